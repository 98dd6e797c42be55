"""Diagnostic: what the growing mode does at scale, step by step (bench.py's refscene.new_landmarks_at_scale scene).
   python scripts/gpu_diag_grow_scale.py [P] [L0] [U] [spare] [steps]"""
import math
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import parakeet_slam_amd as pk  # noqa: E402

P, L0, U, spare, steps = [int(a) for a in sys.argv[1:6]] if len(sys.argv) >= 6 else (20000, 2000, 6, 16, 14)
general = os.environ.get("GROW_GENERAL") == "1"


class Scan(object):
    pass


class Node(object):
    pass


means, covs, _ = bench.synthetic_inputs(L0 + U, 1)
feats = [pk.Feature(mean=means[l], covar=covs[l]) for l in range(L0)]
pk.msgs.Time.set_now(0.0)
random.seed(7)
fs = pk.FastSLAM(feats, num_particles=P, device=0, weight_domain="log", rng="device", seed=7, new_landmarks=True, spare_landmarks=spare, publish_debug=False)
if general:
    fs._filter.set_option("fast_observe", 0)
node = Node()
node.last_sensor_reading = Scan()
tw = pk.msgs.Twist()
tw.linear.x, tw.angular.z = 0.5, 0.2
fs.last_control = tw
pose = (0.0, 0.0, 0.0)
prev = np.zeros((P, 4))
for s_ in range(steps):
    h1 = pose[2] + 0.2 * 0.1
    pose = (pose[0] + 0.1 * math.cos(h1), pose[1] + 0.1 * math.sin(h1), pose[2] + 0.04)
    pk.msgs.Time.set_now(0.2 * (s_ + 1))
    b = np.empty((L0 + U, 4))
    b[:, 0] = np.arctan2(means[:, 1] - pose[1], means[:, 0] - pose[0]) - pose[2]
    b[:, 1:] = means[:, 2:]
    node.last_sensor_reading.observes = b
    t0 = time.perf_counter()
    fs.cam_cb(node)
    fs.summary()
    dt = time.perf_counter() - t0
    c = fs._filter.grow_download(readings=False, slot_ids=False)[0].astype(np.int64)
    fl = fs._filter.observe_flagged()
    st = fs._filter.observe_pub_stats() if hasattr(fs._filter, "observe_pub_stats") else None
    print("step %2d  %.2f ms  route %s published %s flagged %s | readings stored mean %.2f max %d | spare used mean %.2f max %d | next_id mean %.1f | dropped %d | stats %s"
          % (s_, dt * 1e3, fs._filter.observe_route(), fs._filter.observe_published(), fl, c[:, 0].mean(), c[:, 0].max(), c[:, 1].mean(), c[:, 1].max(),
             c[:, 2].mean(), c[:, 3].sum(), st), flush=True)
fs.close()
