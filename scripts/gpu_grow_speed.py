"""Section 8(f4) at device speed: ms per cam_cb of the facade in the growing mode, the per-particle bookkeeping on the device
(pk_k_grow.hip, the default) against the host loop (bookkeeping="host"), same scene, same seeds.  Prints one JSON line.
    python scripts/gpu_grow_speed.py [P] [L0] [U] [steps]"""
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import parakeet_slam_amd as pk  # noqa: E402


def ring_world(L, seed=123):
    rs = np.random.RandomState(seed)
    phi = -np.pi + 2 * np.pi * np.arange(L) / float(L) + 0.01
    rho = rs.uniform(8.0, 30.0, size=L)
    m = np.empty((L, 5))
    m[:, 0], m[:, 1] = rho * np.cos(phi), rho * np.sin(phi)
    m[:, 2:] = rs.uniform(0.0, 255.0, size=(L, 3))
    return m


def scan(world, pose):
    x, y, h = pose
    b = np.empty((world.shape[0], 4))
    b[:, 0] = np.arctan2(world[:, 1] - y, world[:, 0] - x) - h
    b[:, 1:] = world[:, 2:]
    return b


class View(object):
    def __init__(self, blobs):
        class Scan(object):
            pass

        self.last_sensor_reading = Scan()
        self.last_sensor_reading.observes = blobs  # (B, 4) array: the facade's fast path


def run(P, L0, U, steps, bookkeeping):
    world = ring_world(L0 + U)
    feats = [pk.Feature(mean=world[l].copy(), covar=0.25 * np.identity(5)) for l in range(L0)]
    random.seed(3)
    pk.msgs.Time.set_now(0.0)
    fs = pk.FastSLAM(feats, num_particles=P, weight_domain="log", rng="device", seed=3, new_landmarks=True, spare_landmarks=U + 2,
                     bookkeeping=bookkeeping, publish_debug=False)
    tw = pk.msgs.Twist()
    tw.linear.x, tw.angular.z = 0.5, 0.2
    fs.last_control = tw
    pose, t = (0.0, 0.0, 0.0), []
    for s in range(steps):
        h1 = pose[2] + 0.2 * 0.1
        pose = (pose[0] + 0.1 * np.cos(h1), pose[1] + 0.1 * np.sin(h1), pose[2] + 0.04)  # v = 0.5, w = 0.2, dt = 0.2, no noise
        pk.msgs.Time.set_now(0.2 * (s + 1))
        v = View(scan(world, pose))
        t0 = time.perf_counter()
        fs.cam_cb(v)
        fs.summary()
        t.append((time.perf_counter() - t0) * 1e3)
    used = np.asarray(fs._used)
    k = fs._filter.download_landmarks(means=False, covs=False)[2][:, L0:]
    out = dict(ms_per_step=[round(x, 3) for x in t], median_ms=round(float(np.median(t[1:])), 3), spare_slots_in_use_mean=round(float(used.mean()), 2),
               promoted_per_particle=round(float((((k & 0x40000000) == 0) & (np.arange(k.shape[1])[None, :] < used[:, None])).sum() / float(P)), 2),
               readings_stored_mean=round(float(np.mean([len(h) for h in fs._hyp])), 2), readings_dropped=fs.readings_dropped())
    fs.close()
    return out


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L0 = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    U = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    res = dict(particles=P, preset_landmarks=L0, unknown_landmarks=U, steps=steps, device=run(P, L0, U, steps, "device"))
    res["host"] = run(P, L0, U, steps, "host")
    res["speedup"] = round(res["host"]["median_ms"] / res["device"]["median_ms"], 1)
    print(json.dumps(res))
