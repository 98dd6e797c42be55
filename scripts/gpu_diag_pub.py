"""Diagnostic: k_step_pub against k_step_regs / the general kernels on one scan (test_sweep_observe_large_maps' world)."""
import os, sys, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parakeet_slam_amd import _lib
from oracle.fastslam_oracle import OracleFilter, synthetic_scan, synthetic_world
L = int(os.environ.get("DG_L", 513)); P = int(os.environ.get("DG_P", 24))
rs = np.random.RandomState(200 + L)
means, covs = synthetic_world(L)
n = len(means[3::7])
means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))
blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
poses = np.zeros((P, 4)); poses[:, 0] = rs.uniform(-1, 1, P); poses[:, 1] = rs.uniform(-1, 1, P); poses[:, 2] = rs.uniform(-0.1, 0.1, P); poses[:, 3] = 1.0
def run(opts):
    f = _lib.DeviceFilter(P, L)
    for k, v in opts.items(): f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25)); f.upload_poses(poses)
    f.observe(blobs)
    out = (f.download_poses(), f.download_landmarks(), f.observe_route(), f.observe_flagged(), f.observe_published())
    f.close(); return out
pub = run({}); regs = run({"pub_step": 0}); gen = run({"fast_observe": 0})
print("routes", pub[2], pub[3], pub[4], "|", regs[2], regs[3], regs[4], "|", gen[2])
lw = lambda o: np.log(o[0][:, 3])
print("logw diff pub-gen in units of log 0.1:", np.round((lw(pub) - lw(gen)) / math.log(0.1), 3))
print("logw diff regs-gen:", np.round((lw(regs) - lw(gen)) / math.log(0.1), 6))
for name, o in (("pub", pub), ("regs", regs)):
    m, c, k = o[1]; gm, gc, gk = gen[1]
    print(name, "means equal", np.array_equal(m, gm), "cov equal", np.array_equal(c, gc), "counts equal", np.array_equal(k, gk), "count diffs", int((k != gk).sum()))
    if not np.array_equal(k, gk):
        bad = np.argwhere(k != gk)[:10]; print(" first count diffs (particle, landmark, got, want):", [(int(a), int(b), int(k[a, b]), int(gk[a, b])) for a, b in bad])
