"""Diagnostic: k_step_pub against the general kernels on a random world of tests/test_gpu_random_worlds.py."""
import os, sys, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parakeet_slam_amd import _lib
import test_gpu_random_worlds as T
seed = int(os.environ.get("DG_SEED", 1022))
case = T.random_case(seed)
L, P, means, covs, immutable, poses, blobs, qt = case
print("L", L, "P", P, "B", len(blobs))
def run(opts):
    f = _lib.DeviceFilter(P, L)
    for k, v in opts.items(): f.set_option(k, v)
    f.set_measurement_noise(qt)
    f.upload_map(means, covs.reshape(L, 25), immutable); f.upload_poses(poses)
    ids = None
    if opts.get("fast_observe", 1) == 0:
        ids = f.observe(blobs, return_ids=True)
    else:
        f.observe(blobs)
    out = (f.download_log_weights(), f.download_landmarks(), f.observe_route(), f.observe_flagged(), f.observe_published(), ids)
    f.close(); return out
pub = run({}); regs = run({"pub_step": 0}); gen = run({"fast_observe": 0})
print("routes", pub[2], pub[3], pub[4], "|", regs[2], regs[3], regs[4], "|", gen[2])
print("logw diff pub-gen:", pub[0] - gen[0]); print("logw diff regs-gen:", regs[0] - gen[0])
ids = gen[5]
for name, o in (("pub", pub), ("regs", regs)):
    m, c, k = o[1]; gm, gc, gk = gen[1]
    print(name, "means close", np.allclose(m, gm, rtol=1e-11, atol=1e-13), "counts equal", np.array_equal(k, gk), "count diffs", int((k != gk).sum()))
    if not np.array_equal(k, gk):
        bad = np.argwhere(k != gk)[:10]
        for a, b in bad:
            print("  particle", int(a), "landmark", int(b) + 1, "count", int(k[a, b]), "want", int(gk[a, b]), "blobs matched to it (general ids):", np.nonzero(ids[a] == b + 1)[0])
