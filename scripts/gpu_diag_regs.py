"""Diagnostic: k_step_regs against the two-sweep route on one small case; where do they differ?  PK_BENCH_LIB picks the build."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parakeet_slam_amd import _lib
if os.environ.get("PK_BENCH_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "parakeet_slam_amd", os.environ["PK_BENCH_LIB"])
import bench
L, P = int(os.environ.get("ST_L", 513)), int(os.environ.get("ST_P", 4))
means, covs, scans = bench.synthetic_inputs(L, 2)
out = {}
for name, opts in (("regs", {}), ("sweep", {"regs_step": 0})):
    f = _lib.DeviceFilter(P, L)
    for k, v in opts.items():
        f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25))
    f.reset_weights()
    f.motion(0.2, 0.1, 0.1, seed=3, draw=0)
    f.observe(scans[0])
    out[name] = (f.observe_route(), f.observe_flagged(), f.download_log_weights(), f.download_landmarks())
    f.close()
a, b = out["regs"], out["sweep"]
print("routes", a[0], b[0], "flagged", a[1], b[1])
print("logw regs ", a[2])
print("logw sweep", b[2])
m1, c1, k1 = a[3]
m2, c2, k2 = b[3]
bad = ~np.isclose(m1, m2, rtol=1e-9, atol=1e-9, equal_nan=False)
print("mean entries that differ:", bad.sum(), "landmarks:", np.unique(np.nonzero(bad)[1])[:20], "fields", np.unique(np.nonzero(bad)[2]))
badc = ~np.isclose(c1, c2, rtol=1e-9, atol=1e-12)
print("cov entries that differ:", badc.sum(), "landmarks:", np.unique(np.nonzero(badc)[1])[:20])
print("count differs:", (k1 != k2).sum())
if bad.sum():
    p, l = np.nonzero(bad)[0][0], np.nonzero(bad)[1][0]
    print("first:", p, l, m1[p, l], m2[p, l])
