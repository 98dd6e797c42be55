"""Diagnostic: wall time of every step of the bench workload (synchronised per step), with the flagged counts."""
import os, sys, time, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from parakeet_slam_amd import _lib
P, L, S = int(os.environ.get("ST_P", 100000)), int(os.environ.get("ST_L", 2000)), int(os.environ.get("ST_S", 70))
means, covs, scans = bench.synthetic_inputs(L, S + 2)
ws = bench.synthetic_controls(S + 2)
f = _lib.DeviceFilter(P, L)
for name in ("PK_OPT_FAST_OBSERVE", "PK_OPT_REGS_STEP"):
    if os.environ.get(name):
        f.set_option(name[7:].lower(), int(os.environ[name]))
f.upload_map(means, covs.reshape(L, 25))
rnd = random.Random(7)
f.enable_timing(True)
out = []
for s in range(S):
    f.reset_timings()
    f.synchronize()
    t0 = time.perf_counter()
    f.step(0.2, ws[s], 0.1, scans[s], rnd.random(), seed=7, draw=s, domain=1)
    f.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    tm = f.timings()
    out.append((s, dt, f.observe_flagged(), tm))
for s, dt, fl, tm in out:
    if s % 4 == 3 or fl[0] > 0:
        print("step %3d  %.2f ms  flagged %s  spans %s" % (s, dt, fl, {k: (round(v[0], 2) if isinstance(v, tuple) else round(v, 2)) for k, v in tm.items()}))
