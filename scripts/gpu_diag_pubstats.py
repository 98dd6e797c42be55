"""Diagnostic: every step of the bench workload (step 0 included), synchronised per step: wall time, flagged particles, what the
scan's publish table came to (pk_observe_pub_stats) and the spans -- what finds a slow step in the warm-up and says why."""
import os, sys, time, random, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from parakeet_slam_amd import _lib
P, L, S = int(os.environ.get("ST_P", 20000)), int(os.environ.get("ST_L", 5000)), int(os.environ.get("ST_S", 50))
means, covs, scans = bench.synthetic_inputs(L, S + 2)
ws = bench.synthetic_controls(S + 2)
f = _lib.DeviceFilter(P, L)
for name in os.environ.get("ST_OPTS", "").split(","):
    if "=" in name:
        k, v = name.split("=")
        f.set_option(k, int(v))
f.upload_map(means, covs.reshape(L, 25))
rnd = random.Random(7)
f.enable_timing(True)
rows = []
for s in range(S):
    f.reset_timings()
    f.synchronize()
    t0 = time.perf_counter()
    f.step(0.2, ws[s], 0.1, scans[s], rnd.random(), seed=7, draw=s, domain=1)
    f.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    tm = f.timings()
    st = f.observe_pub_stats()
    fl = f.observe_flagged()
    src = np.unique(f.download_sources()).size
    rows.append(dict(step=s, ms=round(dt, 3), flagged=fl[0], cand_overflow=fl[1], distinct_sources=int(src), **st,
                     spans={k: (round(v[0], 3) if isinstance(v, tuple) else round(v, 3)) for k, v in tm.items()}))
for r in rows:
    print(json.dumps(r))
out = os.environ.get("ST_OUT")
if out:
    json.dump(dict(P=P, L=L, steps=rows), open(out, "w"), indent=0)
