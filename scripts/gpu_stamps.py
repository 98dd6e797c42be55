"""Diagnostic: per-phase cycle shares of k_assoc_grid / k_step_fused from the -DPK_STAMPS build
(python -m parakeet_slam_amd.build --stamps)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parakeet_slam_amd import _lib, build as _build
# the stamps library is rebuilt whenever a source is newer than it (a stale one used to leave a traceback in profiles/)
_so = os.path.join(ROOT, "parakeet_slam_amd", "libparakeet_slam_stamps.so")
_srcs = [os.path.join(_build.CSRC, x) for x in os.listdir(_build.CSRC) if x.endswith((".hip", ".hpp", ".cpp"))] + [os.path.join(ROOT, "include", "parakeet_slam.h")]
if not os.path.exists(_so) or any(os.path.getmtime(x) > os.path.getmtime(_so) for x in _srcs):
    try:
        _build.build_stamps(verbose=False)
    except Exception as e:  # noqa: BLE001
        print("gpu_stamps: cannot build the stamps library: %s" % e, file=sys.stderr)
        sys.exit(2)
_lib.LIB_PATH = os.path.join(ROOT, "parakeet_slam_amd", "libparakeet_slam_stamps.so")
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
P = int(os.environ.get("ST_P", 10000)); L = int(os.environ.get("ST_L", 500))
WARM = int(os.environ.get("ST_WARM", 3))  # steps before the measured three (the far pruning of round 5 needs a map seen a few times)
means, covs, scans = bench.synthetic_inputs(L, WARM + 3)
f = _lib.DeviceFilter(P, L)
f.upload_map(means, covs.reshape(L, 25))
print("options from the environment:", bench.env_options(f))
so = _lib.load()
so.pk_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()
for s in range(WARM):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, domain=1)
f.synchronize(); so.pk_debug_stamps(buf, 1)
wbuf = (ctypes.c_ulonglong * 96)()
if hasattr(so, "pk_debug_pub_wave_stamps"):
    so.pk_debug_pub_wave_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    so.pk_debug_pub_wave_stamps(wbuf, 1)
for s in range(WARM, WARM + 3):
    f.step(0.2, 0.1, 0.1, scans[s], 0.3, seed=7, draw=s, domain=1)
f.synchronize(); so.pk_debug_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64)
print("steps before the measured three:", WARM)
print("git", os.environ.get("PK_GIT_SHA") or os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % ROOT).read().strip() or "unknown", "P", P, "L", L)
if L > 2048 and hasattr(so, "pk_debug_pub_wave_stamps"):  # k_step_pub_big ran (its stamps are kept per wave only)
    so.pk_debug_pub_wave_stamps(wbuf, 1)
    w = np.array(list(wbuf), dtype=np.float64).reshape(8, 12)
    if w[:, 9].sum() > 0:
        bn = ["pass 1: waits for records and rows", "pass 1: gates", "pass 1: verdicts", "pass 1: next rows asked for", "barrier A",
              "settling, barrier B", "take, barrier C", "pass 2: waits for rows", "pass 2: updates", None,
              "pass 2: stores, next rows asked for", "particle"]
        tot = w.sum(axis=0)
        for i, n in enumerate(bn):
            if n:
                print("big %-40s %12.4g  %5.1f%%" % (n, tot[i], 100 * tot[i] / tot[11]))
        print("big cycles per particle and wave: %.0f" % (tot[11] / max(tot[9], 1.0)))
        print("per wave, cycles per particle: p1 wait | gates | verdicts | ask | A | settle+B | take+C | p2 wait | updates | stores+ask")
        for i in range(8):
            n = max(w[i, 9], 1.0)
            print("  wave %d  " % i + " | ".join("%6.0f" % (w[i, k] / n) for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10)))
        sys.exit(0)
if v[48 + 8] > 0:  # k_step_pub ran (512 < L <= 2048, publish table in LDS)
    pn = ["scalars, requests", "gates (waits for candidate records, means)", "  of which: keys of the gate-passing blobs", "barrier A",
          "unseen blobs, subscribe", "barrier B", "updates, stores issued", "wave sum", "particle (wave lifetime)"]
    life = v[48 + 8]
    for i, n in enumerate(pn):
        print("pub %-44s %12.4g  %5.1f%%" % (n, v[48 + i], 100 * v[48 + i] / life))
    print("pub %-44s %12.4g  %5.1f%%" % ("  of the updates: stores + next rows issued", v[48 + 10], 100 * v[48 + 10] / life))
    print("pub cycles per particle and wave: %.0f" % (life / max(v[48 + 9], 1.0)))
    if hasattr(so, "pk_debug_pub_wave_stamps"):
        so.pk_debug_pub_wave_stamps(wbuf, 1)
        w = np.array(list(wbuf), dtype=np.float64).reshape(8, 12)
        print("per wave of the workgroup, cycles per particle: requests | gates + keys (of which keys) | wait at barrier A | settling | B..C | updates + stores (of which stores + next rows issued)")
        for i in range(8):
            n = max(w[i, 9], 1.0)
            print("  wave %d  %6.0f | %6.0f (%6.0f) | %6.0f | %6.0f | %6.0f | %6.0f (%6.0f)" % (i, w[i, 0] / n, w[i, 1] / n, w[i, 2] / n, w[i, 3] / n, w[i, 4] / n, w[i, 5] / n, w[i, 6] / n, w[i, 10] / n))
    sys.exit(0)
if v[32 + 14] > 0:  # k_step_regs ran (512 < L <= 2048)
    rn = ["scalars, requests, zeroing", "barrier", "gate arguments (scalar loads)", "gates 1st landmark (waits for means)",
          "gates 2nd landmark", "barrier behind the gates", "warming, unseen blobs", "round 1: prepare (waits for cov rows)",
          "round 1: queue evaluation", "round 2: prepare", "round 2: queue evaluation", "bids: barriers, win, collect",
          "updates + stores issued", "block sum + tail", "particle", "-"]
    life = v[32 + 14]
    for i, n in enumerate(rn[:15]):
        print("regs %-40s %12.4g  %5.1f%%" % (n, v[32 + i], 100 * v[32 + i] / life))
    print("regs queued probabilities per particle: %.1f" % (v[32 + 15] / (3.0 * P)))
    sys.exit(0)
if v[16 + 13] > 0:  # k_step_fused ran (L <= 512)
    fn = ["tables->LDS + barrier", "wait for the means", "atan2, cell, walk", "exact gates", "barrier after gates",
          "counts + barrier", "prepare (settling)", "barrier after prepare", "queue evaluation + barrier",
          "collect + barrier", "apply (EKF)", "stores issued", "block sum", "lifetime",
          "  (of the first:) scalar prologue", "  (of the first:) table words arrived"]
    life = v[16 + 13]
    for i, n in enumerate(fn):
        print("fused %-28s %12.4g  %5.1f%%" % (n, v[16 + i], 100 * v[16 + i] / life))
    sys.exit(0)
names = ["init+sync", "S1 atan2+cell", "S1 phase1 walk", "S1 phase2 exact", "S1 total", "S1 barrier wait", "S2", "S3", "S4", "writeout"]
tot = v[0] + v[4] + v[5] + v[6] + v[7] + v[8] + v[9]
for i, n in enumerate(names):
    print("%-18s %12.4g  %5.1f%%" % (n, v[i], 100 * v[i] / tot))

