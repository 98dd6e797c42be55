"""Diagnostic (build: python -m parakeet_slam_amd.build --variant rounds -DPK_STAMPS -DPK_DIAG_ROUNDS): how full the verdict rounds of
k_step_pub_big are, step by step along the bench trajectory -- per call of pub_keysN (one wave, 128 landmarks): rounds in which some lane
computes a key, how many of them beyond the first, and how many (landmark, blob) pairs those later rounds really hold."""
import ctypes, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from parakeet_slam_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "parakeet_slam_amd", "libpk_rounds.so")
P, L, S = int(os.environ.get("ST_P", 20480)), int(os.environ.get("ST_L", 5000)), int(os.environ.get("ST_S", 50))
means, covs, scans = bench.synthetic_inputs(L, S + 2)
ws = bench.synthetic_controls(S + 2)
f = _lib.DeviceFilter(P, L)
f.upload_map(means, covs.reshape(L, 25))
so = _lib.load()
so.pk_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 128)()
rnd = random.Random(7)
for s in range(S):
    f.synchronize(); so.pk_debug_stamps(buf, 1)
    f.step(0.2, ws[s], 0.1, scans[s], rnd.random(), seed=7, draw=s, domain=1)
    f.synchronize(); so.pk_debug_stamps(buf, 1)
    calls, heavy, later, pairs = buf[48 + 12], buf[48 + 13], buf[48 + 14], buf[48 + 15]
    if calls and (s < 8 or s % 4 == 0):
        print("step %2d  per wave and turn: %.2f rounds with arithmetic, %.2f of them beyond the first, holding %.1f (landmark, blob) pairs in all (of 128 per round)  route %s"
              % (s, heavy / calls, later / calls, pairs / calls, f.observe_route()))
