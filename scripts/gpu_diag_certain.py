"""Diagnostic (build: python -m parakeet_slam_amd.build --variant cert -DPK_STAMPS -DPK_DIAG_BIG_CERTAIN): in k_step_pub_big, how many
wave.pairs (64 lanes x 2 landmarks) hold only landmarks whose gate-passing blobs no other landmark lists -- the ones a 'finish in
pass 1' variant could update without the verdicts -- step by step along the bench trajectory."""
import ctypes, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from parakeet_slam_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "parakeet_slam_amd", "libpk_cert.so")
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
    a, b, c = buf[48 + 12], buf[48 + 13], buf[48 + 14]
    if b and (s < 8 or s % 4 == 0):
        print("step %2d  wave.pairs all certain %5.1f %%   lanes (pairs of landmarks) certain %5.1f %%   route %s" % (s, 100.0 * a / b, 100.0 * c / (64.0 * b), f.observe_route()))
