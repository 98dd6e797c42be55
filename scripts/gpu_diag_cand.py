"""Diagnostic: spread of the particles' expected bearings / colours around particle 0 (the reference of k_candidates)."""
import os, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from parakeet_slam_amd import _lib
P, L, S = int(os.environ.get("ST_P", 20000)), int(os.environ.get("ST_L", 2000)), int(os.environ.get("ST_S", 25))
means, covs, scans = bench.synthetic_inputs(L, S + 2)
ws = bench.synthetic_controls(S + 2)
f = _lib.DeviceFilter(P, L)
f.upload_map(means, covs.reshape(L, 25))
rnd = random.Random(7)
for s in range(S):
    f.step(0.2, ws[s], 0.1, scans[s], rnd.random(), seed=7, draw=s, domain=1)
    if s % 6 == 5 or s == S - 1:
        print("step", s, "flagged/over", f.observe_flagged(), "unique sources", np.unique(f.download_sources()).size)
# one more motion, then look at the generation the next observe would see
f.motion(0.2, ws[S], 0.1, seed=7, draw=S)
poses = f.download_poses()
print("pose std x y h:", poses[:, 0].std(), poses[:, 1].std(), poses[:, 2].std(), " h range", poses[:, 2].min(), poses[:, 2].max())
n = 256
m, c, k = f.download_landmarks(0, n)
eb = np.arctan2(m[:, :, 1] - poses[:n, 1:2], m[:, :, 0] - poses[:n, 0:1]) - poses[:n, 2:3]
d = eb - eb[0:1]
d = (d + np.pi) % (2 * np.pi) - np.pi
print("max |eb - eb0| per particle: median %.4f  90%% %.4f  max %.4f" % (np.median(np.abs(d).max(1)), np.percentile(np.abs(d).max(1), 90), np.abs(d).max()))
print("landmark xy spread (std over particles, max over landmarks):", m[:, :, 0].std(0).max(), m[:, :, 1].std(0).max())
dc = np.abs(m[:, :, 2:] - m[0:1, :, 2:]).max()
print("max colour deviation from particle 0:", dc)
worst = np.unravel_index(np.abs(d).argmax(), d.shape)
print("worst particle, landmark:", worst, "dist", np.hypot(m[worst[0], worst[1], 0], m[worst[0], worst[1], 1]), "xy", m[worst[0], worst[1], :2], "ref xy", m[0, worst[1], :2], "counts", k[worst[0], worst[1]], k[0, worst[1]])
