"""Diagnostic: who flags the particles of the underflow-edge scene at L = 2 304 (k_step_pub_big)?"""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parakeet_slam_amd as pk
from parakeet_slam_amd import _lib as lib
from oracle.fastslam_oracle import synthetic_scan, synthetic_world
from test_gpu_pub import run
L = int(os.environ.get("DL", 2304))
for target, shift in ((1470.0, True), (1470.0, False), (None, False), (1300.0, True), (1000.0, True)):
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = np.zeros((4, 4)); poses[:, 3] = 1.0
    d2 = 200.0
    lm = 17
    for lm in range(17, L):
        shifted = means[lm, 2:] + np.array([math.sqrt(d2), 0.0, 0.0])
        others = np.delete(np.arange(L), lm)
        if np.min(np.sum((means[others, 2:] - shifted) ** 2, axis=1)) > 500.0:
            break
    if target:
        kconst = 5.0 * math.log(2.0 * math.pi) + math.log(0.25 * 0.25)
        lo, hi = 1e-3, 0.25
        for _ in range(200):
            c = 0.5 * (lo + hi)
            key = kconst + 3.0 * math.log(c) + d2 / c
            lo, hi = (c, hi) if key > target else (lo, c)
        covs[lm, 2:, 2:] = np.identity(3) * c
    if shift:
        blobs[lm, 1] += math.sqrt(d2)
    out = run(lib, means, covs, poses, blobs)
    print("target", target, "shift", shift, "lm", lm, "route", out["route"], "published", out["published"], "flagged", out["flagged"])
