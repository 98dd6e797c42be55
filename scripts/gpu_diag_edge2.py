"""Diagnostic: plain ring scenes at several map sizes / seeds -- how many particles does k_step_pub_big flag?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parakeet_slam_amd import _lib as lib
from oracle.fastslam_oracle import synthetic_scan, synthetic_world
from test_gpu_pub import run
for L in (2300, 2304, 2305, 2432, 2560, 3000, 4096, 5000):
    for seed in (123, 5, 77):
        means, covs = synthetic_world(L, seed)
        blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
        poses = np.zeros((4, 4)); poses[:, 3] = 1.0
        out = run(lib, means, covs, poses, blobs)
        # longest candidate list by brute force (widened gates are a little wider than these)
        print("L", L, "seed", seed, "route", out["route"], "published", out["published"], "flagged", out["flagged"], flush=True)
