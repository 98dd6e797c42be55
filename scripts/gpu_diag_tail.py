"""Diagnostic: the tail scene of tests/test_gpu_pub.py at a few sizes, production route against the general kernels
(PK_TEST_LIB picks the build)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from parakeet_slam_amd import _lib  # noqa: E402

if os.environ.get("PK_TEST_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "parakeet_slam_amd", os.environ["PK_TEST_LIB"])
import test_gpu_pub as T  # noqa: E402

for L, opts in ((2000, {}), (1536, {}), (1535, {}), (1664, {}), (768, {}), (384, {"pub_small": 1}), (5008, {})):
    rs = np.random.RandomState(4000 + L)
    means, covs, blobs = T.crowded_tail_scene(L, rs)
    P = 3
    poses = T.poses_around(rs, P, 0.05)
    pub = T.run(_lib, means, covs, poses, blobs, opts)
    gen = T.run(_lib, means, covs, poses, blobs, {"fast_observe": 0})
    n_last = (gen["ids"] == L).sum(axis=1)
    print(os.environ.get("PK_TEST_LIB", "HEAD"), "L", L, "route", pub["route"], "published", pub["published"], "flagged", pub["flagged"],
          "blobs taken by the last landmark", n_last, "logw pub - gen", pub["logw"] - gen["logw"],
          "maps equal", all(np.array_equal(x, y) for x, y in zip(pub["maps"], gen["maps"])))
