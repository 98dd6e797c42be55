import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parakeet_slam_amd import _lib
import test_gpu_pub as T
from oracle.fastslam_oracle import synthetic_scan, synthetic_world
for L, P in [(700, 6), (1024, 4), (1026, 4), (2000, 3), (2048, 3)]:
    rs = np.random.RandomState(900 + L)
    means, covs = synthetic_world(L)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))
    imm = (rs.uniform(size=L) < 0.1).astype(np.uint8)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))[rs.permutation(L)]
    poses = T.poses_around(rs, P, 0.2)
    for opts in ({}, {"pub_step": 0}):
        o = T.run(_lib, means, covs, poses, blobs, opts, immutable=imm)
        f = _lib.DeviceFilter(P, L); [f.set_option(k, v) for k, v in opts.items()]
        f.upload_map(means, covs.reshape(L, 25), imm); f.upload_poses(poses); f.observe(blobs); fl = f.observe_flagged(); f.close()
        print(L, P, opts, "published", o["published"], "flagged/overflow", fl)
    # passers
    mx = 0
    for x, y, h, _ in poses:
        eb = np.arctan2(means[:, 1] - y, means[:, 0] - x) - h
        ok = (np.abs(blobs[:, 0][None, :] - eb[:, None]) <= 0.5) & (((blobs[None, :, 1:] - means[:, None, 2:]) ** 2).sum(-1) <= 300)
        mx = max(mx, int(ok.sum(1).max()))
    print("   max passers", mx)
