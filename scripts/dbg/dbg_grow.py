import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parakeet_slam_amd import _lib as lib
import test_gpu_grow as T
from oracle.fastslam_oracle import synthetic_scan, truth_step
P, L0, U, spare, steps = 12, 700, 3, 5, 6
world, covs, known, kcov = T._scene(L0, U)
fa = T._device_filter(lib, P, known, kcov, spare, 30.0, 64)
fb = T._device_filter(lib, P, known, kcov, spare, 30.0, 64); fb.set_option("fast_observe", 0)
rs = np.random.RandomState(7 + L0)
pose = (0.0, 0.0, 0.0)
for s in range(steps):
    pose = truth_step(pose, 0.8, 0.35, 0.5)
    blobs = synthetic_scan(world, pose)
    z = rs.standard_normal((P, 3))
    fa.motion(0.8, 0.35, 0.5, z=z); fb.motion(0.8, 0.35, 0.5, z=z)
    fa.observe(blobs, fresh=True); ids = fb.observe(blobs, fresh=True, return_ids=True)
    flags = fa.observe_flags()
    ca, ra, sa = fa.grow_download(); cb, rb, sb = fb.grow_download()
    print("step", s, "route", fa.observe_route(), fa.observe_published(), "flagged", fa.observe_flagged(), "flags", flags.tolist(), "stats", fa.observe_pub_stats())
    print("   unmatched per particle (general):", [(np.nonzero(ids[i] == 0)[0]).tolist() for i in range(P)][:6])
    print("   cnt a", ca[:, :3].tolist()[:6]); print("   cnt b", cb[:, :3].tolist()[:6])
    bad = [i for i in range(P) if not (np.array_equal(ca[i], cb[i]) and np.allclose(ra[i], rb[i]))]
    print("   particles that differ:", bad)
    for i in bad[:2]:
        n = max(ca[i, 0], cb[i, 0])
        print("   p", i, "a ids", ra[i, :n, 0].tolist(), "bearing", np.round(ra[i, :n, 4], 4).tolist()); print("   p", i, "b ids", rb[i, :n, 0].tolist(), "bearing", np.round(rb[i, :n, 4], 4).tolist())
    u = float(rs.uniform())
    anc = fa.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True); ancb = fb.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True)
    print("   ancestors equal", np.array_equal(anc, ancb), "logw close", np.allclose(fa.download_log_weights(), fb.download_log_weights()))
