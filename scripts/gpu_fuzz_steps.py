"""Whole filter steps (motion, association + EKF, log-weights, resampling) on random
maps through the production routes against the general kernels: ancestors and maps identical step after step.  As a script: FUZZ_N
runs, FUZZ_SEED; a short run is part of the suite (tests/test_gpu_fuzz.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parakeet_slam_amd import _lib as lib
from oracle.fastslam_oracle import synthetic_scan, synthetic_world, truth_step

# options of the production run from the environment: FUZZ_OPTS="pub_duo=1,pub_small=1" (round 6: the fuzzers on the optional instances)
FUZZ_OPTS = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("FUZZ_OPTS", "").split(",") if kv}


def fuzz_steps(N, seed0, lib=lib):
    bad = 0; routes = {}; t0 = time.time()
    for k in range(N):
        rs = np.random.RandomState(seed0 * 1000 + k)
        big = rs.uniform() < 0.4
        L = int(rs.randint(2049, 5000)) if big else int(rs.randint(513, 2049))
        if os.environ.get("FUZZ_SMALL"):  # maps of at most 512 landmarks (with FUZZ_OPTS=pub_small=1: the 256-lane publish / subscribe instance)
            L = int(rs.randint(40, 513))
        P = int(rs.choice([64, 200, 512, 1000]))
        steps = int(rs.randint(3, 7))
        means, covs = synthetic_world(L, seed=int(rs.randint(1, 10**6)))
        covs[:, 2:, 2:] = rs.choice([0.25, 0.04, 0.01]) * np.identity(3)
        imm = (rs.uniform(size=L) < rs.choice([0.0, 0.1])).astype(np.uint8)
        outs = []
        for opts in (dict(FUZZ_OPTS), {"fast_observe": 0}):
            f = lib.DeviceFilter(P, L)
            for kk, v in opts.items():
                f.set_option(kk, v)
            f.upload_map(means, covs.reshape(L, 25), imm)
            pose, rec = (0.0, 0.0, 0.0), []
            rs2 = np.random.RandomState(k)
            for s in range(steps):
                pose = truth_step(pose, 0.2, 0.1, 0.1)
                scan = synthetic_scan(means, pose)
                scan[:, 0] += rs2.normal(0, 0.002, L); scan[:, 1:] += rs2.normal(0, 0.3, (L, 3))
                f.step(0.2, 0.1, 0.1, scan[rs2.permutation(L)], 0.1 + 0.13 * s, seed=11 + k, draw=s, domain=lib.PK_WEIGHTS_LOG)
                rec.append((f.download_poses().copy(), [x.copy() for x in f.download_landmarks()]))
            outs.append((rec, f.observe_route()))
            f.close()
        routes[outs[0][1]] = routes.get(outs[0][1], 0) + 1
        ok = True
        for (pa, ma), (pb, mb) in zip(outs[0][0], outs[1][0]):
            ok &= np.array_equal(pa[:, :3], pb[:, :3]) and np.allclose(pa[:, 3], pb[:, 3], rtol=1e-9, atol=1e-12)
            ok &= all(np.array_equal(x, y) for x, y in zip(ma, mb))
        if not ok:
            bad += 1
            print("MISMATCH scene", k, "L", L, "P", P, "steps", steps, "route", outs[0][1], flush=True)
    print("scenes", N, "mismatches", bad, "routes", routes, "seconds %.0f" % (time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if fuzz_steps(int(os.environ.get("FUZZ_N", 120)), int(os.environ.get("FUZZ_SEED", 1))) else 0)
