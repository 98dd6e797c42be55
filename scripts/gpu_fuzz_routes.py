"""Random maps, scans and particle clouds through the production routes (k_step_pub,
k_step_pub_big) against the general kernels -- bit-identical maps, log-weights to 1e-11.  As a script: FUZZ_N scenes, FUZZ_SEED (the
long runs whose logs are in profiles/r04/); a short run is part of the suite (tests/test_gpu_fuzz.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parakeet_slam_amd import _lib as lib
from oracle.fastslam_oracle import synthetic_scan, synthetic_world
from test_gpu_pub import run, poses_around

# options of the production run from the environment: FUZZ_OPTS="pub_duo=1,pub_small=1" (round 6: the fuzzers on the optional instances)
FUZZ_OPTS = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("FUZZ_OPTS", "").split(",") if kv}


def fuzz_routes(N, seed0, lib=lib):
    bad = 0; routes = {}; flagged_all = 0; t0 = time.time()
    for k in range(N):
        rs = np.random.RandomState(seed0 * 1000 + k)
        big = rs.uniform() < 0.4
        L = int(rs.randint(2049, 5600)) if big else int(rs.randint(513, 2049))
        if os.environ.get("FUZZ_SMALL"):  # maps of at most 512 landmarks (with FUZZ_OPTS=pub_small=1: the 256-lane publish / subscribe instance)
            L = int(rs.randint(40, 513))
        P = int(rs.randint(2, 7))
        means, covs = synthetic_world(L, seed=int(rs.randint(1, 10**6)))
        tight = rs.uniform() < 0.7
        if tight:
            covs[:, 2:, 2:] = rs.choice([0.01, 0.04, 0.0025]) * np.identity(3)
            covs[:, :2, :2] = rs.choice([0.25, 0.05, 1.0]) * np.identity(2)
        n = len(means[3::7])
        if rs.uniform() < 0.7:
            means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))  # look-alikes: contested blobs
        imm = (rs.uniform(size=L) < rs.choice([0.0, 0.1])).astype(np.uint8)
        pose = (rs.normal(0, 0.05), rs.normal(0, 0.05), rs.normal(0, 0.02))
        blobs = synthetic_scan(means, pose)
        blobs[:, 0] += rs.normal(0, 0.002, L); blobs[:, 1:] += rs.normal(0, 0.3, (L, 3))
        blobs = blobs[rs.permutation(L)]
        if rs.uniform() < 0.5:
            blobs = blobs[: int(L * rs.uniform(0.5, 1.0))]
        if L > 5000:
            blobs = blobs[:3500]
        poses = poses_around(rs, P, rs.choice([0.02, 0.05, 0.2]))
        a = run(lib, means, covs, poses, blobs, dict(FUZZ_OPTS), immutable=imm)
        g = run(lib, means, covs, poses, blobs, {"fast_observe": 0}, immutable=imm)
        routes[a["route"]] = routes.get(a["route"], 0) + 1
        ok = np.allclose(a["logw"], g["logw"], rtol=1e-11, atol=1e-9) and all(np.array_equal(x, y) for x, y in zip(a["maps"], g["maps"]))
        flagged_all += int(a["flagged"] == P)
        if not ok:
            bad += 1
            print("MISMATCH scene", k, "L", L, "P", P, "B", len(blobs), "route", a["route"], "flagged", a["flagged"], "tight", tight, flush=True)
    print("scenes", N, "mismatches", bad, "routes", routes, "scenes with every particle flagged", flagged_all, "seconds %.0f" % (time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if fuzz_routes(int(os.environ.get("FUZZ_N", 120)), int(os.environ.get("FUZZ_SEED", 1))) else 0)
