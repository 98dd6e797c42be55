"""GPU: section 8(f4) on the device -- potential features (prkt_core_v2.py:109-118: the reference's negative ids).
A landmark whose count word carries PK_LANDMARK_POTENTIAL is matched and updated like any other, but a match weighs
0.1 (`no_match_weight`) instead of the importance factor, and the flag falls when its update count passes 5.  The rule
sits in the one EKF device function, so every route must show it: the one-pass kernels, the hand-off routes, the general
kernels, supplied ids, the dense layout."""
import numpy as np
import pytest

from oracle.fastslam_oracle import OracleFilter, synthetic_scan, synthetic_world, truth_step

pytestmark = pytest.mark.gpu


def run(lib, P, L, opts, dense=False, ids=None, steps=2, seed=3):
    rs = np.random.RandomState(seed)
    means, covs = synthetic_world(L)
    if dense:  # position-colour coupling: the dense layout and its kernels
        covs = covs.copy()
        covs[:, 0, 2] = covs[:, 2, 0] = 0.01
    pot = rs.uniform(size=(P, L)) < 0.3
    cnt = rs.randint(0, 4, size=(P, L)) * 2
    f = lib.DeviceFilter(P, L)
    for k, v in opts.items():
        f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25))
    m, c, _ = f.download_landmarks()
    f.upload_landmarks(0, P, m, c, (cnt | np.where(pot, lib.PK_LANDMARK_POTENTIAL, 0)).astype(np.int32))
    o = OracleFilter(P, means, covs)
    o.count[:] = cnt
    o.potential[:] = pot
    pose = (0.0, 0.0, 0.0)
    z = rs.normal(size=(steps, P, 3))
    for s in range(steps):
        pose = truth_step(pose, 0.2, 0.1, 0.1)
        blobs = synthetic_scan(means, pose)
        use = None if ids is None else np.arange(1, L + 1)
        f.reset_weights()
        f.motion(0.2, 0.1, 0.1, z=z[s])
        f.observe(blobs, ids=use)
        o.reset_weights()
        o.motion(0.2, 0.1, 0.1, z[s])
        o.observe(blobs, use)
        lw = f.download_log_weights()
        assert np.allclose(lw, o.logw, rtol=1e-9, atol=1e-9), (s, f.observe_route())
        left_now = int(o.potential.sum())  # before the resample thins the cloud out
        u = float(rs.uniform())
        anc = f.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True)
        o.gather(anc)
    gm, gc, gk = f.download_landmarks()
    route = f.observe_route()
    f.close()
    assert np.array_equal(gk & ~lib.PK_LANDMARK_POTENTIAL, o.count), route
    assert np.array_equal((gk & lib.PK_LANDMARK_POTENTIAL) != 0, o.potential), route
    assert np.allclose(gm, o.mean, rtol=1e-9, atol=1e-11)
    return route, left_now, int(pot.sum())


@pytest.mark.parametrize("L,opts,route", [(40, {}, "ml_fused"), (40, {"fused_step": 0}, "ml_handoff"), (40, {"fast_observe": 0}, "ml_general"),
                                          (700, {}, "ml_regs"), (700, {"regs_step": 0}, "ml_sweep"), (700, {"owner_step": 1}, "ml_owner")])
def test_potential_features_weigh_a_tenth_and_are_promoted(lib, L, opts, route):
    got, left, start = run(lib, 48, L, opts)
    assert got == route
    assert 0 < left < start  # some promoted (count passed 5), some still potential


def test_potential_features_with_supplied_ids_and_on_the_dense_layout(lib):
    got, left, start = run(lib, 32, 60, {}, ids=True)
    assert got == "known_ids" and 0 < left < start
    got, left, start = run(lib, 16, 24, {}, dense=True)
    assert got == "dense" and 0 < left < start
