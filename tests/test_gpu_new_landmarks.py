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
                                          (700, {}, "ml_regs"), (700, {"regs_step": 0}, "ml_sweep"), (700, {"pub_step": 0}, "ml_regs"), (2500, {}, "ml_pub_big")])
def test_potential_features_weigh_a_tenth_and_are_promoted(lib, L, opts, route):
    got, left, start = run(lib, 48, L, opts)
    assert got == route
    assert 0 < left < start  # some promoted (count passed 5), some still potential


def test_potential_features_with_supplied_ids_and_on_the_dense_layout(lib):
    got, left, start = run(lib, 32, 60, {}, ids=True)
    assert got == "known_ids" and 0 < left < start
    got, left, start = run(lib, 16, 24, {}, dense=True)
    assert got == "dense" and 0 < left < start


# ---------------------------------------------------------------------------------------------------------------------
# The whole of f4 through the facade: unknown landmarks are triangulated from two sightings (:546-746 with the working pairing
# rule), tracked as potential features (device kernels), promoted past update_count 5 -- against the NumPy GrowingOracle.
class _View(object):
    def __init__(self, pk, blobs):
        class Scan(object):
            pass

        self.last_sensor_reading = Scan()
        obs = []
        for b in blobs:
            z = pk.msgs.Blob()
            z.bearing = float(b[0])
            z.color.r, z.color.g, z.color.b = float(b[1]), float(b[2]), float(b[3])
            obs.append(z)
        self.last_sensor_reading.observes = obs


@pytest.mark.parametrize("bookkeeping", ["device", "host"])
def test_unknown_landmarks_are_triangulated_tracked_and_promoted(lib, bookkeeping):
    import random

    import parakeet_slam_amd as pk
    from oracle.fastslam_oracle import GrowingOracle, low_variance_ancestors

    L0, U, P, spare, steps, thr = 10, 3, 12, 5, 8, 30.0
    v, w, dt = 0.8, 0.35, 0.5
    world, covs = synthetic_world(L0 + U)
    known, kcov = world[:L0], covs[:L0]
    np.random.seed(5)
    random.seed(5)
    pk.msgs.Time.set_now(0.0)
    fs = pk.FastSLAM([pk.Feature(mean=m.copy(), covar=c.copy()) for m, c in zip(known, kcov)], num_particles=P,
                     weight_domain="log", new_landmarks=True, spare_landmarks=spare, pair_threshold=thr, bookkeeping=bookkeeping,
                     record_ids=True)
    assert fs._nl_device == (bookkeeping == "device")
    tw = pk.msgs.Twist()
    tw.linear.x, tw.angular.z = v, w
    fs.last_control = tw
    o = GrowingOracle(P, known, kcov, spare, thr)
    np_rs = np.random.RandomState(5)  # the facade draws from numpy.random / random seeded the same way
    py_rs = random.Random(5)
    pose = (0.0, 0.0, 0.0)
    created = promoted = 0
    for s in range(steps):
        pose = truth_step(pose, v, w, dt)
        blobs = synthetic_scan(world, pose)  # the scan sees the three unknown landmarks too
        pk.msgs.Time.set_now(dt * (s + 1))
        fs.cam_cb(_View(pk, blobs))
        o.f.reset_weights()
        o.f.motion(v, w, dt, np_rs.standard_normal((P, 3)))
        oids = o.observe(blobs)
        assert np.array_equal(np.asarray(fs.last_ids), oids), "step %d: association differs" % s
        wts = np.exp(o.f.logw - o.f.logw.max())
        anc = low_variance_ancestors(wts, py_rs.random())
        assert np.array_equal(np.asarray(fs.last_ancestors), anc), s
        o.gather(anc)
        m, c, k = fs._filter.download_landmarks()
        assert np.array_equal((k & lib.PK_LANDMARK_POTENTIAL) != 0, o.f.potential), s
        assert np.array_equal(k & ~lib.PK_LANDMARK_POTENTIAL, o.f.count), s
        assert np.allclose(m, o.f.mean, rtol=1e-9, atol=1e-9), s
        assert fs._next_id == o.next_id and fs._used == o.used and fs.readings_dropped() == 0
        assert [len(h) for h in fs._hyp] == [len(h) for h in o.hyp]
        created = max(created, max(o.used))
        promoted = max(promoted, int(((o.f.count[:, L0:] > 5) & ~o.f.potential[:, L0:]).sum()))
    assert created >= 2, "the unknown landmarks were never triangulated"
    assert promoted >= 1, "no potential feature reached the full feature set"
    # the object view (:278-292): promoted features under their positive id, potential ones under the negative one
    p0 = fs.particles[0]
    ids_full = [i for i in p0.feature_set.keys() if i > L0]
    assert len(ids_full) + len(p0.potential_features) == o.used[0]
    assert all(i < 0 for i in p0.potential_features) and len(p0.hypothesis_set) == len(o.hyp[0])
    assert p0.next_id == o.next_id[0]
    # the features sit on the rays they were triangulated from: seen from the pose of the step that created them they were
    # at the blobs' bearings; after the updates they still are within the bearing gate of the true landmarks' directions
    got = np.array([f.mean[:2] for i, f in list(p0.feature_set.items()) + list(p0.potential_features.items()) if abs(i) > L0])
    assert len(got) == o.used[0] and np.isfinite(got).all()
    # snapshot round trip in the growing mode: landmarks (spare slots and flags included) and the host bookkeeping
    import os
    import tempfile

    path = os.path.join(tempfile.mkdtemp(), "grow.npz")
    fs.save_state(path)
    with np.load(path, allow_pickle=False) as snap:  # plain numeric arrays: loading a snapshot never needs pickle
        assert all(snap[k].dtype != object for k in snap.files) and "nl_readings" in snap.files
    fs2 = pk.FastSLAM([pk.Feature(mean=m.copy(), covar=c.copy()) for m, c in zip(known, kcov)], num_particles=P,
                      weight_domain="log", new_landmarks=True, spare_landmarks=spare, pair_threshold=thr, bookkeeping=bookkeeping)
    fs2.load_state(path)
    for xa, xb in zip(fs._filter.download_landmarks(), fs2._filter.download_landmarks()):
        assert np.array_equal(xa, xb)
    assert fs2._next_id == fs._next_id and fs2._used == fs._used and fs2._slot_id == fs._slot_id
    assert [list(map(tuple, h)) for h in fs2._hyp] == [list(map(tuple, h)) for h in fs._hyp]
    q0 = fs2.particles[0]
    assert set(q0.potential_features) == set(p0.potential_features) and set(q0.feature_set.keys()) == set(p0.feature_set.keys())
    fs2.close()
    fs.close()


@pytest.mark.parametrize("opts,route", [({}, "ml_fused"), ({"fused_step": 0}, "ml_handoff"), ({"fast_observe": 0}, "ml_general")])
def test_potential_features_against_the_reference_golden(lib, opts, route):
    """tests/golden/step_potential.npz: cam_cb of the unmodified reference with hand-populated potential_features
    (oracle/make_golden.py::scene_potential).  The same start through pk_upload_landmarks(PK_LANDMARK_POTENTIAL): ids (the
    reference's -(slot + 1) is the device's slot + 1), weights, means, covariances, counts, which features are still
    potential after every step, ancestors."""
    from conftest import load_golden

    g = load_golden("step_potential")
    P, L0, NP = int(g["P"]), int(g["L0"]), int(g["NP"])
    L = L0 + NP
    means = np.vstack([g["full_means"], g["pot_mean0"][0]])
    covs = np.concatenate([g["full_covs"], g["pot_cov0"][0]])
    imm = np.concatenate([g["full_immutable"], g["pot_immutable"]])
    f = lib.DeviceFilter(P, L)
    for k, v in opts.items():
        f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25), imm)
    m, c, k = f.download_landmarks()
    m[:, L0:] = g["pot_mean0"]
    c[:, L0:] = g["pot_cov0"]
    k[:, L0:] = g["pot_count0"].astype(np.int32) | lib.PK_LANDMARK_POTENTIAL
    f.upload_landmarks(0, P, m, c, k.astype(np.int32))
    for s in range(len(g["u"])):
        f.reset_weights()
        f.motion(float(g["v"]), float(g["w"]), float(g["dt"]), z=g["z"][s])
        ids = f.observe(g["blobs"][s], return_ids=True)
        assert np.array_equal(ids, np.abs(g["ids"][s]))
        poses = f.download_poses()
        assert np.allclose(poses[:, 3], g["weights"][s], rtol=1e-9, atol=0)
        gm, gc, gk = f.download_landmarks()
        assert np.allclose(gm, g["mean"][s], rtol=1e-10, atol=1e-12)
        assert np.allclose(gc, g["cov"][s], rtol=1e-9, atol=1e-13)
        assert np.array_equal(gk & ~lib.PK_LANDMARK_POTENTIAL, g["count"][s])
        assert np.array_equal((gk & lib.PK_LANDMARK_POTENTIAL) != 0, g["potential"][s])
        anc = f.resample(float(g["u"][s]), return_ancestors=True)
        assert np.array_equal(anc, g["ancestors"][s])
    f.close()


@pytest.mark.parametrize("opts,route", [({}, "ml_fused"), ({"fused_step": 0}, "ml_handoff")])
def test_potential_golden_on_the_production_routes_without_ids(lib, opts, route):
    # the same fixture with no ids asked for: the one-pass / hand-off kernels themselves (asking for ids selects the general route)
    from conftest import load_golden

    g = load_golden("step_potential")
    P, L0, NP = int(g["P"]), int(g["L0"]), int(g["NP"])
    L = L0 + NP
    means = np.vstack([g["full_means"], g["pot_mean0"][0]])
    covs = np.concatenate([g["full_covs"], g["pot_cov0"][0]])
    f = lib.DeviceFilter(P, L)
    for k, v in opts.items():
        f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25), np.concatenate([g["full_immutable"], g["pot_immutable"]]))
    m, c, k = f.download_landmarks()
    m[:, L0:] = g["pot_mean0"]
    c[:, L0:] = g["pot_cov0"]
    k[:, L0:] = g["pot_count0"].astype(np.int32) | lib.PK_LANDMARK_POTENTIAL
    f.upload_landmarks(0, P, m, c, k.astype(np.int32))
    for s in range(len(g["u"])):
        f.reset_weights()
        f.motion(float(g["v"]), float(g["w"]), float(g["dt"]), z=g["z"][s])
        f.observe(g["blobs"][s])
        assert f.observe_route() == route
        assert np.allclose(f.download_poses()[:, 3], g["weights"][s], rtol=1e-9, atol=0)
        gm, gc, gk = f.download_landmarks()
        assert np.allclose(gm, g["mean"][s], rtol=1e-10, atol=1e-12)
        assert np.array_equal(gk & ~lib.PK_LANDMARK_POTENTIAL, g["count"][s])
        assert np.array_equal((gk & lib.PK_LANDMARK_POTENTIAL) != 0, g["potential"][s])
        assert np.array_equal(f.resample(float(g["u"][s]), return_ancestors=True), g["ancestors"][s])
    f.close()
