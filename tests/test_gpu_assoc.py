"""GPU: maximum-likelihood data association (match_features_to_scan, prkt_core_v2.py:317-381).
The colour-grid kernel, the brute-force reference kernel and the NumPy oracle must return
identical ids, including on inputs built to stress the grid: ties between identical
landmarks, many blobs on one landmark, colours far outside / at the edge of the blob grid,
clustered colours (everything in one cell), and more than one landmark per thread."""
import math

import numpy as np
import pytest

from oracle.fastslam_oracle import OracleFilter, synthetic_scan, synthetic_world

pytestmark = pytest.mark.gpu


def device_ids(lib, P, means, covs, poses, blobs, kernel):
    L = means.shape[0]
    f = lib.DeviceFilter(P, L)
    f.set_option("assoc_kernel", kernel)
    f.upload_map(means, covs.reshape(L, 25))
    f.upload_poses(poses)
    ids = f.associate(blobs)
    f.close()
    return ids


def oracle_ids(P, means, covs, poses, blobs):
    o = OracleFilter(P, means, covs)
    o.x, o.y, o.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
    return o.associate(blobs)


def check(lib, P, means, covs, poses, blobs, expect_unmatched=None):
    ref = oracle_ids(P, means, covs, poses, blobs)
    g = device_ids(lib, P, means, covs, poses, blobs, 0)
    b = device_ids(lib, P, means, covs, poses, blobs, 1)
    assert np.array_equal(b, ref), "brute-force kernel differs from the oracle"
    assert np.array_equal(g, ref), "grid kernel differs from the oracle"
    if expect_unmatched is not None:
        assert (ref == 0).any() == expect_unmatched
    return ref


def rand_poses(rs, P, spread=0.3):
    poses = np.zeros((P, 4))
    poses[:, 0] = rs.normal(0, spread, P)
    poses[:, 1] = rs.normal(0, spread, P)
    poses[:, 2] = rs.normal(0, 0.05, P)
    poses[:, 3] = 1.0
    return poses


@pytest.mark.parametrize("L", [1, 7, 50, 300, 700])
def test_synthetic_ring(lib, L):
    rs = np.random.RandomState(L)
    means, covs = synthetic_world(L)
    P = 64
    poses = rand_poses(rs, P)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    ids = check(lib, P, means, covs, poses, blobs)
    assert (ids > 0).mean() > 0.5


def test_ties_between_identical_landmarks_keep_the_earliest(lib):
    means, covs = synthetic_world(12)
    means = np.vstack([means, means[3:6], means[3:6]])  # landmarks 13-15 and 16-18 duplicate 4-6
    covs = np.vstack([covs, covs[3:6], covs[3:6]])
    poses = rand_poses(np.random.RandomState(1), 32)
    blobs = synthetic_scan(means[:12], (0.0, 0.0, 0.0))
    ids = check(lib, 32, means, covs, poses, blobs)
    assert set(np.unique(ids[:, 3:6])) <= {0, 4, 5, 6}  # never the later duplicates


def test_many_blobs_on_one_landmark_and_strays(lib):
    rs = np.random.RandomState(2)
    means, covs = synthetic_world(40)
    scan = synthetic_scan(means, (0.0, 0.0, 0.0))
    dup = np.repeat(scan[5:6], 9, axis=0)
    dup[:, 0] += rs.uniform(-0.05, 0.05, 9)
    dup[:, 1:] += rs.uniform(-3, 3, (9, 3))
    strays = np.column_stack([rs.uniform(-3, 3, 6), rs.uniform(0, 255, (6, 3))])
    blobs = np.vstack([scan, dup, strays])
    ids = check(lib, 48, means, covs, rand_poses(rs, 48), blobs)
    assert (ids[:, 40:49] == 6).mean() > 0.9


def test_all_colours_equal_everything_contested(lib):
    # every blob passes the colour gate of every landmark: the contested second sweep decides
    rs = np.random.RandomState(3)
    L = 90
    means, covs = synthetic_world(L)
    means[:, 2:] = 100.0
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    check(lib, 40, means, covs, rand_poses(rs, 40, 0.1), blobs)


def test_colours_outside_and_at_the_edge_of_the_grid(lib):
    rs = np.random.RandomState(4)
    L = 60
    means, covs = synthetic_world(L)
    means[:20, 2:] = rs.uniform(-500, -400, (20, 3))      # far below every other colour
    means[20:40, 2:] = rs.uniform(900, 1500, (20, 3))      # beyond 16 cells: clamped cells
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    blobs[::3, 1:] += rs.uniform(-9.9, 9.9, (len(blobs[::3]), 3))  # cross cell borders, stay in the gate
    blobs[1::3, 1] += 17.3  # just inside the gate along one channel (17.3^2 = 299.29)
    far = blobs[:5].copy()
    far[:, 1:] += 1e6  # landmark colours far outside the blob range never match
    means2 = means.copy()
    means2[55:, 2:] += 1e7
    check(lib, 32, means2, covs, rand_poses(rs, 32, 0.1), np.vstack([blobs, far]), expect_unmatched=True)


def test_gate_boundaries(lib):
    # bearing gate 0.5 rad (:433) and colour gate 300 (:441) straddled on both sides
    means = np.array([[10.0, 0.0, 100, 100, 100], [0.0, 10.0, 30, 200, 60]])
    covs = np.broadcast_to(0.25 * np.identity(5), (2, 5, 5)).copy()
    poses = np.array([[0.0, 0.0, 0.0, 1.0]])
    eps = 1e-9
    rows = []
    for db in (0.5 - eps, 0.5 + eps, -0.5 + eps, -0.5 - eps):
        rows.append([0.0 + db, 100, 100, 100])
    for dc in (math.sqrt(300) - 1e-9, math.sqrt(300) + 1e-9):
        rows.append([math.pi / 2, 30 + dc, 200, 60])
        rows.append([math.pi / 2, 30, 200 - dc, 60])
    ids = check(lib, 1, means, covs, poses, np.array(rows))
    assert list(ids[0]) == [1, 0, 1, 0, 2, 2, 0, 0]


def test_probability_underflow_means_no_match(lib):
    # tiny covariance far from the ray: pdf underflows to exactly 0 -> strict '>' never matches
    means = np.array([[20.0, 0.0, 50, 50, 50]])
    covs = 1e-6 * np.identity(5)[None]
    poses = np.array([[0.0, 0.0, 0.0, 1.0]])
    blobs = np.array([[0.3, 50, 50, 50], [0.0, 50, 50, 50]])
    ids = check(lib, 1, means, covs, poses, blobs)
    assert list(ids[0]) == [0, 1]


def test_large_scan_many_landmarks_per_thread(lib):
    rs = np.random.RandomState(6)
    L = 1500
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.05, 0.02, 0.02))
    g = device_ids(lib, 6, means, covs, rand_poses(rs, 6), blobs, 0)
    b = device_ids(lib, 6, means, covs, rand_poses(np.random.RandomState(6), 6), blobs, 1)
    assert np.array_equal(g, b)
    ref = oracle_ids(2, means, covs, rand_poses(np.random.RandomState(6), 6)[:2], blobs)
    assert np.array_equal(g[:2], ref)


def test_observe_with_both_kernels_gives_identical_state(lib):
    rs = np.random.RandomState(7)
    L, P = 120, 200
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = rand_poses(rs, P)
    out = []
    for k in (0, 1):
        f = lib.DeviceFilter(P, L)
        f.set_option("assoc_kernel", k)
        f.upload_map(means, covs.reshape(L, 25))
        f.upload_poses(poses)
        ids = f.observe(blobs, return_ids=True)
        out.append((ids, f.download_poses(), f.download_landmarks()))
        f.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------- fast hand-off path (L <= 512)
def observe_state(lib, P, means, covs, poses, blobs, fast, immutable=None, dup=1, fused=1, want_flagged=False):
    L = means.shape[0]
    f = lib.DeviceFilter(P, L)
    f.set_option("fast_observe", fast)
    f.set_option("assoc_dup", dup)
    f.set_option("fused_step", fused)
    f.set_option("regs_step", fused)  # "one kernel": k_step_fused (L <= 512) / k_step_regs (L <= 2048); 0: hand-off + second kernel
    f.upload_map(means, covs.reshape(L, 25), immutable)
    f.upload_poses(poses)
    f.observe(blobs)  # no ids requested: the production route
    out = (f.download_poses(), f.download_landmarks())
    if want_flagged:
        out = out + (f.observe_flagged(), f.observe_route())
    f.close()
    return out


def oracle_state(P, means, covs, poses, blobs, immutable=None):
    o = OracleFilter(P, means, covs, immutable)
    o.x, o.y, o.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
    o.observe(blobs)
    return o


def check_fast(lib, P, means, covs, poses, blobs, immutable=None):
    """fast_observe = 1: k_step_fused, or with fused_step = 0 the hand-off + k_observe_fast
    (L <= 512) / k_observe_sweep (above); 2: k_observe_sweep for every L; 3: the same with eight
    hand-off slots per landmark; 0: the general kernels.  All against the oracle and each other."""
    fast = observe_state(lib, P, means, covs, poses, blobs, 1, immutable)
    two = observe_state(lib, P, means, covs, poses, blobs, 1, immutable, fused=0)
    if means.shape[0] <= 512:
        assert np.array_equal(fast[0], two[0])  # k_step_fused vs hand-off + k_observe_fast: same arithmetic, same order
    else:  # k_step_regs vs hand-off + k_observe_sweep: the log-weight is summed in another order, the maps are not
        assert np.array_equal(fast[0][:, :3], two[0][:, :3]) and np.allclose(fast[0][:, 3], two[0][:, 3], rtol=1e-12, atol=0)
    for x, y in zip(fast[1], two[1]):
        assert np.array_equal(x, y)
    gen = observe_state(lib, P, means, covs, poses, blobs, 0, immutable)
    sweep = observe_state(lib, P, means, covs, poses, blobs, 2, immutable)
    sweep8 = observe_state(lib, P, means, covs, poses, blobs, 3, immutable)
    o = oracle_state(P, means, covs, poses, blobs, immutable)
    assert np.array_equal(fast[0][:, :3], gen[0][:, :3]) and np.allclose(fast[0][:, 3], gen[0][:, 3], rtol=1e-11, atol=0)
    for x, y in zip(fast[1], gen[1]):
        assert np.array_equal(x, y)  # same device functions on the same inputs: the maps agree bit for bit
    for sw in (sweep, sweep8):
        assert np.allclose(sw[0], gen[0], rtol=1e-11, atol=0)
        assert np.allclose(sw[1][0], gen[1][0], rtol=1e-12, atol=1e-14)
        assert np.allclose(sw[1][1], gen[1][1], rtol=1e-11, atol=1e-15)
        assert np.array_equal(sw[1][2], gen[1][2])
    for got in (fast, gen, sweep, sweep8):
        w = got[0][:, 3]
        assert np.allclose(w, o.weights(), rtol=1e-9, atol=0)
        m, c, k = got[1]
        assert np.allclose(m, o.mean, rtol=1e-10, atol=1e-12)
        assert np.allclose(c, o.cov, rtol=1e-9, atol=1e-13)
        assert np.array_equal(k, o.count)
    # the two device routes agree far tighter than either agrees with NumPy
    assert np.allclose(fast[0], gen[0], rtol=1e-11, atol=0)
    assert np.allclose(fast[1][0], gen[1][0], rtol=1e-12, atol=1e-14)
    assert np.array_equal(fast[1][2], gen[1][2])


@pytest.mark.parametrize("L", [1, 2, 7, 50, 255, 500, 512])
def test_fast_observe_synthetic(lib, L):
    rs = np.random.RandomState(100 + L)
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    check_fast(lib, 40, means, covs, rand_poses(rs, 40), blobs)


def max_passers(means, blobs, poses):
    """Most blobs any landmark of any particle lets through both gates (:433, :441); above
    kFastSlots = 4 the particle is flagged and leaves the hand-off kernels for the general route."""
    m = 0
    for x, y, h, _ in poses:
        eb = np.arctan2(means[:, 1] - y, means[:, 0] - x) - h
        ok = (np.abs(blobs[:, 0][None, :] - eb[:, None]) <= 0.5) & (
            ((blobs[None, :, 1:] - means[:, None, 2:]) ** 2).sum(-1) <= 300)
        m = max(m, int(ok.sum(1).max()))
    return m


@pytest.mark.parametrize("L,P,most", [(513, 24, 4), (700, 16, 4), (1400, 6, 4), (1500, 6, 5)])
def test_sweep_observe_large_maps(lib, L, P, most):
    # L > 512: the hand-off (up to eight gate-passing blobs per landmark) is settled by
    # k_observe_sweep (landmark chunks, two sweeps); in the last case some landmark passes five
    rs = np.random.RandomState(200 + L)
    means, covs = synthetic_world(L)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))  # look-alikes three bearings apart: contested blobs
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    poses = rand_poses(rs, P)
    assert max_passers(means, blobs, poses) == most
    check_fast(lib, P, means, covs, poses, blobs)


def test_sweep_observe_up_to_eight_blobs_per_landmark_and_beyond(lib):
    # groups of seven look-alike bearing neighbours: every landmark passes seven blobs (hand-off,
    # eight slots); then twelve blobs on one landmark: more than eight -> particle flagged -> general route
    rs = np.random.RandomState(13)
    L = 700
    means, covs = synthetic_world(L)
    g = np.arange(L // 7)
    lattice = np.stack([g % 5, (g // 5) % 5, g // 25], axis=1) * 50.0 + 10.0
    means[:, 2:] = np.repeat(lattice, 7, axis=0) + rs.uniform(-2, 2, (L, 3))
    covs = covs * rs.uniform(0.5, 2.0, (L, 1, 1))
    perm = rs.permutation(L)
    means, covs = means[perm], covs[perm]
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = rand_poses(rs, 5, 0.1)
    assert 5 <= max_passers(means, blobs, poses) <= 8
    check_fast(lib, 5, means, covs, poses, blobs)
    dup = np.repeat(blobs[5:6], 12, axis=0)
    dup[:, 0] += rs.uniform(-0.05, 0.05, 12)
    dup[:, 1:] += rs.uniform(-1, 1, (12, 3))
    blobs2 = np.vstack([blobs, dup])
    assert max_passers(means, blobs2, poses) > 8
    check_fast(lib, 5, means, covs, poses, blobs2)


def test_regs_flagged_particles_get_a_second_chance_before_the_general_kernels(lib):
    """k_step_regs keeps four gate-passing blobs per landmark in registers; particles in which some landmark passes more are
    flagged.  With "regs_retry" (default) they go through the eight-slot hand-off + k_observe_sweep, without it through the
    general kernels: same maps, weights to rounding, and the oracle's."""
    rs = np.random.RandomState(17)
    L = 1400
    means, covs = synthetic_world(L)
    g = np.arange(L // 7)
    lattice = np.stack([g % 6, (g // 6) % 6, g // 36], axis=1) * 40.0 + 10.0
    means[:, 2:] = np.repeat(lattice, 7, axis=0) + rs.uniform(-2, 2, (L, 3))
    perm = rs.permutation(L)
    means, covs = means[perm], covs[perm]
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    P = 6
    poses = rand_poses(rs, P, 0.1)
    assert 5 <= max_passers(means, blobs, poses) <= 8
    out = {}
    for retry in (1, 0):
        f = lib.DeviceFilter(P, L)
        f.set_option("regs_retry", retry)
        f.upload_map(means, covs.reshape(L, 25))
        f.upload_poses(poses)
        f.observe(blobs)
        assert f.observe_route() == "ml_regs" and f.observe_flagged()[0] == P  # every particle has such a landmark
        out[retry] = (f.download_poses(), f.download_landmarks())
        f.close()
    a, b = out[1], out[0]
    assert np.array_equal(a[0][:, :3], b[0][:, :3]) and np.allclose(a[0][:, 3], b[0][:, 3], rtol=1e-11, atol=0)
    assert np.allclose(a[1][0], b[1][0], rtol=1e-12, atol=1e-14) and np.allclose(a[1][1], b[1][1], rtol=1e-11, atol=1e-15)
    assert np.array_equal(a[1][2], b[1][2])
    o = oracle_state(P, means, covs, poses, blobs)
    assert np.allclose(a[0][:, 3], o.weights(), rtol=1e-9, atol=0)
    assert np.allclose(a[1][0], o.mean, rtol=1e-10, atol=1e-12) and np.array_equal(a[1][2], o.count)


def test_sweep_observe_everything_contested_across_chunks(lib):
    # three bearing neighbours share each colour: every blob is contested by three landmarks, the
    # rivals sit in other landmark chunks (permuted order), and there are more contested pairs
    # (2700) than queue entries
    rs = np.random.RandomState(11)
    L = 900
    means, covs = synthetic_world(L)
    g = np.arange(L // 3)
    lattice = np.stack([g % 7, (g // 7) % 7, g // 49], axis=1) * 36.0 + 10.0  # groups two gates apart
    means[:, 2:] = np.repeat(lattice, 3, axis=0) + rs.uniform(-2, 2, (L, 3))
    covs = covs * rs.uniform(0.5, 2.0, (L, 1, 1))
    perm = rs.permutation(L)  # neighbours in bearing are far apart in landmark index
    means, covs = means[perm], covs[perm]
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = rand_poses(rs, 6, 0.1)
    assert max_passers(means, blobs, poses) <= 4
    check_fast(lib, 6, means, covs, poses, blobs)


def test_fused_queue_overflow_goes_the_general_way(lib):
    # L = 512 in groups of four look-alike bearing neighbours: every blob is contested by four
    # landmarks, 2048 probabilities are wanted and the LDS queue of k_step_fused holds 512 -> the
    # particle is handed to the general kernels (nothing was written yet); the two-kernel route
    # evaluates the excess in place.  All against the oracle; the fused route must then be the
    # general route bit for bit.
    rs = np.random.RandomState(17)
    L, P = 512, 6
    means, covs = synthetic_world(L)
    g = np.arange(L // 4)
    lattice = np.stack([g % 8, (g // 8) % 4, g // 32], axis=1) * 36.0 + 10.0  # groups two gates apart
    means[:, 2:] = np.repeat(lattice, 4, axis=0) + rs.uniform(-2, 2, (L, 3))
    covs = covs * rs.uniform(0.5, 2.0, (L, 1, 1))
    perm = rs.permutation(L)
    means, covs = means[perm], covs[perm]
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = rand_poses(rs, P, 0.1)
    assert max_passers(means, blobs, poses) == 4
    fused = observe_state(lib, P, means, covs, poses, blobs, 1)
    two = observe_state(lib, P, means, covs, poses, blobs, 1, fused=0)
    gen = observe_state(lib, P, means, covs, poses, blobs, 0)
    o = oracle_state(P, means, covs, poses, blobs)
    for got in (fused, two, gen):
        assert np.allclose(got[0][:, 3], o.weights(), rtol=1e-9, atol=0)
        m, c, k = got[1]
        assert np.allclose(m, o.mean, rtol=1e-10, atol=1e-12)
        assert np.allclose(c, o.cov, rtol=1e-9, atol=1e-13)
        assert np.array_equal(k, o.count)
    assert np.array_equal(fused[0], gen[0])
    for x, y in zip(fused[1], gen[1]):
        assert np.array_equal(x, y)


def test_sweep_observe_ties_across_chunks_keep_the_earliest(lib):
    # exact duplicates 600 landmarks apart: equal probabilities from different chunks, earliest wins (:377)
    rs = np.random.RandomState(12)
    base, bcov = synthetic_world(600)
    means = np.vstack([base, base[:50]])
    covs = np.vstack([bcov, bcov[:50]])
    blobs = synthetic_scan(base, (0.01, 0.0, 0.0))
    check_fast(lib, 6, means, covs, rand_poses(rs, 6, 0.1), blobs)


@pytest.mark.parametrize("L,P", [(40, 64), (500, 32), (1400, 8), (2000, 6), (3000, 4)])
def test_one_pass_routes_tight_cloud_settle_contested_blobs_themselves(lib, L, P):
    """Particles as close together as a filter's are after a resample (centimetres, hundredths of a radian): everyone sits
    inside the margins of the reference particle's candidate lists, and what the one-pass route of the map's size (k_step_fused,
    k_step_pub, k_step_pub_big -- with the second chance for what they flag) decides -- look-alike landmarks a few bearings apart contest every
    seventh blob -- is what the oracle and the general kernels decide."""
    rs = np.random.RandomState(900 + L)
    means, covs = synthetic_world(L)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))  # look-alikes three bearings apart: contested blobs
    covs = covs * rs.uniform(0.5, 2.0, (L, 1, 1))
    imm = (rs.uniform(size=L) < 0.1).astype(np.uint8)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    blobs = np.vstack([blobs, blobs[5:8] + [0.01, 1.0, -1.0, 0.5],  # second sightings
                       np.column_stack([rs.uniform(-3, 3, 5), rs.uniform(0, 255, (5, 3))])])  # strays
    blobs = blobs[rs.permutation(len(blobs))]
    poses = np.zeros((P, 4))
    poses[:, :3] = np.array([0.02, -0.01, 0.01]) + rs.normal(0, [0.02, 0.02, 0.01], (P, 3))
    poses[:, 3] = 1.0
    own = observe_state(lib, P, means, covs, poses, blobs, 1, imm, want_flagged=True)
    assert own[3] == ("ml_fused" if L <= 512 else "ml_regs" if L <= 2048 else "ml_pub_big")
    if L <= 512:
        assert own[2] == (0, 0), "k_step_fused: nobody flagged"
    elif L <= 2048:  # k_step_pub hands particles with a landmark that passes more than four look-alike blobs to the second chance
        assert own[2][1] == 0, "no candidate list overflows"
    gen = observe_state(lib, P, means, covs, poses, blobs, 0, imm)
    o = oracle_state(P, means, covs, poses, blobs, imm)
    assert np.allclose(own[0][:, 3], o.weights(), rtol=1e-9, atol=0)
    m, c, k = own[1]
    assert np.allclose(m, o.mean, rtol=1e-10, atol=1e-12) and np.allclose(c, o.cov, rtol=1e-9, atol=1e-13) and np.array_equal(k, o.count)
    assert np.allclose(own[0][:, 3], gen[0][:, 3], rtol=1e-11, atol=0)
    for x, y in zip(own[1], gen[1]):
        assert np.array_equal(x, y)


def test_regs_ties_across_the_two_rounds_keep_the_earliest(lib):
    # k_step_regs settles a lane's two landmarks (l and l + 1024) in two rounds: exact duplicates 1100 landmarks
    # apart bid in different rounds with equal probabilities, the earliest landmark must win (:377)
    rs = np.random.RandomState(21)
    base, bcov = synthetic_world(1100)
    means = np.vstack([base, base[:60]])
    covs = np.vstack([bcov, bcov[:60]])
    blobs = synthetic_scan(base, (0.01, 0.0, 0.0))
    check_fast(lib, 5, means, covs, rand_poses(rs, 5, 0.1), blobs)


@pytest.mark.parametrize("L,P", [(1025, 4), (2047, 3), (2048, 3)])
def test_regs_observe_map_sizes_around_the_two_landmarks_per_lane_limit(lib, L, P):
    rs = np.random.RandomState(400 + L)
    means, covs = synthetic_world(L)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))  # look-alikes: contested blobs
    imm = (rs.uniform(size=L) < 0.1).astype(np.uint8)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    check_fast(lib, P, means, covs, rand_poses(rs, P), blobs[rs.permutation(L)], imm)


@pytest.mark.parametrize("L", [60, 500, 900])
def test_nine_range_walk_without_the_duplicated_index_list(lib, L):
    # assoc_dup = 0: the association kernels walk the nine (r, g) columns of the colour grid instead
    # of one duplicated list (what scans too large for the list in LDS get); ids and state unchanged
    rs = np.random.RandomState(300 + L)
    means, covs = synthetic_world(L)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    poses = rand_poses(rs, 12)
    P = 12
    o = oracle_state(P, means, covs, poses, blobs)
    for fast in (0, 1):
        got = observe_state(lib, P, means, covs, poses, blobs, fast, dup=0)
        ref = observe_state(lib, P, means, covs, poses, blobs, fast, dup=1)
        if fast == 1 and L > 512:  # without the list: hand-off + k_observe_sweep; with it: k_step_regs (another summation order)
            assert np.allclose(got[0], ref[0], rtol=1e-12, atol=0)
        else:
            assert np.array_equal(got[0], ref[0])
        for x, y in zip(got[1], ref[1]):
            assert np.array_equal(x, y)
        assert np.allclose(got[0][:, 3], o.weights(), rtol=1e-9, atol=0)
        assert np.allclose(got[1][0], o.mean, rtol=1e-10, atol=1e-12)
    f = lib.DeviceFilter(P, L)
    f.set_option("assoc_dup", 0)
    f.upload_map(means, covs.reshape(L, 25))
    f.upload_poses(poses)
    assert np.array_equal(f.associate(blobs), oracle_ids(P, means, covs, poses, blobs))
    f.close()


def test_fast_observe_contested_and_ties(lib):
    rs = np.random.RandomState(5)
    means, covs = synthetic_world(60)
    means = np.vstack([means, means[10:14]])  # duplicates: exact ties, earliest must win
    covs = np.vstack([covs, covs[10:14]])
    means[20:40, 2:] = 120.0  # colour clash: many contested blobs
    blobs = synthetic_scan(means[:60], (0.0, 0.0, 0.0))
    check_fast(lib, 64, means, covs, rand_poses(rs, 64, 0.1), blobs)


def test_fast_observe_more_contested_pairs_than_the_probability_queue_holds(lib):
    # every landmark exists twice and the copies' covariances differ, so each blob is contested by
    # two landmarks: 2 x 500 queued probabilities > the 512-entry LDS queue of k_observe_fast;
    # the excess is evaluated in place and must give the same winners
    rs = np.random.RandomState(9)
    base, bcov = synthetic_world(250)
    means = np.vstack([base, base])
    covs = np.vstack([bcov, bcov * rs.uniform(0.5, 2.0, (250, 1, 1))])
    means[250:, :2] += rs.normal(0, 0.05, (250, 2))
    blobs = synthetic_scan(base, (0.01, 0.0, 0.0))
    check_fast(lib, 24, means, covs, rand_poses(rs, 24, 0.1), blobs)


def test_fast_observe_flagged_particles_take_the_general_route(lib):
    # ten blobs on one landmark: more than two gate-passing blobs for it -> particle flagged
    rs = np.random.RandomState(6)
    means, covs = synthetic_world(40)
    scan = synthetic_scan(means, (0.0, 0.0, 0.0))
    dup = np.repeat(scan[5:6], 9, axis=0)
    dup[:, 0] += rs.uniform(-0.05, 0.05, 9)
    dup[:, 1:] += rs.uniform(-3, 3, (9, 3))
    strays = np.column_stack([rs.uniform(-3, 3, 6), rs.uniform(0, 255, (6, 3))])
    imm = np.zeros(40, dtype=np.uint8)
    imm[7] = 1
    check_fast(lib, 48, means, covs, rand_poses(rs, 48), np.vstack([scan, dup, strays]), imm)


def test_fast_observe_two_blobs_on_one_landmark_in_scan_order(lib):
    rs = np.random.RandomState(8)
    means, covs = synthetic_world(30)
    scan = synthetic_scan(means, (0.0, 0.0, 0.0))
    second = scan[[3, 11, 20]].copy()
    second[:, 0] += 0.03
    second[:, 1:] += rs.uniform(-2, 2, (3, 3))
    blobs = np.vstack([second[:1], scan, second[1:]])  # one duplicate BEFORE, two AFTER in scan order
    check_fast(lib, 32, means, covs, rand_poses(rs, 32, 0.05), blobs)


def test_fast_observe_underflow_and_empty_scan(lib):
    means = np.array([[20.0, 0.0, 50, 50, 50], [0.0, 20.0, 200, 50, 50]])
    covs = np.stack([1e-6 * np.identity(5), 0.25 * np.identity(5)])
    poses = np.array([[0.0, 0.0, 0.0, 1.0]] * 4)
    check_fast(lib, 4, means, covs, poses, np.array([[0.3, 50, 50, 50], [0.0, 50, 50, 50], [1.5708, 200, 50, 50]]))
    got = observe_state(lib, 4, means, covs, poses, np.zeros((0, 4)), 1)
    assert np.array_equal(got[0][:, 3], np.ones(4))
