"""GPU: k_step_pub -- the register route (512 < L <= 2048) with contested blobs settled by static publish / subscribe
(match_one's argmax, prkt_core_v2.py:353-381, strict '>' from 0.0 and the earliest landmark on a tie).  The kernel
compares KEYS (-2 log probability) instead of probabilities and hands a particle to the general kernels wherever keys
cannot be trusted to order like probabilities: the cases below sit exactly there."""
import math

import numpy as np
import pytest

from oracle.fastslam_oracle import OracleFilter, synthetic_scan, synthetic_world

pytestmark = pytest.mark.gpu


def run(lib, means, covs, poses, blobs, opts=None, immutable=None):
    L, P = means.shape[0], poses.shape[0]
    f = lib.DeviceFilter(P, L)
    for k, v in (opts or {}).items():
        f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25), immutable)
    f.upload_poses(poses)
    if (opts or {}).get("fast_observe", 1) == 0:
        ids = f.observe(blobs, return_ids=True)
    else:
        ids = None
        f.observe(blobs)
    out = dict(logw=f.download_log_weights(), maps=f.download_landmarks(), route=f.observe_route(), flagged=f.observe_flagged()[0],
               published=f.observe_published(), ids=ids)
    f.close()
    return out


def poses_around(rs, P, spread=0.05):
    poses = np.zeros((P, 4))
    poses[:, 0] = rs.normal(0, spread, P)
    poses[:, 1] = rs.normal(0, spread, P)
    poses[:, 2] = rs.normal(0, 0.01, P)
    poses[:, 3] = 1.0
    return poses


def same_state(a, b, logw_rtol=1e-12):
    assert np.allclose(a["logw"], b["logw"], rtol=logw_rtol, atol=1e-9)
    for x, y in zip(a["maps"], b["maps"]):
        assert np.array_equal(x, y)  # same update function on the same inputs: bit for bit


def against_oracle(got, means, covs, poses, blobs, immutable=None):
    o = OracleFilter(poses.shape[0], means, covs, immutable)
    o.x, o.y, o.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
    o.observe(blobs)
    m, c, k = got["maps"]
    assert np.allclose(got["logw"], o.logw, rtol=1e-10, atol=1e-9)
    assert np.allclose(m, o.mean, rtol=1e-10, atol=1e-12) and np.allclose(c, o.cov, rtol=1e-9, atol=1e-13) and np.array_equal(k, o.count)


@pytest.mark.parametrize("L,P,tight", [(700, 6, False), (1024, 4, False), (1026, 4, False), (2000, 3, False), (2048, 3, False), (700, 6, True), (1500, 4, True),
                                       (2000, 3, True)])
def test_pub_is_the_default_instance_and_agrees_with_regs_and_the_general_kernels(lib, L, P, tight):
    rs = np.random.RandomState(900 + L)
    means, covs = synthetic_world(L)
    if tight:  # (colour blocks of a map that has been seen a few times: nobody may be flagged -- the kernel itself is held to the others)
        covs[:, 2:, 2:] = 0.01 * np.identity(3)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))  # look-alikes three bearings apart: contested blobs
    imm = (rs.uniform(size=L) < 0.1).astype(np.uint8)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))[rs.permutation(L)]
    poses = poses_around(rs, P, 0.2)
    pub = run(lib, means, covs, poses, blobs, immutable=imm)
    regs = run(lib, means, covs, poses, blobs, {"pub_step": 0}, immutable=imm)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0}, immutable=imm)
    assert pub["route"] == "ml_regs" and (pub["published"] or L == 2048)  # (2 048 blobs with look-alikes: the table does not fit LDS)
    assert regs["route"] == "ml_regs" and not regs["published"]
    # k_step_regs flags a particle for any landmark that passes more than four blobs (some here pass 5-7); k_step_pub since round 4
    # only when more than four of them have a positive probability
    assert pub["flagged"] <= regs["flagged"]
    if tight:
        assert pub["published"] and pub["flagged"] == 0
    same_state(pub, regs)
    same_state(pub, gen, 1e-11)
    against_oracle(pub, means, covs, poses, blobs, imm)


def test_a_publish_table_that_does_not_fit_leaves_the_scan_to_the_fall_back_kernels(lib):
    # (round 5: the scan's candidate lists have had their far look-alikes taken off -- k_step_regs' candidate-list instance,
    # which cannot hold a landmark's own bound against the scan's, stands back too, and every particle is flagged)
    rs = np.random.RandomState(7)
    L = 900
    means, covs = synthetic_world(L)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = poses_around(rs, 4, 0.1)
    small = run(lib, means, covs, poses, blobs, {"pub_entry_limit": 64})
    full = run(lib, means, covs, poses, blobs)
    assert full["published"] and not small["published"] and small["route"] == "ml_regs"
    same_state(small, full)


def test_identical_landmarks_tie_and_the_earliest_takes_the_blob(lib):
    # exact duplicates publish identical keys: a tie, settled by landmark order (:377), nobody flagged
    rs = np.random.RandomState(21)
    base, bcov = synthetic_world(1100)
    means = np.vstack([base, base[:60]])
    covs = np.vstack([bcov, bcov[:60]])
    blobs = synthetic_scan(base, (0.01, 0.0, 0.0))
    poses = poses_around(rs, 5, 0.1)
    pub = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["published"] and pub["flagged"] == 0
    same_state(pub, gen, 1e-11)
    m, c, k = pub["maps"]
    assert (k[:, :60] == 2).all() and (k[:, 1100:] == 0).all()  # the earlier copy took every contested blob


def test_keys_too_close_to_call_send_the_particle_to_the_general_kernels(lib):
    # two landmarks that differ by 1e-6 in one colour mean contest one blob: their keys differ by 4e-12 -- far inside the
    # 1e-7 margin and not identical -- so every particle is flagged, and the general kernels' probabilities decide
    rs = np.random.RandomState(3)
    L = 600
    means, covs = synthetic_world(L)
    means = np.vstack([means, means[40:41]])
    covs = np.vstack([covs, covs[40:41]])
    means[L, 2] += 1e-6
    blobs = synthetic_scan(means[:L], (0.0, 0.0, 0.0))
    poses = poses_around(rs, 6, 0.1)
    pub = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["published"] and pub["flagged"] == 6
    same_state(pub, gen, 1e-11)
    against_oracle(pub, means, covs, poses, blobs)


@pytest.mark.parametrize("L", [640, 2300])
@pytest.mark.parametrize("target", [1470.0, 1488.5, 1489.6, 1490.2, 1490.4, 1491.0, 1492.03, 1492.5, 1520.0])
def test_probabilities_around_the_float64_underflow_edge(lib, target, L):
    # one landmark whose own blob is so far off in colour (inside the gate) that its probability lands around the smallest
    # subnormal: -2 log pr = target.  Positive below 1490.27, zero above; between 1489 and 1491.5 the kernel evaluates the
    # probability as the reference does.  Matched or not must agree with the general kernels and the oracle.
    # (L = 2 300: k_step_pub_big, whose GATES already leave out a blob that is certainly beyond the edge -- from the float copy of
    # its colour, less a margin: 1 492.03 is beyond the edge but inside that margin, 1 492.5 beyond both)
    rs = np.random.RandomState(int(target * 10))
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    P = 4
    poses = np.zeros((P, 4))
    poses[:, 3] = 1.0  # on the spot: the position term is 0 for every landmark
    d2 = 200.0  # squared colour distance of the landmark's own blob (gate: 300)
    lm = 17
    if L > 640:  # (a denser ring: a landmark whose shifted blob no OTHER landmark has inside its colour gate -- a second contender
        # next to a subnormal winner rightly sends the particle to the general kernels, which is another test's subject)
        for lm in range(17, L):
            shifted = means[lm, 2:] + np.array([math.sqrt(d2), 0.0, 0.0])
            others = np.delete(np.arange(L), lm)
            if np.min(np.sum((means[others, 2:] - shifted) ** 2, axis=1)) > 500.0:
                break
    # key = 5 log 2pi + log det2 + log det3 + d2 / c with covariances 0.25 I_2 and c I_3
    kconst = 5.0 * math.log(2.0 * math.pi) + math.log(0.25 * 0.25)
    lo, hi = 1e-3, 0.25
    for _ in range(200):  # bisection on c: the key falls as c grows (d2 / c dominates 3 log c)
        c = 0.5 * (lo + hi)
        key = kconst + 3.0 * math.log(c) + d2 / c
        lo, hi = (c, hi) if key > target else (lo, c)
    covs[lm, 2:, 2:] = np.identity(3) * c
    blobs[lm, 1] += math.sqrt(d2)
    blobs[lm, 1:] = np.clip(blobs[lm, 1:], -1e9, 1e9)
    pub = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["published"] and pub["flagged"] == 0
    same_state(pub, gen, 1e-11)
    against_oracle(pub, means, covs, poses, blobs)
    matched = gen["ids"][:, lm] == lm + 1
    if target <= 1489.0:
        assert matched.all()
    if target >= 1491.5:
        assert not matched.any()


def test_a_subnormal_winner_next_to_another_contender_is_left_to_the_general_kernels(lib):
    # two look-alikes whose probabilities for one blob are both subnormal-small (keys ~1420 and ~1450): keys order such
    # probabilities only roughly -- the particle is flagged
    rs = np.random.RandomState(11)
    L = 640
    means, covs = synthetic_world(L)
    means = np.vstack([means, means[30:31]])
    covs = np.vstack([covs, covs[30:31]])
    kconst = 5.0 * math.log(2.0 * math.pi) + math.log(0.25 * 0.25)
    d2 = 150.0
    for idx, target in ((30, 1420.0), (L, 1450.0)):
        lo, hi = 1e-3, 0.25
        for _ in range(200):
            c = 0.5 * (lo + hi)
            key = kconst + 3.0 * math.log(c) + d2 / c
            lo, hi = (c, hi) if key > target else (lo, c)
        covs[idx, 2:, 2:] = np.identity(3) * c
    blobs = synthetic_scan(means[:L], (0.0, 0.0, 0.0))
    blobs[30, 1] += math.sqrt(d2)
    poses = np.zeros((3, 4))
    poses[:, 3] = 1.0
    pub = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["published"] and pub["flagged"] == 3
    same_state(pub, gen, 1e-11)
    against_oracle(pub, means, covs, poses, blobs)


def test_whole_steps_with_resampling_pub_against_regs(lib):
    # a short run at a register-route map size: ancestors identical, maps bit for bit
    L, P = 1500, 512
    means, covs = synthetic_world(L)
    outs = []
    for opts in ({}, {"pub_step": 0}):
        f = lib.DeviceFilter(P, L)
        for k, v in opts.items():
            f.set_option(k, v)
        f.upload_map(means, covs.reshape(L, 25))
        pose = (0.0, 0.0, 0.0)
        anc = []
        from oracle.fastslam_oracle import truth_step
        for s in range(4):
            pose = truth_step(pose, 0.2, 0.1, 0.1)
            blobs = synthetic_scan(means, pose)
            f.reset_weights()
            f.motion(0.2, 0.1, 0.1, seed=5, draw=s)
            f.observe(blobs)
            anc.append(f.resample(0.37 + 0.1 * s, return_ancestors=True, domain=lib.PK_WEIGHTS_LOG))
        outs.append((anc, f.download_poses(), f.download_landmarks(), f.observe_published()))
        f.close()
    assert outs[0][3] and not outs[1][3]
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(outs[0][1][:, :3], outs[1][1][:, :3])
    for x, y in zip(outs[0][2], outs[1][2]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("L,P", [(7, 40), (255, 20), (500, 12), (512, 8)])
def test_the_256_lane_instance_for_small_maps_is_exact_too(lib, L, P):
    # "pub_small" = 1: L <= 512 through k_step_pub<256 lanes> (three workgroups per CU) instead of k_step_fused -- not the default
    # (no faster), kept exact
    rs = np.random.RandomState(300 + L)
    means, covs = synthetic_world(L)
    if L > 20:
        n = len(means[3::7])
        means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    blobs = np.vstack([blobs, blobs[:2] + [0.01, 0.5, -0.5, 0.25]])  # two landmarks sighted twice: sequential double updates
    poses = poses_around(rs, P, 0.2)
    pub = run(lib, means, covs, poses, blobs, {"pub_small": 1})
    fused = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["route"] == "ml_fused" and pub["published"] and not fused["published"]
    same_state(pub, fused)
    same_state(pub, gen, 1e-11)
    against_oracle(pub, means, covs, poses, blobs)


@pytest.mark.parametrize("L,P,opts", [(700, 5, {}), (2000, 3, {}), (2300, 3, {}), (5000, 2, {}), (400, 6, {"pub_small": 1})])
def test_the_product_of_the_updates_norms_is_folded_before_it_overflows(lib, L, P, opts):
    """One logarithm per lane and particle (round 6, pub_fold_norms): a lane multiplies the squared Frobenius norms of its updates'
    Q (importance_factor, prkt_core_v2.py:835-849) and takes the logarithm of the product.  Colour blocks of 9e18 I give every update a
    squared norm of 2.4e38: the product of a lane's four (k_step_pub) or ten (the two-pass kernels) leaves the float64 range unless it is
    folded on the way.  Held to the general kernels, which take a logarithm per update.  (Not to the oracle: at such covariances the
    reference's own arithmetic -- (I - K H) Sigma in float64 -- has lost the updated colour block altogether, 0 where it is 0.1.)"""
    rs = np.random.RandomState(4100 + L)
    means, covs = synthetic_world(L)
    covs[:, 2:, 2:] = 9e18 * np.identity(3)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))[rs.permutation(L)]
    poses = poses_around(rs, P, 0.05)
    got = run(lib, means, covs, poses, blobs, opts)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert got["published"] and got["flagged"] == 0, (got["route"], got["flagged"])
    assert np.isfinite(got["logw"]).all() and (got["logw"] < -20.0 * L).all()  # (each update's factor is about e^-45)
    same_state(got, gen, 1e-11)


def test_small_maps_take_the_256_lane_instance_where_it_was_measured_faster(lib):
    # "pub_small" = -1 (the default): k_step_pub<256 lanes> from 5e6 particle.landmarks and 128 landmarks on (its per-scan kernels cost 27 us
    # whatever the number of particles: profiles/r06/pub_small_sweep*.log), k_step_fused below; either way the same state
    L = 320
    rs = np.random.RandomState(77)
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    for P, want in ((15625, True), (15624, False)):
        poses = poses_around(rs, P, 0.1)
        auto = run(lib, means, covs, poses, blobs)
        off = run(lib, means, covs, poses, blobs, {"pub_small": 0})
        on = run(lib, means, covs, poses, blobs, {"pub_small": 1})
        assert auto["route"] == "ml_fused" and auto["published"] == want and on["published"] and not off["published"], (P, auto["published"])
        same_state(auto, off)
        same_state(auto, on)
    means, covs = synthetic_world(112)  # (below 128 landmarks nothing was measured: k_step_fused)
    poses = poses_around(rs, 50000, 0.1)
    assert not run(lib, means, covs, poses, synthetic_scan(means, (0.02, -0.01, 0.01)))["published"]


@pytest.mark.parametrize("L,P,tight", [(2049, 3, False), (3000, 3, False), (4096, 2, False), (5000, 4, False), (2300, 3, True), (3000, 3, True), (5000, 4, True),
                                       (5200, 2, True), (5632, 2, True)])
def test_two_pass_instance_for_maps_beyond_2048_landmarks(lib, L, P, tight):
    """k_step_pub_big (2 048 < L <= 6 144): sixteen-entry candidate lists both ways, verdicts published in a first pass over
    the map, updates in a second -- against the two-sweep route it replaces as the default, the general kernels and the oracle.
    With the fresh map's loose colour blocks (0.25 I) some landmark of a ring of several thousand has five blobs with a positive
    probability inside its gates and the kernel hands every particle to the fall-back kernels (a landmark keeps four): those cases
    hold the hand-over; the TIGHT ones (0.01 I: a look-alike four colour units away is beyond the underflow edge, and the gates
    leave it out) hold the kernel itself -- nobody may be flagged there."""
    rs = np.random.RandomState(1200 + L)
    means, covs = synthetic_world(L)
    if tight:
        covs[:, 2:, 2:] = 0.01 * np.identity(3)
    n = len(means[3::7])
    means[0:7 * n:7, 2:] = means[3::7, 2:] + rs.uniform(-4, 4, (n, 3))  # look-alikes: contested blobs
    imm = (rs.uniform(size=L) < 0.1).astype(np.uint8)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))[rs.permutation(L)]
    if L > 5120:  # (the six-chunk instance: the fall-back sweep's tables hold no more than some 5 000 blobs, so a part of the scan)
        blobs = blobs[:3500]
    poses = poses_around(rs, P, 0.05)
    big = run(lib, means, covs, poses, blobs, immutable=imm)
    sweep = run(lib, means, covs, poses, blobs, {"pub_step": 0}, immutable=imm)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0}, immutable=imm)
    assert big["route"] == "ml_pub_big" and sweep["route"] == "ml_sweep" and gen["route"] == "ml_general"
    if tight:
        assert big["published"] and big["flagged"] == 0
    same_state(big, sweep, 1e-11)
    same_state(big, gen, 1e-11)
    if L <= 3000:  # (the NumPy oracle takes a while at 5 000 x 5 000)
        against_oracle(big, means, covs, poses, blobs, imm)


def test_two_pass_gates_first_look_at_the_edges_of_its_margins(lib):
    """k_step_pub_big's gates look at a FLOAT copy of a candidate's bearing and colour first (one 16-byte gather) and read the exact
    record only where that look cannot decide: blobs a hair inside and a hair outside the bearing gate (0.5 rad, :433) and the colour
    gate (300, :441) -- closer to the edge than the float copy resolves --, a blob whose colours are beyond the range the margins
    were derived for (NaN in the table: always looked at exactly), and blobs just inside / outside the margins themselves."""
    L, P = 2300, 2
    rs = np.random.RandomState(77)
    means, covs = synthetic_world(L)
    means[1400, 2:] = [1500.0, 20.0, 30.0]  # beyond |colour| <= 1 000
    pose = (0.0, 0.0, 0.0)
    base = synthetic_scan(means, pose)
    extra, expect_pass = [], []
    edge = math.sqrt(300.0)
    for i, (l, kind, delta) in enumerate([(100, "b", -2e-7), (230, "b", +2e-7), (360, "b", -5e-6), (490, "b", +5e-6), (620, "b", -3e-8), (750, "b", +3e-8),
                                          (880, "c", -1e-4), (1010, "c", +1e-4), (1140, "c", -2e-3), (1270, "c", +2e-3), (1400, "c", -1e-4), (1530, "c", -1e-7),
                                          (1660, "c", +1e-7), (1790, "bneg", -2e-7), (1920, "bneg", +2e-7)]):
        z = base[l].copy()
        if kind == "b":
            z[0] = base[l, 0] + (0.5 + delta)
        elif kind == "bneg":
            z[0] = base[l, 0] - (0.5 + delta)
        else:
            z[1] = means[l, 2] + (edge + delta)
        extra.append(z)
        expect_pass.append(delta < 0)
    blobs = np.vstack([base, np.array(extra)])
    blobs = blobs[rs.permutation(len(blobs))]
    poses = np.zeros((P, 4))
    poses[:, 3] = 1.0
    big = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert big["route"] == "ml_pub_big" and big["published"] and gen["route"] == "ml_general"
    assert big["flagged"] == 0  # the kernel itself decided: nobody went to the fall-back kernels
    same_state(big, gen, 1e-11)
    against_oracle(big, means, covs, poses, blobs)
    # the scene does what it says: a landmark whose extra blob is inside its gates was updated twice (count 2 + 2), the others once
    counts = big["maps"][2][0]
    for (l, want) in zip([100, 230, 360, 490, 620, 750, 880, 1010, 1140, 1270, 1400, 1530, 1660, 1790, 1920], expect_pass):
        assert counts[l] == (4 if want else 2), (l, counts[l], want)


def test_two_pass_instance_whole_steps_against_the_sweep_route(lib):
    L, P = 2600, 256
    means, covs = synthetic_world(L)
    outs = []
    from oracle.fastslam_oracle import truth_step
    for opts in ({}, {"pub_step": 0}):
        f = lib.DeviceFilter(P, L)
        for k, v in opts.items():
            f.set_option(k, v)
        f.upload_map(means, covs.reshape(L, 25))
        pose, anc = (0.0, 0.0, 0.0), []
        for s in range(4):
            pose = truth_step(pose, 0.2, 0.1, 0.1)
            f.reset_weights()
            f.motion(0.2, 0.1, 0.1, seed=5, draw=s)
            f.observe(synthetic_scan(means, pose))
            anc.append(f.resample(0.37 + 0.1 * s, return_ancestors=True, domain=lib.PK_WEIGHTS_LOG))
        outs.append((anc, f.download_poses(), f.download_landmarks(), f.observe_route(), f.observe_published(), f.observe_flagged()))
        f.close()
    assert outs[0][3] == "ml_pub_big" and outs[0][4] and outs[1][3] == "ml_sweep"
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(outs[0][1][:, :3], outs[1][1][:, :3])
    for x, y in zip(outs[0][2], outs[1][2]):
        assert np.array_equal(x, y)


def crowded_tail_scene(L, rs, n_extra=2):
    """A scene that loads the LAST pair of the map.  Landmark L - 1 is sighted 1 + n_extra times; landmark L - 2 is its
    near-look-alike: 18.5 colour units away -- outside the colour gate (:441, sqrt 300 = 17.3) but inside the widened gate of the
    candidate lists (19.9) -- so each lists the other's blobs (every blob of the pair is contested: it owns a publish entry)
    without passing them.  L - 1's list holds four blobs of which it takes three: wherever k_candidates' atomics put them,
    at least one taken blob sits at list index >= 2.  Colour blocks of 0.01 I: a look-alike that does pass a gate has
    probability 0 (at most four blobs of a landmark may have a positive one in the two-pass kernel)."""
    means, covs = synthetic_world(L)
    means[L - 2, 2:] = means[L - 1, 2:] + [18.5, 0.0, 0.0]
    covs[:, 2:, 2:] = 0.01 * np.identity(3)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    extra = np.repeat(blobs[L - 1:L], n_extra, axis=0)
    extra[:, 0] += 0.004 * (1 + np.arange(n_extra))
    extra[:, 1:] += rs.uniform(-0.4, 0.4, (n_extra, 3))
    blobs = np.vstack([blobs, extra])[rs.permutation(L + n_extra)]
    return means, covs, blobs


# (sizes at which no landmark of the synthetic world passes more than the four blobs k_step_pub keeps: nothing is flagged)
@pytest.mark.parametrize("L,opts", [(2000, {}), (1664, {}), (1536, {}), (1919, {}), (1535, {}), (1008, {}), (768, {}),
                                    (496, {"pub_small": 1}), (384, {"pub_small": 1}), (5008, {}), (4096, {}), (3072, {})])
def test_lanes_beyond_the_map_do_not_repeat_the_last_pair(lib, L, opts):
    """The lanes of k_step_pub / k_step_pub_big that stand beyond the map hold the last pair's rows once more.  With that
    pair's candidate lists (round 3: only list word 0 was blanked) they gated, took and WEIGHED its blobs at list index >= 2
    a second time, and from another wave (Lp a multiple of 128) overwrote keys the pair's own lane had published.  Map sizes
    where the last pair is real (L = Lp or Lp - 1) and the workgroup has lanes to spare, every instance of the kernel;
    match_features_to_scan :317-351 / the weight product :95,:124 against the oracle."""
    rs = np.random.RandomState(4000 + L)
    means, covs, blobs = crowded_tail_scene(L, rs)
    P = 3 if L <= 2048 else 2
    poses = poses_around(rs, P, 0.05)
    pub = run(lib, means, covs, poses, blobs, opts)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["published"] and pub["flagged"] == 0  # the production kernel itself did the work
    assert pub["route"] == ("ml_pub_big" if L > 2048 else "ml_fused" if L <= 512 else "ml_regs")
    assert (gen["ids"] == L).sum(axis=1).max() == 3  # the last landmark really takes its three blobs (a particle on the other side of atan2's branch cut takes none: unwrapped bearings, :408-423)
    same_state(pub, gen, 1e-11)
    if L <= 2048:  # (the big maps against the oracle: test_gpu_audit.py, particle by particle)
        against_oracle(pub, means, covs, poses, blobs)


def crowded_landmark_scene(L, rs, n_lookalike, n_sightings, tight=True):
    """Landmark 40 with n_lookalike near-copies a few bearings away (their blobs pass its gates, and its blob theirs) and
    n_sightings blobs of its own.  tight: colour blocks of 0.01 I, under which a look-alike's blob -- 2-4 colour units off -- has
    probability 0; else the initial 0.25 I, under which every gate-passing blob has a positive one."""
    means, covs = synthetic_world(L)
    lm = 40
    for i in range(n_lookalike):
        means[lm + 3 * (i + 1), 2:] = means[lm, 2:] + rs.uniform(2.0, 4.0, 3) * rs.choice([-1.0, 1.0], 3)
    if tight:
        covs[:, 2:, 2:] = 0.01 * np.identity(3)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    extra = np.repeat(blobs[lm:lm + 1], n_sightings - 1, axis=0)
    extra[:, 0] += 0.003 * (1 + np.arange(n_sightings - 1))
    extra[:, 1:] += rs.uniform(-0.05, 0.05, (n_sightings - 1, 3))
    blobs = np.vstack([blobs, extra])[rs.permutation(L + n_sightings - 1)]
    return means, covs, blobs


# (the blobs left over -- n_lookalike + n_sightings - 4 -- must find room in the slots that the first turn frees, at least
# 4 - n_sightings: the last case has one too many, and the particles go to the fall-back kernels as before)
@pytest.mark.parametrize("L,n_lookalike,n_sightings,settled", [(1500, 4, 1, True), (1800, 5, 1, True), (1024, 3, 2, True), (1300, 4, 2, True),
                                                               (700, 2, 3, True), (1800, 5, 2, False)])
def test_a_landmark_that_passes_more_blobs_than_it_has_slots_is_settled_in_the_kernel(lib, L, n_lookalike, n_sightings, settled):
    """Round 4: five to eight blobs inside a landmark's gates no longer send the particle to the second-chance kernels as long
    as at most four of them have a positive probability (match_one's argmax ignores the others, prkt_core_v2.py:369-379): the
    blobs the four slots no longer hold get their verdicts in a second turn of the key rounds."""
    rs = np.random.RandomState(70 + L + n_lookalike)
    means, covs, blobs = crowded_landmark_scene(L, rs, n_lookalike, n_sightings)
    poses = poses_around(rs, 4, 0.05)
    pub = run(lib, means, covs, poses, blobs)
    regs = run(lib, means, covs, poses, blobs, {"pub_step": 0})
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["published"] and pub["route"] == "ml_regs"
    assert regs["flagged"] == 4  # landmark 40 passes n_lookalike + n_sightings > 4 blobs: k_step_regs hands every particle on
    assert pub["flagged"] == (0 if settled else 4)   # ... k_step_pub settles them itself
    assert ((gen["ids"] == 41).sum(axis=1) == n_sightings).all()
    same_state(pub, regs)
    same_state(pub, gen, 1e-11)
    against_oracle(pub, means, covs, poses, blobs)


@pytest.mark.parametrize("L,n_lookalike,n_sightings", [(2600, 4, 1), (2600, 6, 1), (3000, 5, 2), (5000, 7, 1), (4096, 3, 3)])
def test_the_two_pass_kernel_parks_the_fifth_to_eighth_blob_of_positive_probability(lib, L, n_lookalike, n_sightings):
    """Round 6 (missing #4: the 99-ms step in the warm-up).  On the first scan of a fresh map of several thousand
    landmarks (0.25 I colour blocks) every blob inside a landmark's gates has a probability > 0, a hundred landmarks of every particle
    have five or more, and k_step_pub_big -- four slots a landmark -- handed EVERY particle of that step to the fall-back kernels.
    Now the fifth to eighth wait in the publish table's unused entries and move into a slot the landmark did not take behind the
    settling: nobody is flagged, and the state is the general kernels' bit for bit (match_one's argmax :353-381 over all of them)."""
    rs = np.random.RandomState(700 + L + n_lookalike)
    means, covs, blobs = crowded_landmark_scene(L, rs, n_lookalike, n_sightings, tight=False)
    poses = poses_around(rs, 3, 0.05)
    big = run(lib, means, covs, poses, blobs, {"pub_duo": 0})
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert big["route"] == "ml_pub_big" and big["published"]
    assert (gen["ids"] == 41).sum(axis=1).min() >= n_sightings  # landmark 40 takes its own sightings, whoever else wanted them
    assert big["flagged"] == 0, big["flagged"]
    same_state(big, gen, 1e-11)


def test_more_than_four_blobs_with_a_positive_probability_still_go_to_the_fallback_kernels(lib):
    rs = np.random.RandomState(99)
    L = 1200
    means, covs, blobs = crowded_landmark_scene(L, rs, 1, 5, tight=False)  # six gate-passing blobs, all with a positive probability
    poses = poses_around(rs, 3, 0.05)
    pub = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert pub["published"] and pub["flagged"] == 3
    same_state(pub, gen, 1e-11)
    against_oracle(pub, means, covs, poses, blobs)


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: look-alikes that are certainly beyond the underflow edge leave the candidate lists ONCE PER SCAN (k_candidates on the
# reference particle, with margins; pk_pub_math.hpp).  That verdict holds for a particle's landmark only while its own bound is
# at least the scan's: a landmark with a much WIDER colour block than the reference particle's (it missed most of the updates)
# has to look at its far list itself -- and is handed to the fall-back kernels when a blob there could match it.
def run_stale(lib, means, covs, poses, blobs, stale, opts=None):
    """One observe on a filter whose particles share (means, covs) except `stale` = {particle: (landmarks, covs)} overrides."""
    L, P = means.shape[0], poses.shape[0]
    f = lib.DeviceFilter(P, L)
    for k, v in (opts or {}).items():
        f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25))
    for p, (ls, cv) in stale.items():
        m, c, k = f.download_landmarks(p, p + 1)
        c[0, ls] = cv
        f.upload_landmarks(p, p + 1, m, c.reshape(1, L, 25), k)
    f.upload_poses(poses)
    f.observe(blobs)
    out = dict(logw=f.download_log_weights(), maps=f.download_landmarks(), route=f.observe_route(), flagged=f.observe_flagged()[0],
               flags=f.observe_flags(), published=f.observe_published())
    f.close()
    return out


@pytest.mark.parametrize("L", [700, 1500, 2600, 5000])
def test_a_landmark_with_a_weaker_bound_than_the_scans_looks_at_its_far_list_itself(lib, L):
    rs = np.random.RandomState(50 + L)
    means, covs = synthetic_world(L)
    covs[:, 2:, 2:] = 0.004 * np.identity(3)  # colour blocks of a map seen some twenty-five times (C_n = 1 / (4 + 10 n))
    covs[:, :2, :2] = 0.02 * np.identity(2)
    # look-alikes: landmark 7 k gets the colour of landmark 7 k + 3 (three bearings on: inside the bearing gate) shifted by ten --
    # inside the colour gate (radius 17.3), and FAR for the tight block: |d|^2 / c = 100 / 0.004
    n = len(means[3::7])
    shift = rs.normal(size=(n, 3))
    means[0:7 * n:7, 2:] = means[3::7, 2:] + 10.0 * shift / np.linalg.norm(shift, axis=1, keepdims=True)
    blobs = synthetic_scan(means, (0.01, -0.02, 0.005))[rs.permutation(L)]
    P = 8
    poses = poses_around(rs, P, 0.05)
    # particles 3 and 6 hold WIDE colour blocks for some of those landmarks (and for a few others): for them the look-alike's blob has
    # a positive probability -- it contends, may win, may be a second sighting of the landmark
    wide = 0.25 * np.identity(5)
    stale = {3: (np.arange(0, 7 * n, 7)[::3], wide), 6: (np.concatenate([np.arange(3, L, 7)[::4], np.arange(5, L, 50)]), wide)}
    pruned = run_stale(lib, means, covs, poses, blobs, stale)
    plain = run_stale(lib, means, covs, poses, blobs, stale, {"far_prune": 0})
    gen = run_stale(lib, means, covs, poses, blobs, stale, {"fast_observe": 0})
    assert pruned["route"] == plain["route"] == ("ml_regs" if L <= 2048 else "ml_pub_big") and pruned["published"]
    # the stale particles, and only they, were handed on by the pruned scan (their landmarks' own bounds do not meet the scan's, and a
    # far-listed blob passes their gates with a probability that is not certainly 0)
    assert set(np.nonzero(pruned["flags"])[0]) == {3, 6}, np.nonzero(pruned["flags"])[0]
    same_state(pruned, plain, 1e-11)
    same_state(pruned, gen, 1e-11)
    # (without the pruning the kernels judge every look-alike themselves, for every particle: whom they hand on is another matter --
    # at most the stale ones, where a landmark now has more than four blobs with a positive probability)
    assert set(np.nonzero(plain["flags"])[0]) <= {3, 6}


def test_a_weaker_bound_without_a_matching_far_blob_costs_nothing(lib):
    # wide colour blocks on landmarks that have NO look-alike: their bounds do not meet the scan's either, but their far lists are
    # empty or hold nothing that passes their gates -- nobody is flagged
    rs = np.random.RandomState(77)
    L = 1200
    means, covs = synthetic_world(L)
    covs[:, 2:, 2:] = 0.004 * np.identity(3)
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = poses_around(rs, 6, 0.05)
    d2 = ((means[:, None, 2:] - means[None, :, 2:]) ** 2).sum(-1) + 1e9 * np.eye(L)
    lonely = np.nonzero(d2.min(1) > 700.0)[0][:40]  # no other landmark's colour within the (widened) colour gate
    assert lonely.size >= 10
    stale = {2: (lonely, 0.25 * np.identity(5))}
    pruned = run_stale(lib, means, covs, poses, blobs, stale)
    gen = run_stale(lib, means, covs, poses, blobs, stale, {"fast_observe": 0})
    assert pruned["published"] and pruned["flagged"] == 0
    same_state(pruned, gen, 1e-11)


def test_a_whole_scan_the_kernel_stands_back_from_gets_second_chance_rows_for_every_particle(lib):
    """a scan whose publish table does not fit flags ALL particles; the second chance has rows for P / 16 (>= 1 024)
    and sent the rest through the general kernels, every scan.  The rows now grow to what the last scan wanted."""
    rs = np.random.RandomState(5)
    L, P = 2300, 3000
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    f = lib.DeviceFilter(P, L)
    f.set_option("pub_entry_limit", 8)  # no table: the two-pass kernel stands back from every scan
    f.upload_map(means, covs.reshape(L, 25))
    f.upload_poses(poses_around(rs, P, 0.05))
    f.observe(blobs)
    assert f.observe_route() == "ml_pub_big" and f.observe_flagged()[0] == P
    wanted, cap = f.observe_retry_rows()
    assert wanted == P and cap == 1024
    a = (f.download_log_weights(), f.download_landmarks())
    f.observe(blobs)  # (this scan finds rows for everybody)
    wanted, cap = f.observe_retry_rows()
    assert wanted == P and cap == P
    f.close()
    g = lib.DeviceFilter(P, L)
    g.set_option("fast_observe", 0)
    g.upload_map(means, covs.reshape(L, 25))
    g.upload_poses(poses_around(np.random.RandomState(5), P, 0.05))
    g.observe(blobs)
    assert np.allclose(a[0], g.download_log_weights(), rtol=1e-11, atol=1e-9)
    for x, y in zip(a[1], g.download_landmarks()):
        assert np.array_equal(x, y)
    g.close()


@pytest.mark.parametrize("L,reserve", [(600, 16), (600, 10), (600, 3), (2600, 16), (2600, 5)])
def test_the_xcd_walk_covers_every_particle_of_odd_ranges_and_grids(lib, L, reserve):
    """Round 5: the publish / subscribe kernels let every XCD walk a contiguous eighth of the launch's particles (pub_walk_*).  An
    observe in pieces -- ranges that are no multiple of anything, a first piece launched on a grid that is (reserve 16 -> 240
    workgroups) or is not (10 -> 246, 3 -> 253: the plain deal) a multiple of eight -- must leave exactly the state of the observe
    in one piece: every particle taken once, none twice."""
    from oracle.fastslam_oracle import synthetic_scan, synthetic_world, truth_step

    P = 1237
    means, covs = synthetic_world(L)
    rs = np.random.RandomState(3)
    z = rs.standard_normal((P, 3))
    pose = truth_step((0.0, 0.0, 0.0), 0.2, 0.1, 0.1)
    blobs = synthetic_scan(means, pose)
    out = []
    for pieces in (None, [(0, 1), (1, 8), (8, 13), (13, 700), (700, 1236), (1236, 1237)], [(0, 1237)], [(0, 530), (530, 1237)]):
        f = lib.DeviceFilter(P, L)
        f.set_option("split_reserve_cus", reserve)
        f.upload_map(means, covs.reshape(L, 25))
        f.motion(0.2, 0.1, 0.1, z=z)
        f.stage_scan(blobs)
        if pieces is None:
            f.observe_staged(fresh=True)
        else:
            assert f.staged_takes_regs()
            for i, (p0, p1) in enumerate(pieces):
                f.observe_staged_range(True, p0, p1, i == 0, i == len(pieces) - 1)
        route = f.observe_route()
        out.append((f.download_log_weights(), f.download_landmarks(), f.observe_flagged()))
        f.close()
    assert route in ("ml_regs", "ml_pub_big")
    for lw, (m, c, k), fl in out[1:]:
        assert np.array_equal(k, out[0][1][2]) and np.array_equal(m, out[0][1][0]) and np.array_equal(c, out[0][1][1])
        assert np.allclose(lw, out[0][0], rtol=1e-12, atol=1e-12)
