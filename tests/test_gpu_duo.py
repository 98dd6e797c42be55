"""GPU: k_step_pub_duo -- the two-pass publish / subscribe kernel of maps beyond 2 048 landmarks (match_features_to_scan
prkt_core_v2.py:317-381, the EKF pieces :748-930) at <= 128 VGPRs, two workgroups per CU.  It is made of k_step_pub_big's device
functions, so its maps must be that kernel's BIT FOR BIT (and the general kernels', and the oracle's to rounding); what is its own
-- one carried word per landmark, the overflow area for landmarks with several blobs of probability > 0, the held area for
landmarks that take several, the per-scan choice between the two instances -- is what the scenes below load."""
import numpy as np
import pytest

from oracle.fastslam_oracle import OracleFilter, synthetic_scan, synthetic_world, truth_step

pytestmark = pytest.mark.gpu


def run(lib, means, covs, poses, blobs, opts=None, immutable=None):
    L, P = means.shape[0], poses.shape[0]
    f = lib.DeviceFilter(P, L)
    for k, v in (opts or {}).items():
        f.set_option(k, v)
    f.upload_map(means, covs.reshape(L, 25), immutable)
    f.upload_poses(poses)
    ids = None
    if (opts or {}).get("fast_observe", 1) == 0:
        ids = f.observe(blobs, return_ids=True)
    else:
        f.observe(blobs)
    out = dict(logw=f.download_log_weights(), maps=f.download_landmarks(), route=f.observe_route(), flagged=f.observe_flagged()[0],
               published=f.observe_published(), stats=f.observe_pub_stats(), ids=ids)
    f.close()
    return out


def poses_around(rs, P, spread=0.05):
    poses = np.zeros((P, 4))
    poses[:, 0] = rs.normal(0, spread, P)
    poses[:, 1] = rs.normal(0, spread, P)
    poses[:, 2] = rs.normal(0, 0.01, P)
    poses[:, 3] = 1.0
    return poses


def same_state(a, b, logw_rtol=1e-11):
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


def lookalike_world(L, rs, every, colour_var, spread=4.0):
    """The synthetic ring with every `every`-th landmark given a near-copy three bearings on (their blobs pass each other's gates:
    contested blobs).  colour_var: the colour blocks -- 0.01: a look-alike's blob is beyond the underflow edge (one blob of probability
    > 0 per landmark: the one-word case); 0.05, or 0.02 with a smaller spread: its probability is > 0 (both landmarks of the couple
    park their slots)."""
    means, covs = synthetic_world(L)
    covs[:, 2:, 2:] = colour_var * np.identity(3)
    n = len(means[3::every])
    means[0:every * n:every, 2:] = means[3::every, 2:] + rs.uniform(-spread, spread, (n, 3))
    return means, covs


# (0.05 at several thousand landmarks: the ring's chance look-alikes are contenders too, and the table no longer fits half a CU's LDS)
@pytest.mark.parametrize("L,P,every,colour_var", [(2049, 3, 7, 0.01), (2300, 3, 7, 0.05), (3000, 3, 7, 0.05), (3072, 2, 9, 0.05), (4096, 2, 11, 0.02),
                                                   (5000, 4, 14, 0.02), (5008, 3, 14, 0.01), (5120, 2, 14, 0.02)])
def test_the_two_instances_of_the_two_pass_kernel_agree_bit_for_bit(lib, L, P, every, colour_var):
    rs = np.random.RandomState(6000 + L)
    means, covs = lookalike_world(L, rs, every, colour_var, 4.0 if colour_var > 0.03 else 2.5)
    imm = (rs.uniform(size=L) < 0.1).astype(np.uint8)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))[rs.permutation(L)]
    if L > 5008:  # (the fall-back sweep's tables hold no more than some 5 000 blobs: a part of the scan, as in test_gpu_pub.py)
        blobs = blobs[:3500]
    poses = poses_around(rs, P, 0.05)
    duo = run(lib, means, covs, poses, blobs, {"pub_duo": 1}, immutable=imm)
    trio = run(lib, means, covs, poses, blobs, {"pub_duo": 2}, immutable=imm)
    big = run(lib, means, covs, poses, blobs, {"pub_duo": 0}, immutable=imm)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0}, immutable=imm)
    assert duo["route"] == trio["route"] == big["route"] == "ml_pub_big" and gen["route"] == "ml_general"
    assert duo["stats"]["instance"] == 2 and big["stats"]["instance"] == 1, (duo["stats"], big["stats"])
    # (three workgroups per CU: a third of the LDS each -- the busiest of these scenes' tables do not fit, and the scan stays with the one-workgroup instance)
    assert trio["stats"]["instance"] in (1, 3) and (trio["stats"]["instance"] == 3 or colour_var > 0.015 or L > 4096), trio["stats"]
    assert duo["flagged"] == 0 and big["flagged"] == 0 and trio["flagged"] == 0  # the kernels themselves did the work
    same_state(trio, big)
    if colour_var > 0.015:  # the scene does what it says: landmarks with two blobs inside their gates that both count
        assert duo["stats"]["multi_landmarks"] > L // (2 * every)
    same_state(duo, big)
    same_state(duo, gen)
    if L <= 3000:  # (the NumPy oracle takes a while at 5 000 x 5 000; test_gpu_audit.py holds the big maps to it particle by particle)
        against_oracle(duo, means, covs, poses, blobs, imm)


def test_maps_beyond_ten_turns_stay_with_the_one_workgroup_instance(lib):
    """Ten carried words are what 128 VGPRs hold beside the update (twelve: two spilled): 5 121 .. 6 144 landmarks are k_step_pub_big<6>'s."""
    L, P = 5632, 2
    rs = np.random.RandomState(13)
    means, covs = lookalike_world(L, rs, 16, 0.01)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))[rs.permutation(L)][:3500]
    poses = poses_around(rs, P, 0.05)
    a = run(lib, means, covs, poses, blobs)
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert a["route"] == "ml_pub_big" and a["stats"]["instance"] == 1 and a["flagged"] == 0
    same_state(a, gen)


def test_a_fresh_maps_loose_colour_blocks_leave_the_scan_to_the_one_workgroup_instance(lib):
    """With the initial 0.25 I every look-alike is a real contender: the publish table and the landmarks with several blobs exceed
    half a CU's LDS, k_cand_entries gives the scan to k_step_pub_big (decided on the device, per scan)."""
    L, P = 5000, 2
    rs = np.random.RandomState(11)
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = poses_around(rs, P, 0.05)
    duo = run(lib, means, covs, poses, blobs, {"pub_duo": 1})
    big = run(lib, means, covs, poses, blobs, {"pub_duo": 0})
    assert duo["stats"]["instance"] == 1 and big["stats"]["instance"] == 1
    assert duo["stats"]["entries"] == big["stats"]["entries"] > 4000
    same_state(duo, big, 1e-13)


def sighted_twice_world(L, rs, n_twice, stride):
    """n_twice landmarks sighted TWICE (two blobs of their own colour, a hair apart in bearing: both have a positive probability,
    nobody else lists them -- the landmark takes both, in scan order, :88): each parks its slots in pass 1 and keeps them in the held
    area for pass 2."""
    means, covs = synthetic_world(L)
    covs[:, 2:, 2:] = 0.01 * np.identity(3)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    idx = 40 + stride * np.arange(n_twice)
    extra = blobs[idx].copy()
    extra[:, 0] += 0.003
    extra[:, 1:] += rs.uniform(-0.05, 0.05, (n_twice, 3))
    blobs = np.vstack([blobs, extra])[rs.permutation(L + n_twice)]
    return means, covs, blobs, idx


@pytest.mark.parametrize("L,n_twice,flagged", [(2600, 1, False), (3000, 30, False), (4900, 60, False), (4800, 100, True)])
def test_landmarks_that_take_two_blobs_keep_their_slots_for_pass_two(lib, L, n_twice, flagged):
    """... 64 of them per particle (kDuoHeld); a particle with more goes to the fall-back kernels, as exact as ever."""
    rs = np.random.RandomState(300 + n_twice)
    means, covs, blobs, idx = sighted_twice_world(L, rs, n_twice, 37)
    poses = poses_around(rs, 3, 0.05)
    duo = run(lib, means, covs, poses, blobs, {"pub_duo": 1})
    trio = run(lib, means, covs, poses, blobs, {"pub_duo": 2})
    big = run(lib, means, covs, poses, blobs, {"pub_duo": 0})
    gen = run(lib, means, covs, poses, blobs, {"fast_observe": 0})
    assert duo["stats"]["instance"] == 2 and big["stats"]["instance"] == 1 and trio["stats"]["instance"] == 3
    assert duo["flagged"] == (3 if flagged else 0) and big["flagged"] == 0 and trio["flagged"] == duo["flagged"]
    same_state(duo, big)
    same_state(trio, big)
    same_state(duo, gen)
    counts = duo["maps"][2]
    # two updates each (+2 per update, :914, :930); the others one -- or none where a particle stands on the other side of atan2's
    # branch cut (unwrapped bearings, :408-423)
    assert (counts[:, idx] == 4).all() and np.isin(np.delete(counts, idx, axis=1), (0, 2)).all()


def test_an_overflow_area_that_is_too_small_sends_the_particle_to_the_fall_back_kernels(lib):
    L, P = 3000, 4
    rs = np.random.RandomState(5)
    means, covs = lookalike_world(L, rs, 7, 0.05)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))[rs.permutation(L)]
    poses = poses_around(rs, P, 0.05)
    full = run(lib, means, covs, poses, blobs, {"pub_duo": 1})
    small = run(lib, means, covs, poses, blobs, {"pub_duo": 1, "pub_duo_park_limit": 100})
    none = run(lib, means, covs, poses, blobs, {"pub_duo": 1, "pub_duo_park_limit": 0})
    assert full["stats"]["instance"] == small["stats"]["instance"] == none["stats"]["instance"] == 2
    assert full["stats"]["multi_landmarks"] > 100
    assert full["flagged"] == 0 and small["flagged"] == P and none["flagged"] == P
    same_state(small, full)
    same_state(none, full)


def test_a_publish_table_limit_makes_both_instances_stand_back(lib):
    L, P = 2600, 3
    rs = np.random.RandomState(8)
    means, covs = lookalike_world(L, rs, 7, 0.05)
    blobs = synthetic_scan(means, (0.0, 0.0, 0.0))
    poses = poses_around(rs, P, 0.05)
    full = run(lib, means, covs, poses, blobs, {"pub_duo": 1})
    lim = run(lib, means, covs, poses, blobs, {"pub_duo": 1, "pub_entry_limit": 64})
    assert full["stats"]["instance"] == 2 and lim["stats"]["instance"] == 0 and not lim["published"] and lim["flagged"] == P
    same_state(lim, full)


def test_whole_steps_with_resampling_on_either_instance(lib):
    """Six whole steps (motion, observe, resample) from a fresh map: the first scans go to the one-workgroup instance, the later ones
    -- the colour blocks tighten -- to k_step_pub_duo; ancestors, poses and maps equal the run that never leaves k_step_pub_big and
    the run on the two-sweep route."""
    L, P = 2600, 384
    means, covs = synthetic_world(L)
    outs = []
    for opts in ({"pub_duo": 1}, {"pub_duo": 0}, {"pub_step": 0}, {"pub_duo": 2}):
        f = lib.DeviceFilter(P, L)
        for k, v in opts.items():
            f.set_option(k, v)
        f.upload_map(means, covs.reshape(L, 25))
        pose, anc, inst = (0.0, 0.0, 0.0), [], []
        for s in range(6):
            pose = truth_step(pose, 0.2, 0.1, 0.1)
            f.reset_weights()
            f.motion(0.2, 0.1, 0.1, seed=5, draw=s)
            f.observe(synthetic_scan(means, pose))
            inst.append(f.observe_pub_stats()["instance"])
            anc.append(f.resample(0.37 + 0.1 * s, return_ancestors=True, domain=lib.PK_WEIGHTS_LOG))
        outs.append((anc, f.download_poses(), f.download_landmarks(), inst, f.observe_route()))
        f.close()
    assert outs[0][4] == outs[1][4] == outs[3][4] == "ml_pub_big" and outs[2][4] == "ml_sweep"
    assert 2 in outs[0][3] and outs[0][3][-1] == 2 and set(outs[1][3]) == {1} and outs[3][3][-1] == 3, (outs[0][3], outs[1][3], outs[3][3])
    for other in (outs[1], outs[2], outs[3]):
        for a, b in zip(outs[0][0], other[0]):
            assert np.array_equal(a, b)
        assert np.array_equal(outs[0][1][:, :3], other[1][:, :3])
        for x, y in zip(outs[0][2], other[2]):
            assert np.array_equal(x, y)


def test_particle_ranges_and_reserved_cus(lib):
    """The split step of the sharded filter runs the kernel on particle ranges with CUs held back (pk_observe_staged_range): an
    observe in pieces equals the observe in one piece."""
    L, P = 2600, 700
    rs = np.random.RandomState(77)
    means, covs = lookalike_world(L, rs, 9, 0.02, 2.5)
    blobs = synthetic_scan(means, (0.02, -0.01, 0.01))
    poses = poses_around(rs, P, 0.05)
    outs = []
    for pieces, nl in ((None, 1), ([(0, 13), (13, 400), (400, 700)], 1), ([(250, 700), (0, 250)], 1), ([(0, 301), (301, 700)], 2)):
        f = lib.DeviceFilter(P, L)
        f.set_option("split_reserve_cus", 11)
        f.set_option("pub_duo", nl)
        f.upload_map(means, covs.reshape(L, 25))
        f.upload_poses(poses)
        f.stage_scan(blobs)
        if pieces is None:
            f.observe_staged(fresh=True)
        else:
            assert f.staged_takes_regs()
            for i, (a, b) in enumerate(pieces):
                f.observe_staged_range(True, a, b, i == 0, i == len(pieces) - 1)
        assert f.observe_pub_stats()["instance"] == 1 + nl and f.observe_flagged()[0] == 0
        outs.append((f.download_log_weights(), f.download_landmarks()))
        f.close()
    for o in outs[1:]:
        assert np.allclose(o[0], outs[0][0], rtol=1e-12, atol=1e-9)
        for x, y in zip(o[1], outs[0][1]):
            assert np.array_equal(x, y)
