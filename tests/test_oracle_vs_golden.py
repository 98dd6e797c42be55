"""CPU: the NumPy oracle against the golden vectors captured from the unmodified reference
(oracle/make_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP
path with both."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import fastslam_oracle as O


def test_triples_all_functions():
    g = load_golden("ka_triples")
    n = len(g["pom"])
    assert n >= 90
    for i in range(n):
        pose, mean, cov, blob = g["pose"][i], g["mean"][i], g["cov"][i], g["blob"][i]
        assert relerr(O.probability_of_match(pose[0], pose[1], pose[2], blob, mean, cov), g["pom"][i]) < 1e-11, i
        ppm = O.prob_position_match(mean[0], mean[1], (cov[0, 0], cov[0, 1], cov[1, 1]), pose[0], pose[1], blob[0])
        assert relerr(ppm, g["ppm"][i]) < 1e-11, i
        cx, cy = O.closest_point(mean[0], mean[1], pose[0], pose[1], blob[0])
        assert np.allclose([cx, cy], g["closest"][i], rtol=0, atol=1e-14), i
        pcm = O.prob_color_match(mean[2:], (cov[2, 2], cov[2, 3], cov[2, 4], cov[3, 3], cov[3, 4], cov[4, 4]), blob[1:])
        assert relerr(pcm, g["pcm"][i]) < 1e-11, i
        nm, nc, w, aux = O.ekf_update_dense(pose[0], pose[1], mean, cov, blob, g["Qt"])
        assert np.allclose(aux["zhat"], g["zhat"][i], rtol=1e-15, atol=1e-15), i  # np.arctan2 vs math.atan2: 1 ulp
        assert np.allclose(aux["H"], g["H"][i], rtol=1e-15, atol=0), i
        assert np.allclose(aux["Q"], g["Q"][i], rtol=1e-14, atol=1e-16), i
        assert np.allclose(aux["K"], g["K"][i], rtol=1e-12, atol=1e-15), i
        assert relerr(w, g["weight"][i]) < 1e-11, i
        assert np.allclose(nm, g["new_mean"][i], rtol=1e-13, atol=1e-13), i
        assert np.allclose(nc, g["new_cov"][i], rtol=1e-12, atol=1e-14), i
        assert relerr(np.exp(aux["logweight"]), g["weight"][i]) < 1e-11, i


@pytest.mark.parametrize("name", ["step_small", "step_refscene", "step_config1"])
def test_step_trajectories(name):
    g = load_golden(name)
    P = int(g["P"])
    f = O.OracleFilter(P, g["means0"], g["covs0"], g["immutable"], g["Qt"])
    lsel = g["lsel"] if "lsel" in g.files else slice(None)
    for s in range(len(g["u"])):
        f.reset_weights()
        f.motion(float(g["v"]), float(g["w"]), float(g["dts"][s]), g["z"][s])
        assert np.allclose(np.stack([f.x, f.y, f.h], 1), g["post_motion"][s][:, :3], rtol=0, atol=1e-15)
        ids = f.observe(g["blobs"][s])
        assert np.array_equal(ids, g["ids"][s])
        assert relerr(f.weights(), g["weights"][s]) < 1e-12
        anc = f.resample(float(g["u"][s]))
        assert np.array_equal(anc, g["ancestors"][s])
        assert relerr(f.mean[:, lsel], g["mean"][s]) < 1e-13
        assert np.allclose(f.cov[:, lsel], g["cov"][s], rtol=1e-12, atol=1e-15)
        assert np.array_equal(f.count[:, lsel], g["count"][s])
        assert np.allclose(f.summary(), g["summary"][s], rtol=0, atol=1e-15)
    if name == "step_small":
        # the fixture exercises: an unmatched blob, a doubly matched landmark, an immutable one
        assert (g["ids"][0] == 0).any()
        assert any(np.sum(g["ids"][0][0] == k) == 2 for k in range(1, 7))
        assert g["immutable"].any()


def test_motion_sequence_with_heading_wrap():
    g = load_golden("motion")
    P = g["start"].shape[0]
    f = O.OracleFilter(P, [[1, 1, 1, 1, 1.0]], [np.identity(5)])
    f.x, f.y, f.h = g["start"][:, 0].copy(), g["start"][:, 1].copy(), g["start"][:, 2].copy()
    assert np.abs(g["post"][..., 2]).max() <= np.pi and np.abs(g["start"][:, 2]).max() <= np.pi
    for s in range(len(g["dts"])):
        f.motion(g["controls"][s, 0], g["controls"][s, 1], g["dts"][s], g["z"][s])
        assert np.allclose(np.stack([f.x, f.y, f.h], 1), g["post"][s][:, :3], rtol=0, atol=5e-15), s


def test_resample_ancestors():
    g = load_golden("resample")
    names = [k[2:] for k in g.files if k.startswith("w_")]
    for nm in names:
        w, u, a = g["w_" + nm], float(g["u_" + nm]), g["a_" + nm]
        assert np.array_equal(O.low_variance_ancestors(w, u), a), nm
        assert np.array_equal(O.low_variance_ancestors_sequential(w, u), a), nm
    assert np.all(g["a_zeros"] == 0)  # all-zero weights: P copies of particle 0


def test_log_domain_matches_linear_when_nothing_underflows():
    g = load_golden("step_config1")
    P = int(g["P"])
    f1 = O.OracleFilter(P, g["means0"], g["covs0"])
    f2 = O.OracleFilter(P, g["means0"], g["covs0"])
    for s in range(2):
        _, a1 = f1.step(float(g["v"]), float(g["w"]), float(g["dts"][s]), g["z"][s], g["blobs"][s], float(g["u"][s]))
        _, a2 = f2.step(float(g["v"]), float(g["w"]), float(g["dts"][s]), g["z"][s], g["blobs"][s], float(g["u"][s]),
                        domain="log")
        assert np.array_equal(a1, a2)


def potential_filter(g):
    """The oracle loaded with the step_potential fixture's start: five full features, and in slots 5..8 of every
    particle the potential features the reference carried under the ids -6..-9 (prkt_core_v2.py:109-118)."""
    P, L0, NP = int(g["P"]), int(g["L0"]), int(g["NP"])
    means = np.vstack([g["full_means"], g["pot_mean0"][0]])
    covs = np.concatenate([g["full_covs"], g["pot_cov0"][0]])
    imm = np.concatenate([g["full_immutable"], g["pot_immutable"]])
    f = O.OracleFilter(P, means, covs, imm, g["Qt"])
    f.mean[:, L0:] = g["pot_mean0"]
    f.cov[:, L0:] = g["pot_cov0"]
    f.count[:, L0:] = g["pot_count0"]
    f.potential[:, L0:] = True
    return f


def test_potential_feature_branch_against_the_reference():
    """What pins OracleFilter.potential (and through it the device rule): cam_cb of the unmodified reference with
    hand-populated potential_features -- matched like full features (listed behind them, :366-367), updated, weight
    0.1 instead of the importance factor (:111-112), promoted to the positive id past update_count 5 (:113-117)."""
    g = load_golden("step_potential")
    P, L0 = int(g["P"]), int(g["L0"])
    f = potential_filter(g)
    ar = np.arange(P)[:, None]
    saw_potential = saw_promoted = False
    for s in range(len(g["u"])):
        f.reset_weights()
        f.motion(float(g["v"]), float(g["w"]), float(g["dt"]), g["z"][s])
        assert np.allclose(np.stack([f.x, f.y, f.h], 1), g["post_motion"][s][:, :3], rtol=0, atol=1e-15)
        assert relerr(f.mean, g["pre_mean"][s]) < 1e-13 and np.array_equal(f.count, g["pre_count"][s])
        assert np.array_equal(f.potential, g["pre_potential"][s])
        pre_pot = f.potential.copy()
        ids = f.observe(g["blobs"][s])
        ref = g["ids"][s]
        assert (ref != 0).all() and np.array_equal(ids, np.abs(ref))  # the reference's negative id -(slot + 1): the same slot
        assert np.array_equal(ref < 0, pre_pot[ar, np.abs(ref) - 1])   # ... negative exactly while the feature is potential
        saw_potential |= bool((ref < 0).any())
        saw_promoted |= bool((ref > L0).any())
        assert relerr(f.weights(), g["weights"][s]) < 1e-12
        assert relerr(f.mean, g["mean"][s]) < 1e-13
        assert np.allclose(f.cov, g["cov"][s], rtol=1e-12, atol=1e-15)
        assert np.array_equal(f.count, g["count"][s]) and np.array_equal(f.potential, g["potential"][s])
        anc = f.resample(float(g["u"][s]))
        assert np.array_equal(anc, g["ancestors"][s])
    assert saw_potential and saw_promoted
    assert g["potential"][-1][:, L0 + 3].all() and (g["count"][-1][:, L0 + 3] == g["pot_count0"][3]).all()  # the immutable one: never
