"""GPU parity: the HIP path (through the C ABI) against the golden vectors captured from
the unmodified reference (tests/golden, made by oracle/make_golden.py) and against the
NumPy oracle on the same seeded inputs.

Tolerances: north_star asks for 1e-5 relative on poses and landmark means; everything here
is float64 on both sides, so the tests hold the device to 1e-9 (1e-12 where the arithmetic
is a handful of flops).  Ancestors and association ids are discrete: exact.
"""
import math

import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def is_block_diag(cov):
    return np.all(cov[:2, 2:] == 0) and np.all(cov[2:, :2] == 0)


def test_probe_known_answers(lib):
    """Every scalar function on the path vs the reference, per (pose, landmark, blob)."""
    g = load_golden("ka_triples")
    n = len(g["pom"])
    checked = 0
    for i in range(n):
        cov = g["cov"][i]
        r = lib.probe(g["pose"][i], g["mean"][i], cov, g["blob"][i], g["Qt"])
        checked += 1
        assert relerr(r["probability_of_match"], g["pom"][i]) < 1e-10, i
        assert relerr(r["prob_position_match"], g["ppm"][i]) < 1e-10, i
        assert np.allclose(r["closest_point"], g["closest"][i], rtol=1e-12, atol=1e-12), i
        assert relerr(r["prob_color_match"], g["pcm"][i]) < 1e-10, i
        assert np.allclose(r["zhat"], g["zhat"][i], rtol=1e-13, atol=1e-13), i
        assert np.allclose(r["H0"], g["H"][i][0, :2], rtol=1e-13, atol=1e-15), i
        assert np.allclose(r["Q"], g["Q"][i], rtol=1e-12, atol=1e-14), i
        assert np.allclose(r["K"], g["K"][i], rtol=1e-11, atol=1e-13), i
        assert relerr(r["weight"], g["weight"][i]) < 1e-10, i
        assert np.allclose(r["new_mean"], g["new_mean"][i], rtol=1e-12, atol=1e-12), i
        assert np.allclose(r["new_cov"], g["new_cov"][i], rtol=1e-10, atol=1e-13), i
    assert checked == n and n >= 90  # the dense triples (xy-rgb coupling) included: the general dense device functions
    assert sum(not is_block_diag(c) for c in g["cov"]) >= 20


def test_survey_known_answer(lib):
    """The vector quoted in SURVEY.md 8a."""
    r = lib.probe((0.5, -0.25, 0.3), (3, 4, 100, 150, 200), 0.25 * np.identity(5), (0.9, 101, 149, 202))
    assert relerr(r["probability_of_match"], 7.804697433262233e-07) < 1e-11
    assert relerr(r["prob_position_match"], 0.2500746500315802) < 1e-12
    assert relerr(r["prob_color_match"], 3.120947058119098e-06) < 1e-11
    assert relerr(r["weight"], 8.819701295333626e-05) < 1e-11
    assert np.allclose(r["new_mean"], [2.944889780603, 3.967582223884, 100.714285714286, 149.285714285714,
                                       201.428571428571], rtol=1e-11)


def run_fixture(lib, name, assoc):
    g = load_golden(name)
    P, L = int(g["P"]), int(g["L"])
    f = lib.DeviceFilter(P, L)
    f.set_measurement_noise(g["Qt"])
    f.upload_map(g["means0"], g["covs0"].reshape(L, 25), g["immutable"])
    S = len(g["u"])
    lsel = g["lsel"] if "lsel" in g.files else np.arange(L)
    for s in range(S):
        f.reset_weights()
        f.motion(float(g["v"]), float(g["w"]), float(g["dts"][s]), z=g["z"][s])
        pm = f.download_poses()
        assert np.allclose(pm[:, :3], g["post_motion"][s][:, :3], rtol=1e-12, atol=1e-13), (name, s)
        if assoc == "ml":
            ids = f.observe(g["blobs"][s], return_ids=True)
            assert np.array_equal(ids, g["ids"][s]), (name, s)
        else:
            # ids shared by all particles: only valid when the golden ids agree across particles
            ids0 = g["ids"][s][0]
            assert np.all(g["ids"][s] == ids0[None, :])
            f.observe(g["blobs"][s], ids=ids0)
        w = f.download_poses()[:, 3]
        assert relerr(w, g["weights"][s]) < RTOL, (name, s, relerr(w, g["weights"][s]))
        anc = f.resample(float(g["u"][s]), return_ancestors=True)
        assert np.array_equal(anc, g["ancestors"][s]), (name, s)
        pr = f.download_poses()
        assert np.allclose(pr[:, :3], g["post_resample"][s][:, :3], rtol=1e-12, atol=1e-13), (name, s)
        assert relerr(pr[:, 3], g["post_resample"][s][:, 3]) < RTOL
        m, c, k = f.download_landmarks()
        assert relerr(m[:, lsel], g["mean"][s]) < RTOL, (name, s)
        assert np.allclose(c[:, lsel], g["cov"][s], rtol=RTOL, atol=1e-13), (name, s)
        assert np.array_equal(k[:, lsel], g["count"][s]), (name, s)
        sm = f.summary()
        assert np.allclose(sm, g["summary"][s], rtol=1e-12, atol=1e-14), (name, s)
    f.close()


@pytest.mark.parametrize("name", ["step_small", "step_refscene", "step_config1"])
def test_step_fixture_ml(lib, name):
    run_fixture(lib, name, "ml")


@pytest.mark.parametrize("name", ["step_refscene", "step_config1"])
def test_step_fixture_known_ids(lib, name):
    run_fixture(lib, name, "known")


def test_motion_fixture(lib):
    g = load_golden("motion")
    P = g["start"].shape[0]
    f = lib.DeviceFilter(P, 1)
    f.upload_poses(g["start"])
    for s in range(len(g["dts"])):
        f.motion(g["controls"][s, 0], g["controls"][s, 1], g["dts"][s], z=g["z"][s])
        got = f.download_poses()
        assert np.allclose(got[:, :3], g["post"][s][:, :3], rtol=1e-12, atol=1e-13), s
    f.close()


def test_resample_fixture(lib):
    g = load_golden("resample")
    names = [k[2:] for k in g.files if k.startswith("w_")]
    assert len(names) >= 9
    for nm in names:
        w = g["w_" + nm]
        P = len(w)
        f = lib.DeviceFilter(P, 1)
        poses = np.zeros((P, 4))
        poses[:, 0] = np.arange(P)
        poses[:, 3] = w
        f.upload_poses(poses)
        for dom in (lib.PK_WEIGHTS_LINEAR, lib.PK_WEIGHTS_LOG):
            f.upload_poses(poses)
            anc = f.resample(float(g["u_" + nm]), domain=dom, return_ancestors=True)
            if nm == "zeros" and dom == lib.PK_WEIGHTS_LOG:
                assert np.all(anc == 0)
                continue
            assert np.array_equal(anc, g["a_" + nm]), (nm, dom)
            got = f.download_poses()
            assert np.array_equal(got[:, 0], np.arange(P)[g["a_" + nm]].astype(float)), nm
        f.close()


# ---------------------------------------------------------------- device noise (throughput mode)
def philox4x32_10(c, k):
    """NumPy restatement of pk_philox.hpp (Philox4x32-10): c (n,4) uint32 counters, k (2,) key."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c = c.astype(np.uint64).copy()
    k0, k1 = int(k[0]), int(k[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ np.uint64(k0)) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ np.uint64(k1)) & mask
        n3 = p0 & mask
        c = np.stack([n0, n1, n2, n3], axis=1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c


def device_normals(n, seed, draw, offset=0):
    g = np.arange(n, dtype=np.uint64) + np.uint64(offset)
    lo, hi = g & np.uint64(0xFFFFFFFF), g >> np.uint64(32)
    d_lo, d_hi = np.uint64(draw & 0xFFFFFFFF), np.uint64((draw >> 32) & 0x7FFFFFFF)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = philox4x32_10(np.stack([lo, hi, np.full(n, d_lo), np.full(n, d_hi)], 1), key)
    b = philox4x32_10(np.stack([lo, hi, np.full(n, d_lo), np.full(n, d_hi | np.uint64(0x80000000))], 1), key)

    def u53(x, y):
        return ((x >> np.uint64(5)).astype(np.float64) * 67108864.0 + (y >> np.uint64(6)).astype(np.float64) + 0.5) / 9007199254740992.0

    u1, u2, u3, u4 = u53(a[:, 0], a[:, 1]), u53(a[:, 2], a[:, 3]), u53(b[:, 0], b[:, 1]), u53(b[:, 2], b[:, 3])
    r1, r2 = np.sqrt(-2.0 * np.log(u1)), np.sqrt(-2.0 * np.log(u3))
    return np.stack([r1 * np.cos(2 * np.pi * u2), r1 * np.sin(2 * np.pi * u2), r2 * np.cos(2 * np.pi * u4)], 1)


def test_device_noise_matches_numpy_philox_and_is_shard_invariant(lib):
    from oracle.fastslam_oracle import OracleFilter

    P, seed, draw = 1000, 0x1234ABCD5678, 17
    f = lib.DeviceFilter(P, 1)
    f.motion(0.7, -0.3, 0.2, seed=seed, draw=draw)
    got = f.download_poses()
    o = OracleFilter(P, [[1, 1, 1, 1, 1.0]], [np.identity(5)])
    o.motion(0.7, -0.3, 0.2, device_normals(P, seed, draw))
    assert np.allclose(got[:, 0], o.x, rtol=1e-12, atol=1e-15)
    assert np.allclose(got[:, 1], o.y, rtol=1e-12, atol=1e-15)
    assert np.allclose(got[:, 2], o.h, rtol=1e-12, atol=1e-15)
    # counters use the GLOBAL particle index: two shards reproduce the one-filter noise bit for bit
    halves = []
    for r in range(2):
        h = lib.DeviceFilter(P // 2, 1)
        h.set_shard(r * (P // 2))
        h.motion(0.7, -0.3, 0.2, seed=seed, draw=draw)
        halves.append(h.download_poses())
        h.close()
    assert np.array_equal(np.vstack(halves), got)
    z = device_normals(200000, 99, 3)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.01
    f.close()


@pytest.mark.parametrize("ids_given", [False, True])
def test_observe_fresh_equals_reset_then_observe(lib, ids_given):
    """pk_observe_fresh = pk_reset_weights + pk_observe (prkt_core_v2.py:73 fused into the kernels)."""
    from oracle.fastslam_oracle import synthetic_scan, synthetic_world

    rs = np.random.RandomState(5)
    L, P = 90, 70
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.03, 0.01, -0.02))
    poses = np.column_stack([rs.normal(0, 0.2, P), rs.normal(0, 0.2, P), rs.normal(0, 0.05, P), rs.uniform(0.1, 2.0, P)])
    ids = np.arange(1, L + 1, dtype=np.int32) if ids_given else None
    out = []
    for fresh in (False, True):
        f = lib.DeviceFilter(P, L)
        f.upload_map(means, covs.reshape(L, 25))
        f.upload_poses(poses)
        if fresh:
            f.observe(blobs, ids=ids, fresh=True)
        else:
            f.reset_weights()
            f.observe(blobs, ids=ids)
        out.append((f.download_poses(), f.download_landmarks(), f.observe_route()))
        f.close()
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    assert out[0][2] == out[1][2] == ("known_ids" if ids_given else "ml_fused")
    # and without the reset the old weights stay in the product
    f = lib.DeviceFilter(P, L)
    f.upload_map(means, covs.reshape(L, 25))
    f.upload_poses(poses)
    f.observe(blobs, ids=ids)
    w = f.download_poses()[:, 3]
    f.close()
    assert np.allclose(w, poses[:, 3] * out[1][0][:, 3], rtol=1e-12)


def test_staged_observe_equals_observe(lib):
    """pk_stage_scan + pk_observe_staged = pk_observe_fresh; an observe in between discards the staged scan."""
    from oracle.fastslam_oracle import synthetic_scan, synthetic_world

    rs = np.random.RandomState(6)
    L, P = 120, 50
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.02, 0.0, 0.01))
    poses = np.column_stack([rs.normal(0, 0.2, P), rs.normal(0, 0.2, P), rs.normal(0, 0.05, P), np.ones(P)])
    out = []
    for staged in (False, True):
        f = lib.DeviceFilter(P, L)
        f.upload_map(means, covs.reshape(L, 25))
        f.upload_poses(poses)
        if staged:
            f.stage_scan(blobs)
            f.motion(0.2, 0.1, 0.1, seed=1, draw=0)  # other work between the two halves
            f.observe_staged(fresh=True)
        else:
            f.motion(0.2, 0.1, 0.1, seed=1, draw=0)
            f.observe(blobs, fresh=True)
        out.append((f.download_poses(), f.download_landmarks()))
        f.close()
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    f = lib.DeviceFilter(P, L)
    f.upload_map(means, covs.reshape(L, 25))
    with pytest.raises(Exception):
        f.observe_staged()  # nothing staged
    f.stage_scan(blobs)
    f.observe(blobs, ids=np.arange(1, L + 1))  # discards it
    with pytest.raises(Exception):
        f.observe_staged()
    f.close()


def test_upload_kernel_and_copy_engine_give_the_same_step(lib):
    """"upload_kernel" = 0 (hipMemcpyAsync) and 1 (k_upload reading the pinned staging block) feed the
    kernels the same scan block; "timing_stride" only thins out the event probe."""
    from oracle.fastslam_oracle import synthetic_scan, synthetic_world

    L, P = 70, 300
    means, covs = synthetic_world(L)
    blobs = synthetic_scan(means, (0.02, 0.0, 0.01))
    out = []
    for uk, stride in ((1, 1), (0, 1), (1, 3)):
        f = lib.DeviceFilter(P, L)
        f.set_option("upload_kernel", uk)
        f.set_option("timing_stride", stride)
        f.enable_timing(True)
        f.upload_map(means, covs.reshape(L, 25))
        for s in range(6):
            f.step(0.2, 0.1, 0.1, blobs, 0.1 + 0.1 * s, seed=3, draw=s, domain=lib.PK_WEIGHTS_LOG)
        tm = f.timings()
        out.append((f.download_poses(), f.download_landmarks(), tm["observe"][1]))
        f.close()
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0])
        for a, b in zip(out[0][1], other[1]):
            assert np.array_equal(a, b)
    assert out[0][2] == 6 and out[2][2] == 2  # launches bracketed: every step / every third


def test_expected_bearing_atan2_within_a_few_ulp_of_numpy(lib):
    """pk_math.hpp::pk_atan2 (round 4: one division, degree-10 polynomial, scalar-register coefficients) is what every observe
    kernel takes its expected bearings from (generate_measurement, prkt_core_v2.py:871; :408; :473).  Through pk_probe on 400
    random (pose, landmark) pairs in all octants and over six decades of distance, plus the axes, the diagonals and the point
    itself: within 4 ulp of numpy.arctan2, exact where the reference's own tests are (test_generate_measurement: pi / 4)."""
    rs = np.random.RandomState(77)
    cov = 0.25 * np.identity(5)
    cases = []
    for _ in range(400):
        ang = rs.uniform(-np.pi, np.pi)
        rad = 10.0 ** rs.uniform(-3, 3)
        pose = np.array([rs.uniform(-5, 5), rs.uniform(-5, 5), 0.0])
        cases.append((pose, pose[0] + rad * np.cos(ang), pose[1] + rad * np.sin(ang)))
    for dx, dy in ((1, 0), (0, 1), (-1, 0), (0, -1), (1, 1), (-1, 1), (-1, -1), (1, -1), (0, 0), (3, 1.2426406871192852), (1e-300, 1e-300)):
        cases.append((np.zeros(3), float(dx), float(dy)))
    worst = 0.0
    for pose, fx, fy in cases:
        mean = np.array([fx, fy, 10.0, 20.0, 30.0])
        got = lib.probe(pose, mean, cov, np.array([0.1, 10.0, 20.0, 30.0]))["zhat"][0]
        ref = np.arctan2(fy - pose[1], fx - pose[0])
        if ref == 0.0:
            assert got == 0.0
            continue
        worst = max(worst, abs(got - ref) / np.spacing(abs(ref)))
    assert worst <= 4.0, worst
    z = lib.probe(np.array([-1.0, -1.0, 0.0]), np.array([0.0, 0.0, 73.0, 165.0, 255.0]), cov, np.array([0.1, 73.0, 165.0, 255.0]))["zhat"][0]
    assert z == np.pi / 4  # test_prkt_ros2.py:403-423
    # NaN in, NaN out: the kernels rely on a NaN state failing every comparison, and v_max / v_min in pk_atan2
    # drop NaNs -- the NaN is put back by arithmetic (0 * (x + y): not foldable without fast-math, which the build does not use)
    for fx, fy in ((np.nan, 1.0), (1.0, np.nan), (np.nan, np.nan)):
        try:
            z = lib.probe(np.zeros(3), np.array([fx, fy, 10.0, 20.0, 30.0]), cov, np.array([0.1, 10.0, 20.0, 30.0]))["zhat"][0]
        except Exception:  # (a probe that refuses non-finite input is as good)
            continue
        assert np.isnan(z), (fx, fy, z)
    # signed zeros: dy = -0.0 on the positive x axis is -0.0, as the library's; dx = fx - sx is never -0.0 for finite operands
    # (a - a is +0 in round-to-nearest), so atan2(+-0, -0) = +-pi -- where pk_atan2 gives +-0 -- cannot come up
    z = lib.probe(np.array([0.0, 0.0, 0.0]), np.array([2.0, -0.0, 10.0, 20.0, 30.0]), cov, np.array([0.1, 10.0, 20.0, 30.0]))["zhat"][0]
    assert z == 0.0
