"""GPU: error behaviour and degenerate sizes of the C ABI (status codes instead of the Python
exceptions of the reference; nothing crosses the boundary silently)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_call_order_and_argument_errors(lib):
    f = lib.DeviceFilter(16, 3)
    blobs = np.array([[0.1, 1, 2, 3.0]])
    with pytest.raises(lib.PkError) as e:
        f.observe(blobs)
    assert e.value.status == lib.PK_ERR_STATE  # no map yet
    means = np.array([[5.0, 0, 1, 2, 3], [0, 5.0, 100, 2, 3], [-5.0, 0, 1, 200, 3]])
    covs = np.broadcast_to(0.25 * np.identity(5), (3, 5, 5)).copy()
    f.upload_map(means, covs.reshape(3, 25))
    with pytest.raises(lib.PkError) as e:
        f.observe(blobs, ids=[4])
    assert e.value.status == lib.PK_ERR_INVALID  # id beyond L
    with pytest.raises(lib.PkError) as e:
        f.observe(np.array([[np.nan, 1, 2, 3.0]]))
    assert e.value.status == lib.PK_ERR_INVALID
    with pytest.raises(lib.PkError) as e:
        f.resample(1.0)
    assert e.value.status == lib.PK_ERR_INVALID  # u must be in [0, 1)
    bad = covs.copy()
    bad[1, 0, 1] = np.inf
    with pytest.raises(lib.PkError) as e:
        f.upload_map(means, bad.reshape(3, 25))
    assert e.value.status == lib.PK_ERR_INVALID  # not finite
    Qt = 0.1 * np.identity(4)
    Qt[1, 2] = np.nan
    with pytest.raises(lib.PkError) as e:
        f.set_measurement_noise(Qt)
    assert e.value.status == lib.PK_ERR_INVALID
    poses = np.zeros((16, 4))
    poses[3, 3] = -1.0
    with pytest.raises(lib.PkError):
        f.upload_poses(poses)
    # an infinite weight passes "w >= 0" and would turn the log-domain scan into NaN
    poses[3, 3] = np.inf
    with pytest.raises(lib.PkError) as e:
        f.upload_poses(poses)
    assert e.value.status == lib.PK_ERR_INVALID
    for bad_w in (np.inf, np.nan, -0.5):
        with pytest.raises(lib.PkError) as e:
            f.upload_pose(2, (0.0, 0.0, 0.0, bad_w))
        assert e.value.status == lib.PK_ERR_INVALID
    # the handle is still usable after every refused call
    f.observe(blobs)
    assert np.isfinite(f.download_poses()).all()
    f.close()
    with pytest.raises(lib.PkError):
        lib.DeviceFilter(0, 3)
    with pytest.raises(lib.PkError):
        lib.DeviceFilter(4, 3, device=99)


def test_filter_without_landmarks(lib):
    # FastSLAM() with no preset features (prkt_core_v2.py:38): every blob is unseen -> weight *= 0.1 each (:94-95)
    f = lib.DeviceFilter(32, 0)
    blobs = np.array([[0.1, 1, 2, 3.0], [1.0, 4, 5, 6.0], [-2.0, 7, 8, 9.0]])
    ids = f.observe(blobs, return_ids=True)
    assert ids.shape == (32, 3) and not ids.any()
    assert np.allclose(f.download_poses()[:, 3], 1e-3, rtol=1e-13)
    anc = f.resample(0.25, return_ancestors=True)
    assert np.array_equal(anc, np.arange(32))
    assert f.summary() == (0.0, 0.0, 0.0)
    f.close()


def test_single_particle_single_landmark(lib):
    f = lib.DeviceFilter(1, 1)
    f.upload_map(np.array([[4.0, 3.0, 10, 20, 30]]), 0.25 * np.identity(5).reshape(1, 25))
    f.step(0.2, 0.1, 0.1, np.array([[np.arctan2(3, 4), 10, 20, 30.0]]), 0.99, z=np.zeros((1, 3)))
    m, c, k = f.download_landmarks()
    assert k[0, 0] == 2 and np.all(np.diag(c[0, 0]) < 0.25)
    f.close()


def test_many_blobs_few_landmarks_and_odd_counts(lib):
    from oracle.fastslam_oracle import OracleFilter

    rs = np.random.RandomState(0)
    L, B, P = 3, 700, 9  # odd L (padded slot), B >> L, P not a multiple of anything
    means = np.column_stack([rs.uniform(-10, 10, (L, 2)), rs.uniform(0, 255, (L, 3))])
    covs = np.broadcast_to(0.25 * np.identity(5), (L, 5, 5)).copy()
    blobs = np.column_stack([rs.uniform(-3, 3, B), rs.uniform(0, 255, (B, 3))])
    own = np.column_stack([np.arctan2(means[:, 1], means[:, 0]), means[:, 2:]])
    blobs[[5, 300, 699]] = own
    f = lib.DeviceFilter(P, L)
    f.upload_map(means, covs.reshape(L, 25))
    ids = f.observe(blobs, return_ids=True)
    o = OracleFilter(P, means, covs)
    assert np.array_equal(ids, o.observe(blobs))
    assert np.allclose(f.download_poses()[:, 3], o.weights(), rtol=1e-9, atol=0)
    f.close()


def _few_landmarks_many_blobs(L, B, seed):
    rs = np.random.RandomState(seed)
    means = np.column_stack([rs.uniform(-10, 10, (L, 2)), rs.uniform(0, 255, (L, 3))])
    covs = np.broadcast_to(0.25 * np.identity(5), (L, 5, 5)).copy()
    blobs = np.column_stack([rs.uniform(-3, 3, B), rs.uniform(0, 255, (B, 3))])
    own = np.column_stack([np.arctan2(means[:, 1], means[:, 0]), means[:, 2:]])
    where = rs.choice(B, size=3 * L, replace=False)
    blobs[where] = np.tile(own, (3, 1)) + rs.normal(0, 0.01, (3 * L, 4))  # every landmark is sighted three times
    return means, covs, blobs


def test_scans_beyond_the_default_64k_of_lds(lib):
    """B = 9 000 blobs: the brute-force association kernel needs 12 B of LDS per blob (108 KB), the general
    EKF kernel 4 (Lp + 2 B) = 72 KB for its per-particle chains -- both above the 64 KB a kernel gets
    without the max-dynamic-LDS attribute.  Every route must still agree with the oracle."""
    from oracle.fastslam_oracle import OracleFilter

    L, B, P = 24, 9000, 6
    means, covs, blobs = _few_landmarks_many_blobs(L, B, 11)
    o = OracleFilter(P, means, covs)
    ids_o = o.observe(blobs)
    assert (ids_o > 0).sum() >= 3 * L * P // 2
    for opts in ({"fast_observe": 0, "assoc_kernel": 1}, {"fast_observe": 0}, {}):
        f = lib.DeviceFilter(P, L)
        for k, v in opts.items():
            f.set_option(k, v)
        f.upload_map(means, covs.reshape(L, 25))
        ids = f.observe(blobs, return_ids=True)
        assert np.array_equal(ids, ids_o), opts
        assert np.allclose(f.download_log_weights(), o.logw, rtol=1e-9, atol=1e-9)
        m, c, k = f.download_landmarks()
        assert np.allclose(m, o.mean, rtol=1e-9, atol=1e-11) and np.array_equal(k, o.count)
        f.close()


def test_scan_too_large_for_the_lds_chains_is_refused_loudly(lib):
    """B = 20 000: the per-particle chains of the general EKF kernel (the last resort of every ML route)
    would need 160 KB of LDS.  ML association is refused with PK_ERR_UNSUPPORTED -- not launched to fail
    silently -- while supplied ids (chains shared by all particles, in HBM) still work."""
    L, B, P = 24, 20000, 4
    means, covs, blobs = _few_landmarks_many_blobs(L, B, 12)
    f = lib.DeviceFilter(P, L)
    f.upload_map(means, covs.reshape(L, 25))
    before = f.download_landmarks()
    with pytest.raises(lib.PkError) as e:
        f.observe(blobs)
    assert e.value.status == lib.PK_ERR_UNSUPPORTED and "LDS" in str(e.value)
    with pytest.raises(lib.PkError) as e:
        f.associate(blobs)
    assert e.value.status == lib.PK_ERR_UNSUPPORTED
    after = f.download_landmarks()
    assert all(np.array_equal(a, b) for a, b in zip(before, after)), "a refused observe must not touch the state"
    f.observe(blobs, ids=np.zeros(B, dtype=np.int32))
    assert np.allclose(f.download_log_weights(), B * np.log(0.1), rtol=1e-12)
    f.close()


def test_staged_scan_discarded_by_a_supplied_ids_observe(lib):
    """pk_stage_scan followed by a supplied-ids observe discards the staged scan; the staging ring must keep
    turning (every slot's reuse waits for the upload that last read it, recorded or not)."""
    L, P = 40, 64
    rs = np.random.RandomState(2)
    means = np.column_stack([rs.uniform(-10, 10, (L, 2)), rs.uniform(0, 255, (L, 3))])
    covs = np.broadcast_to(0.25 * np.identity(5), (L, 5, 5)).copy()
    blobs = np.column_stack([np.arctan2(means[:, 1], means[:, 0]), means[:, 2:]])
    ids = np.arange(1, L + 1, dtype=np.int32)
    f, g = lib.DeviceFilter(P, L), lib.DeviceFilter(P, L)
    f.upload_map(means, covs.reshape(L, 25))
    g.upload_map(means, covs.reshape(L, 25))
    for i in range(40):  # five trips round the ring of eight
        f.stage_scan(blobs)
        if i % 3 == 0:
            f.observe(blobs, ids=ids)       # discards the staged scan
            g.observe(blobs, ids=ids)
        else:
            f.observe_staged()
            g.observe(blobs)
    assert np.array_equal(f.download_log_weights(), g.download_log_weights())
    assert all(np.array_equal(a, b) for a, b in zip(f.download_landmarks(), g.download_landmarks()))
    f.close()
    g.close()


def test_upload_landmarks_refuses_non_finite_values_in_both_layouts(lib):
    """pk_upload_landmarks validates what it is given like pk_upload_map does (Feature.__init__ :882-895 takes anything; a NaN
    mean or covariance would sit in HBM until some later step turned every weight into NaN) -- in the compact layout and,
    since round 3, in the dense one as well -- and leaves the loaded maps untouched."""
    P, L = 6, 4
    means = np.array([[5.0, 0, 10, 20, 30], [0, 5.0, 100, 20, 30], [-5.0, 0, 10, 200, 30], [0, -5.0, 10, 20, 230]])
    covs = np.broadcast_to(0.25 * np.identity(5), (L, 5, 5)).copy()
    for dense in (False, True):
        f = lib.DeviceFilter(P, L)
        c0 = covs.copy()
        if dense:
            c0[1, 0, 3] = c0[1, 3, 0] = 0.01  # position-colour coupling: the dense 30-row layout
        f.upload_map(means, c0.reshape(L, 25))
        before = f.download_landmarks()
        m = np.broadcast_to(means, (1, L, 5)).copy()
        c = np.broadcast_to(c0.reshape(L, 25), (1, L, 25)).copy()
        bad_m, bad_c = m.copy(), c.copy()
        bad_m[0, 2, 1] = np.nan
        bad_c[0, 3, 6] = np.inf
        for kw in ({"means": bad_m}, {"covs": bad_c}, {"means": m, "covs": bad_c}):
            with pytest.raises(lib.PkError) as e:
                f.upload_landmarks(2, 3, **kw)
            assert e.value.status == lib.PK_ERR_INVALID, (dense, list(kw))
        after = f.download_landmarks()
        for a, b in zip(before, after):
            assert np.array_equal(a, b)
        f.upload_landmarks(2, 3, means=m, covs=c)  # the handle still takes good values
        f.observe(np.array([[0.1, 10, 20, 30.0]]))
        assert f.observe_route() == ("dense" if dense else "ml_fused")
        assert np.isfinite(f.download_poses()).all()
        f.close()
