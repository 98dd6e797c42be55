"""GPU: the general dense path -- landmark covariances that couple position and colour (full 5x5) and measurement noise
that couples bearing and colour.  The reference takes any Feature(mean, covar) and any 4x4 Qt (prkt_core_v2.py:882-895,
:50-53) and updates with dense algebra (:804-833, :897-930); such inputs switch the filter to the 30-row layout and the
general dense kernel (route "dense").  Everything against the NumPy oracle, which is dense by construction and pinned
to the reference by the goldens (97 triples, 23 of them coupled)."""
import random

import numpy as np
import pytest

from oracle.fastslam_oracle import OracleFilter

pytestmark = pytest.mark.gpu


def random_spd(rs, n, scale):
    a = rs.normal(size=(n, n))
    q, _ = np.linalg.qr(a)
    return (q * (scale * rs.uniform(0.3, 3.0, n))) @ q.T


def dense_world(seed, L):
    rs = np.random.RandomState(seed)
    means = np.empty((L, 5))
    phi = rs.uniform(-np.pi, np.pi, L)
    rho = rs.uniform(4.0, 30.0, L)
    means[:, 0] = rho * np.cos(phi)
    means[:, 1] = rho * np.sin(phi)
    means[:, 2:] = rs.uniform(0, 255, (L, 3))
    covs = np.stack([random_spd(rs, 5, 10.0 ** rs.uniform(-1.0, 0.3)) for _ in range(L)])  # dense: xy-rgb cross terms
    immutable = (rs.uniform(size=L) < 0.15).astype(np.uint8)
    return rs, means, covs, immutable


def scan_of(rs, means, pose, frac=0.9):
    L = len(means)
    seen = np.flatnonzero(rs.uniform(size=L) < frac)
    again = seen[rs.uniform(size=len(seen)) < 0.15]
    src = np.concatenate([seen, again])
    blobs = np.empty((len(src), 4))
    blobs[:, 0] = np.arctan2(means[src, 1] - pose[1], means[src, 0] - pose[0]) - pose[2] + rs.normal(0, 0.01, len(src))
    blobs[:, 1:] = means[src, 2:] + rs.normal(0, 1.0, (len(src), 3))
    strays = np.column_stack([rs.uniform(-3, 3, 3), rs.uniform(0, 255, (3, 3))])
    blobs = np.vstack([blobs, strays])
    return blobs[rs.permutation(len(blobs))]


def close(f, o, lib):
    assert np.allclose(f.download_log_weights(), o.logw, rtol=1e-9, atol=1e-8)
    m, c, k = f.download_landmarks()
    assert np.allclose(m, o.mean, rtol=1e-9, atol=1e-10)
    assert np.allclose(c, o.cov, rtol=1e-8, atol=1e-11)
    assert np.array_equal(k, o.count)


@pytest.mark.parametrize("seed,L,P,coupled_qt", [(1, 1, 3, False), (2, 9, 17, False), (3, 60, 40, True), (4, 300, 12, False), (5, 513, 5, True)])
def test_dense_observe_ml_and_supplied_ids(lib, seed, L, P, coupled_qt):
    rs, means, covs, immutable = dense_world(seed, L)
    Qt = 0.1 * np.identity(4)
    if coupled_qt:
        Qt = random_spd(rs, 4, 0.1)  # bearing-colour coupling
    pose = np.array([0.3, -0.2, 0.1])
    poses = np.zeros((P, 4))
    poses[:, :3] = pose + rs.normal(0, [0.1, 0.1, 0.03], (P, 3))
    poses[:, 3] = rs.uniform(0.2, 1.5, P)
    blobs = scan_of(rs, means, pose)
    for supplied in (False, True):
        o = OracleFilter(P, means, covs, immutable)
        o.Qt = Qt.copy()
        o.x, o.y, o.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
        o.logw = np.log(poses[:, 3])
        f = lib.DeviceFilter(P, L)
        f.set_measurement_noise(Qt)
        f.upload_map(means, covs.reshape(L, 25), immutable)
        f.upload_poses(poses)
        if supplied:
            ids = rs.randint(0, L + 1, len(blobs)).astype(np.int32)
            o.observe(blobs, ids=ids)
            got = f.observe(blobs, ids=ids, return_ids=True)
            assert np.array_equal(got, np.tile(ids, (P, 1)))
        else:
            ids_o = o.observe(blobs)
            assert np.array_equal(f.associate(blobs), ids_o)  # association alone leaves the state alone
            got = f.observe(blobs, return_ids=True)
            assert np.array_equal(got, ids_o)
        assert f.observe_route() == "dense"
        close(f, o, lib)
        f.close()


def test_dense_whole_steps_with_resample(lib):
    rs, means, covs, immutable = dense_world(11, 40)
    P = 300
    o = OracleFilter(P, means, covs, immutable)
    f = lib.DeviceFilter(P, 40)
    f.upload_map(means, covs.reshape(40, 25), immutable)
    pose = np.zeros(3)
    for s in range(4):
        v, w, dt = 0.2, 0.1, 0.1
        h1 = pose[2] + w * dt / 2
        pose = np.array([pose[0] + v * dt * np.cos(h1), pose[1] + v * dt * np.sin(h1), h1 + w * dt / 2])
        blobs = scan_of(rs, means, pose)
        z = rs.standard_normal((P, 3))
        u = rs.uniform()
        o.reset_weights()
        o.motion(v, w, dt, z)
        o.observe(blobs)
        logw = o.logw.copy()
        anc = o.resample(u, domain="log")
        if s % 2 == 0:
            f.step(v, w, dt, blobs, u, z=z, domain=lib.PK_WEIGHTS_LOG)  # the one-call form
        else:
            f.motion(v, w, dt, z=z)
            f.observe(blobs, fresh=True)
            assert np.allclose(f.download_log_weights(), logw, rtol=1e-9, atol=1e-8)
            assert np.array_equal(f.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True), anc)
        ps = f.download_poses()
        assert np.allclose(ps[:, 0], o.x, rtol=1e-10, atol=1e-12) and np.allclose(ps[:, 2], o.h, rtol=1e-10, atol=1e-12)
    m, c, k = f.download_landmarks()
    assert np.allclose(m, o.mean, rtol=1e-9, atol=1e-10) and np.allclose(c, o.cov, rtol=1e-8, atol=1e-11) and np.array_equal(k, o.count)
    assert np.allclose(f.summary(), o.summary(), rtol=1e-10, atol=1e-12)
    f.close()


def test_compact_filter_turns_dense_and_back(lib):
    """A block-diagonal map runs on the fast kernels; a coupled Qt set afterwards converts the loaded maps to the dense
    layout (states preserved), coupled covariances uploaded for some particles do the same; a fresh block-diagonal map with
    a block-diagonal Qt returns the filter to the compact layout."""
    rs, means, covs, immutable = dense_world(21, 30)
    block = np.zeros_like(covs)
    block[:, :2, :2] = covs[:, :2, :2]
    block[:, 2:, 2:] = covs[:, 2:, 2:]
    P = 24
    pose = np.array([0.1, 0.0, 0.05])
    poses = np.zeros((P, 4))
    poses[:, :3] = pose + rs.normal(0, [0.05, 0.05, 0.02], (P, 3))
    poses[:, 3] = 1.0
    blobs = scan_of(rs, means, pose)
    o = OracleFilter(P, means, block, immutable)
    o.x, o.y, o.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
    f = lib.DeviceFilter(P, 30)
    f.upload_map(means, block.reshape(30, 25), immutable)
    f.upload_poses(poses)
    o.observe(blobs)
    f.observe(blobs)
    assert f.observe_route() in ("ml_fused", "ml_handoff")
    close(f, o, lib)
    Qt = random_spd(rs, 4, 0.2)
    o.Qt = Qt.copy()
    f.set_measurement_noise(Qt)  # loaded maps are converted
    close(f, o, lib)
    o.observe(blobs)
    f.observe(blobs)
    assert f.observe_route() == "dense"
    close(f, o, lib)
    # coupled covariances for two particles only
    newc = o.cov.copy()
    newc[3] = covs
    newc[7] = covs
    o.cov = newc
    f.upload_landmarks(3, 4, covs=covs.reshape(1, 30, 25))
    f.upload_landmarks(7, 8, covs=covs.reshape(1, 30, 25))
    o.observe(blobs)
    f.observe(blobs)
    close(f, o, lib)
    # back to the compact layout
    f.set_measurement_noise(0.1 * np.identity(4))
    f.upload_map(means, block.reshape(30, 25), immutable)
    f.upload_poses(poses)
    o2 = OracleFilter(P, means, block, immutable)
    o2.x, o2.y, o2.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
    o2.observe(blobs)
    f.observe(blobs)
    assert f.observe_route() in ("ml_fused", "ml_handoff")
    close(f, o2, lib)
    f.close()
