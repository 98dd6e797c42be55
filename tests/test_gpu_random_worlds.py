"""GPU: randomised worlds (fixed seeds) through every maximum-likelihood route against the oracle.

Each case draws its own map size, scan (landmark sightings with noise, repeated sightings, stray
blobs, shuffled scan order), block-diagonal SPD covariances of random scale and shape, immutable
flags, measurement noise and particle poses, then compares weights, means, covariances and update
counts of: k_step_fused (default, L <= 512), hand-off + k_observe_fast, hand-off + k_observe_sweep
(four and eight slots), the general kernels, with the NumPy oracle."""
import numpy as np
import pytest

from oracle.fastslam_oracle import OracleFilter

pytestmark = pytest.mark.gpu


def random_spd(rs, n, scale):
    a = rs.normal(size=(n, n))
    q, _ = np.linalg.qr(a)
    ev = scale * rs.uniform(0.3, 3.0, n)
    return (q * ev) @ q.T


def random_case(seed, L=None, P=None):
    rs = np.random.RandomState(seed)
    L0 = int(rs.choice([1, 2, 3, 9, 33, 64, 130, 257, 400, 512, 513, 640]))
    P0 = int(rs.choice([1, 5, 17, 40]))
    L = L0 if L is None else L
    P = P0 if P is None else P
    means = np.empty((L, 5))
    phi = rs.uniform(-np.pi, np.pi, L)
    rho = rs.uniform(4.0, 30.0, L)
    means[:, 0] = rho * np.cos(phi)
    means[:, 1] = rho * np.sin(phi)
    ncol = max(1, int(L * rs.choice([0.2, 0.6, 1.0])))  # fewer colours than landmarks: look-alikes
    palette = rs.uniform(0, 255, (ncol, 3))
    means[:, 2:] = palette[rs.randint(0, ncol, L)] + rs.normal(0, 1.5, (L, 3))
    covs = np.zeros((L, 5, 5))
    for l in range(L):
        covs[l, :2, :2] = random_spd(rs, 2, 10.0 ** rs.uniform(-2, 0.5))
        covs[l, 2:, 2:] = random_spd(rs, 3, 10.0 ** rs.uniform(-1.5, 1.0))
    immutable = (rs.uniform(size=L) < 0.15).astype(np.uint8)
    pose = np.array([rs.normal(0, 0.5), rs.normal(0, 0.5), rs.normal(0, 0.4)])
    poses = np.zeros((P, 4))
    poses[:, :3] = pose + rs.normal(0, [0.15, 0.15, 0.03], (P, 3))
    poses[:, 3] = rs.uniform(0.2, 1.5, P)
    seen = np.flatnonzero(rs.uniform(size=L) < rs.choice([0.5, 0.9, 1.0]))
    again = seen[rs.uniform(size=len(seen)) < 0.1]  # sighted twice
    src = np.concatenate([seen, again])
    blobs = np.empty((len(src), 4))
    blobs[:, 0] = np.arctan2(means[src, 1] - pose[1], means[src, 0] - pose[0]) - pose[2] + rs.normal(0, 0.01, len(src))
    blobs[:, 1:] = means[src, 2:] + rs.normal(0, 1.0, (len(src), 3))
    strays = np.column_stack([rs.uniform(-3, 3, 4), rs.uniform(0, 255, (4, 3))])
    blobs = np.vstack([blobs, strays])
    blobs = blobs[rs.permutation(len(blobs))]
    qt = np.zeros((4, 4))
    qt[0, 0] = 10.0 ** rs.uniform(-2, -0.5)
    qt[1:, 1:] = random_spd(rs, 3, 10.0 ** rs.uniform(-1.5, 0))
    return L, P, means, covs, immutable, poses, blobs, qt


def device_state(lib, case, opts):
    L, P, means, covs, immutable, poses, blobs, qt = case
    f = lib.DeviceFilter(P, L)
    for k, v in opts.items():
        f.set_option(k, v)
    f.set_measurement_noise(qt)
    f.upload_map(means, covs.reshape(L, 25), immutable)
    f.upload_poses(poses)
    f.observe(blobs)
    out = (f.download_log_weights(), f.download_landmarks(), f.observe_route())
    f.close()
    return out


@pytest.mark.parametrize("seed", range(24))
def test_random_world_all_routes(lib, seed):
    case = random_case(1000 + seed)
    L, P, means, covs, immutable, poses, blobs, qt = case
    o = OracleFilter(P, means, covs, immutable)
    o.Qt = qt.copy()
    o.x, o.y, o.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
    o.logw = np.log(poses[:, 3])
    o.observe(blobs)
    routes = {
        "default": {},
        "two_kernel": {"fused_step": 0, "regs_step": 0},
        "sweep4": {"fast_observe": 2},
        "sweep8": {"fast_observe": 3},
        "general": {"fast_observe": 0},
        "general_brute": {"fast_observe": 0, "assoc_kernel": 1},
        "regs_without_pub": {"pub_step": 0},  # L > 512: k_step_regs instead of k_step_pub (L <= 512: the default route again)
    }
    got = {name: device_state(lib, case, opts) for name, opts in routes.items()}
    assert got["default"][2] == ("ml_fused" if L <= 512 else "ml_regs")
    assert got["two_kernel"][2] == ("ml_handoff" if L <= 512 else "ml_sweep")
    assert got["general"][2] == "ml_general"
    for name, (logw, (m, c, k), _) in got.items():
        assert np.allclose(logw, o.logw, rtol=1e-10, atol=1e-9), name  # log domain: the weights themselves underflow
        assert np.allclose(m, o.mean, rtol=1e-9, atol=1e-11), name
        assert np.allclose(c, o.cov, rtol=1e-8, atol=1e-13), name
        assert np.array_equal(k, o.count), name


@pytest.mark.parametrize("seed,L,P", [(0, 700, 5), (1, 1024, 3), (2, 1025, 3), (3, 1500, 3), (4, 2048, 2), (5, 2000, 2)])
def test_random_world_large_maps_one_pass_route(lib, seed, L, P):
    """The same random worlds at map sizes of the one-pass register-resident route (k_step_regs, 512 < L <= 2048):
    one pass, two sweeps, general kernels -- all against the oracle, and the maps of the one-pass and the
    two-sweep routes bit for bit (same device functions)."""
    case = random_case(7000 + seed, L=L, P=P)
    L, P, means, covs, immutable, poses, blobs, qt = case
    o = OracleFilter(P, means, covs, immutable)
    o.Qt = qt.copy()
    o.x, o.y, o.h = poses[:, 0].copy(), poses[:, 1].copy(), poses[:, 2].copy()
    o.logw = np.log(poses[:, 3])
    o.observe(blobs)
    routes = {"default": {}, "regs_without_pub": {"pub_step": 0}, "sweep": {"regs_step": 0}, "general": {"fast_observe": 0}}
    got = {name: device_state(lib, case, opts) for name, opts in routes.items()}
    # the one-pass route needs the scan tables AND its probability queue in LDS (about 2 000 blobs, depending on
    # how the colours fill the grid); larger scans take the two-sweep route
    assert got["default"][2] in ("ml_regs", "ml_sweep") and (len(blobs) > 1800 or got["default"][2] == "ml_regs")
    assert got["sweep"][2] == "ml_sweep"
    for name, (logw, (m, c, k), _) in got.items():
        assert np.allclose(logw, o.logw, rtol=1e-10, atol=1e-9), name
        assert np.allclose(m, o.mean, rtol=1e-9, atol=1e-11), name
        assert np.allclose(c, o.cov, rtol=1e-8, atol=1e-13), name
        assert np.array_equal(k, o.count), name
    for x, y in zip(got["default"][1], got["sweep"][1]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("seed", range(8))
def test_random_world_several_steps(lib, seed):
    """Whole steps (motion with host-supplied normals, ML association, EKF, log-domain resample) on a
    random world: ancestors exact, log-weights / poses / maps within tolerance of the oracle after
    every step, for the default route and the general kernels."""
    rs = np.random.RandomState(5000 + seed)
    L, _, means, covs, immutable, _, _, qt = random_case(5000 + seed)
    P = 1030 if L <= 64 else (200 if L <= 257 else 48)  # more than one scan block where the oracle can afford it
    o = OracleFilter(P, means, covs, immutable)
    o.Qt = qt.copy()
    filters = []
    for opts in ({}, {"fast_observe": 0}):
        f = lib.DeviceFilter(P, L)
        for k, v in opts.items():
            f.set_option(k, v)
        f.set_measurement_noise(qt)
        f.upload_map(means, covs.reshape(L, 25), immutable)
        filters.append(f)
    pose = np.zeros(3)
    for s in range(4):
        v, w, dt = 0.2 + 0.1 * rs.uniform(), 0.1 * rs.normal(), 0.1
        h1 = pose[2] + w * dt / 2
        pose = np.array([pose[0] + v * dt * np.cos(h1), pose[1] + v * dt * np.sin(h1), h1 + w * dt / 2])
        seen = np.flatnonzero(rs.uniform(size=L) < 0.9)
        blobs = np.empty((len(seen), 4))
        blobs[:, 0] = np.arctan2(means[seen, 1] - pose[1], means[seen, 0] - pose[0]) - pose[2] + rs.normal(0, 0.01, len(seen))
        blobs[:, 1:] = means[seen, 2:] + rs.normal(0, 1.0, (len(seen), 3))
        blobs = blobs[rs.permutation(len(blobs))]
        z = rs.standard_normal((P, 3))
        u = rs.uniform()
        o.reset_weights()
        o.motion(v, w, dt, z)
        o.observe(blobs)
        logw = o.logw.copy()
        anc = o.resample(u, domain="log")  # gathers the oracle's particles as well
        for f in filters:
            f.motion(v, w, dt, z=z)
            f.observe(blobs, fresh=True)
            assert np.allclose(f.download_log_weights(), logw, rtol=1e-10, atol=1e-9)
            got = f.resample(u, domain=lib.PK_WEIGHTS_LOG, return_ancestors=True)
            assert np.array_equal(got, anc), "ancestors differ from the oracle at step %d" % s
            ps = f.download_poses()
            assert np.allclose(ps[:, 0], o.x, rtol=1e-10, atol=1e-13) and np.allclose(ps[:, 1], o.y, rtol=1e-10, atol=1e-13)
            assert np.allclose(ps[:, 2], o.h, rtol=1e-10, atol=1e-13)
    m, c, k = filters[0].download_landmarks()
    assert np.allclose(m, o.mean, rtol=1e-9, atol=1e-11)
    assert np.allclose(c, o.cov, rtol=1e-8, atol=1e-13)
    assert np.array_equal(k, o.count)
    for f in filters:
        f.close()
