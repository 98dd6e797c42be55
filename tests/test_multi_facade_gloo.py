"""CPU, two ranks over gloo: ``FastSLAM(preset_features, devices=[...])`` -- the reference's class surface with the particles
sharded over child processes (parakeet_slam_amd/multi.py).  The per-particle arithmetic is the test-only OracleShard; the
front end, the command pipes, the replicated draw and the sharded resample are the product's.  The trajectory is the
reference's own (tests/golden/step_small.npz, captured from the unmodified prkt_core_v2.cam_cb)."""
import random

import numpy as np
import pytest

from conftest import load_golden, relerr
from sharded_common import make_oracle_shard


class View(object):
    def __init__(self, pk, blobs):
        class Scan(object):
            pass

        self.last_sensor_reading = Scan()
        obs = []
        for b in blobs:
            o = pk.msgs.Blob()
            o.bearing = float(b[0])
            o.color.r, o.color.g, o.color.b = float(b[1]), float(b[2]), float(b[3])
            obs.append(o)
        self.last_sensor_reading.observes = obs


@pytest.fixture(scope="module")
def pk():
    import parakeet_slam_amd

    return parakeet_slam_amd


@pytest.mark.parametrize("world", [2, 4])
def test_devices_keyword_runs_the_reference_trajectory_on_sharded_ranks(pk, world):
    g = load_golden("step_small")
    P, L = int(g["P"]), int(g["L"])
    seed = int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    pk.msgs.Time.set_now(0.0)
    feats = []
    for l in range(L):
        f = pk.Feature(mean=g["means0"][l], covar=g["covs0"][l])
        f.__immutable__ = bool(g["immutable"][l])
        feats.append(f)
    fs = pk.FastSLAM(feats, num_particles=P, devices=list(range(world)), weight_domain="linear", rng="global", backend="gloo",
                     _shard_factory=make_oracle_shard)
    try:
        from parakeet_slam_amd.multi import ShardedFastSLAM

        assert isinstance(fs, ShardedFastSLAM) and fs.num_particles == P and len(fs.particles) == P
        assert fs.Qt.shape == (4, 4) and isinstance(fs.last_control, pk.msgs.Twist)
        tw = pk.msgs.Twist()
        tw.linear.x = float(g["v"])
        tw.angular.z = float(g["w"])
        fs.last_control = tw
        t = 0.0
        for s in range(len(g["u"])):
            t += float(g["dts"][s])
            pk.msgs.Time.set_now(t)
            fs.cam_cb(View(pk, g["blobs"][s]))
            got = fs.download_poses()
            ref = g["post_resample"][s]
            assert np.allclose(got[:, :3], ref[:, :3], rtol=1e-12, atol=1e-13), s
            assert relerr(got[:, 3], ref[:, 3]) < 1e-11
            assert np.allclose(fs.summary(), g["summary"][s], rtol=1e-12, atol=1e-13)
            for i in (0, P // 2, P - 1):  # lazy views, fetched from the owning rank
                p = fs.particles[i]
                assert abs(p.state.pose.pose.position.x - ref[i, 0]) < 1e-12
                for l in range(L):
                    f = p.feature_set[l + 1]
                    assert relerr(f.mean, g["mean"][s][i, l]) < 1e-12
                    assert f.update_count == g["count"][s][i, l]
        # motion_update (:148-166): moves with the PREVIOUS control, then takes the new one
        before = fs.download_poses().copy()
        pk.msgs.Time.set_now(t + 0.5)
        tw2 = pk.msgs.Twist()
        tw2.linear.x = 1.0
        fs.motion_update(tw2)
        after = fs.download_poses()
        assert fs.last_control is tw2 and not np.array_equal(before[:, :2], after[:, :2])
        assert np.allclose(np.hypot(after[:, 0] - before[:, 0], after[:, 1] - before[:, 1]), 0.2 * 0.5, atol=0.05)
    finally:
        fs.close()


def test_devices_must_divide_the_particles(pk):
    with pytest.raises(ValueError):
        pk.FastSLAM([], num_particles=5, devices=[0, 1], backend="gloo", _shard_factory=make_oracle_shard)
