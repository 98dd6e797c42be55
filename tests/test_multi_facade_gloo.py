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


@pytest.mark.parametrize("world", [2, 4, 8])
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


def small_filter(pk, P=8, world=2, L=3, **kw):
    pk.msgs.Time.set_now(0.0)
    feats = [pk.Feature(mean=np.array([5.0 + l, 1.0 - l, 10.0 * l, 20.0, 30.0]), covar=0.25 * np.identity(5)) for l in range(L)]
    return pk.FastSLAM(feats, P, devices=list(range(world)), backend="gloo", _shard_factory=make_oracle_shard, **kw)


def test_devices_keyword_binds_positional_arguments_and_keeps_fastslam_defaults(pk):
    """FastSLAM(feats, 1000, devices=[0, 1]) ran with 50 particles (positional arguments were dropped) and
    with other defaults than FastSLAM's; unsupported options were ignored without a word."""
    from parakeet_slam_amd import _lib
    from parakeet_slam_amd.multi import ShardedFastSLAM

    fs = small_filter(pk, P=8)
    try:
        assert isinstance(fs, ShardedFastSLAM) and fs.num_particles == 8
        assert fs._rng == "global" and fs._domain == _lib.PK_WEIGHTS_LINEAR  # FastSLAM's defaults, not ShardedFastSLAM's
    finally:
        fs.close()
    with pytest.raises(ValueError):
        small_filter(pk, new_landmarks=True, spare_landmarks=4)
    with pytest.raises(TypeError):
        pk.FastSLAM([], 8, backend="gloo")  # (raised before anything touches a GPU)


def test_sharded_surface_motion_model_resample_and_particle_assignment(pk):
    """The rest of the class surface SURVEY 8(b) lists: motion_model (prkt_core_v2.py:168-208), low_variance_resample
    (:210-252), particles[i] = p (:162) -- on the rank that owns slot i."""
    P = 8
    fs = small_filter(pk, P=P, seed=3)
    try:
        np.random.seed(5)
        random.seed(5)
        # motion_model: a moved COPY; the filter's particles stay where they are
        before = fs.download_poses().copy()
        tw = pk.msgs.Twist()
        tw.linear.x = 1.0
        p0 = fs.particles[0]
        moved = fs.motion_model(p0, tw, pk.msgs.Duration(0.5))
        assert abs(moved.state.pose.pose.position.x - 0.5) < 0.2 and p0.state.pose.pose.position.x == 0.0
        assert np.array_equal(fs.download_poses(), before)
        # particles[i] = p: slot 5 lives on rank 1
        q = fs.particles[5]
        q.state.pose.pose.position.x = 3.25
        q.state.pose.pose.position.y = -1.5
        q.weight = 7.0
        f2 = q.feature_set[2]
        f2.mean = np.array([9.0, 8.0, 7.0, 6.0, 5.0])
        f2.update_count = 4
        fs.particles[5] = q
        got = fs.download_poses()
        assert np.allclose(got[5], [3.25, -1.5, 0.0, 7.0]) and np.array_equal(got[[0, 4, 6]], before[[0, 4, 6]])
        back = fs.particles[5].feature_set[2]
        assert np.array_equal(back.mean, f2.mean) and back.update_count == 4
        assert np.allclose(fs.particles[4].feature_set[2].mean, [6.0, 0.0, 10.0, 20.0, 30.0])
        # low_variance_resample on the weights as they stand: slot 5 weighs 7 of 14 -> it fills half of the slots (:233-250)
        fs.low_variance_resample()
        after = fs.download_poses()
        n5 = int(np.sum(after[:, 0] == 3.25))
        assert 3 <= n5 <= 5 and np.all(np.diff(np.flatnonzero(after[:, 0] == 3.25)) == 1)  # systematic: one contiguous run
        assert fs.particles[int(np.flatnonzero(after[:, 0] == 3.25)[0])].feature_set[2].update_count == 4  # the map travelled with it
        # snapshot round trip in the single-GPU facade's format
        import os
        import tempfile

        path = os.path.join(tempfile.mkdtemp(), "snap.npz")
        fs.save_state(path)
        fs.particles[0] = q
        fs.load_state(path)
        assert np.array_equal(fs.download_poses(), after)
    finally:
        fs.close()


def test_a_failing_rank_shuts_the_facade_down_instead_of_desynchronising_it(pk):
    """_collect raised at the first ERR without draining the other ranks' replies, so every later command
    read a stale reply.  Now every rank's reply is read, the facade is marked failed, the children are stopped."""
    fs = small_filter(pk)
    try:
        fs._conns[0].send(("poses",))
        fs._conns[1].send(("fail",))
        with pytest.raises(RuntimeError, match="asked to fail"):
            fs._collect("poses")
        assert fs._closed and fs._failed
        assert not any(p.is_alive() for p in fs._procs)
        with pytest.raises(RuntimeError, match="failed and was shut down"):
            fs.summary()
    finally:
        fs.close()


def test_a_dead_rank_is_reported_at_once(pk):
    import time

    fs = small_filter(pk)
    try:
        fs._procs[1].terminate()
        fs._procs[1].join(10)
        t0 = time.monotonic()
        with pytest.raises(RuntimeError, match="rank 1"):
            fs.summary()  # rank 0 waits in the all-reduce for a peer that is gone: not for the 600 s of the old poll
        assert time.monotonic() - t0 < 60
        assert not any(p.is_alive() for p in fs._procs)
    finally:
        fs.close()


def test_sharded_publishers_in_ros_mode():
    """/particle_track (:126-127), /aged_particles (:237), /resampled_particles (:242): once per particle per cam_cb from
    gathered pose downloads.  Fresh interpreter: the rospy stub must be in place before the package binds its types."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import ros_stubs
ros_stubs.install()
import rospy
from viz_feature_sim.msg import Blob, VizScan
import parakeet_slam_amd as pk
from sharded_common import make_oracle_shard
if __name__ == "__main__":
    rospy.Time.set_now(0.0)
    feats = [pk.Feature(mean=np.array([5.0, 1, 10, 20, 30]), covar=0.25 * np.identity(5))]
    fs = pk.FastSLAM(feats, 12, devices=[0, 1], backend="gloo", _shard_factory=make_oracle_shard)
    class Node: pass
    node = Node(); node.last_sensor_reading = VizScan([Blob(0.2, 10, 20, 30)])
    rospy.Time.advance(0.1)
    fs.cam_cb(node)
    ok = fs.particle_track_pub.count == 12 and fs.aged_particles_pub.count == 12 and fs.resampled_particles_pub.count == 12
    s = fs.summary()
    fs.close()
    print("SHARDED-ROS-MODE-OK" if ok else "COUNTS", fs.particle_track_pub.count, fs.aged_particles_pub.count, fs.resampled_particles_pub.count, s)
''' % (root, os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "SHARDED-ROS-MODE-OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("world", [2, 3, 8])
def test_new_landmarks_over_sharded_ranks_grow_the_same_maps_as_one_filter(pk, world, tmp_path):
    """SURVEY 8 row (f4) over several ranks: FastSLAM(new_landmarks=True, devices=[...]).  The front end, the command pipes, the
    balanced exchange and the record's tail are the product's protocol; the per-particle arithmetic is the test-only
    GrowingOracleShard.  Against ONE GrowingOracle holding every particle, same numpy / random streams: the snapshot -- poses,
    maps, potential flags, readings, id counters, slot ids, in the single filter's particle order -- must be equal."""
    from oracle.fastslam_oracle import GrowingOracle, low_variance_ancestors, synthetic_scan, synthetic_world, truth_step
    from sharded_common import make_growing_oracle_shard

    L0, U, P, spare, steps, thr = 10, 3, 12 * world, 5, 7, 30.0
    v, w, dt = 0.8, 0.35, 0.5
    world_m, covs = synthetic_world(L0 + U)
    known, kcov = world_m[:L0], covs[:L0]
    np.random.seed(5)
    random.seed(5)
    pk.msgs.Time.set_now(0.0)
    fs = pk.FastSLAM([pk.Feature(mean=m.copy(), covar=c.copy()) for m, c in zip(known, kcov)], num_particles=P,
                     devices=list(range(world)), weight_domain="log", rng="global", backend="gloo", new_landmarks=True,
                     spare_landmarks=spare, pair_threshold=thr, _shard_factory=make_growing_oracle_shard)
    try:
        tw = pk.msgs.Twist()
        tw.linear.x, tw.angular.z = v, w
        fs.last_control = tw
        o = GrowingOracle(P, known, kcov, spare, thr)
        np_rs, py_rs = np.random.RandomState(5), random.Random(5)
        path = str(tmp_path / "grow.npz")

        def same_as_the_one_filter():
            fs.save_state(path)
            d = np.load(path)
            assert np.allclose(d["poses"][:, :3], np.stack([o.f.x, o.f.y, o.f.h], 1), rtol=1e-12, atol=1e-13)
            assert np.allclose(d["means"], o.f.mean, rtol=1e-10, atol=1e-10)
            assert np.array_equal(d["counts"] & ~0x40000000, o.f.count) and np.array_equal((d["counts"] & 0x40000000) != 0, o.f.potential)
            assert d["nl_next_id"].tolist() == o.next_id and d["nl_used"].tolist() == o.used
            assert np.diff(d["nl_offsets"]).tolist() == [len(h) for h in o.hyp]
            flat = np.array([rd for h in o.hyp for rd in h]).reshape(-1, 8)
            assert np.allclose(d["nl_readings"], flat, rtol=1e-12, atol=1e-13)
            assert [tuple(t) for t in d["nl_slot_id"].tolist()] == [(i, sl, v_) for i, m in enumerate(o.slot_id) for sl, v_ in sorted(m.items())]
            return len(set(zip(o.used, [len(h) for h in o.hyp], o.next_id)))

        kinds = 0
        pose = (0.0, 0.0, 0.0)
        for s in range(steps):
            pose = truth_step(pose, v, w, dt)
            blobs = synthetic_scan(world_m, pose)
            pk.msgs.Time.set_now(dt * (s + 1))
            fs.cam_cb(View(pk, blobs))
            o.f.reset_weights()
            o.f.motion(v, w, dt, np_rs.standard_normal((P, 3)))
            o.observe(blobs)
            wts = np.exp(o.f.logw - o.f.logw.max())
            o.gather(low_variance_ancestors(wts, py_rs.random()))
            if s in (1, 3):
                kinds = max(kinds, same_as_the_one_filter())
        kinds = max(kinds, same_as_the_one_filter())
        assert max(o.used) >= 2 and kinds > 1, "the scene shows nothing: every particle held the same bookkeeping at every check"
        assert fs.readings_dropped() == 0
        # the object view of a particle wherever it lives now (:278-292)
        i = P - 2
        p = fs.particles[i]
        assert p.next_id == o.next_id[i] and len(p.hypothesis_set) == len(o.hyp[i])
        assert len(p.potential_features) + len([k for k in p.feature_set.keys() if k > L0]) == o.used[i]
        # the snapshot goes back in (every rank takes its rows again) and the next step still agrees
        fs.load_state(path)
        pose = truth_step(pose, v, w, dt)
        blobs = synthetic_scan(world_m, pose)
        pk.msgs.Time.set_now(dt * (steps + 1))
        fs.cam_cb(View(pk, blobs))
        o.f.reset_weights()
        o.f.motion(v, w, dt, np_rs.standard_normal((P, 3)))
        o.observe(blobs)
        wts = np.exp(o.f.logw - o.f.logw.max())
        o.gather(low_variance_ancestors(wts, py_rs.random()))
        fs.save_state(path)
        d = np.load(path)
        assert d["nl_next_id"].tolist() == o.next_id and d["nl_used"].tolist() == o.used
        assert np.allclose(d["means"], o.f.mean, rtol=1e-10, atol=1e-10)
    finally:
        fs.close()
