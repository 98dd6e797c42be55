"""GPU: the facade under the reference's two threads.  rospy runs the /cmd_vel callback -- CamSlam360.motion_update ->
FastSLAM.motion_update (prkt_ros.py:113-121) -- on a subscriber thread while the main loop is inside cam_cb replacing particles
(prkt_core_v2.py:148-166 against :59-137, :252): a real data race in the reference, no lock anywhere.  The facade
holds one re-entrant lock per filter (core.py, multi.py); the C ABI is single-caller per handle.  What the lock promises is that a run
with two threads equals SOME serial interleaving of its calls: the order in which the calls got the lock is recorded, replayed by ONE
thread on a fresh filter, and the two final states must agree bit for bit."""
import random
import threading
import time

import numpy as np
import pytest

from oracle.fastslam_oracle import synthetic_scan, synthetic_world, truth_step

pytestmark = pytest.mark.gpu


class _View(object):
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


class _TickClock(object):
    """rospy.Time.now() (prkt_core_v2.py:158) as a counter: every reading is 13 ms after the one before it.  Both callers read it
    inside the filter's lock, so the k-th reading belongs to the k-th call that got the lock -- in the threaded run and in its replay."""

    def __init__(self, pk):
        self.pk, self.t, self.lock = pk, 0.0, threading.Lock()

    def install(self):
        clock = self

        def now(cls):
            with clock.lock:
                clock.t += 0.013
                return cls(clock.t)

        self._old = self.pk.msgs.Time.__dict__["now"]
        self.pk.msgs.Time.now = classmethod(now)

    def remove(self):
        self.pk.msgs.Time.now = self._old


def _logged(fs, log):
    """Every entry point of the boundary (prkt_ros.py:84, :94, :118) notes itself once it HOLDS the filter's lock (re-entrant)."""
    cam, motion, summary = fs.cam_cb, fs.motion_update, fs.summary

    def cam_cb(view):
        with fs._lock:
            log.append(("cam", view.index))
            return cam(view)

    def motion_update(tw):
        with fs._lock:
            log.append(("motion", float(tw.linear.x), float(tw.angular.z)))
            return motion(tw)

    def summary_():
        with fs._lock:
            log.append(("summary",))
            return summary()

    return cam_cb, motion_update, summary_


def _twist(pk, v, w):
    tw = pk.msgs.Twist()
    tw.linear.x, tw.angular.z = v, w
    return tw


def _final_state(fs, sharded):
    if sharded:
        poses = fs.download_poses() if hasattr(fs, "download_poses") else None
        parts = fs.particles
        picks = [0, len(parts) // 2, len(parts) - 1]
        st = [np.array([parts[i].state.pose.pose.position.x, parts[i].state.pose.pose.position.y, parts[i].weight]) for i in picks]
        maps = [np.asarray(parts[i].feature_set[3].mean, dtype=np.float64).copy() for i in picks]
        covs = [np.asarray(parts[i].feature_set[3].covar, dtype=np.float64).copy() for i in picks]
        return dict(poses=poses, st=st, maps=maps, covs=covs)
    f = fs._filter
    m, c, k = f.download_landmarks()
    return dict(poses=f.download_poses(), logw=f.download_log_weights(), m=m, c=c, k=k,
                anc=None if fs.last_ancestors is None else np.asarray(fs.last_ancestors).copy())


def _same(a, b):
    assert a.keys() == b.keys()
    for key in a:
        x, y = a[key], b[key]
        if x is None or y is None:
            assert x is None and y is None, key
        elif isinstance(x, list):
            for u, v in zip(x, y):
                assert np.array_equal(u, v), key
        else:
            assert np.array_equal(x, y), key


def _run_threaded_then_replay(pk, make, sharded, steps, scans):
    clock = _TickClock(pk)
    clock.install()
    try:
        # ---- two threads
        random.seed(11)
        clock.t = 0.0
        fs = make()
        fs.last_control = _twist(pk, 0.2, 0.1)
        log, sums = [], []
        cam_cb, motion_update, summary = _logged(fs, log)
        stop, errors = threading.Event(), []

        def driver():  # simple_driver.py:15-24 publishes a Twist at 11 Hz; here faster, and never the same one twice
            k = 0
            try:
                while not stop.is_set():
                    motion_update(_twist(pk, 0.2 + 0.01 * (k % 7), 0.1 - 0.02 * (k % 5)))
                    k += 1
                    time.sleep(0.0009)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        th = threading.Thread(target=driver, name="cmd_vel")
        th.start()
        try:
            for s in range(steps):
                v = _View(pk, scans[s])
                v.index = s
                cam_cb(v)
                sums.append(summary())
                time.sleep(0.001)
        finally:
            stop.set()
            th.join(timeout=60)
        assert not errors, errors
        assert not th.is_alive()
        n_motion = sum(1 for e in log if e[0] == "motion")
        assert n_motion >= 10, "the second thread hardly ran: %d motion updates beside %d steps" % (n_motion, steps)
        # (the calls did interleave: motion updates stand between the steps, not all in front of or behind them)
        kinds = [e[0] for e in log]
        first_cam, last_cam = kinds.index("cam"), len(kinds) - 1 - kinds[::-1].index("cam")
        assert "motion" in kinds[first_cam:last_cam]
        threaded = _final_state(fs, sharded)
        fs.close()
        # ---- the same calls, in the order they got the lock, by one thread
        random.seed(11)
        clock.t = 0.0
        fs2 = make()
        fs2.last_control = _twist(pk, 0.2, 0.1)
        sums2 = []
        for e in log:
            if e[0] == "cam":
                v = _View(pk, scans[e[1]])
                v.index = e[1]
                fs2.cam_cb(v)
            elif e[0] == "motion":
                fs2.motion_update(_twist(pk, e[1], e[2]))
            else:
                sums2.append(fs2.summary())
        replayed = _final_state(fs2, sharded)
        fs2.close()
        assert len(sums) == len(sums2) == steps
        assert np.array_equal(np.array(sums), np.array(sums2))  # every summary() the main thread read, bit for bit
        _same(threaded, replayed)
    finally:
        clock.remove()


def _scans(L, steps):
    means, covs = synthetic_world(L)
    pose, scans = (0.0, 0.0, 0.0), []
    for _ in range(steps):
        pose = truth_step(pose, 0.2, 0.1, 0.1)
        scans.append(synthetic_scan(means, pose))
    return means, covs, scans


def test_motion_update_from_a_second_thread_while_cam_cb_runs_equals_a_serial_interleaving():
    import parakeet_slam_amd as pk

    L, P, steps = 60, 20000, 50
    means, covs, scans = _scans(L, steps)
    feats = [pk.Feature(mean=means[l].copy(), covar=covs[l].copy()) for l in range(L)]
    _run_threaded_then_replay(pk, lambda: pk.FastSLAM(feats, num_particles=P, device=0, weight_domain="log", rng="device", seed=3), False, steps, scans)


def test_the_same_on_the_several_gpu_facade_two_ranks_on_one_gpu():
    import parakeet_slam_amd as pk

    L, P, steps = 40, 4096, 50
    means, covs, scans = _scans(L, steps)
    feats = [pk.Feature(mean=means[l].copy(), covar=covs[l].copy()) for l in range(L)]

    def make():
        fs = pk.FastSLAM(feats, num_particles=P, devices=[0, 0], weight_domain="log", rng="device", seed=3, backend="gloo")
        assert isinstance(fs, pk.ShardedFastSLAM)
        return fs

    _run_threaded_then_replay(pk, make, True, steps, scans)
