"""The reference's ``FastSLAM`` class surface over several GPUs: ``ShardedFastSLAM`` (also reached as
``FastSLAM(preset_features, devices=[0, 1, ...])``).

prkt_ros.py drives the core through four calls -- ``FastSLAM(preset_features)`` :91, ``core.cam_cb(self)`` :84,
``core.motion_update(msg)`` :118, ``core.summary()`` :94 -- from ONE process.  Here that process stays the front end and
never touches a GPU: it starts one child process per device (``multiprocessing`` "spawn": a fresh interpreter, started
BEFORE anything in this process could have initialised HIP; nothing is ever ``exec``-ed), each child owns one contiguous
shard of the particles with their whole maps (``ShardedFilter``, sharded.py: RCCL over xGMI between the children) and
serves commands from a pipe:

    step          one cam_cb (:59-137): weight reset, motion with the previous control (:163), association + EKF + weights,
                  global systematic resample -- the draw u (``random.random()``, :226) is made HERE and sent to every rank
                  (the replicated draw of the north star), as are the scan, Qt and, in the reference's RNG mode, each
                  shard's slice of the ``numpy.random`` normals (:185-193)
    observe       the same without the resample (the debug publishers of :126-127, :237 want the poses in between)
    resample      low_variance_resample (:210-252) on the weights as they stand
    motion        motion_update (:148-166)
    summary       (:254-276): every rank all-reduces four pose sums; rank 0 answers
    poses / landmarks / set_particle / set_state   views for ``particles[i]`` (lazy, like the single-GPU facade),
                  ``particles[i] = p`` (:162) on the owning rank, snapshots
    motion_model  :168-208 on one host particle (rank 0's device; nothing of the filter changes)

Particles are independent until the resample (:216-252), so nothing else crosses ranks.

Failure handling: a rank that raises answers ("ERR", traceback); the front end always reads ONE reply from EVERY rank
before it looks at any of them, so the pipes never fall out of step; on any error, timeout or dead child it marks itself
failed, stops the children (a rank that failed before a collective leaves its peers blocked inside RCCL: they are
terminated by handle) and every later call raises.
"""
from __future__ import annotations

import copy as _copy
import multiprocessing as mp
import multiprocessing.connection as mpc
import os
import random as _pyrandom
import tempfile
import threading
import time
import traceback

import numpy as np

from . import msgs
from .msgs import Odometry, Twist


def _worker_main(rank, world, device, store_path, backend, P_local, L, means, covs, imm, domain, shard_factory, conn, grow=None):
    """One rank: joins the process group, builds its shard, serves the pipe.  Runs in a freshly spawned interpreter."""
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on these hosts
        import torch
        import torch.distributed as dist

        if backend == "nccl":
            torch.cuda.set_device(device)
        dist.init_process_group(backend, store=dist.FileStore(store_path, world), rank=rank, world_size=world)
        from .sharded import ShardedFilter, TorchComm

        shard = shard_factory(P_local, means, covs, imm) if shard_factory is not None else None
        # (a world of one defaults to the contiguous placement, which the new-landmark bookkeeping refuses -- the
        # bookkeeping rides behind the records of the balanced exchange: ask for that placement whenever the maps grow)
        sf = ShardedFilter(P_local, L, device=device, comm=TorchComm(), shard=shard, placement="balanced" if grow is not None else None)
        if L:
            sf.upload_map(means, covs.reshape(L, 25), imm)
        if grow is not None:  # (preset landmarks, reading capacity, pair threshold): section 8(f4) on every rank's device
            sf.grow_enable(*grow)
        probe = None  # a one-particle filter for motion_model, made on first use and kept
        conn.send(("ok", None))
        while True:
            cmd = conn.recv()
            op = cmd[0]
            try:
                if op == "step":
                    _, v, w, dt, z, seed, draw, blobs, qt, u = cmd
                    sf.set_measurement_noise(qt)
                    sf.step(v, w, dt, blobs, u, z=z, seed=seed, draw=draw, domain=domain)
                    conn.send(("ok", None))
                elif op == "observe":
                    _, v, w, dt, z, seed, draw, blobs, qt = cmd
                    sf.set_measurement_noise(qt)
                    sf.motion(v, w, dt, z=z, seed=seed, draw=draw)
                    sf.observe(blobs, fresh=True)
                    conn.send(("ok", None))
                elif op == "resample":
                    sf.resample(cmd[1], domain=domain)
                    conn.send(("ok", None))
                elif op == "motion":
                    _, v, w, dt, z, seed, draw = cmd
                    sf.motion(v, w, dt, z=z, seed=seed, draw=draw)
                    conn.send(("ok", None))
                elif op == "summary":
                    conn.send(("ok", sf.summary()))
                elif op == "poses":  # with the logical index of every slot: the front end answers in the single filter's order
                    conn.send(("ok", (sf.download_poses(), sf.logical_index())))
                elif op == "landmarks":
                    conn.send(("ok", sf.download_landmarks(cmd[1], cmd[2])))
                elif op == "grow":  # the new-landmark bookkeeping of local slots [j0, j1)
                    conn.send(("ok", sf.grow_download(cmd[1], cmd[2])))
                elif op == "set_grow":
                    sf.grow_upload(0, P_local, *cmd[1:4])
                    conn.send(("ok", None))
                elif op == "set_particle":
                    _, j, pose, m, c, k = cmd
                    sf.upload_pose(j, pose)  # (one particle: the shard's other log-weights are not sent through exp and log)
                    if m is not None:
                        sf.upload_landmarks(j, j + 1, m, c, k)
                    conn.send(("ok", None))
                elif op == "set_state":
                    _, poses, m, c, k = cmd
                    sf.upload_poses(poses)
                    if m is not None:
                        sf.upload_landmarks(0, P_local, m, c, k)
                    sf.reset_placement()  # the snapshot's order: rank r holds the logical particles [r P, (r + 1) P) again
                    conn.send(("ok", None))
                elif op == "motion_model":
                    _, pose, v, w, dt, z, seed, draw = cmd
                    if probe is None:
                        if shard_factory is not None:
                            probe = shard_factory(1, means, covs, imm)
                        else:
                            from . import _lib

                            probe = _lib.DeviceFilter(1, 0, device=device)
                    probe.upload_poses(np.array([[pose[0], pose[1], pose[2], 1.0]]))
                    probe.motion(v, w, dt, z=z, seed=seed, draw=draw)
                    conn.send(("ok", probe.download_poses()[0]))
                elif op == "fail":  # tests: a rank that raises in the middle of a command
                    raise RuntimeError("rank %d was asked to fail" % rank)
                elif op == "close":
                    conn.send(("ok", None))
                    break
                else:
                    conn.send(("ERR", "unknown command %r" % (op,)))
            except Exception:  # noqa: BLE001 -- reported to the front end, which raises
                conn.send(("ERR", traceback.format_exc()))
        if probe is not None and hasattr(probe, "close"):
            probe.close()
        if hasattr(sf.f, "close"):
            sf.f.close()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        try:
            conn.send(("ERR", traceback.format_exc()))
        except Exception:  # noqa: BLE001
            pass


class _ShardedParticleList(object):
    """``fs.particles``: list-like views fetched from the owning rank on demand; ``particles[i] = p`` (:162) writes the
    host particle's pose, weight and landmark estimates into slot i on the rank that owns it."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return self._o.num_particles

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def _index(self, i):
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("particle index out of range")
        return i

    def __getitem__(self, i):
        return self._o._particle_view(self._index(i))

    def __setitem__(self, i, particle):
        self._o._store_particle(self._index(i), particle)

    def __repr__(self):
        return "<%d particles on GPUs %s>" % (len(self), self._o.devices)


class ShardedFastSLAM(object):
    """prkt_core_v2.py:37-276 with the particles sharded over ``devices`` (one child process per GPU).

    ShardedFastSLAM(preset_features=[], num_particles=50, devices=(0, 1), weight_domain="log", rng="device", seed=0,
                    backend="nccl", publish_debug=None)

    ``num_particles`` must be a multiple of ``len(devices)``.  rng as in ``FastSLAM``: "device" (Philox per global particle
    index: the noise does not depend on the shard count) or "global" (the reference's own ``numpy.random`` / ``random``
    streams, drawn here and handed to the shards).  publish_debug as in ``FastSLAM``: the three debug publishers of
    :55-57 (/particle_track :126-127, /aged_particles :237, /resampled_particles :242) fed from pose downloads gathered over
    the ranks.  ``backend="gloo"`` with ``_shard_factory`` is the CPU rehearsal the tests use.

    new_landmarks=True with spare_landmarks=S: the working new-landmark initialisation of ``FastSLAM`` (:546-746, core.py) on
    every rank's device (pk_grow_enable); a particle that migrates in the resample takes its orphaned readings, id counter and
    spare-slot ids along behind its map, so the filter grows the same maps whatever the number of GPUs.

    (``FastSLAM(..., devices=[...])`` arrives here with FastSLAM's OWN defaults -- weight_domain "linear", rng "global".)"""

    def __init__(self, preset_features=[], num_particles=50, devices=(0, 1), weight_domain="log", rng="device", seed=0,
                 backend="nccl", publish_debug=None, _shard_factory=None, new_landmarks=False, spare_landmarks=0,
                 pair_threshold=30.0, reading_capacity=64):
        from . import _lib  # constants only; loads nothing

        self._lock = threading.RLock()
        self.last_control = Twist()
        self.last_update = msgs.now()
        self.devices = [int(d) for d in devices]
        world = len(self.devices)
        if world < 1:
            raise ValueError("devices must name at least one GPU")
        self.num_particles = int(num_particles)
        if self.num_particles % world:
            raise ValueError("num_particles (%d) must be a multiple of the number of devices (%d)" % (self.num_particles, world))
        if rng not in ("global", "device"):
            raise ValueError("rng must be 'global' or 'device'")
        self._P_local = self.num_particles // world
        self._features = list(preset_features)
        L0 = len(self._features)
        self._grow = bool(new_landmarks) and int(spare_landmarks) > 0
        self._spare = int(spare_landmarks) if self._grow else 0
        if self._grow and _shard_factory is not None and not getattr(_shard_factory, "grows", False):
            raise ValueError("ShardedFastSLAM: new_landmarks needs shards that keep the bookkeeping (the HIP shards do; this "
                             "_shard_factory does not say it does)")
        self._L0 = L0
        L = L0 + self._spare  # every particle's map: the preset landmarks, then the spare slots
        self._L = L
        means = np.zeros((L, 5))
        covs = np.tile(np.identity(5), (L, 1, 1))
        imm = np.zeros(L, dtype=np.uint8)
        if L0:
            means[:L0] = np.array([np.asarray(f.mean, dtype=np.float64).reshape(5) for f in self._features])
            covs[:L0] = np.array([np.asarray(f.covar, dtype=np.float64).reshape(5, 5) for f in self._features])
            imm[:L0] = np.array([bool(f.__immutable__) for f in self._features], dtype=np.uint8)
        from .core import FastSLAM as _Single

        means[L0:, 2:] = _Single.EMPTY_COLOUR  # a spare slot that holds nothing yet fails every colour gate
        self._reading_capacity = int(reading_capacity)
        grow = (L0, int(reading_capacity), float(pair_threshold)) if self._grow else None
        self.Qt = np.identity(4) * 0.1  # prkt_core_v2.py:50-53
        self._domain = {"linear": _lib.PK_WEIGHTS_LINEAR, "log": _lib.PK_WEIGHTS_LOG}[weight_domain]
        self._rng, self._seed, self._draw = rng, int(seed), 0
        self._pose_cache = None
        r = msgs.ros()
        self._publish = (r is not None) if publish_debug is None else bool(publish_debug)
        if r is not None:  # :55-57
            self.aged_particles_pub = r.Publisher('/aged_particles', Odometry, queue_size=1)
            self.resampled_particles_pub = r.Publisher('/resampled_particles', Odometry, queue_size=1)
            self.particle_track_pub = r.Publisher('/particle_track', Odometry, queue_size=1)
        else:
            self.aged_particles_pub = self.resampled_particles_pub = self.particle_track_pub = None
            self._publish = False
        ctx = mp.get_context("spawn")  # fresh interpreters: no child inherits (or replaces) a process that touched a GPU
        fd, self._store = tempfile.mkstemp(prefix="pk_store_")
        os.close(fd)
        os.unlink(self._store)
        self._conns, self._procs = [], []
        self._closed, self._failed = False, None
        self._unsent = set()
        for rk, dev in enumerate(self.devices):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(rk, world, dev, self._store, backend, self._P_local, L, means, covs, imm,
                                                       self._domain, _shard_factory, b, grow), daemon=True)
            p.start()
            self._conns.append(a)
            self._procs.append(p)
        self._collect("start")
        self.particles = _ShardedParticleList(self)

    # ------------------------------------------------------------------ plumbing
    def _check_open(self):
        if self._failed is not None:
            raise RuntimeError("this ShardedFastSLAM has failed and was shut down: %s" % self._failed)
        if self._closed:
            raise RuntimeError("this ShardedFastSLAM is closed")

    def _post(self, r, cmd):
        """Send one command to rank r; a pipe that is gone (the child died) is remembered for _collect."""
        try:
            self._conns[r].send(cmd)
        except (BrokenPipeError, OSError):
            self._unsent.add(r)

    def _collect(self, what, timeout=600.0, ranks=None):
        """ONE reply from every rank asked, whatever the others say; then errors, if any.  A dead child is noticed at once
        (its pipe would stay silent for the whole timeout)."""
        ranks = list(range(len(self._conns))) if ranks is None else list(ranks)
        replies, errors = {}, []
        for r in sorted(self._unsent.intersection(ranks)):
            errors.append("rank %d is gone: '%s' could not be sent (exit code %r)" % (r, what, self._procs[r].exitcode))
        pending = {r: self._conns[r] for r in ranks if r not in self._unsent}
        self._unsent = set()
        deadline = time.monotonic() + timeout
        while pending:
            ready = mpc.wait(list(pending.values()), timeout=0.5)
            for c in ready:
                r = next(k for k, v in pending.items() if v is c)
                try:
                    replies[r] = c.recv()
                except (EOFError, OSError):
                    errors.append("rank %d closed its pipe during '%s' (exit code %r)" % (r, what, self._procs[r].exitcode))
                del pending[r]
            if ready:
                continue
            for r in list(pending):
                if not self._procs[r].is_alive() and not pending[r].poll(0):
                    errors.append("rank %d died during '%s' (exit code %r)" % (r, what, self._procs[r].exitcode))
                    del pending[r]
            if errors and pending:
                # a rank is gone or has failed: its peers may be waiting for it inside a collective -- give them a moment, no more
                deadline = min(deadline, time.monotonic() + 5.0)
            if pending and time.monotonic() > deadline:
                for r in pending:
                    errors.append("rank %d did not answer '%s' in time" % (r, what))
                pending.clear()
            if not errors:  # a reported failure shortens everybody else's wait as well
                for r, rep in replies.items():
                    if rep[0] != "ok":
                        deadline = min(deadline, time.monotonic() + 5.0)
                        break
        for r in sorted(replies):
            status, payload = replies[r]
            if status != "ok":
                errors.append("rank %d failed in '%s':\n%s" % (r, what, payload))
        if errors:
            self._failed = "; ".join(e.splitlines()[0] for e in errors)
            self._shutdown(graceful=False)
            raise RuntimeError("\n".join(errors))
        return [replies[r][1] for r in ranks]

    def _all(self, *cmd):
        self._check_open()
        for r in range(len(self._conns)):
            self._post(r, cmd)
        return self._collect(cmd[0])

    def _one(self, r, *cmd):
        self._check_open()
        self._post(r, cmd)
        return self._collect(cmd[0], ranks=[r])[0]

    def _noise(self):
        if self._rng == "global":  # numpy.random.normal(0, s, 1) x 3 per particle, particle-major (:185-193)
            # the whole filter's normals in the reference's (= logical) particle order: every rank takes the rows of the
            # particles it holds (ShardedFilter.motion)
            z = np.random.standard_normal((self.num_particles, 3))
            return [z] * len(self.devices)
        return [None] * len(self.devices)

    def _publish_all(self, pub, poses):
        if not self._publish or pub is None:
            return
        from .core import _make_state

        for x, y, h, _w in poses:
            st = _make_state(x, y, h)
            st.header.frame_id = 'odom'
            pub.publish(st)

    # ------------------------------------------------------------------ the reference's surface
    def cam_cb(self, ros_view):
        """One filter step (:59-137)."""
        with self._lock:
            self._check_open()
            dt = msgs.now() - self.last_update  # :158, the motion update of :75-77
            v, w = float(self.last_control.linear.x), float(self.last_control.angular.z)
            from .core import _blob_matrix

            blobs = _blob_matrix(ros_view.last_sensor_reading.observes)  # :82
            zs = self._noise()
            qt = np.asarray(self.Qt, dtype=np.float64).reshape(4, 4)
            if self._publish:
                # the debug publishers want the poses between the weighting and the resample: two commands, and ONE gathered
                # pose download that feeds /particle_track (:126-127) and /aged_particles (:237)
                for r in range(len(self._conns)):
                    self._post(r, ("observe", v, w, float(dt.to_sec()), zs[r], self._seed, self._draw, blobs, qt))
                self._collect("observe")
                self._pose_cache = None
                self._publish_all(self.particle_track_pub, self.download_poses())
                self._draw += 1
                self.last_update = self.last_update + dt
                self.low_variance_resample()
                return
            u = _pyrandom.random()  # :226 -- drawn once, here, and replicated to every rank
            for r in range(len(self._conns)):
                self._post(r, ("step", v, w, float(dt.to_sec()), zs[r], self._seed, self._draw, blobs, qt, float(u)))
            self._collect("step")
            self._draw += 1
            self.last_update = self.last_update + dt  # :165 (last_control stays: cam_cb moves with it, :77)
            self._pose_cache = None

    def motion_update(self, new_twist):
        """:148-166: moves every particle with the PREVIOUS control, then takes the new one."""
        with self._lock:
            self._check_open()
            dt = msgs.now() - self.last_update
            v, w = float(self.last_control.linear.x), float(self.last_control.angular.z)
            zs = self._noise()
            for r in range(len(self._conns)):
                self._post(r, ("motion", v, w, float(dt.to_sec()), zs[r], self._seed, self._draw))
            self._collect("motion")
            self._draw += 1
            self.last_update = self.last_update + dt
            self.last_control = new_twist
            self._pose_cache = None

    def motion_model(self, particle, twist, dt):
        """Move ONE host particle (:168-208) and return the moved copy; the filter's particles are not touched.  Runs the
        motion kernel on a one-particle filter of rank 0's device."""
        from .core import _state_pose
        from .msgs import heading_to_quaternion

        with self._lock:
            x, y, h = _state_pose(particle.state)
            z = np.random.standard_normal((1, 3)) if self._rng == "global" else None
            nx, ny, nh, _ = self._one(0, "motion_model", (x, y, h), float(twist.linear.x), float(twist.angular.z),
                                      float(dt.to_sec()), z, self._seed, self._draw)
            self._draw += 1
        new_particle = _copy.deepcopy(particle)
        new_particle.state = _copy.deepcopy(particle.state)
        new_particle.state.pose.pose.position.x = float(nx)
        new_particle.state.pose.pose.position.y = float(ny)
        new_particle.state.pose.pose.orientation = heading_to_quaternion(float(nh))
        return new_particle

    def low_variance_resample(self):
        """:210-252 over all shards on the weights as they stand: one draw u (:226), made here, replicated to every rank."""
        with self._lock:
            self._check_open()
            if self._publish:
                self._publish_all(self.aged_particles_pub, self.download_poses())  # :237
            u = _pyrandom.random()
            self._all("resample", float(u))
            self._pose_cache = None
            if self._publish:
                self._publish_all(self.resampled_particles_pub, self.download_poses())  # :242

    def summary(self):
        """:254-276 over all shards."""
        with self._lock:
            return tuple(self._all("summary")[0])

    def odom_motion_update(self, odom):
        pass  # :140-146 alpha feature, empty in the reference

    # ------------------------------------------------------------------ views
    def download_poses(self):
        """(P, 4) poses and weights in the particle order of ONE filter (prkt_core_v2.py:43-44's list): the ranks answer with the
        logical index of every slot they hold (balanced placement, sharded.py), the rows are put in that order."""
        with self._lock:
            if self._pose_cache is None:
                parts = self._all("poses")
                logical = np.concatenate([np.asarray(p[1], dtype=np.int64) for p in parts])
                if not np.array_equal(np.sort(logical), np.arange(self.num_particles)):
                    raise RuntimeError("the ranks' logical indices are not a permutation of the particles")
                poses = np.empty((self.num_particles, 4))
                poses[logical] = np.concatenate([p[0] for p in parts])
                self._where = np.empty(self.num_particles, dtype=np.int64)  # logical index -> physical place rank * P_local + j
                self._where[logical] = np.arange(self.num_particles)
                self._pose_cache = poses
            return self._pose_cache

    def _place(self, i):
        """(rank, slot) that holds logical particle i now."""
        self.download_poses()
        return divmod(int(self._where[i]), self._P_local)

    def _particle_view(self, i):
        from .core import Feature, FilterParticle, _FeatureSet, _make_state

        with self._lock:
            x, y, h, w = self.download_poses()[i]
            p = FilterParticle(_make_state(x, y, h))
            p.weight = float(w)
            p.Qt = self.Qt
            r, j = self._place(i)
            feats, lock, L0, one = self._features, self._lock, self._L0, self._one
            slot_id = {}
            if self._grow:
                from . import _lib
                from .core import _nl_unpack
                from .msgs import Blob

                nl = _nl_unpack(*one(r, "grow", j, j + 1), L0=L0)
                p.next_id, slot_id = nl["next_id"][0], nl["slot_id"][0]
                for rd in nl["hyp"][0]:
                    b = Blob()
                    b.bearing = rd[4]
                    b.color.r, b.color.g, b.color.b = rd[5], rd[6], rd[7]
                    p.hypothesis_set[rd[0]] = (_make_state(rd[1], rd[2], rd[3]), b)
            else:
                p.next_id = L0 + 1
            pot_bit = 0x40000000  # PK_LANDMARK_POTENTIAL

            def load_all():
                with lock:
                    m, c, k = one(r, "landmarks", j, j + 1)
                full, potential = {}, {}
                for l in range(L0):
                    f = Feature(mean=m[0, l], covar=c[0, l])
                    f.update_count = int(k[0, l]) & ~pot_bit
                    f.__immutable__ = bool(feats[l].__immutable__)
                    full[l + 1] = f
                for slot, id_ in slot_id.items():
                    f = Feature(mean=m[0, slot], covar=c[0, slot])
                    f.update_count = int(k[0, slot]) & ~pot_bit
                    if int(k[0, slot]) & pot_bit:
                        potential[-id_] = f  # :685
                    else:
                        full[id_] = f        # promoted (:115-116)
                return full, potential

            p.feature_set = _FeatureSet(lambda: load_all()[0])
            if slot_id:
                p.potential_features = load_all()[1]
            return p

    def _store_particle(self, i, particle):
        """``fs.particles[i] = particle`` (:162): pose, weight and -- when the particle carries all L of them -- landmark
        estimates go to slot i of the rank that owns it."""
        from .core import _state_pose

        with self._lock:
            x, y, h = _state_pose(particle.state)
            r, j = self._place(i)
            L = self._L0
            m = c = k = None
            ids = range(1, L + 1)
            if L and all(q in particle.feature_set for q in ids):
                fs_ = [particle.feature_set[q] for q in ids]
                m = np.array([np.asarray(f.mean, dtype=np.float64) for f in fs_]).reshape(1, L, 5)
                c = np.array([np.asarray(f.covar, dtype=np.float64) for f in fs_]).reshape(1, L, 25)
                k = np.array([int(f.update_count) for f in fs_], dtype=np.int32).reshape(1, L)
                if self._spare:  # the preset landmarks only: the spare slots keep what the filter put there
                    m0, c0, k0 = self._one(r, "landmarks", j, j + 1)
                    c0 = c0.reshape(1, -1, 25)
                    m0[:, :L], c0[:, :L], k0[:, :L] = m, c, k
                    m, c, k = m0, c0, k0
            self._one(r, "set_particle", j, (x, y, h, float(particle.weight)), m, c, k)
            self._pose_cache = None

    def readings_dropped(self):
        """As ``FastSLAM.readings_dropped``: orphaned readings that found a particle's ring full, over all ranks (0 = the device
        bookkeeping is the reference's)."""
        if not self._grow:
            return 0
        with self._lock:
            return int(sum(int(np.asarray(g[0])[:, 3].sum()) for g in self._all("grow", 0, self._P_local)))

    # ------------------------------------------------------------------ snapshot / restore (the single-GPU facade's format)
    def save_state(self, path):
        with self._lock:
            poses = self.download_poses()
            parts = self._all("landmarks", 0, self._P_local)
            m = np.concatenate([p[0] for p in parts])[self._where]  # the single filter's particle order
            c = np.concatenate([p[1] for p in parts])[self._where]
            k = np.concatenate([p[2] for p in parts])[self._where]
            extra = {}
            if self._grow:  # the single-GPU facade's nl_* arrays, in the single filter's particle order
                from .core import _nl_snapshot_arrays

                g = self._all("grow", 0, self._P_local)
                extra = _nl_snapshot_arrays(*[np.concatenate([q[a] for q in g])[self._where] for a in range(3)], L0=self._L0)
            np.savez_compressed(
                path, poses=poses, means=m, covs=c, counts=k, Qt=np.asarray(self.Qt, dtype=np.float64),
                immutable=np.array([bool(f.__immutable__) for f in self._features], dtype=np.uint8),
                last_control=np.array([float(self.last_control.linear.x), float(self.last_control.angular.z)]),
                last_update=float(self.last_update.to_sec()), draw=self._draw, **extra)

    def load_state(self, path):
        with self._lock:
            self._check_open()
            d = np.load(path, allow_pickle=False)
            P, L, Pl = self.num_particles, self._L, self._P_local
            if d["poses"].shape != (P, 4) or d["means"].shape != (P, L, 5):
                raise ValueError("snapshot is for %s particles x %s landmarks, this filter has %d x %d"
                                 % (d["poses"].shape[0], d["means"].shape[1], P, L))
            nl = None
            if self._grow:
                from .core import _nl_arrays_from_snapshot, _nl_check_snapshot

                if "nl_offsets" not in d.files:
                    raise ValueError("snapshot: no new-landmark bookkeeping (nl_* arrays) for a filter with new_landmarks=True")
                _nl_check_snapshot(d, P, self._L0, L, self._spare, self._reading_capacity)  # before anything is assigned
                nl = _nl_arrays_from_snapshot(d, P, self._L0, self._spare, self._reading_capacity)
            for r in range(len(self._conns)):
                s = slice(r * Pl, (r + 1) * Pl)
                self._post(r, ("set_state", d["poses"][s], d["means"][s] if L else None,
                           d["covs"][s].reshape(Pl, L, 25) if L else None, d["counts"][s].astype(np.int32) if L else None))
            self._collect("set_state")
            if nl is not None:  # (set_state put rank r's slots back in the snapshot's order)
                for r in range(len(self._conns)):
                    s = slice(r * Pl, (r + 1) * Pl)
                    self._post(r, ("set_grow", nl[0][s], nl[1][s], nl[2][s]))
                self._collect("set_grow")
            self.Qt = d["Qt"].copy()
            self.last_control.linear.x = float(d["last_control"][0])
            self.last_control.angular.z = float(d["last_control"][1])
            self._draw = int(d["draw"])
            self._pose_cache = None

    # ------------------------------------------------------------------ shutdown
    def _shutdown(self, graceful):
        if self._closed:
            return
        self._closed = True
        if graceful:
            for c in self._conns:
                try:
                    c.send(("close",))
                except Exception:  # noqa: BLE001
                    pass
            for c in self._conns:
                try:
                    if c.poll(30.0):
                        c.recv()
                except Exception:  # noqa: BLE001
                    pass
        for p in self._procs:
            p.join(timeout=30 if graceful else 0.2)
            if p.is_alive():
                p.terminate()  # our own child, by handle
                p.join(timeout=10)
        for c in self._conns:
            try:
                c.close()
            except Exception:  # noqa: BLE001
                pass
        try:
            os.unlink(self._store)
        except OSError:
            pass

    def close(self):
        self._shutdown(graceful=self._failed is None)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
