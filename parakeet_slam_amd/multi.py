"""The reference's ``FastSLAM`` class surface over several GPUs: ``ShardedFastSLAM`` (also reached as
``FastSLAM(preset_features, devices=[0, 1, ...])``).

prkt_ros.py drives the core through four calls -- ``FastSLAM(preset_features)`` :91, ``core.cam_cb(self)`` :84,
``core.motion_update(msg)`` :118, ``core.summary()`` :94 -- from ONE process.  Here that process stays the front end and
never touches a GPU: it starts one child process per device (``multiprocessing`` "spawn": a fresh interpreter, started
BEFORE anything in this process could have initialised HIP; nothing is ever ``exec``-ed), each child owns one contiguous
shard of the particles with their whole maps (``ShardedFilter``, sharded.py: RCCL over xGMI between the children) and
serves commands from a pipe:

    step     one cam_cb (:59-137): weight reset, motion with the previous control (:163), association + EKF + weights,
             global systematic resample -- the draw u (``random.random()``, :226) is made HERE and sent to every rank
             (the replicated draw of the north star), as are the scan, Qt and, in the reference's RNG mode, each shard's
             slice of the ``numpy.random`` normals (:185-193)
    motion   motion_update (:148-166)
    summary  (:254-276): every rank all-reduces four pose sums; rank 0 answers
    poses / landmarks   views for ``particles[i]`` (lazy, like the single-GPU facade)

Particles are independent until the resample (:216-252), so nothing else crosses ranks.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import random as _pyrandom
import tempfile
import threading
import traceback

import numpy as np

from . import msgs
from .msgs import Twist


def _worker_main(rank, world, device, store_path, backend, P_local, L, means, covs, imm, domain, shard_factory, conn):
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
        sf = ShardedFilter(P_local, L, device=device, comm=TorchComm(), shard=shard)
        if L:
            sf.upload_map(means, covs.reshape(L, 25), imm)
        conn.send(("ok", None))
        while True:
            cmd = conn.recv()
            op = cmd[0]
            try:
                if op == "step":
                    _, v, w, dt, z, seed, draw, blobs, qt, u = cmd
                    if hasattr(sf.f, "set_measurement_noise"):
                        sf.f.set_measurement_noise(qt)
                    sf.step(v, w, dt, blobs, u, z=z, seed=seed, draw=draw, domain=domain)
                    conn.send(("ok", None))
                elif op == "motion":
                    _, v, w, dt, z, seed, draw = cmd
                    sf.motion(v, w, dt, z=z, seed=seed, draw=draw)
                    conn.send(("ok", None))
                elif op == "summary":
                    conn.send(("ok", sf.summary()))
                elif op == "poses":
                    conn.send(("ok", sf.download_poses()))
                elif op == "landmarks":
                    conn.send(("ok", sf.download_landmarks(cmd[1], cmd[2])))
                elif op == "close":
                    conn.send(("ok", None))
                    break
                else:
                    conn.send(("ERR", "unknown command %r" % (op,)))
            except Exception:  # noqa: BLE001 -- reported to the front end, which raises
                conn.send(("ERR", traceback.format_exc()))
        if hasattr(sf.f, "close"):
            sf.f.close()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        try:
            conn.send(("ERR", traceback.format_exc()))
        except Exception:  # noqa: BLE001
            pass


class _ShardedParticleList(object):
    """``fs.particles``: list-like, read-only views fetched from the owning rank on demand."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return self._o.num_particles

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, i):
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("particle index out of range")
        return self._o._particle_view(i)


class ShardedFastSLAM(object):
    """prkt_core_v2.py:37-276 with the particles sharded over ``devices`` (one child process per GPU).

    ShardedFastSLAM(preset_features=[], num_particles=50, devices=(0, 1), weight_domain="log", rng="device", seed=0,
                    backend="nccl")

    ``num_particles`` must be a multiple of ``len(devices)``.  rng as in ``FastSLAM``: "device" (Philox per global particle
    index: the noise does not depend on the shard count) or "global" (the reference's own ``numpy.random`` / ``random``
    streams, drawn here and handed to the shards).  ``backend="gloo"`` with ``_shard_factory`` is the CPU rehearsal the
    tests use."""

    def __init__(self, preset_features=[], num_particles=50, devices=(0, 1), weight_domain="log", rng="device", seed=0,
                 backend="nccl", _shard_factory=None):
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
        L = len(self._features)
        self._L = L
        means = np.array([np.asarray(f.mean, dtype=np.float64).reshape(5) for f in self._features]).reshape(L, 5)
        covs = np.array([np.asarray(f.covar, dtype=np.float64).reshape(5, 5) for f in self._features]).reshape(L, 5, 5)
        imm = np.array([bool(f.__immutable__) for f in self._features], dtype=np.uint8)
        self.Qt = np.identity(4) * 0.1  # prkt_core_v2.py:50-53
        self._domain = {"linear": _lib.PK_WEIGHTS_LINEAR, "log": _lib.PK_WEIGHTS_LOG}[weight_domain]
        self._rng, self._seed, self._draw = rng, int(seed), 0
        self._pose_cache = None
        ctx = mp.get_context("spawn")  # fresh interpreters: no child inherits (or replaces) a process that touched a GPU
        fd, self._store = tempfile.mkstemp(prefix="pk_store_")
        os.close(fd)
        os.unlink(self._store)
        self._conns, self._procs = [], []
        for r, dev in enumerate(self.devices):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(r, world, dev, self._store, backend, self._P_local, L, means, covs, imm,
                                                       self._domain, _shard_factory, b), daemon=True)
            p.start()
            self._conns.append(a)
            self._procs.append(p)
        self._closed = False
        try:
            self._collect("start")
        except Exception:
            self.close()
            raise
        self.particles = _ShardedParticleList(self)

    # ------------------------------------------------------------------ plumbing
    def _collect(self, what, timeout=600.0):
        out = []
        for r, c in enumerate(self._conns):
            if not c.poll(timeout):
                raise RuntimeError("rank %d did not answer '%s' within %.0f s" % (r, what, timeout))
            status, payload = c.recv()
            if status != "ok":
                raise RuntimeError("rank %d failed in '%s':\n%s" % (r, what, payload))
            out.append(payload)
        return out

    def _all(self, *cmd):
        for c in self._conns:
            c.send(cmd)
        return self._collect(cmd[0])

    def _noise(self):
        if self._rng == "global":  # numpy.random.normal(0, s, 1) x 3 per particle, particle-major (:185-193)
            z = np.random.standard_normal((self.num_particles, 3))
            return [z[r * self._P_local:(r + 1) * self._P_local] for r in range(len(self.devices))]
        return [None] * len(self.devices)

    # ------------------------------------------------------------------ the reference's surface
    def cam_cb(self, ros_view):
        """One filter step (:59-137)."""
        with self._lock:
            dt = msgs.now() - self.last_update  # :158, the motion update of :75-77
            v, w = float(self.last_control.linear.x), float(self.last_control.angular.z)
            observes = list(ros_view.last_sensor_reading.observes)  # :82
            blobs = np.array([[float(b.bearing), float(b.color.r), float(b.color.g), float(b.color.b)] for b in observes],
                             dtype=np.float64).reshape(-1, 4)
            zs = self._noise()
            u = _pyrandom.random()  # :226 -- drawn once, here, and replicated to every rank
            qt = np.asarray(self.Qt, dtype=np.float64).reshape(4, 4)
            for r, c in enumerate(self._conns):
                c.send(("step", v, w, float(dt.to_sec()), zs[r], self._seed, self._draw, blobs, qt, float(u)))
            self._collect("step")
            self._draw += 1
            self.last_update = self.last_update + dt  # :165 (last_control stays: cam_cb moves with it, :77)
            self._pose_cache = None

    def motion_update(self, new_twist):
        """:148-166: moves every particle with the PREVIOUS control, then takes the new one."""
        with self._lock:
            dt = msgs.now() - self.last_update
            v, w = float(self.last_control.linear.x), float(self.last_control.angular.z)
            zs = self._noise()
            for r, c in enumerate(self._conns):
                c.send(("motion", v, w, float(dt.to_sec()), zs[r], self._seed, self._draw))
            self._collect("motion")
            self._draw += 1
            self.last_update = self.last_update + dt
            self.last_control = new_twist
            self._pose_cache = None

    def summary(self):
        """:254-276 over all shards."""
        with self._lock:
            return tuple(self._all("summary")[0])

    def odom_motion_update(self, odom):
        pass  # :140-146 alpha feature, empty in the reference

    # ------------------------------------------------------------------ views
    def download_poses(self):
        with self._lock:
            if self._pose_cache is None:
                self._pose_cache = np.concatenate(self._all("poses"))
            return self._pose_cache

    def _particle_view(self, i):
        from .core import Feature, FilterParticle, _FeatureSet, _make_state

        with self._lock:
            x, y, h, w = self.download_poses()[i]
            p = FilterParticle(_make_state(x, y, h))
            p.weight = float(w)
            p.Qt = self.Qt
            r, j = divmod(i, self._P_local)
            conn, feats, lock, L = self._conns[r], self._features, self._lock, self._L

            def load_all():
                with lock:
                    conn.send(("landmarks", j, j + 1))
                    status, payload = conn.recv()
                if status != "ok":
                    raise RuntimeError(payload)
                m, c, k = payload
                full = {}
                for l in range(L):
                    f = Feature(mean=m[0, l], covar=c[0, l])
                    f.update_count = int(k[0, l])
                    f.__immutable__ = bool(feats[l].__immutable__)
                    full[l + 1] = f
                return full

            p.feature_set = _FeatureSet(load_all)
            return p

    def close(self):
        if self._closed:
            return
        self._closed = True
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
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()  # our own child, by handle
        try:
            os.unlink(self._store)
        except OSError:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
