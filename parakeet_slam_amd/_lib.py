"""ctypes binding of ``libparakeet_slam.so`` (include/parakeet_slam.h).

The library is the product path.  There is no CPU fallback: if the shared object
is missing, or no HIP device is visible when a filter is created, this fails
loudly (``PkError`` / ``OSError``) instead of computing anything on the host.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libparakeet_slam.so")

PK_OK = 0
PK_ERR_INVALID, PK_ERR_HIP, PK_ERR_STATE, PK_ERR_UNSUPPORTED, PK_ERR_NOMEM = -1, -2, -3, -4, -5
PK_WEIGHTS_LINEAR, PK_WEIGHTS_LOG = 0, 1
PK_LANDMARK_POTENTIAL = 0x40000000  # flag in a landmark's count word: potential feature (prkt_core_v2.py:109-118)
PK_T_NAMES = ("motion", "assoc", "observe", "weights", "resample", "summary", "materialise")
PK_T_COUNT = len(PK_T_NAMES)
PK_PROBE_LEN = 79
PK_ABI_VERSION = 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_bp = C.POINTER(C.c_uint8)
_h = C.c_void_p

# name -> (restype, argtypes); mirrors include/parakeet_slam.h one to one
SIGNATURES = {
    "pk_abi_version": (C.c_int, []),
    "pk_status_string": (C.c_char_p, [C.c_int]),
    "pk_last_error": (C.c_char_p, []),
    "pk_device_count": (C.c_int, []),
    "pk_create": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(_h)]),
    "pk_destroy": (C.c_int, [_h]),
    "pk_set_stream": (C.c_int, [_h, C.c_void_p]),
    "pk_synchronize": (C.c_int, [_h]),
    "pk_num_particles": (C.c_int64, [_h]),
    "pk_num_landmarks": (C.c_int32, [_h]),
    "pk_device_bytes": (C.c_int64, [_h]),
    "pk_set_option": (C.c_int, [_h, C.c_char_p, C.c_int64]),
    "pk_set_measurement_noise": (C.c_int, [_h, _dp]),
    "pk_upload_map": (C.c_int, [_h, _dp, _dp, _bp]),
    "pk_upload_poses": (C.c_int, [_h, _dp]),
    "pk_download_poses": (C.c_int, [_h, _dp]),
    "pk_upload_pose": (C.c_int, [_h, C.c_int64, _dp]),
    "pk_download_landmarks": (C.c_int, [_h, C.c_int64, C.c_int64, _dp, _dp, _ip]),
    "pk_upload_landmarks": (C.c_int, [_h, C.c_int64, C.c_int64, _dp, _dp, _ip]),
    "pk_reset_weights": (C.c_int, [_h]),
    "pk_motion": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_uint64, C.c_uint64]),
    "pk_download_log_weights": (C.c_int, [_h, _dp]),
    "pk_observe": (C.c_int, [_h, _dp, C.c_int32, _ip, _ip]),
    "pk_observe_fresh": (C.c_int, [_h, _dp, C.c_int32, _ip, _ip]),
    "pk_stage_scan": (C.c_int, [_h, _dp, C.c_int32]),
    "pk_observe_staged": (C.c_int, [_h, C.c_int32]),
    "pk_associate": (C.c_int, [_h, _dp, C.c_int32, _ip]),
    "pk_resample": (C.c_int, [_h, C.c_double, C.c_int32, _lp]),
    "pk_summary": (C.c_int, [_h, _dp]),
    "pk_pose_sums": (C.c_int, [_h, _dp]),
    "pk_step": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, _dp, C.c_uint64, C.c_uint64, _dp, C.c_int32,
                          _ip, C.c_double, C.c_int32]),
    "pk_shard_max_logw": (C.c_int, [_h, _dp]),
    "pk_shard_num_blocks": (C.c_int64, [_h]),
    "pk_shard_block_totals": (C.c_int, [_h, C.c_double, C.c_int32, _dp]),
    "pk_set_shard": (C.c_int, [_h, C.c_int64]),
    "pk_shard_offspring": (C.c_int, [_h, _dp, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int32, _lp]),
    "pk_shard_max_logw_dev": (C.c_int, [_h, C.c_void_p]),
    "pk_shard_block_totals_dev": (C.c_int, [_h, C.c_void_p, C.c_int32, C.c_void_p]),
    "pk_shard_plan_dev": (C.c_int, [_h, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int32, C.c_int32,
                                    C.c_void_p]),
    "pk_shard_logw_dev": (C.c_int, [_h, C.c_void_p]),
    "pk_shard_plan_global_dev": (C.c_int, [_h, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                           C.c_void_p]),
    "pk_shard_download_offspring": (C.c_int, [_h, _lp]),
    "pk_shard_pack_dev": (C.c_int, [_h, _lp, C.c_int32, C.c_int32, C.c_void_p]),
    "pk_shard_pack_slots_dev": (C.c_int, [_h, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "pk_shard_adopt_dev": (C.c_int, [_h, C.c_int32, C.c_void_p, C.c_int64]),
    "pk_shard_local_span_dev": (C.c_int, [_h, C.c_void_p]),
    "pk_shard_adopt_local_dev": (C.c_int, [_h, C.c_int32]),
    "pk_shard_adopt_remote_dev": (C.c_int, [_h, C.c_int32, C.c_void_p, C.c_int64]),
    "pk_shard_state_dev": (C.c_int, [_h, C.c_void_p]),
    "pk_shard_plan_balanced_dev": (C.c_int, [_h, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                             C.c_void_p]),
    "pk_shard_pack_balanced_dev": (C.c_int, [_h, _lp, C.c_int32, C.c_int32, C.c_void_p]),
    "pk_shard_adopt_balanced_dev": (C.c_int, [_h, _lp, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32]),
    "pk_shard_pack_balanced_loop_dev": (C.c_int, [_h, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "pk_shard_download_logical": (C.c_int, [_h, _lp]),
    "pk_shard_reset_placement": (C.c_int, [_h]),
    "pk_shard_upload_logical": (C.c_int, [_h, _lp]),
    "pk_shard_download_balanced_plan": (C.c_int, [_h, _lp, _lp, _ip]),
    "pk_shard_download_balanced_offspring": (C.c_int, [_h, C.c_int64, _lp]),
    "pk_shard_balanced_errors": (C.c_int, [_h, _lp]),
    "pk_motion_range": (C.c_int, [_h, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64]),
    "pk_staged_takes_regs": (C.c_int, [_h]),
    "pk_observe_staged_range": (C.c_int, [_h, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "pk_particle_bytes": (C.c_int64, [_h]),
    "pk_pack_particles": (C.c_int, [_h, _lp, C.c_int64, C.c_void_p]),
    "pk_adopt_particles": (C.c_int, [_h, _lp, C.c_void_p, C.c_int64]),
    "pk_probe": (C.c_int, [C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp]),
    "pk_enable_timing": (C.c_int, [_h, C.c_int32]),
    "pk_reset_timings": (C.c_int, [_h]),
    "pk_timings": (C.c_int, [_h, _dp, _lp]),
    "pk_observe_bytes": (C.c_int, [_h, C.c_int32, _lp, _lp]),
    "pk_observe_route": (C.c_int, [_h]),
    "pk_download_sources": (C.c_int, [_h, _ip]),
    "pk_observe_flagged": (C.c_int, [_h, _lp, _lp]),
    "pk_observe_retry_rows": (C.c_int, [_h, _lp, _lp]),
    "pk_grow_enable": (C.c_int, [_h, C.c_int32, C.c_int32, C.c_double]),
    "pk_grow_download": (C.c_int, [_h, C.c_int64, C.c_int64, _ip, _dp, _ip]),
    "pk_grow_upload": (C.c_int, [_h, C.c_int64, C.c_int64, _ip, _dp, _ip]),
    "pk_grow_shape": (C.c_int, [_h, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pk_observe_published": (C.c_int, [_h, C.POINTER(C.c_int32)]),
    "pk_observe_flags": (C.c_int, [_h, _bp]),
    "pk_observe_pub_stats": (C.c_int, [_h, _lp]),
    "pk_rng_create_numpy": (C.c_int, [C.c_uint32, C.POINTER(_h)]),
    "pk_rng_create_python": (C.c_int, [C.c_uint32, C.POINTER(_h)]),
    "pk_rng_destroy": (C.c_int, [_h]),
    "pk_rng_standard_normal": (C.c_int, [_h, C.c_int64, _dp]),
    "pk_rng_random": (C.c_int, [_h, C.c_int64, _dp]),
}


class PkError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("parakeet_slam [%d]: %s" % (status, message))
        self.status = status


_lib = None


def load():
    """Load the shared library (once).  Raises OSError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            "%s is missing: build it with `python -m parakeet_slam_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the particle update." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    ver = lib.pk_abi_version()
    if ver != PK_ABI_VERSION:
        raise OSError("libparakeet_slam.so has ABI %d, the binding expects %d: rebuild" % (ver, PK_ABI_VERSION))
    _lib = lib
    return lib


def check(status):
    if status != PK_OK:
        lib = load()
        msg = lib.pk_last_error().decode("utf-8", "replace") or lib.pk_status_string(status).decode()
        raise PkError(status, msg)


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def lptr(a):
    return a.ctypes.data_as(_lp) if a is not None else None


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class DeviceFilter(object):
    """Thin, numpy-in / numpy-out wrapper of one ``pk_filter`` handle.

    This is the layer the FastSLAM facade, the parity tests and bench.py share.
    """

    def __init__(self, num_particles, num_landmarks, device=0):
        self._lib = load()
        self._h = _h()
        check(self._lib.pk_create(int(num_particles), int(num_landmarks), int(device), C.byref(self._h)))
        self.P = int(num_particles)
        self.L = int(num_landmarks)
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pk_destroy(self._h)
            self._h = _h()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ---------------------------------------------------
    def set_stream(self, stream_ptr):
        check(self._lib.pk_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        check(self._lib.pk_synchronize(self._h))

    def device_bytes(self):
        return int(self._lib.pk_device_bytes(self._h))

    def set_option(self, name, value):
        check(self._lib.pk_set_option(self._h, name.encode(), int(value)))

    def set_measurement_noise(self, Qt):
        q = f64(Qt, (16,))
        check(self._lib.pk_set_measurement_noise(self._h, dptr(q)))

    def upload_map(self, means, covs, immutable=None):
        m = f64(means, (self.L, 5))
        c = f64(covs, (self.L, 25))
        im = None
        if immutable is not None:
            im = np.ascontiguousarray(immutable, dtype=np.uint8).reshape(self.L)
        check(self._lib.pk_upload_map(self._h, dptr(m), dptr(c), im.ctypes.data_as(_bp) if im is not None else None))

    def upload_poses(self, xyhw):
        a = f64(xyhw, (self.P, 4))
        check(self._lib.pk_upload_poses(self._h, dptr(a)))

    def upload_pose(self, p, xyhw):
        """One particle's (x, y, heading, weight); the other particles keep their log-weights bit for bit."""
        a = f64(xyhw, (4,))
        check(self._lib.pk_upload_pose(self._h, int(p), dptr(a)))

    def download_poses(self):
        out = np.empty((self.P, 4), dtype=np.float64)
        check(self._lib.pk_download_poses(self._h, dptr(out)))
        return out

    def download_landmarks(self, p0=0, p1=None, means=True, covs=True, counts=True):
        p1 = self.P if p1 is None else p1
        n = p1 - p0
        m = np.empty((n, self.L, 5)) if means else None
        c = np.empty((n, self.L, 5, 5)) if covs else None
        k = np.empty((n, self.L), dtype=np.int32) if counts else None
        check(self._lib.pk_download_landmarks(self._h, p0, p1, dptr(m), dptr(c), iptr(k)))
        return m, c, k

    def upload_landmarks(self, p0, p1, means=None, covs=None, counts=None):
        n = p1 - p0
        m = f64(means, (n, self.L, 5)) if means is not None else None
        c = f64(covs, (n, self.L, 25)) if covs is not None else None
        k = np.ascontiguousarray(counts, dtype=np.int32).reshape(n, self.L) if counts is not None else None
        check(self._lib.pk_upload_landmarks(self._h, p0, p1, dptr(m), dptr(c), iptr(k)))

    # -- the step ----------------------------------------------------------
    def reset_weights(self):
        check(self._lib.pk_reset_weights(self._h))

    def motion(self, v, w, dt, z=None, seed=0, draw=0):
        zz = f64(z, (self.P, 3)) if z is not None else None
        check(self._lib.pk_motion(self._h, float(v), float(w), float(dt), dptr(zz), int(seed), int(draw)))

    def download_log_weights(self):
        out = np.empty(self.P, dtype=np.float64)
        check(self._lib.pk_download_log_weights(self._h, dptr(out)))
        return out

    def observe(self, blobs, ids=None, return_ids=False, fresh=False):
        """fresh: the weights restart from 1 first (prkt_core_v2.py:73), fused into the same kernels."""
        b = f64(blobs).reshape(-1, 4)
        B = b.shape[0]
        i = np.ascontiguousarray(ids, dtype=np.int32).reshape(B) if ids is not None else None
        out = np.empty((self.P, B), dtype=np.int32) if return_ids else None
        fn = self._lib.pk_observe_fresh if fresh else self._lib.pk_observe
        check(fn(self._h, dptr(b), B, iptr(i), iptr(out)))
        return out

    def stage_scan(self, blobs):
        """Host half of an ML observe (tables into a pinned staging slot), no kernel launched."""
        b = f64(blobs).reshape(-1, 4)
        check(self._lib.pk_stage_scan(self._h, dptr(b), b.shape[0]))

    def observe_staged(self, fresh=False):
        """Device half: upload + kernels for the scan staged last."""
        check(self._lib.pk_observe_staged(self._h, 1 if fresh else 0))

    def associate(self, blobs):
        b = f64(blobs).reshape(-1, 4)
        out = np.zeros((self.P, b.shape[0]), dtype=np.int32)
        check(self._lib.pk_associate(self._h, dptr(b), b.shape[0], iptr(out)))
        return out

    def resample(self, u, domain=PK_WEIGHTS_LINEAR, return_ancestors=False):
        out = np.empty(self.P, dtype=np.int64) if return_ancestors else None
        check(self._lib.pk_resample(self._h, float(u), int(domain), lptr(out)))
        return out

    def summary(self):
        out = np.empty(3, dtype=np.float64)
        check(self._lib.pk_summary(self._h, dptr(out)))
        return float(out[0]), float(out[1]), float(out[2])

    def pose_sums(self):
        out = np.empty(4, dtype=np.float64)
        check(self._lib.pk_pose_sums(self._h, dptr(out)))
        return out

    def step(self, v, w, dt, blobs, u, z=None, seed=0, draw=0, ids=None, domain=PK_WEIGHTS_LINEAR):
        b = f64(blobs).reshape(-1, 4)
        B = b.shape[0]
        zz = f64(z, (self.P, 3)) if z is not None else None
        i = np.ascontiguousarray(ids, dtype=np.int32).reshape(B) if ids is not None else None
        check(self._lib.pk_step(self._h, float(v), float(w), float(dt), dptr(zz), int(seed), int(draw), dptr(b), B,
                                iptr(i), float(u), int(domain)))

    # -- sharded resample (see sharded.py) ---------------------------------------
    def set_shard(self, global_offset):
        check(self._lib.pk_set_shard(self._h, int(global_offset)))

    def shard_max_logw(self):
        v = C.c_double()
        check(self._lib.pk_shard_max_logw(self._h, C.byref(v)))
        return float(v.value)

    def shard_num_blocks(self):
        return int(self._lib.pk_shard_num_blocks(self._h))

    def shard_block_totals(self, gmax, domain):
        out = np.empty(self.shard_num_blocks(), dtype=np.float64)
        check(self._lib.pk_shard_block_totals(self._h, float(gmax), int(domain), dptr(out)))
        return out

    def shard_offspring(self, global_totals, first_block, global_particles, u, last_shard):
        t = f64(global_totals)
        out = np.empty(self.P + 1, dtype=np.int64)
        check(self._lib.pk_shard_offspring(self._h, dptr(t), t.size, int(first_block), int(global_particles), float(u),
                                           1 if last_shard else 0, lptr(out)))
        return out

    def shard_max_logw_dev(self, out_ptr):
        check(self._lib.pk_shard_max_logw_dev(self._h, C.c_void_p(out_ptr)))

    def shard_block_totals_dev(self, gmax_ptr, domain, totals_ptr):
        check(self._lib.pk_shard_block_totals_dev(self._h, C.c_void_p(gmax_ptr), int(domain), C.c_void_p(totals_ptr)))

    def shard_plan_dev(self, gtotals_ptr, n_blocks, first_block, global_particles, u, last_shard, world, ranges_ptr):
        check(self._lib.pk_shard_plan_dev(self._h, C.c_void_p(gtotals_ptr), int(n_blocks), int(first_block),
                                          int(global_particles), float(u), 1 if last_shard else 0, int(world),
                                          C.c_void_p(ranges_ptr)))

    def shard_logw_dev(self, out_ptr):
        check(self._lib.pk_shard_logw_dev(self._h, C.c_void_p(out_ptr)))

    def shard_plan_global_dev(self, glogw_ptr, global_particles, gmax_ptr, domain, u, last_shard, world, ranges_ptr):
        check(self._lib.pk_shard_plan_global_dev(self._h, C.c_void_p(glogw_ptr), int(global_particles), C.c_void_p(gmax_ptr),
                                                 int(domain), float(u), 1 if last_shard else 0, int(world), C.c_void_p(ranges_ptr)))

    def shard_download_offspring(self):
        out = np.empty(self.P + 1, dtype=np.int64)
        check(self._lib.pk_shard_download_offspring(self._h, lptr(out)))
        return out

    def shard_pack_dev(self, ranges, world, rank, buf_ptr):
        r = np.ascontiguousarray(ranges, dtype=np.int64)
        check(self._lib.pk_shard_pack_dev(self._h, lptr(r), int(world), int(rank), C.c_void_p(buf_ptr)))

    def shard_pack_slots_dev(self, j0, j1, slot_lo, slot_hi, buf_ptr):
        check(self._lib.pk_shard_pack_slots_dev(self._h, int(j0), int(j1), int(slot_lo), int(slot_hi), C.c_void_p(buf_ptr)))

    def shard_adopt_dev(self, rank, recv_ptr, n_received):
        check(self._lib.pk_shard_adopt_dev(self._h, int(rank), C.c_void_p(recv_ptr), int(n_received)))

    # -- the split step (the exchange overlapped with the work on the particles that stay) --
    def shard_local_span_dev(self, out_ptr):
        check(self._lib.pk_shard_local_span_dev(self._h, C.c_void_p(out_ptr)))

    def shard_adopt_local_dev(self, rank):
        check(self._lib.pk_shard_adopt_local_dev(self._h, int(rank)))

    def shard_adopt_remote_dev(self, rank, recv_ptr, n_received):
        check(self._lib.pk_shard_adopt_remote_dev(self._h, int(rank), C.c_void_p(recv_ptr), int(n_received)))

    # -- balanced placement (minimum migration) --
    def shard_state_dev(self, out_ptr):
        check(self._lib.pk_shard_state_dev(self._h, C.c_void_p(out_ptr)))

    def shard_plan_balanced_dev(self, gstate_ptr, global_particles, gmax_ptr, domain, u, world, rank, table_ptr):
        check(self._lib.pk_shard_plan_balanced_dev(self._h, C.c_void_p(gstate_ptr), int(global_particles), C.c_void_p(gmax_ptr),
                                                   int(domain), float(u), int(world), int(rank), C.c_void_p(table_ptr)))

    def shard_pack_balanced_dev(self, table, world, rank, buf_ptr):
        t = np.ascontiguousarray(table, dtype=np.int64)
        check(self._lib.pk_shard_pack_balanced_dev(self._h, lptr(t), int(world), int(rank), C.c_void_p(buf_ptr)))

    def shard_adopt_balanced_dev(self, table, world, rank, recv_ptr, n_received, mode):
        t = np.ascontiguousarray(table, dtype=np.int64)
        check(self._lib.pk_shard_adopt_balanced_dev(self._h, lptr(t), int(world), int(rank), C.c_void_p(recv_ptr), int(n_received),
                                                    int(mode)))

    def shard_pack_balanced_loop_dev(self, keep, a0, a1, buf_ptr):
        check(self._lib.pk_shard_pack_balanced_loop_dev(self._h, int(keep), int(a0), int(a1), C.c_void_p(buf_ptr)))

    def download_logical(self):
        out = np.empty(self.P, dtype=np.int64)
        check(self._lib.pk_shard_download_logical(self._h, lptr(out)))
        return out

    def upload_logical(self, logical):
        a = np.ascontiguousarray(logical, dtype=np.int64).reshape(self.P)
        check(self._lib.pk_shard_upload_logical(self._h, lptr(a)))

    def shard_download_balanced_plan(self):
        """(rel int64[P + 1], Hl int64[P], alive int32[n_alive]) of the last balanced plan (tests)."""
        rel, Hl, alive = np.empty(self.P + 1, dtype=np.int64), np.empty(self.P, dtype=np.int64), np.empty(self.P, dtype=np.int32)
        check(self._lib.pk_shard_download_balanced_plan(self._h, lptr(rel), lptr(Hl), iptr(alive)))
        return rel, Hl, alive[alive >= 0]

    def reset_placement(self):
        check(self._lib.pk_shard_reset_placement(self._h))

    def shard_download_balanced_offspring(self, global_particles):
        out = np.empty(int(global_particles) + 1, dtype=np.int64)
        check(self._lib.pk_shard_download_balanced_offspring(self._h, int(global_particles), lptr(out)))
        return out

    def shard_balanced_errors(self):
        out = np.zeros(1, dtype=np.int64)
        check(self._lib.pk_shard_balanced_errors(self._h, lptr(out)))
        return int(out[0])

    def motion_range(self, v, w, dt, seed, draw, p0, p1):
        check(self._lib.pk_motion_range(self._h, float(v), float(w), float(dt), int(seed), int(draw), int(p0), int(p1)))

    def staged_takes_regs(self):
        return bool(self._lib.pk_staged_takes_regs(self._h))

    def observe_staged_range(self, fresh, p0, p1, first, last):
        check(self._lib.pk_observe_staged_range(self._h, 1 if fresh else 0, int(p0), int(p1), 1 if first else 0, 1 if last else 0))

    ROUTES = {0: "none", 1: "known_ids", 2: "ml_general", 3: "ml_handoff", 4: "ml_sweep", 5: "ml_fused", 6: "ml_regs", 8: "dense", 9: "ml_pub_big"}

    def observe_route(self):
        """Kernels the last observe / step used for association + EKF update (pk_observe_route)."""
        return self.ROUTES[int(self._lib.pk_observe_route(self._h))]

    def observe_flagged(self):
        """(particles the last ML observe handed to the general kernels, candidate-list overflows) -- pk_observe_flagged."""
        a, b = C.c_int64(), C.c_int64()
        check(self._lib.pk_observe_flagged(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def observe_pub_stats(self):
        """The last scan's publish table (pk_observe_pub_stats): dict of entries, contested blobs, landmarks of the reference
        particle with several blobs inside their gates, longest candidate list, entry capacity, instance (0 none, 1 one
        workgroup per CU, 2 k_step_pub_duo)."""
        a = np.zeros(6, dtype=np.int64)
        check(self._lib.pk_observe_pub_stats(self._h, lptr(a)))
        return dict(zip(("entries", "contested_blobs", "multi_landmarks", "longest_list", "entry_capacity", "instance"), (int(v) for v in a)))

    # ---- new landmarks on the device (pk_grow_enable) ----
    def grow_enable(self, preset_landmarks, reading_capacity=64, pair_threshold=30.0):
        check(self._lib.pk_grow_enable(self._h, int(preset_landmarks), int(reading_capacity), float(pair_threshold)))

    def grow_shape(self):
        """(preset landmarks, spare slots, reading capacity); zeros while the bookkeeping is off."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._lib.pk_grow_shape(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def grow_download(self, p0=0, p1=None, counters=True, readings=True, slot_ids=True):
        """(counters (n, 4) int32: readings stored / spare slots in use / next_id / readings dropped,
        readings (n, R, 8): id, x, y, heading, bearing, r, g, b, slot_ids (n, S) int32); None for what was not asked for."""
        p1 = self.P if p1 is None else p1
        _, S, R = self.grow_shape()
        n = p1 - p0
        c = np.empty((n, 4), dtype=np.int32) if counters else None
        r = np.empty((n, R, 8), dtype=np.float64) if readings else None
        s = np.empty((n, S), dtype=np.int32) if slot_ids else None
        check(self._lib.pk_grow_download(self._h, int(p0), int(p1), iptr(c), dptr(r), iptr(s)))
        return c, r, s

    def grow_upload(self, p0, p1, counters=None, readings=None, slot_ids=None):
        _, S, R = self.grow_shape()
        n = p1 - p0
        c = None if counters is None else np.ascontiguousarray(counters, dtype=np.int32).reshape(n, 4)
        r = None if readings is None else f64(readings, (n, R, 8))
        s = None if slot_ids is None else np.ascontiguousarray(slot_ids, dtype=np.int32).reshape(n, S)
        check(self._lib.pk_grow_upload(self._h, int(p0), int(p1), iptr(c), dptr(r), iptr(s)))

    def observe_retry_rows(self):
        """(second-chance rows the last scan wanted, rows the next scan will find) -- pk_observe_retry_rows."""
        a, b = C.c_int64(), C.c_int64()
        check(self._lib.pk_observe_retry_rows(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def observe_published(self):
        """True when k_step_pub (static publish / subscribe settling) worked on the last register-route scan."""
        v = C.c_int32()
        check(self._lib.pk_observe_published(self._h, C.byref(v)))
        return bool(v.value)

    def observe_flags(self):
        """Per particle: 0 = settled by the one-pass kernel, 2 = second-chance route, 1 = general kernels (pk_observe_flags)."""
        out = np.zeros(self.P, dtype=np.uint8)
        check(self._lib.pk_observe_flags(self._h, out.ctypes.data_as(_bp)))
        return out

    def download_sources(self):
        """Map slot each particle's landmarks currently live in (pk_download_sources)."""
        out = np.empty(self.P, dtype=np.int32)
        check(self._lib.pk_download_sources(self._h, iptr(out)))
        return out

    def particle_bytes(self):
        return int(self._lib.pk_particle_bytes(self._h))

    def pack_particles(self, local_idx, dev_ptr):
        idx = np.ascontiguousarray(local_idx, dtype=np.int64)
        check(self._lib.pk_pack_particles(self._h, lptr(idx), idx.size, C.c_void_p(dev_ptr)))

    def adopt_particles(self, src, dev_ptr, n_received):
        s_ = np.ascontiguousarray(src, dtype=np.int64).reshape(self.P)
        check(self._lib.pk_adopt_particles(self._h, lptr(s_), C.c_void_p(dev_ptr), int(n_received)))

    # -- instrumentation -----------------------------------------------------
    def enable_timing(self, mask=True):
        """mask: True = every kernel slot, False/0 = off, int = bitmask over PK_T_NAMES."""
        m = -1 if mask is True else (0 if mask is False else int(mask))
        check(self._lib.pk_enable_timing(self._h, m))

    def reset_timings(self):
        check(self._lib.pk_reset_timings(self._h))

    def timings(self):
        ms = np.zeros(PK_T_COUNT, dtype=np.float64)
        n = np.zeros(PK_T_COUNT, dtype=np.int64)
        check(self._lib.pk_timings(self._h, dptr(ms), lptr(n)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(PK_T_NAMES)}

    def observe_bytes(self, num_blobs):
        a = C.c_int64()
        m = C.c_int64()
        check(self._lib.pk_observe_bytes(self._h, int(num_blobs), C.byref(a), C.byref(m)))
        return int(a.value), int(m.value)


def probe(pose, mean, cov, blob, Qt=None, device=0):
    """Evaluate every scalar function of the path on one (pose, landmark, blob) on the GPU."""
    lib = load()
    Qt = 0.1 * np.identity(4) if Qt is None else Qt
    out = np.empty(PK_PROBE_LEN, dtype=np.float64)
    check(lib.pk_probe(int(device), dptr(f64(pose, (3,))), dptr(f64(mean, (5,))), dptr(f64(cov, (25,))),
                       dptr(f64(blob, (4,))), dptr(f64(Qt, (16,))), dptr(out)))
    return dict(
        probability_of_match=out[0], prob_position_match=out[1], closest_point=out[2:4].copy(),
        prob_color_match=out[4], zhat=out[5:9].copy(), H0=out[9:11].copy(), Q=out[11:27].reshape(4, 4).copy(),
        K=out[27:47].reshape(5, 4).copy(), weight=out[47], new_mean=out[48:53].copy(),
        new_cov=out[53:78].reshape(5, 5).copy(), log_weight=out[78],
    )


class HostRng(object):
    """Host reproduction of numpy's legacy normal stream / CPython's random() (pk_rng_*)."""

    def __init__(self, seed, kind="numpy"):
        self._lib = load()
        self._h = _h()
        fn = self._lib.pk_rng_create_numpy if kind == "numpy" else self._lib.pk_rng_create_python
        check(fn(int(seed) & 0xFFFFFFFF, C.byref(self._h)))

    def standard_normal(self, n):
        out = np.empty(int(n), dtype=np.float64)
        check(self._lib.pk_rng_standard_normal(self._h, int(n), dptr(out)))
        return out

    def random(self, n=None):
        out = np.empty(1 if n is None else int(n), dtype=np.float64)
        check(self._lib.pk_rng_random(self._h, out.size, dptr(out)))
        return float(out[0]) if n is None else out

    def __del__(self):
        try:
            if self._h.value:
                self._lib.pk_rng_destroy(self._h)
                self._h = _h()
        except Exception:
            pass
