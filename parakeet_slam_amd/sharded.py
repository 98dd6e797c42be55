"""Particles sharded over one process per GPU.

Everything except ``low_variance_resample`` is independent per particle
(prkt_core_v2.py:67-134 touches only ``self.particles[i]``), so a shard runs motion,
association and the EKF on its own particles with no communication.  The resample is the
one coupling point (prkt_core_v2.py:216-252: a global weight sum and one ordered walk over
all particles).  Per step a ShardedFilter therefore does:

  1. all-reduce(MAX) of the shard's max log-weight                     (1 float64; log domain only)
  2. all-gather of the shard's weight-scan block totals                (P_local / 1024 float64)
  3. every shard scans the SAME global block totals in the SAME order, so the comb
     u r + k r lands on identical cumulative sums everywhere: each shard knows, for each of
     its own particles, which global output slots it fills -- no further communication
  4. all-to-all of the particles whose slots belong to another shard   (pose + landmark map;
     the only bandwidth-relevant traffic, and only for migrating particles)

The draw u is replicated (same value on every rank), as the north star asks.  With shards
that are multiples of 1024 particles the ancestors are bit-identical to the 1-GPU run.

The collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on ROCm
(device tensors, all_to_all_single); "gloo" is used by the CPU / single-GPU tests.
"""
from __future__ import annotations

import contextlib

import numpy as np

from . import _lib

_nullcontext = contextlib.nullcontext

SCAN_BLOCK = 1024
UNASSIGNED = np.iinfo(np.int64).min  # record r is encoded as -(r + 1), so -1 is taken


# --------------------------------------------------------------------------- planning
def plan_exchange(hi, rank, world, p_local):
    """Who sends what where.  Pure NumPy (tested on CPU).

    hi: int64[p_local + 1], hi[0] = output slots filled by earlier shards, hi[1 + j] = after
    local particle j (pk_shard_offspring).  Slot k belongs to rank k // p_local.

    Returns (local_src, sends):
      local_src  int64[p_local]: for the slots this rank owns, the local ancestor index where
                 the ancestor is local, else UNASSIGNED (to be filled from received records)
      sends      list over destination ranks of (idx, lo, hi) int64 arrays: local particle idx
                 fills global slots [lo, hi) of that destination (clipped to its interval);
                 sends[rank] is empty.
    """
    hi = np.maximum.accumulate(np.asarray(hi, dtype=np.int64))
    lo_j, hi_j = hi[:-1], hi[1:]
    local_src = np.full(p_local, UNASSIGNED, dtype=np.int64)
    sends = []
    for dest in range(world):
        start, end = dest * p_local, (dest + 1) * p_local
        j0 = int(np.searchsorted(hi_j, start, side="right"))
        j1 = int(np.searchsorted(lo_j, end, side="left"))
        if j1 <= j0:
            sends.append((np.empty(0, np.int64),) * 3)
            continue
        idx = np.arange(j0, j1, dtype=np.int64)
        lo = np.maximum(lo_j[j0:j1], start)
        up = np.minimum(hi_j[j0:j1], end)
        keep = up > lo
        idx, lo, up = idx[keep], lo[keep], up[keep]
        if dest == rank:
            counts = up - lo
            local_src[np.repeat(lo - start, counts) + _ranges(counts)] = np.repeat(idx, counts)
            sends.append((np.empty(0, np.int64),) * 3)
        else:
            sends.append((idx, lo, up))
    return local_src, sends


def _ranges(counts):
    """concatenate([arange(c) for c in counts]) without a Python loop."""
    counts = np.asarray(counts, dtype=np.int64)
    total = int(counts.sum())
    if total == 0:
        return np.empty(0, np.int64)
    starts = np.cumsum(counts) - counts
    return np.arange(total, dtype=np.int64) - np.repeat(starts, counts)


def fill_from_received(local_src, rank, p_local, recv_ranges):
    """recv_ranges: list over source ranks of (lo, hi) int64 arrays in record order.  Record r
    (numbered over sources in rank order) fills the local slots [lo - start, hi - start)."""
    start = rank * p_local
    r0 = 0
    for lo, up in recv_ranges:
        n = len(lo)
        if n:
            counts = up - lo
            local_src[np.repeat(lo - start, counts) + _ranges(counts)] = -(np.repeat(np.arange(r0, r0 + n), counts) + 1)
        r0 += n
    if (local_src == UNASSIGNED).any():
        raise RuntimeError("sharded resample: %d local slots were not assigned an ancestor"
                           % int((local_src == UNASSIGNED).sum()))
    return local_src, r0


# --------------------------------------------------------------------------- communicators
class TorchComm(object):
    """torch.distributed wrapper: 'nccl' (= RCCL) with device tensors, 'gloo' with host tensors."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.on_device = self.backend == "nccl"
        self.device = torch.device("cuda", device if device is not None else torch.cuda.current_device()) \
            if self.on_device else torch.device("cpu")

    def allreduce_max(self, v):
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allreduce_sum(self, a):
        t = self.torch.as_tensor(np.asarray(a, dtype=np.float64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather(self, a):
        t = self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)
        out = self.torch.empty(self.world * t.numel(), dtype=self.torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(out, t) if self.on_device else self.dist.all_gather(
            list(out.view(self.world, -1).unbind(0)), t)
        return out.cpu().numpy()

    def alltoall_i64(self, per_dest):
        """per_dest: list of int64 arrays (one per destination) -> list per source."""
        if not self.on_device:
            gathered = [None] * self.world
            self.dist.all_gather_object(gathered, [np.asarray(a, dtype=np.int64) for a in per_dest])
            return [gathered[src][self.rank] for src in range(self.world)]
        torch = self.torch
        counts = torch.tensor([len(a) for a in per_dest], dtype=torch.int64, device=self.device)
        rcounts = torch.empty_like(counts)
        self.dist.all_to_all_single(rcounts, counts)
        rc = rcounts.cpu().tolist()
        send = torch.as_tensor(np.concatenate([np.asarray(a, dtype=np.int64) for a in per_dest])
                               if sum(len(a) for a in per_dest) else np.empty(0, np.int64)).to(self.device)
        recv = torch.empty(int(sum(rc)), dtype=torch.int64, device=self.device)
        self.dist.all_to_all_single(recv, send, rc, [len(a) for a in per_dest])
        out, o = [], 0
        r = recv.cpu().numpy()
        for c in rc:
            out.append(r[o:o + c])
            o += c
        return out

    def alltoall_records(self, send_buf, send_counts, recv_counts, record_bytes):
        """send_buf: torch uint8 tensor on the GPU with the packed records in destination order
        (a NumPy uint8 array from the CPU test backend).  Returns the received records in source
        order, same kind of buffer."""
        torch = self.torch
        n_recv = int(sum(recv_counts))
        if isinstance(send_buf, np.ndarray):
            parts, o = [], 0
            for c in send_counts:
                parts.append(send_buf[o:o + c * record_bytes].copy())
                o += c * record_bytes
            gathered = [None] * self.world
            self.dist.all_gather_object(gathered, parts)
            if not n_recv:
                return np.empty(0, np.uint8)
            return np.concatenate([gathered[src][self.rank] for src in range(self.world)])
        recv = torch.empty(max(n_recv, 1) * record_bytes, dtype=torch.uint8, device=send_buf.device)
        if self.on_device:
            self.dist.all_to_all_single(recv[: n_recv * record_bytes], send_buf[: int(sum(send_counts)) * record_bytes],
                                        [c * record_bytes for c in recv_counts], [c * record_bytes for c in send_counts])
            return recv
        host = send_buf.cpu().numpy()
        parts, o = [], 0
        for c in send_counts:
            parts.append(host[o:o + c * record_bytes])
            o += c * record_bytes
        gathered = [None] * self.world
        self.dist.all_gather_object(gathered, parts)
        got = np.concatenate([gathered[src][self.rank] for src in range(self.world)]) if n_recv else np.empty(0, np.uint8)
        if n_recv:
            recv[: n_recv * record_bytes] = torch.as_tensor(got).to(send_buf.device)
        return recv

    def barrier(self):
        self.dist.barrier()


# --------------------------------------------------------------------------- the filter
class HipShard(_lib.DeviceFilter):
    """The product backend of a shard: the HIP library plus torch-allocated device buffers for
    the migrating particle records (torch is plumbing here: device memory for the collective)."""

    def __init__(self, num_particles, num_landmarks, device=0):
        import torch

        super().__init__(num_particles, num_landmarks, device=device)
        # One stream for the kernels, the record buffers and the collectives: torch orders
        # RCCL work and allocator reuse against the *current* stream, so the library must run
        # on it too, or its kernels would race the all-to-all that fills the buffers they read.
        self.torch = torch
        self.stream = torch.cuda.Stream(device=torch.device("cuda", device))
        self.set_stream(self.stream.cuda_stream)

    def on_stream(self):
        return self.torch.cuda.stream(self.stream)

    def alloc_records(self, n):
        torch = self.torch
        return torch.empty(max(int(n), 1) * self.particle_bytes(), dtype=torch.uint8,
                           device=torch.device("cuda", self.device))

    def pack_records(self, local_idx, buf):
        self.pack_particles(local_idx, int(buf.data_ptr()))

    def adopt_records(self, src, buf, n_received):
        self.adopt_particles(src, int(buf.data_ptr()) if (buf is not None and n_received) else 0, n_received)


class ShardedFilter(object):
    """One shard of a FastSLAM filter: same methods as ``_lib.DeviceFilter`` for what bench.py
    and the tests use, with the resample made global across ranks."""

    def __init__(self, particles_per_rank, num_landmarks, device=0, comm=None, shard=None):
        self.comm = comm if comm is not None else TorchComm(device)
        self.rank, self.world = self.comm.rank, self.comm.world
        self.P = int(particles_per_rank)
        self.L = int(num_landmarks)
        self.P_global = self.P * self.world
        # the compute backend is the HIP library; tests may inject an object with the same
        # shard_* / pack / adopt methods to exercise the exchange logic without a GPU
        self.f = shard if shard is not None else HipShard(self.P, self.L, device=device)
        self.f.set_shard(self.rank * self.P)
        self._recv_keepalive = None
        self.last_migrated = 0

    # -- pass-throughs -------------------------------------------------------------
    def upload_map(self, *a, **k):
        return self.f.upload_map(*a, **k)

    def set_measurement_noise(self, Qt):
        return self.f.set_measurement_noise(Qt)

    def set_option(self, name, value):
        return self.f.set_option(name, value)

    def reset_weights(self):
        return self.f.reset_weights()

    def motion(self, v, w, dt, z=None, seed=0, draw=0):
        return self.f.motion(v, w, dt, z=z, seed=seed, draw=draw)

    def observe(self, blobs, ids=None, return_ids=False):
        out = self.f.observe(blobs, ids=ids, return_ids=return_ids)
        self._recv_keepalive = None  # adopted slots were rewritten into the shard's own map
        return out

    def download_poses(self):
        return self.f.download_poses()

    def download_landmarks(self, *a, **k):
        out = self.f.download_landmarks(*a, **k)
        self._recv_keepalive = None
        return out

    def synchronize(self):
        return self.f.synchronize()

    def enable_timing(self, mask=True):
        return self.f.enable_timing(mask)

    def reset_timings(self):
        return self.f.reset_timings()

    def timings(self):
        return self.f.timings()

    def close(self):
        self.f.close()

    # -- the coupled part ----------------------------------------------------------
    def resample(self, u, domain=_lib.PK_WEIGHTS_LINEAR, return_ancestors=False):
        """Global systematic resample (prkt_core_v2.py:210-252) with a replicated draw u."""
        ctx = self.f.on_stream() if hasattr(self.f, "on_stream") else _nullcontext()
        with ctx:
            return self._resample(u, domain, return_ancestors)

    def _resample(self, u, domain, return_ancestors):
        f, comm = self.f, self.comm
        gmax = comm.allreduce_max(f.shard_max_logw()) if domain == _lib.PK_WEIGHTS_LOG else 0.0
        totals = f.shard_block_totals(gmax, domain)
        nb = totals.size
        gtotals = comm.allgather(totals) if self.world > 1 else totals
        hi = f.shard_offspring(gtotals, self.rank * nb, self.P_global, u, self.rank == self.world - 1)
        local_src, sends = plan_exchange(hi, self.rank, self.world, self.P)
        rec_bytes = f.particle_bytes()
        send_counts = [len(s[0]) for s in sends]
        recv_lo = comm.alltoall_i64([s[1] for s in sends]) if self.world > 1 else [np.empty(0, np.int64)]
        recv_hi = comm.alltoall_i64([s[2] for s in sends]) if self.world > 1 else [np.empty(0, np.int64)]
        recv_counts = [len(a) for a in recv_lo]
        local_src, n_recv = fill_from_received(local_src, self.rank, self.P, list(zip(recv_lo, recv_hi)))
        n_send = int(sum(send_counts))
        self.last_migrated = n_send
        recv = None
        if self.world > 1:  # every rank takes part in the exchange, even with nothing to move
            send_buf = f.alloc_records(n_send)
            if n_send:
                f.pack_records(np.concatenate([s[0] for s in sends]), send_buf)
            recv = comm.alltoall_records(send_buf, send_counts, recv_counts, rec_bytes)
        f.adopt_records(local_src, recv, n_recv)
        self._recv_keepalive = recv if n_recv else None
        if return_ancestors:
            return self._global_ancestors(hi, local_src)
        return None

    def _global_ancestors(self, hi, local_src):
        # global ancestor index of every local output slot, for tests: reconstruct from all hi
        allhi = self.comm.allgather(hi[1:].astype(np.float64)).astype(np.int64) if self.world > 1 else hi[1:]
        allhi = np.maximum.accumulate(allhi)
        slots = np.arange(self.rank * self.P, (self.rank + 1) * self.P)
        return np.searchsorted(allhi, slots, side="right").astype(np.int64)

    def step(self, v, w, dt, blobs, u, z=None, seed=0, draw=0, ids=None, domain=_lib.PK_WEIGHTS_LINEAR):
        self.reset_weights()
        self.motion(v, w, dt, z=z, seed=seed, draw=draw)
        self.observe(blobs, ids=ids)
        self.resample(u, domain=domain)

    def summary(self):
        """FastSLAM.summary (prkt_core_v2.py:254-276) over all shards: all-reduce of four sums."""
        s = self.f.pose_sums()
        if self.world > 1:
            ctx = self.f.on_stream() if hasattr(self.f, "on_stream") else _nullcontext()
            with ctx:
                s = self.comm.allreduce_sum(s)
        n = float(self.P_global)
        return float(s[0] / n), float(s[1] / n), float(np.arctan2(s[2], s[3]))
