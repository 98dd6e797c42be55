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
     about weights.  An all-gather of (2 x world) particle-range bounds sizes step 4.
  4. all-to-all of the particles whose slots belong to another shard   (pose + landmark map;
     the only bandwidth-relevant traffic, and only for migrating particles); each record's
     header names the slots it fills, so no per-particle metadata travels separately

All per-step tensors live on the GPU and the kernels that fill / consume them are enqueued on
the same stream as the collectives; the host synchronises once per resample.

The draw u is replicated (same value on every rank), as the north star asks.  With shards
that are multiples of 1024 particles the ancestors are bit-identical to the 1-GPU run.

``ShardedFilter.step`` overlaps step 4 with the next step's work (``split_step``, on when the scan takes the register
route): the output slots a shard fills with its OWN particles form one contiguous run (ancestors are monotone in the slot
index), so the motion update and k_step_regs start on that run while the migrating particles -- a whole map each, the
bulk of a step's bytes on the wire -- are still travelling, and the slots at either end follow when they have arrived.

Shards of any other size take the GLOBAL-SCAN variant of steps 2-3 (``global_scan``; chosen by
itself when P_local % 1024 != 0): the ranks all-gather their log-weights (8 B per particle of
the whole filter; the migrating maps of step 4 are 10^4 times that) and every rank runs the
1-GPU scan kernels over the whole array -- the same 1024-particle blocks, the same additions,
hence the same bits as one GPU -- and then derives its own particles' output slots from it.

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


# --------------------------------------------------------------------------- balanced placement (minimum migration)
def balanced_tables(n, p_local):
    """Per-rank bookkeeping of the balanced placement, from n[r] = children of rank r's particles (sum = world * p_local).

    Rank r keeps the first m[r] = min(n[r], p_local) of its children in its physical slots [0, m[r]); the other e[r] (its
    EXCESS) leave; a rank with n[r] < p_local has dd[r] = p_local - n[r] free slots [m[r], p_local) (its DEFICIT).  All excess
    children, numbered E = ebase[r] + (q - p_local) in (rank, child position q) order, fill all free slots, numbered
    D = dbase[r] + (slot - m[r]) in (rank, slot) order: child E goes to free slot D = E.  Nothing else moves, and
    sum(e) is the least number of children that can change rank with p_local slots per rank."""
    n = np.asarray(n, dtype=np.int64)
    m = np.minimum(n, p_local)
    e = n - m
    dd = p_local - m
    ebase = np.concatenate([[0], np.cumsum(e)[:-1]]).astype(np.int64)
    dbase = np.concatenate([[0], np.cumsum(dd)[:-1]]).astype(np.int64)
    return m, e, dd, ebase, dbase


def plan_balanced(H, logical_all, world, p_local):
    """The whole balanced plan in NumPy -- the readable reference of pk_shard_plan_balanced_dev, and what the CPU tests check
    the device tables against.

    H: int64[Pg + 1], the single-filter offspring table in LOGICAL order (logical particle l fills output slots
       [H[l], H[l + 1]) -- prkt_core_v2.py:233-250's ordered walk); logical_all: int64[Pg], the logical index of the particle
       in physical place g = rank * p_local + j.

    Returns (cq, nz, tables, pairs): cq int64[Pg + 1] the exclusive prefix of the children counts in PHYSICAL order, nz the
    exclusive prefix of "has children" (a particle without children is never packed: after a resample three of four are
    dead); tables = (n,) + balanced_tables(...); pairs int64[world, world, 2]: rank s sends the particles number [a0, a1)
    of ITS LIST OF PARTICLES WITH CHILDREN to rank d."""
    H = np.asarray(H, dtype=np.int64)
    logical_all = np.asarray(logical_all, dtype=np.int64)
    cnt = H[logical_all + 1] - H[logical_all]
    cq = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    nz = np.concatenate([[0], np.cumsum(cnt > 0)]).astype(np.int64)
    bounds = cq[np.arange(world + 1) * p_local]
    n = np.diff(bounds)
    m, e, dd, ebase, dbase = balanced_tables(n, p_local)
    pairs = np.zeros((world, world, 2), dtype=np.int64)
    for s in range(world):
        rel = cq[s * p_local:(s + 1) * p_local + 1] - bounds[s]  # child positions of rank s's particles, rank-relative
        nzr = nz[s * p_local:(s + 1) * p_local + 1] - nz[s * p_local]
        for d in range(world):
            lo, up = max(ebase[s], dbase[d]), min(ebase[s] + e[s], dbase[d] + dd[d])
            if s == d or up <= lo:
                continue
            q0, q1 = p_local + lo - ebase[s], p_local + up - ebase[s]
            j0 = np.searchsorted(rel[1:], q0, side="right")  # first j with rel[j + 1] > q0 (it has children)
            j1 = np.searchsorted(rel[:-1], q1, side="left")  # first j with rel[j] >= q1
            pairs[s, d] = nzr[j0], nzr[j1]
    return cq, nz, (n, m, e, dd, ebase, dbase), pairs


def balanced_record_ranges(rel, Hl, alive, a0, a1, p_local, ebase_s, dbase_d, dd_d, m_d):
    """Headers of the records rank s packs for rank d: for the particles j = alive[a0 .. a1) the free slots [lo, up) of rank d
    their excess children fill and klo, the logical index of the child in slot lo (the child in slot k is klo + k - lo).
    rel: the rank-relative exclusive child-position prefix (p_local + 1), Hl[j] = H[logical[j]], alive: the rank's particles
    with children, ascending.  Returns (j, lo, up, klo); every range is non-empty."""
    j = np.asarray(alive, dtype=np.int64)[a0:a1]
    e0 = ebase_s + np.maximum(rel[j], p_local) - p_local
    e1 = ebase_s + np.maximum(rel[j + 1], p_local) - p_local
    elo = np.maximum(e0, dbase_d)
    eup = np.maximum(np.minimum(e1, dbase_d + dd_d), elo)
    lo = m_d + elo - dbase_d
    up = m_d + eup - dbase_d
    klo = Hl[j] + (elo - ebase_s + p_local - rel[j])
    return j, lo, up, klo


# --------------------------------------------------------------------------- communicators
class TorchComm(object):
    """torch.distributed on torch tensors, in place.  'nccl' (= RCCL over xGMI) runs on the device
    tensors directly; 'gloo' (tests) stages device tensors through the host."""

    def __init__(self):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.direct = dist.get_backend() == "nccl"

    def _stage(self, t):
        return t if (self.direct or not t.is_cuda) else t.cpu()

    def all_reduce_max_(self, t):
        h = self._stage(t)
        self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX)
        if h is not t:
            t.copy_(h)

    def all_reduce_sum_(self, t):
        h = self._stage(t)
        self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
        if h is not t:
            t.copy_(h)

    def all_gather_(self, out, t):
        """out: world * t.numel() elements, rank-major."""
        h_in = self._stage(t)
        if self.direct:
            self.dist.all_gather_into_tensor(out, t)
            return
        h_out = out if not out.is_cuda else self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather(list(h_out.view(self.world, -1).unbind(0)), h_in)
        if h_out is not out:
            out.copy_(h_out)

    def all_to_all_records(self, send, send_counts, recv_counts, record_bytes):
        """send: uint8 tensor with the packed records in destination-rank order.  Returns a uint8
        tensor on the same device with the received records in source-rank order."""
        torch = self.torch
        n_recv, n_send = int(sum(recv_counts)), int(sum(send_counts))
        recv = torch.empty(max(n_recv, 1) * record_bytes, dtype=torch.uint8, device=send.device)
        if self.direct:
            self.dist.all_to_all_single(recv[: n_recv * record_bytes], send[: n_send * record_bytes],
                                        [c * record_bytes for c in recv_counts],
                                        [c * record_bytes for c in send_counts])
            return recv
        # the gloo stand-in (tests, rehearsals on one GPU): pairwise host transfers -- a rank holds its own send and receive
        # bytes only (an all_gather_object of everybody's records took world x that on every rank and brought a box down at
        # four ranks x 100 000 particles)
        host = send[: n_send * record_bytes].cpu()
        hrecv = torch.empty(max(n_recv, 1) * record_bytes, dtype=torch.uint8)
        ops, so, ro = [], 0, 0
        for peer in range(self.world):
            sb, rb = send_counts[peer] * record_bytes, recv_counts[peer] * record_bytes
            if peer == self.rank:
                if sb:
                    hrecv[ro:ro + rb] = host[so:so + sb]
            else:
                if sb:
                    ops.append(self.dist.P2POp(self.dist.isend, host[so:so + sb], peer))
                if rb:
                    ops.append(self.dist.P2POp(self.dist.irecv, hrecv[ro:ro + rb], peer))
            so += sb
            ro += rb
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        if n_recv:
            recv[: n_recv * record_bytes] = hrecv[: n_recv * record_bytes].to(send.device)
        return recv

    def all_to_all_records_async(self, send, send_counts, recv_counts, record_bytes):
        """The same, not waited for: returns (recv, work); ``work.wait()`` makes the current stream wait for the transfer
        (None: already complete -- the gloo stand-in of the tests is synchronous)."""
        if not self.direct:
            return self.all_to_all_records(send, send_counts, recv_counts, record_bytes), None
        torch = self.torch
        n_recv, n_send = int(sum(recv_counts)), int(sum(send_counts))
        recv = torch.empty(max(n_recv, 1) * record_bytes, dtype=torch.uint8, device=send.device)
        work = self.dist.all_to_all_single(recv[: n_recv * record_bytes], send[: n_send * record_bytes],
                                           [c * record_bytes for c in recv_counts],
                                           [c * record_bytes for c in send_counts], async_op=True)
        return recv, work

    def barrier(self):
        self.dist.barrier()


# --------------------------------------------------------------------------- the filter
class HipShard(_lib.DeviceFilter):
    """The product backend of a shard: the HIP library plus torch-allocated device buffers for
    the small per-step exchange tensors and the migrating particle records (torch is plumbing
    here: device memory the collectives can run on)."""

    def __init__(self, num_particles, num_landmarks, device=0):
        import torch

        super().__init__(num_particles, num_landmarks, device=device)
        # One stream for the kernels, the buffers and the collectives: torch orders RCCL work
        # and allocator reuse against the *current* stream, so the library must run on it too,
        # or its kernels would race the all-to-all that fills the buffers they read.
        self.torch = torch
        self.tdev = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=self.tdev)
        self.set_stream(self.stream.cuda_stream)
        self._host_reads = {}

    def on_stream(self):
        return self.torch.cuda.stream(self.stream)

    def new_f64(self, n):
        return self.torch.zeros(int(n), dtype=self.torch.float64, device=self.tdev)

    def new_i64(self, n):
        return self.torch.zeros(int(n), dtype=self.torch.int64, device=self.tdev)

    def alloc_records(self, n):
        return self.torch.empty(max(int(n), 1) * self.particle_bytes(), dtype=self.torch.uint8, device=self.tdev)

    def max_logw_into(self, t):
        self.shard_max_logw_dev(t.data_ptr())

    def block_totals_into(self, gmax_t, domain, totals_t):
        self.shard_block_totals_dev(gmax_t.data_ptr() if gmax_t is not None else 0, domain, totals_t.data_ptr())

    def plan_into(self, gtotals_t, first_block, global_particles, u, last_shard, world, ranges_t):
        self.shard_plan_dev(gtotals_t.data_ptr(), gtotals_t.numel(), first_block, global_particles, u, last_shard, world,
                            ranges_t.data_ptr())

    def logw_into(self, t):
        self.shard_logw_dev(t.data_ptr())

    def plan_global_into(self, glogw_t, global_particles, gmax_t, domain, u, last_shard, world, ranges_t):
        self.shard_plan_global_dev(glogw_t.data_ptr(), global_particles, gmax_t.data_ptr() if gmax_t is not None else 0, domain,
                                   u, last_shard, world, ranges_t.data_ptr())

    def pack_into(self, ranges, world, rank, buf):
        self.shard_pack_dev(ranges, world, rank, buf.data_ptr())

    def adopt_from(self, rank, recv, n_received):
        self.shard_adopt_dev(rank, recv.data_ptr() if (recv is not None and n_received) else 0, n_received)

    # -- balanced placement --
    def state_into(self, t):
        self.shard_state_dev(t.data_ptr())

    def plan_balanced_into(self, gstate_t, global_particles, gmax_t, domain, u, world, rank, table_t):
        self._bal_pg = int(global_particles)
        self.shard_plan_balanced_dev(gstate_t.data_ptr(), global_particles, gmax_t.data_ptr() if gmax_t is not None else 0, domain, u,
                                     world, rank, table_t.data_ptr())

    def pack_balanced_into(self, table, world, rank, buf):
        self.shard_pack_balanced_dev(table, world, rank, buf.data_ptr())

    def adopt_balanced(self, table, world, rank, recv, n_received, mode=0):
        self.shard_adopt_balanced_dev(table, world, rank, recv.data_ptr() if (recv is not None and n_received) else 0, n_received,
                                      mode)

    def download_balanced_offspring(self):
        return self.shard_download_balanced_offspring(self._bal_pg)

    # -- the split step --
    def local_span_into(self, t2):
        self.shard_local_span_dev(t2.data_ptr())

    def adopt_local(self, rank):
        self.shard_adopt_local_dev(rank)

    def adopt_remote(self, rank, recv, n_received):
        self.shard_adopt_remote_dev(rank, recv.data_ptr() if (recv is not None and n_received) else 0, n_received)

    def start_host_read(self, t):
        """Begin copying a small device tensor to pinned host memory on the shard's stream; returns a
        handle for finish_host_read.  Lets the caller do host work before it has to wait."""
        torch = self.torch
        # the pinned landing buffer and the event are kept (a pinned allocation per step costs tens of
        # microseconds of host time in front of the next step's launches); one read is pending at a time
        key = (tuple(t.shape), t.dtype)
        slot = self._host_reads.get(key)
        if slot is None:
            slot = (torch.empty(t.shape, dtype=t.dtype, pin_memory=True), torch.cuda.Event())
            self._host_reads[key] = slot
        host, ev = slot
        host.copy_(t, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.tdev))
        return host, ev

    @staticmethod
    def finish_host_read(handle):
        host, ev = handle
        ev.synchronize()
        return host.numpy().copy()  # the buffer is reused by the next read


class ShardedFilter(object):
    """One shard of a FastSLAM filter: same methods as ``_lib.DeviceFilter`` for what bench.py
    and the tests use, with the resample made global across ranks."""

    def __init__(self, particles_per_rank, num_landmarks, device=0, comm=None, shard=None, global_scan=None, split_step=None,
                 loopback=None, placement=None):
        self.comm = comm if comm is not None else TorchComm()
        self.rank, self.world = self.comm.rank, self.comm.world
        self.P = int(particles_per_rank)
        self.L = int(num_landmarks)
        self.P_global = self.P * self.world
        # the compute backend is the HIP library; tests may inject an object with the same
        # *_into / pack_into / adopt_from methods to exercise the exchange logic without a GPU
        self.f = shard if shard is not None else HipShard(self.P, self.L, device=device)
        self.f.set_shard(self.rank * self.P)
        self.nb = (self.P + SCAN_BLOCK - 1) // SCAN_BLOCK
        f = self.f
        self._gmax = f.new_f64(1)
        self._totals = f.new_f64(self.nb)
        self._gtotals = f.new_f64(self.nb * self.world)
        # shards that do not end on scan-block boundaries: scan the whole filter's weights on every rank (module docstring)
        self.global_scan = (self.P % SCAN_BLOCK != 0 and self.world > 1) if global_scan is None else bool(global_scan)
        if self.global_scan:
            self._logw = f.new_f64(self.P)
            self._glogw = f.new_f64(self.P_global) if self.world > 1 else self._logw
        # per destination (j0, j1), then the global bounds of the slots this shard's own particles fill (the split step)
        self._row = 2 * self.world + 2
        self._ranges = f.new_i64(self._row)
        self._all_ranges = f.new_i64(self._row * self.world)
        self.split_step = (self.world > 1) if split_step is None else bool(split_step)
        self.split_steps_done = 0
        # debug (one-rank tests): (n_front, n_back) -- the split step treats that many slots at either end of the shard as
        # "remote": the particles that fill them are packed, sent through the asynchronous all-to-all TO THIS RANK ITSELF and
        # adopted from the received records, so pack -> all_to_all_single(async_op) -> wait -> adopt_remote runs with records
        # that really travelled even where only one device exists
        self.loopback = tuple(int(v) for v in loopback) if loopback is not None else None
        self.loopback_records = 0
        # "balanced" (the default between ranks): the physical slots carry their LOGICAL index (the index the particle has in
        # one filter holding everything -- what keys the Philox streams and orders the weight scan); a rank keeps its own
        # children and ships only its excess to whichever rank is short (balanced_tables above).  "contiguous": rank r
        # holds the logical slots [r P, (r + 1) P), whatever that moves (rounds 1-4).
        if placement is None:
            placement = "balanced" if (self.world > 1 and self.loopback is None and hasattr(f, "plan_balanced_into")) else "contiguous"
        if placement not in ("balanced", "contiguous"):
            raise ValueError("placement must be 'balanced' or 'contiguous'")
        # The balanced loopback (round 6; one rank, debug): (0, n_back) -- the rank keeps its children [0, P - n_back) and sends the
        # rest, as records with the balanced protocol's 64-byte header (and the bookkeeping's tail), through the all-to-all TO ITSELF;
        # the state all-gather runs through the communicator as well.  Everything the default placement does between ranks then runs
        # over RCCL where only one device exists.
        if placement == "balanced" and self.loopback is not None and (self.world != 1 or self.loopback[0] != 0):
            raise ValueError("the balanced loopback is (0, n_back) on a world of one")
        self.placement = placement
        self._bal_loop = placement == "balanced" and self.loopback is not None
        if placement == "balanced":
            self._state = f.new_f64(2 * self.P)  # [log-weights | logical indices (int64 bits)]
            self._gstate = f.new_f64(2 * self.P * self.world) if (self.world > 1 or self._bal_loop) else self._state
            self._brow = 2 * self.world + 4
            self._btable = f.new_i64(self._brow * self.world)
        self._sums = f.new_f64(4)
        self._recv_keepalive = None
        self._pending = None  # a resample whose exchange has been planned on the GPU but not carried out yet
        self.last_migrated = 0
        self.total_migrated = 0  # particles this rank sent to other ranks since the counter was last cleared (bench.py)

    def _ctx(self):
        return self.f.on_stream() if hasattr(self.f, "on_stream") else _nullcontext()

    # -- pass-throughs -------------------------------------------------------------
    def upload_map(self, *a, **k):
        return self.f.upload_map(*a, **k)

    def set_measurement_noise(self, Qt):
        return self.f.set_measurement_noise(Qt)

    def set_option(self, name, value):
        return self.f.set_option(name, value)

    def reset_weights(self):
        self._complete()
        return self.f.reset_weights()

    def motion(self, v, w, dt, z=None, seed=0, draw=0):
        """z (the reference's RNG mode): the normals of THIS rank's particles in physical order (P rows), or those of the
        whole filter in logical order (P_global rows: each rank takes the rows of the particles it holds)."""
        self._complete()
        return self.f.motion(v, w, dt, z=self._my_rows(z), seed=seed, draw=draw)

    def _my_rows(self, z):
        if z is None:
            return None
        z = np.asarray(z, dtype=np.float64)
        if z.shape[0] == self.P_global and self.world > 1:
            return np.ascontiguousarray(z[self.logical_index()])
        return z

    def logical_index(self):
        """int64[P]: the logical index (the particle's index in one filter holding everything) of every physical slot."""
        self._complete()
        if self.placement == "balanced":
            return np.asarray(self.f.download_logical(), dtype=np.int64)
        return self.rank * self.P + np.arange(self.P, dtype=np.int64)

    def reset_placement(self):
        """Physical slot j holds logical particle rank * P + j again (after a state upload in that order)."""
        self._complete()
        if self.placement == "balanced":
            self.f.reset_placement()

    def observe(self, blobs, ids=None, return_ids=False, fresh=False):
        self._complete()
        out = self.f.observe(blobs, ids=ids, return_ids=return_ids, fresh=fresh) if fresh else \
            self.f.observe(blobs, ids=ids, return_ids=return_ids)
        self._recv_keepalive = None  # adopted slots were rewritten into the shard's own map (stream-ordered free)
        return out

    def download_poses(self):
        self._complete()
        return self.f.download_poses()

    def upload_poses(self, xyhw):
        self._complete()
        return self.f.upload_poses(xyhw)

    def upload_pose(self, p, xyhw):
        self._complete()
        return self.f.upload_pose(p, xyhw)

    def upload_landmarks(self, p0, p1, means=None, covs=None, counts=None):
        self._complete()
        out = self.f.upload_landmarks(p0, p1, means, covs, counts)
        self._recv_keepalive = None  # (the upload materialised the map: adopted records were copied into the shard's own slots)
        return out

    def download_landmarks(self, *a, **k):
        self._complete()
        out = self.f.download_landmarks(*a, **k)
        self._recv_keepalive = None
        return out

    # ---- new landmarks on the device: every particle's bookkeeping rides behind its map in the exchange ----
    def grow_enable(self, preset_landmarks, reading_capacity=64, pair_threshold=30.0):
        if self.placement != "balanced":
            raise ValueError("grow_enable: the new-landmark bookkeeping travels with the balanced placement (placement='balanced')")
        self._complete()
        self.split_step = False  # (the bookkeeping kernel wants the ids in HBM: whole observes on the general route)
        return self.f.grow_enable(preset_landmarks, reading_capacity, pair_threshold)

    def grow_download(self, *a, **k):
        self._complete()
        return self.f.grow_download(*a, **k)

    def grow_upload(self, *a, **k):
        self._complete()
        return self.f.grow_upload(*a, **k)

    def synchronize(self):
        self._complete()
        return self.f.synchronize()

    def enable_timing(self, mask=True):
        return self.f.enable_timing(mask)

    def reset_timings(self):
        return self.f.reset_timings()

    def timings(self):
        return self.f.timings()

    def observe_route(self):
        return self.f.observe_route() if hasattr(self.f, "observe_route") else "none"

    def observe_published(self):
        """Which instance of the register route worked on this shard's last scan (pk_observe_published)."""
        return bool(self.f.observe_published()) if hasattr(self.f, "observe_published") else False

    def observe_flagged(self):
        """(particles of THIS shard the one-pass kernel handed on, candidate-list overflows) -- pk_observe_flagged."""
        return self.f.observe_flagged() if hasattr(self.f, "observe_flagged") else (0, 0)

    def close(self):
        self._pending = None
        self.f.close()

    # -- the coupled part ----------------------------------------------------------
    def resample(self, u, domain=_lib.PK_WEIGHTS_LINEAR, return_ancestors=False, defer=False):
        """Global systematic resample (prkt_core_v2.py:210-252) with a replicated draw u.
        One host synchronisation per call: reading the (2 x world x world) table of particle
        ranges that sizes the all-to-all.  defer=True returns once the plan is enqueued; the
        exchange is carried out by the next call that needs the particles (step() stages the
        next scan on the host in between)."""
        self._complete()
        with self._ctx():
            self._plan(u, domain)
        if defer and not return_ancestors:
            return None
        self._complete()
        if return_ancestors:
            return self._global_ancestors(u, domain)
        return None

    def _plan(self, u, domain):
        f, comm, W, R = self.f, self.comm, self.world, self.rank
        gmax = None
        if domain == _lib.PK_WEIGHTS_LOG:
            gmax = self._gmax
            f.max_logw_into(gmax)
            if W > 1:
                comm.all_reduce_max_(gmax)
        if self.placement == "balanced":
            # ONE all-gather (16 B per particle of the whole filter); every rank then runs the 1-GPU scan on the weights in
            # logical order and derives the whole plan -- who keeps what, who sends which particles to whom -- by itself
            f.state_into(self._state)
            if W > 1 or self._bal_loop:
                comm.all_gather_(self._gstate, self._state)
            f.plan_balanced_into(self._gstate, self.P_global, gmax, domain, u, W, R, self._btable)
            handle = f.start_host_read(self._btable) if hasattr(f, "start_host_read") else None
            self._pending = (self._btable, handle)
            return
        if self.global_scan:
            f.logw_into(self._logw)
            if W > 1:
                comm.all_gather_(self._glogw, self._logw)
            f.plan_global_into(self._glogw, self.P_global, gmax, domain, u, R == W - 1, W, self._ranges)
        else:
            f.block_totals_into(gmax, domain, self._totals)
            if W > 1:
                comm.all_gather_(self._gtotals, self._totals)
                gtot = self._gtotals
            else:
                gtot = self._totals
            f.plan_into(gtot, R * self.nb, self.P_global, u, R == W - 1, W, self._ranges)
        if hasattr(f, "local_span_into"):
            f.local_span_into(self._ranges[2 * W:])
        if W > 1:
            comm.all_gather_(self._all_ranges, self._ranges)
            table = self._all_ranges
        else:
            table = self._ranges
        # the one host read of the step, started now and waited for in _complete()
        handle = f.start_host_read(table) if hasattr(f, "start_host_read") else None
        self._pending = (table, handle)

    def _read_plan(self):
        """The pending plan's range table on the host: (pairs[source][destination] -> (j0, j1), send / receive counts of this
        rank, number of records that change rank anywhere, local slot run [a, b) of this rank)."""
        table, handle = self._pending
        self._pending = None
        f, W, R = self.f, self.world, self.rank
        host = f.finish_host_read(handle) if handle is not None else table.cpu().numpy()
        if self.placement == "balanced":
            rows = np.asarray(host, dtype=np.int64).reshape(W, self._brow)
            allr = rows[:, :2 * W].reshape(W, W, 2)
            counts = allr[:, :, 1] - allr[:, :, 0]
            send_counts = [int(counts[R, d]) for d in range(W)]
            recv_counts = [int(counts[s, R]) for s in range(W)]
            self._bhost = rows
            return rows, send_counts, recv_counts, int(counts.sum()), 0, int(rows[R, 2 * W + 1])
        rows = np.asarray(host).reshape(W, self._row)
        allr = rows[:, :2 * W].reshape(W, W, 2)  # [source][destination] -> (j0, j1)
        counts = allr[:, :, 1] - allr[:, :, 0]
        send_counts = [int(counts[R, d]) if d != R else 0 for d in range(W)]
        recv_counts = [int(counts[s, R]) if s != R else 0 for s in range(W)]
        moving = int(counts.sum() - np.trace(counts))  # same number on every rank (all-gathered table)
        lo, up = int(rows[R, 2 * W]) - R * self.P, int(rows[R, 2 * W + 1]) - R * self.P
        a = min(max(lo, 0), self.P)
        b = min(max(up, a), self.P)
        return allr, send_counts, recv_counts, moving, a, b

    def _complete(self):
        """Carry out the exchange of a planned resample: read the range table, all-to-all of the
        migrating particles, adoption."""
        if self._pending is None:
            return
        with self._ctx():
            f, comm, W, R = self.f, self.comm, self.world, self.rank
            allr, send_counts, recv_counts, moving, _a, _b = self._read_plan()
            n_send, n_recv = sum(send_counts), sum(recv_counts)
            self.last_migrated = n_send
            self.total_migrated += n_send
            recv = None
            bal = self.placement == "balanced"
            if W > 1 and moving > 0:  # every rank takes part in the exchange, even with nothing of its own to move
                send = f.alloc_records(n_send)
                if n_send:
                    if bal:
                        f.pack_balanced_into(allr.reshape(-1), W, R, send)
                    else:
                        f.pack_into(allr[R].reshape(-1), W, R, send)
                recv = comm.all_to_all_records(send, send_counts, recv_counts, f.particle_bytes())
            elif self._bal_loop:
                send, n_recv, _keep = self._pack_balanced_loop()
                recv = comm.all_to_all_records(send, [n_recv], [n_recv], f.particle_bytes())
                self.loopback_records += n_recv
            if bal:
                f.adopt_balanced(allr.reshape(-1), W, R, recv, n_recv, 0)
            else:
                f.adopt_from(R, recv, n_recv)
            self._recv_keepalive = recv if n_recv else None

    def _complete_and_step_split(self, v, w, dt, seed, draw):
        """The pending exchange AND the motion + observe of the next step, overlapped: the particles that stay on this rank
        (output slots [a, b) of the new generation) are moved and observed while the migrating ones travel; the slots at
        either end follow.  The staged scan must take the register route."""
        with self._ctx():
            f, comm, W, R = self.f, self.comm, self.world, self.rank
            allr, send_counts, recv_counts, moving, a, b = self._read_plan()
            n_send, n_recv = sum(send_counts), sum(recv_counts)
            self.last_migrated = n_send
            self.total_migrated += n_send
            recv, work = None, None
            bal = self.placement == "balanced"
            if W > 1 and moving > 0:
                send = f.alloc_records(n_send)
                if n_send:  # from the OLD generation: before the adoption below
                    if bal:
                        f.pack_balanced_into(allr.reshape(-1), W, R, send)
                    else:
                        f.pack_into(allr[R].reshape(-1), W, R, send)
                recv, work = comm.all_to_all_records_async(send, send_counts, recv_counts, f.particle_bytes())
            elif self._bal_loop:
                # one rank, balanced placement: its children from position keep on through the exchange, to itself (see __init__)
                send, n_recv, keep = self._pack_balanced_loop()
                recv, work = comm.all_to_all_records_async(send, [n_recv], [n_recv], f.particle_bytes())
                a, b = 0, keep
                self.loopback_records += n_recv
            elif W == 1 and self.loopback is not None and sum(self.loopback) > 0:
                # one rank, slots [0, n_front) and [P - n_back, P) through the exchange (see __init__)
                nf, nb = self.loopback
                lo_s, hi_s = min(nf, self.P), max(self.P - nb, min(nf, self.P))
                hi = np.maximum.accumulate(f.shard_download_offspring())
                jf0, jf1 = int(np.searchsorted(hi[1:], 0, side="right")), int(np.searchsorted(hi[:-1], lo_s, side="left"))
                jb0, jb1 = int(np.searchsorted(hi[1:], hi_s, side="right")), int(np.searchsorted(hi[:-1], self.P, side="left"))
                jf1, jb1 = max(jf1, jf0), max(jb1, jb0)
                n_recv = (jf1 - jf0) + (jb1 - jb0)
                send = f.alloc_records(n_recv)
                stride = f.particle_bytes()
                if jf1 > jf0:
                    f.shard_pack_slots_dev(jf0, jf1, 0, lo_s, send.data_ptr())
                if jb1 > jb0:
                    f.shard_pack_slots_dev(jb0, jb1, hi_s, self.P, send.data_ptr() + (jf1 - jf0) * stride)
                recv, work = comm.all_to_all_records_async(send, [n_recv], [n_recv], stride)
                f.set_option("split_loopback_lo", lo_s)
                f.set_option("split_loopback_hi", hi_s)
                a, b = lo_s, hi_s
                self.loopback_records += n_recv
            if bal:  # the rank's own children fill [0, m) = [a, b); what it is sent fills [m, P)
                f.adopt_balanced(allr.reshape(-1), W, R, None, 0, 1)
            else:
                f.adopt_local(R)
            if recv is None:  # nobody changes rank this time: every slot is filled locally, one launch over all of them
                if bal:
                    f.adopt_balanced(allr.reshape(-1), W, R, None, 0, 2)
                else:
                    f.adopt_remote(R, None, 0)
                f.motion_range(v, w, dt, seed, draw, 0, self.P)
                f.observe_staged_range(True, 0, self.P, True, True)
            else:
                f.motion_range(v, w, dt, seed, draw, a, b)
                f.observe_staged_range(True, a, b, True, False)  # weight reset (:73) fused in; consumes the staged scan
                if work is not None:
                    work.wait()  # the stream waits for the records, the host does not
                if bal:
                    f.adopt_balanced(allr.reshape(-1), W, R, recv, n_recv, 2)
                else:
                    f.adopt_remote(R, recv, n_recv)
                f.motion_range(v, w, dt, seed, draw, 0, a)
                f.motion_range(v, w, dt, seed, draw, b, self.P)
                f.observe_staged_range(True, 0, a, False, False)
                f.observe_staged_range(True, b, self.P, False, True)
            self._recv_keepalive = recv if n_recv else None  # read by the launches above: freed in stream order later
            self.split_steps_done += 1

    def _pack_balanced_loop(self):
        """The balanced loopback's records (debug, one rank): the particles whose children reach beyond position keep = P - n_back,
        packed for this rank's own slots [keep, P).  Returns (send buffer, number of records, keep).  Reads the plan's tables on the
        host: a debug path."""
        f = self.f
        keep = max(self.P - self.loopback[1], 0)
        rel, _Hl, alive = f.shard_download_balanced_plan()
        first = np.searchsorted(rel[alive + 1], keep, side="right") if len(alive) else 0  # first alive particle with a child at position >= keep
        a0, a1 = int(first), int(len(alive))
        if keep >= self.P:
            a0 = a1
        n = a1 - a0
        send = f.alloc_records(n)
        if n:
            f.shard_pack_balanced_loop_dev(keep, a0, a1, send.data_ptr())
        f.set_option("balanced_loopback_keep", keep)
        return send, n, keep

    def _global_ancestors(self, u, domain):
        """Tests only: global ancestor index of every local output slot, from the host-array
        variant of the same plan (pk_shard_offspring).  Balanced placement: the ancestors' LOGICAL indices, in the physical
        order of this rank's slots (logical_index() says which output slot of the single filter each of them is)."""
        if self.placement == "balanced":
            H = np.maximum.accumulate(np.asarray(self.f.download_balanced_offspring(), dtype=np.int64))
            return np.searchsorted(H[1:], self.logical_index(), side="right").astype(np.int64)
        if self.global_scan:
            hi = self.f.shard_download_offspring()  # the table the plan itself used
        else:
            gt = self._gtotals if self.world > 1 else self._totals
            hi = self.f.shard_offspring(gt.cpu().numpy(), self.rank * self.nb, self.P_global, u, self.rank == self.world - 1)
        allhi = self.f.new_f64(self.P * self.world)
        mine = self.f.new_f64(self.P)
        mine.copy_(self._to_dev(hi[1:].astype(np.float64), mine))
        if self.world > 1:
            self.comm.all_gather_(allhi, mine)
        else:
            allhi = mine
        allhi = np.maximum.accumulate(allhi.cpu().numpy().astype(np.int64))
        slots = np.arange(self.rank * self.P, (self.rank + 1) * self.P)
        return np.searchsorted(allhi, slots, side="right").astype(np.int64)

    @staticmethod
    def _to_dev(a, like):
        import torch

        return torch.from_numpy(np.ascontiguousarray(a)).to(like.device)

    def step(self, v, w, dt, blobs, u, z=None, seed=0, draw=0, ids=None, domain=_lib.PK_WEIGHTS_LINEAR):
        """One cam_cb.  The host half of the scan upload (association tables) is done BEFORE the
        previous step's exchange is completed, i.e. while the GPU is still busy with that step."""
        staged = False
        if ids is None and hasattr(self.f, "stage_scan") and len(blobs) > 0:
            self.f.stage_scan(blobs)
            staged = True
        if (staged and self.split_step and z is None and self._pending is not None and hasattr(self.f, "staged_takes_regs")
                and self.f.staged_takes_regs()):
            self._complete_and_step_split(v, w, dt, seed, draw)
        else:
            self._complete()
            self.f.motion(v, w, dt, z=self._my_rows(z), seed=seed, draw=draw)
            if staged:
                self.f.observe_staged(fresh=True)  # weight reset (:73) fused into the observe kernels
                self._recv_keepalive = None
            else:
                self.observe(blobs, ids=ids, fresh=True)
        self.resample(u, domain=domain, defer=True)

    def summary(self):
        """FastSLAM.summary (prkt_core_v2.py:254-276) over all shards: all-reduce of four sums."""
        self._complete()
        s = self.f.pose_sums()
        if self.world > 1:
            with self._ctx():
                self._sums.copy_(self._to_dev(np.asarray(s, dtype=np.float64), self._sums))
                self.comm.all_reduce_sum_(self._sums)
                s = self._sums.cpu().numpy()
        n = float(self.P_global)
        return float(s[0] / n), float(s[1] / n), float(np.arctan2(s[2], s[3]))
