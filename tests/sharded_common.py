"""Shared pieces of the sharded-resample tests (CPU gloo and single-GPU gloo)."""
import math
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.fastslam_oracle import OracleFilter, synthetic_scan, synthetic_world, truth_step  # noqa: E402

SCAN_BLOCK = 1024


class OracleShard(object):
    """TEST-ONLY compute backend for ShardedFilter: the NumPy oracle for the per-particle work
    plus a NumPy restatement of the shard_* protocol (block totals in 1024-particle blocks,
    sequential scan of the global totals, owner-computes offspring).  Lets the exchange logic
    run under gloo on a machine without a GPU.  Never used by the product."""

    def __init__(self, P, means, covs, immutable=None):
        self.P = P
        self.o = OracleFilter(P, means, covs, immutable)
        self.L = self.o.L
        self.off = 0
        self.device = -1

    def set_shard(self, off):
        self.off = off
        self.logical = off + np.arange(self.P, dtype=np.int64)  # balanced placement: logical index of the particle in place j

    def upload_map(self, *a, **k):
        pass

    def reset_weights(self):
        self.o.reset_weights()

    def motion(self, v, w, dt, z=None, seed=0, draw=0):
        self.o.motion(v, w, dt, z)

    def download_logical(self):
        return self.logical.copy()

    def reset_placement(self):
        self.logical = self.off + np.arange(self.P, dtype=np.int64)

    def observe(self, blobs, ids=None, return_ids=False, fresh=False):
        if fresh:
            self.o.reset_weights()
        return self.o.observe(blobs, ids)

    def download_poses(self):
        return np.stack([self.o.x, self.o.y, self.o.h, self.o.weights()], 1)

    def upload_poses(self, xyhw):
        a = np.asarray(xyhw, dtype=np.float64).reshape(self.P, 4)
        self.o.x, self.o.y, self.o.h = a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()
        with np.errstate(divide="ignore"):
            self.o.logw = np.log(a[:, 3])

    def upload_pose(self, p, xyhw):
        self.o.x[p], self.o.y[p], self.o.h[p] = float(xyhw[0]), float(xyhw[1]), float(xyhw[2])
        with np.errstate(divide="ignore"):
            self.o.logw[p] = np.log(float(xyhw[3]))

    def upload_landmarks(self, p0, p1, means=None, covs=None, counts=None):
        n = p1 - p0
        if means is not None:
            self.o.mean[p0:p1] = np.asarray(means, dtype=np.float64).reshape(n, self.L, 5)
        if covs is not None:
            self.o.cov[p0:p1] = np.asarray(covs, dtype=np.float64).reshape(n, self.L, 5, 5)
        if counts is not None:
            self.o.count[p0:p1] = np.asarray(counts).reshape(n, self.L)

    def set_measurement_noise(self, Qt):
        self.o.Qt = np.asarray(Qt, dtype=np.float64).reshape(4, 4)

    def download_landmarks(self, p0=0, p1=None):
        p1 = self.P if p1 is None else p1
        return self.o.mean[p0:p1].copy(), self.o.cov[p0:p1].copy(), self.o.count[p0:p1].copy()

    def pose_sums(self):
        return np.array([self.o.x.sum(), self.o.y.sum(), np.sin(self.o.h).sum(), np.cos(self.o.h).sum()])

    # ---- protocol ----
    def shard_max_logw(self):
        return float(self.o.logw.max())

    def _w(self, gmax, domain):
        return np.exp(self.o.logw - (gmax if domain == 1 and gmax > -np.inf else 0.0))

    def shard_block_totals(self, gmax, domain):
        w = self._w(gmax, domain)
        nb = (self.P + SCAN_BLOCK - 1) // SCAN_BLOCK
        self.clocal = np.empty(self.P)
        tot = np.empty(nb)
        for b in range(nb):
            c = np.cumsum(w[b * SCAN_BLOCK:(b + 1) * SCAN_BLOCK])
            self.clocal[b * SCAN_BLOCK:b * SCAN_BLOCK + len(c)] = c
            tot[b] = c[-1]
        return tot

    def shard_offspring(self, gtotals, first_block, P_global, u, last_shard):
        offs = np.concatenate([[0.0], np.cumsum(gtotals)])
        S = offs[-1]
        r = S / float(P_global)
        t = u * r + np.arange(P_global, dtype=np.float64) * r
        C = offs[first_block + np.arange(self.P) // SCAN_BLOCK] + self.clocal
        hi = np.empty(self.P + 1, dtype=np.int64)
        hi[0] = 0 if first_block == 0 else np.searchsorted(t, offs[first_block], side="right")
        hi[1:] = np.searchsorted(t, C, side="right")
        if last_shard:
            hi[-1] = P_global
        return hi

    # ---- tensor-based protocol (mirrors HipShard; torch CPU tensors stand in for HBM) ----
    def new_f64(self, n):
        import torch

        return torch.zeros(int(n), dtype=torch.float64)

    def new_i64(self, n):
        import torch

        return torch.zeros(int(n), dtype=torch.int64)

    HEAD = 8  # words in a record's header: x, y, h, logw, lo, up, klo, spare

    def particle_bytes(self):
        return 8 * (self.HEAD + self.L * 30 + self.L)

    def alloc_records(self, n):
        import torch

        return torch.zeros(max(int(n), 1) * self.particle_bytes(), dtype=torch.uint8)

    def max_logw_into(self, t):
        t[0] = self.shard_max_logw()

    def block_totals_into(self, gmax_t, domain, totals_t):
        import torch

        totals_t.copy_(torch.from_numpy(self.shard_block_totals(float(gmax_t[0]) if gmax_t is not None else 0.0, domain)))

    def plan_into(self, gtotals_t, first_block, P_global, u, last_shard, world, ranges_t):
        self.hi = self.shard_offspring(gtotals_t.numpy(), first_block, P_global, u, last_shard)
        self._ranges_from_hi(world, ranges_t)

    # ---- global-scan variant (shards of any size): the 1-GPU blocked scan on the all-gathered log-weights ----
    def logw_into(self, t):
        import torch

        t.copy_(torch.from_numpy(self.o.logw.copy()))

    def plan_global_into(self, glogw_t, P_global, gmax_t, domain, u, last_shard, world, ranges_t):
        glogw = glogw_t.numpy()
        gmax = float(gmax_t[0]) if gmax_t is not None else 0.0
        w = np.exp(glogw - (gmax if domain == 1 and gmax > -np.inf else 0.0))
        nb = (P_global + SCAN_BLOCK - 1) // SCAN_BLOCK
        C = np.empty(P_global)
        run = 0.0
        for b in range(nb):  # block-local inclusive sums + the sequential exclusive scan of the block totals
            c = np.cumsum(w[b * SCAN_BLOCK:(b + 1) * SCAN_BLOCK])
            C[b * SCAN_BLOCK:b * SCAN_BLOCK + len(c)] = run + c
            run = run + c[-1]
        r = run / float(P_global)
        t = u * r + np.arange(P_global, dtype=np.float64) * r
        off, P = self.off, self.P
        hi = np.empty(P + 1, dtype=np.int64)
        hi[0] = 0 if off == 0 else np.searchsorted(t, C[off - 1], side="right")
        hi[1:] = np.searchsorted(t, C[off:off + P], side="right")
        if last_shard:
            hi[-1] = P_global
        self.hi = hi
        self._ranges_from_hi(world, ranges_t)

    def shard_download_offspring(self):
        return self.hi.copy()

    def local_span_into(self, t2):
        t2[0] = int(self.hi[0])
        t2[1] = int(self.hi[self.P])

    def _ranges_from_hi(self, world, ranges_t):
        hi, P = self.hi, self.P
        for d in range(world):
            start, end = d * P, (d + 1) * P
            j0 = int(np.searchsorted(hi[1:], start, side="right"))
            j1 = int(np.searchsorted(hi[:-1], end, side="left"))
            ranges_t[2 * d] = j0
            ranges_t[2 * d + 1] = max(j1, j0)

    def _record(self, j, lo, up, klo=0):
        o = self.o
        head = np.array([o.x[j], o.y[j], o.h[j], o.logw[j], 0.0, 0.0, 0.0, 0.0])
        head[4:7] = np.array([lo, up, klo], dtype=np.int64).view(np.float64)
        return np.concatenate([head, o.mean[j].ravel(), o.cov[j].ravel(), o.count[j].astype(np.float64)])

    def pack_into(self, ranges, world, rank, buf):
        rb, P, hi = self.particle_bytes(), self.P, self.hi
        out = buf.numpy()
        i = 0
        for d in range(world):
            if d == rank:
                continue
            for j in range(int(ranges[2 * d]), int(ranges[2 * d + 1])):
                lo = max(int(hi[j]), d * P)
                up = max(min(int(hi[j + 1]), (d + 1) * P), lo)
                out[i * rb:(i + 1) * rb] = self._record(j, lo, up).view(np.uint8)
                i += 1

    def adopt_from(self, rank, recv, n_received):
        o, L, rb, P, hi = self.o, self.L, self.particle_bytes(), self.P, self.hi
        buf = recv.numpy() if recv is not None else None
        recs = [buf[r * rb:(r + 1) * rb].view(np.float64) for r in range(n_received)]
        rhi = np.array([int(r[4:6].view(np.int64)[1]) for r in recs], dtype=np.int64)
        x, y, h, lw = o.x.copy(), o.y.copy(), o.h.copy(), o.logw.copy()
        mean, cov, cnt = o.mean.copy(), o.cov.copy(), o.count.copy()
        for k in range(P):
            K = rank * P + k
            if hi[0] <= K < hi[P]:
                a = int(np.searchsorted(hi[1:], K, side="right"))
                o.x[k], o.y[k], o.h[k], o.logw[k] = x[a], y[a], h[a], lw[a]
                o.mean[k], o.cov[k], o.count[k] = mean[a], cov[a], cnt[a]
            else:
                rec = recs[int(np.searchsorted(rhi, K, side="right"))]
                lo_, up_ = rec[4:6].view(np.int64)
                assert lo_ <= K < up_
                o.x[k], o.y[k], o.h[k], o.logw[k] = rec[:4]
                self._take_record(k, rec)

    def _take_record(self, k, rec):
        o, L, H = self.o, self.L, self.HEAD
        o.x[k], o.y[k], o.h[k], o.logw[k] = rec[:4]
        o.mean[k] = rec[H:H + 5 * L].reshape(L, 5)
        o.cov[k] = rec[H + 5 * L:H + 30 * L].reshape(L, 5, 5)
        o.count[k] = rec[H + 30 * L:].astype(np.int64)

    # ---- balanced placement (minimum migration): NumPy mirror of pk_shard_state_dev / pk_shard_plan_balanced_dev /
    # pk_shard_pack_balanced_dev / pk_shard_adopt_balanced_dev ----
    def state_into(self, t):
        import torch

        t[:self.P] = torch.from_numpy(self.o.logw.copy())
        t[self.P:] = torch.from_numpy(self.logical.view(np.float64).copy())

    def plan_balanced_into(self, gstate_t, P_global, gmax_t, domain, u, world, rank, table_t):
        from parakeet_slam_amd.sharded import plan_balanced

        P = self.P
        g = gstate_t.numpy().reshape(world, 2 * P)
        glogw_phys = g[:, :P].reshape(-1)
        glog = g[:, P:].copy().view(np.int64).reshape(-1)
        glogw = np.empty(P_global)
        glogw[glog] = glogw_phys  # the single filter's order
        gmax = float(gmax_t[0]) if gmax_t is not None else 0.0
        w = np.exp(glogw - (gmax if domain == 1 and gmax > -np.inf else 0.0))
        nb = (P_global + SCAN_BLOCK - 1) // SCAN_BLOCK
        C = np.empty(P_global)
        run = 0.0
        for b in range(nb):  # the 1-GPU blocked scan
            c = np.cumsum(w[b * SCAN_BLOCK:(b + 1) * SCAN_BLOCK])
            C[b * SCAN_BLOCK:b * SCAN_BLOCK + len(c)] = run + c
            run = run + c[-1]
        r = run / float(P_global)
        t = u * r + np.arange(P_global, dtype=np.float64) * r
        H = np.empty(P_global + 1, dtype=np.int64)
        H[0] = 0
        H[1:] = np.searchsorted(t, C, side="right")
        H[-1] = P_global
        H = np.maximum.accumulate(H)
        cq, _nz, (n, m, e, dd, ebase, dbase), pairs = plan_balanced(H, glog, world, P)
        rel = cq[rank * P:(rank + 1) * P + 1] - cq[rank * P]
        self.bal = dict(H=H, Hl=H[self.logical], rel=rel, alive=np.nonzero(np.diff(rel) > 0)[0], n=n, m=m, e=e, dd=dd,
                        ebase=ebase, dbase=dbase)
        row = 2 * world + 4
        tab = np.zeros((world, row), dtype=np.int64)
        tab[:, :2 * world] = pairs.reshape(world, 2 * world)
        tab[:, 2 * world], tab[:, 2 * world + 1], tab[:, 2 * world + 2], tab[:, 2 * world + 3] = n, m, ebase, dbase
        import torch

        table_t.copy_(torch.from_numpy(tab.reshape(-1)))

    def download_balanced_offspring(self):
        return self.bal["H"].copy()

    def pack_balanced_into(self, table, world, rank, buf):
        from parakeet_slam_amd.sharded import balanced_record_ranges

        b, rb, P = self.bal, self.particle_bytes(), self.P
        tab = np.asarray(table).reshape(world, 2 * world + 4)
        out = buf.numpy()
        i = 0
        for d in range(world):
            a0, a1 = int(tab[rank, 2 * d]), int(tab[rank, 2 * d + 1])
            if d == rank or a1 <= a0:
                continue
            js, lo, up, klo = balanced_record_ranges(b["rel"], b["Hl"], b["alive"], a0, a1, P, int(b["ebase"][rank]),
                                                     int(b["dbase"][d]), int(b["dd"][d]), int(b["m"][d]))
            for j, a, c, k in zip(js, lo, up, klo):
                assert c > a  # no record travels without a child at the destination
                out[i * rb:(i + 1) * rb] = self._record(int(j), int(a), int(c), int(k)).view(np.uint8)
                i += 1

    def adopt_balanced(self, table, world, rank, recv, n_received, mode=0):
        """mode 0: every slot; 1: the slots [0, m) this rank fills with its own children (and the new generation becomes
        current); 2: the slots [m, P) from the received records."""
        o, b, P, rb = self.o, self.bal, self.P, self.particle_bytes()
        m = int(b["m"][rank])
        if mode != 2:
            rel, Hl = b["rel"], b["Hl"]
            k = np.arange(m)
            a = np.searchsorted(rel[1:], k, side="right")  # the particle whose children hold position k
            self._old = (o.x.copy(), o.y.copy(), o.h.copy(), o.logw.copy(), o.mean.copy(), o.cov.copy(), o.count.copy())
            x, y, h, lw, mean, cov, cnt = self._old
            o.x[:m], o.y[:m], o.h[:m], o.logw[:m] = x[a], y[a], h[a], lw[a]
            o.mean[:m], o.cov[:m], o.count[:m] = mean[a], cov[a], cnt[a]
            newlog = self.logical.copy()
            newlog[:m] = Hl[a] + (k - rel[a])
            anc = np.full(P, -1, dtype=np.int64)
            anc[:m] = self.logical[a]
            self.logical, self.anc_logical = newlog, anc
        if mode != 1 and m < P:
            buf = recv.numpy()
            recs = [buf[r * rb:(r + 1) * rb].view(np.float64) for r in range(n_received)]
            heads = np.array([r[4:7].view(np.int64) for r in recs], dtype=np.int64).reshape(-1, 3)
            for k in range(m, P):
                r = int(np.searchsorted(heads[:, 1], k, side="right"))
                lo_, up_, klo = heads[r]
                assert lo_ <= k < up_, (k, lo_, up_)
                self._take_record(k, recs[r])
                self.logical[k] = klo + (k - lo_)


def scenario(L, steps, seed=5):
    means, covs = synthetic_world(L)
    pose = (0.0, 0.0, 0.0)
    scans = []
    for _ in range(steps):
        pose = truth_step(pose, 0.2, 0.1, 0.1)
        scans.append(synthetic_scan(means, pose))
    return means, covs, scans


def noise(P_global, steps, seed):
    rs = np.random.RandomState(seed)
    return rs.standard_normal((steps, P_global, 3)), rs.uniform(0, 1, steps)


def init_gloo(rank, world, store_path):
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method="file://" + store_path, rank=rank, world_size=world)


def store_file():
    fd, path = tempfile.mkstemp(prefix="pk_gloo_")
    os.close(fd)
    os.unlink(path)
    return path


def make_oracle_shard(P, means, covs, immutable=None):
    """Shard factory for ShardedFastSLAM's CPU rehearsal (module-level: picklable for the spawned ranks)."""
    return OracleShard(P, means, covs, immutable)


class GrowingOracleShard(OracleShard):
    """OracleShard + the new-landmark bookkeeping: GrowingOracle per shard, and the bookkeeping of a migrating
    particle behind its map in the record -- the NumPy mirror of pk_grow_enable / k_new_landmarks / k_grow_gather and of the tail
    pk_shard_pack_balanced_dev appends (here as float64 words: counters (4) | slot ids (S) | readings (R x 8))."""

    grow = None

    def grow_enable(self, preset_landmarks, reading_capacity=64, pair_threshold=30.0):
        from oracle.fastslam_oracle import GrowingOracle

        g = GrowingOracle.__new__(GrowingOracle)  # around the shard's own OracleFilter (its map already holds the spare slots)
        g.f, g.L0, g.spare, g.thr = self.o, int(preset_landmarks), self.L - int(preset_landmarks), float(pair_threshold)
        g.hyp = [[] for _ in range(self.P)]
        g.next_id = [g.L0 + 1] * self.P
        g.used = [0] * self.P
        g.slot_id = [dict() for _ in range(self.P)]
        self.grow, self.R = g, int(reading_capacity)

    def grow_shape(self):
        return (self.grow.L0, self.grow.spare, self.R) if self.grow else (0, 0, 0)

    def observe(self, blobs, ids=None, return_ids=False, fresh=False):
        if self.grow is None or ids is not None or len(blobs) == 0:
            return OracleShard.observe(self, blobs, ids, return_ids, fresh)
        if fresh:
            self.o.reset_weights()
        return self.grow.observe(blobs)

    def _tail_words(self):
        return 4 + self.grow.spare + 8 * self.R if self.grow else 0

    def particle_bytes(self):
        return OracleShard.particle_bytes(self) + 8 * self._tail_words()

    def _tail(self, j):
        g = self.grow
        t = np.zeros(self._tail_words())
        assert len(g.hyp[j]) <= self.R, "the rehearsal's ring is too small for this scene"
        t[:3] = (len(g.hyp[j]), g.used[j], g.next_id[j])
        for k in range(g.used[j]):
            t[4 + k] = g.slot_id[j][g.L0 + k]
        if g.hyp[j]:
            t[4 + g.spare:4 + g.spare + 8 * len(g.hyp[j])] = np.asarray(g.hyp[j], dtype=np.float64).ravel()
        return t

    def _set_tail(self, k, t):
        g = self.grow
        n, used = int(t[0]), int(t[1])
        g.next_id[k], g.used[k] = int(t[2]), used
        g.slot_id[k] = {g.L0 + i: int(t[4 + i]) for i in range(used)}
        rd = t[4 + g.spare:4 + g.spare + 8 * n].reshape(n, 8)
        g.hyp[k] = [(int(r[0]),) + tuple(float(v) for v in r[1:]) for r in rd]

    POT = 0x40000000  # PK_LANDMARK_POTENTIAL: the device keeps the potential flag in the count word, so do the records here

    def _record(self, j, lo, up, klo=0):
        rec = OracleShard._record(self, j, lo, up, klo)
        if self.grow is None:
            return rec
        rec[self.HEAD + 30 * self.L:] += np.where(self.o.potential[j], float(self.POT), 0.0)
        return np.concatenate([rec, self._tail(j)])

    def _take_record(self, k, rec):
        if self.grow is None:
            return OracleShard._take_record(self, k, rec)
        n = self._tail_words()
        OracleShard._take_record(self, k, rec[:-n])
        self.o.potential[k] = (self.o.count[k] & self.POT) != 0
        self.o.count[k] &= ~self.POT
        self._set_tail(k, rec[-n:])

    def download_landmarks(self, p0=0, p1=None):
        m, c, k = OracleShard.download_landmarks(self, p0, p1)
        p1 = self.P if p1 is None else p1
        return m, c, k | np.where(self.o.potential[p0:p1], self.POT, 0)

    def upload_landmarks(self, p0, p1, means=None, covs=None, counts=None):
        OracleShard.upload_landmarks(self, p0, p1, means, covs, None)
        if counts is not None:
            k = np.asarray(counts).reshape(p1 - p0, self.L).astype(np.int64)
            self.o.count[p0:p1] = k & ~self.POT
            self.o.potential[p0:p1] = (k & self.POT) != 0

    def adopt_balanced(self, table, world, rank, recv, n_received, mode=0):
        if self.grow is not None:
            assert mode == 0
            g, b = self.grow, self.bal
            m = int(b["m"][rank])
            a = np.searchsorted(b["rel"][1:], np.arange(m), side="right")  # the particle whose children hold position k
            old = (g.hyp, g.next_id, g.used, g.slot_id)
            g.hyp = [list(old[0][i]) for i in a] + [[] for _ in range(self.P - m)]
            g.next_id = [old[1][i] for i in a] + [0] * (self.P - m)
            g.used = [old[2][i] for i in a] + [0] * (self.P - m)
            g.slot_id = [dict(old[3][i]) for i in a] + [dict() for _ in range(self.P - m)]
            pot = self.o.potential.copy()
            self.o.potential[:m] = pot[a]
        return OracleShard.adopt_balanced(self, table, world, rank, recv, n_received, mode)

    def adopt_from(self, rank, recv, n_received):
        assert self.grow is None, "the bookkeeping travels with the balanced placement only"
        return OracleShard.adopt_from(self, rank, recv, n_received)

    def grow_download(self, p0=0, p1=None, counters=True, readings=True, slot_ids=True):
        g = self.grow
        p1 = self.P if p1 is None else p1
        n = p1 - p0
        cnt = np.zeros((n, 4), dtype=np.int32)
        rd = np.zeros((n, self.R, 8))
        sid = np.zeros((n, g.spare), dtype=np.int32)
        for i in range(n):
            t = self._tail(p0 + i)
            cnt[i] = t[:4]
            sid[i] = t[4:4 + g.spare]
            rd[i] = t[4 + g.spare:].reshape(self.R, 8)
        return cnt, rd, sid

    def grow_upload(self, p0, p1, counters=None, readings=None, slot_ids=None):
        g = self.grow
        for i in range(p1 - p0):
            t = np.concatenate([np.asarray(counters[i], dtype=np.float64), np.asarray(slot_ids[i], dtype=np.float64),
                                np.asarray(readings[i], dtype=np.float64).ravel()])
            self._set_tail(p0 + i, t)


def make_growing_oracle_shard(P, means, covs, immutable=None):
    return GrowingOracleShard(P, means, covs, immutable)


make_growing_oracle_shard.grows = True  # (ShardedFastSLAM(new_landmarks=True) asks the factory whether its shards can)
