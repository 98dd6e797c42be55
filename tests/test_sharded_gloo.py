"""CPU, world_size 2 and 4 over gloo: the multi-GPU path of the resample.

The exchange logic (plan, metadata all-to-all, record all-to-all, adoption) is the product's
own (parakeet_slam_amd/sharded.py); the per-particle arithmetic is supplied by the test-only
OracleShard so this runs without a GPU.  Claims checked: G shards give the SAME ancestors,
poses, weights and landmark maps as one filter holding all particles, including when weights
are so skewed that nearly every particle migrates."""
import numpy as np
import pytest
import torch.multiprocessing as mp

from sharded_common import OracleShard, init_gloo, noise, scenario, store_file
from oracle.fastslam_oracle import OracleFilter


def reference_run(P, L, steps, skew):
    means, covs, scans = scenario(L, steps)
    z, us = noise(P, steps, 11)
    o = OracleFilter(P, means, covs)
    out = []
    for s in range(steps):
        o.reset_weights()
        o.motion(0.2, 0.1, 0.1, z[s])
        o.observe(scans[s], ids=np.arange(1, L + 1))
        if skew:
            o.logw += np.linspace(0.0, skew, P)  # heavier weights at high indices: mass moves down-rank
        # canonical blocked scan (same association as the device / OracleShard)
        sh = OracleShard(P, means, covs)
        sh.o = o
        tot = sh.shard_block_totals(float(o.logw.max()), 1)
        hi = sh.shard_offspring(tot, 0, P, float(us[s]), True)
        anc = np.minimum(np.searchsorted(np.maximum.accumulate(hi[1:]), np.arange(P), side="right"), P - 1)
        o.gather(anc)
        out.append((anc.copy(), o.x.copy(), o.y.copy(), o.h.copy(), o.logw.copy(), o.mean.copy(), o.count.copy()))
    return out


def worker(rank, world, store, P_local, L, steps, skew, q, placement="contiguous"):
    try:
        init_gloo(rank, world, store)
        from parakeet_slam_amd.sharded import ShardedFilter, TorchComm

        means, covs, scans = scenario(L, steps)
        P = P_local * world
        z, us = noise(P, steps, 11)
        sf = ShardedFilter(P_local, L, comm=TorchComm(), shard=OracleShard(P_local, means, covs), placement=placement)
        assert sf.placement == placement
        res = []
        for s in range(steps):
            sf.reset_weights()
            sf.motion(0.2, 0.1, 0.1, z=z[s])  # the whole filter's normals in logical order: the rank takes its particles' rows
            sf.observe(scans[s], ids=np.arange(1, L + 1))
            if skew:
                sf.f.o.logw += np.linspace(0.0, skew, P)[sf.logical_index()]
            anc = sf.resample(float(us[s]), domain=1, return_ancestors=True)
            o = sf.f.o
            res.append((anc, o.x.copy(), o.y.copy(), o.h.copy(), o.logw.copy(), o.mean.copy(), o.count.copy(),
                        sf.last_migrated, sf.summary(), sf.logical_index()))
        q.put((rank, res))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + traceback.format_exc()))


CASES = [(2, 1024, 0.0), (2, 1024, 6.0), (4, 1024, 3.0), (2, 300, 2.0), (4, 300, 3.0), (3, 1500, 6.0), (8, 1024, 3.0), (8, 300, 3.0),
         (8, 300, 12.0)]


@pytest.mark.parametrize("placement", ["balanced", "contiguous"])
@pytest.mark.parametrize("world,P_local,skew", CASES)
def test_shards_reproduce_single_filter(world, P_local, skew, placement):
    """G shards give the single filter's ancestors, poses, weights and maps exactly.  Contiguous placement: rank r holds
    the logical slots [r P, (r + 1) P).  Balanced placement (the default): each physical slot carries its logical index and
    only a rank's excess children move; the comparison is made in logical order."""
    L, steps = 6, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    store = store_file()
    procs = [ctx.Process(target=worker, args=(r, world, store, P_local, L, steps, skew, q, placement)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res = q.get(timeout=300)
        assert not isinstance(res, str), res
        got[r] = res
    for p in procs:
        p.join(timeout=60)
    P = world * P_local
    ref = reference_run(P, L, steps, skew)
    migrated = 0
    for s in range(steps):
        logical = np.concatenate([got[r][s][9] for r in range(world)])
        assert np.array_equal(np.sort(logical), np.arange(P)), "the logical indices must be a permutation of the filter"
        if placement == "contiguous":
            assert np.array_equal(logical, np.arange(P))

        def whole_of(fld):  # the field of every rank, put into the single filter's order
            cat = np.concatenate([got[r][s][fld] for r in range(world)])
            out = np.empty_like(cat)
            out[logical] = cat
            return out

        anc = whole_of(0)
        # any shard size: the 1024-aligned ones through the block-total plan, the others through the global scan
        assert np.array_equal(anc, ref[s][0]), "ancestors differ from the single-filter run"
        for fld in range(1, 7):
            assert np.array_equal(whole_of(fld), ref[s][fld]), (s, fld)
        migrated += sum(got[r][s][7] for r in range(world))
        # summaries agree across ranks and with the concatenated state
        x = whole_of(1)
        h = whole_of(3)
        for r in range(world):
            sm = got[r][s][8]
            assert abs(sm[0] - x.mean()) < 1e-12 and abs(sm[2] - np.arctan2(np.sin(h).sum(), np.cos(h).sum())) < 1e-12
        # every output slot got exactly one ancestor, ancestors are monotone (systematic resampling)
        assert np.all(np.diff(anc) >= 0) and anc.min() >= 0 and anc.max() < P
    if skew >= 3.0:
        assert migrated > 0, "the skewed case must actually move particles between ranks"
    _MIGRATED[(world, P_local, skew, placement)] = migrated
    other = _MIGRATED.get((world, P_local, skew, "contiguous" if placement == "balanced" else "balanced"))
    if other is not None:  # the balanced placement never moves more records than the contiguous one
        bal, con = (migrated, other) if placement == "balanced" else (other, migrated)
        assert bal <= con, (bal, con)


_MIGRATED = {}


def test_plan_exchange_covers_every_slot():
    from parakeet_slam_amd.sharded import fill_from_received, plan_exchange

    rs = np.random.RandomState(0)
    world, P_local = 4, 50
    P = world * P_local
    for trial in range(50):
        w = np.exp(rs.normal(0, rs.uniform(0.1, 5), P))
        c = np.cumsum(w)
        r = c[-1] / P
        t = rs.uniform() * r + np.arange(P) * r
        hi_all = np.searchsorted(t, c, side="right")
        hi_all[-1] = P
        anc = np.searchsorted(hi_all, np.arange(P), side="right")
        plans = []
        for rank in range(world):
            hi = np.concatenate([[0 if rank == 0 else hi_all[rank * P_local - 1]], hi_all[rank * P_local:(rank + 1) * P_local]])
            plans.append(plan_exchange(hi, rank, world, P_local))
        for rank in range(world):
            local_src = plans[rank][0].copy()
            ranges, owners = [], []
            for src in range(world):
                idx, lo, up = plans[src][1][rank]
                ranges.append((lo, up))
                owners.extend((src * P_local + idx).tolist())
            local_src, n = fill_from_received(local_src, rank, P_local, ranges)
            glob = np.where(local_src >= 0, rank * P_local + local_src, 0)
            for k in np.nonzero(local_src < 0)[0]:
                glob[k] = owners[-(local_src[k] + 1)]
            assert np.array_equal(glob, anc[rank * P_local:(rank + 1) * P_local]), (trial, rank)


def test_plan_balanced_places_every_child_exactly_once():
    """sharded.plan_balanced (the readable reference of pk_shard_plan_balanced_dev) on random weights, random placements and world sizes
    1-8: the new logical indices are a permutation of the filter, a rank keeps min(n, P) of its own children, every record fills free
    slots of its destination and none travels without a child, the records reach a destination in slot order (what the adoption's
    binary search over the received headers relies on), and what moves is exactly the ranks' excess."""
    from parakeet_slam_amd.sharded import balanced_record_ranges, plan_balanced

    rs = np.random.RandomState(1)
    for trial in range(200):
        W = int(rs.choice([1, 2, 3, 4, 8]))
        P = int(rs.choice([1, 2, 5, 50, 300]))
        Pg = W * P
        w = np.exp(rs.normal(0, rs.uniform(0.1, 6), Pg))
        if rs.uniform() < 0.2:  # most particles dead
            w[rs.uniform(size=Pg) < 0.7] = 0
            w[rs.randint(Pg)] = 1
        c = np.cumsum(w)
        r = c[-1] / Pg
        t = rs.uniform() * r + np.arange(Pg) * r
        H = np.concatenate([[0], np.searchsorted(t, c, side="right")])
        H[-1] = Pg
        H = np.maximum.accumulate(H)
        logical = rs.permutation(Pg)
        cq, nz, (n, m, e, dd, eb, db), pairs = plan_balanced(H, logical, W, P)
        assert n.sum() == Pg and e.sum() == dd.sum() and np.array_equal(m, np.minimum(n, P))
        newlog = np.full(Pg, -1)
        moved = 0
        for R in range(W):
            rel = cq[R * P:(R + 1) * P + 1] - cq[R * P]
            Hl = H[logical[R * P:(R + 1) * P]]
            alive = np.nonzero(np.diff(rel) > 0)[0]
            k = np.arange(m[R])
            a = np.searchsorted(rel[1:], k, side="right")
            newlog[R * P + k] = Hl[a] + (k - rel[a])
            for d in range(W):
                a0, a1 = pairs[R, d]
                if a1 <= a0:
                    continue
                assert d != R
                _, lo, up, klo = balanced_record_ranges(rel, Hl, alive, a0, a1, P, eb[R], db[d], dd[d], m[d])
                assert (up > lo).all()  # no record without a child at the destination
                for a_, b_, k_ in zip(lo, up, klo):
                    assert m[d] <= a_ < b_ <= P and (newlog[d * P + a_:d * P + b_] == -1).all()
                    newlog[d * P + a_:d * P + b_] = k_ + np.arange(b_ - a_)
                    moved += b_ - a_
        assert np.array_equal(np.sort(newlog), np.arange(Pg)), (trial, W, P)
        assert moved == e.sum()
        for d in range(W):  # arrival order = slot order
            los = []
            for s_ in range(W):
                a0, a1 = pairs[s_, d]
                if a1 > a0:
                    rel = cq[s_ * P:(s_ + 1) * P + 1] - cq[s_ * P]
                    alive = np.nonzero(np.diff(rel) > 0)[0]
                    los += list(balanced_record_ranges(rel, H[logical[s_ * P:(s_ + 1) * P]], alive, a0, a1, P, eb[s_], db[d], dd[d], m[d])[1])
            assert los == sorted(los)
