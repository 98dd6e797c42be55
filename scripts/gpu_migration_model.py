"""What the sharded resample would move at world sizes one box cannot run (VERDICT round 4 #1c: a GPU box allows six
processes on its card, BASELINE configs[4] has eight ranks).

ONE filter holds all W x P_local particles and runs the bench's trajectory; at every resample the device's own ancestors give
the offspring table H, and the exchange of W ranks is PLANNED on the host by the product's own planner
(parakeet_slam_amd.sharded.plan_balanced -- what pk_shard_plan_balanced_dev restates, tests/test_sharded_gloo.py) without being
carried out: records and children that would change rank under
  contiguous          rank r holds the logical slots [r P, (r + 1) P); every particle of a contiguous index range is packed,
                      with or without children (rounds 1-4)
  contiguous_compact  the same placement, only particles with children at the destination packed
  balanced            physical slots carry their logical index, a rank keeps its own children, only its excess travels,
                      only particles with children are packed (round 5, the default)
The sharded filter's results are bit-identical to the one filter's (tests), so these ARE the counts an N-rank run sees; the
4-rank rehearsal on one GPU (bench.py --gpus 4, PK_BENCH_SAME_GPU) is the cross-check.

    python scripts/gpu_migration_model.py P_total L steps [worlds]     -> one JSON line
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (inputs of the bench's own trajectory)
from parakeet_slam_amd import _lib  # noqa: E402
from parakeet_slam_amd.sharded import _ranges, plan_balanced  # noqa: E402


def contiguous_counts(H, W, P):
    """(records of rounds 1-4, records with dead particles left out, children) that cross ranks with rank r = slots [rP, (r+1)P)."""
    Pg = W * P
    lo, up = H[:-1], H[1:]
    rec_all = rec_alive = children = 0
    for s in range(W):
        l, u = lo[s * P:(s + 1) * P], up[s * P:(s + 1) * P]
        for d in range(W):
            if d == s:
                continue
            start, end = d * P, (d + 1) * P
            j0 = int(np.searchsorted(u, start, side="right"))
            j1 = max(int(np.searchsorted(l, end, side="left")), j0)
            rec_all += j1 - j0
            c = np.minimum(u[j0:j1], end) - np.maximum(l[j0:j1], start)
            rec_alive += int((c > 0).sum())
            children += int(np.maximum(c, 0).sum())
    assert 0 <= children <= Pg
    return rec_all, rec_alive, children


def balanced_step(H, logical, W, P):
    """(records, children) that cross ranks, and the new logical index of every physical slot."""
    cq, nz, (n, m, e, dd, ebase, dbase), pairs = plan_balanced(H, logical, W, P)
    records = int((pairs[:, :, 1] - pairs[:, :, 0]).sum())
    cnt = np.diff(cq)
    new = np.empty_like(logical)
    excess, free = [], []
    for s in range(W):
        c = cnt[s * P:(s + 1) * P]
        ch = np.repeat(H[logical[s * P:(s + 1) * P]], c) + _ranges(c)  # this rank's children, by parent then output slot
        new[s * P:s * P + m[s]] = ch[:m[s]]
        excess.append(ch[P:])
        free.append(s * P + np.arange(m[s], P))
    new[np.concatenate(free)] = np.concatenate(excess)
    return records, int(e.sum()), new


def main():
    Pt, L, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    worlds = [int(w) for w in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["2", "4", "8"])]
    means, covs, scans = bench.synthetic_inputs(L, steps + 1)
    ws = bench.synthetic_controls(steps + 1)
    import random

    rnd = random.Random(7)
    us = [rnd.random() for _ in range(steps + 1)]
    f = _lib.DeviceFilter(Pt, L)
    f.upload_map(means, covs.reshape(L, 25))
    stride = f.particle_bytes()
    logical = {w: np.arange(Pt, dtype=np.int64) for w in worlds}
    tot = {w: dict(contiguous=0, contiguous_compact=0, contiguous_children=0, balanced=0, balanced_children=0) for w in worlds}
    skip = 2  # the first resamples of a fresh filter (all particles at one pose) are not the steady state
    for s in range(steps):
        f.reset_weights()
        f.motion(0.2, ws[s], 0.1, seed=7, draw=s)
        f.observe(scans[s])
        anc = f.resample(us[s], domain=_lib.PK_WEIGHTS_LOG, return_ancestors=True)
        H = np.concatenate([[0], np.cumsum(np.bincount(anc, minlength=Pt))]).astype(np.int64)
        for w in worlds:
            P = Pt // w
            ra, rc, ch = contiguous_counts(H, w, P)
            rb, cb, logical[w] = balanced_step(H, logical[w], w, P)
            if s >= skip:
                t = tot[w]
                t["contiguous"] += ra
                t["contiguous_compact"] += rc
                t["contiguous_children"] += ch
                t["balanced"] += rb
                t["balanced_children"] += cb
        sys.stderr.write("step %d done\n" % s)
    n = steps - skip
    out = {"what": "records (pose + whole map, %d bytes each) that change rank per resample, planned on the weights of ONE filter of "
                   "%d x %d on the bench's trajectory, mean of steps %d..%d" % (stride, Pt, L, skip, steps - 1),
           "particles_total": Pt, "landmarks": L, "record_bytes": stride, "git": os.environ.get("PK_GIT_SHA", "unknown"), "worlds": {}}
    for w in worlds:
        t = tot[w]
        P = Pt // w
        out["worlds"][str(w)] = {
            "particles_per_rank": P,
            "records_per_step": {k: t[k] / n for k in ("contiguous", "contiguous_compact", "balanced")},
            "children_per_step": {"contiguous": t["contiguous_children"] / n, "balanced": t["balanced_children"] / n},
            "fraction_of_particles": {k: t[k] / n / Pt for k in ("contiguous", "contiguous_compact", "balanced")},
            "bytes_per_step_all_ranks": {k: t[k] / n * stride for k in ("contiguous", "contiguous_compact", "balanced")},
            "bytes_per_rank_and_step": {k: t[k] / n * stride / w for k in ("contiguous", "contiguous_compact", "balanced")},
            "balanced_over_contiguous": t["balanced"] / max(t["contiguous"], 1),
        }
    f.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
