#!/usr/bin/env python3
"""Headline benchmark: FastSLAM filter steps on synthetic 360-degree bearing+colour scans.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P] [--landmarks L]
                    [--assoc ml|known] [--no-cpu-baseline]

One "step" = one whole cam_cb (prkt_core_v2.py:59-137): weight reset, motion sample,
maximum-likelihood data association, per particle x landmark EKF update + weight,
systematic resample -- all on the GPU through the C ABI (include/parakeet_slam.h).
The default workload is BASELINE.json configs[1]: 10 000 particles x 500 landmarks per
GPU, B = L blobs per scan, float64 like the reference.  N > 1: one process per GPU
(launched by torch.distributed.run), particles sharded, weak scaling.

Prints ONE JSON line on rank 0 (contract in the task description), with two extra
objects: "roofline" for the dominant HBM kernel (the fused EKF+weight kernel k_observe,
timed with hipEvents on its own stream inside the timed region) and "cpu_baseline"
(the NumPy oracle of the same step on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def measured_traffic(P, L, variant):
    """HBM bytes per k_observe launch from the committed rocprofv3 PMC run (profiles/*/pmc_traffic.json,
    made by scripts/gpu_pmc_traffic.sh) when it was taken on this very configuration, else None."""
    best = None
    for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        path = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
        if os.path.exists(path):
            d = json.load(open(path))
            if d["config"]["particles"] == P and d["config"]["landmarks"] == L and variant in d["bytes_per_launch"]:
                best = d["bytes_per_launch"][variant]
    return best


ROUTE_KERNEL = {
    "known_ids": "k_observe<known ids> (fused EKF update + log-weight)",
    "ml_fused": "k_step_fused (association gates + settling of contested blobs + EKF update + log-weight in ONE kernel)",
    "ml_handoff": "k_observe_fast (EKF update + log-weight + settling of contested associations; includes the "
                  "near-empty general k_observe launch for flagged particles; the association kernel is separate)",
    "ml_sweep": "k_observe_sweep (two sweeps over landmark chunks: settling of contested associations, EKF update + "
                "log-weight; the association kernel is separate)",
    "ml_general": "k_observe<ML general> (fused EKF update + log-weight)",
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_UPDATE = 224  # 14 fp64 read + 14 written per particle.landmark


def synthetic_controls(steps, w=0.1, dt=0.1):
    """Angular velocity per step.  The reference evaluates |atan2(fy - sy, fx - sx) - blob.bearing|
    > pi/2 with the robot-frame bearing (prkt_core_v2.py:473-475, frames mixed): once the robot's
    heading passes pi/2 EVERY blob gets probability 0 and no EKF update happens any more -- a
    degenerate workload (measured: the step gets 25 % faster).  So SURVEY 8d's control
    (v, w) = (0.2, 0.1) is kept for the first 60 steps of 0.1 s and the robot then turns back
    and forth between -0.6 and +0.6 rad."""
    out, h, sign = [], 0.0, 1.0
    for _ in range(steps):
        if abs(h + sign * w * dt) > 0.6:
            sign = -sign
        out.append(sign * w)
        h += sign * w * dt
    return out


def synthetic_inputs(L, steps, v=0.2, w=0.1, dt=0.1):
    """World + noise-free scans along the true trajectory.  Restated here so the
    timed path does not import the oracle."""
    rs = np.random.RandomState(123)
    phi = -math.pi + 2.0 * math.pi * np.arange(L) / float(L) + 0.01
    rho = rs.uniform(8.0, 30.0, size=L)
    col = rs.uniform(0.0, 255.0, size=(L, 3))
    if os.environ.get("PK_BENCH_LATTICE"):  # experiment: colours on a lattice, no blob passes two landmarks' gates
        n = int(math.ceil(L ** (1.0 / 3.0)))
        idx = np.arange(L)
        col = np.stack([idx % n, (idx // n) % n, idx // (n * n)], axis=1) * (255.0 / max(n - 1, 1))
    means = np.empty((L, 5))
    means[:, 0] = rho * np.cos(phi)
    means[:, 1] = rho * np.sin(phi)
    means[:, 2:] = col
    covs = np.broadcast_to(0.25 * np.identity(5), (L, 5, 5)).copy()
    x = y = h = 0.0
    scans = []
    for ws in synthetic_controls(steps, w, dt):
        h1 = h + ws * dt / 2
        x, y = x + v * dt * math.cos(h1), y + v * dt * math.sin(h1)
        h = math.atan2(math.sin(h1 + ws * dt / 2), math.cos(h1 + ws * dt / 2))
        blobs = np.empty((L, 4))
        blobs[:, 0] = np.arctan2(means[:, 1] - y, means[:, 0] - x) - h
        blobs[:, 1:] = col
        scans.append(blobs)
    return means, covs, scans


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def _cpu_run(job):
    """One oracle filter of Ps particles stepped nsteps times; returns its wall time."""
    from oracle.fastslam_oracle import OracleFilter

    try:  # one thread per process, whatever BLAS/OpenMP pools NumPy was built with
        from threadpoolctl import threadpool_limits

        threadpool_limits(1)
    except Exception:
        pass
    L, Ps, nsteps, seed = job
    means, covs, scans = synthetic_inputs(L, nsteps)
    rs = np.random.RandomState(seed)
    rnd = random.Random(seed)
    f = OracleFilter(Ps, means, covs)
    t0 = time.perf_counter()
    for s in range(nsteps):
        f.step(0.2, 0.1, 0.1, rs.standard_normal((Ps, 3)), scans[s], rnd.random())
    return time.perf_counter() - t0


def cpu_baseline(L, budget_s=20.0):
    """NumPy oracle ("port" of the reference step, validated against the reference in tests/)
    timed on a bounded particle sample of the same workload: first on ONE host thread, then on
    every host core with the sample's particles split over a fork()ed process pool (particles
    are independent until the weight sum).  Must run BEFORE the GPU is initialised
    (fork).  `value` is the all-core figure; the 1-thread figure rides along."""
    import multiprocessing as mp

    cores = min(usable_cores(), 64)
    nsteps = 2
    half = budget_s / 2.0
    Ps = 4
    t1 = _cpu_run((L, Ps, nsteps, 7))
    while t1 < half / 1.5 and Ps < 4096:  # the oracle vectorises over particles: grow until the sample fills the budget
        Ps = int(min(4096, max(Ps + 1, Ps * min(8.0, 0.8 * half / t1))))
        t1 = _cpu_run((L, Ps, nsteps, 7))
    single = Ps * L * nsteps / t1
    out = {
        "value": single,
        "unit": "particle*landmark EKF updates/s",
        "cores": 1,
        "kind": "port",
        "steps_per_s": nsteps / t1,
        "single_thread_value": single,
        "sample": "%d particles x %d landmarks x %d blobs, %d full steps (ML association), NumPy oracle, "
        "%.1f s on 1 of %d host cores" % (Ps, L, L, nsteps, t1, os.cpu_count() or 1),
    }
    if cores > 1:
        try:
            with mp.get_context("fork").Pool(cores) as pool:
                # page the workers in; a pool that is much slower than one thread alone (cores
                # fewer than reported) times out and leaves the 1-thread figure
                pool.map_async(_cpu_run, [(L, 2, 1, 7)] * cores).get(timeout=60)
                t0 = time.perf_counter()
                pool.map_async(_cpu_run, [(L, Ps, nsteps, 7 + i) for i in range(cores)],
                               chunksize=1).get(timeout=max(30.0, 6.0 * t1))
                tw = time.perf_counter() - t0
            out.update({
                "value": cores * Ps * L * nsteps / tw,
                "cores": cores,
                "steps_per_s": nsteps / tw,
                "sample": "%d particles (%d per process x %d processes) x %d landmarks x %d blobs, %d full steps "
                "(ML association), NumPy oracle, %.1f s wall on all %d host cores; one thread alone: %.3g updates/s "
                "(%d particles, %.1f s)" % (cores * Ps, Ps, cores, L, L, nsteps, tw, cores, single, Ps, t1),
            })
        except Exception as e:  # a pool that cannot start leaves the 1-thread figure
            out["pool_error"] = repr(e)
    return out


def main():
    # Exactly ONE line may reach stdout (the JSON).  RCCL and friends print banners to fd 1,
    # so park the real stdout and point fd 1 at stderr until the result is ready.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--particles", type=int, default=10000, help="particles per GPU")
    ap.add_argument("--landmarks", type=int, default=500)
    ap.add_argument("--assoc", choices=["ml", "known"], default="ml")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (collectives included) even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch

    if torch.cuda.device_count() == 0:  # does not initialise the GPU
        raise SystemExit("bench.py needs a GPU: the particle update has no CPU fallback")
    # CPU baseline first: it fork()s a process pool, which must happen before any HIP call.
    # Rank 0 at N=1 only (the other ranks would just wait on it).
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.landmarks, args.cpu_budget)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the particle update has no CPU fallback")
    if os.environ.get("PK_BENCH_SAME_GPU"):  # rehearsal of the N > 1 control flow on a one-GPU box (with PK_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from parakeet_slam_amd import _lib

    if os.environ.get("PK_BENCH_LIB"):  # tuning experiments only: A/B of two builds on one box
        _lib.LIB_PATH = os.path.join(ROOT, "parakeet_slam_amd", os.environ["PK_BENCH_LIB"])
    P, L = args.particles, args.landmarks
    K, W = args.steps, args.warmup
    EXTRA = 32  # untimed steps after the timed region (association share, supplied-ids route)
    means, covs, scans = synthetic_inputs(L, K + W + EXTRA)
    ws = synthetic_controls(K + W + EXTRA)
    ids = np.arange(1, L + 1, dtype=np.int32) if args.assoc == "known" else None

    if world > 1 or args.force_sharded:
        import torch.distributed as dist

        if not dist.is_initialized():  # --force-sharded with a single rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            dist.init_process_group("nccl", rank=0, world_size=1)
        from parakeet_slam_amd.sharded import ShardedFilter

        filt = ShardedFilter(P, L, device=local_rank)
    else:
        filt = _lib.DeviceFilter(P, L, device=local_rank)
    filt.upload_map(means, covs.reshape(L, 25))
    if os.environ.get("PK_OBSERVE_NV"):  # tuning experiments only
        filt.set_option("observe_landmarks_per_lane", int(os.environ["PK_OBSERVE_NV"]))
    rnd = random.Random(7)
    us = [rnd.random() for _ in range(K + W + EXTRA)]

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()
        filt.synchronize()

    def one_step(s):
        filt.step(0.2, ws[s], 0.1, scans[s], us[s], seed=7, draw=s, ids=ids, domain=_lib.PK_WEIGHTS_LOG)

    for s in range(W):
        one_step(s)
    barrier()
    # hipEvents around the dominant kernel only (the observe launch), on the library's own stream,
    # inside the timed region; every bracketed launch costs two event records
    # (two event records per bracketed launch cost the stream ~8 us: every 4th step is sampled)
    stride = 4 if K >= 16 else 1
    filt.set_option("timing_stride", stride)
    filt.enable_timing(0b0000100)
    filt.reset_timings()
    barrier()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        one_step(s)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    route = filt.observe_route() if hasattr(filt, "observe_route") else ("known_ids" if args.assoc == "known" else "ml")
    tm = filt.timings()
    filt.enable_timing(0)
    filt.set_option("timing_stride", 1)
    summary = filt.summary()
    # validity probe (untimed): share of the blobs of the last timed scan that the particles, as they
    # stand now, still associate with some landmark (the workload degenerates when this collapses)
    matched = None
    if world == 1 and not args.force_sharded and float(P) * L <= 2e7 and hasattr(filt, "associate"):
        matched = float((filt.associate(scans[W + K - 1]) > 0).mean())
    # the association kernel's share, from a few extra (untimed) steps
    filt.enable_timing(0b0000010)
    filt.reset_timings()
    for s in range(W + K, W + K + 10):
        one_step(s)
    barrier()
    tm["assoc"] = filt.timings()["assoc"]
    filt.enable_timing(0)

    # the EKF stage on its own (SURVEY 8d defines the HBM roofline on it): the same filter, reset to
    # the initial map and poses, stepped with SUPPLIED ids -- whole steps, resample included, exactly
    # what `--assoc known` times -- so both routes' rooflines come from one run
    known = None
    if args.assoc == "ml" and world == 1 and not args.force_sharded:
        kids = np.arange(1, L + 1, dtype=np.int32)
        kn, kw = (20, 3) if K >= 20 else (K, 2)
        filt.upload_map(means, covs.reshape(L, 25))
        filt.upload_poses(np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (P, 1)))
        for s in range(kw):
            filt.step(0.2, ws[s], 0.1, scans[s], us[s], seed=7, draw=s, ids=kids, domain=_lib.PK_WEIGHTS_LOG)
        filt.enable_timing(0b0000100)
        filt.reset_timings()
        barrier()
        k0 = time.perf_counter()
        for s in range(kw, kw + kn):
            filt.step(0.2, ws[s], 0.1, scans[s], us[s], seed=7, draw=s, ids=kids, domain=_lib.PK_WEIGHTS_LOG)
        barrier()
        k1 = time.perf_counter()
        kms, kcnt = filt.timings()["observe"]
        filt.enable_timing(0)
        kavg = (kms / max(kcnt, 1)) * 1e-3
        known = {
            "what": "the filter reset to its initial state, %d whole steps with ids = 1..L (no association)" % kn,
            "ms_per_step": (k1 - k0) / kn * 1e3,
            "value": float(P) * L * kn / (k1 - k0),
            "kernel": ROUTE_KERNEL["known_ids"],
            "avg_launch_ms": kavg * 1e3,
            "launches": kcnt,
            "achieved": float(P) * L * BYTES_PER_UPDATE / kavg / 1e9 if kavg > 0 else 0.0,
            "frac": float(P) * L * BYTES_PER_UPDATE / kavg / 1e9 / HBM_PEAK_GBS if kavg > 0 else 0.0,
            "unit": "GB/s",
            "traffic": measured_traffic(P, L, "observe_known"),
        }

    # what a plain device-to-device copy reaches on THIS box (read + write bytes / time), outside the
    # timed region: the practical ceiling next to the 8 TB/s vendor figure
    copy_gbs = None
    if rank == 0:
        n = 1 << 27  # 2 x 1 GiB of float64
        a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * 8 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b

    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        total_updates = float(P) * world * L * K
        obs_ms, obs_n = tm["observe"]
        assoc_ms, assoc_n = tm["assoc"]
        obs_avg_s = (obs_ms / max(obs_n, 1)) * 1e-3
        alg_bytes = float(P) * L * BYTES_PER_UPDATE
        achieved = alg_bytes / obs_avg_s / 1e9 if obs_avg_s > 0 else 0.0
        out = {
            "metric": "particle*landmark EKF updates/sec (whole filter step)",
            "value": total_updates / elapsed,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "filter_steps_per_sec": K / elapsed,
            "config": {
                "workload": "BASELINE.json configs[1]: %d particles/GPU x %d landmarks, B=%d blobs/scan, "
                "synthetic 360deg bearing+colour obs, assoc=%s, resample every step" % (P, L, L, args.assoc),
                "particles_per_gpu": P,
                "landmarks": L,
                "blobs": L,
                "assoc": args.assoc,
                "global_particles": P * world,
                "parallelism": "particles sharded over %d GPU(s); RCCL all-reduce(max) + all-gather(block weight "
                "totals) + all-to-all(migrating particles) per resample" % world if (world > 1 or args.force_sharded)
                else "single GPU",
            },
            "roofline": {
                "kernel": ROUTE_KERNEL.get(route, route),
                "route": route,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(P, L, {"known_ids": "observe_known", "ml_fused": "step_fused",
                                                   "ml_handoff": "observe_ml"}.get(route, "none")),
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes = (2*FETCH + WRITE)*1024, "
                "see profiles/*/pmc_traffic.json",
                "copy_measured": copy_gbs,
                "frac_of_copy": achieved / copy_gbs if copy_gbs else None,
                "avg_launch_ms": obs_avg_s * 1e3,
                "launches": obs_n,
                "launches_note": "hipEvent-bracketed launches inside the timed region (every %d-th step of %d)" % (stride, K),
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "kernel_ms_per_step": {"observe": obs_ms / max(obs_n, 1), "assoc": assoc_ms / max(assoc_n, 1)},
            "summary": list(summary),
            "matched_fraction_last_timed_scan": matched,  # validity: the scans stayed matchable to the end
        }
        if known is not None:
            out["ekf_stage_supplied_ids"] = known
        if cpu is not None:
            out["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(0 if os.environ.get("PK_BENCH_SAME_GPU") else int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(os.environ.get("PK_BENCH_BACKEND", "nccl"))  # "nccl" = RCCL over xGMI
    main()
