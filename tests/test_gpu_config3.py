"""GPU: BASELINE.json configs[3] at FULL size on one MI355X -- 400 000 particles x 2 000 landmarks (186 GB of map state).
First one filter holding all the particles; then the configuration as BASELINE states it, four shards of 100 000
particles -- four processes on the one device, gloo between them, the exchange overlapped with the next step (split
step) -- which must reproduce the one filter: ancestors and poses exactly (100 000 is not a multiple of the 1 024-particle
scan block: the global-scan plan), the log-weights to rounding (each shard builds its candidate lists around its own mean
pose, so the fall-back kernels see different particles and add the same terms in another order), the maps of sampled
particles bit for bit.  (prkt_core_v2.py:210-252 is the one place where particles meet.)"""
import numpy as np
import pytest
import torch.multiprocessing as mp

from sharded_common import init_gloo, store_file

pytestmark = pytest.mark.gpu

P_TOTAL, WORLD, L, STEPS = 400000, 4, 2000, 3
V, W = 0.2, 0.1
US = (0.37, 0.81, 0.14)
SAMPLE = np.array([0, 1, 99999, 100000, 123456, 199999, 200000, 287654, 300000, 399999])


def scenario():
    from oracle.fastslam_oracle import synthetic_scan, synthetic_world, truth_step

    means, covs = synthetic_world(L)
    pose, scans = (0.0, 0.0, 0.0), []
    for _ in range(STEPS):
        pose = truth_step(pose, V, W, 0.1)
        scans.append(synthetic_scan(means, pose))
    return means, covs, scans, pose


def _one_filter(q):
    try:
        from parakeet_slam_amd import _lib

        means, covs, scans, pose = scenario()
        f = _lib.DeviceFilter(P_TOTAL, L)
        assert f.device_bytes() > 180e9
        f.upload_map(means, covs.reshape(L, 25))
        anc, flagged = [], []
        for s in range(STEPS):
            f.reset_weights()
            f.motion(V, W, 0.1, seed=9, draw=s)
            f.observe(scans[s])
            flagged.append(f.observe_flagged()[0])
            anc.append(f.resample(US[s], domain=_lib.PK_WEIGHTS_LOG, return_ancestors=True))
        poses = f.download_poses()
        maps = [f.download_landmarks(int(p), int(p) + 1) for p in SAMPLE]
        out = dict(anc=anc, poses=poses, maps=maps, summary=f.summary(), route=f.observe_route(), published=f.observe_published(),
                   flagged=flagged, truth=pose)
        f.close()
        q.put(("ok", out))
    except Exception:  # pragma: no cover
        import traceback

        q.put(("ERR", traceback.format_exc()))


def _shard(rank, store, q):
    try:
        init_gloo(rank, WORLD, store)
        from parakeet_slam_amd import _lib
        from parakeet_slam_amd.sharded import ShardedFilter, TorchComm

        means, covs, scans, _ = scenario()
        P_local = P_TOTAL // WORLD
        sf = ShardedFilter(P_local, L, device=0, comm=TorchComm(), split_step=True)
        sf.upload_map(means, covs.reshape(L, 25))
        for s in range(STEPS):
            sf.step(V, W, 0.1, scans[s], US[s], seed=9, draw=s, domain=_lib.PK_WEIGHTS_LOG)
        poses = sf.download_poses()  # (completes the last resample's exchange)
        # balanced placement: every slot carries its logical index (its index in the one filter); the sampled particles are
        # wherever the exchange left them
        logical = sf.logical_index()
        assert sf.placement == "balanced" and sf.f.shard_balanced_errors() == 0
        where = {int(l): j for j, l in enumerate(logical)}
        maps = {int(p): sf.download_landmarks(where[int(p)], where[int(p)] + 1) for p in SAMPLE if int(p) in where}
        q.put((rank, dict(poses=poses, logical=logical, maps=maps, summary=sf.summary(), split=sf.split_steps_done,
                          migrated=sf.total_migrated, bytes_per_particle=sf.f.particle_bytes())))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + traceback.format_exc()))


def test_config3_one_filter_then_four_shards_on_one_gpu():
    ctx = mp.get_context("spawn")
    # ---- 1. one filter, 400 000 x 2 000 (its own process: all of its 186 GB must be gone before the shards start)
    q = ctx.Queue()
    p = ctx.Process(target=_one_filter, args=(q,))
    p.start()
    status, one = q.get(timeout=1500)
    p.join(timeout=120)
    assert status == "ok", one
    assert one["route"] == "ml_regs" and one["published"]
    poses = one["poses"]
    assert np.isfinite(poses).all()
    for a in one["anc"]:
        assert np.all(np.diff(a) >= 0) and a[0] >= 0 and a[-1] < P_TOTAL
    assert abs(one["summary"][0] - one["truth"][0]) < 0.05 and abs(one["summary"][1] - one["truth"][1]) < 0.05
    # ---- 2. four shards of 100 000 on the same device
    q = ctx.Queue()
    store = store_file()
    procs = [ctx.Process(target=_shard, args=(r, store, q)) for r in range(WORLD)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(WORLD):
        r, res = q.get(timeout=1500)
        assert not isinstance(res, str), res
        got[r] = res
    for pr in procs:
        pr.join(timeout=120)
    logical = np.concatenate([got[r]["logical"] for r in range(WORLD)])
    assert np.array_equal(np.sort(logical), np.arange(P_TOTAL))
    sharded = np.empty((P_TOTAL, 4))
    sharded[logical] = np.concatenate([got[r]["poses"] for r in range(WORLD)])
    # the poses after the last resample are the ancestors' poses: equal poses = equal ancestors at every step (the noise of
    # step s + 1 is drawn per global particle index on the poses step s left)
    assert np.array_equal(sharded[:, :3], poses[:, :3])
    assert np.allclose(np.log(sharded[:, 3]), np.log(poses[:, 3]), rtol=1e-9, atol=1e-9)
    k = 0
    index_of = {int(p): i for i, p in enumerate(SAMPLE)}
    for r in range(WORLD):
        assert got[r]["split"] == STEPS - 1  # every step after the first overlapped its exchange
        for p, m in got[r]["maps"].items():
            ref = one["maps"][index_of[p]]
            k += 1
            for x, y in zip(m, ref):
                assert np.array_equal(x, y)
        assert np.allclose(got[r]["summary"], one["summary"], rtol=1e-12, atol=1e-13)
    assert k == len(SAMPLE)
    moved = sum(got[r]["migrated"] for r in range(WORLD))
    assert moved > 0
    print("configs[3] on one GPU: %d particles changed rank over %d resamples = %.3g GB per step over all ranks" %
          (moved, STEPS, moved / STEPS * got[0]["bytes_per_particle"] / 1e9))
