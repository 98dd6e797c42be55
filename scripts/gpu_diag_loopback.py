"""Diagnostic: the one-rank loopback of the split exchange against the plain filter -- which slots differ after which step."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from parakeet_slam_amd import _lib
from parakeet_slam_amd.sharded import ShardedFilter, TorchComm
import test_gpu_sharded as T

comm = TorchComm()
Ls = 600
ms, cs, scs = T._fast_scenario(Ls, 4)
for nf, nb, Pl in ((200, 130, 1500), (0, 257, 1500), (300, 0, 1500)):
    for nsteps in (2, 3):
        sl = ShardedFilter(Pl, Ls, device=0, comm=comm, split_step=True, loopback=(nf, nb))
        sl.upload_map(ms, cs.reshape(Ls, 25))
        pl = _lib.DeviceFilter(Pl, Ls)
        pl.upload_map(ms, cs.reshape(Ls, 25))
        for st in range(nsteps):
            sl.step(T._V, T._W, 0.1, scs[st], 0.37 + 0.2 * st, seed=4, draw=st, domain=1)
            pl.step(T._V, T._W, 0.1, scs[st], 0.37 + 0.2 * st, seed=4, draw=st, domain=1)
        fa, fb = sl.f.observe_flags(), pl.observe_flags()
        print("   flagged in the last observe: loopback", int((fa != 0).sum()), np.flatnonzero(fa != 0)[:12], "plain", int((fb != 0).sum()), np.flatnonzero(fb != 0)[:12],
              "logw max abs diff", float(np.abs(sl.f.download_log_weights() - pl.download_log_weights()).max()))
        a, b = sl.download_poses(), pl.download_poses()
        bad = np.flatnonzero((a != b).any(axis=1))
        ma, mb = sl.download_landmarks()[0], pl.download_landmarks()[0]
        badm = np.flatnonzero((ma != mb).any(axis=(1, 2)))
        print("loopback", (nf, nb), "steps", nsteps, "split steps", sl.split_steps_done, "records", sl.loopback_records, "pose rows differing", bad.size,
              (bad.min(), bad.max()) if bad.size else "", "cols", np.flatnonzero((a != b).any(axis=0)), "map rows differing", badm.size,
              (badm.min(), badm.max()) if badm.size else "")
        if bad.size:
            k = bad[0]
            print("   first differing slot", k, a[k], b[k], "last", bad[-1], a[bad[-1]], b[bad[-1]])
            print("   differing slots:", bad[:40])
        sl.close()
        pl.close()
dist.destroy_process_group()
