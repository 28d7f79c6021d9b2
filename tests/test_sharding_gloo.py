"""The N > 1 path on CPU: two processes (torch.distributed, gloo) each take the point shard the
engine's planner assigns to their rank, build the shard's partial reduced camera system with the CPU
oracle, all-reduce (SUM) it -- the exchange libmi_ba.so does over RCCL with the same layout
[S | ea/E | cost] -- and must obtain the single-process system of the whole scene."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    from conftest import pkg as _pkg
    from oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, capi = _pkg("scene"), _pkg("capi")
    sc = scene.make_scene("x", m=14, n=900, d=5, seed=21)
    pl = capi.Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam, rank, world)
    st = pl.stats()
    mine = np.sort(pl.get("perm")[st["pt_begin"]:st["pt_end"]])
    # sub-scene: all cameras, only this rank's points
    obs = np.concatenate([np.arange(sc.pt_ptr[i], sc.pt_ptr[i + 1]) for i in mine])
    m, cnp = sc.m, sc.cnp
    sub = scene.Scene(n=len(mine), m=m, cnp=cnp, est_focal=1, undistort=0, explicit_centers=1, proj_sign=-1.0,
                      pt_ptr=np.concatenate([[0], np.cumsum(np.diff(sc.pt_ptr)[mine])]).astype(np.int32),
                      obs_cam=sc.obs_cam[obs], x=sc.x[obs], R0=sc.R0, f_fixed=sc.f_fixed,
                      p0=np.concatenate([sc.p0[:m * cnp], sc.p0[m * cnp:].reshape(-1, 3)[mine].reshape(-1)]), p_true=sc.p0)
    lin = Oracle().linearize(sub, fd_jacobian=0)
    buf = torch.from_numpy(np.concatenate([lin["S"].reshape(-1), lin["ea"].reshape(-1), [lin["cost"], float(len(mine))]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    mx = torch.tensor([np.abs(lin["eb"]).max()], dtype=torch.float64)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if rank == 0:
        full = Oracle().linearize(sc, fd_jacobian=0)
        N = m * cnp
        S = buf[:N * N].numpy().reshape(N, N)
        ok = (np.abs(S - full["S"]).max() < 1e-9 * np.abs(full["U"]).max()
              and np.allclose(buf[N * N:N * N + N].numpy(), full["ea"].reshape(-1), rtol=1e-10)
              and abs(buf[-2].item() - full["cost"]) < 1e-10 * full["cost"] and int(buf[-1].item()) == sc.n
              and abs(mx.item() - np.abs(full["eb"]).max()) < 1e-12)
        open(out, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_point_sharded_system_equals_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
