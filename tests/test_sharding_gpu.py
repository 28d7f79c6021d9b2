"""The point-sharded engine END TO END with several ranks on the ONE GPU of the test box: every rank is a process with
its own engine (rank r of world N: its shard of the points in HBM, the block pattern of the whole scene), the per-trial
exchanges go through the engine's host-callback transport (mi_ba_comm_set_host_allreduce) and gloo instead of RCCL --
the RCCL calls themselves are covered by test_engine_gpu.py::test_rccl_exchange_on_a_one_rank_communicator.  What this
checks is everything else about N > 1: shard plans, partial sums, replicated PCG and LM decisions staying in step, the
gather of the points at the end."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, outdir, with_long_tracks):
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from conftest import pkg as _pkg
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, capi = _pkg("scene"), _pkg("capi")
    if with_long_tracks:
        from helpers import ladybug_like_scene
        sc = ladybug_like_scene(m=120, n=3000, dmax=90, seed=9)
    else:
        sc = scene.make_scene("x", m=40, n=6000, d=6, seed=77)
    eng = capi.Engine(capi.default_config(rank=rank, world=world, device=0, pcg_tol=1e-12, pcg_maxit=400, mcon=1))

    def allreduce(buf, op):
        t = torch.from_numpy(buf)                       # shares memory with the engine's staging buffer
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
    eng.set_host_allreduce(allreduce)
    eng.load_scene(sc)
    eng.prepare()
    ret, info = eng.run()
    p = eng.packed_params()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), ret=ret, info=info, p=p, trace=eng.trace())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,long_tracks", [(2, False), (3, True)])
def test_ranks_sharing_one_gpu_reproduce_the_single_rank_solve(tmp_path, world, long_tracks):
    import torch.multiprocessing as mp
    from conftest import pkg
    port = 29600 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), long_tracks), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # every rank took the same decisions and holds the same answer, bit for bit
    for r in res[1:]:
        assert int(r["ret"]) == int(res[0]["ret"]) and np.array_equal(r["info"], res[0]["info"])
        assert np.array_equal(r["p"], res[0]["p"]) and np.array_equal(r["trace"], res[0]["trace"])
    # and it is the single-rank answer (the partial sums are added in another order: rounding-level differences)
    scene, capi = pkg("scene"), pkg("capi")
    if long_tracks:
        from helpers import ladybug_like_scene
        sc = ladybug_like_scene(m=120, n=3000, dmax=90, seed=9)
    else:
        sc = scene.make_scene("x", m=40, n=6000, d=6, seed=77)
    eng = capi.Engine(capi.default_config(pcg_tol=1e-12, pcg_maxit=400, mcon=1))
    eng.load_scene(sc); eng.prepare()
    ret, info = eng.run()
    assert abs(int(res[0]["ret"]) - ret) <= 1 and res[0]["info"][6] == info[6]
    assert abs(res[0]["info"][1] - info[1]) < 1e-9 * info[1]
    assert np.abs(res[0]["p"] - eng.packed_params()).max() < 1e-6
