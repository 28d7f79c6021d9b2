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


def _case(kind):
    """scene, config keywords and the extras applied to an engine before prepare()"""
    from conftest import pkg as _pkg
    scene = _pkg("scene")
    off = int(os.environ.get("MI_FUZZ_SEED_OFFSET", "0"))    # other scenes: set the offset
    if kind == "long":
        from helpers import ladybug_like_scene
        return ladybug_like_scene(m=120, n=3000, dmax=90, seed=9 + off), {}, lambda eng: None
    if kind == "det":      # deterministic mode (ordered hand-over of the register tiles; the ragged tracks take the row path), on both ranks
        from helpers import ladybug_like_scene
        return ladybug_like_scene(m=100, n=2500, dmax=80, seed=12 + off), dict(deterministic=1), lambda eng: None
    if kind == "pba":      # PBA's Levenberg-Marquardt rules (lm_variant 1): damping on the diagonals, dL and the step norm in the metric of diag J'J
        return scene.make_scene("x", m=40, n=6000, d=6, seed=79 + off), dict(lm_variant=1, pcg_minit=10, opts=(1e-3, 0.0, 1e-6, 1e-10, 1e5, 0.0), itmax=40), lambda eng: None
    if kind == "resume":   # every trial's first run of CG iterations falls short (engine.cpp solve_phase): the update phase -- and with
        os.environ["MI_BA_PCG_FIRST_RUN"] = "2"   # it the second exchange and its per-rank maxima slots -- runs twice per trial
        return scene.make_scene("x", m=40, n=6000, d=6, seed=77 + off), {}, lambda eng: None
    if kind == "extras":   # radial coefficient, links between cameras, soft constraints on cameras and points: the camera-side
        sc = scene.make_scene("x", m=40, n=6000, d=6, seed=78 + off, undistort=1, nk=1)   # terms are replicated, the point-side sharded
        rng = np.random.default_rng(4)
        la = np.arange(0, 12); lb = (la + 1) % 12
        w = np.zeros((12, sc.cnp)); w[:, 6] = 2e6
        cam_w = np.zeros((sc.m, sc.cnp)); cam_w[5:9, :3] = 50.0
        cam_v = sc.p0[:sc.m * sc.cnp].reshape(sc.m, sc.cnp).copy()
        pt_w = np.where(rng.random(sc.n) < 0.05, 1e-3, 0.0)
        pt_v = sc.p0[sc.m * sc.cnp:].reshape(sc.n, 3) + rng.normal(0, 0.01, (sc.n, 3))

        def extras(eng):
            eng.set_camera_links(la, lb, w)
            eng.set_constraints(cam_w, cam_v, pt_w, pt_v)
        return sc, dict(undistort=1), extras
    return scene.make_scene("x", m=40, n=6000, d=6, seed=77 + off), {}, lambda eng: None


def _worker(rank, world, port, outdir, kind):
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from conftest import pkg as _pkg
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    capi = _pkg("capi")
    sc, kw, extras = _case(kind)
    eng = capi.Engine(capi.default_config(rank=rank, world=world, device=0, pcg_tol=1e-12, pcg_maxit=400, mcon=1, **kw))

    def allreduce(buf, op):
        t = torch.from_numpy(buf)                       # shares memory with the engine's staging buffer
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
    eng.set_host_allreduce(allreduce)
    eng.load_scene(sc)
    extras(eng)
    eng.prepare()
    ret, info = eng.run()
    p = eng.packed_params()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), ret=ret, info=info, p=p, trace=eng.trace())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "plain"), (3, "long"), (2, "extras"), (2, "resume"), (2, "det"), (2, "pba")])
def test_ranks_sharing_one_gpu_reproduce_the_single_rank_solve(tmp_path, world, kind):
    import torch.multiprocessing as mp
    from conftest import pkg
    port = 29600 + (os.getpid() % 2000) + world + 7 * len(kind)
    mp.spawn(_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # every rank took the same decisions and holds the same answer, bit for bit
    for r in res[1:]:
        assert int(r["ret"]) == int(res[0]["ret"]) and np.array_equal(r["info"], res[0]["info"])
        assert np.array_equal(r["p"], res[0]["p"]) and np.array_equal(r["trace"], res[0]["trace"])
    # and it is the single-rank answer (the partial sums are added in another order: rounding-level differences)
    capi = pkg("capi")
    sc, kw, extras = _case(kind)
    os.environ.pop("MI_BA_PCG_FIRST_RUN", None)          # the single-rank solve it is compared with runs as usual
    eng = capi.Engine(capi.default_config(pcg_tol=1e-12, pcg_maxit=400, mcon=1, **kw))
    eng.load_scene(sc); extras(eng); eng.prepare()
    ret, info = eng.run()
    assert abs(int(res[0]["ret"]) - ret) <= 1 and res[0]["info"][6] == info[6]
    assert abs(res[0]["info"][1] - info[1]) < 1e-9 * info[1]
    assert np.abs(res[0]["p"] - eng.packed_params()).max() < 1e-6


def _worker_rccl(rank, world, outdir):
    """one rank per GPU, the exchange through RCCL itself (ncclAllReduce on the engine's stream); the unique id travels through a file"""
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import time
    from conftest import pkg as _pkg
    capi = _pkg("capi")
    sc, kw, extras = _case("long")
    idfile = os.path.join(outdir, "uid.bin")
    if rank == 0:
        uid = capi.comm_unique_id()
        with open(idfile + ".tmp", "wb") as fo:
            fo.write(bytes(uid))
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
        uid = open(idfile, "rb").read()
    eng = capi.Engine(capi.default_config(rank=rank, world=world, device=rank, pcg_tol=1e-12, pcg_maxit=400, mcon=1, **kw))
    eng.comm_init(bytes(uid))
    eng.load_scene(sc)
    eng.prepare()
    ret, info = eng.run()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), ret=ret, info=info, p=eng.packed_params(), trace=eng.trace())
    eng.close()


def test_two_ranks_on_two_gpus_through_rccl(tmp_path):
    """the real transport: ncclAllReduce between two GPUs (skipped on a one-GPU box -- the driver's 8-GPU node runs it)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: the RCCL exchange between ranks (one-rank communicator: test_engine_gpu.py)")
    import torch.multiprocessing as mp
    from conftest import pkg
    mp.spawn(_worker_rccl, args=(2, str(tmp_path)), nprocs=2, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    assert int(res[1]["ret"]) == int(res[0]["ret"]) and np.array_equal(res[1]["info"], res[0]["info"])
    assert np.array_equal(res[1]["p"], res[0]["p"]) and np.array_equal(res[1]["trace"], res[0]["trace"])
    capi = pkg("capi")
    sc, kw, extras = _case("long")
    eng = capi.Engine(capi.default_config(pcg_tol=1e-12, pcg_maxit=400, mcon=1, **kw))
    eng.load_scene(sc); eng.prepare()
    ret, info = eng.run()
    assert abs(int(res[0]["ret"]) - ret) <= 1 and abs(res[0]["info"][1] - info[1]) < 1e-9 * info[1]
    assert np.abs(res[0]["p"] - eng.packed_params()).max() < 1e-6
    eng.close()
    # ... and bench.py itself as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU, RCCL): the first multi-GPU
    # scaling run must not be the first execution of that path.  The line carries what RCCL reported about every rank's communicator.
    import json
    import subprocess
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--config", "C2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "rehearsal" not in d and d["rccl"]["transport"] == "rccl"
    assert sorted(r["rank"] for r in d["rccl"]["ranks"]) == [0, 1] and all(r["nranks"] == 2 for r in d["rccl"]["ranks"])
    assert len({r["device"] for r in d["rccl"]["ranks"]}) == 2
    assert d["kernel_ms_per_step"]["exchange"] > 0 and d["rmse_px"]["after_timed_steps"] < 0.5 * d["rmse_px"]["initial"]


def test_bench_with_two_ranks_rehearsed_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), rehearsed here with both ranks on
    the one GPU of the box (MI_BA_BENCH_ONE_GPU: gloo + the engine's host transport instead of RCCL): the sharded engine, the
    barrier / max-over-ranks timing and the one JSON line of rank 0"""
    import json
    import subprocess
    env = dict(os.environ, MI_BA_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--config", "C1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                              # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0 and "rehearsal" in d and d["config"]["parallelism"] == "points/2"
    assert d["rmse_px"]["after_timed_steps"] < 0.5 * d["rmse_px"]["initial"] and d["kernel_ms_per_step"]["allreduce_host"] > 0
    assert d["kernel_ms_per_step"]["exchange"] > 0 and d["kernel_ms_per_step"]["replicated"] > 0      # what a scaling run has to explain itself with
    assert d["rccl"]["transport"].startswith("host") and sorted(r["rank"] for r in d["rccl"]["ranks"]) == [0, 1]    # every rank reported in
