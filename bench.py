#!/usr/bin/env python
"""bench.py -- LM iterations/sec (+ reprojection RMSE) of the MI355X bundle-adjustment engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE Levenberg-Marquardt damping trial (sba's nlss unit: one reduced camera system
built, solved and gain-ratio tested) over the whole synthetic scene, inputs resident in HBM.
The reference's stop rules are ON: when the solve ends (13 trials on C3) the next one starts from
the same initial guess, restored from a copy kept in HBM (mi_ba_restart), as a pipeline would go
on to its next problem -- so every timed step is a full trial however large --steps is.
Workload = BASELINE.json's metric config, C3: 1k cameras / 1M points / 10M observations
(seeded, large-scale-sfm_amd/scene.py).  --config C4 = the BAL-Ladybug-like scene (camera sets that do not repeat,
tracks up to 120 views: the block-major row path); C1 / C2 = the smaller configs.  With N > 1 the SAME scene is
point-sharded over the ranks (RCCL all-reduce of the Schur partials inside libmi_ba.so), so scaling is "strong".
Rank 0 prints one JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 f64 FMA lanes x 2 flops x 2.4 GHz


def cpu_baseline(scene_mod, gpu_value, config, full_c3=False):
    """The reference's CPU solve timed on this box's host cores, on a bounded sample.  The metric's configuration C3 costs the
    real sba-1.5 about three minutes per LM iteration (O(m n) index scans, finite-difference Jacobian, a 7000 x 7000 dense
    Cholesky), so by default two iterations of C2 (the same generator, 1/10 of the observations) are timed here and scaled to C3
    by the ratio of the reference's own seconds per iteration on the two configurations as measured in the build container
    (tests/golden/ref_C3.npz, ref_C2.npz) -- flagged as extrapolated.  --cpu-baseline-c3 times one real C3 iteration instead."""
    # sba-1.5 itself is single-threaded; the LAPACK it calls (MKL here: dpotrf / dpotrs of the reduced system, dsytrf / dsytri per point)
    # is pinned to ONE thread too, before MKL is loaded, so that "cores" is a fact and not a label
    os.environ["MKL_NUM_THREADS"] = "1"
    os.environ["OMP_NUM_THREADS"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle, Reference
    golden = os.path.join(ROOT, "tests", "golden")
    out = {"unit": "LM iterations/s", "cores": 1, "mkl_num_threads": 1, "host_cores": os.cpu_count()}
    try:                                   # (and through MKL's own call, should the library be initialised already)
        import ctypes
        mkl = ctypes.CDLL("/opt/conda/lib/libmkl_rt.so.1")   # the library oracle/_ref/libref_sfm.so is linked with (oracle/Makefile)
        mkl.MKL_Set_Num_Threads(1)
        out["mkl_num_threads"] = int(mkl.MKL_Get_Max_Threads())
    except Exception:
        pass
    c3 = np.load(os.path.join(golden, "ref_C3.npz")) if os.path.exists(os.path.join(golden, "ref_C3.npz")) else None
    try:
        if Reference.available():
            ref = Reference()
            if full_c3:
                sc = scene_mod.make_scene("C3")
                t0 = time.time()
                r = ref.run(sc, itmax=1)
                dt = time.time() - t0
                out.update({"value": float(r["info"][9] / dt), "kind": "reference", "config": "C3",
                            "sample": "config C3 itself: 1 LM iteration of the real sba-1.5 (FD Jacobian, dense Cholesky via MKL, MKL_NUM_THREADS=1), %.1f s" % dt})
            else:
                sc = scene_mod.make_scene("C2")
                t0 = time.time()
                r = ref.run(sc, itmax=2)
                dt = time.time() - t0
                out.update({"value": float(r["info"][9] / dt), "kind": "reference", "config": "C2",
                            "sample": "config C2 (200 cams / 100k pts / 1M obs, same generator), 2 LM iterations of the real sba-1.5 (FD Jacobian, "
                            "dense Cholesky via MKL with MKL_NUM_THREADS=1: one core in all), %.1f s" % dt,
                            "rmse_after": float(np.sqrt(r["info"][1] / sc.nobs))})
                if c3 is not None:
                    c2 = np.load(os.path.join(golden, "ref_C2.npz"))
                    s3 = float(c3["seconds_per_lm_iteration"])
                    s2 = float(c2["seconds_per_lm_iteration"]) if "seconds_per_lm_iteration" in c2 else 102.0 / 13.0   # make_golden_c2.py's log
                    out["extrapolated_to_C3"] = {
                        "value": out["value"] * s2 / s3, "unit": "LM iterations/s",
                        "law": "C3 it/s = C2 it/s measured here x (s per LM iteration on C2 / on C3 of the real sba-1.5, both measured in the "
                               "build container: %.1f s / %.1f s)" % (s2, s3),
                        "reference_solve_of_C3": {"lm_iterations": int(c3["ret_final"]), "seconds": float(c3["seconds_total"]),
                                                  "threads": int(c3["threads"]), "final_rmse_px": float(np.sqrt(c3["info_final"][1] / 1e7))}}
    except Exception as ex:  # the prebuilt reference may not load on another box
        sys.stderr.write("cpu_baseline: reference unavailable (%s), timing the port\n" % ex)
    if "value" not in out:
        o = Oracle()
        sc = scene_mod.make_scene("C2")
        t0 = time.time()
        r = o.run(sc, itmax=6, fd_jacobian=1)
        dt = time.time() - t0
        out.update({"value": float(r["info"][9] / dt), "kind": "port", "config": "C2",
                    "sample": "config C2, 6 LM iterations of oracle/ba_oracle.c (FD Jacobian, dense Cholesky), %.1f s" % dt,
                    "rmse_after": float(np.sqrt(r["info"][1] / sc.nobs))})
    if config == "C3":
        same = out["value"] if out.get("config") == "C3" else out.get("extrapolated_to_C3", {}).get("value")
        if same:
            out["gpu_over_cpu_on_C3"] = {"ratio": gpu_value / same, "cpu_figure": "measured" if out.get("config") == "C3" else "extrapolated"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-c3", action="store_true", help="time one real sba-1.5 iteration on C3 itself (minutes)")
    ap.add_argument("--deterministic", type=int, default=0, help="0: f64 atomics (default); 1: ordered hand-over, per-tile copies summed in tile order (no f64 atomics, full speed); "
                    "2: every point on the block-major row path")
    ap.add_argument("--shared-focal", type=float, default=0.0, help="weight of a ring of focal-length couplings between the cameras (PBA's SetFocalMask with one group); 0 = every camera its own focal length")
    ap.add_argument("--pcg-tol", type=float, default=1e-3)
    ap.add_argument("--pcg-maxit", type=int, default=100)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # MI_BA_BENCH_ONE_GPU=1: a rehearsal of the N > 1 control flow on a box with ONE GPU -- every rank on device 0, the
    # exchanges through the engine's host transport and gloo instead of RCCL.  Marked in the output; never a measurement.
    rehearsal = world > 1 and os.environ.get("MI_BA_BENCH_ONE_GPU") == "1"
    device = 0 if rehearsal else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    capi = importlib.import_module("large-scale-sfm_amd.capi")
    scene_mod = importlib.import_module("large-scale-sfm_amd.scene")

    t0 = time.time()
    sc = scene_mod.make_scene(args.config)
    t_gen = time.time() - t0
    cfg = capi.default_config(device=device, rank=rank, world=world, pcg_tol=args.pcg_tol, pcg_maxit=args.pcg_maxit,
                              deterministic=args.deterministic, tile_max_batches=int(os.environ.get("MI_BENCH_TILE_MAX_BATCHES", "0")))   # (probe: 0 = the planner's choice)
    eng = capi.Engine(cfg)
    if rehearsal:
        def host_allreduce(buf, op):
            dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
        eng.set_host_allreduce(host_allreduce)
    elif world > 1:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        eng.comm_init(bytes(uid.cpu().numpy().tobytes()))
    # what RCCL itself reports about the communicator every rank joined (ncclCommCount / CuDevice / UserRank inside libmi_ba.so), gathered
    # on rank 0: the line below is refused unless N ranks on N distinct devices were really seen
    comm = eng.comm_info()
    comm["hip_device"] = device
    comm["pid"] = os.getpid()
    comms = [comm]
    if dist is not None:
        comms = [None] * world
        dist.all_gather_object(comms, comm)
        if not rehearsal:
            bad = [c for c in comms if not c["rccl"] or c["nranks"] != world]
            if bad or len({c["device"] for c in comms}) != world or sorted(c["rank"] for c in comms) != list(range(world)):
                raise SystemExit("bench.py: the RCCL communicator does not span %d ranks on %d devices: %s" % (world, world, comms))
    eng.set_timing_detail(1)            # HIP events around the build kernels: the roofline figure is measured in the timed steps
    eng.load_scene(sc)
    if args.shared_focal > 0 and not (cfg.est_focal and sc.cnp >= 7):
        raise SystemExit("--shared-focal couples focal lengths: the scene must estimate them (cnp >= 7)")
    if args.shared_focal > 0:           # BASELINE's C4 is a shared-intrinsics problem: one focal length for all cameras, as PBA's SetFocalMask
        la = np.arange(sc.m, dtype=np.int32); lb = (la + 1) % sc.m      # does it -- a ring of quadratic couplings between the cameras' focal lengths
        w = np.zeros((sc.m, sc.cnp)); w[:, 6] = (np.sqrt(sc.nobs / sc.m) * args.shared_focal) ** 2 / cfg.f_scale ** 2   # (pba_shim.cpp, ProcessWeightCameraQ)
        eng.set_camera_links(la, lb, w)
    t0 = time.time()
    eng.prepare()
    t_plan = time.time() - t0
    eng.begin()
    rmse0 = float(np.sqrt(eng.info()[0] / sc.nobs))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    done = {"solves": 0, "accepted": 0, "rmse": None, "iters": None}

    def one_step():
        st = eng.step()
        done["accepted"] += st == 1
        if st == 2:                                    # a stop rule ended this solve with the trial just made
            info = eng.info()
            done["solves"] += 1; done["rmse"] = float(np.sqrt(info[1] / sc.nobs)); done["iters"] = int(info[5])
            eng.restart()

    for _ in range(args.warmup):
        one_step()
    eng.reset_timing()
    done["accepted"] = 0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tm = eng.timing()
    info = eng.info()
    rmse_after = float(np.sqrt(info[1] / sc.nobs))
    # the per-phase breakdown: the same trials once more, untimed, with HIP events around the reduced solve and the
    # back-substitution as well (an event costs ~5 us of idle GPU, so the timed pass only brackets the build kernels)
    eng.set_timing_detail(2)
    eng.restart()
    for _ in range(args.warmup):
        eng.step() == 2 and eng.restart()
    eng.reset_timing()
    for _ in range(args.steps):
        eng.step() == 2 and eng.restart()
    tm2 = eng.timing()
    eng.set_timing_detail(1)

    if rank == 0:
        # every fourth build of the timed steps is bracketed by HIP events (a pair costs ~12 us of idle GPU); a run too short to
        # hold one takes the figure from the untimed second pass, where every build is
        build_src = "HIP events around every fourth build of the timed steps (%d launches)" % int(tm["build_launches"])
        if tm["build_launches"] > 0:
            build_ms = tm["build_ms"] / tm["build_launches"]
        else:
            build_ms = tm2["build_ms"] / max(tm2["build_launches"], 1)
            build_src = "HIP events around every build of the untimed second pass (the timed steps held no sampled build)"
        # HBM traffic of one k_build launch from the PMC passes committed under profiles/ (rocprofv3 cannot run
        # inside this process); only quoted for the configuration it was measured on
        traffic, tpath = None, None
        if world == 1 and not args.deterministic:
            import glob
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from build_traffic import build_source_sha
            for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_build_traffic_%s.json" % args.config)), reverse=True):
                rec = json.load(open(cand))
                if rec.get("build_source_sha") == build_source_sha():      # measured on these very kernels and this planner
                    traffic, tpath = rec["hbm_bytes_gfx950_corrected"], cand
                    break
        achieved = tm["build_bytes"] / (build_ms * 1e-3) / 1e9 if build_ms > 0 else 0.0
        out = {
            "metric": "LM iterations/sec (+ final reprojection RMSE), 1k cams / 1M pts",
            "step_definition": "one LM damping trial = one reduced camera system built, solved and gain-ratio tested (sba's nlss unit, "
                               "info[9]); rejected trials and the restart probe of a finished solve count as steps",
            "value": args.steps / dt, "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            **({"rehearsal": "ranks share one GPU, host transport: control flow only, not a measurement"} if rehearsal else {}),
            "rccl": {"transport": "host test transport (rehearsal)" if rehearsal else ("rccl" if world > 1 else "none (one rank: no exchange)"),
                     "ranks": [{k: c[k] for k in ("rank", "nranks", "device", "hip_device", "pid")} for c in comms]},
            "config": {"workload": f"{args.config}: {sc.m} cams / {sc.n} pts / {sc.nobs} obs, cnp 7 (pose + focal), "
                       "point-sharded Schur + block-Jacobi PCG", "parallelism": f"points/{world}",
                       "pcg_tol": args.pcg_tol, "pcg_maxit": args.pcg_maxit,
                       "pcg_matrix": "f32 storage of the scaled matrix (> 64 MB: streamed every iteration), f64 vectors and sums" if tm.get("pcg_matrix_f32") else "f64", "deterministic": args.deterministic, "shared_focal_weight": args.shared_focal},
            "rmse_px": {"initial": rmse0, "after_timed_steps": rmse_after, "final_of_last_completed_solve": done["rmse"]},
            "accepted_steps": int(done["accepted"]), "completed_solves": done["solves"], "lm_iterations_of_last_completed_solve": done["iters"],
            "pcg_iters_per_step": tm["pcg_iters"] / max(tm["pcg_solves"], 1), "pcg_resumes": int(tm["pcg_resumes"]),
            "trials_linearised_ahead": int(tm["spec_adopted"]),
            "kernel_ms_per_step": {"build": build_ms, "pcg": tm2["pcg_ms"] / max(tm2["pcg_timed"], 1),
                                   "backsub": tm2["backsub_ms"] / max(tm2["backsub_launches"], 1),
                                   "note": "build: " + build_src + "; pcg, backsub: the same trials repeated untimed with events around them",
                                   "allreduce_host": tm["allreduce_ms"] / max(tm["allreduces"], 1) if world > 1 else 0.0,
                                   # several ranks: what does not shrink with N -- the exchange itself and the work every rank repeats
                                   "exchange": tm2["exchange_ms"] / max(tm2["exchanges_timed"], 1) if world > 1 else 0.0,
                                   "replicated": tm2["replicated_ms"] / max(tm2["replicated_timed"], 1) if world > 1 else None,
                                   "overlap": 0.0},
            "setup_s": {"scene": t_gen, "plan_upload": t_plan},
            "roofline": {"bound": "hbm", "kernel": "the build: k_build (fused linearise + Schur complement in register tiles) + the row path "
                                                    "(k_build row tiles, k_long_reduce, k_schur_pairs, k_schur_super, k_schur_blocks) between one pair of HIP events",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": os.path.relpath(tpath, ROOT) + " (rocprofv3 FETCH_SIZE / WRITE_SIZE passes)" if traffic
                         else "none: no PMC pass under profiles/ was taken on the present kernels.hip / plan.cpp (tools/profile_round.sh)",
                         "algorithmic_bytes_per_launch": tm["build_bytes"],
                         "algorithmic_bytes_formula": "24 k + 96 n + 192 m + 512 nb_up (SURVEY 8d Pass A in f64; the engine's own tables are not in it)",
                         "engine_table_bytes_per_launch": tm["build_index_bytes"], "block_handovers_per_launch": tm.get("build_block_handovers"), "avg_launch_ms": build_ms,
                         # the kernel is not HBM-bound: its arithmetic against the f64 vector peak
                         "valu_f64": {"achieved": tm["build_flops"] / (build_ms * 1e-3) / 1e12 if build_ms > 0 else 0.0,
                                      "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tm["build_flops"] / (build_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS if build_ms > 0 else 0.0,
                                      "algorithmic_flops_per_launch": tm["build_flops"]}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_mod, out["value"], args.config, args.cpu_baseline_c3)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
