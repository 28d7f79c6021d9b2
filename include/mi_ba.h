/*
 * mi_ba.h -- C ABI of the MI355X-native bundle-adjustment engine (libmi_ba.so).
 *
 * Drop-in boundary for the Optimizer/ hot path of large-scale-sfm.  Plain C, host pointers
 * and sizes only; the library owns every HIP allocation, stream, graph and RCCL communicator.
 * The C++ plugin surface of the reference (class ParallelBA, Optimizer/pba/src/pba/pba.h:52-133)
 * and its C solver entry points (sba_motstr_levmar, Optimizer/sba-1.5/sba.h:95-108; run_sfm,
 * SiftGPU/ba_libs/lib/sfm-driver/sfm.h:68-86) are thin shims over THIS header -- see
 * include/mi_pba.h, include/mi_sba.h and INTEGRATION.md.
 *
 * Parameterisation = the one Bundler's run_sfm hands to sba (sfm.c:657-709): per camera
 *   a_j = [ C_j or t_j (3) | omega_j (3) | f_j * f_scale (if est_focal) | k1_j * k_scale (if undistort) ]
 * around a fixed initial rotation R0_j (R_j = exp([omega_j]x) R0_j, sfm.c:78-117,545-553), per
 * point b_i = [X Y Z].  Projection (sfm.c:303-374):
 *   q = R_j (X - C_j)   or   R_j X + t_j ;   x^ = proj_sign * f * q.xy / q.z * (1 + k1 r^2),  r^2 = |q.xy / q.z|^2
 * proj_sign = -1 is Bundler's camera (sfm.c:333), +1 the VisualSFM / PBA one (+z forward).
 * All arithmetic on the device is IEEE double.
 *
 * Error behaviour follows the reference: entry points return >= 0 on success and MI_BA_ERROR (-1,
 * = SBA_ERROR, sba.h:57) on failure with a message in mi_ba_last_error(); nothing throws; a
 * missing GPU / HIP runtime failure is an error, never a silent CPU fallback.
 */
#ifndef MI_BA_H
#define MI_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_BA_ERROR (-1)
#define MI_BA_INFOSZ 10      /* SBA_INFOSZ, sba.h:56 */
#define MI_BA_VERSION 100

typedef struct mi_ba_engine mi_ba_engine;
typedef struct mi_ba_plan mi_ba_plan;

typedef struct {
    /* model (sfm.c:598-617 arguments of run_sfm) */
    int est_focal;          /* 1: focal length is a parameter (cnp 7), 0: fixed (cnp 6) */
    int undistort;          /* 1: radial coefficients are parameters (Bundler's projection distortion, sfm.c:360-373): k1 alone
                             * (nk = 1, the one-coefficient model of PBA's CameraT) or k1 and k2 (nk = 2);
                             * 2: PBA's measurement distortion, residual x (1 + k |x|^2) - f q.xy / q.z
                             * (SparseBundleCPU.cpp:1188-1193), k in the same parameter slot */
    int explicit_centers;   /* 1: camera params hold the centre C, 0: the translation t */
    int mcon;               /* first mcon cameras are held fixed (sba.h:95 'mcon') */
    double proj_sign;       /* -1 Bundler, +1 VisualSFM/PBA */
    double f_scale;         /* 0.001, sfm.c:640 */
    double k_scale;         /* 5.0,   sfm.c:641 */
    /* Levenberg-Marquardt controller (sba_levmar.c:506-511,609-612) */
    int itmax;              /* 150 = MAX_ITERS, sfm.c:820 */
    double opts[6];         /* tau, eps1, eps2, eps3, eps4, eps5 (sfm.c:711-719) */
    int disable_stops;      /* bench only: keep iterating after the stop rules would fire */
    /* reduced camera system: block-Jacobi PCG (role of sba_Axb_Chol, sba_lapack.c:374) */
    int pcg_maxit;
    double pcg_tol;         /* stop when sqrt(r'z / r0'z0) < pcg_tol */
    /* placement */
    int device;             /* HIP device ordinal */
    int rank, world;        /* point shard of this process; world > 1 needs mi_ba_comm_init */
    int verbose;
    /* planner knobs (0 = default) */
    int tile_max_batches;
    /* 0: cameras + points (sba_motstr_levmar), 1: cameras only, points fixed (sba_mot_levmar, sba.h:110-116;
     * ParallelBA::BUNDLE_ONLY_MOTION), 2: points only, cameras fixed (sba_str_levmar, sba.h:118-123;
     * BUNDLE_ONLY_STRUCTURE) */
    int mode;
    /* 0 (default): points whose camera set repeats accumulate their blocks of S in registers and hand them over with f64 atomics;
     * results differ from run to run at rounding level.  1: the same tiles leave per-tile copies of their blocks and camera
     * vectors instead, and a second kernel sums the copies in tile order (csrc/kernels.hip k_tile_reduce_*): no floating-point
     * atomics anywhere, every sum has a fixed order, two runs give the same bits (the reference is deterministic:
     * sba_levmar.c:1182-1339) -- at about the speed of 0.  2: every point takes the block-major row path (csrc/plan.h),
     * round 2's first deterministic mode (4x slower on scenes whose camera sets repeat; what nine-parameter cameras always use). */
    int deterministic;
    /* radial coefficients that are parameters with undistort == 1: 1 (default; 0 means 1) = k1, 2 = k1 and k2, Bundler's full
     * model (sfm.c:360-373, 648-649).  With the focal length that is NINE camera parameters (cnp 9): the engine then keeps camera j
     * as two rows of its 8-wide camera-side arrays (parameters 0-7, parameter 8) and every point takes the row path. */
    int nk;
    /* Levenberg-Marquardt variant.  0 (default): sba's (sba_levmar.c:1085-1617) -- additive damping mu I started at tau max diag,
     * the stop rules of opts[1..5].  1: PBA's (SparseBundleCPU.cpp:3797-3983, what ParallelBA::RunBundleAdjustment runs) --
     * multiplicative damping, U* = U + mu diag(U), V* = V + mu diag(V) (:1813-1814), started at opts[0] itself
     * (__lm_initial_damp); a trial is accepted when the cost goes down; the gain ratio is the cost change over
     * d'(mu D d + g), D = diag(J'J) (the exact-solve form of :3740-3745); mu <- mu max(1/3, 1 - (2 rho - 1)^3) kept within
     * [opts[3], opts[4]] after a success (__lm_minimum_damp, __lm_maximum_damp), mu <- mu nu, nu <- 2 nu after a failure;
     * stops: sqrt(d'D d) <= opts[2] (__lm_delta_threshold; stop code 2), gradient |g|_inf <= opts[1] when opts[1] > 0
     * (__lm_check_gradient; stop code 1), itmax TRIALS (stop code 3).  The mean-squared-error and time-budget rules are the
     * caller's (pba_shim.cpp).  opts[5] > 0 is PBA's __lm_damping_auto_switch (:3962-3966): after a failed trial whose mu is above
     * it the solve continues as variant 2 from mu = opts[5]; the upper bound opts[4] holds under variant 1 only (:3948).
     * 2: the same controller with PBA's identity damping (mu I; its
     * __lm_use_diagonal_damp off). */
    int lm_variant;
    int pcg_minit;          /* the reduced solve's tolerance test counts from this iteration on (PBA: __cg_min_iteration = 10) */
    int reserved[2];
} mi_ba_config;

/* fills the defaults above (Bundler's run_sfm settings, cnp 7, explicit centres) */
void mi_ba_default_config(mi_ba_config *cfg);

/* ---- engine life cycle ---------------------------------------------------------------- */
mi_ba_engine *mi_ba_create(const mi_ba_config *cfg);     /* NULL when no usable HIP device */
void mi_ba_destroy(mi_ba_engine *e);
const char *mi_ba_last_error(const mi_ba_engine *e);      /* e == NULL: error of mi_ba_create */
int mi_ba_cnp(const mi_ba_engine *e);

/* ---- problem definition; roles of ParallelBA::SetCameraData / SetPointData / SetProjection
 *      (pba.h:96-101).  Host arrays are copied; the caller keeps ownership. ------------------- */
int mi_ba_set_cameras(mi_ba_engine *e, int m, const double *a0 /* m*cnp */, const double *R0 /* m*9 */,
                      const double *f_fixed /* m, used when !est_focal; may be NULL otherwise */);
/* optional: cameras held constant beyond the leading mcon (CameraT::constant_camera, DataInterface.h:72);
 * mask[j] != 0 fixes camera j; NULL clears.  Call after mi_ba_set_cameras. */
int mi_ba_set_fixed_cameras(mi_ba_engine *e, const unsigned char *mask /* m */);
/* optional: single camera parameters held constant -- mask[j*cnp + r] != 0 keeps parameter r of camera j at its initial
 * value while the rest of the camera moves (its row and column leave the reduced system).  What PBA does with the distortion
 * coefficient under SetFixedIntrinsics(true): applied, not estimated (JacobianOne, SparseBundleCPU.cpp:1362-1363).  NULL
 * clears.  Call after mi_ba_set_cameras and before prepare. */
int mi_ba_set_fixed_params(mi_ba_engine *e, const unsigned char *mask /* m*cnp */);
int mi_ba_set_points(mi_ba_engine *e, int n, const double *xyz /* n*3 */);
/* observations sorted by point (pt_idx non-decreasing, like pba's driver.cpp:158-161 requires) */
int mi_ba_set_observations(mi_ba_engine *e, int64_t nobs, const double *xy /* nobs*2 */,
                           const int *pt_idx, const int *cam_idx);
/* Bundler's soft constraints (camera_constraints_t / point_constraints_t, sba.h:76-86): penalty
 *   sum_j,r w_jr (v_jr - a_jr)^2  +  nobs * sum_i w_i |v_i - b_i|^2
 * added to the cost, U_j / V_i diagonals and gradients as sba_levmar.c:809-842, 952-963, 1017-1029 do -- including
 * the reference's lag (the penalty of a trial point is evaluated at the old p, :1498).  A weight of 0 leaves the
 * entry free; NULL clears.  Call after set_cameras / set_points and before prepare. */
int mi_ba_set_camera_constraints(mi_ba_engine *e, const double *weights /* m*cnp */, const double *values /* m*cnp */);
int mi_ba_set_point_constraints(mi_ba_engine *e, const double *weight /* n */, const double *values /* n*3 */);
/* quadratic couplings between the parameters of two cameras: penalty  sum_q,r w_qr (a_{A(q),r} - a_{B(q),r})^2.
 * PBA's shared-intrinsics "Q" constraints (ParallelBA::SetFocalMask, pba.h:124; SparseBundleCPU.cpp:1262-1329,
 * 2781-2892) are links on the focal-length (and distortion) entries around a ring of cameras.  The couplings
 * enter U (diagonal and the (A,B) off-diagonal block, which the plan adds to the pattern of S), the gradient and
 * the cost exactly; weights are in the units of the internal parameters a_j.  nlink = 0 clears.  Call after
 * set_cameras and before prepare. */
int mi_ba_set_camera_links(mi_ba_engine *e, int nlink, const int *cam_a, const int *cam_b, const double *weights /* nlink*cnp */);
/* builds the shard plan, uploads everything to HBM */
int mi_ba_prepare(mi_ba_engine *e);

/* ---- solve: the role of sba_motstr_levmar / RunBundleAdjustment ------------------------- */
/* returns the number of LM iterations (>= 0) or MI_BA_ERROR; info[] as sba's (sba_levmar.c:512-527) */
int mi_ba_run(mi_ba_engine *e, double info[MI_BA_INFOSZ]);
/* step-wise driving (bench, tests): begin() evaluates the initial cost and picks mu0,
 * step() = ONE damping trial = one reduced system built + solved + gain-ratio test.
 * step() returns 1 accepted, 0 rejected, 2 stopped (see info[6]), MI_BA_ERROR on failure. */
int mi_ba_begin(mi_ba_engine *e);
/* back to the parameters of the last mi_ba_set_params / mi_ba_prepare (kept in HBM: no host traffic), then mi_ba_begin */
int mi_ba_restart(mi_ba_engine *e);
int mi_ba_step(mi_ba_engine *e);
int mi_ba_get_info(const mi_ba_engine *e, double info[MI_BA_INFOSZ]);
/* per trial: itno, mu, cost_before, cost_trial, accepted, pcg_iterations (6 doubles each) */
int mi_ba_get_trace(const mi_ba_engine *e, double *trace, int cap);
/* current estimate, original point order */
int mi_ba_get_params(mi_ba_engine *e, double *a /* m*cnp */, double *xyz /* n*3 */);
/* reset the estimate (keeps the plan) */
int mi_ba_set_params(mi_ba_engine *e, const double *a, const double *xyz);
/* final rotations R_j = exp([omega_j]x) R0_j, m*9 (sfm.c:897) */
int mi_ba_get_rotations(mi_ba_engine *e, double *R);

/* ---- parity hooks: what sba exports through Uout/Vout/Sout (sba_levmar.c:1633-2026) -------
 * One linearisation at the current estimate with damping mu (mu = 0 allowed when every V_i is
 * invertible).  Any pointer may be NULL.
 *   S   (m*cnp)^2 row-major reduced camera matrix  U* - W V*^-1 W^T
 *   E   m*cnp  reduced right-hand side,  ea m*cnp = sum A^T e,  eb n*3 = sum B^T e,
 *   V   n*9 undamped V_i (full symmetric),  diagU m*cnp, returns cost in *cost              */
int mi_ba_debug_linearize(mi_ba_engine *e, double mu, double *S, double *E, double *ea, double *eb,
                          double *V, double *diagU, double *cost);
/* after debug_linearize on an engine whose every point takes the row path (mi_ba_config.deterministic = 2, one rank): the full blocks
 * sba exports through Uout / Wout -- U m*cnp*cnp (row-major U_j, camera-constraint weights on the diagonal), W nobs*cnp*3 (W_ij = A^T B
 * of observation k in the caller's order).  Either may be NULL. */
int mi_ba_debug_blocks(mi_ba_engine *e, double *U, double *W);
/* after debug_linearize: solve S da = E with the PCG, back-substitute; da m*cnp, db n*3,
 * cost at p + dp; returns PCG iterations or MI_BA_ERROR */
int mi_ba_debug_solve(mi_ba_engine *e, double mu, double *da, double *db, double *cost_new);

/* ---- measurement: device-side timings from HIP events on the engine's stream ------------ */
typedef struct {
    double build_ms, build_launches;       /* fused linearise + Schur kernel(s) */
    double pcg_ms, pcg_solves, pcg_iters;
    double backsub_ms, backsub_launches;
    double step_ms, steps;                 /* whole trial incl. host control */
    double allreduce_ms, allreduces;
    double build_bytes;                    /* algorithmic HBM bytes of one build launch */
    double pcg_resumes;                    /* solves whose unchecked first run of PCG iterations fell short (engine.cpp solve_phase) */
    double build_flops;                    /* algorithmic f64 flops of one build launch: 400 per observation (residual, Jacobian,
                                            * V / e_b / E terms), 2 x 138 per pair block, 120 per point (3x3 inverse, h) */
    double pcg_timed;                      /* solves that pcg_ms covers (mi_ba_set_timing_detail >= 2) */
    double spec_adopted;                   /* trials whose linear phase had been enqueued behind the previous trial (engine.cpp solve_phase) */
    double build_index_bytes;              /* the engine's own tables and scratch one build launch also moves (pair entries, row-path records) */
    /* several ranks, mi_ba_set_timing_detail >= 2: device time of the per-trial exchange (the all-reduce of [S_up | camvec | scal] on the
     * engine's stream) and of what every rank then computes REDUNDANTLY (finalize, block inverses, scaled matrix, the whole CG, camera
     * update): the part of a trial that does not shrink with the number of GPUs */
    double exchange_ms, exchanges_timed;
    double replicated_ms, replicated_timed;
    double pcg_persistent;                 /* reduced solves done by the one persistent launch (k_pcg_persist) instead of a launch per iteration */
    double pcg_matrix_f32;                 /* 1: the reduced solve streams its scaled matrix in single precision (systems whose matrix exceeds 64 MB at the
                                            * default tolerance; vectors, dot products and recurrences f64) -- engine.cpp pcg_f32_ */
    double pcg_push;                       /* 1: the reduced solve reads its neighbours' vectors from per-block mailboxes (opt-in MI_BA_PCG_PUSH=1) */
    double build_block_handovers;          /* blocks of S the tiles of one build launch hand over with f64 atomics (or as per-tile copies): the plan's slots summed
                                            * over the tiles -- exact for the tiles of one camera set (one hand-over per block and tile), a lower bound for the others */
    double reserved[7];                    /* the struct stays 32 doubles from here on: new figures come out of this tail */
} mi_ba_timing;
/* ABI note: the struct grew twice before it got its tail (17 doubles in round 1, 21 in round 2, 32 now).  mi_ba_get_timing is the
 * symbol every one of those headers declares, so it only ever writes the 17 doubles they all share (build_ms .. build_index_bytes);
 * the whole block comes through mi_ba_get_timing_n with the CALLER's sizeof (at most that many bytes are written). */
int mi_ba_get_timing(const mi_ba_engine *e, mi_ba_timing *t);
int mi_ba_get_timing_n(const mi_ba_engine *e, void *t, size_t bytes);
int mi_ba_reset_timing(mi_ba_engine *e);
/* which phases are bracketed by HIP events: 0 (default) none, 1 the build kernels (every fourth build: build_ms / build_launches is
 * the mean over those), 2 every build, the reduced solve and the
 * back-substitution.  An event on the stream costs ~6 us of idle GPU, so device-side times are something to ask for
 * (bench.py: level 1 in its timed steps, level 2 in a second, untimed pass), not something every trial pays. */
int mi_ba_set_timing_detail(mi_ba_engine *e, int level);

/* ---- multi-GPU: one process per GPU, points sharded, RCCL all-reduce of the Schur partials */
#define MI_BA_UNIQUE_ID_BYTES 128
int mi_ba_comm_unique_id(char id[MI_BA_UNIQUE_ID_BYTES]);                 /* rank 0 */
int mi_ba_comm_init(mi_ba_engine *e, const char id[MI_BA_UNIQUE_ID_BYTES]); /* every rank, before prepare */
/* what RCCL itself says about this rank's communicator: out[0] = ncclCommCount, out[1] = ncclCommCuDevice (the HIP device the
 * communicator is bound to), out[2] = ncclCommUserRank, out[3] = 1 when the exchanges go through RCCL (0: one rank and no
 * communicator yet, or the host test transport).  Lets a launcher prove that N ranks on N devices were really joined. */
int mi_ba_comm_info(const mi_ba_engine *e, int out[4]);
/* test transport: instead of RCCL, every exchange goes device -> host, through `fn` (which must leave the element-wise
 * sum (op 0) or maximum (op 1) over all ranks in buf), and back.  Lets the sharded engine be exercised by several
 * processes that share ONE GPU (tests/test_sharding_gpu.py, gloo underneath); never used by bench.py. */
typedef int (*mi_ba_host_allreduce_fn)(void *user, double *buf, int64_t count, int op);
int mi_ba_comm_set_host_allreduce(mi_ba_engine *e, mi_ba_host_allreduce_fn fn, void *user);

/* ---- planner (host only; usable without a GPU) -------------------------------------------
 * Same code the engine runs in mi_ba_prepare: point permutation (visibility-signature sort),
 * shard cut, wave batches, LDS tiles, pair tables and the block-sparse pattern of S.          */
mi_ba_plan *mi_ba_plan_build(int n, int m, int64_t nobs, const int *pt_idx, const int *cam_idx,
                             int rank, int world, int tile_max_batches);
/* the same with every point on the row path (mi_ba_config.deterministic) */
mi_ba_plan *mi_ba_plan_build_rows(int n, int m, int64_t nobs, const int *pt_idx, const int *cam_idx,
                                  int rank, int world, int tile_max_batches);
void mi_ba_plan_free(mi_ba_plan *p);
/* stats: see MI_BA_PLAN_* indices */
enum {
    MI_BA_PLAN_NPTS_LOCAL = 0, MI_BA_PLAN_NOBS_LOCAL, MI_BA_PLAN_NBATCH, MI_BA_PLAN_NTILE,
    MI_BA_PLAN_NPAIR, MI_BA_PLAN_NB_UPPER, MI_BA_PLAN_NB_FULL, MI_BA_PLAN_NSLOT_TOTAL,
    MI_BA_PLAN_NBATCH_ROWS, MI_BA_PLAN_NPAIR_ROWS, MI_BA_PLAN_NLONG, MI_BA_PLAN_PT_BEGIN,
    MI_BA_PLAN_PT_END, MI_BA_PLAN_MAX_SLOTS, MI_BA_PLAN_NREC, MI_BA_PLAN_NROW_PTS, MI_BA_PLAN_NSTATS = 16
};
int mi_ba_plan_stats(const mi_ba_plan *p, int64_t stats[MI_BA_PLAN_NSTATS]);
/* array access for tests: returns element count, copies min(cap, count) elements.
 * names: "perm" (int32, internal->original point), "obs_cam", "obs_pt" (int32, internal local),
 * "obs_src" (int64 original observation index), "batch_obs", "batch_pair", "batch_pt", "tile_batch",
 * "tile_slot", "slot_gblk", "tile_cslot", "cslot_cam", "pair" (uint32),
 * "obs_cslot" (uint8), "up_row" (int32 m+1), "up_col" (int32), "bsr_row", "bsr_col", "bsr_src",
 * "tile_mode" (uint8: 1 = blocks in registers, 2 = blocks in an LDS table, 0 = row tile); the row path: "obs_rec" (int32 per
 * observation, -1 = none), "rec_cam" (m+1), "wk_blk", "wk_q" (int64, nwork+1), "rpair" (uint32, 2 per pair),
 * "wk_diag_cam", "wk_diag_blk"                                                                  */
int64_t mi_ba_plan_get(const mi_ba_plan *p, const char *name, void *out, int64_t cap_elems);

#ifdef __cplusplus
}
#endif
#endif /* MI_BA_H */
