// kernels.h -- device-side data layout and launch wrappers (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "plan.h"

namespace miba {

// ---- HBM layout ---------------------------------------------------------------------------
// cam_a    [m][8]   camera parameters a_j, padded to 8 doubles (64 B rows)
// cam_lin  [m][24]  per-linearisation camera record: R(9) = exp([w]x) R0, Jl(9) = left Jacobian of
//                   SO(3) at w, c(3) = C or t, f, k1, k2   (192 B, L2-resident, gathered per obs)
// pts      [npl][3] this rank's points, internal (signature-sorted) order
// obs_xy   [nol]    double2 measurement; obs_cam/obs_pt int32; obs_cslot uint8
// S_up     [nb_up][64]  upper blocks of the reduced camera matrix (8x8 padded, row-major)
// camvec   [3][m][8]    ea_j | E_j partial (= sum A^T(e - B h)) | diag U_j
// ptout    [npl][12]    (V_i + mu I)^-1 (6: 00 01 02 11 12 22) | h_i = V*^-1 eb_i (3) | eb_i (3)
// S_full   [nb][64]     BSR values of the reduced system, Linv [m][64] inverse Cholesky factors of its diagonal blocks,
// S_pcg    [nb][64]     Linv S Linv^T, what the CG multiplies with
constexpr int kCamLin = 24;
constexpr int kPtOut = 12;
constexpr int kPartW = 8;      // per-workgroup partial row: [0,4) sums, [4,8) maxima
constexpr int kBacksubGridMax = 16384; // the most the MI_BA_BACKSUB_GRID probe may ask for: the partial-row buffer holds that many rows
constexpr int kBacksubGrid = 6144;    // workgroups of k_backsub (each walks a contiguous chunk of batches): two rounds of the 3 072 the chip holds at 3 wavefronts per SIMD
constexpr int kReduceStage = 64;      // first-stage workgroups of k_reduce_partials

struct ModelCfg {
    int explicit_centers, est_focal, undistort, mcon;
    int nk;             // radial coefficients that are parameters with undistort == 1: 1 (k1) or 2 (k1, k2: Bundler, sfm.c:648-649)
    int split;          // 9 camera parameters: camera j is rows 2j (parameters 0-7) and 2j + 1 (parameter 8) of every camera-side array
    int points_fixed;   // motion-only bundle: B_ij := 0 (sba_mot_levmar_x has no structure block)
    int pba_update;     // PBA's UpdateCamera (SparseBundleCPU.cpp:1073-1105): f = max(f + d, 1e-10) and R <- exp(d omega) R with omega back at 0 after every
                        // step -- the rotation the camera is linearised about moves with it (R0 is double-buffered like the parameters)
    int damp_diag;      // PBA's damping: U* = U + mu diag(U), V* = V + mu diag(V) (SparseBundleCPU.cpp:1813-1814) instead of sba's + mu I
    double proj_sign, f_scale, k_scale;
    double inv_f_scale, inv_k_scale;   // 1 / f_scale, 1 / k_scale (the per-observation kernels multiply; a division by a uniform is 13 instructions there)
};

// scalar block (doubles) shared by kernels and the host LM controller; lives at the tail of the
// accumulation buffer (engine.h) so the SUM part of all-reduce #1 is one contiguous range
enum Scal {
    // [0,8)  k_build sums over this rank's observations / points      (all-reduce #1, SUM)
    SC_COST = 0, SC_PL2_PTS = 1, SC_CONS_PTS = 2,   // CONS_PTS: soft point-constraint penalty at p
    // [8,16) k_build maxima                                            (all-reduce #1, MAX)
    SC_EB_INF = 8, SC_MAXDIAG_V = 9, SC_SINGULAR = 10,
    SC_SPEC_GO = 13, SC_SPEC_MU = 14,   // k_lm_peek's verdict, copied here so that it travels to the host with the rest
    SC_PCG_FINAL = 12,   // (not a maximum: zero through all-reduce #1, set by k_pcg_tail afterwards) the solve is over: converged,
                         // broken down or out of iterations -- what the back-substitution waits for
    // [16,32) camera-side values, computed redundantly on every rank: every sum is folded in a FIXED order
    // (per-wave partial rows + k_reduce_partials), so all ranks hold the same bits and take the same LM decisions
    SC_PL2_CAMS = 16, SC_CONS_CAM = 17, SC_LINK = 18,   // k_finalize_vec sums  [16,20)
    SC_EA_INF = 20, SC_MAXDIAG_U = 21,                  // k_finalize_vec maxima [20,24)
    SC_DL_CAMS = 24, SC_DP2_CAMS = 25, SC_LINK_NEW = 26, // k_cam_update sums    [24,28)
    SC_PCG_ITERS = 28, SC_PCG_STATUS, SC_PCG_DONE, SC_NOT_PD,
    // [32,40) k_backsub sums                                           (all-reduce #2, SUM)
    SC_COST_NEW = 32, SC_DL_PTS = 33, SC_DP2_PTS = 34,
    // [40,48) k_backsub maxima                                         (all-reduce #2, MAX)
    SC_MAX_PCT = 40, SC_SPARE_MAX = 44,                 // [44,48): dump for unused maxima columns
    SC_COUNT = 48
};

// quadratic couplings between parameters of two cameras, cost += w_r (a_j[r] - a_k[r])^2: PBA's shared-intrinsics
// "Q" constraints (SetFocalMask, SparseBundleCPU.cpp:1262-1329, 2781-2892).  Every link is listed at both ends.
struct CamLinks {
    const int *row;        // m + 1, CSR over cameras; nullptr = no links
    const int *other;      // linked camera of each entry
    const double *w;       // kCP weights per entry (internal parameter units), 0 = uncoupled
    const int *blk_pair;   // per block of the full BSR pattern: index into pair_w, or -1
    const double *pair_w;  // kCP summed weights per linked camera pair
};

struct BuildArgs {
    const unsigned char *tile_mode; // per tile: 1 register-accumulated blocks, 2 LDS block table, 3 one camera set on the matrix pipe, 4 one camera set on the vector pipe (k_build_uni), 0 row tiles
    int has_table, has_reg, has_mfma, has_uni; // some tile of this launch range has mode 2 / 1 / 3 / 4 (has_uni: with a view count other than kUniFixedViews)
    int has_uni_fixed;                        // ... mode 4 with kUniFixedViews views: k_build_uni's instantiation with the view count as a constant
    const int *tile_batch, *tile_slot, *tile_cslot, *slot_gblk, *cslot_cam;
    const int *batch_obs, *batch_pt;
    const long long *batch_pair;
    const int *pt_obs;
    const int *obs_cam, *obs_pt;
    const unsigned char *obs_cslot, *cam_fixed;   // cam_fixed [m]: 1 = camera held constant (A_ij := 0)
    const double2 *obs_xy;
    const unsigned *pair;
    const int *obs_rec;                     // row tiles: record of every observation (plan.h)
    double *rec;                            // the records, kRec doubles each
    int tile0, m, want_diag;                // want_diag: also accumulate diag U_j (mu0 and info[4] only)
    const double *cam_lin, *pts;
    const double *pt_cw, *pt_cv;            // soft point constraints (weight already x nvis), internal order; NULL = none
    double mu;
    const double *spec;                     // speculative launch (engine.cpp solve_phase): spec[0] == 0 -> do nothing, else mu = spec[1]; nullptr: mu above
    unsigned long long *dbg;                // diagnostic stamp sums (MIBA_STAMPS build), else NULL
    double *S_up, *camvec, *ptout, *part;   // part: [ntile][kPartW] cost, |X|^2, -, -, |eb|inf, max diag V, singular
    double *tile_blk, *tile_cam;            // ordered hand-over (deterministic = 1): per-tile copies [ntile][kNSlot][cnp^2], [ntile][kNCSlot][24]; nullptr: f64 atomics into S_up, camvec
    ModelCfg cfg;
};

struct BacksubArgs {
    const int *batch_obs, *batch_pt, *pt_obs, *obs_cam, *obs_pt;
    const int *batch_obs_end, *batch_pt_end;   // batch b = observations [batch_obs[b], batch_obs_end[b]), points [batch_pt[b], batch_pt_end[b]): runs of consecutive points with gaps between batches (plan.h bs_uni)
    const unsigned char *cam_fixed;
    const double2 *obs_xy;
    const double *cam_lin, *cam_lin_new, *pts, *ptout, *dA;
    double *pts_new, *dB, *part;            // part: [ngrid][kPartW] cost_new, dL, |db|^2, -, max pct change
    double mu, eps5;
    int nbatch, ngrid;
    ModelCfg cfg;
    const double *scal;                     // SC_PCG_FINAL == 0: the reduced solve has not finished, nothing to do yet
    // k_backsub_uni: the tiles of one camera set (plan.h mode 4) -- the build's tile tables; their points are not in batch_obs / batch_pt above
    const unsigned char *tile_mode;
    const int *tile_batch, *tile_cslot, *cslot_cam, *bld_batch_obs, *bld_batch_pt;
    int uni_ntile, uni_row0;                // workgroups (= LDS tiles of the plan; 0: no uniform tile), first partial row of theirs
    int uni_fixed, uni_other;               // some uniform tile has kUniFixedViews views (the instantiation with that constant) / another view count
};

// long tracks (> 64 views): one workgroup per point; the records go where the row tiles' go
struct LongArgs {
    const int *long_pts, *pt_obs, *obs_cam, *obs_rec;
    const unsigned char *cam_fixed;
    const double2 *obs_xy;
    const double *cam_lin, *cam_lin_new, *pts, *dA, *pt_cw, *pt_cv;
    double *rec, *ptout, *part, *pts_new, *dB;
    double mu, eps5;
    int nlong, m, part_row0;
    ModelCfg cfg;
    const double *scal;
    const double *spec;                     // as in BuildArgs (the linear-phase kernels)
};

// k_schur_blocks: the blocks of S (and the camera-side vectors) of the row path, summed from the observation records
struct SchurArgs {
    const double *rec;
    const int *wk_blk;            // off-diagonal work items: upper block ...
    const long long *wk_q;        // ... and its pair range
    const unsigned *rpair;        // 2 per pair: record of the row camera, record of the column camera
    const int *diag_cam, *diag_blk, *rec_cam;   // diagonal work items: camera, its diagonal block, its record range
    int nwork, ndiag, m, want_diag;
    int mv;                       // rows of the camera-side arrays (2 m with nine camera parameters: ModelCfg.split)
    const int *wk_vblk, *diag_vblk;   // split: the 4 virtual blocks of every work item (diagonal items: 0, 1 and 3 used)
    const int *sb_blk, *sb_ent;       // super-blocks (plan.h): 16 blocks per work item; 8 records per entry, -1 = none
    const long long *sb_q;            // ... work item w: entries [sb_q[w], sb_q[w + 1]), a multiple of four
    int nsuper;
    const int *sb8_blk, *sb8_ent;     // ... the same with groups of eight cameras: 64 blocks per work item, 16 records per entry
    const long long *sb8_q;
    int nsuper8;
    double *S_up, *camvec;
    const double *spec;           // as in BuildArgs
};

// where the CG starts: written by k_block_inv, which holds L^-1 of every camera's diagonal block anyway
struct PcgStart { const double *E; double *x, *r0, *p, *s0, *s1; unsigned long long *bar; const int *list /* k_scale_blocks' work list: 8 ints per block on or above the diagonal (kernels.hip) */; int nlist; int ell /* with PcgArgs.ell_w: the blocks go to their fixed-pitch positions */; int layout /* of the scaled matrix in S_pcg: 0 f64 blocks row-major, 1 single precision (PcgArgs.S32), 2 f64 in the product's column form (PcgArgs.cm) */; /* k_pcg_persist's barrier words, zeroed here; or nullptr */ };
// grid-barrier words of the persistent reduced solve (u64, 64-byte apart): root counter, generation, abort, one counter per group
constexpr int kBarRoot = 0, kBarGen = 8, kBarAbort = 16, kBarGroup0 = 32, kBarMaxGroups = 32, kBarWords = kBarGroup0 + 8 * kBarMaxGroups;

// Two-level preconditioner of the reduced solve: block-Jacobi (the scaling) + an additive coarse correction on aggregates of
// consecutive cameras, M~^-1 = I + Z~ (Z'SZ)^-1 Z~', Z~ = L'Z -- applied INSIDE the one launch an iteration is (kernels.hip, k_pcg2_*).
constexpr int kBcSlots = 4;     // aggregates a block row may reach (its own included); more -> the coarse space is not used
constexpr int kCoarseMax = 16;  // aggregates at most: the coarse matrix (8 ng)^2 is inverted by one workgroup, in registers
struct CoarseArgs {
    int ng, nc, wg_per_agg;           // aggregates (0 = off), coarse unknowns = 8 ng, workgroups (4 block rows each) per aggregate
    const int *bc_agg;                // [m][kBcSlots]: the aggregates row j's blocks reach, -1 = unused
    double *Ac, *Acinv;               // Z'SZ and its inverse, nc x nc
    double *U, *Bc;                   // per row and slot: U = sum of the row's blocks of S in that aggregate, Bc = L^-1 U (= S~ Z~): [m][kBcSlots][64]
    double *Lfac;                     // L_j of S_jj = L_j L_j', [m][64]
    double *z;                        // preconditioned residual, own rows only
    double *cpart;                    // [2][3][nwg][8]: per workgroup the coarse projections Z~'(r | w | s) of its rows
};

struct PcgArgs {
    const int *bsr_row, *bsr_col;
    const double *S, *Linv, *E;                        // S = the scaled matrix S~ (k_scale_blocks)
    const float *S32;                                  // ... or, non-null, S~ in single precision in the same buffer (large systems: the product streams half the bytes)
    int cm;                                            // S (f64) holds each block in the product's column form (k_scale_blocks layout 2) instead of row-major
    double *mb0, *mb1;                                 // push halo (with cm): per block q of the pattern the r | w | s (3 x 8) of block row bsr_col[q], by iteration parity; nullptr: gathered by column index
    const int *bsr_tq;                                 // ... and where block q's transpose sits: bsr_col[bsr_tq[q]] = row of q
    int ell_w;                                         // > 0: S and bsr_col hold every block row at the fixed pitch ell_w (padded with zero blocks in the own column): row j = [j ell_w, (j + 1) ell_w)
                                                       // -- no row pointer between the kernel's arguments and the first matrix loads (engine.cpp pcg_ell_)
    double *x, *da, *p, *r0, *r1, *w0, *w1, *s0, *s1;  // r, w, s alternate by iteration parity (read by the neighbours' rows)
    double *gp, *dp, *rz_hist, *alpha_hist, *scal;     // gp, dp [2][nwg]: partials of r.r and w.r per workgroup; histories [maxit+2]
    double tol;
    int last;                                          // k_pcg_tail: no iteration will follow this one
    int tail;                                          // k_cam_update: do k_pcg_tail's work as well
    int m, ng, nwg, it;                                // ng = ceil(m / 8) (k_pcg_init); product kernels: nwg = m (one block row per
                                                       // workgroup, few cameras) or ceil(m / 4) (one per wavefront)
    int wpr;                                           // nwg = ceil(m / 4): wavefronts per block row, 1 or 2 (long rows)
    int minit;                                         // the tolerance test counts from this iteration on (PBA: __cg_min_iteration)
    double *tot;                                       // [2][2] or nullptr: the two dot products summed once per launch by k_pcg_totals (many workgroups:
                                                       // every workgroup adding up all nwg partials itself is O(nwg^2) loads per iteration)
    CoarseArgs co;                                     // co.ng == 0: plain block-Jacobi
    unsigned long long *stamps;                        // diagnostic build (make stamp, MI_BA_PCG_STAMPS=1): [it][16] wall-clock stamps of the first and the last workgroup of k_pcg_iter
};

// the whole reduced solve in one cooperative launch (kernels.hip k_pcg_persist): the scaled matrix in LDS, a grid barrier per iteration
struct PersistArgs {
    PcgArgs a;
    unsigned long long *bar;      // kBarWords barrier words, zeroed by k_block_inv every trial
    int rows_per_wg, maxit, group; // block rows per workgroup (<= 8), iteration cap, workgroups per barrier group
    const int *win_ptr, *win_col;  // per workgroup the distinct columns of its rows, ascending (CSR over workgroups): its window of r, w, s
    const int *blk_loc;            // per block of the full pattern: the position of its column in its workgroup's window
};
int pcg_persist_capacity(size_t lds_bytes);                                    // co-resident workgroups at that much dynamic LDS, 0 = cannot run
bool launch_pcg_persist(hipStream_t st, const PersistArgs &g, int nwg, size_t lds_bytes);   // false: the cooperative launch was refused

void launch_zero(hipStream_t st, double *p, size_t n);   // n doubles to zero (p 16-byte aligned)
void launch_cam_prep(hipStream_t st, int m, int cnp, const ModelCfg &cfg, const double *cam_a, const double *R0,
                     const double *f_fixed, double *cam_lin);
void launch_build(hipStream_t st, int cnp, bool lds_mode, int ntiles, const BuildArgs &a);
// sums of the per-tile copies in tile order (ordered hand-over): S_up blocks and camera vectors
void launch_tile_reduce(hipStream_t st, int cnp, int nb_up, int m, const int *blk_ptr, const int *blk_list, const int *cam_ptr, const int *cam_list,
                        const double *tile_blk, const double *tile_cam, double *S_up, double *camvec);
void launch_finalize(hipStream_t st, int cnp, int m, int nb, const unsigned char *fixed, const unsigned char *pfix, double mu, const double *spec, const int *bsr_rowidx, const int *bsr_col,
                     const int *bsr_src, const unsigned char *bsr_trans, const int *bsr_diag, const double *S_up, double *S_full /* or nullptr */,
                     double *camvec, const double *cam_a, const double *cam_cw, const double *cam_cv, const CamLinks &links,
                     double *E, double *cam_part, double *stage, double *scal, double *Linv, double *S_pcg,
                     bool reduce_now /* false: the camera-side sums are taken by launch_reduce_final at the end of the trial */,
                     const PcgStart &ps, const CoarseArgs &co /* co.ng != 0: also Z'SZ, its inverse, S~Z~ */, const int *bsr_row,
                     bool damp_diag /* PBA's damping: + mu diag(U) */);
void launch_pcg_init(hipStream_t st, const PcgArgs &a);  // w_0 = S~ r_0 and the first partials (r_0: k_block_inv)
void launch_pcg_reinit(hipStream_t st, const PcgArgs &a); // r_0, x~_0 again, for a second solve on the same linearisation
void launch_pcg_iter(hipStream_t st, const PcgArgs &a); // iteration a.it: one launch
void launch_pcg_tail(hipStream_t st, const PcgArgs &a); // rz_hist[a.it] = r_it . M^-1 r_it
void launch_reduce_partials(hipStream_t st, int nrows, const double *part, double *stage, double *out_sum, double *out_max);
void launch_fold_max(hipStream_t st, double *slots, int world, double *out8);
// the LM controller's accept test and damping update on the device (sba_levmar.c:1543-1552), from the trial's scalars:
// spec[0] = 1 when the trial will be accepted and the solve goes on, spec[1] = the damping of the next trial
struct PeekArgs {
    double c_lag, mu;                                  // the controller's state the scalars do not hold
    double eps2_sq, eps3_sq, eps4_sq, eps5, pl2_inactive;
    int itno, mode, stops, last;                       // last: the iteration cap ends the solve with this trial
    int variant;                                       // 0: sba's controller; 1, 2: PBA's (accept on any decrease, its damping bounds, its step stop)
    double mu_min, mu_max, d_thr;                      // PBA: opts[3], opts[4] (0 or variant 2: no upper bound), opts[2]
};
void launch_lm_peek(hipStream_t st, double *scal, const PeekArgs &pk, double *spec);
void launch_reduce_final(hipStream_t st, int bs_rows, const double *bs_part, double *stage, double *bs_sum, double *bs_max,
                         int m /* real cameras */, int mv /* rows of the camera-side arrays */, const double *fv_part, const double *cu_part,
                         double *scal, double *spec /* or nullptr */, const PeekArgs &pk,
                         int bp_rows /* the build's partial rows, still to be summed (0: done after the build) */, const double *bp_part, double *bp_sum, double *bp_max,
                         int fv_rows /* partial rows of the camera-side vectors: mv (inside k_block_inv) or ceil(8 mv / 64) (k_finalize_vec) */);
 // out8[c] = max over the ranks' slots, which are zeroed
void launch_cam_update(hipStream_t st, int m, int cnp, const ModelCfg &cfg, const unsigned char *fixed, double mu, const double *cam_a, const PcgArgs &pa,
                       double *dA, const double *camvec, const double *R0, const double *f_fixed, const CamLinks &links, double *cam_a_new,
                       double *cam_lin_new, double *cam_part, double *R0_new /* cfg.pba_update: the trial's rotations, or nullptr */);
// cfg.pba_update, at the start of a solve: R0 <- exp([omega]x) R0, omega <- 0, J_l <- I (idempotent)
void launch_cam_rebase(hipStream_t st, int m, int row_stride /* doubles between cameras in cam_a */, double *cam_a, double *cam_lin, double *R0);   // ends the scaled solve (da = Linv^T x~, + k_pcg_tail's work with pa.tail), then a + da
void launch_backsub(hipStream_t st, int cnp, const BacksubArgs &a);
void launch_long_build(hipStream_t st, int cnp, const LongArgs &a);   // k_long_reduce
void launch_schur_blocks(hipStream_t st, int cnp, const SchurArgs &a); // after every record of this linearisation has been written
void launch_long_backsub(hipStream_t st, int cnp, const LongArgs &a);
void launch_rotations(hipStream_t st, int m, const double *cam_lin, double *R);

} // namespace miba
