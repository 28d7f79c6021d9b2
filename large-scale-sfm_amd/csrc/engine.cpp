// engine.cpp -- see engine.h.  Host orchestration only; every number is produced by kernels.hip.
#include "engine.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

namespace miba {

namespace {
constexpr double kEpsilonSq = 1E-12 * 1E-12; // SBA_EPSILON_SQ, sba_levmar.c:37-38
constexpr double kOneThird = 0.3333333334;   // SBA_ONE_THIRD, sba_levmar.c:40
constexpr int kBuildTimedEvery = 4;          // mi_ba_set_timing_detail(1): every fourth build is bracketed by HIP events (both buffers of the double-buffered pair in turn)
constexpr double kPcgEllMaxFill = 1.25;        // the CG's matrix at a fixed row pitch while padding every block row to the longest costs at most this factor
constexpr size_t kPcgF32FromBytes = 64u << 20;   // the scaled matrix goes single precision above this size (pcg_f32_)
constexpr int kPcgChunk = 8;                 // PCG iterations launched between two host checks
constexpr int kPcgRowPerWorkgroupMax = 512; // up to this many cameras a block row of S~ gets a whole workgroup (4 wavefronts), above: a wavefront
constexpr int kPcgTotalsFrom = 1024;          // workgroups of the CG product from which the dot products are summed by a launch of their own (k_pcg_totals)
constexpr int kPcgTwoWavesPerRowBelow = 128; // blocks per block row of S~ on average below which two wavefronts share a row in the CG product (kernels.hip) ...
constexpr int kPcgTwoWavesPerRowUpTo = 512;  // ... while the launch is a single round of workgroups (5 000 cameras, 1 250 workgroups: 3.06 ms per solve with two, 2.34 with one)
constexpr int kTwoLevelMinCams = 400;       // the coarse space is considered from this many cameras on (fewer: the CG needs few iterations anyway)
constexpr int kTwoLevelFrom = 12;           // ... and built from the first trial after a solve that took this many iterations (per trial it
                                            // costs ~35 us: coarse matrix, its inverse, the band S~Z~)

// Every public entry point runs with the engine's device current and puts the caller's back afterwards: buffers are allocated
// and copied on "the current device" of the calling THREAD, which another engine, another library or another thread of the host
// program may have changed since the constructor ran (ParallelBA with device != 0, the sharded C ABI used in-process).
struct DeviceGuard {
    int prev = -1, want;
    hipError_t status = hipSuccess;             // of the switch to the engine's device: entry points that allocate or create check it
    explicit DeviceGuard(int dev) : want(dev)   // dev < 0: nothing to do (an engine that never opened its device)
    {
        if (want < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) status = hipSetDevice(want);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// a host thread that is joined on every way out of the scope that started it (an exception from the planner included)
struct JoinOnExit {
    std::thread &t;
    ~JoinOnExit() { if (t.joinable()) t.join(); }
};

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
} // namespace

bool Engine::hipok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

Engine::Engine(const mi_ba_config &c) : cfg(c)
{
    if (cfg.undistort < 0 || cfg.undistort > 2) { err = "undistort must be 0, 1 (projection distortion) or 2 (measurement distortion)"; return; }
    if (cfg.nk != 0 && cfg.nk != 1 && cfg.nk != 2) { err = "nk must be 1 (k1) or 2 (k1 and k2)"; return; }
    if (cfg.lm_variant < 0 || cfg.lm_variant > 2) { err = "lm_variant must be 0 (sba), 1 (PBA) or 2 (PBA, identity damping)"; return; }
    if (cfg.nk == 0) cfg.nk = 1;
    lm_variant_cfg_ = cfg.lm_variant;
    cnp = 6 + (cfg.est_focal ? 1 : 0) + (cfg.undistort == 1 ? cfg.nk : cfg.undistort == 2 ? 1 : 0);
    split_ = cnp == 9;       // Bundler's full camera (sfm.c:648-649): blocks wider than the 8 x 8 the camera-side kernels are shaped for
    if (cfg.mode < 0 || cfg.mode > 2) { err = "bad bundle mode"; return; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        err = std::string("no HIP device: ") + (e == hipSuccess ? "device count is 0" : hipGetErrorString(e)) +
              " (the engine has no CPU fallback)";
        return;
    }
    if (cfg.device < 0 || cfg.device >= ndev) { err = "bad device ordinal"; return; }
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return;
    if (!hipok(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking), "hipStreamCreate")) return;
    for (auto &ev : ev_) if (!hipok(hipEventCreate(&ev), "hipEventCreate")) return;
    for (auto &ev : evx_) if (!hipok(hipEventCreate(&ev), "hipEventCreate")) return;
    speculate_ = std::getenv("MI_BA_NO_SPECULATION") == nullptr;
    if (!hipok(hipHostMalloc((void **)&h_scal_, sizeof(double) * 1024, hipHostMallocDefault), "hipHostMalloc")) return;
    created_ = true;
}

Engine::~Engine()
{
    DeviceGuard on_device(st_ ? cfg.device : -1);
    if (d_dbg_.p) { // diagnostic build: per-phase cycle sums of k_build over the engine's life
        unsigned long long h[8];
        if (hipMemcpy(h, d_dbg_.p, 64, hipMemcpyDeviceToHost) == hipSuccess) {
            double tot = 0; for (auto v : h) tot += (double)v;
            std::fprintf(stderr, "[mi_ba stamps] k_build cycle shares: prologue %.3f | ph1 %.3f | ph2 %.3f | ph3 %.3f | ph4 %.3f | ph5 %.3f | flush %.3f | ph1 loads issued %.3f (total %.3g)\n",
                         h[0] / tot, h[1] / tot, h[2] / tot, h[3] / tot, h[4] / tot, h[5] / tot, h[6] / tot, h[7] / tot, tot);
        }
    }
    if (d_pstamps_.p) { // diagnostic build: where the launches of the last reduced solve spent their time (100 MHz wall clock: 10 ns ticks)
        std::vector<unsigned long long> h(d_pstamps_.n);
        if (hipMemcpy(h.data(), d_pstamps_.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            for (int w = 0; w < 2; ++w) {
                double seg[5] = { 0, 0, 0, 0, 0 }; int cnt = 0;
                for (int it = 2; it + 1 < cfg.pcg_maxit && h[(size_t)(it + 1) * 16 + 8 * w + 4]; ++it) {
                    const unsigned long long *t = &h[(size_t)it * 16 + 8 * w], *tn = &h[(size_t)(it + 1) * 16 + 8 * w];
                    if (!t[4] || tn[0] < t[4]) break;
                    for (int i = 0; i < 4; ++i) seg[i] += (double)(t[i + 1] - t[i]);
                    seg[4] += (double)(tn[0] - t[4]); ++cnt;
                }
                if (cnt) std::fprintf(stderr, "[mi_ba stamps] k_pcg_iter %s workgroup, %d launches, us: flags %.2f | product %.2f | scalars+update %.2f | publish %.2f | exit -> next entry %.2f\n",
                                      w ? "last" : "first", cnt, seg[0] / cnt / 100, seg[1] / cnt / 100, seg[2] / cnt / 100, seg[3] / cnt / 100, seg[4] / cnt / 100);
            }
        }
    }
    if (comm_) ncclCommDestroy((ncclComm_t)comm_);
    if (h_scal_) (void)hipHostFree(h_scal_);
    for (auto &ev : ev_) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : evx_) if (ev) (void)hipEventDestroy(ev);
    if (st_) (void)hipStreamDestroy(st_);
}

ModelCfg Engine::model() const
{
    ModelCfg mc;
    mc.explicit_centers = cfg.explicit_centers; mc.est_focal = cfg.est_focal; mc.undistort = cfg.undistort;
    mc.mcon = cfg.mcon; mc.points_fixed = cfg.mode == 1; mc.proj_sign = cfg.proj_sign; mc.f_scale = cfg.f_scale; mc.k_scale = cfg.k_scale;
    mc.inv_f_scale = 1.0 / cfg.f_scale; mc.inv_k_scale = 1.0 / cfg.k_scale;
    mc.nk = cfg.undistort == 1 ? cfg.nk : 1; mc.split = split_ ? 1 : 0;
    mc.pba_update = lm_variant_cfg_ >= 1 ? 1 : 0;   // PBA's controller comes with PBA's UpdateCamera (f clamp, R <- exp(d omega) R)
    mc.damp_diag = cfg.lm_variant == 1;         // (2: PBA's controller with identity damping, __lm_use_diagonal_damp off)
    return mc;
}

int Engine::set_cameras(int m, const double *a0, const double *R0, const double *f_fixed)
{
    if (m <= 0 || !a0 || !R0) { err = "set_cameras: bad arguments"; return MI_BA_ERROR; }
    if (!cfg.est_focal && !f_fixed) { err = "set_cameras: f_fixed is required when the focal length is not estimated"; return MI_BA_ERROR; }
    m_ = m; mv_ = split_ ? 2 * m : m;
    h_a0_.assign(a0, a0 + (size_t)m * cnp);
    h_R0_.assign(R0, R0 + (size_t)m * 9);
    have_ff_ = f_fixed != nullptr;
    if (have_ff_) h_ff_.assign(f_fixed, f_fixed + m); else h_ff_.assign(m, 1.0);
    prepared_ = begun_ = false;
    return 0;
}

int Engine::set_fixed_cameras(const unsigned char *mask)
{
    if (!m_) { err = "set_fixed_cameras: set_cameras first"; return MI_BA_ERROR; }
    if (mask) h_fixed_.assign(mask, mask + m_); else h_fixed_.assign(m_, 0);
    prepared_ = begun_ = false;
    return 0;
}

int Engine::set_fixed_params(const unsigned char *mask)
{
    if (!m_) { err = "set_fixed_params: set_cameras first"; return MI_BA_ERROR; }
    h_pfix_.assign(m_, 0);
    if (mask)
        for (int j = 0; j < m_; ++j)
            for (int r = 0; r < cnp; ++r)
                if (mask[(size_t)j * cnp + r]) h_pfix_[j] |= (unsigned short)(1u << r);
    prepared_ = begun_ = false;
    return 0;
}

int Engine::set_camera_constraints(const double *weights, const double *values)
{
    if (!m_) { err = "set_camera_constraints: set_cameras first"; return MI_BA_ERROR; }
    if (weights && !values) { err = "set_camera_constraints: values missing"; return MI_BA_ERROR; }
    if (weights) { h_cam_cw_.assign(weights, weights + (size_t)m_ * cnp); h_cam_cv_.assign(values, values + (size_t)m_ * cnp); }
    else { h_cam_cw_.clear(); h_cam_cv_.clear(); }
    prepared_ = begun_ = false;
    return 0;
}

int Engine::set_camera_links(int nlink, const int *cam_a, const int *cam_b, const double *weights)
{
    if (!m_) { err = "set_camera_links: set_cameras first"; return MI_BA_ERROR; }
    h_link_cam_.clear(); h_link_w_.clear();
    prepared_ = begun_ = false;
    if (nlink <= 0) return 0;
    if (!cam_a || !cam_b || !weights) { err = "set_camera_links: null argument"; return MI_BA_ERROR; }
    for (int q = 0; q < nlink; ++q) {
        if (cam_a[q] < 0 || cam_a[q] >= m_ || cam_b[q] < 0 || cam_b[q] >= m_ || cam_a[q] == cam_b[q]) {
            h_link_cam_.clear(); h_link_w_.clear();
            err = "set_camera_links: bad camera index"; return MI_BA_ERROR;
        }
        h_link_cam_.push_back(cam_a[q]); h_link_cam_.push_back(cam_b[q]);
    }
    h_link_w_.assign(weights, weights + (size_t)nlink * cnp);
    return 0;
}

int Engine::set_point_constraints(const double *weight, const double *values)
{
    if (!n_) { err = "set_point_constraints: set_points first"; return MI_BA_ERROR; }
    if (weight && !values) { err = "set_point_constraints: values missing"; return MI_BA_ERROR; }
    if (weight) { h_pt_cw_.assign(weight, weight + n_); h_pt_cv_.assign(values, values + (size_t)n_ * 3); }
    else { h_pt_cw_.clear(); h_pt_cv_.clear(); }
    prepared_ = begun_ = false;
    return 0;
}

int Engine::set_points(int n, const double *xyz)
{
    if (n <= 0 || !xyz) { err = "set_points: bad arguments"; return MI_BA_ERROR; }
    n_ = n;
    h_xyz_.assign(xyz, xyz + (size_t)n * 3);
    prepared_ = begun_ = false;
    return 0;
}

int Engine::set_observations(int64_t nobs, const double *xy, const int *pt_idx, const int *cam_idx)
{
    if (nobs <= 0 || !xy || !pt_idx || !cam_idx) { err = "set_observations: bad arguments"; return MI_BA_ERROR; }
    nobs_ = nobs;
    h_xy_.assign(xy, xy + (size_t)nobs * 2);
    h_pt_.assign(pt_idx, pt_idx + nobs);
    h_cam_.assign(cam_idx, cam_idx + nobs);
    prepared_ = begun_ = false;
    return 0;
}

// what RCCL reports about the communicator (include/mi_ba.h)
int Engine::comm_info(int out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!comm_) { out[0] = host_ar_ ? cfg.world : 1; out[1] = cfg.device; out[2] = cfg.rank; return 0; }
    int n = 0, dev = -1, rk = -1;
    if (ncclCommCount((ncclComm_t)comm_, &n) != ncclSuccess || ncclCommCuDevice((ncclComm_t)comm_, &dev) != ncclSuccess ||
        ncclCommUserRank((ncclComm_t)comm_, &rk) != ncclSuccess) { err = "ncclCommCount / CuDevice / UserRank failed"; return MI_BA_ERROR; }
    out[0] = n; out[1] = dev; out[2] = rk; out[3] = 1;
    return 0;
}

int Engine::comm_init(const char *id)
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (cfg.world <= 1) return 0;
    ncclUniqueId uid;
    static_assert(sizeof(uid) <= MI_BA_UNIQUE_ID_BYTES, "unique id size");
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t c;
    ncclResult_t r = ncclCommInitRank(&c, cfg.world, uid, cfg.rank);
    if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return MI_BA_ERROR; }
    comm_ = c;
    return 0;
}

int Engine::prepare()
{
    DeviceGuard on_device(cfg.device);
    if (!created_) return MI_BA_ERROR;
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!m_ || !n_ || !nobs_) { err = "prepare: cameras, points and observations must be set first"; return MI_BA_ERROR; }
    if (cfg.mcon < 0 || cfg.mcon > m_) { err = "prepare: bad mcon"; return MI_BA_ERROR; }
    if (cfg.world > 1 && !comm_ && !host_ar_) { err = "prepare: world > 1 needs mi_ba_comm_init first"; return MI_BA_ERROR; }
    if (cfg.world == 1 && !comm_ && std::getenv("MI_BA_FORCE_COMM")) {
        // test hook: run the RCCL exchange on a one-rank communicator, so the collective calls (buffers,
        // counts, stream) are exercised on a single-GPU box
        ncclUniqueId uid;
        ncclComm_t c;
        if (ncclGetUniqueId(&uid) != ncclSuccess || ncclCommInitRank(&c, 1, uid, 0) != ncclSuccess) {
            err = "prepare: MI_BA_FORCE_COMM: cannot create a one-rank RCCL communicator";
            return MI_BA_ERROR;
        }
        comm_ = c;
    }
    // sba_levmar.c:645-650: fewer measurements than unknowns is an error
    if (2 * nobs_ < (cfg.mode == 1 ? 0 : 3 * (int64_t)n_) + (cfg.mode == 2 ? 0 : (int64_t)m_ * cnp)) {
        err = "cannot solve a problem with fewer measurements than unknowns";
        return MI_BA_ERROR;
    }
    if (!h_link_cam_.empty() && cfg.mode == 2) { err = "prepare: camera links make no sense in a structure-only bundle"; return MI_BA_ERROR; }
    const bool plan_timing = std::getenv("MI_BA_PLAN_TIMING") != nullptr;
    double t_stage = now_ms();
    auto stage = [&](const char *what) {
        if (!plan_timing) return;
        (void)hipDeviceSynchronize();
        const double t = now_ms();
        std::fprintf(stderr, "[mi_ba prepare] %-26s %8.1f ms\n", what, t - t_stage);
        t_stage = t;
    };
    // the observation streams are final long before the planner is done (plan.h shard_arrays_ready): a second host thread gathers
    // the measurements into the plan's order and uploads them with the two index streams while the tiling runs
    std::thread early_upload;
    JoinOnExit join_early_upload{early_upload};
    hipError_t early_err = hipSuccess;
    auto start_early_upload = [&]() {
        early_upload = std::thread([&]() {
            if ((early_err = hipSetDevice(cfg.device)) != hipSuccess) return;
            const Plan &q = plan_;
            if ((early_err = d_obs_cam_.upload(q.obs_cam)) != hipSuccess || (early_err = d_obs_pt_.upload(q.obs_pt)) != hipSuccess) return;
            const int64_t nol = q.nobs_local();
            std::vector<double2> xy(nol);
            const int nt = nol < (1 << 18) ? 1 : (int)std::min<unsigned>(4u, std::max(1u, std::thread::hardware_concurrency() / 2));
            std::vector<std::thread> th;
            for (int k = 0; k < nt; ++k)
                th.emplace_back([&, k]() {
                    for (int64_t o = nol * k / nt, e = nol * (k + 1) / nt; o < e; ++o) { xy[o].x = h_xy_[2 * q.obs_src[o]]; xy[o].y = h_xy_[2 * q.obs_src[o] + 1]; }
                });
            for (auto &t : th) t.join();
            early_err = d_obs_xy_.upload(xy);
        });
    };
    std::string perr;
    try {
        perr = build_plan(plan_, n_, m_, nobs_, h_pt_.data(), h_cam_.data(), cfg.rank, cfg.world, cfg.tile_max_batches,
                          h_link_cam_.empty() ? nullptr : &h_link_cam_, cfg.deterministic == 2, split_, start_early_upload);
    } catch (const std::exception &ex) { perr = std::string("out of host memory or worse (") + ex.what() + ")"; }
    if (early_upload.joinable()) early_upload.join();
    if (perr.empty() && !hipok(early_err, "upload of the observation streams")) return MI_BA_ERROR;
    if (!perr.empty()) { err = "plan: " + perr; return MI_BA_ERROR; }
    const Plan &pl = plan_;
#define UP(buf, vec) if (!hipok((buf).upload(vec), "upload " #vec)) return MI_BA_ERROR
    UP(d_tile_batch_, pl.tile_batch); UP(d_tile_slot_, pl.tile_slot); UP(d_tile_cslot_, pl.tile_cslot); UP(d_tile_mode_, pl.tile_mode);
    uni_fixed_ = uni_other_ = false;      // which instantiations of k_build_uni have tiles: counted from the tables the kernels read
    for (int t = 0; t < pl.ntile_lds; ++t)
        if (pl.tile_mode[t] == 4) { if (pl.tile_cslot[t + 1] - pl.tile_cslot[t] == kUniFixedViews) uni_fixed_ = true; else uni_other_ = true; }
    UP(d_slot_gblk_, pl.slot_gblk); UP(d_cslot_cam_, pl.cslot_cam);
    UP(d_batch_obs_, pl.batch_obs); UP(d_batch_pt_, pl.batch_pt); UP(d_pt_obs_, pl.pt_obs);
    UP(d_bs_batch_obs_, pl.bs_batch_obs); UP(d_bs_batch_pt_, pl.bs_batch_pt); UP(d_bs_batch_obs_end_, pl.bs_batch_obs_end); UP(d_bs_batch_pt_end_, pl.bs_batch_pt_end);
    UP(d_obs_cslot_, pl.obs_cslot);                              // (obs_cam, obs_pt, obs_xy: the early upload above)
    UP(d_pair_, pl.pair);
    UP(d_obs_rec_, pl.obs_rec); UP(d_rec_cam_, pl.rec_cam); UP(d_wk_blk_, pl.wk_blk); UP(d_rpair_, pl.rpair);
    UP(d_wk_diag_cam_, pl.wk_diag_cam); UP(d_wk_diag_blk_, pl.wk_diag_blk);
    UP(d_wk_vblk_, pl.wk_vblk); UP(d_wk_diag_vblk_, pl.wk_diag_vblk);
    if (!hipok(d_rec_.alloc((size_t)pl.rec_cam[m_] * (split_ ? kRec9 : kRec)), "alloc observation records")) return MI_BA_ERROR;
    UP(d_bsr_row_, pl.bsr_row); UP(d_bsr_col_, pl.bsr_col); UP(d_bsr_src_, pl.bsr_src); UP(d_bsr_trans_, pl.bsr_trans);
    {
        std::vector<long long> bp(pl.batch_pair.begin(), pl.batch_pair.end()), wq(pl.wk_q.begin(), pl.wk_q.end());
        UP(d_batch_pair_, bp); UP(d_wk_q_, wq);
        std::vector<long long> sq(pl.sb_q.begin(), pl.sb_q.end());
        UP(d_sb_q_, sq); UP(d_sb_blk_, pl.sb_blk); UP(d_sb_ent_, pl.sb_ent);
        std::vector<long long> sq8(pl.sb8_q.begin(), pl.sb8_q.end());
        UP(d_sb8_q_, sq8); UP(d_sb8_blk_, pl.sb8_blk); UP(d_sb8_ent_, pl.sb8_ent);
        std::vector<int> rowidx(pl.nb_full()), diag(mv_, -1);
        for (int j = 0; j < mv_; ++j)
            for (int q = pl.bsr_row[j]; q < pl.bsr_row[j + 1]; ++q) { rowidx[q] = j; if (pl.bsr_col[q] == j) diag[j] = q; }
        UP(d_bsr_rowidx_, rowidx); UP(d_bsr_diag_, diag);
        // where every block's transpose sits (the pattern is symmetric): the mailbox a block row pushes its vectors to (kernels.hip push_row)
        std::vector<int> tq(pl.nb_full(), -1);
        bool sym = true;
        for (int j = 0; j < mv_ && sym; ++j)
            for (int q = pl.bsr_row[j]; q < pl.bsr_row[j + 1]; ++q) {
                const int k = pl.bsr_col[q];
                const int *b = pl.bsr_col.data() + pl.bsr_row[k], *e = pl.bsr_col.data() + pl.bsr_row[k + 1];
                const int *f = std::lower_bound(b, e, j);
                if (f == e || *f != j) f = std::find(b, e, j);          // (columns not ascending: look through the row)
                if (f == e) { sym = false; break; }
                tq[q] = (int)(f - pl.bsr_col.data());
            }
        if (!sym) { err = "internal: the pattern of S is not symmetric"; return MI_BA_ERROR; }
        // (opt-in, MI_BA_PCG_PUSH=1: measured SLOWER than the gather by column index -- C3 0.38 against 0.26 ms per solve, C2 0.208 / 0.202, C1 0.066 / 0.056:
        //  the mailboxes are 37 x the vectors' bytes, written and read across the XCDs every iteration)
        {   // fixed-pitch copy of the pattern for the CG (PcgArgs.ell_w): every block row padded to the longest one, the pads zero blocks in the own column
            int W = 0;
            for (int j = 0; j < mv_; ++j) W = std::max(W, pl.bsr_row[j + 1] - pl.bsr_row[j]);
            const char *env = std::getenv("MI_BA_PCG_ELL");
            pcg_ell_w_ = (env ? std::atoi(env) != 0 : true) && W > 0 && (double)mv_ * W <= kPcgEllMaxFill * pl.nb_full() ? W : 0;
            if (pcg_ell_w_) {
                std::vector<int> ecol((size_t)mv_ * W);
                for (int j = 0; j < mv_; ++j)
                    for (int t = 0; t < W; ++t) {
                        const int q = pl.bsr_row[j] + t;
                        const bool real = q < pl.bsr_row[j + 1];
                        ecol[(size_t)j * W + t] = real ? pl.bsr_col[q] : j;
                    }
                UP(d_ell_col_, ecol);
            }
            // k_scale_blocks' work list: the blocks on and above the diagonal with everything the kernel needs to know about them and their mirror blocks
            std::vector<int> sl;
            sl.reserve((size_t)pl.nb_up() * 8);
            for (int j = 0; j < mv_; ++j)
                for (int q = pl.bsr_row[j]; q < pl.bsr_row[j + 1]; ++q) {
                    const int k = pl.bsr_col[q];
                    if (k < j) continue;
                    const int bm = k == j ? -1 : tq[q];
                    const int pe = pcg_ell_w_ ? j * W + (q - pl.bsr_row[j]) : q, pme = bm < 0 ? -1 : pcg_ell_w_ ? k * W + (bm - pl.bsr_row[k]) : bm;
                    const int e[8] = { j, k, pl.bsr_src[q], q, bm, pe, pme, 0 };
                    sl.insert(sl.end(), e, e + 8);
                }
            UP(d_scale_list_, sl);
        }
        pcg_push_ok_ = sym && (size_t)pl.nb_full() * 64 * sizeof(double) <= kPcgF32FromBytes && std::getenv("MI_BA_PCG_PUSH") && std::atoi(std::getenv("MI_BA_PCG_PUSH")) != 0;
        if (pcg_push_ok_) UP(d_bsr_tq_, tq);
    }
    stage("plan + index uploads");
    if (cfg.deterministic == 1 && pl.ntile_lds > 0) {   // ordered hand-over: who contributes to every block of S and to every camera, in tile order
        std::vector<int> bp((size_t)pl.nb_up() + 1, 0), cp((size_t)mv_ + 1, 0);
        for (int t = 0; t < pl.ntile_lds; ++t) {
            for (int q = pl.tile_slot[t]; q < pl.tile_slot[t + 1]; ++q) bp[(size_t)pl.slot_gblk[q] + 1]++;
            for (int q = pl.tile_cslot[t]; q < pl.tile_cslot[t + 1]; ++q) cp[(size_t)pl.cslot_cam[q] + 1]++;
        }
        for (size_t i = 0; i + 1 < bp.size(); ++i) bp[i + 1] += bp[i];
        for (size_t i = 0; i + 1 < cp.size(); ++i) cp[i + 1] += cp[i];
        std::vector<int> bl((size_t)bp.back()), cl((size_t)cp.back()), bf(bp.begin(), bp.end() - 1), cf(cp.begin(), cp.end() - 1);
        for (int t = 0; t < pl.ntile_lds; ++t) {
            for (int q = pl.tile_slot[t]; q < pl.tile_slot[t + 1]; ++q) bl[(size_t)bf[pl.slot_gblk[q]]++] = t * kNSlot + (q - pl.tile_slot[t]);
            for (int q = pl.tile_cslot[t]; q < pl.tile_cslot[t + 1]; ++q) cl[(size_t)cf[pl.cslot_cam[q]]++] = t * kNCSlot + (q - pl.tile_cslot[t]);
        }
        UP(d_tb_ptr_, bp); UP(d_tb_list_, bl); UP(d_tc_ptr_, cp); UP(d_tc_list_, cl);
        if (!hipok(d_tile_blk_.alloc((size_t)pl.ntile_lds * kNSlot * cnp_cam_() * cnp_cam_()), "alloc per-tile blocks") ||
            !hipok(d_tile_cam_.alloc((size_t)pl.ntile_lds * kNCSlot * 3 * kCP), "alloc per-tile camera vectors")) return MI_BA_ERROR;
    }
    {   // two-level preconditioner: aggregates of consecutive block rows (whole workgroups of four), their coarse blocks, the band slots
        co_ = CoarseArgs{};
        co_on_ = co_active_ = false;
        const char *ev = std::getenv("MI_BA_TWO_LEVEL");
        co_force_ = ev ? std::atoi(ev) : 0;      // opt-in (fewer iterations, no faster on the bench scenes)
        const int nwg4 = (mv_ + 3) / 4;
        bool want = co_force_ != 0 && !split_ && cfg.mode != 2 && nwg4 >= 2 && (co_force_ == 1 || mv_ >= kTwoLevelMinCams);
        if (want) {
            const int wpa = (nwg4 + kCoarseMax - 1) / kCoarseMax, ng = (nwg4 + wpa - 1) / wpa, cpa = wpa * 4;
            std::vector<int> bc((size_t)mv_ * kBcSlots, -1);
            for (int j = 0; j < mv_ && want; ++j) {
                int used = 0;
                auto add = [&](int H) {
                    for (int t = 0; t < used; ++t) if (bc[(size_t)j * kBcSlots + t] == H) return true;
                    if (used == kBcSlots) return false;
                    bc[(size_t)j * kBcSlots + used++] = H;
                    return true;
                };
                for (int q = pl.bsr_row[j]; q < pl.bsr_row[j + 1] && want; ++q) want = add(pl.bsr_col[q] / cpa);
            }
            if (want) {
                UP(d_bc_agg_, bc);
                const int nc = 8 * ng;
                if (!hipok(d_Ac_.alloc((size_t)nc * nc), "alloc coarse matrix") || !hipok(d_Acinv_.alloc((size_t)nc * nc), "alloc coarse inverse") ||
                    !hipok(d_Bc_.alloc((size_t)mv_ * kBcSlots * 64), "alloc coarse band") || !hipok(d_U_.alloc((size_t)mv_ * kBcSlots * 64), "alloc coarse row sums") || !hipok(d_Lfac_.alloc((size_t)mv_ * 64), "alloc L") ||
                    !hipok(d_z_.alloc((size_t)mv_ * kCP), "alloc z") || !hipok(d_cpart_.alloc((size_t)2 * 3 * nwg4 * 8), "alloc coarse partials"))
                    return MI_BA_ERROR;
                co_.ng = ng; co_.nc = nc; co_.wg_per_agg = wpa;
                co_.bc_agg = d_bc_agg_.p; co_.U = d_U_.p;
                co_.Ac = d_Ac_.p; co_.Acinv = d_Acinv_.p; co_.Bc = d_Bc_.p; co_.Lfac = d_Lfac_.p; co_.z = d_z_.p; co_.cpart = d_cpart_.p;
            }
        }
    }
    UP(d_R0_[0], h_R0_); UP(d_R0_[1], h_R0_); UP(d_R0_init_, h_R0_); UP(d_ff_, h_ff_);
    {   // soft constraints: camera side always present (zeros = none), point side only when set
        if (!h_pt_cw_.empty() && cfg.mode != 0) { err = "prepare: point constraints need the full bundle (mode 0)"; return MI_BA_ERROR; }
        if (!h_cam_cw_.empty() && cfg.mode == 2) { err = "prepare: camera constraints make no sense in a structure-only bundle"; return MI_BA_ERROR; }
        std::vector<double> cw((size_t)mv_ * kCP, 0.0), cv((size_t)mv_ * kCP, 0.0);
        if (!h_cam_cw_.empty())
            for (int j = 0; j < m_; ++j)
                for (int r = 0; r < cnp; ++r) { cw[slot_(j, r)] = h_cam_cw_[(size_t)j * cnp + r]; cv[slot_(j, r)] = h_cam_cv_[(size_t)j * cnp + r]; }
        UP(d_cam_cw_, cw); UP(d_cam_cv_, cv);
        d_pt_cw_.release(); d_pt_cv_.release();
        if (!h_pt_cw_.empty()) {
            const size_t npl_ = (size_t)pl.npts_local();
            std::vector<double> pw(npl_), pv(npl_ * 3);
            for (size_t t = 0; t < npl_; ++t) {
                const size_t i = (size_t)pl.perm[pl.pt_begin + t];
                pw[t] = h_pt_cw_[i] * (double)nobs_;                 // "nvis * weight", sba_levmar.c:1026
                for (int r = 0; r < 3; ++r) pv[3 * t + r] = h_pt_cv_[3 * i + r];
            }
            UP(d_pt_cw_, pw); UP(d_pt_cv_, pv);
        }
    }
    links_ = CamLinks{};
    if (!h_link_cam_.empty()) { // camera links: CSR over cameras (each link at both ends), summed weight per linked pair
        const int nl = (int)h_link_cam_.size() / 2;
        // (links couple parameters 0-7: with nine parameters they sit on the even rows, camera j = row 2j)
        const int vs = split_ ? 2 : 1;
        std::vector<int> row(mv_ + 1, 0), other(2 * (size_t)nl), fill;
        std::vector<double> w(2 * (size_t)nl * kCP, 0.0), pw;
        for (int q = 0; q < 2 * nl; ++q) row[vs * h_link_cam_[q] + 1]++;
        for (int j = 0; j < mv_; ++j) row[j + 1] += row[j];
        fill.assign(row.begin(), row.end() - 1);
        std::unordered_map<uint64_t, int> pairs;
        for (int q = 0; q < nl; ++q) {
            const int a = vs * h_link_cam_[2 * q], b = vs * h_link_cam_[2 * q + 1];
            const int ea = fill[a]++, eb = fill[b]++;
            other[ea] = b; other[eb] = a;
            const uint64_t key = ((uint64_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b);
            auto it = pairs.find(key);
            if (it == pairs.end()) { it = pairs.emplace(key, (int)pairs.size()).first; pw.resize(pairs.size() * kCP, 0.0); }
            for (int r = 0; r < cnp_cam_(); ++r) {
                const double wr = h_link_w_[(size_t)q * cnp + r];
                if (!(wr >= 0.0)) { err = "prepare: camera link weights must be >= 0"; return MI_BA_ERROR; }
                w[(size_t)ea * kCP + r] = wr; w[(size_t)eb * kCP + r] = wr;
                pw[(size_t)it->second * kCP + r] += wr;
            }
        }
        std::vector<int> blk(pl.nb_full(), -1);
        for (int j = 0; j < mv_; ++j)
            for (int q = pl.bsr_row[j]; q < pl.bsr_row[j + 1]; ++q) {
                const int k = pl.bsr_col[q];
                if (k == j) continue;
                auto it = pairs.find(((uint64_t)std::min(j, k) << 32) | (uint32_t)std::max(j, k));
                if (it != pairs.end()) blk[q] = it->second;
            }
        UP(d_link_row_, row); UP(d_link_other_, other); UP(d_link_w_, w); UP(d_link_blk_, blk); UP(d_link_pair_w_, pw);
        links_.row = d_link_row_.p; links_.other = d_link_other_.p; links_.w = d_link_w_.p;
        links_.blk_pair = d_link_blk_.p; links_.pair_w = d_link_pair_w_.p;
    }
    UP(d_long_pts_, pl.long_pts);
    {
        if ((int)h_fixed_.size() != m_) h_fixed_.assign(m_, 0);
        std::vector<unsigned char> fx(h_fixed_);
        for (int j = 0; j < cfg.mcon && j < m_; ++j) fx[j] = 1; // sba's leading mcon cameras
        if (cfg.mode == 2) std::fill(fx.begin(), fx.end(), (unsigned char)1); // structure-only: A_ij := 0 everywhere
        UP(d_fixed_real_, fx);
        h_fixed_eff_ = fx;
        if ((int)h_pfix_.size() != m_) h_pfix_.assign(m_, 0);
        std::vector<unsigned char> fxv(mv_), pfv(mv_);
        for (int j = 0; j < m_; ++j) {
            if (!split_) { fxv[j] = fx[j]; pfv[j] = (unsigned char)(h_pfix_[j] & 0xFF); continue; }
            fxv[2 * j] = fxv[2 * j + 1] = fx[j];
            pfv[2 * j] = (unsigned char)(h_pfix_[j] & 0xFF);
            pfv[2 * j + 1] = (unsigned char)(0xFE | ((h_pfix_[j] >> 8) & 1));     // row 2j + 1: parameter 8 in column 0, the other columns are padding
        }
        UP(d_fixed_, fxv); UP(d_pfix_, pfv);
        nfree_cams_ = 0;
        for (unsigned char f : fx) nfree_cams_ += f ? 0 : 1;
    }
#undef UP
    const size_t npl = (size_t)pl.npts_local();
    for (int b = 0; b < 2; ++b) {
        if (!hipok(d_cam_a_[b].alloc((size_t)mv_ * kCP), "alloc cam_a")) return MI_BA_ERROR;
        if (!hipok(d_cam_lin_[b].alloc((size_t)m_ * kCamLin), "alloc cam_lin")) return MI_BA_ERROR;
        if (!hipok(d_pts_[b].alloc(npl * 3), "alloc pts")) return MI_BA_ERROR;
    }
    const size_t acc_n = 2 * nslot_() + (size_t)pl.nb_up() * 64 + (size_t)3 * mv_ * kCP + kScalCount;
    if (!hipok(d_ptout_.alloc(npl * kPtOut), "alloc ptout") || !hipok(d_acc_.alloc(acc_n), "alloc acc") ||
        !hipok(d_S_full_.alloc((size_t)pl.nb_full() * 64), "alloc S") || !hipok(d_Linv_.alloc((size_t)mv_ * 64), "alloc Linv") || !hipok(d_S_pcg_.alloc(std::max((size_t)pl.nb_full(), (size_t)mv_ * pcg_ell_w_) * 64), "alloc S~") ||
        !hipok(d_E_.alloc((size_t)mv_ * kCP), "alloc E") || !hipok(d_dA_.alloc((size_t)mv_ * kCP), "alloc dA") ||
        !hipok(d_dB_.alloc(npl * 3), "alloc dB") || !hipok(d_pcg_.alloc((size_t)mv_ * kCP * 8), "alloc pcg") ||
        !hipok(d_rzpq_.alloc(2 * (size_t)(cfg.pcg_maxit + 4) + 4 * (size_t)mv_), "alloc rz/pq") ||
        // partial rows: the build's (a row per tile, per long track), then -- from bs_row0_() on: both sets are summed together at the trial's end -- the
        // back-substitution's (k_backsub's grid | the long tracks | the uniform tiles)
        !hipok(d_part_.alloc((size_t)(2 * (pl.ntile() + pl.long_pts.size()) + kBacksubGridMax + pl.ntile() + 2) * kPartW), "alloc partials") ||
        !hipok(d_stage_.alloc((size_t)2 * kReduceStage * kPartW), "alloc reduce stage") ||
        !hipok(d_cam_part_.alloc((size_t)(mv_ + 1) * kPartW), "alloc camera partials") ||
        !hipok(d_cam_part2_.alloc((size_t)((m_ + 7) / 8 + 1) * kPartW), "alloc camera partials"))
        return MI_BA_ERROR;
    if (cfg.world > 1 && !hipok(d_full_pts_.alloc((size_t)n_ * 3), "alloc full pts")) return MI_BA_ERROR;
    if (!hipok(hipMemset(d_rzpq_.p, 0, d_rzpq_.n * sizeof(double)), "memset rz/pq")) return MI_BA_ERROR;
    if (pcg_push_ok_ && (!hipok(d_mb_.alloc(2 * (size_t)pl.nb_full() * 24), "alloc CG mailboxes") || !hipok(hipMemset(d_mb_.p, 0, d_mb_.n * sizeof(double)), "memset CG mailboxes"))) return MI_BA_ERROR;
    if (pcg_ell_w_ && !hipok(hipMemset(d_S_pcg_.p, 0, d_S_pcg_.n * sizeof(double)), "memset S~")) return MI_BA_ERROR;   // (the pads stay zero: only blocks of the pattern are ever written)
    if (!hipok(d_spec_.alloc(2), "alloc spec") || !hipok(hipMemset(d_spec_.p, 0, 2 * sizeof(double)), "memset spec")) return MI_BA_ERROR;
    if (cfg.pcg_maxit + 4 > 500) { err = "pcg_maxit too large (max 496)"; return MI_BA_ERROR; }
    if (std::getenv("MI_BA_PCG_STAMPS")) {
        if (!hipok(d_pstamps_.alloc((size_t)(cfg.pcg_maxit + 4) * 16), "alloc pcg stamps") || !hipok(hipMemset(d_pstamps_.p, 0, d_pstamps_.n * 8), "memset pcg stamps")) return MI_BA_ERROR;
    }
    if (std::getenv("MI_BA_STAMPS")) {
        if (!hipok(d_dbg_.alloc(8), "alloc dbg") || !hipok(hipMemset(d_dbg_.p, 0, 64), "memset dbg")) return MI_BA_ERROR;
    }
    setup_persist();
    {   // the single-precision matrix of the reduced solve (pcg_f32_): MI_BA_PCG_F32 = 0 / 1 forces, MI_BA_PCG_F32_FROM_MB moves the size threshold (probes, tests)
        const char *f = std::getenv("MI_BA_PCG_F32"), *mb = std::getenv("MI_BA_PCG_F32_FROM_MB");
        pcg_f32_force_ = f ? std::atoi(f) : -1;
        pcg_f32_from_ = mb ? (size_t)std::atoll(mb) << 20 : kPcgF32FromBytes;
    }
    stage("parameters, buffers");
    prepared_ = true;
    if (set_params(h_a0_.data(), h_xyz_.data()) != 0) return MI_BA_ERROR;
    // algorithmic HBM bytes of one linearisation + Schur build, SURVEY.md 8(d) "Pass A" with the f64 state the engine really
    // moves: per observation 2 f64 measurement + 2 i32 indices (24 B), per point 3 f64 in, V*^-1 (6) + h (3) out (96 B), per
    // camera its 192-B record, every upper block of S once (512 B).  The engine's own index tables (pair entries, batch and
    // tile descriptors) and the row path's observation records are NOT problem data: they are reported beside it.
    timing.build_bytes = 24.0 * (double)pl.nobs_local() + 96.0 * (double)npl + 192.0 * m_ + 512.0 * pl.nb_up();
    timing.build_block_handovers = (double)pl.slot_gblk.size();
    timing.build_index_bytes = 1.0 * (double)pl.nobs_local() + 4.0 * (double)pl.pair.size() + 8.0 * (double)(pl.rpair.size() / 2) + 4.0 * (double)(pl.sb_ent.size() + pl.sb8_ent.size()) +
                               2.0 * 8.0 * kRec * (double)pl.rec_cam[m_];
    {   // arithmetic of the same launch (for the f64 vector roofline beside the HBM one)
        double pairs = 0.0;
        for (size_t t = 0; t < npl; ++t) { const double d = pl.pt_obs[t + 1] - pl.pt_obs[t]; pairs += 0.5 * d * (d + 1); }
        timing.build_flops = 400.0 * (double)pl.nobs_local() + 276.0 * pairs + 120.0 * (double)npl;
    }
    if (cfg.verbose)
        std::fprintf(stderr, "[mi_ba] rank %d/%d: %lld pts %lld obs, %d batches, %d tiles (%d LDS), %zu pair entries; row path: %lld pts, %d records, %zu pairs in %zu + %zu blocks, %lld pairs in %zu super-block entries; S: %d upper / %d full blocks\n",
                     cfg.rank, cfg.world, (long long)npl, (long long)pl.nobs_local(), pl.nbatch(), pl.ntile(), pl.ntile_lds,
                     pl.pair.size(), (long long)pl.nrow_pts, pl.rec_cam[m_], pl.rpair.size() / 2, pl.wk_blk.size(), pl.wk_diag_cam.size(), (long long)pl.nsuper_pairs, pl.sb_ent.size() / 8 + pl.sb8_ent.size() / 16, pl.nb_up(), pl.nb_full());
    return 0;
}

// The reduced solve in one cooperative launch (k_pcg_persist), OPT-IN (MI_BA_PCG_PERSIST=1): measured slower than a launch per
// iteration -- C3 12.0 us per iteration against 7.4, C2 8.8 against 6.2 (profiles/r03_persistent_cg_probe.txt): the grid barrier's
// three dependent cross-XCD round trips cost more than a kernel boundary, and the rest of the iteration is the same chain either way.
// Usable when every workgroup's share of S~ -- the blocks of <= 8 consecutive block rows -- fits its LDS and all workgroups are
// resident at once with a margin (a stranded workgroup would stall a barrier until its timeout).  Not with the coarse space (its own
// kernels), not under the host transport (test set-ups put several ranks' cooperative launches on ONE GPU, where none may be fully resident).
void Engine::setup_persist()
{
    persist_ok_ = false;
    const char *ev = std::getenv("MI_BA_PCG_PERSIST");
    if (!(ev && ev[0] == '1')) return;
    if (co_.ng || host_ar_ || cfg.mode == 2 || mv_ <= 0) return;
    const Plan &pl = plan_;
    for (int rows = 4; rows <= 8; ++rows) {
        const int nwg = (mv_ + rows - 1) / rows;
        // per workgroup: its blocks, and the distinct columns they reach (its window of the neighbours' vectors)
        std::vector<int> win_ptr(nwg + 1, 0), win_col, blk_loc(pl.nb_full(), 0), mark(mv_, -1);
        size_t lds = 0;
        for (int b = 0; b < nwg; ++b) {
            const int q0 = pl.bsr_row[b * rows], q1 = pl.bsr_row[std::min(mv_, (b + 1) * rows)];
            std::vector<int> cols(pl.bsr_col.begin() + q0, pl.bsr_col.begin() + q1);
            std::sort(cols.begin(), cols.end());
            cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            for (size_t k = 0; k < cols.size(); ++k) mark[cols[k]] = (int)k;
            for (int q = q0; q < q1; ++q) blk_loc[q] = mark[pl.bsr_col[q]];
            win_col.insert(win_col.end(), cols.begin(), cols.end());
            win_ptr[b + 1] = (int)win_col.size();
            lds = std::max(lds, (size_t)(q1 - q0) * (64 * sizeof(double) + sizeof(int)) + cols.size() * (24 * sizeof(double) + sizeof(int)));
        }
        lds = (lds + 15) & ~(size_t)15;
        if (lds > 150 * 1024) break;                               // more rows per workgroup only need more
        const int cap = pcg_persist_capacity(lds);
        if (cap <= 0 || nwg > cap - cap / 8) continue;             // a margin of an eighth of the chip
        const int group = std::max(8, (nwg + kBarMaxGroups - 1) / kBarMaxGroups);
        if (!hipok(d_bar_.alloc(kBarWords), "alloc barrier words") || !hipok(hipMemset(d_bar_.p, 0, kBarWords * sizeof(unsigned long long)), "memset barrier words") ||
            !hipok(d_win_ptr_.upload(win_ptr), "upload windows") || !hipok(d_win_col_.upload(win_col), "upload windows") || !hipok(d_blk_loc_.upload(blk_loc), "upload windows")) {
            err.clear();               // (an opt-in that could not be set up is simply off: prepare() goes on with the launched iteration)
            return;
        }
        persist_rows_ = rows; persist_nwg_ = nwg; persist_lds_ = lds; persist_group_ = group;
        persist_ok_ = true;
        if (cfg.verbose && cfg.rank == 0)
            std::fprintf(stderr, "[mi_ba] reduced solve: one persistent launch, %d workgroups x %d block rows, %zu KB of LDS each, barrier groups of %d\n", nwg, rows, lds >> 10, group);
        return;
    }
}

int Engine::set_params(const double *a, const double *xyz)
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_) { err = "set_params: prepare first"; return MI_BA_ERROR; }
    std::vector<double> ap((size_t)mv_ * kCP, 0.0);
    for (int j = 0; j < m_; ++j)
        for (int r = 0; r < cnp; ++r) ap[slot_(j, r)] = a[(size_t)j * cnp + r];
    const Plan &pl = plan_;
    const size_t npl = (size_t)pl.npts_local();
    std::vector<double> pp(npl * 3);
    for (size_t t = 0; t < npl; ++t)
        for (int r = 0; r < 3; ++r) pp[3 * t + r] = xyz[3 * (size_t)pl.perm[pl.pt_begin + t] + r];
    pl2_inactive_ = 0.0;
    for (int t = pl.n_active; t < n_; ++t)
        for (int r = 0; r < 3; ++r) { double v = xyz[3 * (size_t)pl.perm[t] + r]; pl2_inactive_ += v * v; }
    if (xyz != h_xyz_.data()) h_xyz_.assign(xyz, xyz + (size_t)n_ * 3);
    cur_ = 0; cam_lin_valid_ = false;
    if (!hipok(hipMemcpy(d_R0_[0].p, d_R0_init_.p, d_R0_init_.n * sizeof(double), hipMemcpyDeviceToDevice), "restore R0")) return MI_BA_ERROR;   // (PBA's update moves it)
    if (!hipok(hipMemcpy(d_cam_a_[0].p, ap.data(), ap.size() * sizeof(double), hipMemcpyHostToDevice), "upload cam_a")) return MI_BA_ERROR;
    if (npl && !hipok(hipMemcpy(d_pts_[0].p, pp.data(), pp.size() * sizeof(double), hipMemcpyHostToDevice), "upload pts")) return MI_BA_ERROR;
    // a device-resident copy of the starting point, for restart()
    if (!hipok(d_cam_a_init_.alloc(ap.size()), "alloc cam_a copy") || !hipok(d_pts_init_.alloc(pp.size()), "alloc pts copy")) return MI_BA_ERROR;
    if (!hipok(hipMemcpy(d_cam_a_init_.p, d_cam_a_[0].p, ap.size() * sizeof(double), hipMemcpyDeviceToDevice), "copy cam_a")) return MI_BA_ERROR;
    if (npl && !hipok(hipMemcpy(d_pts_init_.p, d_pts_[0].p, pp.size() * sizeof(double), hipMemcpyDeviceToDevice), "copy pts")) return MI_BA_ERROR;
    begun_ = false;
    return 0;
}

// back to the parameters of the last set_params / prepare without touching the host (a 24 MB device copy at C3), then begin()
int Engine::restart()
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_ || !d_cam_a_init_.p) { err = "restart: prepare first"; return MI_BA_ERROR; }
    cur_ = 0; cam_lin_valid_ = false;
    if (!hipok(hipMemcpyAsync(d_R0_[0].p, d_R0_init_.p, d_R0_init_.n * sizeof(double), hipMemcpyDeviceToDevice, st_), "restore R0")) return MI_BA_ERROR;
    if (!hipok(hipMemcpyAsync(d_cam_a_[0].p, d_cam_a_init_.p, d_cam_a_init_.n * sizeof(double), hipMemcpyDeviceToDevice, st_), "restore cam_a")) return MI_BA_ERROR;
    if (d_pts_init_.n && !hipok(hipMemcpyAsync(d_pts_[0].p, d_pts_init_.p, d_pts_init_.n * sizeof(double), hipMemcpyDeviceToDevice, st_), "restore pts")) return MI_BA_ERROR;
    return begin();
}

// one element-wise reduction over the ranks, in place on the device: RCCL on the engine's stream, or (tests only)
// the host callback
bool Engine::exchange(double *dev, size_t count, int op)
{
    if (host_ar_) {
        h_ar_.resize(count);
        if (!hipok(hipMemcpyAsync(h_ar_.data(), dev, count * sizeof(double), hipMemcpyDeviceToHost, st_), "exchange: download") ||
            !hipok(hipStreamSynchronize(st_), "exchange: sync")) return false;
        if (host_ar_(host_ar_user_, h_ar_.data(), (int64_t)count, op) != 0) return fail("host all-reduce callback failed");
        return hipok(hipMemcpyAsync(dev, h_ar_.data(), count * sizeof(double), hipMemcpyHostToDevice, st_), "exchange: upload") &&
               hipok(hipStreamSynchronize(st_), "exchange: sync");
    }
    ncclResult_t r = ncclAllReduce(dev, dev, count, ncclDouble, op ? ncclMax : ncclSum, (ncclComm_t)comm_, st_);
    if (r != ncclSuccess) return fail(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return true;
}

bool Engine::allreduce1()
{
    if (!comm_ && !host_ar_) return true;
    double t0 = now_ms();
    const size_t nsum = nslot_() + (size_t)plan_.nb_up() * 64 + (size_t)3 * mv_ * kCP + 8;   // slots1, S_up, camvec, scal[0, 8)
    const int xp = lin_pair_ ? 2 : 0;
    if (timing_detail >= 2) (void)hipEventRecord(evx_[xp], st_);
    if (!exchange(d_acc_.p, nsum, 0)) return false;
    if (cfg.world > 1) launch_fold_max(st_, slots1_(), cfg.world, scal_() + 8);
    if (timing_detail >= 2) (void)hipEventRecord(evx_[xp + 1], st_);
    timing.allreduce_ms += now_ms() - t0; timing.allreduces += 1;
    return true;
}

bool Engine::allreduce2()
{
    if (!comm_ && !host_ar_) return true;
    if (!exchange(scal_() + 32, 16 + nslot_(), 0)) return false;                              // scal[32, 48), slots2
    if (cfg.world > 1) launch_fold_max(st_, slots2_(), cfg.world, scal_() + 40);
    return true;
}

LongArgs Engine::long_args(double mu, double *dB_dev, int part_row0)
{
    LongArgs a{};
    a.long_pts = d_long_pts_.p; a.pt_obs = d_pt_obs_.p; a.obs_cam = d_obs_cam_.p; a.obs_rec = d_obs_rec_.p;
    a.cam_fixed = d_fixed_real_.p; a.obs_xy = d_obs_xy_.p;
    a.cam_lin = d_cam_lin_[cur_].p; a.cam_lin_new = d_cam_lin_[cur_ ^ 1].p; a.pts = d_pts_[cur_].p; a.dA = d_dA_.p;
    a.pt_cw = d_pt_cw_.p; a.pt_cv = d_pt_cv_.p;
    a.rec = d_rec_.p; a.ptout = d_ptout_.p; a.part = d_part_.p;
    a.pts_new = d_pts_[cur_ ^ 1].p; a.dB = dB_dev; a.mu = mu; a.eps5 = cfg.lm_variant >= 1 ? 0.0 : cfg.opts[5];
    a.nlong = (int)plan_.long_pts.size(); a.m = m_; a.part_row0 = part_row0; a.cfg = model();
    a.scal = scal_();
    return a;
}

// linearise at the current estimate with damping mu and build the reduced camera system
bool Engine::linear_phase(double mu, bool want_diag, const double *spec)
{
    const Plan &pl = plan_;
    const ModelCfg mc = model();
    launch_zero(st_, d_acc_.p, d_acc_.n);
    // the camera records of the current estimate: after an accepted trial they are the ones k_cam_update wrote for the trial point
    if (!cam_lin_valid_) {
        launch_cam_prep(st_, m_, cnp, mc, d_cam_a_[cur_].p, d_R0_[cur_].p, d_ff_.p, d_cam_lin_[cur_].p);
        // PBA's parametrisation: whatever omega the caller passed is folded into the rotation the camera is linearised about
        if (mc.pba_update) launch_cam_rebase(st_, m_, split_ ? 2 * kCP : kCP, d_cam_a_[cur_].p, d_cam_lin_[cur_].p, d_R0_[cur_].p);
        cam_lin_valid_ = true;
    }
    BuildArgs a{};
    a.tile_batch = d_tile_batch_.p; a.tile_slot = d_tile_slot_.p; a.tile_cslot = d_tile_cslot_.p;
    a.tile_mode = d_tile_mode_.p; a.has_table = pl.ntile_table > 0; a.has_mfma = pl.ntile_mfma > 0; a.has_uni = uni_other_; a.has_uni_fixed = uni_fixed_; a.has_reg = pl.ntile_lds - pl.ntile_table - pl.ntile_mfma - pl.ntile_uni > 0;
    a.slot_gblk = d_slot_gblk_.p; a.cslot_cam = d_cslot_cam_.p;
    a.batch_obs = d_batch_obs_.p; a.batch_pt = d_batch_pt_.p; a.batch_pair = d_batch_pair_.p; a.pt_obs = d_pt_obs_.p;
    a.obs_cam = d_obs_cam_.p; a.obs_pt = d_obs_pt_.p; a.obs_cslot = d_obs_cslot_.p; a.obs_xy = d_obs_xy_.p; a.cam_fixed = d_fixed_real_.p;
    a.pair = d_pair_.p; a.obs_rec = d_obs_rec_.p; a.rec = d_rec_.p;
    a.m = mv_; a.cam_lin = d_cam_lin_[cur_].p; a.pts = d_pts_[cur_].p; a.mu = mu;
    a.S_up = S_up_(); a.camvec = camvec_(); a.ptout = d_ptout_.p; a.part = d_part_.p; a.cfg = mc; a.spec = spec;
    const bool pba = cfg.lm_variant == 1;       // PBA's diagonal damping needs diag U of every linearisation
    a.dbg = d_dbg_.p; a.want_diag = (want_diag || pba) ? 1 : 0;
    a.pt_cw = d_pt_cw_.p; a.pt_cv = d_pt_cv_.p;
    lin_pair_ = (lin_count_ & 1) ? 6 : 0;
    // level 1 (bench.py's timed steps): a pair of events costs ~12 us of idle GPU, so only every kBuildTimedEvery-th build is bracketed
    const bool timed = timing_detail >= 2 || (timing_detail == 1 && (lin_count_ / 2) % kBuildTimedEvery == 0);
    lin_timed_[lin_pair_ ? 1 : 0] = timed;
    ++lin_count_;
    if (timed) (void)hipEventRecord(ev_[lin_pair_], st_);
    const bool ordered = cfg.deterministic == 1 && pl.ntile_lds > 0;   // per-tile copies + sums in tile order instead of f64 atomics
    if (ordered) { a.tile_blk = d_tile_blk_.p; a.tile_cam = d_tile_cam_.p; }
    a.tile0 = 0;
    launch_build(st_, cnp, true, pl.ntile_lds, a);
    if (ordered) launch_tile_reduce(st_, cnp, pl.nb_up(), mv_, d_tb_ptr_.p, d_tb_list_.p, d_tc_ptr_.p, d_tc_list_.p, d_tile_blk_.p, d_tile_cam_.p, S_up_(), camvec_());
    a.tile0 = pl.ntile_lds;
    launch_build(st_, cnp, false, pl.ntile() - pl.ntile_lds, a);
    {
        LongArgs la = long_args(mu, nullptr, pl.ntile());
        la.spec = spec;
        launch_long_build(st_, cnp, la);
        // the row path's blocks of S and camera-side vectors, from the records the two launches above have left
        SchurArgs sa{};
        sa.rec = d_rec_.p; sa.wk_blk = d_wk_blk_.p; sa.wk_q = d_wk_q_.p; sa.rpair = d_rpair_.p;
        sa.diag_cam = d_wk_diag_cam_.p; sa.diag_blk = d_wk_diag_blk_.p; sa.rec_cam = d_rec_cam_.p;
        sa.nwork = (int)pl.wk_blk.size(); sa.ndiag = (int)pl.wk_diag_cam.size(); sa.m = m_; sa.mv = mv_; sa.want_diag = (want_diag || pba) ? 1 : 0;
        sa.wk_vblk = d_wk_vblk_.p; sa.diag_vblk = d_wk_diag_vblk_.p;
        sa.sb_blk = d_sb_blk_.p; sa.sb_ent = d_sb_ent_.p; sa.sb_q = d_sb_q_.p; sa.nsuper = (int)pl.sb_q.size() - 1;
        sa.sb8_blk = d_sb8_blk_.p; sa.sb8_ent = d_sb8_ent_.p; sa.sb8_q = d_sb8_q_.p; sa.nsuper8 = (int)pl.sb8_q.size() - 1;
        sa.S_up = S_up_(); sa.camvec = camvec_(); sa.spec = spec;
        launch_schur_blocks(st_, cnp, sa);
    }
    if (timed) (void)hipEventRecord(ev_[lin_pair_ + 1], st_);
    // the build's partial rows (cost, |X|^2, |e_b|inf, ...): nothing on the device needs their sums before the trial's end on one rank -- k_reduce_final
    // adds them up with the back-substitution's (two launches less per trial); several ranks exchange them now, the parity hooks read them now
    build_part_pending_ = cfg.world == 1 && !want_diag;
    if (!build_part_pending_)
        launch_reduce_partials(st_, pl.ntile() + (int)pl.long_pts.size(), d_part_.p, d_stage_.p, scal_() + SC_COST,
                               cfg.world > 1 ? slots1_() + (size_t)cfg.rank * 8 : scal_() + SC_EB_INF);
    if (!allreduce1()) return false;
    // the coarse space is built from the first trial after a solve that needed many iterations, and then for the rest of the run
    if (co_.ng && !co_on_ && (co_force_ == 1 || pcg_pred_ >= kTwoLevelFrom)) co_on_ = true;
    co_active_ = co_.ng && co_on_;
    if (pcg_ell_w_) {   // the fixed-pitch matrix relies on its pads being zero: a trial that wrote the buffer in another layout (block b at b) has soiled them
        const bool ell_now = pcg_args().ell_w != 0;
        if (ell_now && ell_pads_dirty_ && !hipok(hipMemsetAsync(d_S_pcg_.p, 0, d_S_pcg_.n * sizeof(double), st_), "memset S~")) return false;
        ell_pads_dirty_ = !ell_now;
    }
    fv_rows_ = want_diag ? (mv_ * kCP + 63) / 64 : mv_;   // (launch_finalize: the camera-side vectors inside k_block_inv unless the hooks want the full matrix / the sums now)
    launch_finalize(st_, cnp_cam_(), mv_, pl.nb_full(), d_fixed_.p, d_pfix_.p, mu, spec, d_bsr_rowidx_.p, d_bsr_col_.p, d_bsr_src_.p, d_bsr_trans_.p, d_bsr_diag_.p,
                    S_up_(), want_diag ? d_S_full_.p : nullptr, camvec_(), d_cam_a_[cur_].p, d_cam_cw_.p, d_cam_cv_.p, links_, d_E_.p,
                    d_cam_part_.p, d_stage_.p, scal_(), d_Linv_.p, d_S_pcg_.p, want_diag /* a linear phase on its own (begin, parity hooks) */, pcg_start(),
                    co_active_ ? co_ : CoarseArgs{}, d_bsr_row_.p, pba);
    cg_start_fresh_ = true;
    return hipok(hipGetLastError(), "launch (linear phase)");
}

PeekArgs Engine::peek_args(double mu) const
{
    PeekArgs k{};
    k.c_lag = c_lag_; k.mu = mu;
    k.eps2_sq = cfg.opts[2] * cfg.opts[2]; k.eps3_sq = cfg.opts[3] * cfg.opts[3]; k.eps4_sq = cfg.opts[4] * cfg.opts[4]; k.eps5 = cfg.opts[5];
    k.pl2_inactive = pl2_inactive_; k.itno = itno_; k.mode = cfg.mode; k.stops = cfg.disable_stops ? 0 : 1;
    k.last = itno_ + 1 >= cfg.itmax ? 1 : 0;
    k.variant = cfg.lm_variant;
    if (cfg.lm_variant >= 1) {     // PBA's controller: opts[] mean other things there (mi_ba.h), and its cap counts trials
        k.mu_min = cfg.opts[3]; k.mu_max = cfg.opts[4]; k.d_thr = cfg.opts[2];
        k.last = trials_ + 1 >= cfg.itmax ? 1 : 0;
    }
    return k;
}

// The scaled matrix of the reduced solve in single precision (vectors, dot products and recurrences stay f64): for systems whose matrix
// is streamed from HBM / the Infinity Cache every iteration (> kPcgF32FromBytes; C4: 208 MB at 4.9 TB/s = 43 us per iteration) and a
// tolerance the rounding of the matrix (6e-8 relative) is far below.  Small systems (C1-C3, C5: L2 / cache resident, launch-bound) keep
// f64 -- nothing changes for them.  Not with the coarse space or the persistent solve (their kernels read f64).  MI_BA_PCG_F32=0 / 1 forces.
bool Engine::pcg_f32_() const
{
    if (co_.ng || persist_ok_) return false;
    if (pcg_f32_force_ >= 0) return pcg_f32_force_ != 0;
    return (size_t)plan_.nb_full() * 64 * sizeof(double) > pcg_f32_from_ && cfg.pcg_tol >= 1e-6;
}

// the f64 matrix in the product's column form (kernels.hip row_products): whenever the plain launched iteration is the only reader
bool Engine::pcg_cm_() const { return !co_.ng && !persist_ok_ && !pcg_f32_(); }

PcgStart Engine::pcg_start()
{
    const PcgArgs p = pcg_args();
    timing.pcg_matrix_f32 = pcg_f32_() ? 1.0 : 0.0;
    timing.pcg_push = p.mb0 ? 1.0 : 0.0;
    return PcgStart{ p.E, p.x, p.r0, p.p, p.s0, p.s1, persist_ok_ ? d_bar_.p : nullptr, d_scale_list_.p, (int)(d_scale_list_.n / 8), p.ell_w ? 1 : 0, pcg_f32_() ? 1 : pcg_cm_() ? 2 : 0 };
}

PcgArgs Engine::pcg_args()
{
    const size_t vec = (size_t)mv_ * kCP;
    PcgArgs p{};
    p.bsr_row = d_bsr_row_.p; p.bsr_col = d_bsr_col_.p; p.S = d_S_pcg_.p; p.Linv = d_Linv_.p; p.E = d_E_.p;
    p.S32 = pcg_f32_() ? reinterpret_cast<const float *>(d_S_pcg_.p) : nullptr;
    p.cm = pcg_cm_() ? 1 : 0;
    if (p.cm && pcg_ell_w_ && !pcg_push_ok_) { p.ell_w = pcg_ell_w_; p.bsr_col = d_ell_col_.p; }
    if (p.cm && pcg_push_ok_) { p.mb0 = d_mb_.p; p.mb1 = d_mb_.p + (size_t)plan_.nb_full() * 24; p.bsr_tq = d_bsr_tq_.p; }
    double *v = d_pcg_.p;
    p.da = d_dA_.p; p.x = v; p.p = v + vec; p.r0 = v + 2 * vec; p.r1 = v + 3 * vec; p.w0 = v + 4 * vec; p.w1 = v + 5 * vec;
    p.s0 = v + 6 * vec; p.s1 = v + 7 * vec;
    // a block row of S~ gets a whole workgroup (four wavefronts share its blocks) when there are few rows.  (Measured with more
    // rows, MI_BA_PCG_ROW_WG=1: slower -- C3 0.27 -> 0.37 ms per solve, C4 (1 700 rows of ~240 blocks, S~ = 208 MB) 2.35 -> 2.62.)
    static const int force_row_wg = std::getenv("MI_BA_PCG_ROW_WG") ? std::atoi(std::getenv("MI_BA_PCG_ROW_WG")) : -1;
    const bool row_wg = co_.ng ? false : force_row_wg >= 0 ? force_row_wg != 0 : mv_ <= kPcgRowPerWorkgroupMax;   // (the coarse space is laid out for four rows per workgroup)
    p.ng = (mv_ + 7) / 8; p.nwg = row_wg ? mv_ : (mv_ + 3) / 4;
    static const int force_wpr = std::getenv("MI_BA_PCG_WPR") ? std::atoi(std::getenv("MI_BA_PCG_WPR")) : 0;
    p.wpr = force_wpr == 1 || force_wpr == 2 ? force_wpr
                                                : ((size_t)plan_.nb_full() < (size_t)kPcgTwoWavesPerRowBelow * mv_ && p.nwg <= kPcgTwoWavesPerRowUpTo ? 2 : 1);
    p.rz_hist = d_rzpq_.p; p.alpha_hist = p.rz_hist + (cfg.pcg_maxit + 4); p.gp = p.alpha_hist + (cfg.pcg_maxit + 4); p.dp = p.gp + 2 * (size_t)p.nwg;
    p.scal = scal_(); p.tol = cfg.pcg_tol; p.minit = cfg.pcg_minit; p.m = mv_; p.it = 0;
    static const int force_tot = std::getenv("MI_BA_PCG_TOTALS") ? std::atoi(std::getenv("MI_BA_PCG_TOTALS")) : -1;
    if (!row_wg && !co_active_ && (force_tot >= 0 ? force_tot != 0 : p.nwg >= kPcgTotalsFrom)) p.tot = p.dp + 2 * (size_t)p.nwg;   // (room: 4 mv doubles are allocated for the partials)
    if (co_active_) p.co = co_;
    p.stamps = d_pstamps_.p;
    return p;
}

// CG iterations [it0, pcg_maxit) in chunks with a host look after each (device-side convergence is sticky)
bool Engine::pcg_chunks(int it0)
{
    PcgArgs p = pcg_args();
    int it = it0;
    while (it < cfg.pcg_maxit) {
        const int chunk = std::min(kPcgChunk, cfg.pcg_maxit - it);
        for (int c = 0; c < chunk; ++c) { p.it = it + c; launch_pcg_iter(st_, p); }
        it += chunk;
        p.it = it; p.last = it >= cfg.pcg_maxit; launch_pcg_tail(st_, p); // rz_hist[it] and the flags after the chunk's last update
        if (!read_scalars()) return false;
        if (h_scal_[SC_PCG_FINAL] != 0.0) break;
    }
    return true;
}

// CG on S~ x~ = E~, then the update phase, then the trial's scalars on the host (h_scal_).  A host look at the solve costs
// a round trip during which the GPU idles (~30 us), so nothing here waits for one: the expected number of iterations is
// enqueued in one go -- iterations past convergence are 2 us no-op launches (sticky device-side flag) -- followed by the
// camera update and the back-substitution, which look at SC_PCG_FINAL themselves: when the guess fell short (the flag is
// still down) the back-substitution does nothing, and the host, which sees the flag with the trial's scalars, continues the
// iterations in checked chunks and launches the update phase again.
bool Engine::solve_phase(double mu, double *dB_dev, bool may_speculate)
{
    PcgArgs p = pcg_args();
    // (histories and partials need no reset: every slot a launch reads was written by an earlier launch of this solve)
    if (timing_detail >= 2) (void)hipEventRecord(ev_[2], st_);
    if (!cg_start_fresh_) launch_pcg_reinit(st_, p);   // a second solve on the same linearisation (parity hooks)
    const bool fresh = cg_start_fresh_;
    cg_start_fresh_ = false;
    bool persistent = persist_ok_ && !co_active_ && fresh;   // (a second solve's barrier words are not at zero: the launched iteration then)
    if (persistent) {
        PersistArgs pg{};
        pg.a = p; pg.bar = d_bar_.p; pg.rows_per_wg = persist_rows_; pg.maxit = cfg.pcg_maxit; pg.group = persist_group_;
        pg.win_ptr = d_win_ptr_.p; pg.win_col = d_win_col_.p; pg.blk_loc = d_blk_loc_.p;
        if (!launch_pcg_persist(st_, pg, persist_nwg_, persist_lds_)) { persist_ok_ = false; persistent = false; }   // refused: never again
        else timing.pcg_persistent += 1;
    }
    if (!persistent) launch_pcg_init(st_, p);
    // the count grows geometrically from trial to trial while the damping falls by 1/3 (measured on C3: 9, 14, 20, 33, 54, 93,
    // then the cap): extrapolate the last ratio, with a margin -- an iteration too many costs 2 us, one too few a round trip
    const double ratio = pcg_pred_prev_ > 0 ? std::min(2.0, std::max(1.0, (double)pcg_pred_ / pcg_pred_prev_)) : 1.0;
    int it0 = std::min(cfg.pcg_maxit, std::max(kPcgChunk, (int)std::ceil(pcg_pred_ * ratio * 1.25) + 3));
    if (const char *force = std::getenv("MI_BA_PCG_FIRST_RUN")) it0 = std::min(cfg.pcg_maxit, std::max(1, std::atoi(force)));   // test hook: make the guess fall short
    if (!persistent) for (int it = 0; it < it0; ++it) { p.it = it; launch_pcg_iter(st_, p); }
    p.it = it0; p.last = it0 >= cfg.pcg_maxit; p.tail = persistent ? 0 : 1;   // (the persistent launch has set the flags itself)
    upd_p_ = p;                  // k_cam_update: what k_pcg_tail does, then da = Linv^T x~
    p.tail = 0;
    if (timing_detail >= 2) (void)hipEventRecord(ev_[3], st_);
    const bool spec = may_speculate && speculate_;
    if (!update_phase(mu, dB_dev, spec && cfg.world == 1)) return false;
    spec_enqueued_ = false;
    if (spec) {
        // The host needs this trial's scalars to decide (accept? which damping next? stop?), and until it has decided the GPU
        // would idle: ~35 us per trial, a fifth of a trial on a 20-camera scene.  So the common decision is taken on the
        // device as well (k_lm_peek: the controller's accept test and damping update, operation for operation) and the
        // next trial's linear phase is enqueued behind it at once, reading its damping from HBM; the host waits only for the
        // copy of the scalars, runs the controller as ever, and adopts what was built when its own decision -- accepted,
        // no stop, the same damping bit for bit -- is the one the device acted on.  Otherwise (rejected: k_build saw the
        // flag and did nothing; stopped; a reduced solve that needs more iterations) it linearises again as before.
        if (cfg.world > 1) launch_lm_peek(st_, scal_(), peek_args(mu), d_spec_.p);   // after the exchange; on one rank k_reduce_final has done it
        if (!hipok(hipMemcpyAsync(h_scal_, scal_(), sizeof(double) * kScalCount, hipMemcpyDeviceToHost, st_), "copy scal")) return false;
        (void)hipEventRecord(ev_[8], st_);
        cur_ ^= 1;                                   // the trial point, where an accepted trial continues
        const bool ok = linear_phase(0.0, false, d_spec_.p);
        cur_ ^= 1;
        spec_pair_ = lin_pair_;
        if (!ok) return false;
        spec_enqueued_ = true;
        if (!hipok(hipEventSynchronize(ev_[8]), "sync (scalars)")) return false;
    } else if (!read_scalars()) return false;
    if (h_scal_[SC_PCG_FINAL] == 0.0) { // the guess fell short -- or the persistent launch gave up at a barrier (status 3)
        timing.pcg_resumes += 1;
        const bool gave_up = persistent && h_scal_[SC_PCG_STATUS] == 3.0;
        if (persistent) {
            persist_ok_ = false;        // whatever happened, the launched iteration from here on
            if (cfg.verbose && cfg.rank == 0) std::fprintf(stderr, "[mi_ba] persistent reduced solve %s: launched iterations from now on\n", gave_up ? "timed out at a barrier" : "did not finish");
        }
        if (spec_enqueued_ || persistent) {   // the linear phase enqueued behind it has overwritten this trial's system (or the vectors are a half-done persistent solve's): build it again
            spec_enqueued_ = false;
            if (!linear_phase(mu, false)) return false;
            cg_start_fresh_ = false;
            launch_pcg_init(st_, p);
            it0 = 0;
        }
        if (!pcg_chunks(it0)) return false;
        upd_p_ = p; upd_p_.tail = 0;
        if (!update_phase(mu, dB_dev) || !read_scalars()) return false;
    }
    pcg_pred_prev_ = pcg_pred_ > 0 ? pcg_pred_ : (int)h_scal_[SC_PCG_ITERS];
    pcg_pred_ = (int)h_scal_[SC_PCG_ITERS];
    return true;
}

// camera update, back-substitution, cost at the trial point
// workgroups of k_backsub: whole rounds of what the chip holds at once (3 wavefronts per SIMD at 168 VGPRs: 3 072), so that no
// SIMD idles through a part-filled last round; MI_BA_BACKSUB_GRID overrides (probe)
static int backsub_grid()   // never above kBacksubGridMax: prepare() sizes the partial rows for that many (+ the long tracks)
{
    static const int v = std::getenv("MI_BA_BACKSUB_GRID") ? std::atoi(std::getenv("MI_BA_BACKSUB_GRID")) : 0;
    return v > 0 ? std::min(v, kBacksubGridMax) : kBacksubGrid;
}

bool Engine::update_phase(double mu, double *dB_dev, bool peek)
{
    const Plan &pl = plan_;
    const ModelCfg mc = model();
    launch_cam_update(st_, m_, cnp, mc, d_fixed_.p, mu, d_cam_a_[cur_].p, upd_p_, d_dA_.p, camvec_(), d_R0_[cur_].p, d_ff_.p, links_,
                      d_cam_a_[cur_ ^ 1].p, d_cam_lin_[cur_ ^ 1].p, d_cam_part2_.p, mc.pba_update ? d_R0_[cur_ ^ 1].p : nullptr);
    BacksubArgs b{};
    b.batch_obs = d_bs_batch_obs_.p; b.batch_pt = d_bs_batch_pt_.p; b.batch_obs_end = d_bs_batch_obs_end_.p; b.batch_pt_end = d_bs_batch_pt_end_.p;
    b.tile_mode = d_tile_mode_.p; b.tile_batch = d_tile_batch_.p; b.tile_cslot = d_tile_cslot_.p; b.cslot_cam = d_cslot_cam_.p;
    b.bld_batch_obs = d_batch_obs_.p; b.bld_batch_pt = d_batch_pt_.p;
    b.pt_obs = d_pt_obs_.p; b.obs_cam = d_obs_cam_.p; b.obs_pt = d_obs_pt_.p;
    b.obs_xy = d_obs_xy_.p; b.cam_fixed = d_fixed_real_.p; b.cam_lin = d_cam_lin_[cur_].p; b.cam_lin_new = d_cam_lin_[cur_ ^ 1].p; b.pts = d_pts_[cur_].p;
    b.ptout = d_ptout_.p; b.dA = d_dA_.p; b.pts_new = d_pts_[cur_ ^ 1].p; b.dB = dB_dev; b.part = d_part_.p + bs_row0_() * kPartW;
    b.mu = mu; b.eps5 = cfg.lm_variant >= 1 ? 0.0 : cfg.opts[5];   /* (Bundler's rule; under PBA's controller opts[5] is the auto-switch threshold) */ b.nbatch = (int)pl.bs_batch_obs.size(); b.ngrid = std::min(b.nbatch, backsub_grid()); b.cfg = mc; b.scal = scal_();
    const int nlong = (int)pl.long_pts.size();
    b.uni_ntile = pl.bs_uni ? pl.ntile_lds : 0; b.uni_row0 = b.ngrid + nlong; b.uni_fixed = uni_fixed_; b.uni_other = uni_other_;     // partial rows: k_backsub's | the long tracks' | the uniform tiles'
    if (timing_detail >= 2) (void)hipEventRecord(ev_[4], st_);
    launch_backsub(st_, cnp, b);
    { LongArgs la = long_args(mu, dB_dev, b.ngrid); la.part = b.part; launch_long_backsub(st_, cnp, la); }
    if (timing_detail >= 2) (void)hipEventRecord(ev_[5], st_);
    launch_reduce_final(st_, b.ngrid + nlong + b.uni_ntile, b.part, d_stage_.p, scal_() + SC_COST_NEW,
                        cfg.world > 1 ? slots2_() + (size_t)cfg.rank * 8 : scal_() + SC_MAX_PCT,
                        m_, mv_, d_cam_part_.p, d_cam_part2_.p, scal_(), peek ? d_spec_.p : nullptr, peek_args(mu),
                        build_part_pending_ ? pl.ntile() + nlong : 0, d_part_.p, scal_() + SC_COST, scal_() + SC_EB_INF, fv_rows_);
    if (!allreduce2()) return false;
    return hipok(hipGetLastError(), "launch (solve phase)");
}

bool Engine::read_scalars()
{
    if (!hipok(hipMemcpyAsync(h_scal_, scal_(), sizeof(double) * kScalCount, hipMemcpyDeviceToHost, st_), "copy scal")) return false;
    return hipok(hipStreamSynchronize(st_), "sync");
}

int Engine::begin()
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_) { err = "begin: prepare first"; return MI_BA_ERROR; }
    cfg.lm_variant = lm_variant_cfg_;   // (a solve that switched to identity damping starts the next one as configured)
    mu_ = 0; nu_ = 2; itno_ = 0; stop_ = 0; nfev_ = 1; njev_ = 0; nlss_ = 0; dp_L2_ = DBL_MAX; new_outer_ = true;
    pcg_pred_ = pcg_pred_prev_ = 0;      // a new solve: nothing to extrapolate from
    co_on_ = false;
    spec_ready_ = spec_enqueued_ = false;
    trace.clear();
    // probe pass: cost at p0 and the largest diagonal entry of U, V -> mu0 = tau * max (sba_levmar.c:1124-1128)
    if (!linear_phase(1.0, true) || !read_scalars()) return MI_BA_ERROR;
    c_lag_ = h_scal_[SC_CONS_CAM] + h_scal_[SC_CONS_PTS];       // sba_levmar.c:809-842
    p_eL2_ = init_p_eL2_ = h_scal_[SC_COST] + c_lag_ + h_scal_[SC_LINK];
    maxdiag_ = measured_maxdiag();
    eab_inf_ = std::max(h_scal_[SC_EA_INF], h_scal_[SC_EB_INF]);
    mu_ = cfg.lm_variant >= 1 ? std::fabs(cfg.opts[0]) : std::fabs(cfg.opts[0]) * maxdiag_;   // PBA: __lm_initial_damp multiplies the diagonals itself
    trials_ = 0;
    if (cfg.verbose && cfg.rank == 0)   // sba's own line, sba_levmar.c:844
        std::printf("initial motstr-SBA error %g [%g]\n", p_eL2_ / (double)nobs_, p_eL2_);
    if (!std::isfinite(p_eL2_)) stop_ = 7;
    begun_ = true;
    return 0;
}

void Engine::accept_trial() { cur_ ^= 1; }

// largest diagonal entry of the free blocks: U and V, U only (points fixed) or V only (cameras fixed)
double Engine::measured_maxdiag() const
{
    const double u = cfg.mode == 2 ? 0.0 : h_scal_[SC_MAXDIAG_U], v = cfg.mode == 1 ? 0.0 : h_scal_[SC_MAXDIAG_V];
    return std::max(std::max(u, v), DBL_MIN);
}

// one damping trial; the control flow is sba_levmar.c:1085-1614 unrolled into a resumable step
int Engine::step()
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!begun_) { err = "step: begin first"; return MI_BA_ERROR; }
    if (stop_ || (cfg.lm_variant >= 1 ? trials_ : itno_) >= cfg.itmax) { if (!stop_) stop_ = 3; return 2; }
    const double t0 = now_ms();
    const double eps1 = std::fabs(cfg.opts[1]), eps2 = std::fabs(cfg.opts[2]), eps2_sq = cfg.opts[2] * cfg.opts[2],
                 eps3_sq = cfg.opts[3] * cfg.opts[3], eps4_sq = cfg.opts[4] * cfg.opts[4], eps5 = cfg.opts[5];
    const bool stops = !cfg.disable_stops;
    if (new_outer_) { ++njev_; new_outer_ = false; }
    const double mu = mu_;
    int pair = spec_pair_;
    if (spec_ready_) { spec_ready_ = false; timing.spec_adopted += 1; }   // this trial's linear phase was enqueued behind the previous trial (solve_phase)
    else { if (!linear_phase(mu, false)) return MI_BA_ERROR; pair = lin_pair_; }
    if (!solve_phase(mu, nullptr, true)) return MI_BA_ERROR;   // h_scal_ holds the trial's scalars
    float ms = 0;
    if (timing_detail >= 1 && lin_timed_[pair ? 1 : 0] && hipEventElapsedTime(&ms, ev_[pair], ev_[pair + 1]) == hipSuccess) { timing.build_ms += ms; timing.build_launches += 1; }
    if (timing_detail >= 2 && hipEventElapsedTime(&ms, ev_[2], ev_[3]) == hipSuccess) { timing.pcg_ms += ms; timing.pcg_timed += 1; }
    if (timing_detail >= 2 && hipEventElapsedTime(&ms, ev_[4], ev_[5]) == hipSuccess) { timing.backsub_ms += ms; timing.backsub_launches += 1; }
    if (timing_detail >= 2 && (comm_ || host_ar_)) {   // the exchange of this trial's linear phase, and the replicated stretch up to the back-substitution
        const int xp = pair ? 2 : 0;
        if (hipEventElapsedTime(&ms, evx_[xp], evx_[xp + 1]) == hipSuccess) { timing.exchange_ms += ms; timing.exchanges_timed += 1; }
        if (hipEventElapsedTime(&ms, evx_[xp + 1], ev_[4]) == hipSuccess) { timing.replicated_ms += ms; timing.replicated_timed += 1; }
    }
    timing.pcg_solves += 1;
    const double *s = h_scal_;
    timing.pcg_iters += s[SC_PCG_ITERS];
    // cost carried by the reference = residuals at p + the constraint penalty of the previous accepted iterate
    const double c_cur = s[SC_CONS_CAM] + s[SC_CONS_PTS];
    p_eL2_ = s[SC_COST] + c_lag_ + s[SC_LINK];           // links: exact at p, no lag (they are not Bundler's)
    // the truncated drivers' parameter vector is the free block only (sba_levmar.c:2234, 2838)
    p_L2_ = (cfg.mode == 2 ? 0.0 : s[SC_PL2_CAMS]) + (cfg.mode == 1 ? 0.0 : s[SC_PL2_PTS] + pl2_inactive_);
    eab_inf_ = std::max(s[SC_EA_INF], s[SC_EB_INF]);
    // max diag(U, V) enters mu0 (begin) and info[4] only; diag U is not accumulated per trial, so
    // info[4] is normalised by the value of the last linearisation that measured it
    int ret = 0;
    auto end_outer = [&]() { // tail of the outer for-loop body, sba_levmar.c:1614 + the ++itno
        if (p_eL2_ <= eps3_sq && stops) stop_ = 5;
        ++itno_; new_outer_ = true;
        if (itno_ >= cfg.itmax) stop_ = 3; // sba_levmar.c:1617
    };
    auto push_trace = [&](double cost_trial, int accepted) {
        const double row[6] = { (double)itno_, mu, p_eL2_, cost_trial, (double)accepted, s[SC_PCG_ITERS] };
        trace.insert(trace.end(), row, row + 6);
    };
    if (cfg.lm_variant >= 1) {
        // PBA's controller, SparseBundleCPU.cpp:3843-3976 (one pass of its for-loop).  SC_DL_* and SC_DP2_* are in the metric of
        // D = diag(J'J) here (k_cam_update, k_backsub): dL = d'(mu D d + g), |d|_D^2 = d'D d -- PBA's step norm, which it takes on
        // its column-normalised Jacobian (:3872, :3779-3793).
        const double g_thr = cfg.opts[1], d_thr = cfg.opts[2], mu_min = cfg.opts[3], mu_max = cfg.opts[4];
        do {
            if (s[SC_SINGULAR] != 0.0 || s[SC_PCG_STATUS] != 0.0 || s[SC_NOT_PD] != 0.0) {   // :3858-3868 ('I': more damping, the pass does not count)
                push_trace(NAN, 0);
                mu_ *= nu_;
                const int nu2 = (int)((unsigned)nu_ << 1);
                if (nu2 <= nu_ || !std::isfinite(mu_)) { stop_ = 6; ret = 2; } else nu_ = nu2;
                break;
            }
            ++trials_; ++nlss_;
            dp_L2_ = s[SC_DP2_CAMS] + s[SC_DP2_PTS];
            if (std::sqrt(dp_L2_) <= d_thr && stops) { stop_ = 2; ret = 2; break; }          // :3875-3884 "quit on too small change"
            ++nfev_;
            const double pdp_eL2 = s[SC_COST_NEW] + c_cur + s[SC_LINK_NEW];
            const double dF = p_eL2_ - pdp_eL2;
            if (std::isfinite(pdp_eL2) && dF > 0.0) {                                        // :3892
                double dL = s[SC_DL_CAMS] + s[SC_DL_PTS];
                if (!(dL > 0.0)) dL = 0.001 * dF;                                            // :3748
                push_trace(pdp_eL2, 1);
                if (cfg.verbose > 1 && cfg.rank == 0)
                    std::printf("[mi_ba] it %3d  mu %.3e  cost %.9g -> %.9g  pcg %3d  gain %.3f\n", itno_, mu, p_eL2_, pdp_eL2, (int)s[SC_PCG_ITERS], dF / dL);
                accept_trial(); p_eL2_ = pdp_eL2; c_lag_ = c_cur;
                ++itno_; new_outer_ = true;
                if (g_thr > 0.0 && eab_inf_ < g_thr && stops) { stop_ = 1; ret = 2; break; }  // :3927 (the gradient of the point just left)
                double tmp = 2.0 * dF / dL - 1.0;
                tmp = 1.0 - tmp * tmp * tmp;
                mu_ *= (tmp >= kOneThird) ? tmp : kOneThird;                                  // :3940-3946
                nu_ = 2;
                if (mu_ < mu_min) mu_ = mu_min; else if (cfg.lm_variant == 1 && mu_max > 0.0 && mu_ > mu_max) mu_ = mu_max;   // :3948 the upper bound holds under diagonal damping only
                if (trials_ >= cfg.itmax) stop_ = 3;
                ret = stop_ ? 2 : 1;
                spec_ready_ = ret == 1 && spec_enqueued_ && s[SC_SPEC_GO] == 1.0 && s[SC_SPEC_MU] == mu_;
                break;
            }
            push_trace(pdp_eL2, 0);
            if (cfg.lm_variant == 1 && cfg.opts[5] > 0.0 && mu_ > cfg.opts[5]) {            // :3962-3966 __lm_damping_auto_switch: from here on mu I
                cfg.lm_variant = 2; mu_ = cfg.opts[5]; nu_ = 2;
                if (cfg.verbose > 1 && cfg.rank == 0) std::printf("[mi_ba] NOTE: switch to damping with an identity matrix\n");
            } else {
                mu_ *= nu_;                                                                   // :3969-3974
                const int nu2 = (int)((unsigned)nu_ << 1);
                if (nu2 <= nu_ || !std::isfinite(mu_)) { stop_ = 6; ret = 2; break; }
                nu_ = nu2;
            }
            if (trials_ >= cfg.itmax) { stop_ = 3; ret = 2; break; }
            ret = 0;
        } while (false);
        timing.step_ms += now_ms() - t0; timing.steps += 1;
        if (stop_) cfg.lm_variant = lm_variant_cfg_;   // (the auto-switch to identity damping belongs to the solve it happened in: the parity hooks
        return ret;                                    //  and model() read the configured variant again once the solve is over)
    }
    do {
        if (eab_inf_ <= eps1 && stops) { dp_L2_ = 0.0; stop_ = 1; ret = 2; break; }          // :1117-1121
        if (s[SC_SINGULAR] != 0.0) { push_trace(NAN, 0); goto moredamping; }                 // :1156-1158
        nlss_ += cfg.mode == 1 ? nfree_cams_ : cfg.mode == 2 ? n_ : 1; // one system per camera / point there (:2510, :3083)
        if (s[SC_PCG_STATUS] != 0.0 || s[SC_NOT_PD] != 0.0) { push_trace(NAN, 0); goto moredamping; } // issolved == 0
        dp_L2_ = s[SC_DP2_CAMS] + s[SC_DP2_PTS];
        if (dp_L2_ <= eps2_sq * p_L2_ && stops) { stop_ = 2; end_outer(); ret = 2; break; }  // :1450-1454
        if (dp_L2_ >= (p_L2_ + eps2) / kEpsilonSq) {                                          // :1456-1462
            err = "the matrix of the augmented normal equations is almost singular";
            return MI_BA_ERROR;
        }
        ++nfev_;
        {
            const double pdp_eL2 = s[SC_COST_NEW] + c_cur + s[SC_LINK_NEW]; // QUIRK: Bundler's penalty at the OLD p (sba_levmar.c:1498, 1514)
            if (!std::isfinite(pdp_eL2)) { stop_ = 7; end_outer(); ret = 2; break; }         // :1479-1485
            const double dL = s[SC_DL_CAMS] + s[SC_DL_PTS], dF = p_eL2_ - pdp_eL2;
            if (dL > 0.0 && dF > 0.0) {                                                       // :1543
                push_trace(pdp_eL2, 1);
                double tmp = 2.0 * dF / dL - 1.0;
                tmp = 1.0 - tmp * tmp * tmp;
                mu_ = mu_ * ((tmp >= kOneThird) ? tmp : kOneThird);
                nu_ = 2;
                bool discard = false;
                if (stops) {
                    if (pdp_eL2 - 2.0 * std::sqrt(p_eL2_ * pdp_eL2) < (eps4_sq - 1.0) * p_eL2_) stop_ = 4; // :1567
                    // QUIRK kept (sba_levmar.c:1569-1572): stop 8 leaves before p <- p + dp
                    if (cfg.mode == 0 && s[SC_MAX_PCT] < eps5 && itno_ >= 4) { stop_ = 8; discard = true; } // motstr only
                }
                if (cfg.verbose > 1 && cfg.rank == 0)   // per accepted step: iteration, damping, cost, PCG iterations, Bundler's max_pct_change (:1563)
                    std::printf("[mi_ba] it %3d  mu %.3e  cost %.9g -> %.9g  pcg %3d  max_pct_change %s eps5\n", itno_, mu, p_eL2_, pdp_eL2,
                                (int)s[SC_PCG_ITERS], s[SC_MAX_PCT] < eps5 ? "<" : ">=");   // (the kernels keep one bit of it: pct_change_flag)
                if (!discard) { accept_trial(); p_eL2_ = pdp_eL2; c_lag_ = c_cur; }
                end_outer();
                ret = stop_ ? 2 : 1;
                // the linear phase enqueued behind this trial is the next trial's if the device acted on this very decision
                spec_ready_ = ret == 1 && spec_enqueued_ && s[SC_SPEC_GO] == 1.0 && s[SC_SPEC_MU] == mu_;
                break;
            }
            push_trace(pdp_eL2, 0);
        }
    moredamping:                                                                               // :1584-1597
        mu_ *= nu_;
        {
            const int nu2 = (int)((unsigned)nu_ << 1);
            if (nu2 <= nu_) { stop_ = 6; end_outer(); ret = 2; break; }
            nu_ = nu2;
        }
        ret = 0;
    } while (false);
    timing.step_ms += now_ms() - t0; timing.steps += 1;
    return ret;
}

void Engine::fill_info(double info[MI_BA_INFOSZ]) const
{
    info[0] = init_p_eL2_; info[1] = p_eL2_; info[2] = eab_inf_; info[3] = dp_L2_;
    info[4] = mu_ / maxdiag_; info[5] = itno_; info[6] = stop_;
    info[7] = (double)nfev_ * (double)nobs_; info[8] = (double)njev_ * (double)nobs_; info[9] = nlss_;
}

int Engine::run(double info[MI_BA_INFOSZ])
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (begin() != 0) return MI_BA_ERROR;
    while (!stop_ && itno_ < cfg.itmax) {
        int r = step();
        if (r == MI_BA_ERROR) return MI_BA_ERROR;
    }
    if (itno_ >= cfg.itmax) stop_ = 3;
    // info[4] = mu / max diag(U, V) at the final estimate (sba_levmar.c:2033-2043): one more linearisation
    if (info && stop_ != 7) {
        if (!linear_phase(1.0, true) || !read_scalars()) return MI_BA_ERROR;
        maxdiag_ = measured_maxdiag();
    }
    if (info) fill_info(info);
    return stop_ != 7 ? itno_ : MI_BA_ERROR;
}

int Engine::get_params(double *a, double *xyz)
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_) { err = "get_params: prepare first"; return MI_BA_ERROR; }
    const Plan &pl = plan_;
    if (a) {
        std::vector<double> ap((size_t)mv_ * kCP);
        if (!hipok(hipMemcpy(ap.data(), d_cam_a_[cur_].p, ap.size() * sizeof(double), hipMemcpyDeviceToHost), "download cam_a")) return MI_BA_ERROR;
        for (int j = 0; j < m_; ++j)
            for (int r = 0; r < cnp; ++r) a[(size_t)j * cnp + r] = ap[slot_(j, r)];
        if (lm_variant_cfg_ >= 1) {
            // PBA's update keeps the whole rotation in the (moving) R0 and omega at 0; the interface's omega is relative to the rotation the
            // caller passed: omega = log(R R0^T)
            std::vector<double> Rc((size_t)m_ * 9);
            if (!hipok(hipMemcpy(Rc.data(), d_R0_[cur_].p, Rc.size() * sizeof(double), hipMemcpyDeviceToHost), "download R")) return MI_BA_ERROR;
            for (int j = 0; j < m_; ++j) {
                const double *R = &Rc[(size_t)j * 9], *Q = &h_R0_[(size_t)j * 9];
                double w[3] = { ap[slot_(j, 3)], ap[slot_(j, 4)], ap[slot_(j, 5)] };
                if (w[0] == 0.0 && w[1] == 0.0 && w[2] == 0.0) {          // (before the first linearisation of a solve omega is still the caller's)
                    double D[9];
                    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) D[3 * r + c] = R[3 * r] * Q[3 * c] + R[3 * r + 1] * Q[3 * c + 1] + R[3 * r + 2] * Q[3 * c + 2];
                    const double cs = std::min(1.0, std::max(-1.0, 0.5 * (D[0] + D[4] + D[8] - 1.0))), th = std::acos(cs);
                    const double v[3] = { D[7] - D[5], D[2] - D[6], D[3] - D[1] }, sn = std::sin(th);
                    const double k = sn < 1e-12 ? 0.5 : th / (2.0 * sn);
                    for (int r = 0; r < 3; ++r) w[r] = k * v[r];
                }
                for (int r = 0; r < 3; ++r) a[(size_t)j * cnp + 3 + r] = w[r];
            }
        }
    }
    if (xyz) {
        const size_t npl = (size_t)pl.npts_local();
        std::vector<double> pp(npl * 3);
        if (npl && !hipok(hipMemcpy(pp.data(), d_pts_[cur_].p, pp.size() * sizeof(double), hipMemcpyDeviceToHost), "download pts")) return MI_BA_ERROR;
        std::memcpy(xyz, h_xyz_.data(), sizeof(double) * (size_t)n_ * 3); // untouched (unobserved / other ranks')
        if (cfg.world > 1) {
            // every rank contributes its shard to a zero-filled full array; SUM all-reduce = all-gather
            std::vector<double> full((size_t)n_ * 3, 0.0);
            for (size_t t = 0; t < npl; ++t)
                for (int r = 0; r < 3; ++r) full[3 * (size_t)pl.perm[pl.pt_begin + t] + r] = pp[3 * t + r];
            if (!hipok(hipMemcpy(d_full_pts_.p, full.data(), full.size() * sizeof(double), hipMemcpyHostToDevice), "upload full pts")) return MI_BA_ERROR;
            if (!exchange(d_full_pts_.p, full.size(), 0)) return MI_BA_ERROR;
            if (!hipok(hipStreamSynchronize(st_), "sync")) return MI_BA_ERROR;
            if (!hipok(hipMemcpy(full.data(), d_full_pts_.p, full.size() * sizeof(double), hipMemcpyDeviceToHost), "download full pts")) return MI_BA_ERROR;
            for (int t = 0; t < pl.n_active; ++t)
                for (int r = 0; r < 3; ++r) xyz[3 * (size_t)pl.perm[t] + r] = full[3 * (size_t)pl.perm[t] + r];
        } else {
            for (size_t t = 0; t < npl; ++t)
                for (int r = 0; r < 3; ++r) xyz[3 * (size_t)pl.perm[pl.pt_begin + t] + r] = pp[3 * t + r];
        }
    }
    return 0;
}

int Engine::get_rotations(double *R)
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_) { err = "get_rotations: prepare first"; return MI_BA_ERROR; }
    launch_cam_prep(st_, m_, cnp, model(), d_cam_a_[cur_].p, d_R0_[cur_].p, d_ff_.p, d_cam_lin_[cur_].p);
    DevBuf<double> tmp;
    if (!hipok(tmp.alloc((size_t)m_ * 9), "alloc R")) return MI_BA_ERROR;
    launch_rotations(st_, m_, d_cam_lin_[cur_].p, tmp.p);
    if (!hipok(hipStreamSynchronize(st_), "sync")) return MI_BA_ERROR;
    return hipok(hipMemcpy(R, tmp.p, sizeof(double) * (size_t)m_ * 9, hipMemcpyDeviceToHost), "download R") ? 0 : MI_BA_ERROR;
}

// U_j = sum_i A_ij^T A_ij (+ the soft camera constraints' weights on the diagonal) and W_ij = A_ij^T B_ij of the LAST linearisation, from
// the observation records the row path leaves in HBM (A | B | e | ..., kernels.hip RecL) -- what sba hands out through Uout / Wout
// (sba_levmar.c:1633-1700, 1812-1850).  Needs every point on the row path (mi_ba_config.deterministic = 2) and one rank.
//   U  m * cnp * cnp row-major blocks;  W  nobs * cnp * 3, observation k of the caller's order: W[k][r][c]
int Engine::debug_blocks(double *U, double *W)
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_) { err = "debug_blocks: prepare first"; return MI_BA_ERROR; }
    const Plan &pl = plan_;
    if (cfg.deterministic != 2 || cfg.world > 1) { err = "debug_blocks: needs deterministic = 2 (every observation leaves a record) and world == 1"; return MI_BA_ERROR; }
    const int rs = split_ ? kRec9 : kRec, ap = split_ ? 10 : 8;
    const size_t nrec = (size_t)pl.rec_cam[m_];
    std::vector<double> rec(nrec * rs);
    if (nrec && !hipok(hipMemcpy(rec.data(), d_rec_.p, rec.size() * sizeof(double), hipMemcpyDeviceToHost), "download records")) return MI_BA_ERROR;
    if (U) std::fill(U, U + (size_t)m_ * cnp * cnp, 0.0);
    const int64_t nol = pl.nobs_local();
    for (int64_t o = 0; o < nol; ++o) {
        const int r = pl.obs_rec[o], j = pl.obs_cam[o];
        if (r < 0) { err = "debug_blocks: an observation without a record"; return MI_BA_ERROR; }
        const double *q = rec.data() + (size_t)r * rs, *A0 = q, *A1 = q + ap, *B = q + 2 * ap;
        if (U) {
            double *u = U + (size_t)j * cnp * cnp;
            for (int a = 0; a < cnp; ++a) for (int b = 0; b < cnp; ++b) u[a * cnp + b] += A0[a] * A0[b] + A1[a] * A1[b];
        }
        if (W) {
            double *w = W + (size_t)pl.obs_src[o] * cnp * 3;
            for (int a = 0; a < cnp; ++a) for (int c = 0; c < 3; ++c) w[a * 3 + c] = A0[a] * B[c] + A1[a] * B[3 + c];
        }
    }
    if (U && !h_cam_cw_.empty())
        for (int j = 0; j < m_; ++j) for (int a = 0; a < cnp; ++a) U[(size_t)j * cnp * cnp + a * cnp + a] += h_cam_cw_[(size_t)j * cnp + a];
    return 0;
}

int Engine::debug_linearize(double mu, double *S, double *E, double *ea, double *eb, double *V, double *diagU, double *cost)
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_) { err = "debug_linearize: prepare first"; return MI_BA_ERROR; }
    if (cfg.world > 1 && (eb || V)) { err = "debug_linearize: point outputs need world == 1"; return MI_BA_ERROR; }
    spec_ready_ = false;   // whatever was enqueued for the next trial is overwritten here
    if (!linear_phase(mu, true) || !read_scalars()) return MI_BA_ERROR;
    if (mu == 0.0 && h_scal_[SC_SINGULAR] != 0.0) { err = "debug_linearize: a V_i is singular at zero damping"; return MI_BA_ERROR; }
    const Plan &pl = plan_;
    if (cost) *cost = h_scal_[SC_COST];
    const size_t vec = (size_t)mv_ * kCP;
    std::vector<double> tmp;
    auto unpad = [&](const double *src, double *dst) {
        for (int j = 0; j < m_; ++j) for (int r = 0; r < cnp; ++r) dst[(size_t)j * cnp + r] = src[slot_(j, r)];
    };
    if (S) {
        tmp.resize((size_t)pl.nb_full() * 64);
        if (!hipok(hipMemcpy(tmp.data(), d_S_full_.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost), "download S")) return MI_BA_ERROR;
        const size_t N = (size_t)m_ * cnp;
        std::fill(S, S + N * N, 0.0);
        // row / column of the dense matrix a (row of the camera-side arrays, parameter) pair stands for, -1 = padding
        auto dense = [&](int v, int r) -> long long {
            if (!split_) return r < cnp ? (long long)v * cnp + r : -1;
            return (v & 1) ? (r == 0 ? (long long)(v >> 1) * cnp + 8 : -1) : (long long)(v >> 1) * cnp + r;
        };
        for (int j = 0; j < mv_; ++j)
            for (int q = pl.bsr_row[j]; q < pl.bsr_row[j + 1]; ++q) {
                const int k = pl.bsr_col[q];
                for (int r = 0; r < kCP; ++r)
                    for (int c = 0; c < kCP; ++c) {
                        const long long R = dense(j, r), Cc = dense(k, c);
                        if (R >= 0 && Cc >= 0) S[(size_t)R * N + (size_t)Cc] = tmp[(size_t)q * 64 + r * 8 + c];
                    }
            }
    }
    if (E) { tmp.resize(vec); if (!hipok(hipMemcpy(tmp.data(), d_E_.p, vec * sizeof(double), hipMemcpyDeviceToHost), "download E")) return MI_BA_ERROR; unpad(tmp.data(), E); }
    if (ea || diagU) {
        tmp.resize(3 * vec);
        if (!hipok(hipMemcpy(tmp.data(), camvec_(), 3 * vec * sizeof(double), hipMemcpyDeviceToHost), "download camvec")) return MI_BA_ERROR;
        if (ea) unpad(tmp.data(), ea);
        if (diagU) unpad(tmp.data() + 2 * vec, diagU);
    }
    if (eb || V) {
        const size_t npl = (size_t)pl.npts_local();
        tmp.resize(npl * kPtOut);
        if (npl && !hipok(hipMemcpy(tmp.data(), d_ptout_.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost), "download ptout")) return MI_BA_ERROR;
        if (eb) std::fill(eb, eb + (size_t)n_ * 3, 0.0);
        if (V) std::fill(V, V + (size_t)n_ * 9, 0.0);
        for (size_t t = 0; t < npl; ++t) {
            const double *po = tmp.data() + t * kPtOut;
            const size_t i = (size_t)pl.perm[pl.pt_begin + t];
            if (eb) for (int r = 0; r < 3; ++r) eb[3 * i + r] = po[9 + r];
            if (V) { // V = (V*^-1)^-1 - mu I, inverted on the host for the parity test only
                const double a00 = po[0], a01 = po[1], a02 = po[2], a11 = po[3], a12 = po[4], a22 = po[5];
                const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
                const double det = a00 * c00 + a01 * c01 + a02 * c02, id = det != 0.0 ? 1.0 / det : 0.0;
                double *v = V + 9 * i;
                v[0] = c00 * id - mu; v[1] = v[3] = c01 * id; v[2] = v[6] = c02 * id;
                v[4] = (a00 * a22 - a02 * a02) * id - mu; v[5] = v[7] = (a01 * a02 - a00 * a12) * id;
                v[8] = (a00 * a11 - a01 * a01) * id - mu;
            }
        }
    }
    return 0;
}

int Engine::debug_solve(double mu, double *da, double *db, double *cost_new)
{
    DeviceGuard on_device(cfg.device);
    if (!hipok(on_device.status, "hipSetDevice")) return MI_BA_ERROR;
    if (!prepared_) { err = "debug_solve: prepare first"; return MI_BA_ERROR; }
    if (cfg.world > 1 && db) { err = "debug_solve: point outputs need world == 1"; return MI_BA_ERROR; }
    spec_ready_ = false;
    if (!solve_phase(mu, d_dB_.p)) return MI_BA_ERROR;
    {
        float ms = 0;   // (the solve's device time, for tools/pcg_persist_timing.py)
        if (timing_detail >= 2 && hipEventElapsedTime(&ms, ev_[2], ev_[3]) == hipSuccess) { timing.pcg_ms += ms; timing.pcg_timed += 1; }
    }
    const Plan &pl = plan_;
    if (cost_new) *cost_new = h_scal_[SC_COST_NEW];
    if (da) {
        std::vector<double> tmp((size_t)mv_ * kCP);
        if (!hipok(hipMemcpy(tmp.data(), d_dA_.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost), "download dA")) return MI_BA_ERROR;
        for (int j = 0; j < m_; ++j) for (int r = 0; r < cnp; ++r) da[(size_t)j * cnp + r] = h_fixed_eff_[j] ? 0.0 : tmp[slot_(j, r)];
    }
    if (db) {
        const size_t npl = (size_t)pl.npts_local();
        std::vector<double> tmp(npl * 3);
        if (npl && !hipok(hipMemcpy(tmp.data(), d_dB_.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost), "download dB")) return MI_BA_ERROR;
        std::fill(db, db + (size_t)n_ * 3, 0.0);
        for (size_t t = 0; t < npl; ++t) for (int r = 0; r < 3; ++r) db[3 * (size_t)pl.perm[pl.pt_begin + t] + r] = tmp[3 * t + r];
    }
    if (h_scal_[SC_PCG_STATUS] != 0.0 || h_scal_[SC_NOT_PD] != 0.0) { err = "debug_solve: reduced system not positive definite"; return MI_BA_ERROR; }
    return (int)h_scal_[SC_PCG_ITERS];
}

} // namespace miba
