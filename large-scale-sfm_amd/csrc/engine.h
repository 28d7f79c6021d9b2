// engine.h -- host side of the MI355X bundle-adjustment engine: HBM residency, the
// Levenberg-Marquardt controller (sba_levmar.c:848-1617 restated step-wise) and the RCCL exchange.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi_ba.h"
#include "kernels.h"
#include "plan.h"

namespace miba {

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
    }
    template <class A>
    hipError_t upload(const std::vector<T, A> &v)
    {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// layout of the accumulation buffer `acc` (doubles):  [ S_up nb_up*64 | camvec 3*m*8 | scal 48 ]
// all-reduce #1: SUM over [S_up | camvec | scal[0..8)], MAX over scal[8..16)
// all-reduce #2: SUM over scal[32..40), MAX over scal[40..48)
constexpr int kScalCount = 48;

class Engine {
public:
    explicit Engine(const mi_ba_config &c);
    ~Engine();
    bool ok() const { return created_; }

    int set_cameras(int m, const double *a0, const double *R0, const double *f_fixed);
    int set_fixed_cameras(const unsigned char *mask);
    int set_fixed_params(const unsigned char *mask);
    int set_host_allreduce(mi_ba_host_allreduce_fn fn, void *user) { host_ar_ = fn; host_ar_user_ = user; if (fn) persist_ok_ = false; return 0; }   // (ranks that share a GPU: no cooperative launches)
    int set_camera_constraints(const double *weights, const double *values);
    int set_camera_links(int nlink, const int *cam_a, const int *cam_b, const double *weights);
    int set_point_constraints(const double *weight, const double *values);
    int set_points(int n, const double *xyz);
    int set_observations(int64_t nobs, const double *xy, const int *pt_idx, const int *cam_idx);
    int prepare();
    int begin();
    int restart();
    int step();
    int run(double info[MI_BA_INFOSZ]);
    void fill_info(double info[MI_BA_INFOSZ]) const;
    int get_params(double *a, double *xyz);
    int set_params(const double *a, const double *xyz);
    int get_rotations(double *R);
    int debug_linearize(double mu, double *S, double *E, double *ea, double *eb, double *V, double *diagU, double *cost);
    int debug_solve(double mu, double *da, double *db, double *cost_new);
    int comm_init(const char *id);
    int comm_info(int out[4]);
    bool pcg_f32_() const;
    bool pcg_cm_() const;
    int pcg_ell_w_ = 0;                           // row pitch of the CG's fixed-pitch matrix (0: rows by bsr_row): prepare()
    bool ell_pads_dirty_ = false;                 // ... its pads have been written over by a trial in another layout
    int fv_rows_ = 0;                             // partial rows the camera-side vectors of the present linearisation left (engine.cpp linear_phase)
    bool build_part_pending_ = false;             // the build's partial rows of the present linearisation are summed by the trial's k_reduce_final (engine.cpp linear_phase)
    size_t bs_row0_() const { return (size_t)plan_.ntile() + plan_.long_pts.size(); }   // first partial row of the back-substitution
    bool pcg_push_ok_ = false;                    // the reduced solve's push halo (opt-in MI_BA_PCG_PUSH=1: mailboxes allocated, symmetric pattern)
    int pcg_f32_force_ = -1; size_t pcg_f32_from_ = (size_t)64 << 20;   // read from the environment by prepare()
    int debug_blocks(double *U, double *W);

    mi_ba_config cfg;
    int cnp = 7;
    std::string err;
    std::vector<double> trace;   // 6 doubles per trial
    mi_ba_timing timing{};
    int timing_detail = 0;   // mi_ba_set_timing_detail

private:
    bool fail(const std::string &msg) { err = msg; return false; }
    bool hipok(hipError_t e, const char *what);
    bool linear_phase(double mu, bool want_diag, const double *spec = nullptr); // cam_prep .. build .. all-reduce .. finalize .. block_inv
    bool solve_phase(double mu, double *dB_dev, bool may_speculate = false);    // pcg .. cam_update .. backsub .. all-reduce .. scalars on the host
    bool update_phase(double mu, double *dB_dev, bool peek = false);   // peek: k_reduce_final also takes the controller's decision (one rank)
    bool pcg_chunks(int it0);
    PcgArgs pcg_args();
    PcgStart pcg_start();
    PeekArgs peek_args(double mu) const;
    int pcg_pred_ = 0, pcg_pred_prev_ = 0;        // iterations of the previous two solves of this LM run
    bool read_scalars();
    bool allreduce1();
    bool allreduce2();
    void accept_trial();
    double measured_maxdiag() const;
    ModelCfg model() const;

    bool created_ = false, prepared_ = false, begun_ = false;
    hipStream_t st_ = nullptr;
    hipEvent_t evx_[4] = {};     // exchange bracket of a linear phase, [0],[1] and [2],[3] alternating like the build's
    hipEvent_t ev_[10] = {};      // build bracket: [0],[1] and [6],[7] alternate from one linear phase to the next (the next trial's
                                  // may be enqueued before this trial's is read); [2..5] reduced solve, back-substitution; [8] scalars copied
    int lin_count_ = 0, lin_pair_ = 0, spec_pair_ = 0;
    bool lin_timed_[2] = { false, false };   // whether the build of either event pair was bracketed (timing_detail 1 samples)
    // the next trial's linear phase enqueued before the host has seen this trial's outcome (solve_phase)
    bool speculate_ = true, spec_enqueued_ = false, spec_ready_ = false;
    PcgArgs upd_p_{};               // the solve k_cam_update ends (it, last, tail set by solve_phase)
    // two-level preconditioner of the reduced solve (kernels.h CoarseArgs)
    CoarseArgs co_{};               // ng == 0: this problem has none (few cameras, rows reaching too many aggregates, nine parameters)
    bool co_on_ = false;            // switched on for the rest of this LM run (once a solve needed kTwoLevelFrom iterations)
    bool co_active_ = false;        // the last linear phase built the coarse space: the next solve uses it
    int co_force_ = 0;              // MI_BA_TWO_LEVEL: 0 off (default), 1 from the first trial, -1 once a solve ran long (>= kTwoLevelMinCams cameras)
    DevBuf<int> d_bc_agg_;
    DevBuf<int> d_tb_ptr_, d_tb_list_, d_tc_ptr_, d_tc_list_;   // ordered hand-over (deterministic = 1): contributors per block of S / per camera
    DevBuf<double> d_tile_blk_, d_tile_cam_;
    DevBuf<double> d_Ac_, d_Acinv_, d_Bc_, d_U_, d_Lfac_, d_z_, d_cpart_;
    bool cg_start_fresh_ = false;   // the last linear phase's k_block_inv has set up the CG start and no solve has used it yet
    void *comm_ = nullptr; // ncclComm_t
    mi_ba_host_allreduce_fn host_ar_ = nullptr;   // test transport: the exchange through a host callback instead of RCCL
    void *host_ar_user_ = nullptr;
    std::vector<double> h_ar_;
    bool exchange(double *dev, size_t count, int op);   // op 0 sum, 1 max; RCCL or the host callback

    // host problem
    int m_ = 0, n_ = 0;
    int mv_ = 0;               // rows of every camera-side array and of the reduced system: m_, or 2 m_ with nine camera parameters
    bool split_ = false;       // cnp == 9 (kernels.h ModelCfg.split)
    int cnp_cam_() const { return cnp < kCP ? cnp : kCP; }   // parameters per row of the camera-side arrays
    size_t slot_(int j, int r) const { return split_ ? (r < kCP ? (size_t)(2 * j) * kCP + r : (size_t)(2 * j + 1) * kCP) : (size_t)j * kCP + r; }   // parameter r of camera j
    int64_t nobs_ = 0;
    std::vector<double> h_a0_, h_R0_, h_ff_, h_xyz_, h_xy_;
    std::vector<int> h_pt_, h_cam_;
    std::vector<unsigned char> h_fixed_, h_fixed_eff_;
    std::vector<unsigned short> h_pfix_;                          // per camera, bit r = parameter r held constant (r < 9)
    std::vector<double> h_cam_cw_, h_cam_cv_, h_pt_cw_, h_pt_cv_; // soft constraints (empty = none)
    std::vector<int> h_link_cam_;                                 // camera links: 2 ints per link
    std::vector<double> h_link_w_;                                // cnp weights per link
    CamLinks links_{};                                            // device view (row == nullptr: none)
    double c_lag_ = 0.0; // constraint penalty carried with the cost: the one of the PREVIOUS accepted iterate (QUIRK, sba_levmar.c:1498)
    bool have_ff_ = false;
    double pl2_inactive_ = 0.0;
    Plan plan_;

    // device: plan
    DevBuf<int> d_tile_batch_, d_tile_slot_, d_tile_cslot_, d_slot_gblk_, d_cslot_cam_, d_batch_obs_, d_batch_pt_, d_pt_obs_,
        d_obs_cam_, d_obs_pt_, d_bsr_row_, d_bsr_rowidx_, d_bsr_col_, d_bsr_src_, d_bsr_diag_, d_bsr_tq_, d_ell_col_, d_scale_list_;
    DevBuf<long long> d_batch_pair_;
    DevBuf<unsigned char> d_obs_cslot_, d_bsr_trans_, d_fixed_, d_pfix_;   // per ROW of the camera-side arrays: held camera; bit r = parameter r held constant
    DevBuf<unsigned char> d_fixed_real_;                                    // per real camera (the point-side kernels)
    DevBuf<int> d_wk_vblk_, d_wk_diag_vblk_;
    DevBuf<unsigned> d_pair_, d_rpair_;
    DevBuf<int> d_obs_rec_, d_rec_cam_, d_wk_blk_, d_wk_diag_cam_, d_wk_diag_blk_, d_sb_blk_, d_sb_ent_, d_sb8_blk_, d_sb8_ent_;   // row path (plan.h)
    DevBuf<long long> d_wk_q_, d_sb_q_, d_sb8_q_;
    DevBuf<double> d_rec_;                                                             // kRec doubles per row-path observation
    DevBuf<unsigned long long> d_dbg_, d_pstamps_;
    // the reduced solve as one persistent cooperative launch (kernels.hip k_pcg_persist) when S~ fits the chip's LDS
    DevBuf<unsigned long long> d_bar_;       // its grid-barrier words
    DevBuf<int> d_win_ptr_, d_win_col_, d_blk_loc_;   // per workgroup: the distinct columns of its rows; per block: where its column sits among them
    bool persist_ok_ = false;                // usable for this problem; cleared for good when a launch is refused or a barrier gives up
    int persist_rows_ = 0, persist_nwg_ = 0, persist_group_ = 0;
    size_t persist_lds_ = 0;
    void setup_persist();
    DevBuf<int> d_long_pts_;
    DevBuf<double> d_cam_cw_, d_cam_cv_, d_pt_cw_, d_pt_cv_;
    DevBuf<int> d_link_row_, d_link_other_, d_link_blk_;
    DevBuf<unsigned char> d_tile_mode_;
    bool uni_fixed_ = false, uni_other_ = false;   // uniform tiles (plan.h mode 4) with kUniFixedViews views / with another view count
    DevBuf<int> d_bs_batch_obs_, d_bs_batch_pt_, d_bs_batch_obs_end_, d_bs_batch_pt_end_;
    DevBuf<double> d_cam_a_init_, d_pts_init_;
    DevBuf<double> d_link_w_, d_link_pair_w_, d_cam_part_, d_cam_part2_;   // partial rows of k_finalize_vec / of k_cam_update
    LongArgs long_args(double mu, double *dB_dev, int part_row0);
    DevBuf<double2> d_obs_xy_;
    // device: state
    DevBuf<double> d_cam_a_[2], d_cam_lin_[2], d_pts_[2], d_R0_[2] /* [cur_]: the rotation omega is relative to; moves with the camera under PBA's update */, d_R0_init_, d_ff_, d_ptout_, d_acc_, d_S_full_, d_S_pcg_, d_Linv_, d_E_, d_dA_, d_dB_,
        d_pcg_, d_rzpq_, d_full_pts_, d_part_, d_stage_, d_spec_, d_mb_ /* the CG mailboxes (push halo) */;
    int cur_ = 0; // index of the current estimate in the [2] buffers
    bool cam_lin_valid_ = false; // d_cam_lin_[cur_] holds the records of d_cam_a_[cur_]
    double *h_scal_ = nullptr; // pinned, kScalCount
    // d_acc_ = [slots1 | S_up | camvec | scal | slots2].  With several ranks the maxima travel inside the SUM all-reduces:
    // every rank puts its 8 maxima into its own slot of a world x 8 block of zeros next to the summed range, and
    // k_fold_max takes the maximum over the slots afterwards -- one collective per exchange instead of two.
    size_t nslot_() const { return cfg.world > 1 ? (size_t)cfg.world * 8 : 0; }
    double *slots1_() { return d_acc_.p; }
    double *S_up_() { return d_acc_.p + nslot_(); }
    double *camvec_() { return S_up_() + (size_t)plan_.nb_up() * 64; }
    double *scal_() { return camvec_() + (size_t)3 * mv_ * kCP; }
    double *slots2_() { return scal_() + kScalCount; }

    // LM state (names follow sba_levmar.c:596-613)
    double mu_ = 0, p_eL2_ = 0, init_p_eL2_ = 0, eab_inf_ = 0, dp_L2_ = 0, p_L2_ = 0, maxdiag_ = 0;
    int nu_ = 2, itno_ = 0, stop_ = 0, nfev_ = 0, njev_ = 0, nlss_ = 0;
    int lm_variant_cfg_ = 0;        // cfg.lm_variant as configured (PBA's auto-switch, opts[5], turns 1 into 2 inside a solve)
    int trials_ = 0;                // lm_variant 1: passes of PBA's loop (its iteration cap counts trials, SparseBundleCPU.cpp:3843)
    bool new_outer_ = true;
    int nfree_cams_ = 0;
};

} // namespace miba
