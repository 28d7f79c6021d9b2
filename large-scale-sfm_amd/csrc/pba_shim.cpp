// pba_shim.cpp -- class ParallelBA (include/mi_pba.h) over the C ABI of include/mi_ba.h.
// Mirrors the call sequence of the reference plugin (pba.cpp:27-154 forwarding to
// SparseBundleCPU.cpp:2541-2674); the numbers come from the HIP engine.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi_ba.h"
#include "../../include/mi_pba.h"
#include "focal_links.h"

ConfigBA::ConfigBA()
    : __lm_max_iteration(50), __lm_mse_threshold(0.25f), __bundle_time_budget(0), __bundle_mode_next(0), __cg_max_iteration(100), __cg_norm_threshold(1e-3f), __fixed_intrinsics(false),
      __use_radial_distortion(0), __verbose_level(0), __selected_device(0), __bundle_return_code(0), __num_lm_success(0),
      __num_lm_iteration(0), __initial_mse(0), __final_mse(0), __focal_normalize(true), __depth_normalize(true), __depth_degeneracy_fix(true),
      __honour_data_normalize(false), __data_normalize_median(0.5f), __depth_check_epsilon(0.01f), __focal_scaling(1.0f), __depth_scaling(1.0f),
      __num_point_behind(0), __num_camera_modified(0), __lm_initial_damp(1e-3f), __lm_minimum_damp(1e-10f), __lm_maximum_damp(1e+5f),
      __lm_delta_threshold(1e-6f), __lm_gradient_threshold(1e-10f), __lm_damping_auto_switch(0), __lm_use_diagonal_damp(true),
      __lm_check_gradient(false), __cg_min_iteration(10)
{
    __output_param[0] = 0;
}

// the reference's string-hashed option parser (ConfigBA.cpp:318-541), the options that apply here
void ConfigBA::ParseParam(int argc, char **argv)
{
    for (int i = 0; i < argc; ++i) {
        const char *a = argv[i];
        if (!a || a[0] != '-') continue;
        const char *v = (i + 1 < argc) ? argv[i + 1] : nullptr;
        if (!std::strcmp(a, "-lmi") && v) { __lm_max_iteration = std::atoi(v); ++i; }
        else if (!std::strcmp(a, "-lme") && v) { if (std::atof(v) >= 0) __lm_mse_threshold = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-budg") && v) { if (std::atoi(v) >= 0) __bundle_time_budget = std::atoi(v); ++i; }
        else if (!std::strcmp(a, "-md") || !std::strcmp(a, "-radi")) __use_radial_distortion = -1;
        else if (!std::strcmp(a, "-pd")) __use_radial_distortion = 1;
        else if (!std::strcmp(a, "-vari")) __fixed_intrinsics = false;
        else if (!std::strcmp(a, "-moti")) __bundle_mode_next = 1;
        else if (!std::strcmp(a, "-stru")) __bundle_mode_next = 2;
        else if (!std::strcmp(a, "-cgi") && v) { __cg_max_iteration = std::atoi(v); ++i; }
        else if (!std::strcmp(a, "-cgim") && v) { if (std::atoi(v) > 0) __cg_min_iteration = std::atoi(v); ++i; }        // ConfigBA.cpp:376-413
        else if (!std::strcmp(a, "-lmd") && v) { if (std::atof(v) >= 0) __lm_delta_threshold = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-lmg") && v) { if (std::atof(v) > 0) __lm_gradient_threshold = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-damp") && v) { if (std::atof(v) > 0) __lm_initial_damp = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-dmin") && v) { if (std::atof(v) > 0) __lm_minimum_damp = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-dmax") && v) { if (std::atof(v) > 0) __lm_maximum_damp = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-id")) __lm_use_diagonal_damp = false;                                              // :473-482
        else if (!std::strcmp(a, "-das")) {   // :477-480 switch to identity damping once a failed trial's damping passes this (default 2, at least 0.1)
            float f = 0;
            if (v && std::sscanf(v, "%f", &f) == 1 && f > 0) __lm_damping_auto_switch = f > 0.1f ? f : 0.1f; else __lm_damping_auto_switch = 2.0f;
        }
        else if (!std::strcmp(a, "-chkg")) __lm_check_gradient = true;
        else if (!std::strcmp(a, "-cgn") && v) { __cg_norm_threshold = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-dev") && v) { __selected_device = std::atoi(v); ++i; }
        else if (!std::strcmp(a, "-out") && v) { std::snprintf(__output_param, sizeof(__output_param), "%s", v); ++i; }
        else if (!std::strcmp(a, "-v") && v) { __verbose_level = std::atoi(v); ++i; }
        else if (!std::strcmp(a, "-cali")) __fixed_intrinsics = true;
        else if (!std::strcmp(a, "-nodn")) { __depth_normalize = false; __focal_normalize = false; }       // ConfigBA.cpp:503-506
        else if (!std::strcmp(a, "-nodf")) __depth_degeneracy_fix = false;                                  // :507-509
        else if (!std::strcmp(a, "-norm") && v) { if (std::atof(v) > 0) __data_normalize_median = (float)std::atof(v); ++i; }   // :510-512
        else if (!std::strcmp(a, "-dce") && v) { if (std::atof(v) > 0 && std::atof(v) <= 0.01) __depth_check_epsilon = (float)std::atof(v); ++i; }
        else if (!std::strcmp(a, "-dn")) { __honour_data_normalize = true; __depth_normalize = true; __focal_normalize = true; }   // ours: see mi_pba.h
    }
}

struct ParallelBA::Impl {
    ConfigBA cfg;
    int device = 0;
    bool valid_device = true;
    CameraT *cams = nullptr; size_t ncam = 0;
    Point3D *pts = nullptr; size_t npt = 0;
    const Point2D *meas = nullptr; const int *ptidx = nullptr, *camidx = nullptr; size_t nproj = 0;
    const int *fmask = nullptr; float fweight = 0.0f;
    std::atomic<float> mse{0.0f};
    std::atomic<int> iteration{0};
    std::atomic<bool> abort{false};
};

ParallelBA::ParallelBA(DeviceT device) : _optimizer(new Impl)
{
    _optimizer->valid_device = device != PBA_INVALID_DEVICE;
    _optimizer->device = device >= 0 ? (int)device : 0;
    _optimizer->cfg.__selected_device = _optimizer->device;
}
ParallelBA::~ParallelBA() { delete _optimizer; }
void *ParallelBA::operator new(size_t size) { return std::malloc(size); }
void ParallelBA::operator delete(void *p) { std::free(p); }

void ParallelBA::ParseParam(int narg, char **argv) { _optimizer->cfg.ParseParam(narg, argv); _optimizer->device = _optimizer->cfg.__selected_device; }
ConfigBA *ParallelBA::GetInternalConfig() { return &_optimizer->cfg; }
void ParallelBA::SetFixedIntrinsics(bool fixed) { _optimizer->cfg.__fixed_intrinsics = fixed; }
void ParallelBA::EnableRadialDistortion(DistortionT type) { _optimizer->cfg.__use_radial_distortion = (int)type; }
void ParallelBA::SetNextTimeBudget(int seconds) { _optimizer->cfg.__bundle_time_budget = seconds; }
void ParallelBA::ReserveStorage(size_t, size_t, size_t) {}
void ParallelBA::SetCameraData(size_t ncam, CameraT *cams) { _optimizer->ncam = ncam; _optimizer->cams = cams; }
void ParallelBA::SetPointData(size_t npoint, Point3D *pts) { _optimizer->npt = npoint; _optimizer->pts = pts; }
void ParallelBA::SetProjection(size_t nproj, const Point2D *imgpts, const int *point_idx, const int *cam_idx)
{ _optimizer->nproj = nproj; _optimizer->meas = imgpts; _optimizer->ptidx = point_idx; _optimizer->camidx = cam_idx; }
void ParallelBA::SetNextBundleMode(BundleModeT mode) { _optimizer->cfg.__bundle_mode_next = (int)mode; }
float ParallelBA::GetMeanSquaredError() { return _optimizer->mse.load(); }
void ParallelBA::AbortBundleAdjustment() { _optimizer->abort.store(true); }
int ParallelBA::GetCurrentIteration() { return _optimizer->iteration.load(); }
// shared-intrinsics groups: fmask[i] = index of the group's first camera (fmask[iq] == iq), -1 = on its own
// (pba.h:124, SparseBundleCPU.cpp:2550-2554); the pointer is kept like every other input array
void ParallelBA::SetFocalMask(const int *fmask, float weight) { _optimizer->fmask = fmask; _optimizer->fweight = weight; }

int ParallelBA::RunBundleAdjustment()
{
    Impl &o = *_optimizer;
    ConfigBA &c = o.cfg;
    c.__bundle_return_code = 0; c.__num_lm_success = c.__num_lm_iteration = 0;
    o.abort.store(false); o.iteration.store(0);
    // silent "nothing done" on missing input, SparseBundleCPU.cpp:2587-2590
    if (!o.valid_device || !o.cams || !o.pts || !o.meas || !o.ptidx || !o.camidx || !o.ncam || !o.npt || !o.nproj) return 0;
    // EnableRadialDistortion: +1 distorts the projection, -1 the measurement (pba.h:73-84); with fixed intrinsics the
    // coefficient is applied but not estimated (JacobianOne, SparseBundleCPU.cpp:1362-1363, 1427)
    const int dist = c.__use_radial_distortion;
    bool fixed_projection_distortion = false;   // fixed intrinsics with a projection distortion that is not zero: the coefficient is a
    if (dist > 0 && c.__fixed_intrinsics)       // parameter the engine holds constant (mi_ba_set_fixed_params)
        for (size_t j = 0; j < o.ncam; ++j) fixed_projection_distortion |= o.cams[j].radial != 0;
    mi_ba_config cfg;
    mi_ba_default_config(&cfg);
    cfg.est_focal = c.__fixed_intrinsics ? 0 : 1;
    cfg.undistort = (dist != 0 && !c.__fixed_intrinsics) ? (dist > 0 ? 1 : 2) : fixed_projection_distortion ? 1 : 0;
    if (cfg.undistort == 2) { // the coefficient of |x|^2 (pixels^2) is ~1e-7: scale its parameter by f^2 so that the additive damping sees a balanced column
        std::vector<float> fv(o.ncam);
        for (size_t j = 0; j < o.ncam; ++j) fv[j] = o.cams[j].f;
        std::nth_element(fv.begin(), fv.begin() + fv.size() / 2, fv.end());
        const double fm = fv[fv.size() / 2];
        if (fm > 0) cfg.k_scale = 5.0 * fm * fm;
    }
    cfg.explicit_centers = 0;              // CameraT holds T = -R C (DataInterface.h:51)
    cfg.proj_sign = 1.0;                   // +z forward
    cfg.itmax = c.__lm_max_iteration;
    cfg.opts[5] = 0.0;                     // Bundler's "residuals changed by < 4 %" exit (stop 8) is not a PBA rule
    cfg.pcg_maxit = c.__cg_max_iteration; cfg.pcg_tol = c.__cg_norm_threshold;
    // PBA's own Levenberg-Marquardt (NonlinearOptimizeLM, SparseBundleCPU.cpp:3797-3983) unless MI_PBA_SBA_LM=1 asks for sba's:
    // damping on the diagonals started at __lm_initial_damp, its accept rule, gain ratio, damping bounds, step and gradient
    // stops, the iteration cap counting trials, the CG's minimum iteration count (mi_ba.h lm_variant)
    const char *sba_lm = std::getenv("MI_PBA_SBA_LM");
    if (!(sba_lm && sba_lm[0] == '1')) {
        cfg.lm_variant = c.__lm_use_diagonal_damp ? 1 : 2;
        cfg.opts[0] = c.__lm_initial_damp;
        cfg.opts[1] = c.__lm_check_gradient ? c.__lm_gradient_threshold : 0.0;
        cfg.opts[2] = c.__lm_delta_threshold;
        cfg.opts[3] = c.__lm_minimum_damp;
        cfg.opts[4] = (c.__lm_damping_auto_switch == 0 && c.__lm_use_diagonal_damp) ? c.__lm_maximum_damp : 0.0;   // :3950
        cfg.opts[5] = (c.__lm_damping_auto_switch > 0 && c.__lm_use_diagonal_damp) ? c.__lm_damping_auto_switch : 0.0; // :3962-3966, the switch to mu I
        cfg.pcg_minit = c.__cg_min_iteration;
    }
    cfg.device = o.device; cfg.verbose = c.__verbose_level >= 3 ? 2 : 0;   // level 2 prints the reference-style lines below, 3 the engine's too
    cfg.mode = c.__bundle_mode_next == 1 ? 1 : c.__bundle_mode_next == 2 ? 2 : 0;        // SetNextBundleMode, pba.h:103
    c.__bundle_mode_next = 0;                                                             // ResetTemporarySetting, ConfigBA.cpp:149-157
    // SetFocalMask: soft equality of focal length (and distortion) inside a group, as a ring of links between its
    // cameras (ProcessIndexCameraQ, SparseBundleCPU.cpp:2781-2851), the group's values first set to their means and
    // the weight sqrt(projections per camera of the group) * weight (ProcessWeightCameraQ, :2854-2892)
    miba::FocalLinks fl;
    std::vector<double> link_wr;
    if (o.fmask && o.fweight > 0 && cfg.est_focal && cfg.mode != 2) {
        const int mm = (int)o.ncam;
        if (!miba::focal_mask_links(mm, o.fmask, o.nproj, o.camidx, o.fweight, fl))
            std::fprintf(stderr, "[mi_pba] Error: incorrect constraints\n");   // the reference drops the mask and goes on
        else {
            miba::focal_mask_average(mm, o.fmask, &o.cams[0].f, sizeof(CameraT) / sizeof(float));
            miba::focal_mask_average(mm, o.fmask, &o.cams[0].radial, sizeof(CameraT) / sizeof(float));
            for (size_t q = 0; q < fl.a.size(); ++q) link_wr.push_back(dist != 0 ? fl.w[q] * o.cams[fl.rep[q]].f : 0.0);
        }
    }
    const int m = (int)o.ncam, n = (int)o.npt, cnp = 6 + cfg.est_focal + (cfg.undistort ? 1 : 0), kslot = 6 + cfg.est_focal;
    std::vector<double> a0((size_t)m * cnp, 0.0), R0((size_t)m * 9), ff(m), xyz((size_t)n * 3), xy(o.nproj * 2);
    std::vector<unsigned char> fixed(m, 0);
    bool some_fixed_intrinsics = false;
    for (int j = 0; j < m; ++j) {
        const CameraT &cam = o.cams[j];
        for (int r = 0; r < 3; ++r) a0[(size_t)j * cnp + r] = cam.t[r];
        if (cfg.est_focal) a0[(size_t)j * cnp + 6] = (double)cam.f * cfg.f_scale;
        if (cfg.undistort) a0[(size_t)j * cnp + kslot] = (double)cam.radial * cfg.k_scale;   // PBA reads c[13] whatever its type tag
        for (int r = 0; r < 9; ++r) R0[(size_t)j * 9 + r] = cam.m[r / 3][r % 3];
        ff[j] = cam.f;
        fixed[j] = cam.constant_camera == 1.0f;
        some_fixed_intrinsics |= cam.constant_camera == 2.0f;
    }
    if (some_fixed_intrinsics && cfg.est_focal)
        std::fprintf(stderr, "[mi_pba] per-camera SetFixedIntrinsic() is ignored; use SetFixedIntrinsics(true) for all cameras\n");
    for (int i = 0; i < n; ++i) for (int r = 0; r < 3; ++r) xyz[(size_t)i * 3 + r] = o.pts[i].xyz[r];
    for (size_t k = 0; k < o.nproj; ++k) { xy[2 * k] = o.meas[k].x; xy[2 * k + 1] = o.meas[k].y; }
    if (dist < 0 && c.__fixed_intrinsics) // a fixed measurement distortion only rescales the measurements
        for (size_t k = 0; k < o.nproj; ++k) {
            const int j = o.camidx[k];
            if (j < 0 || j >= m) continue;      // reported by mi_ba_prepare
            const double g = 1.0 + (double)o.cams[j].radial * (xy[2 * k] * xy[2 * k] + xy[2 * k + 1] * xy[2 * k + 1]);
            xy[2 * k] *= g; xy[2 * k + 1] *= g;
        }
    // ---- data normalisation, NormalizeDataD then NormalizeDataF (SparseBundleCPU.cpp:2652-2657, 3211-3306, 3141-3209) on the copies
    // the engine gets; undone after the solve (DenormalizeData, :3311-3345)
    c.__focal_scaling = c.__depth_scaling = 1.0f; c.__num_point_behind = c.__num_camera_modified = 0;
    if (!c.__honour_data_normalize) c.__depth_normalize = c.__focal_normalize = false;     // the reference's copy: :2583-2584
    double depth_scaling = 1.0, focal_scaling = 1.0;
    {
        const double dist_bound = 1.0;
        std::vector<double> oz(o.nproj), cpdist1(m, dist_bound), cpdist2(m, -dist_bound);
        int bad_point_count = 0;
        for (size_t k = 0; k < o.nproj; ++k) {
            const int j = o.camidx[k], i = o.ptidx[k];
            if (j < 0 || j >= m || i < 0 || i >= n) { oz[k] = 1.0; continue; }   // reported by mi_ba_prepare
            const double *R = &R0[(size_t)j * 9], *X = &xyz[(size_t)i * 3], *t = &a0[(size_t)j * cnp];
            oz[k] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
            if (std::fabs(oz[k] / t[2]) < c.__depth_check_epsilon) {               // a point close to the camera's plane, :3234-3253
                ++bad_point_count;
                const double px = ff[j] * (R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0]), py = ff[j] * (R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1]);
                const double mx = o.meas[k].x, my = o.meas[k].y;
                const bool checkx = std::fabs(mx) > std::fabs(my);
                if ((checkx && px * oz[k] * mx < 0 && std::fabs(mx) > 64) || (!checkx && py * oz[k] * my < 0 && std::fabs(my) > 64)) {
                    if (oz[k] > 0) cpdist2[j] = 0; else cpdist1[j] = 0;
                }
                if (oz[k] >= 0) cpdist1[j] = std::min(cpdist1[j], oz[k]); else cpdist2[j] = std::max(cpdist2[j], oz[k]);
            }
            if (oz[k] < 0) c.__num_point_behind++;
        }
        if (bad_point_count > 0 && c.__depth_degeneracy_fix) {                     // :3258-3263
            if (!c.__focal_normalize || !c.__depth_normalize) std::printf("Enable data normalization on degeneracy\n");
            c.__focal_normalize = c.__depth_normalize = true;
        }
        if (c.__depth_normalize) {
            std::nth_element(oz.begin(), oz.begin() + o.nproj / 2, oz.end());
            const double oz_median = oz[o.nproj / 2];
            const double shift_min = std::min(oz_median * 0.001, 1.0), dist_threshold = shift_min * 0.1;
            depth_scaling = (1.0 / oz_median) / c.__data_normalize_median;
            if (c.__verbose_level > 2) std::printf("Depth normalized by %.9g (%.9g)\n", depth_scaling, oz_median);   // the reference's line, :3272
            for (int j = 0; j < m; ++j) {
                double *t = &a0[(size_t)j * cnp];
                if (c.__depth_degeneracy_fix && (cpdist1[j] < dist_threshold || cpdist2[j] > -dist_threshold)) {   // move the camera a little, :3279-3291
                    t[2] += std::max(shift_min, std::fabs(t[2] * 1.1920929e-7));
                    c.__num_camera_modified++;
                }
                t[0] *= depth_scaling; t[1] *= depth_scaling; t[2] *= depth_scaling;
            }
            for (double &v : xyz) v *= depth_scaling;
        }
        if (c.__num_point_behind > 0) std::printf("WARNING: %d points are behind camras.\n", c.__num_point_behind);       // the reference's own words
        if (c.__num_camera_modified > 0) std::printf("WARNING: %d camera moved to avoid degeneracy.\n", c.__num_camera_modified);
        if (c.__focal_normalize) {
            std::vector<double> fv(ff);
            std::nth_element(fv.begin(), fv.begin() + m / 2, fv.end());
            const double median_f = fv[m / 2];
            focal_scaling = c.__data_normalize_median / median_f;
            if (c.__verbose_level > 2) std::printf("Focal length normalized by %.9g\n", focal_scaling);                 // :3180
            const double radial_factor = median_f * median_f * 4.0;                // = 1 / focal_scaling^2 at the default median, :3157
            (void)radial_factor;
            for (double &v : xy) v *= focal_scaling;
            for (double &v : ff) v *= focal_scaling;
            // the engine's camera parameters are f * f_scale and k * k_scale: instead of scaling the parameters, scale the
            // scales -- a6 and a7 keep their values (and their O(1) size, which the additive damping wants), the focal lengths
            // they stand for are the normalised ones, and a measurement distortion, which multiplies |x|^2, the normalised
            // k / focal_scaling^2 = k * radial_factor
            cfg.f_scale /= focal_scaling;
            if (cfg.undistort == 2) cfg.k_scale *= focal_scaling * focal_scaling;
        }
        c.__focal_scaling = (float)focal_scaling; c.__depth_scaling = (float)depth_scaling;
    }
    const double mse_unscale = 1.0 / (focal_scaling * focal_scaling);               // GetMeanSquaredError, :2573-2576
    mi_ba_engine *e = mi_ba_create(&cfg);
    if (!e) { std::fprintf(stderr, "[mi_pba] %s\n", mi_ba_last_error(nullptr)); c.__bundle_return_code = 'E'; return 0; }
    if (mi_ba_cnp(e) != cnp) { std::fprintf(stderr, "[mi_pba] internal: camera block size\n"); c.__bundle_return_code = 'E'; mi_ba_destroy(e); return 0; }
    int ret = 0;
    double info[MI_BA_INFOSZ] = { 0 };
    std::vector<double> link_w(fl.a.size() * (size_t)cnp, 0.0);
    for (size_t q = 0; q < fl.a.size(); ++q) {   // internal units: a6 = f * f_scale, a7 = k * k_scale
        link_w[q * cnp + 6] = fl.w[q] * fl.w[q] / (cfg.f_scale * cfg.f_scale);
        if (cfg.undistort && cfg.est_focal) link_w[q * cnp + 7] = link_wr[q] * link_wr[q] / (cfg.k_scale * cfg.k_scale);
    }
    std::vector<unsigned char> frozen;
    if (fixed_projection_distortion) { frozen.assign((size_t)m * cnp, 0); for (int j = 0; j < m; ++j) frozen[(size_t)j * cnp + kslot] = 1; }
    if (mi_ba_set_cameras(e, m, a0.data(), R0.data(), ff.data()) || mi_ba_set_fixed_cameras(e, fixed.data()) ||
        mi_ba_set_fixed_params(e, frozen.empty() ? nullptr : frozen.data()) ||
        mi_ba_set_camera_links(e, (int)fl.a.size(), fl.a.data(), fl.b.data(), link_w.data()) ||
        mi_ba_set_points(e, n, xyz.data()) || mi_ba_set_observations(e, (int64_t)o.nproj, xy.data(), o.ptidx, o.camidx) ||
        mi_ba_prepare(e) || mi_ba_begin(e)) {
        std::fprintf(stderr, "[mi_pba] %s\n", mi_ba_last_error(e));
        c.__bundle_return_code = 'E';
        mi_ba_destroy(e);
        return 0;
    }
    mi_ba_get_info(e, info);
    c.__initial_mse = (float)(info[0] / (double)o.nproj * mse_unscale);
    o.mse.store(c.__initial_mse);
    int st = 0;
    char early = 0;
    const auto t_start = std::chrono::steady_clock::now();
    while (!o.abort.load()) {
        st = mi_ba_step(e);
        if (st == MI_BA_ERROR) break;
        mi_ba_get_info(e, info);
        o.mse.store((float)(info[1] / (double)o.nproj * mse_unscale));
        o.iteration.store((int)info[5]);
        c.__num_lm_iteration = (int)info[9];
        if (st == 1) {
            c.__num_lm_success++;
            if (c.__verbose_level > 1)   // the reference's per-iteration line, SparseBundleCPU.cpp:3910-3919 (iteration, PCG steps, MSE)
                std::printf("#%-3d %3d  e=%g\n", c.__num_lm_success - 1, (int)c.__num_lm_iteration, (double)o.mse.load());
            // the reference's tests after every successful step (SparseBundleCPU.cpp:3919-3935)
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            if (c.__bundle_time_budget > 0 && secs > c.__bundle_time_budget) { early = 'T'; break; }
            if (o.mse.load() <= c.__lm_mse_threshold) { early = 'M'; break; }
        }
        if (st == 2) break;
    }
    if (c.__verbose_level >= 3)
        std::printf("[mi_pba] engine stop %d after %d trials, %d successful; |dp|^2 %g, cost %g -> %g\n", (int)info[6], (int)info[9], c.__num_lm_success, info[3], info[0], info[1]);
    c.__bundle_time_budget = 0;          // ResetTemporarySetting, ConfigBA.cpp:149-157
    if (st == MI_BA_ERROR) { std::fprintf(stderr, "[mi_pba] %s\n", mi_ba_last_error(e)); c.__bundle_return_code = 'E'; }
    else if (early) c.__bundle_return_code = early;
    else {
        const int stop = (int)info[6];   // sba stop reasons -> PBA return codes (SparseBundleCPU.cpp:3851-3935)
        c.__bundle_return_code = stop == 7 ? 'E' : (stop == 2 || stop == 8) ? 'S' : stop == 1 ? 'G' : 0;
    }
    // write the result back into the caller's arrays (TransferDataToHost, SparseBundleCPU.cpp:2660-2674)
    std::vector<double> a((size_t)m * cnp), R((size_t)m * 9);
    if (mi_ba_get_params(e, a.data(), xyz.data()) == 0 && mi_ba_get_rotations(e, R.data()) == 0) {
        for (int j = 0; j < m; ++j) {
            CameraT &cam = o.cams[j];
            // DenormalizeData, SparseBundleCPU.cpp:3311-3345
            for (int r = 0; r < 3; ++r) cam.t[r] = (float)(a[(size_t)j * cnp + r] / depth_scaling);
            for (int r = 0; r < 9; ++r) cam.m[r / 3][r % 3] = (float)R[(size_t)j * 9 + r];
            if (cfg.est_focal) cam.f = (float)(a[(size_t)j * cnp + 6] / cfg.f_scale / focal_scaling);   // f_scale carries the normalisation
            if (cfg.undistort == 1 && !fixed_projection_distortion) cam.SetProjectionDistortion(a[(size_t)j * cnp + kslot] / cfg.k_scale);
            if (cfg.undistort == 2) cam.SetMeasumentDistortion(a[(size_t)j * cnp + kslot] / cfg.k_scale * focal_scaling * focal_scaling);
        }
        for (int i = 0; i < n; ++i) for (int r = 0; r < 3; ++r) o.pts[i].xyz[r] = (float)(xyz[(size_t)i * 3 + r] / depth_scaling);
        ret = c.__num_lm_success;
    }
    c.__final_mse = o.mse.load();
    mi_ba_destroy(e);
    return ret;
}

extern "C" ParallelBA *NewParallelBA(ParallelBA::DeviceT device) { return new ParallelBA(device); }
extern "C" int ParallelBA_GetVersion() { return 105; } // pba.cpp:151-154
