// sfm_shim.cpp -- run_sfm / sba_motstr_levmar shaped entry points (include/mi_sfm.h) over mi_ba.h.
// Packing and unpacking follow Bundler's driver: sfm.c:657-720 (pack, options), :881-935 (unpack).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mi_ba.h"
#include "../../include/mi_sfm.h"

namespace {

struct Constraints { const double *cam_w = nullptr, *cam_v = nullptr, *pt_w = nullptr, *pt_v = nullptr; };
struct Blocks { double *V = nullptr, *S = nullptr, *U = nullptr, *W = nullptr; bool any() const { return V || S || U || W; } };

// cnp 6 / 7: pose (+ focal length); undistort: + k1, k2 (cnp 8 / 9), Bundler's projection distortion (sfm.c:360-373, 648-649)
int solve(int n, int m, int mcon, const char *vmask, double *p, int cnp, const double *x, const double *R0,
          const double *f_fixed, int explicit_centers, int itmax, int verbose, const double opts[6], double info[10], int mode = 0,
          const Constraints &cons = Constraints(), int undistort = 0, const Blocks &out = Blocks())
{
    const int est_focal = undistort ? cnp == 9 : cnp == 7;
    if (n <= 0 || m <= 0 || !vmask || !p || !x || !R0 || (undistort ? (cnp != 8 && cnp != 9) : (cnp != 6 && cnp != 7))) {
        std::fprintf(stderr, "mi_sba_motstr_levmar: bad arguments (cnp must be 6 or 7, with distortion 8 or 9)\n");
        return MI_BA_ERROR;
    }
    std::vector<int> pt, cam;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j)
            if (vmask[(size_t)i * m + j]) { pt.push_back(i); cam.push_back(j); }
    mi_ba_config cfg;
    mi_ba_default_config(&cfg);
    cfg.est_focal = est_focal; cfg.explicit_centers = explicit_centers; cfg.mcon = mcon; cfg.proj_sign = -1.0;
    cfg.undistort = undistort ? 1 : 0; cfg.nk = undistort ? 2 : 1;
    cfg.itmax = itmax; cfg.verbose = verbose; cfg.mode = mode;
    if (opts) std::memcpy(cfg.opts, opts, sizeof(cfg.opts));
    mi_ba_engine *e = mi_ba_create(&cfg);
    if (!e) { std::fprintf(stderr, "mi_sba_motstr_levmar: %s\n", mi_ba_last_error(nullptr)); return MI_BA_ERROR; }
    int ret = MI_BA_ERROR;
    if (mi_ba_set_cameras(e, m, p, R0, f_fixed) == 0 && mi_ba_set_points(e, n, p + (size_t)m * cnp) == 0 &&
        mi_ba_set_camera_constraints(e, cons.cam_w, cons.cam_v) == 0 && mi_ba_set_point_constraints(e, cons.pt_w, cons.pt_v) == 0 &&
        mi_ba_set_observations(e, (int64_t)pt.size(), x, pt.data(), cam.data()) == 0 && mi_ba_prepare(e) == 0) {
        ret = mi_ba_run(e, info);
        if (ret != MI_BA_ERROR || true) mi_ba_get_params(e, p, p + (size_t)m * cnp);
    }
    if (ret == MI_BA_ERROR) std::fprintf(stderr, "mi_sba_motstr_levmar: %s\n", mi_ba_last_error(e));
    mi_ba_destroy(e);
    // sba's output hooks (sba_levmar.c:1633-2026): one more linearisation at the final estimate, without damping, on an engine whose every
    // observation leaves its A | B record in HBM (deterministic = 2: the block-major row path) -- S and V from the parity hook, U and W
    // formed on the host from the records
    if (ret != MI_BA_ERROR && out.any()) {
        if (mode != 0) { std::fprintf(stderr, "mi_sba_motstr_levmar: Vout / Sout / Uout / Wout exist for the full bundle only (sba_motstr_levmar_x)\n"); return MI_BA_ERROR; }
        cfg.deterministic = 2;
        mi_ba_engine *b = mi_ba_create(&cfg);
        if (!b) { std::fprintf(stderr, "mi_sba_motstr_levmar: %s\n", mi_ba_last_error(nullptr)); return MI_BA_ERROR; }
        const size_t N = (size_t)m * cnp, nobs = pt.size();
        std::vector<double> Wc(out.W ? nobs * cnp * 3 : 0);
        int rc = MI_BA_ERROR;
        if (mi_ba_set_cameras(b, m, p, R0, f_fixed) == 0 && mi_ba_set_points(b, n, p + N) == 0 &&
            mi_ba_set_camera_constraints(b, cons.cam_w, cons.cam_v) == 0 && mi_ba_set_point_constraints(b, cons.pt_w, cons.pt_v) == 0 &&
            mi_ba_set_observations(b, (int64_t)nobs, x, pt.data(), cam.data()) == 0 && mi_ba_prepare(b) == 0 &&
            mi_ba_debug_linearize(b, 0.0, out.S, nullptr, nullptr, nullptr, out.V, nullptr, nullptr) == 0 &&
            ((!out.U && !out.W) || mi_ba_debug_blocks(b, out.U, out.W ? Wc.data() : nullptr) == 0)) rc = 0;
        if (rc != 0) std::fprintf(stderr, "mi_sba_motstr_levmar: output blocks: %s\n", mi_ba_last_error(b));
        mi_ba_destroy(b);
        if (rc != 0) return MI_BA_ERROR;
        if (out.W) {                                   // dense (m cnp) x (3 n), sba_levmar.c:1829-1842
            std::fill(out.W, out.W + N * 3 * (size_t)n, 0.0);
            for (size_t k = 0; k < nobs; ++k)
                for (int r = 0; r < cnp; ++r)
                    for (int c = 0; c < 3; ++c) out.W[((size_t)cam[k] * cnp + r) * 3 * (size_t)n + 3 * (size_t)pt[k] + c] = Wc[(k * cnp + r) * 3 + c];
        }
    }
    return ret;
}

} // namespace

extern "C" {

int mi_sba_motstr_levmar(int n, int m, int mcon, const char *vmask, double *p, int cnp, const double *x,
                         const double *R0, const double *f_fixed, int explicit_camera_centers,
                         int itmax, int verbose, const double opts[6], double info[10])
{
    return solve(n, m, mcon, vmask, p, cnp, x, R0, f_fixed, explicit_camera_centers, itmax, verbose, opts, info);
}

int mi_sba_motstr_levmar_blocks(int n, int m, int mcon, const char *vmask, double *p, int cnp, const double *x,
                                const double *R0, const double *f_fixed, int explicit_camera_centers,
                                int itmax, int verbose, const double opts[6], double info[10],
                                double *Vout, double *Sout, double *Uout, double *Wout)
{
    Blocks out; out.V = Vout; out.S = Sout; out.U = Uout; out.W = Wout;
    return solve(n, m, mcon, vmask, p, cnp, x, R0, f_fixed, explicit_camera_centers, itmax, verbose, opts, info, 0, Constraints(), 0, out);
}

int mi_run_sfm(int num_pts, int num_cameras, int ncons, char *vmask, double *projections,
               int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
               camera_params_t *cams, v3_t *pts, int use_constraints, int use_point_constraints,
               v3_t *pt_constraints, double pt_constraint_weight, int fix_points, int optimize_for_fisheye,
               double eps2, double *Vout, double *Sout, double *Uout, double *Wout, double info_out[10])
{
    if (const_focal_length || optimize_for_fisheye || ((Vout || Sout || Uout || Wout) && fix_points) ||
        (use_point_constraints && (fix_points || !pt_constraints))) {
        std::fprintf(stderr, "[run_sfm] this build supports est_focal_length / undistort / explicit_camera_centers / ncons / fix_points / "
                             "use_constraints / use_point_constraints / Vout, Sout, Uout, Wout (no fisheye, no const_focal_length; the output "
                             "blocks with the full bundle only, as in the reference)\n");
        return MI_BA_ERROR;
    }
    Blocks blocks; blocks.V = Vout; blocks.S = Sout; blocks.U = Uout; blocks.W = Wout;
    const int m = num_cameras, n = num_pts, cnp = (est_focal_length ? 7 : 6) + (undistort ? 2 : 0), kc = est_focal_length ? 7 : 6;
    const double f_scale = 0.001, k_scale = 5.0; // sfm.c:640-641 (TEST_FOCAL)
    std::vector<double> p((size_t)m * cnp + 3 * (size_t)n), R0((size_t)m * 9), ff(m);
    for (int j = 0; j < m; ++j) {                                   // sfm.c:657-702
        for (int r = 0; r < 3; ++r) p[(size_t)j * cnp + r] = cams[j].t[r];
        for (int r = 3; r < 6; ++r) p[(size_t)j * cnp + r] = 0.0;
        if (est_focal_length) p[(size_t)j * cnp + 6] = cams[j].f * f_scale;
        if (undistort) { p[(size_t)j * cnp + kc] = cams[j].k[0] * k_scale; p[(size_t)j * cnp + kc + 1] = cams[j].k[1] * k_scale; }   // :687-699
        std::memcpy(&R0[(size_t)j * 9], cams[j].R, 9 * sizeof(double));
        ff[j] = cams[j].f;
    }
    for (int i = 0; i < n; ++i) for (int r = 0; r < 3; ++r) p[(size_t)m * cnp + 3 * (size_t)i + r] = pts[i].p[r]; // :704-709
    const double opts[6] = { 1.0e-3, 1.0e-10, eps2, 1.0e-12, 0.0, 4.0e-2 };                                         // :711-719
    // soft constraints exactly as run_sfm hands them to sba (sfm.c:727-787), with TEST_FOCAL's rescaling of the focal entry
    std::vector<double> cw, cv, pw, pv;
    Constraints cons;
    if (use_constraints) {
        cw.assign((size_t)m * cnp, 0.0); cv.assign((size_t)m * cnp, 0.0);
        for (int j = 0; j < m; ++j)
            for (int r = 0; r < cnp; ++r) {
                if (!cams[j].constrained[r]) continue;
                double w = cams[j].weights[r], v = cams[j].constraints[r];
                if (est_focal_length && r == 6) { v *= f_scale; w *= 1.0 / (f_scale * f_scale); }                  // :745-749
                if (undistort && (r == 7 || r == 8)) { v *= k_scale; w *= 1.0 / (k_scale * k_scale); }             // :751-757 (entries 7, 8 as written there)
                cw[(size_t)j * cnp + r] = w; cv[(size_t)j * cnp + r] = v;
            }
        cons.cam_w = cw.data(); cons.cam_v = cv.data();
    }
    if (use_point_constraints) {
        pw.assign(n, 0.0); pv.assign((size_t)n * 3, 0.0);
        for (int i = 0; i < n; ++i) {
            const double *c = pt_constraints[i].p;
            if (c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0) continue;                                                 // :769-777
            pw[i] = pt_constraint_weight;
            for (int r = 0; r < 3; ++r) pv[(size_t)i * 3 + r] = c[r];
        }
        cons.pt_w = pw.data(); cons.pt_v = pv.data();
    }
    double info[10] = { 0 };
    const int ret = solve(n, m, ncons, vmask, p.data(), cnp, projections, R0.data(), ff.data(), explicit_camera_centers,
                          150 /* MAX_ITERS, sfm.c:820 */, 0, opts, info, fix_points ? 1 : 0 /* sba_mot_levmar, sfm.c:845-852 */, cons, undistort, blocks);
    if (info_out) std::memcpy(info_out, info, sizeof(info));
    if (ret == MI_BA_ERROR) return ret;
    std::printf("[run_sfm] Number of iterations: %d\n", (int)info[5]);
    std::printf("info[6] = %0.3f\n", info[6]);
    // unpack, sfm.c:881-935: R <- rot_update(R0, w) is what the engine's record holds
    mi_ba_config cfg; mi_ba_default_config(&cfg);
    for (int j = 0; j < m; ++j) {
        const double *a = &p[(size_t)j * cnp];
        for (int r = 0; r < 3; ++r) cams[j].t[r] = a[r];
        // rot_update(R0, w), sfm.c:78-117
        const double w0 = a[3], w1 = a[4], w2 = a[5];
        const double th = std::sqrt(w0 * w0 + w1 * w1 + w2 * w2);
        if (th != 0.0) {
            const double nx[9] = { 0, -w2 / th, w1 / th, w2 / th, 0, -w0 / th, -w1 / th, w0 / th, 0 };
            double dR[9], Rn[9];
            const double s = std::sin(th), omc = 1.0 - std::cos(th);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
                const double sq = nx[3 * r] * nx[c] + nx[3 * r + 1] * nx[3 + c] + nx[3 * r + 2] * nx[6 + c];
                dR[3 * r + c] = (r == c ? 1.0 : 0.0) + s * nx[3 * r + c] + omc * sq;
            }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
                Rn[3 * r + c] = dR[3 * r] * cams[j].R[c] + dR[3 * r + 1] * cams[j].R[3 + c] + dR[3 * r + 2] * cams[j].R[6 + c];
            std::memcpy(cams[j].R, Rn, sizeof(Rn));
        }
        if (est_focal_length) cams[j].f = a[6] / f_scale;
        if (undistort) { cams[j].k[0] = a[kc] / k_scale; cams[j].k[1] = a[kc + 1] / k_scale; }               // :911-920
    }
    for (int i = 0; i < n; ++i) for (int r = 0; r < 3; ++r) pts[i].p[r] = p[(size_t)m * cnp + 3 * (size_t)i + r];
    return ret;
}

void run_sfm(int num_pts, int num_cameras, int ncons, char *vmask, double *projections,
             int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
             camera_params_t *init_camera_params, v3_t *init_pts, int use_constraints, int use_point_constraints,
             v3_t *points_constraints, double point_constraint_weight, int fix_points, int optimize_for_fisheye,
             double eps2, double *Vout, double *Sout, double *Uout, double *Wout)
{
    (void)mi_run_sfm(num_pts, num_cameras, ncons, vmask, projections, est_focal_length, const_focal_length, undistort,
                     explicit_camera_centers, init_camera_params, init_pts, use_constraints, use_point_constraints,
                     points_constraints, point_constraint_weight, fix_points, optimize_for_fisheye, eps2,
                     Vout, Sout, Uout, Wout, nullptr);
}

} // extern "C"
