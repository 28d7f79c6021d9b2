/*
 * mi_sfm.h -- the reference's C solver entry points over the MI355X engine.
 *
 *   run_sfm()                 same signature as Bundler's driver, SiftGPU/ba_libs/lib/sfm-driver/sfm.h:68-86
 *                             (the nearest surviving source of the missing Optimizer/main_all.cpp; the
 *                             globals of Optimizer/utils.h:140-152 name exactly these arguments)
 *   mi_run_sfm()              the same, returning a status instead of void
 *   mi_sba_motstr_levmar()    sba_motstr_levmar (Optimizer/sba-1.5/sba.h:95-108) with the projection
 *                             fixed to the one run_sfm passes (sfm_project_point3, sfm.c:509-558):
 *                             a device cannot call back into a host `proj` function pointer.
 * camera_params_t / v3_t are layout-compatible with sfm.h:29-51 and matrix/vector.h:65-67.
 *
 * Supported: est_focal_length 0/1, undistort 0/1 (k1 and k2 as parameters, sfm.c:648-649: 8 or 9 parameters per camera),
 * explicit_camera_centers 0/1, ncons >= 0, fix_points (motion-only, sba_mot_levmar), use_constraints,
 * use_point_constraints (Bundler's soft constraints).
 * Not supported: optimize_for_fisheye, const_focal_length ("not implemented" in the reference itself, sfm.c:394-397), the
 * Vout/Sout/Uout/Wout exports -- asking for any of them makes mi_run_sfm return MI_BA_ERROR (run_sfm prints the reason
 * and returns).
 */
#ifndef MI_SFM_H
#define MI_SFM_H
#ifdef __cplusplus
extern "C" {
#endif

#define NUM_CAMERA_PARAMS 9
#define POLY_INVERSE_DEGREE 6

typedef struct { double p[3]; } v3_t;
#ifndef Vx
#define Vx(v) (v).p[0]
#define Vy(v) (v).p[1]
#define Vz(v) (v).p[2]
#endif

typedef struct {
    double R[9];
    double t[3];                 /* translation, or camera centre when explicit_camera_centers */
    double f;
    double k[2];
    double k_inv[POLY_INVERSE_DEGREE];
    char constrained[NUM_CAMERA_PARAMS];
    double constraints[NUM_CAMERA_PARAMS];
    double weights[NUM_CAMERA_PARAMS];
    double K_known[9];
    double k_known[5];
    char fisheye;
    char known_intrinsics;
    double f_cx, f_cy;
    double f_rad, f_angle;
    double f_focal;
    double f_scale, k_scale;
} camera_params_t;

void run_sfm(int num_pts, int num_cameras, int ncons, char *vmask, double *projections,
             int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
             camera_params_t *init_camera_params, v3_t *init_pts, int use_constraints, int use_point_constraints,
             v3_t *points_constraints, double point_constraint_weight, int fix_points, int optimize_for_fisheye,
             double eps2, double *Vout, double *Sout, double *Uout, double *Wout);

int mi_run_sfm(int num_pts, int num_cameras, int ncons, char *vmask, double *projections,
               int est_focal_length, int const_focal_length, int undistort, int explicit_camera_centers,
               camera_params_t *init_camera_params, v3_t *init_pts, int use_constraints, int use_point_constraints,
               v3_t *points_constraints, double point_constraint_weight, int fix_points, int optimize_for_fisheye,
               double eps2, double *Vout, double *Sout, double *Uout, double *Wout, double info[10]);

/* n points, m cameras, first mcon fixed, vmask n*m, p = (a_1..a_m, b_1..b_n) updated in place, cnp 6|7,
 * x = compact (i,j)-ordered measurements (sba.h:95-108).  R0 m*9 and f_fixed m describe the camera
 * model of sfm_project_point3; returns iterations or MI_BA_ERROR, info[] as sba. */
int mi_sba_motstr_levmar(int n, int m, int mcon, const char *vmask, double *p, int cnp, const double *x,
                         const double *R0, const double *f_fixed, int explicit_camera_centers,
                         int itmax, int verbose, const double opts[6], double info[10]);

/* the same with sba's four output hooks (sba.h:95-108 "Vout, Sout, Uout, Wout"; filled by sba_motstr_levmar_x after the solve from a
 * linearisation at the final estimate WITHOUT damping, sba_levmar.c:1633-2026).  Any of them may be NULL.  Layouts as the reference's:
 *   Vout  n * 9        V_i = sum_j B_ij^T B_ij (+ point constraints), full symmetric 3 x 3          (:1758-1769)
 *   Uout  m * cnp^2    U_j = sum_i A_ij^T A_ij (+ camera constraints on the diagonal), row-major   (:1694-1695)
 *   Wout  (m cnp) x (3 n) dense row-major: Wout[(j cnp + r) 3 n + 3 i + c] = W_ij[r][c] = (A_ij^T B_ij)[r][c], zero where point i is not
 *         seen by camera j or the camera is fixed (:1829-1842)
 *   Sout  (m cnp)^2    S = U - W V^-1 W^T (symmetric, so the reference's final transposition is immaterial)  (:1908-2019); with mcon > 0
 *         the fixed cameras' rows and columns are identity here (the reference compacts them away)
 * The Jacobian behind them is analytic (the reference's: forward differences): they agree with the reference's to ~1e-5 relative.
 * Singular V_i at zero damping (a point seen from one direction only) -> MI_BA_ERROR (the reference substitutes 100 I, :1776-1784). */
int mi_sba_motstr_levmar_blocks(int n, int m, int mcon, const char *vmask, double *p, int cnp, const double *x,
                                const double *R0, const double *f_fixed, int explicit_camera_centers,
                                int itmax, int verbose, const double opts[6], double info[10],
                                double *Vout, double *Sout, double *Uout, double *Wout);

#ifdef __cplusplus
}
#endif
#endif
