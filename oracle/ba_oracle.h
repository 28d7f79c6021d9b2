/*
 * ba_oracle.h -- CPU restatement of the reference bundle-adjustment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and there only as the checker /
 * the timed CPU baseline.  The product (large-scale-sfm_amd/csrc) never links
 * or calls it and fails loudly when its HIP library is missing.
 *
 * What is restated (reference file:line, tree at /root/reference):
 *   - sparse LM with Schur complement, Bundler-patched sba-1.5:
 *       Optimizer/sba-1.5/sba_levmar.c:457-2081  (sba_motstr_levmar_x)
 *   - finite-difference Jacobian the Bundler driver uses (projac == NULL):
 *       Optimizer/sba-1.5/sba_levmar_wrap.c:163-259
 *   - camera model, parameter packing, rotation update of Bundler's run_sfm:
 *       SiftGPU/ba_libs/lib/sfm-driver/sfm.c:78-117 (rot_update),
 *       :303-374 (sfm_project_rd), :509-558 (sfm_project_point3),
 *       :640-720 (packing, f_scale/k_scale under TEST_FOCAL, opts[0..5])
 *   - dense reduced solve: sba_Axb_Chol (sba_lapack.c:374-485) = LAPACK
 *     dpotrf/dpotrs; restated here as an in-file Cholesky (LAPACK is a
 *     third-party dependency of the reference, unpinned by it: make_g2o:4).
 *   - 3x3 inverse: sba_symat_invert_BK (sba_lapack.c:1053-1140) = LAPACK
 *     dsytrf/dsytri; restated as a closed-form symmetric inverse.
 *
 * Parity pin: tests/golden/ holds vectors produced by the REAL reference
 * (oracle/_ref, built from the reference's own sources by oracle/Makefile)
 * with tests/golden/make_golden.py; tests/test_oracle_golden.py checks this
 * restatement against them.
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_INFOSZ 10
#define ORA_TRACE_W 6   /* per trial: itno, mu, cost_before, cost_trial, accepted, lin_iters */

typedef struct {
    int n;                 /* points */
    int m;                 /* cameras */
    int mcon;              /* first mcon cameras are held fixed (sba_levmar.c:460) */
    int cnp;               /* 6 (+1 if est_focal) (+nk if undistort)  (sfm.c:643-649) */
    int est_focal;         /* sfm.c:643 */
    int undistort;         /* sfm.c:648 */
    int explicit_centers;  /* camera params hold C (1) or t (0): sfm.c:319-330 */
    double proj_sign;      /* -1 = Bundler (sfm.c:333), +1 = VisualSFM/PBA (+z forward) */
    double f_scale;        /* 0.001 under TEST_FOCAL (sfm.c:640) */
    double k_scale;        /* 5.0   under TEST_FOCAL (sfm.c:641) */
    /* structure, CSR by point; cameras ascending inside a point */
    const int *pt_ptr;     /* n+1 */
    const int *obs_cam;    /* nobs */
    const double *x;       /* 2*nobs measurements, same order */
    const double *R0;      /* m*9 row-major initial rotations (fixed during the solve, sfm.c:549) */
    const double *f_fixed; /* m focal lengths used when !est_focal (sfm.c:523) */
    /* Bundler's soft constraints (sba.h:76-86; sba_levmar.c:809-842, 952-963, 1017-1029, 1488-1522); weight 0 = none */
    const double *cam_cw;  /* m*cnp weights or NULL   */
    const double *cam_cv;  /* m*cnp target values     */
    const double *pt_cw;   /* n weights or NULL (the reference multiplies them by the observation count) */
    const double *pt_cv;   /* n*3 target positions    */
    int nk;                /* distortion parameters when undistort: 2 = Bundler's k1,k2 (sfm.c:648-649), 1 = k1 only
                            * with k2 held at 0 (the single-coefficient model of PBA's CameraT, DataInterface.h:53) */
} ora_problem;

typedef struct {
    int itmax;
    int fd_jacobian;       /* 1 = forward differences exactly as sba_levmar_wrap.c:163-259, 0 = analytic */
    int solver;            /* 0 = dense Cholesky (reference), 1 = block-Jacobi PCG (mirror of the GPU engine) */
    int pcg_maxit;
    double pcg_tol;
    double opts[6];        /* tau, eps1, eps2, eps3, eps4, eps5 (sfm.c:711-719; sba_levmar.c:609-610) */
    int disable_stops;     /* bench: ignore stop rules 1,2,4,5,8 so exactly itmax linear systems are solved */
    int mode;              /* 0 cameras + points (sba_motstr_levmar_x), 1 cameras only (sba_mot_levmar_x,
                            * sba_levmar.c:2090-2690), 2 points only (sba_str_levmar_x, :2698-3230) */
} ora_options;

/* p: m*cnp + 3n parameters, updated in place.  Returns iterations or -1. */
int ora_motstr_levmar(const ora_problem *pb, const ora_options *op, double *p,
                      double info[ORA_INFOSZ], double *trace, int trace_cap, int *ntrace);

/* One linearisation at p, no damping: the blocks the reference exports through
 * Uout/Vout/Wout/Sout (sba_levmar.c:1633-2026).  Any output may be NULL.
 *   U m*cnp*cnp, V n*9 (full symmetric), W nobs*cnp*3 (per observation, CSR order),
 *   S (m*cnp)^2 row-major, ea m*cnp, eb n*3, e 2*nobs, returns sum e^2 */
double ora_linearize(const ora_problem *pb, int fd_jacobian, const double *p,
                     double *U, double *V, double *W, double *S,
                     double *ea, double *eb, double *e, double *jac);

/* predicted measurements at p (2*nobs) and cost vs pb->x */
double ora_residuals(const ora_problem *pb, const double *p, double *e);

void ora_rot_update(const double *R, const double *w, double *Rnew);
void ora_project(const ora_problem *pb, int j, const double *aj, const double *bi, double *xij);

#ifdef __cplusplus
}
#endif
#endif
