/*
 * ref_harness.c -- driver glue around the REAL reference solver (test infrastructure).
 *
 * Compiled only where /root/reference exists (this container) by oracle/Makefile,
 * together with the reference's own sources where they lie:
 *   SiftGPU/ba_libs/lib/sba-1.5/{sba_levmar,sba_levmar_wrap,sba_lapack,sba_crsm,sba_chkjac}.c
 *   SiftGPU/ba_libs/lib/sfm-driver/sfm.c, lib/matrix/{matrix,vector}.c, lib/cminpack/lm*.c ...
 * into oracle/_ref/libref_sfm.so (git-ignored).  LAPACK/BLAS = the image's MKL
 * (/opt/conda/lib/libmkl_rt.so.1, LP64), as SURVEY.md section 8c verified.
 *
 * The reference's optimizer main (Optimizer/main_all.cpp) is missing from the tree, so
 * there are two ways in, both kept:
 *   1. run_sfm(...)            -- Bundler's own driver, sfm.c:598, untouched.
 *   2. ref_sba_motstr(...)     -- calls the reference's sba_motstr_levmar (sba.h:95) with a
 *      projection callback that forwards to the reference's exported rot_update (sfm.c:78)
 *      and sfm_project_rd (sfm.c:303) exactly the way its static sfm_project_point3
 *      (sfm.c:509-558) does; this exposes itmax/opts/info and the Uout/Vout/Wout/Sout hooks,
 *      which run_sfm hard-codes or hides.
 * Nothing here restates the solver: every number comes out of reference code.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sba.h"
#include "sfm.h"
#include "triangulate.h"

typedef struct {
    camera_params_t *cams;
    int est_focal, undistort, explicit_centers;
    int nk; /* 2: k1,k2 are parameters (sfm.c:648-649); 1: only k1 is, k2 is passed to sfm_project_rd as 0 */
} glue_t;

/* same call sequence as sfm_project_point3 (static in sfm.c:509-558), minus its
 * last-rotation cache (sfm.c:545-553), which only avoids recomputing rot_update */
static void glue_project(int j, int i, double *aj, double *bi, double *xij, void *adata)
{
    glue_t *g = (glue_t *)adata;
    double K[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, Rnew[9];
    double *dt = aj, *w = aj + 3, *k;
    (void)i;
    if (!g->est_focal) K[0] = K[4] = g->cams[j].f;
    else K[0] = K[4] = aj[6] / g->cams[j].f_scale; /* TEST_FOCAL branch, sfm.c:532 */
    double k1only[2];
    k = g->est_focal ? aj + 7 : aj + 6;
    if (g->undistort && g->nk == 1) { k1only[0] = k[0]; k1only[1] = 0.0; k = k1only; }
    rot_update(g->cams[j].R, w, Rnew);
    sfm_project_rd(g->cams + j, K, k, Rnew, dt, bi, xij, g->undistort, g->explicit_centers);
}

int ref_sba_motstr(int n, int m, int mcon, int cnp, const int *pt_ptr, const int *obs_cam,
                   const double *x, double *p, const double *R0, const double *f_fixed,
                   int est_focal, int undistort, int explicit_centers,
                   int itmax, int verbose, const double *opts6, double *info10,
                   double *Vout, double *Sout, double *Uout, double *Wout, int nk,
                   const double *cam_cw, const double *cam_cv, const double *pt_cw, const double *pt_cv)
{
    camera_constraints_t *cc = NULL;
    point_constraints_t *pc = NULL;
    int nobs = pt_ptr[n], i, j, o, ret;
    char *vmask = (char *)calloc((size_t)n * m, 1);
    camera_params_t *cams = (camera_params_t *)calloc(m, sizeof(camera_params_t));
    double *xx = (double *)malloc(sizeof(double) * 2 * nobs);
    glue_t g;
    if (!vmask || !cams || !xx) return -2;
    for (i = 0; i < n; ++i)
        for (o = pt_ptr[i]; o < pt_ptr[i + 1]; ++o) vmask[(size_t)i * m + obs_cam[o]] = 1;
    memcpy(xx, x, sizeof(double) * 2 * nobs); /* sba wants a non-const, compact (i,j)-ordered x */
    for (j = 0; j < m; ++j) {
        memcpy(cams[j].R, R0 + 9 * j, 9 * sizeof(double));
        cams[j].f = f_fixed ? f_fixed[j] : 1.0;
        cams[j].known_intrinsics = 0;
        cams[j].f_scale = 0.001; /* values run_sfm installs under TEST_FOCAL, sfm.c:640-663 */
        cams[j].k_scale = 5.0;
    }
    g.cams = cams; g.est_focal = est_focal; g.undistort = undistort; g.explicit_centers = explicit_centers; g.nk = nk;
    if (cam_cw) { /* camera_constraints_t as run_sfm builds them, sfm.c:727-760 */
        cc = (camera_constraints_t *)malloc(m * sizeof(camera_constraints_t));
        for (j = 0; j < m; ++j) {
            int r;
            cc[j].constrained = (char *)malloc(cnp);
            cc[j].constraints = (double *)malloc(sizeof(double) * cnp);
            cc[j].weights = (double *)malloc(sizeof(double) * cnp);
            for (r = 0; r < cnp; ++r) {
                cc[j].constrained[r] = cam_cw[(size_t)j * cnp + r] != 0.0;
                cc[j].constraints[r] = cam_cv[(size_t)j * cnp + r];
                cc[j].weights[r] = cam_cw[(size_t)j * cnp + r];
            }
        }
    }
    if (pt_cw) { /* point_constraints_t, sfm.c:763-787 */
        pc = (point_constraints_t *)malloc(n * sizeof(point_constraints_t));
        for (i = 0; i < n; ++i) {
            pc[i].constrained = pt_cw[i] != 0.0;
            pc[i].weight = pt_cw[i];
            pc[i].constraints[0] = pt_cv[3 * i]; pc[i].constraints[1] = pt_cv[3 * i + 1]; pc[i].constraints[2] = pt_cv[3 * i + 2];
        }
    }
    ret = sba_motstr_levmar(n, m, mcon, vmask, p, cnp, 3, xx, NULL, 2, glue_project, NULL, &g,
                            itmax, verbose, opts6, info10, cc != NULL, cc, pc != NULL, pc, Vout, Sout, Uout, Wout);
    if (cc) { for (j = 0; j < m; ++j) { free(cc[j].constrained); free(cc[j].constraints); free(cc[j].weights); } free(cc); }
    free(pc);
    free(vmask); free(cams); free(xx);
    return ret;
}

/* motion-only (points fixed) and structure-only (cameras fixed) solves of the reference:
 * sba_mot_levmar (sba.h:110-116; what run_sfm calls with fix_points, sfm.c:847) and sba_str_levmar
 * (sba.h:118-123), with the same projection glue. */
typedef struct { glue_t g; const double *pts; const double *cams; int cnp; } glue_ms_t;

static void glue_project_mot(int j, int i, double *aj, double *xij, void *adata)
{
    glue_ms_t *m = (glue_ms_t *)adata;
    double b[3] = { m->pts[3 * i], m->pts[3 * i + 1], m->pts[3 * i + 2] };
    glue_project(j, i, aj, b, xij, &m->g);
}

static void glue_project_str(int j, int i, double *bi, double *xij, void *adata)
{
    glue_ms_t *m = (glue_ms_t *)adata;
    double a[9];
    memcpy(a, m->cams + (size_t)j * m->cnp, sizeof(double) * m->cnp);
    glue_project(j, i, a, bi, xij, &m->g);
}

/* mode 1: p = m*cnp camera parameters (updated), pts n*3 fixed; mode 2: p = n*3 points (updated), cams m*cnp fixed */
int ref_sba_partial(int mode, int n, int m, int mcon, int cnp, const int *pt_ptr, const int *obs_cam,
                    const double *x, double *p, const double *fixed_part, const double *R0, const double *f_fixed,
                    int est_focal, int explicit_centers, int itmax, const double *opts6, double *info10)
{
    int nobs = pt_ptr[n], i, j, o, ret;
    char *vmask = (char *)calloc((size_t)n * m, 1);
    camera_params_t *cams = (camera_params_t *)calloc(m, sizeof(camera_params_t));
    double *xx = (double *)malloc(sizeof(double) * 2 * nobs);
    glue_ms_t ms;
    for (i = 0; i < n; ++i)
        for (o = pt_ptr[i]; o < pt_ptr[i + 1]; ++o) vmask[(size_t)i * m + obs_cam[o]] = 1;
    memcpy(xx, x, sizeof(double) * 2 * nobs);
    for (j = 0; j < m; ++j) {
        memcpy(cams[j].R, R0 + 9 * j, 9 * sizeof(double));
        cams[j].f = f_fixed ? f_fixed[j] : 1.0;
        cams[j].f_scale = 0.001; cams[j].k_scale = 5.0;
    }
    ms.g.cams = cams; ms.g.est_focal = est_focal; ms.g.undistort = 0; ms.g.explicit_centers = explicit_centers; ms.g.nk = 2;
    ms.cnp = cnp;
    if (mode == 1) {
        ms.pts = fixed_part; ms.cams = NULL;
        ret = sba_mot_levmar(n, m, mcon, vmask, p, cnp, xx, NULL, 2, glue_project_mot, NULL, &ms, itmax, 0, opts6, info10, 0, NULL);
    } else {
        ms.pts = NULL; ms.cams = fixed_part;
        ret = sba_str_levmar(n, m, vmask, p, 3, xx, NULL, 2, glue_project_str, NULL, &ms, itmax, 0, opts6, info10);
    }
    free(vmask); free(cams); free(xx);
    return ret;
}

/* Bundler's driver as is.  cams_R m*9, cams_t m*3 (t or C), cams_f m, cams_k m*2 are read
 * and written back; pts n*3 likewise. */
void ref_run_sfm_cons(int n, int m, int ncons, const int *pt_ptr, const int *obs_cam, const double *x,
                      int est_focal, int undistort, int explicit_centers,
                      double *cams_R, double *cams_t, double *cams_f, double *cams_k, double *pts, double eps2,
                      const double *cam_w, const double *cam_v, const double *pt_v, double pt_weight);

void ref_run_sfm(int n, int m, int ncons, const int *pt_ptr, const int *obs_cam, const double *x,
                 int est_focal, int undistort, int explicit_centers,
                 double *cams_R, double *cams_t, double *cams_f, double *cams_k, double *pts, double eps2)
{
    ref_run_sfm_cons(n, m, ncons, pt_ptr, obs_cam, x, est_focal, undistort, explicit_centers, cams_R, cams_t, cams_f, cams_k,
                     pts, eps2, NULL, NULL, NULL, 0.0);
}

/* the same with run_sfm's own constraint arguments: cam_w / cam_v are m*9 weights / targets in the units of
 * camera_params_t (weight 0 = unconstrained; sfm.h:37-39), pt_v n*3 targets ((0,0,0) = unconstrained, sfm.c:769) */
void ref_run_sfm_cons(int n, int m, int ncons, const int *pt_ptr, const int *obs_cam, const double *x,
                      int est_focal, int undistort, int explicit_centers,
                      double *cams_R, double *cams_t, double *cams_f, double *cams_k, double *pts, double eps2,
                      const double *cam_w, const double *cam_v, const double *pt_v, double pt_weight)
{
    int nobs = pt_ptr[n], i, j, o;
    char *vmask = (char *)calloc((size_t)n * m, 1);
    camera_params_t *cams = (camera_params_t *)calloc(m, sizeof(camera_params_t));
    double *xx = (double *)malloc(sizeof(double) * 2 * nobs);
    v3_t *P = (v3_t *)malloc(sizeof(v3_t) * n), *PC = NULL;
    for (i = 0; i < n; ++i) {
        for (o = pt_ptr[i]; o < pt_ptr[i + 1]; ++o) vmask[(size_t)i * m + obs_cam[o]] = 1;
        Vx(P[i]) = pts[3 * i]; Vy(P[i]) = pts[3 * i + 1]; Vz(P[i]) = pts[3 * i + 2];
    }
    memcpy(xx, x, sizeof(double) * 2 * nobs);
    for (j = 0; j < m; ++j) {
        memcpy(cams[j].R, cams_R + 9 * j, 9 * sizeof(double));
        memcpy(cams[j].t, cams_t + 3 * j, 3 * sizeof(double));
        cams[j].f = cams_f[j]; cams[j].k[0] = cams_k[2 * j]; cams[j].k[1] = cams_k[2 * j + 1];
        if (cam_w) {
            int r;
            for (r = 0; r < 9; ++r) {
                cams[j].constrained[r] = cam_w[9 * j + r] != 0.0;
                cams[j].weights[r] = cam_w[9 * j + r];
                cams[j].constraints[r] = cam_v[9 * j + r];
            }
        }
    }
    if (pt_v) {
        PC = (v3_t *)malloc(sizeof(v3_t) * n);
        for (i = 0; i < n; ++i) { Vx(PC[i]) = pt_v[3 * i]; Vy(PC[i]) = pt_v[3 * i + 1]; Vz(PC[i]) = pt_v[3 * i + 2]; }
    }
    run_sfm(n, m, ncons, vmask, xx, est_focal, 0, undistort, explicit_centers, cams, P,
            cam_w != NULL, pt_v != NULL, PC, pt_weight, 0, 0, eps2, NULL, NULL, NULL, NULL);
    free(PC);
    for (j = 0; j < m; ++j) {
        memcpy(cams_R + 9 * j, cams[j].R, 9 * sizeof(double));
        memcpy(cams_t + 3 * j, cams[j].t, 3 * sizeof(double));
        cams_f[j] = cams[j].f; cams_k[2 * j] = cams[j].k[0]; cams_k[2 * j + 1] = cams[j].k[1];
    }
    for (i = 0; i < n; ++i) { pts[3 * i] = Vx(P[i]); pts[3 * i + 1] = Vy(P[i]); pts[3 * i + 2] = Vz(P[i]); }
    free(vmask); free(cams); free(xx); free(P);
}

/* Bundler's triangulate_n (imagelib/triangulate.c:181-275) as is: the routine the pipeline's triangulation step copies
 * (TriangulationMultiStep.cpp:566-672 differs only in how the returned error is normalised).  pv 2*nv, Rs 9*nv, ts 3*nv. */
void ref_triangulate_n(int nv, const double *pv, const double *Rs, const double *ts, double *X, double *err)
{
    v2_t *p = (v2_t *)malloc(sizeof(v2_t) * nv);
    double *R = (double *)malloc(sizeof(double) * 9 * nv), *t = (double *)malloc(sizeof(double) * 3 * nv);
    int i;
    v3_t r;
    for (i = 0; i < nv; ++i) { Vx(p[i]) = pv[2 * i]; Vy(p[i]) = pv[2 * i + 1]; }
    memcpy(R, Rs, sizeof(double) * 9 * nv); memcpy(t, ts, sizeof(double) * 3 * nv);
    r = triangulate_n(nv, p, R, t, err);
    X[0] = Vx(r); X[1] = Vy(r); X[2] = Vz(r);
    free(p); free(R); free(t);
}
