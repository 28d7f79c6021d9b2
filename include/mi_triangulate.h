/*
 * mi_triangulate.h -- C ABI of the MI355X multi-view triangulation step.
 *
 * Replaces the per-track loop the pipeline's stage program runs on the CPU between two bundle adjustments:
 *   SiftGPU/TriangulationMultiStep.cpp:1198-1304 (second level; first level :905-1010 = the same loop with
 *   min_views 3 and a 1.0 degree ray angle), with its helpers ComputeRayAngle (:498-530) and triangulate_n (:566-672,
 *   a copy of Bundler's imagelib/triangulate.c:181-275).
 * One GPU lane per track: measurements normalised by the focal length, at least min_views views, one pair of views
 * whose rays meet at more than min_angle_deg, linear triangulation (+ optional MINPACK-lmdif polish), reprojection
 * error, the point kept when the error is below max_error and zero otherwise -- the same arithmetic, in double.
 *
 * polish: the reference's lmdif polish does not run in the reference as built on LP64 (int / long mismatch between
 * matrix.c:489-507 and f2c.h:10; oracle/tri_oracle.c has the details), so 0 -- the linear solution -- reproduces the
 * reference's output and is the default; 1 runs MINPACK's lmdif1(tol 1e-5) as the source intends.
 *
 * Plain pointers and sizes; host arrays in, host arrays out; >= 0 success, -1 error (mi_tri_last_error).
 */
#ifndef MI_TRIANGULATE_H
#define MI_TRIANGULATE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_tri_engine mi_tri_engine;

typedef struct {
    int min_views;          /* 2 (TriangulationMultiStep.cpp:1236) or 3 (:952) */
    double min_angle_deg;   /* 0.5 (:1259) or 1.0 (:969) */
    double max_error;       /* 0.01 (:1274): error = sqrt(sum of squared normalised residuals) / views */
    int float_normalise;    /* 1: x / f passes through a float, as the reference's cv::Point2f does (:1223-1228) */
    int polish;             /* 0 linear solution (the reference's observable behaviour), 1 + MINPACK lmdif1(1e-5) */
    int device;
    int reserved[6];
} mi_tri_config;

/* status per track */
enum { MI_TRI_FEW_VIEWS = 0, MI_TRI_NARROW = 1, MI_TRI_LARGE_ERROR = 2, MI_TRI_KEPT = 3 };

void mi_tri_default_config(mi_tri_config *cfg);   /* second level: 2 views, 0.5 degrees, 0.01, float_normalise 1, polish 0 */
mi_tri_engine *mi_tri_create(const mi_tri_config *cfg);   /* NULL on error: mi_tri_last_error(NULL) */
void mi_tri_destroy(mi_tri_engine *e);
const char *mi_tri_last_error(const mi_tri_engine *e);

/* cameras: focal length (pixels), R row-major world->camera, t (x_cam = R X + t), c = camera centres as the reference
 * keeps them in Camera::cMat (NULL: computed as -R^T t) */
int mi_tri_set_cameras(mi_tri_engine *e, int m, const double *f, const double *R /* m*9 */, const double *t /* m*3 */,
                       const double *c /* m*3 or NULL */);
/* tracks in CSR form; obs_cam < 0 marks a view whose camera is not part of this model (skipped, :1221);
 * xy in pixels relative to the principal point */
int mi_tri_set_tracks(mi_tri_engine *e, int n, const int *pt_ptr /* n+1 */, const int *obs_cam, const double *xy /* 2 per view */);
/* runs on the device; inputs and outputs stay in HBM */
int mi_tri_run(mi_tri_engine *e);
/* xyz n*3 ((0,0,0) unless kept), err n (-1 when the track was not triangulated), status n; any may be NULL.
 * counts[4] = the reference's count1 (enough views), count2 (well conditioned), count3 and count (kept), :1303;
 * sum_sq_px = sum over kept tracks and their views of (error * focal)^2, nproj = those views (:1280-1286) */
int mi_tri_get(mi_tri_engine *e, double *xyz, double *err, int *status, int64_t counts[4], double *sum_sq_px, int64_t *nproj);
double mi_tri_last_ms(const mi_tri_engine *e);    /* device time of the last mi_tri_run (HIP events) */
double mi_tri_algorithmic_bytes(const mi_tri_engine *e);

/* all of the above in one call */
int mi_triangulate_tracks(const mi_tri_config *cfg, int m, const double *f, const double *R, const double *t, const double *c,
                          int n, const int *pt_ptr, const int *obs_cam, const double *xy,
                          double *xyz, double *err, int *status, int64_t counts[4], double *sum_sq_px, int64_t *nproj);

#ifdef __cplusplus
}
#endif
#endif
