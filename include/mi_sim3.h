/*
 * mi_sim3.h -- C ABI of the MI355X similarity alignment of two point sets.
 *
 * Replaces ComputeRigidTransform (SiftGPU/Merge3D3D.cpp:250-413, with ComputeRT :171-248; the same routine is pasted
 * into Merge2D2D_C.cpp:248, Merge3D2D.cpp:437, FinalNVM.cpp:244 and DataGeneration.cpp:281): the step that brings one
 * cluster's reconstruction into the frame of another before their tracks are fused and the merged model is bundle
 * adjusted.  Given n corresponding 3-D points:
 *   scale = the middle element of all n(n-1)/2 ratios |dst_i - dst_j| / |src_i - src_j|            (:255-274)
 *   5000 rounds of: sample (glibc rand(), seed round + 10000, size 10 doubling every 10 rounds up to n), R, t from the
 *   median-centred sample (R = U V^T of the SVD of the cross-covariance, no reflection fix), score = mean residual
 *   over ALL points; the round with the smallest score wins                                        (:329-408)
 * -- including the reference's quirks (the scale is applied a second time when scoring, :379; a duplicate sample is
 * redrawn once without a second check, :342-358).  All n(n-1)/2 ratios are materialised in HBM and the middle one is
 * found by an exact radix select; one workgroup runs one round.
 *
 * Parity is unpinned against the reference (its file needs Eigen and OpenCV, absent here): oracle/sim3_oracle.c is a
 * restatement checked against libc's rand() and numpy; the kernels are checked against it.
 *
 * Plain pointers and sizes; host arrays in and out; 0 success, -1 error (mi_sim3_last_error).
 */
#ifndef MI_SIM3_H
#define MI_SIM3_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int rounds;        /* 5000 (Merge3D3D.cpp:327) */
    int device;
    int reserved[6];
} mi_sim3_config;

#define MI_SIM3_MAX_POINTS 65536   /* the ratio list is n(n-1)/2 doubles: 17 GB at this size */

void mi_sim3_default_config(mi_sim3_config *cfg);
/* src, dst: n*3.  x_dst ~ R (scale x_src) + t.  scores: rounds values or NULL; best_round, device_ms may be NULL. */
int mi_sim3_align(const mi_sim3_config *cfg, int n, const double *src, const double *dst,
                  double *scale, double R[9], double t[3], double *scores, int *best_round, double *device_ms);
const char *mi_sim3_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
