/*
 * sim3_oracle.c -- CPU restatement of the cluster-merge alignment step.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ba_oracle.h): loaded by tests/, never by the product.
 *
 * PARITY UNPINNED against the reference itself: the routine lives in SiftGPU/Merge3D3D.cpp (and copies of it in
 * Merge2D2D_C.cpp:248, Merge3D2D.cpp:437, FinalNVM.cpp:244, DataGeneration.cpp:281), which needs Eigen and OpenCV;
 * neither is in this image, so the reference cannot be built here and no reference test holds vectors for it.
 * What IS pinned: the sampling sequence (glibc's rand(), restated below, checked against this container's libc),
 * and the linear algebra against numpy (median, polar factor of a 3x3 via SVD).
 *
 * What is restated (Merge3D3D.cpp, tree at /root/reference):
 *   ComputeRigidTransform :250-413
 *     - scale = the element at index size/2 of the sorted list of |dst_i - dst_j| / |src_i - src_j| over all pairs
 *       i < j with |src_i - src_j| != 0 (:255-274);
 *     - src *= scale (:277-280);
 *     - 5000 rounds (:329-408): srand(round + 10000); the sample size starts at 10 and doubles every 10 rounds up to
 *       N (:335-338); each sample index is rand() % N, redrawn ONCE (without a second check) when it is already in the
 *       sample (:342-358); R, t from ComputeRT on the sample; score = mean over ALL points of
 *       | R (scale * src_i) + t - dst_i | with src already scaled -- the scale is applied twice there (:373-383), kept;
 *       the first round, then every strictly smaller score, becomes the answer (:396-407).
 *   ComputeRT :171-248
 *     - centres = per-coordinate medians (element size/2 of each sorted coordinate list, :179-204), not means;
 *     - H = D^T S of the centred sample, R = U V^T from its SVD WITHOUT a reflection fix (:218-243; Eigen JacobiSVD
 *       there; U V^T is the orthogonal polar factor of H and does not depend on the SVD algorithm), t = c_dst - R c_src.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- glibc rand(): TYPE_3 additive feedback generator x_i = x_{i-3} + x_{i-31}, output x_i >> 1, seeded by the
 * Lehmer generator 16807 x mod 2^31 - 1 and 310 discarded draws (glibc stdlib/random_r.c) ---- */
typedef struct { unsigned int ring[31]; int f, b; } gl_rand_state;

void gl_srand(gl_rand_state *s, unsigned int seed)
{
    int32_t word;
    if (seed == 0) seed = 1;
    s->ring[0] = seed;
    word = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
        long hi = word / 127773, lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        s->ring[i] = (unsigned int)word;
    }
    s->f = 3; s->b = 0;
    for (int i = 0; i < 310; ++i) {
        s->ring[s->f] += s->ring[s->b];
        s->f = (s->f + 1) % 31; s->b = (s->b + 1) % 31;
    }
}

int gl_rand(gl_rand_state *s)
{
    s->ring[s->f] += s->ring[s->b];
    const unsigned int v = s->ring[s->f];
    s->f = (s->f + 1) % 31; s->b = (s->b + 1) % 31;
    return (int)(v >> 1);
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* Merge3D3D.cpp:255-274; returns the number of ratios (0: no usable pair) */
long long sim3_scale(int n, const double *src, const double *dst, double *scale)
{
    const long long cap = (long long)n * (n - 1) / 2;
    double *list = (double *)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    long long cnt = 0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const double s0 = src[3 * i] - src[3 * j], s1 = src[3 * i + 1] - src[3 * j + 1], s2 = src[3 * i + 2] - src[3 * j + 2];
            const double d0 = dst[3 * i] - dst[3 * j], d1 = dst[3 * i + 1] - dst[3 * j + 1], d2 = dst[3 * i + 2] - dst[3 * j + 2];
            const double ns = sqrt(s0 * s0 + s1 * s1 + s2 * s2);
            if (ns != 0.0) list[cnt++] = sqrt(d0 * d0 + d1 * d1 + d2 * d2) / ns;
        }
    if (cnt) { qsort(list, (size_t)cnt, sizeof(double), cmp_double); *scale = list[cnt / 2]; }
    free(list);
    return cnt;
}

/* orthogonal polar factor U V^T of a 3x3 H (row-major) through the eigen-decomposition of H^T H (cyclic Jacobi) */
void sim3_polar(const double H[9], double R[9])
{
    double A[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[3 * i + j] = H[i] * H[j] + H[3 + i] * H[3 + j] + H[6 + i] * H[6 + j]; /* H^T H */
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off <= 1e-300 || off <= 1e-18 * (fabs(A[0]) + fabs(A[4]) + fabs(A[8]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (apq == 0.0) continue;
                const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 3; ++k) { /* A <- A J */
                    const double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) { /* A <- J^T A */
                    const double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    /* R = H V diag(1/sigma) V^T */
    double HV[9], inv[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) HV[3 * i + j] = H[3 * i] * V[j] + H[3 * i + 1] * V[3 + j] + H[3 * i + 2] * V[6 + j];
    for (int j = 0; j < 3; ++j) { const double l = A[3 * j + j]; inv[j] = l > 0.0 ? 1.0 / sqrt(l) : 0.0; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = HV[3 * i] * inv[0] * V[3 * j] + HV[3 * i + 1] * inv[1] * V[3 * j + 1] + HV[3 * i + 2] * inv[2] * V[3 * j + 2];
}

/* ComputeRT, :171-248, on the sample idx[0..ns) of src (already scaled) / dst */
void sim3_rt(int ns, const int *idx, const double *src, const double *dst, double R[9], double t[3])
{
    double *buf = (double *)malloc(sizeof(double) * (size_t)ns), cs[3], cd[3], H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int c = 0; c < 3; ++c) {
        for (int k = 0; k < ns; ++k) buf[k] = src[3 * idx[k] + c];
        qsort(buf, (size_t)ns, sizeof(double), cmp_double); cs[c] = buf[ns / 2];
        for (int k = 0; k < ns; ++k) buf[k] = dst[3 * idx[k] + c];
        qsort(buf, (size_t)ns, sizeof(double), cmp_double); cd[c] = buf[ns / 2];
    }
    for (int k = 0; k < ns; ++k) { /* H = D^T S */
        const double *s = src + 3 * idx[k], *d = dst + 3 * idx[k];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) H[3 * i + j] += (d[i] - cd[i]) * (s[j] - cs[j]);
    }
    sim3_polar(H, R);
    for (int i = 0; i < 3; ++i) t[i] = cd[i] - (R[3 * i] * cs[0] + R[3 * i + 1] * cs[1] + R[3 * i + 2] * cs[2]);
    free(buf);
}

/* sample of one round, :335-360; returns the sample size */
int sim3_sample(int round, int n, int *idx)
{
    gl_rand_state rs;
    int npt = 10;
    for (int r = 1; r <= round; ++r) if (r % 10 == 0) npt = n > npt * 2 ? npt * 2 : n; /* the size is carried from round to round */
    gl_srand(&rs, (unsigned int)(round + 10000));
    unsigned char *seen = (unsigned char *)calloc((size_t)n, 1);
    for (int i = 0; i < npt; ++i) {
        int r = gl_rand(&rs) % n;
        if (seen[r]) r = gl_rand(&rs) % n;
        seen[r] = 1;
        idx[i] = r;
    }
    free(seen);
    return npt;
}

/* ComputeRigidTransform.  src is NOT modified (the reference scales it in place); rounds = 5000 in the reference.
 * scores (rounds) and best_round may be NULL.  Returns 0, or -1 when no pair of distinct source points exists. */
int sim3_align(int n, const double *src_in, const double *dst, int rounds, double *scale_out, double R[9], double t[3],
               double *scores, int *best_round)
{
    double scale;
    if (sim3_scale(n, src_in, dst, &scale) == 0) return -1;
    double *src = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n > 10 ? n : 10));
    for (int i = 0; i < 3 * n; ++i) src[i] = scale * src_in[i];
    double best = 0.0;
    int bi = -1;
    for (int round = 0; round < rounds; ++round) {
        double Rr[9], tr[3], total = 0.0;
        const int ns = sim3_sample(round, n, idx);
        sim3_rt(ns, idx, src, dst, Rr, tr);
        for (int i = 0; i < n; ++i) {
            const double p0 = scale * src[3 * i], p1 = scale * src[3 * i + 1], p2 = scale * src[3 * i + 2]; /* scaled twice, :379 */
            const double e0 = Rr[0] * p0 + Rr[1] * p1 + Rr[2] * p2 + tr[0] - dst[3 * i];
            const double e1 = Rr[3] * p0 + Rr[4] * p1 + Rr[5] * p2 + tr[1] - dst[3 * i + 1];
            const double e2 = Rr[6] * p0 + Rr[7] * p1 + Rr[8] * p2 + tr[2] - dst[3 * i + 2];
            total += sqrt(e0 * e0 + e1 * e1 + e2 * e2);
        }
        const double mean = total / n;
        if (scores) scores[round] = mean;
        if (round == 0 || best > mean) { best = mean; bi = round; memcpy(R, Rr, sizeof(Rr)); memcpy(t, tr, sizeof(tr)); }
    }
    *scale_out = scale;
    if (best_round) *best_round = bi;
    free(src); free(idx);
    return 0;
}
