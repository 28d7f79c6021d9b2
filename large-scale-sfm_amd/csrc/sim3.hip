// sim3.hip -- similarity alignment of two 3-D point sets on gfx950 (include/mi_sim3.h):
// ComputeRigidTransform / ComputeRT of SiftGPU/Merge3D3D.cpp:171-413.
//
//   k_ratios      all n(n-1)/2 distance ratios, materialised (HBM is 288 GB: 17 GB at the 65 536-point cap)
//   k_hist/k_pick exact radix select of the middle ratio: 8 passes of an 8-bit digit histogram over the list
//   k_round       one workgroup = one RANSAC round: lane 0 replays glibc's rand() for the sample (a duplicate bitmap
//                 in LDS), six simultaneous radix selects give the per-coordinate medians, block reductions the
//                 cross-covariance and the score over all points, lane 0 the 3x3 polar factor
// Bounds: k_ratios / k_hist stream the ratio list (HBM); k_round is latency-bound on the serial sampling.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi_sim3.h"

namespace {

thread_local std::string g_err;
constexpr int kRB = 256; // threads of a round's workgroup

__device__ __forceinline__ unsigned long long key_of(double v) { return (unsigned long long)__double_as_longlong(v); } // v >= 0

// ratio of pair (i, j > i) at offset i n - i (i + 1) / 2 + (j - i - 1); pairs of coincident source points get +inf
// (they are not part of the reference's list, Merge3D3D.cpp:266) and are counted
__global__ __launch_bounds__(256) void k_ratios(int n, const double *__restrict__ src, const double *__restrict__ dst,
                                               double *__restrict__ ratio, unsigned long long *__restrict__ nskip)
{
#pragma clang fp contract(off) // the ratios are ORDERED: no fused multiply-adds, so the list is the reference's list bit for bit
    const int i = blockIdx.y, j = i + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double s0 = src[3 * i] - src[3 * j], s1 = src[3 * i + 1] - src[3 * j + 1], s2 = src[3 * i + 2] - src[3 * j + 2];
    const double d0 = dst[3 * i] - dst[3 * j], d1 = dst[3 * i + 1] - dst[3 * j + 1], d2 = dst[3 * i + 2] - dst[3 * j + 2];
    const double ns = sqrt(s0 * s0 + s1 * s1 + s2 * s2);
    const size_t off = (size_t)i * n - (size_t)i * (i + 1) / 2 + (size_t)(j - i - 1);
    if (ns != 0.0) ratio[off] = sqrt(d0 * d0 + d1 * d1 + d2 * d2) / ns;
    else { ratio[off] = INFINITY; atomicAdd(nskip, 1ull); }
}

// histogram of the 8-bit digit at `shift` over the keys whose higher bits equal `prefix`
__global__ __launch_bounds__(256) void k_hist(size_t count, const double *__restrict__ v, int shift,
                                             const unsigned long long *__restrict__ state, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = state[0];
    const unsigned long long mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long key = key_of(v[k]);
        if ((key & mask) == (prefix & mask)) atomicAdd(&h[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// state[0] = prefix so far, state[1] = rank still to skip inside the prefix; picks the digit, clears the histogram
__global__ void k_pick(int shift, unsigned long long *__restrict__ state, unsigned long long *__restrict__ hist)
{
    if (threadIdx.x == 0) {
        unsigned long long k = state[1], acc = 0;
        int d = 0;
        for (; d < 256; ++d) { if (acc + hist[d] > k) break; acc += hist[d]; }
        if (d == 256) d = 255;
        state[0] |= (unsigned long long)d << shift;
        state[1] = k - acc;
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
}

// ---- glibc rand() (TYPE_3 additive feedback, stdlib/random_r.c), state in LDS, used by lane 0 only ----
__device__ void gl_srand(unsigned int *ring, int &f, int &b, unsigned int seed)
{
    if (seed == 0) seed = 1;
    ring[0] = seed;
    int word = (int)seed;
    for (int i = 1; i < 31; ++i) {
        const long long hi = word / 127773, lo = word % 127773;
        word = (int)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        ring[i] = (unsigned int)word;
    }
    f = 3; b = 0;
    for (int i = 0; i < 310; ++i) { ring[f] += ring[b]; f = f == 30 ? 0 : f + 1; b = b == 30 ? 0 : b + 1; }
}
__device__ __forceinline__ int gl_rand(unsigned int *ring, int &f, int &b)
{
    ring[f] += ring[b];
    const unsigned int v = ring[f];
    f = f == 30 ? 0 : f + 1; b = b == 30 ? 0 : b + 1;
    return (int)(v >> 1);
}

// orthogonal polar factor U V^T of a 3x3 (cyclic Jacobi on H^T H), as oracle/sim3_oracle.c::sim3_polar
__device__ void polar3(const double H[9], double R[9])
{
    double A[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[3 * i + j] = H[i] * H[j] + H[3 + i] * H[3 + j] + H[6 + i] * H[6 + j];
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
                for (int k = 0; k < 3; ++k) { const double x = A[3 * k + p], y = A[3 * k + q]; A[3 * k + p] = c * x - s * y; A[3 * k + q] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = A[3 * p + k], y = A[3 * q + k]; A[3 * p + k] = c * x - s * y; A[3 * q + k] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = V[3 * k + p], y = V[3 * k + q]; V[3 * k + p] = c * x - s * y; V[3 * k + q] = s * x + c * y; }
            }
    }
    double HV[9], inv[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) HV[3 * i + j] = H[3 * i] * V[j] + H[3 * i + 1] * V[3 + j] + H[3 * i + 2] * V[6 + j];
    for (int j = 0; j < 3; ++j) { const double l = A[3 * j + j]; inv[j] = l > 0.0 ? 1.0 / sqrt(l) : 0.0; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = HV[3 * i] * inv[0] * V[3 * j] + HV[3 * i + 1] * inv[1] * V[3 * j + 1] + HV[3 * i + 2] * inv[2] * V[3 * j + 2];
}

// order-preserving map of any double onto unsigned integers (coordinates can be negative)
__device__ __forceinline__ unsigned long long okey(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double okey_inv(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ double block_sum(double v, double *sh) // fixed-order tree: the same bits whatever the schedule
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = kRB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// one RANSAC round per workgroup (Merge3D3D.cpp:329-408).  src is the SCALED source set.
__global__ __launch_bounds__(kRB) void k_round(int n, int round0, double scale, const double *__restrict__ src, const double *__restrict__ dst,
                                              int *__restrict__ idx_all, double *__restrict__ score, double *__restrict__ rt)
{
    extern __shared__ unsigned int sdyn[];            // duplicate bitmap: ceil(n / 32) words
    __shared__ unsigned int ring[31];
    __shared__ unsigned int hist[6][256];
    __shared__ unsigned long long pref[6], rem[6];
    __shared__ double sh[kRB], sRt[12], sCen[6];
    __shared__ int sNs;
    const int round = round0 + blockIdx.x, tid = threadIdx.x;
    int *idx = idx_all + (size_t)blockIdx.x * n;
    const int words = (n + 31) / 32;
    for (int k = tid; k < words; k += kRB) sdyn[k] = 0;
    __syncthreads();
    if (tid == 0) { // the sample, :335-360
        int npt = 10;
        for (int r = 1; r <= round; ++r) if (r % 10 == 0) npt = n > npt * 2 ? npt * 2 : n;
        int f, b;
        gl_srand(ring, f, b, (unsigned int)(round + 10000));
        for (int i = 0; i < npt; ++i) {
            int r = gl_rand(ring, f, b) % n;
            if (sdyn[r >> 5] & (1u << (r & 31))) r = gl_rand(ring, f, b) % n;   // redrawn once, not checked again
            sdyn[r >> 5] |= 1u << (r & 31);
            idx[i] = r;
        }
        sNs = npt;
    }
    __syncthreads();
    const int ns = sNs;
    // per-coordinate medians of the sample = element ns / 2 of each sorted coordinate list (:179-204): six radix selects
    if (tid < 6) { pref[tid] = 0; rem[tid] = (unsigned long long)(ns / 2); }
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int k = tid; k < 6 * 256; k += kRB) (&hist[0][0])[k] = 0;
        __syncthreads();
        const unsigned long long mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
        for (int k = tid; k < ns; k += kRB) {
            const int p = idx[k];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const unsigned long long key = okey(c < 3 ? src[3 * p + c] : dst[3 * p + c - 3]);
                if ((key & mask) == (pref[c] & mask)) atomicAdd(&hist[c][(key >> shift) & 255], 1u);
            }
        }
        __syncthreads();
        if (tid < 6) {
            unsigned long long k = rem[tid], acc = 0;
            int d = 0;
            for (; d < 256; ++d) { if (acc + hist[tid][d] > k) break; acc += hist[tid][d]; }
            if (d == 256) d = 255;
            pref[tid] |= (unsigned long long)d << shift;
            rem[tid] = k - acc;
        }
        __syncthreads();
    }
    if (tid < 6) sCen[tid] = okey_inv(pref[tid]);
    __syncthreads();
    // H = D^T S of the centred sample (:207-216)
    double h[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int k = tid; k < ns; k += kRB) {
        const int p = idx[k];
        const double s0 = src[3 * p] - sCen[0], s1 = src[3 * p + 1] - sCen[1], s2 = src[3 * p + 2] - sCen[2];
        const double d0 = dst[3 * p] - sCen[3], d1 = dst[3 * p + 1] - sCen[4], d2 = dst[3 * p + 2] - sCen[5];
        h[0] += d0 * s0; h[1] += d0 * s1; h[2] += d0 * s2; h[3] += d1 * s0; h[4] += d1 * s1; h[5] += d1 * s2;
        h[6] += d2 * s0; h[7] += d2 * s1; h[8] += d2 * s2;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) h[c] = block_sum(h[c], sh);
    if (tid == 0) {
        double R[9];
        polar3(h, R);
        for (int c = 0; c < 9; ++c) sRt[c] = R[c];
        for (int i = 0; i < 3; ++i) sRt[9 + i] = sCen[3 + i] - (R[3 * i] * sCen[0] + R[3 * i + 1] * sCen[1] + R[3 * i + 2] * sCen[2]);
    }
    __syncthreads();
    // score: mean residual over ALL points, the scale applied once more as the reference does (:373-383)
    double tot = 0.0;
    for (int p = tid; p < n; p += kRB) {
        const double p0 = scale * src[3 * p], p1 = scale * src[3 * p + 1], p2 = scale * src[3 * p + 2];
        const double e0 = sRt[0] * p0 + sRt[1] * p1 + sRt[2] * p2 + sRt[9] - dst[3 * p];
        const double e1 = sRt[3] * p0 + sRt[4] * p1 + sRt[5] * p2 + sRt[10] - dst[3 * p + 1];
        const double e2 = sRt[6] * p0 + sRt[7] * p1 + sRt[8] * p2 + sRt[11] - dst[3 * p + 2];
        tot += sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    }
    tot = block_sum(tot, sh);
    if (tid == 0) score[round] = tot / n;
    if (tid < 12) rt[(size_t)round * 12 + tid] = sRt[tid];
}

__global__ void k_scale_src(int n3, double scale, double *__restrict__ src)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) src[i] = scale * src[i];
}

template <class T>
struct Buf {
    T *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)); }
};

} // namespace

extern "C" {

void mi_sim3_default_config(mi_sim3_config *c) { std::memset(c, 0, sizeof(*c)); c->rounds = 5000; c->device = 0; }
const char *mi_sim3_last_error(void) { return g_err.c_str(); }

int mi_sim3_align(const mi_sim3_config *cfg, int n, const double *src, const double *dst,
                  double *scale, double R[9], double t[3], double *scores, int *best_round, double *device_ms)
{
    mi_sim3_config c;
    if (cfg) c = *cfg; else mi_sim3_default_config(&c);
    if (!src || !dst || !scale || !R || !t || n < 2 || c.rounds < 1) { g_err = "mi_sim3_align: bad argument"; return -1; }
    if (n > MI_SIM3_MAX_POINTS) { g_err = "mi_sim3_align: more than MI_SIM3_MAX_POINTS points"; return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device: the alignment runs on an MI355X only (no CPU fallback)"; return -1; }
    if (c.device < 0 || c.device >= ndev || hipSetDevice(c.device) != hipSuccess) { g_err = "bad device index"; return -1; }
#define CK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_err = std::string(what) + ": " + hipGetErrorString(e_); return -1; } } while (0)
    const size_t npair = (size_t)n * (n - 1) / 2;
    Buf<double> d_src, d_dst, d_ratio, d_score, d_rt;
    Buf<unsigned long long> d_state, d_hist, d_skip;
    Buf<int> d_idx;
    const int chunk = std::max(1, std::min(c.rounds, (int)std::min<size_t>(4096, (size_t)(1ull << 28) / (size_t)n)));   // <= 1 GB of sample indices
    CK(d_src.alloc((size_t)n * 3), "alloc"); CK(d_dst.alloc((size_t)n * 3), "alloc"); CK(d_ratio.alloc(npair), "alloc ratio list");
    CK(d_score.alloc(c.rounds), "alloc"); CK(d_rt.alloc((size_t)c.rounds * 12), "alloc"); CK(d_state.alloc(2), "alloc");
    CK(d_hist.alloc(256), "alloc"); CK(d_skip.alloc(1), "alloc"); CK(d_idx.alloc((size_t)chunk * n), "alloc sample indices");
    CK(hipMemcpy(d_src.p, src, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice), "upload");
    CK(hipMemcpy(d_dst.p, dst, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice), "upload");
    CK(hipMemset(d_skip.p, 0, sizeof(unsigned long long)), "memset"); CK(hipMemset(d_hist.p, 0, 256 * sizeof(unsigned long long)), "memset");
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0), "event"); CK(hipEventCreate(&ev1), "event");
    (void)hipEventRecord(ev0, 0);
    // ---- scale: the middle element of the sorted ratio list
    hipLaunchKernelGGL(k_ratios, dim3((unsigned)((n + 255) / 256), (unsigned)(n - 1)), dim3(256), 0, 0, n, d_src.p, d_dst.p, d_ratio.p, d_skip.p);
    unsigned long long nskip = 0;
    CK(hipMemcpy(&nskip, d_skip.p, sizeof(nskip), hipMemcpyDeviceToHost), "download");
    const unsigned long long valid = (unsigned long long)npair - nskip;
    if (valid == 0) { g_err = "mi_sim3_align: no pair of distinct source points"; (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); return -1; }
    unsigned long long st[2] = { 0ull, valid / 2 };
    CK(hipMemcpy(d_state.p, st, sizeof(st), hipMemcpyHostToDevice), "upload");
    const unsigned hgrid = (unsigned)std::min<size_t>(8192, (npair + 255) / 256);
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_hist, dim3(hgrid), dim3(256), 0, 0, npair, d_ratio.p, shift, d_state.p, d_hist.p);
        hipLaunchKernelGGL(k_pick, dim3(1), dim3(256), 0, 0, shift, d_state.p, d_hist.p);
    }
    CK(hipMemcpy(st, d_state.p, sizeof(st), hipMemcpyDeviceToHost), "download");
    double sc;
    std::memcpy(&sc, &st[0], sizeof(double));
    // ---- the rounds, on the scaled source set
    hipLaunchKernelGGL(k_scale_src, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, 0, 3 * n, sc, d_src.p);
    const size_t lds = (size_t)((n + 31) / 32) * sizeof(unsigned int);
    for (int r0 = 0; r0 < c.rounds; r0 += chunk) {
        const int nr = std::min(chunk, c.rounds - r0);
        hipLaunchKernelGGL(k_round, dim3((unsigned)nr), dim3(kRB), lds, 0, n, r0, sc, d_src.p, d_dst.p, d_idx.p, d_score.p, d_rt.p);
    }
    (void)hipEventRecord(ev1, 0);
    CK(hipGetLastError(), "launch"); CK(hipDeviceSynchronize(), "sync");
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    std::vector<double> h_score(c.rounds), h_rt((size_t)c.rounds * 12);
    CK(hipMemcpy(h_score.data(), d_score.p, sizeof(double) * c.rounds, hipMemcpyDeviceToHost), "download");
    CK(hipMemcpy(h_rt.data(), d_rt.p, sizeof(double) * 12 * c.rounds, hipMemcpyDeviceToHost), "download");
#undef CK
    int bi = 0; // the first round, then every strictly smaller score (:396-407)
    for (int r = 1; r < c.rounds; ++r) if (h_score[bi] > h_score[r]) bi = r;
    *scale = sc;
    std::memcpy(R, h_rt.data() + (size_t)bi * 12, 9 * sizeof(double));
    std::memcpy(t, h_rt.data() + (size_t)bi * 12 + 9, 3 * sizeof(double));
    if (scores) std::memcpy(scores, h_score.data(), sizeof(double) * c.rounds);
    if (best_round) *best_round = bi;
    if (device_ms) *device_ms = ms;
    return 0;
}

} // extern "C"
