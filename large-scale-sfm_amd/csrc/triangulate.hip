// triangulate.hip -- multi-view triangulation of tracks on gfx950 (include/mi_triangulate.h).
//
// The reference runs this loop on one CPU core between two bundle adjustments
// (SiftGPU/TriangulationMultiStep.cpp:1198-1304, helpers :498-530 and :566-672).  Here: one lane per track, three
// passes over the track's views (ray-angle test, normal equations of the linear system, errors) plus, on request,
// MINPACK's lmdif polish on the 3x3 normal equations.  The camera table (128 B per camera: R, t, c, f) is
// L2-resident; the measurement stream is read once from HBM and re-read from L2.  Bound: HBM (20 B per view in,
// 52 B per track out) -- the arithmetic is a few hundred flops per view.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi_triangulate.h"

namespace {

constexpr int kCamRec = 16; // doubles per camera: R (9), t (3), c (3), f

struct TriArgs {
    const double *cam;      // m * kCamRec
    const int *pt_ptr;      // n + 1
    const int *obs_cam;
    const double2 *xy;
    double *xyz, *err, *sumsq;
    int *status, *nviews;
    int n, min_views, float_normalise, polish, stage_cap;
    double min_angle_deg, max_error;
};

__device__ __forceinline__ bool solve3(const double H[6], const double g[3], double x[3])
{
    const double a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (!(fabs(det) > 0.0)) return false;
    const double id = 1.0 / det;
    const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
    x[0] = (c00 * g[0] + c01 * g[1] + c02 * g[2]) * id;
    x[1] = (c01 * g[0] + c11 * g[1] + c12 * g[2]) * id;
    x[2] = (c02 * g[0] + c12 * g[1] + c22 * g[2]) * id;
    return true;
}

constexpr int kTriBlock = 64;     // tracks per workgroup = one wavefront
constexpr int kTriStage = 1024;   // views a workgroup may stage through LDS (20 KB); longer ranges read HBM / L2 directly

// normalised measurement of a view (TriangulationMultiStep.cpp:1223-1228: a cv::Point2f, hence the float)
__device__ __forceinline__ void normalise(const TriArgs &a, double2 m, double f, double &px, double &py)
{
    px = m.x / f; py = m.y / f;
    if (a.float_normalise) { px = (double)(float)px; py = (double)(float)py; }
}

// the views of a workgroup's tracks: a contiguous range of the measurement stream, staged once (coalesced, already
// normalised) in LDS when it fits, else read in place
struct Views {
    const TriArgs &a;
    const double2 *spv; const int *scam;   // LDS images, index o - base
    int base; bool staged;
    __device__ __forceinline__ int cam(int o) const { return staged ? scam[o - base] : a.obs_cam[o]; }
    __device__ __forceinline__ void pv(int o, double f, double &px, double &py) const
    {
        if (staged) { const double2 v = spv[o - base]; px = v.x; py = v.y; }
        else normalise(a, a.xy[o], f, px, py);
    }
};

// the 128-B camera record as eight 16-B loads: the gather is bound by the address pipeline (64 lanes, 64 different
// cameras), so half as many load instructions is half the time
__device__ __forceinline__ void load_cam(const double *__restrict__ table, int j, double cr[kCamRec])
{
    const double2 *p = reinterpret_cast<const double2 *>(table + (size_t)j * kCamRec);
#pragma unroll
    for (int k = 0; k < kCamRec / 2; ++k) { const double2 v = p[k]; cr[2 * k] = v.x; cr[2 * k + 1] = v.y; }
}

// residual of triangulate_n_residual (:548-563) with project() (:533-547)
__device__ __forceinline__ void resid(const double *cr, double px, double py, const double X[3], double &f0, double &f1, double &depth)
{
    const double q0 = (cr[0] * X[0] + cr[1] * X[1] + cr[2] * X[2]) + cr[9];
    const double q1 = (cr[3] * X[0] + cr[4] * X[1] + cr[5] * X[2]) + cr[10];
    const double q2 = (cr[6] * X[0] + cr[7] * X[1] + cr[8] * X[2]) + cr[11];
    f0 = px - q0 / q2; f1 = py - q1 / q2; depth = q2;
}

__device__ double fnorm2(const Views &V, int o0, int o1, const double X[3])
{
    double s = 0.0;
    for (int o = o0; o < o1; ++o) {
        const int j = V.cam(o);
        if (j < 0) continue;
        double cr[kCamRec], px, py, f0, f1, dz;
        load_cam(V.a.cam, j, cr);
        V.pv(o, cr[15], px, py);
        resid(cr, px, py, X, f0, f1, dz);
        s += f0 * f0 + f1 * f1;
    }
    return s;
}

// fdjac2 (forward differences, h = sqrt(epsmch) |x_j|) folded into N = J^T J and g = J^T f
__device__ void fd_normal(const Views &V, int o0, int o1, const double X[3], double N[6], double g[3])
{
    const double eps = 1.4901161193847656e-08; // sqrt(2^-52)
    double h[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { h[j] = eps * fabs(X[j]); if (h[j] == 0.0) h[j] = eps; }
#pragma unroll
    for (int k = 0; k < 6; ++k) N[k] = 0.0;
    g[0] = g[1] = g[2] = 0.0;
    for (int o = o0; o < o1; ++o) {
        const int jc = V.cam(o);
        if (jc < 0) continue;
        double cr[kCamRec], px, py, f0, f1, dz, J[2][3];
        load_cam(V.a.cam, jc, cr);
        V.pv(o, cr[15], px, py);
        resid(cr, px, py, X, f0, f1, dz);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double Xp[3] = { X[0], X[1], X[2] }, e0, e1;
            Xp[j] = X[j] + h[j];
            resid(cr, px, py, Xp, e0, e1, dz);
            J[0][j] = (e0 - f0) / h[j]; J[1][j] = (e1 - f1) / h[j];
        }
        N[0] += J[0][0] * J[0][0] + J[1][0] * J[1][0]; N[1] += J[0][0] * J[0][1] + J[1][0] * J[1][1]; N[2] += J[0][0] * J[0][2] + J[1][0] * J[1][2];
        N[3] += J[0][1] * J[0][1] + J[1][1] * J[1][1]; N[4] += J[0][1] * J[0][2] + J[1][1] * J[1][2]; N[5] += J[0][2] * J[0][2] + J[1][2] * J[1][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) g[j] += J[0][j] * f0 + J[1][j] * f1;
    }
}

__device__ __forceinline__ double quad3(const double H[6], const double y[3])
{
    double z[3];
    if (!solve3(H, y, z)) return -1.0;
    return y[0] * z[0] + y[1] * z[1] + y[2] * z[2];
}

// MINPACK lmpar (cminpack lmpar.c) on the normal equations: R^T R = P^T N P, so every solve with the QR factor is a
// solve with N or N + par D^2
__device__ void lmpar3(const double N[6], const double g[3], const double diag[3], double delta, double &par, double x[3])
{
    const double dwarf = 2.2250738585072014e-308;
    const bool nonsing = solve3(N, g, x);
    if (!nonsing) x[0] = x[1] = x[2] = 0.0;
    double w[3] = { diag[0] * x[0], diag[1] * x[1], diag[2] * x[2] };
    double dxnorm = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { par = 0.0; return; }
    double parl = 0.0;
    if (nonsing) {
        const double y[3] = { diag[0] * (w[0] / dxnorm), diag[1] * (w[1] / dxnorm), diag[2] * (w[2] / dxnorm) };
        const double q = quad3(N, y);
        if (q > 0.0) parl = fp / delta / q;
    }
    const double s[3] = { g[0] / diag[0], g[1] / diag[1], g[2] / diag[2] };
    const double gnorm = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
    par = fmax(par, parl); par = fmin(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;
    for (int iter = 1;; ++iter) {
        if (par == 0.0) par = fmax(dwarf, 0.001 * paru);
        const double Hd[6] = { N[0] + par * diag[0] * diag[0], N[1], N[2], N[3] + par * diag[1] * diag[1], N[4], N[5] + par * diag[2] * diag[2] };
        if (!solve3(Hd, g, x)) { x[0] = x[1] = x[2] = 0.0; }
        w[0] = diag[0] * x[0]; w[1] = diag[1] * x[1]; w[2] = diag[2] * x[2];
        dxnorm = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        const double y[3] = { diag[0] * (w[0] / dxnorm), diag[1] * (w[1] / dxnorm), diag[2] * (w[2] / dxnorm) };
        const double parc = fp / delta / quad3(Hd, y);
        if (fp > 0.0) parl = fmax(parl, par);
        if (fp < 0.0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
}

// MINPACK lmdif1(tol) = lmdif(ftol = xtol = tol, gtol 0, maxfev 200 (n+1), epsfcn 0, mode 1, factor 100)
// (cminpack lmdif1.c, lmdif.c), n = 3
__device__ void lmdif3(const Views &V, int o0, int o1, double X[3], double tol)
{
    const double epsmch = 2.220446049250313e-16, factor = 100.0;
    const int maxfev = 800;
    double fnorm = sqrt(fnorm2(V, o0, o1, X)), par = 0.0, delta = 0.0, xnorm = 0.0, diag[3] = { 1.0, 1.0, 1.0 };
    int nfev = 1, iter = 1;
    for (;;) {
        double N[6], g[3];
        fd_normal(V, o0, o1, X, N, g);
        nfev += 3;
        const double acn[3] = { sqrt(N[0]), sqrt(N[3]), sqrt(N[5]) };
        if (iter == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { diag[j] = acn[j]; if (acn[j] == 0.0) diag[j] = 1.0; }
            xnorm = sqrt(diag[0] * X[0] * diag[0] * X[0] + diag[1] * X[1] * diag[1] * X[1] + diag[2] * X[2] * diag[2] * X[2]);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        double gnorm = 0.0;
        if (fnorm != 0.0)
#pragma unroll
            for (int j = 0; j < 3; ++j) if (acn[j] != 0.0) gnorm = fmax(gnorm, fabs((g[j] / fnorm) / acn[j]));
        if (gnorm <= 0.0) return;
#pragma unroll
        for (int j = 0; j < 3; ++j) diag[j] = fmax(diag[j], acn[j]);
        for (;;) {
            double x[3], p[3], Xn[3];
            lmpar3(N, g, diag, delta, par, x);
#pragma unroll
            for (int j = 0; j < 3; ++j) { p[j] = -x[j]; Xn[j] = X[j] + p[j]; }
            const double pnorm = sqrt(diag[0] * p[0] * diag[0] * p[0] + diag[1] * p[1] * diag[1] * p[1] + diag[2] * p[2] * diag[2] * p[2]);
            if (iter == 1) delta = fmin(delta, pnorm);
            const double fnorm1 = sqrt(fnorm2(V, o0, o1, Xn));
            ++nfev;
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) actred = 1.0 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
            const double Np0 = N[0] * p[0] + N[1] * p[1] + N[2] * p[2], Np1 = N[1] * p[0] + N[3] * p[1] + N[4] * p[2],
                         Np2 = N[2] * p[0] + N[4] * p[1] + N[5] * p[2];
            const double jp2 = fmax(p[0] * Np0 + p[1] * Np1 + p[2] * Np2, 0.0);
            const double temp1 = sqrt(jp2) / fnorm, temp2 = sqrt(par) * pnorm / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5, dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par /= temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
#pragma unroll
                for (int j = 0; j < 3; ++j) X[j] = Xn[j];
                xnorm = sqrt(diag[0] * X[0] * diag[0] * X[0] + diag[1] * X[1] * diag[1] * X[1] + diag[2] * X[2] * diag[2] * X[2]);
                fnorm = fnorm1;
                ++iter;
            }
            const bool small = fabs(actred) <= tol && prered <= tol && 0.5 * ratio <= 1.0;
            if (small || delta <= tol * xnorm) return;                      // info 1, 2, 3
            if (nfev >= maxfev) return;                                     // info 5
            if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) return;
            if (delta <= epsmch * xnorm || gnorm <= epsmch) return;         // info 7, 8
            if (ratio >= 1e-4) break;
        }
    }
}

// ComputeRayAngle, TriangulationMultiStep.cpp:498-530 (with its "+ centre, - centre" round trip), in degrees
__device__ __forceinline__ double ray_angle_deg(const double *__restrict__ ca, const double *__restrict__ cb, double pax, double pay, double pbx, double pby)
{
    double pw[3], qw[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double u = ca[r] * pax + ca[3 + r] * pay + ca[6 + r] * 1.0;
        const double v = cb[r] * pbx + cb[3 + r] * pby + cb[6 + r] * 1.0;
        pw[r] = (u + ca[12 + r]) - ca[12 + r];
        qw[r] = (v + cb[12 + r]) - cb[12 + r];
    }
    const double dot = pw[0] * qw[0] + pw[1] * qw[1] + pw[2] * qw[2];
    const double mag = sqrt(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]) * sqrt(qw[0] * qw[0] + qw[1] * qw[1] + qw[2] * qw[2]);
    double cs = dot / mag;
    cs = cs < -1.0 + 1.0e-8 ? -1.0 + 1.0e-8 : (cs > 1.0 - 1.0e-8 ? 1.0 - 1.0e-8 : cs);
    return acos(cs) * (180.0 / 3.14159265358979323846);
}

// POLISH is a template parameter because lmdif's state triples the register count: the common case keeps its occupancy
template <bool POLISH>
__global__ __launch_bounds__(kTriBlock) void k_triangulate(const TriArgs a)
{
    extern __shared__ double2 sDyn[];
    double2 *sPv = sDyn;
    int *sCam = reinterpret_cast<int *>(sDyn + a.stage_cap);
    const int i0 = blockIdx.x * kTriBlock, i = i0 + threadIdx.x;
    const int ob0 = a.pt_ptr[i0], ob1 = a.pt_ptr[min(i0 + kTriBlock, a.n)];
    Views V{ a, sPv, sCam, ob0, ob1 - ob0 <= a.stage_cap };
    if (V.staged) {
        for (int k = threadIdx.x; k < ob1 - ob0; k += kTriBlock) {
            const int j = a.obs_cam[ob0 + k];
            double px = 0.0, py = 0.0;
            if (j >= 0) normalise(a, a.xy[ob0 + k], a.cam[(size_t)j * kCamRec + 15], px, py);
            sCam[k] = j; sPv[k] = make_double2(px, py);
        }
        __syncthreads();
    }
    if (i >= a.n) return;
    const int o0 = a.pt_ptr[i], o1 = a.pt_ptr[i + 1];
    double X[3] = { 0.0, 0.0, 0.0 }, err = -1.0, sumsq = 0.0;
    int status = MI_TRI_FEW_VIEWS, nv = 0;
    for (int o = o0; o < o1; ++o) nv += V.cam(o) >= 0;
    if (nv >= a.min_views && nv >= 2) {
        // ONE gather of each view's camera serves two things: the normal equations of the linear system (:580-600:
        // rows (R0 - x R2) X = t2 x - t0, (R1 - y R2) X = t2 y - t1) and the ray-angle test of the pairs (first view, b)
        // (:1251-1265 asks whether ANY pair of rays is more than min_angle apart; the order it looks in does not matter)
        bool conditioned = false;
        int ofirst = -1;
        double r0[3] = { 0.0, 0.0, 0.0 }, n0 = 0.0;
        double H[6] = { 0, 0, 0, 0, 0, 0 }, g[3] = { 0, 0, 0 };
        for (int o = o0; o < o1; ++o) {
            const int j = V.cam(o);
            if (j < 0) continue;
            double cr[kCamRec], x, y, a0[3], a1[3], rw[3];
            load_cam(a.cam, j, cr);
            V.pv(o, cr[15], x, y);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a0[k] = cr[k] - x * cr[6 + k]; a1[k] = cr[3 + k] - y * cr[6 + k];
                const double u = cr[k] * x + cr[3 + k] * y + cr[6 + k] * 1.0;      // ComputeRayAngle :505-521
                rw[k] = (u + cr[12 + k]) - cr[12 + k];
            }
            const double b0 = cr[11] * x - cr[9], b1 = cr[11] * y - cr[10];
            H[0] += a0[0] * a0[0] + a1[0] * a1[0]; H[1] += a0[0] * a0[1] + a1[0] * a1[1]; H[2] += a0[0] * a0[2] + a1[0] * a1[2];
            H[3] += a0[1] * a0[1] + a1[1] * a1[1]; H[4] += a0[1] * a0[2] + a1[1] * a1[2]; H[5] += a0[2] * a0[2] + a1[2] * a1[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] += a0[k] * b0 + a1[k] * b1;
            const double nn = sqrt(rw[0] * rw[0] + rw[1] * rw[1] + rw[2] * rw[2]);
            if (ofirst < 0) { ofirst = o; r0[0] = rw[0]; r0[1] = rw[1]; r0[2] = rw[2]; n0 = nn; }
            else if (!conditioned) {
                double cs = (r0[0] * rw[0] + r0[1] * rw[1] + r0[2] * rw[2]) / (n0 * nn);
                cs = cs < -1.0 + 1.0e-8 ? -1.0 + 1.0e-8 : (cs > 1.0 - 1.0e-8 ? 1.0 - 1.0e-8 : cs);
                conditioned = acos(cs) * (180.0 / 3.14159265358979323846) > a.min_angle_deg;
            }
        }
        // rare: every ray is within the angle of the first one; the remaining pairs decide
        for (int oa = ofirst + 1; oa < o1 && !conditioned; ++oa) {
            const int ja = V.cam(oa);
            if (ja < 0) continue;
            const double *ca = a.cam + (size_t)ja * kCamRec;
            double pax, pay;
            V.pv(oa, ca[15], pax, pay);
            for (int ob = oa + 1; ob < o1; ++ob) {
                const int jb = V.cam(ob);
                if (jb < 0) continue;
                const double *cb = a.cam + (size_t)jb * kCamRec;
                double pbx, pby;
                V.pv(ob, cb[15], pbx, pby);
                if (ray_angle_deg(ca, cb, pax, pay, pbx, pby) > a.min_angle_deg) { conditioned = true; break; }
            }
        }
        status = MI_TRI_NARROW;
        if (conditioned) {
            if (!solve3(H, g, X)) { X[0] = X[1] = X[2] = 0.0; }
            if constexpr (POLISH) lmdif3(V, o0, o1, X, 1.0e-5);  // lmdif_driver(..., 1.0e-5), :626-630
            // errors (:632-661)
            double error = 0.0;
            for (int o = o0; o < o1; ++o) {
                const int j = V.cam(o);
                if (j < 0) continue;
                double cr[kCamRec], px, py, f0, f1, dz;
                load_cam(a.cam, j, cr);
                V.pv(o, cr[15], px, py);
                resid(cr, px, py, X, f0, f1, dz);
                const double v = f0 * f0 + f1 * f1, e = sqrt(v);
                error += v;
                if (dz < 0.0001) error = 10000000.0;
                sumsq += e * e * cr[15] * cr[15];
            }
            err = sqrt(error) / nv;
            status = err < a.max_error ? MI_TRI_KEPT : MI_TRI_LARGE_ERROR;
        }
    }
    const bool kept = status == MI_TRI_KEPT;
    a.xyz[3 * (size_t)i] = kept ? X[0] : 0.0; a.xyz[3 * (size_t)i + 1] = kept ? X[1] : 0.0; a.xyz[3 * (size_t)i + 2] = kept ? X[2] : 0.0;
    a.err[i] = err; a.status[i] = status; a.sumsq[i] = kept ? sumsq : 0.0; a.nviews[i] = nv;
}

template <class T>
struct Buf {
    T *p = nullptr;
    size_t n = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count)
    {
        if (p) (void)hipFree(p);
        p = nullptr; n = count;
        return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
    }
    hipError_t upload(const T *src, size_t count)
    {
        hipError_t e = alloc(count);
        if (e != hipSuccess || !count) return e;
        return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    }
};

thread_local std::string g_tri_err;

} // namespace

struct mi_tri_engine {
    mi_tri_config cfg;
    std::string err;
    int m = 0, n = 0;
    int64_t nobs = 0;
    Buf<double> cam, xy, xyz, errv, sumsq;
    Buf<int> pt_ptr, obs_cam, status, nviews;
    std::vector<int> order;       // internal position -> track (tracks sorted by their first cameras, see set_tracks)
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    bool ran = false;
    bool fail(const std::string &what, hipError_t e) { err = what + ": " + hipGetErrorString(e); return false; }
};

extern "C" {

void mi_tri_default_config(mi_tri_config *c)
{
    std::memset(c, 0, sizeof(*c));
    c->min_views = 2; c->min_angle_deg = 0.5; c->max_error = 0.01; c->float_normalise = 1; c->polish = 0; c->device = 0;
}

const char *mi_tri_last_error(const mi_tri_engine *e) { return e ? e->err.c_str() : g_tri_err.c_str(); }

mi_tri_engine *mi_tri_create(const mi_tri_config *cfg)
{
    mi_tri_config c;
    if (cfg) c = *cfg; else mi_tri_default_config(&c);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_tri_err = "no HIP device: the triangulation step runs on an MI355X only (no CPU fallback)"; return nullptr; }
    if (c.device < 0 || c.device >= ndev) { g_tri_err = "bad device index"; return nullptr; }
    if (hipSetDevice(c.device) != hipSuccess) { g_tri_err = "hipSetDevice failed"; return nullptr; }
    mi_tri_engine *e = new mi_tri_engine;
    e->cfg = c;
    if (hipStreamCreate(&e->st) != hipSuccess || hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
        g_tri_err = "cannot create stream / events"; delete e; return nullptr;
    }
    return e;
}

void mi_tri_destroy(mi_tri_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->st) (void)hipStreamDestroy(e->st);
    delete e;
}

int mi_tri_set_cameras(mi_tri_engine *e, int m, const double *f, const double *R, const double *t, const double *c)
{
    if (!e) return -1;
    if (m <= 0 || !f || !R || !t) { e->err = "set_cameras: bad argument"; return -1; }
    (void)hipSetDevice(e->cfg.device);
    std::vector<double> rec((size_t)m * kCamRec);
    for (int j = 0; j < m; ++j) {
        double *r = rec.data() + (size_t)j * kCamRec;
        std::memcpy(r, R + 9 * (size_t)j, 9 * sizeof(double));
        std::memcpy(r + 9, t + 3 * (size_t)j, 3 * sizeof(double));
        for (int k = 0; k < 3; ++k)
            r[12 + k] = c ? c[3 * (size_t)j + k] : -(r[k] * r[9] + r[3 + k] * r[10] + r[6 + k] * r[11]); // -R^T t
        r[15] = f[j];
        if (!(f[j] != 0.0)) { e->err = "set_cameras: zero focal length"; return -1; }
    }
    hipError_t he = e->cam.upload(rec.data(), rec.size());
    if (he != hipSuccess) { e->fail("upload cameras", he); return -1; }
    e->m = m; e->ran = false;
    return 0;
}

int mi_tri_set_tracks(mi_tri_engine *e, int n, const int *pt_ptr, const int *obs_cam, const double *xy)
{
    if (!e) return -1;
    if (n < 0 || !pt_ptr || (n && (!obs_cam || !xy))) { e->err = "set_tracks: bad argument"; return -1; }
    if (!e->m) { e->err = "set_tracks: set_cameras first"; return -1; }
    if (pt_ptr[0] != 0) { e->err = "set_tracks: pt_ptr[0] must be 0"; return -1; }
    for (int i = 0; i < n; ++i) if (pt_ptr[i + 1] < pt_ptr[i]) { e->err = "set_tracks: pt_ptr must be non-decreasing"; return -1; }
    const int64_t nobs = pt_ptr[n];
    for (int64_t o = 0; o < nobs; ++o) if (obs_cam[o] >= e->m) { e->err = "set_tracks: camera index out of range"; return -1; }
    (void)hipSetDevice(e->cfg.device);
    // The kernel gathers a 128-B camera record per view: in the caller's order 64 neighbouring lanes want 64 different
    // cameras and the gathers run at L2 speed.  Tracks are therefore processed in the order of their first cameras
    // (neighbours then share records in L1); the views inside a track keep their order, and results go back in the
    // caller's order, so nothing observable changes.
    e->order.resize(n);
    {
        std::vector<unsigned long long> key(n);
        for (int i = 0; i < n; ++i) {
            unsigned long long k = 0;
            int taken = 0;
            for (int o = pt_ptr[i]; o < pt_ptr[i + 1] && taken < 3; ++o)
                if (obs_cam[o] >= 0) { k |= (unsigned long long)(obs_cam[o] & 0x1fffff) << (42 - 21 * taken); ++taken; }
            key[i] = k;
            e->order[i] = i;
        }
        std::stable_sort(e->order.begin(), e->order.end(), [&](int a, int b) { return key[a] < key[b]; });
    }
    std::vector<int> pp((size_t)n + 1, 0), oc((size_t)nobs);
    std::vector<double> xx((size_t)nobs * 2);
    for (int t = 0; t < n; ++t) {
        const int i = e->order[t], o0 = pt_ptr[i], d = pt_ptr[i + 1] - o0;
        std::memcpy(oc.data() + pp[t], obs_cam + o0, sizeof(int) * (size_t)d);
        std::memcpy(xx.data() + 2 * (size_t)pp[t], xy + 2 * (size_t)o0, sizeof(double) * 2 * (size_t)d);
        pp[t + 1] = pp[t] + d;
    }
    pt_ptr = pp.data(); obs_cam = oc.data(); xy = xx.data();
    hipError_t he;
    if ((he = e->pt_ptr.upload(pt_ptr, (size_t)n + 1)) != hipSuccess || (he = e->obs_cam.upload(obs_cam, (size_t)nobs)) != hipSuccess ||
        (he = e->xy.upload(xy, (size_t)nobs * 2)) != hipSuccess || (he = e->xyz.alloc((size_t)n * 3)) != hipSuccess ||
        (he = e->errv.alloc(n)) != hipSuccess || (he = e->sumsq.alloc(n)) != hipSuccess || (he = e->status.alloc(n)) != hipSuccess ||
        (he = e->nviews.alloc(n)) != hipSuccess) { e->fail("upload tracks", he); return -1; }
    e->n = n; e->nobs = nobs; e->ran = false;
    return 0;
}

int mi_tri_run(mi_tri_engine *e)
{
    if (!e) return -1;
    if (!e->m || !e->pt_ptr.p) { e->err = "run: cameras and tracks must be set first"; return -1; }
    (void)hipSetDevice(e->cfg.device);
    e->ran = true; e->last_ms = 0.0;
    if (!e->n) return 0;
    TriArgs a{};
    a.cam = e->cam.p; a.pt_ptr = e->pt_ptr.p; a.obs_cam = e->obs_cam.p; a.xy = reinterpret_cast<const double2 *>(e->xy.p);
    a.xyz = e->xyz.p; a.err = e->errv.p; a.sumsq = e->sumsq.p; a.status = e->status.p; a.nviews = e->nviews.p;
    a.n = e->n; a.min_views = e->cfg.min_views; a.float_normalise = e->cfg.float_normalise; a.polish = e->cfg.polish;
    a.min_angle_deg = e->cfg.min_angle_deg; a.max_error = e->cfg.max_error;
    (void)hipEventRecord(e->ev0, e->st);
    int cap = kTriStage;
    if (const char *s = std::getenv("MI_TRI_STAGE")) cap = std::max(0, std::min(atoi(s), 6000));   // experiments
    a.stage_cap = cap;
    const size_t lds = (size_t)cap * (sizeof(double2) + sizeof(int));
    const dim3 grid((unsigned)((e->n + kTriBlock - 1) / kTriBlock)), block(kTriBlock);
    if (e->cfg.polish) hipLaunchKernelGGL(k_triangulate<true>, grid, block, lds, e->st, a);
    else hipLaunchKernelGGL(k_triangulate<false>, grid, block, lds, e->st, a);
    (void)hipEventRecord(e->ev1, e->st);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipStreamSynchronize(e->st);
    if (he != hipSuccess) { e->fail("k_triangulate", he); return -1; }
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->ev0, e->ev1) == hipSuccess) e->last_ms = ms;
    return 0;
}

int mi_tri_get(mi_tri_engine *e, double *xyz, double *err, int *status, int64_t counts[4], double *sum_sq_px, int64_t *nproj)
{
    if (!e) return -1;
    if (!e->ran) { e->err = "get: run first"; return -1; }
    (void)hipSetDevice(e->cfg.device);
    const size_t n = (size_t)e->n;
    std::vector<int> st(n), nv(n);
    std::vector<double> ss(n), X(n * 3), er(n);
    hipError_t he = hipSuccess;
    if (n) {
        he = hipMemcpy(X.data(), e->xyz.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(er.data(), e->errv.p, n * sizeof(double), hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(st.data(), e->status.p, n * sizeof(int), hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(nv.data(), e->nviews.p, n * sizeof(int), hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(ss.data(), e->sumsq.p, n * sizeof(double), hipMemcpyDeviceToHost);
        if (he != hipSuccess) { e->fail("download", he); return -1; }
    }
    // back to the caller's order; the reference's counters and its running sum in track order (:1303)
    std::vector<int> where(n);
    for (size_t t = 0; t < n; ++t) where[(size_t)e->order[t]] = (int)t;
    int64_t c1 = 0, c2 = 0, c3 = 0, np = 0;
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const size_t t = (size_t)where[i];
        if (xyz) { xyz[3 * i] = X[3 * t]; xyz[3 * i + 1] = X[3 * t + 1]; xyz[3 * i + 2] = X[3 * t + 2]; }
        if (err) err[i] = er[t];
        if (status) status[i] = st[t];
        if (st[t] != MI_TRI_FEW_VIEWS) ++c1;
        if (st[t] >= MI_TRI_LARGE_ERROR) ++c2;
        if (st[t] == MI_TRI_KEPT) { ++c3; sum += ss[t]; np += nv[t]; }
    }
    if (counts) { counts[0] = c1; counts[1] = c2; counts[2] = c3; counts[3] = c3; }
    if (sum_sq_px) *sum_sq_px = sum;
    if (nproj) *nproj = np;
    return 0;
}

double mi_tri_last_ms(const mi_tri_engine *e) { return e ? e->last_ms : 0.0; }

double mi_tri_algorithmic_bytes(const mi_tri_engine *e)
{
    if (!e) return 0.0;
    // 16 B measurement + 4 B camera index per view, CSR pointer, camera table once, 52 B of results per track
    return 20.0 * (double)e->nobs + 4.0 * ((double)e->n + 1) + 128.0 * e->m + 52.0 * (double)e->n;
}

int mi_triangulate_tracks(const mi_tri_config *cfg, int m, const double *f, const double *R, const double *t, const double *c,
                          int n, const int *pt_ptr, const int *obs_cam, const double *xy,
                          double *xyz, double *err, int *status, int64_t counts[4], double *sum_sq_px, int64_t *nproj)
{
    mi_tri_engine *e = mi_tri_create(cfg);
    if (!e) return -1;
    int rc = -1;
    if (mi_tri_set_cameras(e, m, f, R, t, c) == 0 && mi_tri_set_tracks(e, n, pt_ptr, obs_cam, xy) == 0 && mi_tri_run(e) == 0 &&
        mi_tri_get(e, xyz, err, status, counts, sum_sq_px, nproj) == 0) rc = 0;
    else g_tri_err = e->err;
    mi_tri_destroy(e);
    return rc;
}

} // extern "C"
