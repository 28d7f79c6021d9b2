// kernels.hip -- hand-written HIP kernels of the bundle-adjustment engine, gfx950 (CDNA4) only.
//
// One LM damping trial (sba_levmar.c:1131-1612) is
//   k_cam_prep -> k_build (-> RCCL all-reduce) -> k_finalize{,_vec} -> k_block_inv
//   -> k_scale_blocks -> CG (k_pcg_init, k_pcg_start, k_pcg_iter per iteration, k_pcg_finish) -> k_cam_update -> k_backsub (-> all-reduce)
// All arithmetic is IEEE double.  Wavefront = 64 lanes; k_build and k_backsub use ONE wavefront
// per workgroup so every LDS hand-off is wave-synchronous (the s_barrier of a one-wave workgroup
// costs nothing) and no inter-wave protocol exists anywhere in this file; cross-workgroup
// accumulation is by f64 atomics into zero-initialised HBM buffers only.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace miba {

// layout of a row-path observation record (doubles): A row 0 | A row 1 (AP each) | B (6) | e (2) | G = B V*^-1 (6) | e - B h (2)
template <int CNP>
struct RecL {
    static constexpr int AP = CNP <= 8 ? 8 : 10, A0 = 0, A1 = AP, B = 2 * AP, E = 2 * AP + 6, G = 2 * AP + 8, ER = 2 * AP + 14, SIZE = 2 * AP + 16;
};
static_assert(RecL<7>::SIZE == kRec && RecL<9>::SIZE == kRec9, "record sizes (plan.h)");

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// Wave-wide sum / maximum, the result in every lane.  Inside a row of 16 lanes the partner's value comes through DPP (quad
// permutes, then the half-row and row mirrors), the four row totals through v_readlane: ~20 short instructions.  (The
// butterfly of six __shfl_xor = twelve dependent ds_bpermute took ~0.5 us per double; the small kernels of a trial do
// dozens of these.)  Call with all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror
    v += dpp_move<0x140>(v);   // row_mirror: every lane holds its row's total
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_move<0xB1>(v));
    v = fmax(v, dpp_move<0x4E>(v));
    v = fmax(v, dpp_move<0x141>(v));
    v = fmax(v, dpp_move<0x140>(v));
    return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
}
// max of non-negative doubles through the monotone bit pattern; NaN compares above everything
__device__ __forceinline__ void atomic_max_pos(double *p, double v)
{
    atomicMax(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v));
}
// LDS hand-off inside a ONE-wavefront workgroup: the LDS unit executes a wave's DS instructions in
// issue order, so all that is needed is that the compiler keeps the order and that outstanding DS
// results have landed.  Unlike __syncthreads() this does NOT wait for vmcnt(0): global loads issued
// for the next batch and the ptout / pts_new stores stay in flight across the phases.
__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// 1 / x for the per-observation kernels: v_rcp_f64 and two Newton steps (5 instructions, ~1 ulp) instead of the IEEE division
// sequence (13, v_div_scale / v_div_fmas / v_div_fixup around the same iteration); x is a depth or a determinant, never denormal
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ void gadd(double *p, double v) { unsafeAtomicAdd(p, v); } // global_atomic_add_f64
__device__ __forceinline__ void ladd(double *p, double v)                           // ds_add_f64
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS-DMA (global_load_lds_*): a wave instruction that moves 16 (or 4) bytes per active lane from the lane's global address straight
// into LDS at lds_dst + lane * 16 (or 4) -- no destination registers, counted in vmcnt.  lds_dst is wave-uniform.  Through inline asm:
// with the builtin the compiler drains the transfers (s_waitcnt vmcnt(0)) in front of the next LDS read of ANY array, which exposes
// the HBM latency a prefetch is there to hide; the kernels wait for their transfers themselves.  (M0 is the compiler's: saved, put back.)
__device__ __forceinline__ void glds_b128(const void *src, const void *lds_dst)
{
    const unsigned m0v = (unsigned)(size_t)lds_dst;       // LDS byte address: the low 32 bits of the generic pointer
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
}
__device__ __forceinline__ void glds_b32(const void *src, const void *lds_dst)
{
    const unsigned m0v = (unsigned)(size_t)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
}
__device__ __forceinline__ void wait_lds_dma()            // every transfer (and store) this wave has issued has landed
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------
// camera model: projection (sfm.c:303-374) and its analytic Jacobian (the reference uses forward
// differences, sba_levmar_wrap.c:163-259; the analytic form agrees to ~1e-6 relative, see
// tests/test_oracle_golden.py::test_analytic_jacobian_close_to_fd)
// ------------------------------------------------------------------------------------------
// undistort == 2 is PBA's measurement distortion (ComputeProjection with radial == -1, SparseBundleCPU.cpp:1188-1193):
// residual x (1 + k |x|^2) - f q.xy / q.z, written as x - xhat with xhat = f q.xy / q.z - k |x|^2 x
// FULL = false is the lean instantiation of the common case (no measurement distortion): the hot kernels are register
// bound, and carrying the rarely used model through them costs occupancy
template <bool FULL>
__device__ __forceinline__ void project_obs(const ModelCfg &cfg, const double *__restrict__ cl, const double X[3],
                                            double2 xy, double &u, double &v)
{
    double q[3];
    if (cfg.explicit_centers) {
        double d0 = X[0] - cl[18], d1 = X[1] - cl[19], d2 = X[2] - cl[20];
#pragma unroll
        for (int r = 0; r < 3; ++r) q[r] = cl[3 * r] * d0 + cl[3 * r + 1] * d1 + cl[3 * r + 2] * d2;
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) q[r] = cl[3 * r] * X[0] + cl[3 * r + 1] * X[1] + cl[3 * r + 2] * X[2] + cl[18 + r];
    }
    const bool md = FULL && cfg.undistort == 2;
    double f = cl[21];
    double iz = 1.0 / q[2], mx = q[0] * iz, my = q[1] * iz;
    double g = cfg.proj_sign * f;
    if constexpr (FULL) {
        const double k1 = cl[22], k2 = cl[23], r2 = mx * mx + my * my;
        if (!md) g *= 1.0 + k1 * r2 + k2 * r2 * r2;
    }
    u = g * mx; v = g * my;
    if (md) { const double kk = cl[22] * (xy.x * xy.x + xy.y * xy.y); u -= kk * xy.x; v -= kk * xy.y; }
}

template <int CNP, bool FULL>
__device__ __forceinline__ void linearize_obs(const ModelCfg &cfg, const double *__restrict__ cl, const double X[3],
                                              double2 xy, bool fixed_cam, double e[2], double A[2][CNP], double B[2][3])
{
    double R[9], q[3], qr[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = cl[i];
    const bool md = FULL && cfg.undistort == 2;
    const double m2k = md ? (xy.x * xy.x + xy.y * xy.y) : 0.0;
    const double c0 = cl[18], c1 = cl[19], c2 = cl[20], f = cl[21];
    const double k1 = (FULL && !md) ? cl[22] : 0.0, k2 = FULL ? cl[23] : 0.0;   // lean: no distortion terms at all
    if (cfg.explicit_centers) {
        double d0 = X[0] - c0, d1 = X[1] - c1, d2 = X[2] - c2;
#pragma unroll
        for (int r = 0; r < 3; ++r) { qr[r] = R[3 * r] * d0 + R[3 * r + 1] * d1 + R[3 * r + 2] * d2; q[r] = qr[r]; }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) qr[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2];
        q[0] = qr[0] + c0; q[1] = qr[1] + c1; q[2] = qr[2] + c2;
    }
    const double s = cfg.proj_sign, iz = fast_rcp(q[2]);
    const double mx = q[0] * iz, my = q[1] * iz, r2 = mx * mx + my * my;
    const double D = FULL ? 1.0 + k1 * r2 + k2 * r2 * r2 : 1.0, Dp = FULL ? k1 + 2.0 * k2 * r2 : 0.0;
    e[0] = xy.x - s * f * mx * D;
    e[1] = xy.y - s * f * my * D;
    if (md) { const double kk = cl[22] * m2k; e[0] = fma(kk, xy.x, e[0]); e[1] = fma(kk, xy.y, e[1]); }
    const double dmx[3] = { iz, 0.0, -mx * iz }, dmy[3] = { 0.0, iz, -my * iz };
    double Pq[2][3];
    if constexpr (FULL) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            double dr2 = 2.0 * (mx * dmx[t] + my * dmy[t]);
            Pq[0][t] = s * f * (D * dmx[t] + mx * Dp * dr2);
            Pq[1][t] = s * f * (D * dmy[t] + my * Dp * dr2);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 3; ++t) B[r][t] = Pq[r][0] * R[t] + Pq[r][1] * R[3 + t] + Pq[r][2] * R[6 + t];
    } else {
        // no distortion: d xhat / d q = s f [iz 0 -mx iz; 0 iz -my iz] -- the two zeros written out (the general form multiplies by
        // them at run time: IEEE does not let a compiler drop x * 0); the same values to the bit
        const double sf = s * f;
        Pq[0][0] = sf * dmx[0]; Pq[0][1] = 0.0; Pq[0][2] = sf * dmx[2];
        Pq[1][0] = 0.0; Pq[1][1] = sf * dmy[1]; Pq[1][2] = sf * dmy[2];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            B[0][t] = fma(Pq[0][2], R[6 + t], Pq[0][0] * R[t]);
            B[1][t] = fma(Pq[1][2], R[6 + t], Pq[1][1] * R[3 + t]);
        }
    }
    if (fixed_cam) { // A_ij is taken as zero for the first mcon cameras (sba_levmar.c:460-462)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < CNP; ++t) A[r][t] = 0.0;
        if (cfg.points_fixed) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < 3; ++t) B[r][t] = 0.0;
        }
        return;
    }
    // M = -[qr]x Jl
    const double *Jl = cl + 9;
    double M[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        M[0][t] = -(-qr[2] * Jl[3 + t] + qr[1] * Jl[6 + t]);
        M[1][t] = -(qr[2] * Jl[t] - qr[0] * Jl[6 + t]);
        M[2][t] = -(-qr[1] * Jl[t] + qr[0] * Jl[3 + t]);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            A[r][t] = cfg.explicit_centers ? -B[r][t] : Pq[r][t];
            if constexpr (FULL) A[r][3 + t] = Pq[r][0] * M[0][t] + Pq[r][1] * M[1][t] + Pq[r][2] * M[2][t];
            else A[r][3 + t] = fma(Pq[r][2], M[2][t], Pq[r][r] * M[r][t]);       // (the zero of Pq written out, as above)
        }
        // intrinsics columns, in Bundler's order (sfm.c:679-701): focal length (scaled), then k1 (scaled)
        if constexpr (CNP >= 7) {
            const double mr = (r == 0) ? mx : my;
            int idx = 6;
            if (!FULL || cfg.est_focal) { A[r][6] = s * mr * D * cfg.inv_f_scale; idx = 7; }   // lean: cnp 7 = pose + focal
            if (FULL && cfg.undistort) {
                const double dk = md ? -m2k * (r == 0 ? xy.x : xy.y) * cfg.inv_k_scale     // JacobianOne, SparseBundleCPU.cpp:1427-1432
                                     : s * f * mr * r2 * cfg.inv_k_scale;
                if (idx == 6) A[r][6] = dk;
                if constexpr (CNP >= 8) { if (idx == 7) A[r][7] = dk; }
                if (!md && cfg.nk == 2) {                                              // Bundler's second coefficient: d D / d k2 = r^4
                    const double dk2 = s * f * mr * r2 * r2 * cfg.inv_k_scale;
                    if constexpr (CNP >= 8) { if (idx == 6) A[r][7] = dk2; }
                    if constexpr (CNP >= 9) { if (idx == 7) A[r][8] = dk2; }
                }
            }
        }
    }
    if (cfg.points_fixed) { // after A (A_C = -B above): no structure block in a motion-only bundle
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 3; ++t) B[r][t] = 0.0;
    }
}

// Bundler's "residuals changed by less than eps5" exit (sba_levmar.c:1552-1572) only ever compares the largest relative change
// with eps5, so a trial needs ONE bit of it: did some residual component change by eps5 or more?  |e - en| / |e| >= eps5 is
// tested as |e - en| >= eps5 |e| -- no f64 division per observation (two of them were a tenth of k_backsub's instructions) --
// with the reference's quirks kept: the signed comparison that exempts a component (:1554), 0 / 0 skipped, x / 0 = a change.
// SC_MAX_PCT carries +inf ("some component moved") or 0; "SC_MAX_PCT < eps5" reads as before for every eps5.
__device__ __forceinline__ double pct_change_flag(double e, double en, double eps5)
{
    const double dd = fabs(e - en), ae = fabs(e);
    const bool exempt = (e < eps5) & (en < eps5);                                 // the reference's signed test
    const bool skip = ((dd == 0.0) & (ae == 0.0)) | (ae == __builtin_inf());      // 0 / 0 and x / inf: NaN or 0 there, never the maximum
    const bool moved = dd >= eps5 * ae;                                           // false when a residual is NaN: the reference's max skips NaN too
    return (moved & !exempt & !skip) ? __builtin_inf() : 0.0;                     // (no branches: selects)
}

// What the back-substitution needs of an observation: the residual e, and t = B^T (A da) -- the camera block A only
// through its product with the camera's update.  With the per-camera vectors v0 = dq/d(centre or translation) . dC
// (= -R dC with explicit centres, dt otherwise) and u = J_l(w) dw, A da = Pq (v0 - qr x u) + the intrinsics columns,
// so neither A nor M = -[qr]x J_l is formed (a third of linearize_obs' arithmetic).
struct CamDelta { double v0[3], u[3], df, dk, dk2; };   // df, dk, dk2: the updates of the focal-length and distortion parameters (internal units)

__device__ __forceinline__ CamDelta cam_delta(const ModelCfg &cfg, const double *__restrict__ cl, const double *__restrict__ da, bool fixed_cam, bool full)
{
    CamDelta d;
    if (fixed_cam) { d.v0[0] = d.v0[1] = d.v0[2] = d.u[0] = d.u[1] = d.u[2] = d.df = d.dk = d.dk2 = 0.0; return d; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d.v0[r] = cfg.explicit_centers ? -(cl[3 * r] * da[0] + cl[3 * r + 1] * da[1] + cl[3 * r + 2] * da[2]) : da[r];
        d.u[r] = cl[9 + 3 * r] * da[3] + cl[9 + 3 * r + 1] * da[4] + cl[9 + 3 * r + 2] * da[5];
    }
    const bool has_f = !full || cfg.est_focal;
    d.df = has_f ? da[6] : 0.0;
    d.dk = (full && cfg.undistort) ? da[has_f ? 7 : 6] : 0.0;
    d.dk2 = (full && cfg.undistort == 1 && cfg.nk == 2) ? da[has_f ? 8 : 7] : 0.0;   // da holds 9 entries (the ninth through the split rows)
    return d;
}

template <int CNP, bool FULL>
__device__ __forceinline__ void backsub_obs(const ModelCfg &cfg, const double *__restrict__ cl, const CamDelta &cd, const double X[3], double2 xy,
                                            double e[2], double t[3], double bd[3])
{
    double q[3], qr[3];
    const bool md = FULL && cfg.undistort == 2;
    const double m2k = md ? (xy.x * xy.x + xy.y * xy.y) : 0.0;
    const double f = cl[21];
    const double k1 = (FULL && !md) ? cl[22] : 0.0, k2 = FULL ? cl[23] : 0.0;
    if (cfg.explicit_centers) {
        const double d0 = X[0] - cl[18], d1 = X[1] - cl[19], d2 = X[2] - cl[20];
#pragma unroll
        for (int r = 0; r < 3; ++r) { qr[r] = cl[3 * r] * d0 + cl[3 * r + 1] * d1 + cl[3 * r + 2] * d2; q[r] = qr[r]; }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) { qr[r] = cl[3 * r] * X[0] + cl[3 * r + 1] * X[1] + cl[3 * r + 2] * X[2]; q[r] = qr[r] + cl[18 + r]; }
    }
    const double s = cfg.proj_sign, iz = fast_rcp(q[2]);
    const double mx = q[0] * iz, my = q[1] * iz, r2 = mx * mx + my * my;
    const double D = FULL ? 1.0 + k1 * r2 + k2 * r2 * r2 : 1.0, Dp = FULL ? k1 + 2.0 * k2 * r2 : 0.0;
    e[0] = xy.x - s * f * mx * D;
    e[1] = xy.y - s * f * my * D;
    if (md) { const double kk = cl[22] * m2k; e[0] = fma(kk, xy.x, e[0]); e[1] = fma(kk, xy.y, e[1]); }
    const double dmx[3] = { iz, 0.0, -mx * iz }, dmy[3] = { 0.0, iz, -my * iz };
    double Pq[2][3];
    if constexpr (FULL) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double dr2 = 2.0 * (mx * dmx[c] + my * dmy[c]);
            Pq[0][c] = s * f * (D * dmx[c] + mx * Dp * dr2);
            Pq[1][c] = s * f * (D * dmy[c] + my * Dp * dr2);
        }
    } else {   // (the zeros written out: linearize_obs)
        const double sf = s * f;
        Pq[0][0] = sf * dmx[0]; Pq[0][1] = 0.0; Pq[0][2] = sf * dmx[2];
        Pq[1][0] = 0.0; Pq[1][1] = sf * dmy[1]; Pq[1][2] = sf * dmy[2];
    }
    // dq = v0 + M u with M u = -(qr x u)
    const double dq[3] = { cd.v0[0] - (qr[1] * cd.u[2] - qr[2] * cd.u[1]), cd.v0[1] - (qr[2] * cd.u[0] - qr[0] * cd.u[2]),
                           cd.v0[2] - (qr[0] * cd.u[1] - qr[1] * cd.u[0]) };
    double w[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double mr = (r == 0) ? mx : my;
        if constexpr (FULL) w[r] = Pq[r][0] * dq[0] + Pq[r][1] * dq[1] + Pq[r][2] * dq[2];
        else w[r] = fma(Pq[r][2], dq[2], Pq[r][r] * dq[r]);
        if constexpr (CNP >= 7) {
            w[r] = fma(s * mr * D * cfg.inv_f_scale, cd.df, w[r]);       // zero update when the focal length is not a parameter
            if (FULL) w[r] = fma(md ? -m2k * (r == 0 ? xy.x : xy.y) * cfg.inv_k_scale : s * f * mr * r2 * cfg.inv_k_scale, cd.dk, w[r]);
            if (FULL) w[r] = fma(s * f * mr * r2 * r2 * cfg.inv_k_scale, cd.dk2, w[r]);   // zero update unless k2 is a parameter
        }
    }
    if (cfg.points_fixed) { t[0] = t[1] = t[2] = bd[0] = bd[1] = bd[2] = 0.0; return; }      // a motion-only bundle has no structure block
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double B0 = FULL ? Pq[0][0] * cl[c] + Pq[0][1] * cl[3 + c] + Pq[0][2] * cl[6 + c] : fma(Pq[0][2], cl[6 + c], Pq[0][0] * cl[c]);
        const double B1 = FULL ? Pq[1][0] * cl[c] + Pq[1][1] * cl[3 + c] + Pq[1][2] * cl[6 + c] : fma(Pq[1][2], cl[6 + c], Pq[1][1] * cl[3 + c]);
        t[c] = B0 * w[0] + B1 * w[1];
        bd[c] = fma(B1, B1, B0 * B0);                                  // this observation's share of diag V_i (PBA's damping metric)
    }
}

// the parameters of REAL camera j out of a camera-side array of kCP-double rows: row j, or -- nine parameters -- rows 2j and 2j + 1
__device__ __forceinline__ void load_cam_params(const ModelCfg &cfg, const double *__restrict__ arr, int j, double a[9])
{
    const double *row = arr + (size_t)(cfg.split ? 2 * j : j) * kCP;
#pragma unroll
    for (int r = 0; r < kCP; ++r) a[r] = row[r];
    a[8] = cfg.split ? row[kCP] : 0.0;
}

// exp([w]x) R0 exactly as rot_update (sfm.c:78-117), and the left Jacobian of SO(3) at w
__device__ __forceinline__ void cam_record(const ModelCfg &cfg, int cnp, const double *__restrict__ a,
                                           const double *__restrict__ R0, double f_fix, double *__restrict__ cl)
{
    const double w0 = a[3], w1 = a[4], w2 = a[5];
    const double t2 = w0 * w0 + w1 * w1 + w2 * w2, theta = sqrt(t2);
    double R[9];
    if (theta == 0.0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = R0[i];
    } else {
        const double n0 = w0 / theta, n1 = w1 / theta, n2 = w2 / theta;
        const double nx[9] = { 0.0, -n2, n1, n2, 0.0, -n0, -n1, n0, 0.0 };
        const double sn = sin(theta), omc = 1.0 - cos(theta);
        double dR[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double sq = nx[3 * r] * nx[c] + nx[3 * r + 1] * nx[3 + c] + nx[3 * r + 2] * nx[6 + c];
                dR[3 * r + c] = (r == c ? 1.0 : 0.0) + sn * nx[3 * r + c] + omc * sq;
            }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                R[3 * r + c] = dR[3 * r] * R0[c] + dR[3 * r + 1] * R0[3 + c] + dR[3 * r + 2] * R0[6 + c];
    }
    double ca, cb;
    if (t2 < 1e-8) { ca = 0.5 - t2 / 24.0; cb = 1.0 / 6.0 - t2 / 120.0; }
    else { ca = (1.0 - cos(theta)) / t2; cb = (theta - sin(theta)) / (t2 * theta); }
    const double wx[9] = { 0.0, -w2, w1, w2, 0.0, -w0, -w1, w0, 0.0 };
#pragma unroll
    for (int i = 0; i < 9; ++i) cl[i] = R[i];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double sq = wx[3 * r] * wx[c] + wx[3 * r + 1] * wx[3 + c] + wx[3 * r + 2] * wx[6 + c];
            cl[9 + 3 * r + c] = (r == c ? 1.0 : 0.0) + ca * wx[3 * r + c] + cb * sq;
        }
    cl[18] = a[0]; cl[19] = a[1]; cl[20] = a[2];
    cl[21] = cfg.est_focal ? a[6] / cfg.f_scale : f_fix;
    cl[22] = cfg.undistort ? a[cfg.est_focal ? 7 : 6] / cfg.k_scale : 0.0; // k1 (sfm.c:364, TEST_FOCAL scaling)
    cl[23] = (cfg.undistort == 1 && cfg.nk == 2) ? a[cfg.est_focal ? 8 : 7] / cfg.k_scale : 0.0;   // k2 (a: 9 entries then)
    (void)cnp;
}

__global__ void k_cam_prep(int m, int cnp, ModelCfg cfg, const double *__restrict__ cam_a, const double *__restrict__ R0,
                           const double *__restrict__ f_fixed, double *__restrict__ cam_lin)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double a[9];
    load_cam_params(cfg, cam_a, j, a);
    cam_record(cfg, cnp, a, R0 + (size_t)j * 9, f_fixed ? f_fixed[j] : 1.0, cam_lin + (size_t)j * kCamLin);
}

// ------------------------------------------------------------------------------------------
// k_build: fused linearise + Schur complement, one damping trial.
//
//   workgroup = one 64-lane wavefront = one TILE (a run of batches; plan.h)
//   phase 1  lane = observation: gather camera record + point, residual e, A (2 x CNP), B (2 x 3)
//            -> LDS                                     [sba_levmar.c:856, 919-1030]
//   phase 2  lane = (point, component): V_i (6 upper), eb_i (3) summed from LDS
//   phase 3  lane = point: V*_i = V_i + mu I, closed-form inverse, h_i = V*^-1 eb_i  [:1138-1162]
//   phase 4  lane = observation: ea_j, E_j = A^T (e - B h_i), diag U_j               [:1322-1338]
//   phase 5  lane = observation PAIR (a <= b) of one point:
//               block(j_a, j_b) += A_a^T (delta_ab I - G_a B_b^T) A_b,   G_a = B_a V*^-1
//            which is delta_ab U - Y_ij W_ik^T of sba_levmar.c:1182-1301 written through the
//            2 x 2 matrix G_a B_b^T (measurements are 2-vectors, so every block has rank <= 2
//            and W_ij = A^T B never has to be formed or stored).
//   LDS mode:  pair entries come in rounds of 64 (plan.h); a lane keeps "its" CNP x CNP block in
//              registers for as long as consecutive rounds give it the same block and spills it
//              into the tile's LDS table (ds_add_f64) only when the block changes; the table is
//              flushed once per tile as contiguous f64 atomics (one S block = one wave instruction).
//   row tiles (points whose camera set does not repeat, or with > kMaxViewsLds views): phases 1-4 only; phase 5 leaves one
//              record per observation in HBM (A | B | e | G = B V*^-1 | e - B h) and k_schur_blocks sums the blocks and the
//              camera-side vectors from them -- no atomics.
// ------------------------------------------------------------------------------------------
// what one lane needs before it can start a batch; filled one batch ahead (software pipeline:
// the kernel runs one wavefront per SIMD, so nothing else hides the global-load latency)
template <bool LDS>
struct BatchRegs {
    int o0, nob, p0, npt;
    long long q0, q1;
    int j, ip, cs;
    double2 xy;
};

// batch descriptors of a whole tile (<= 32 batches) sit in registers, one batch per lane, loaded once in
// the prologue; the loop reads them with v_readlane (no scalar-load latency inside the loop)
struct TileDesc { int obs, pt, pair_lo, pair_hi; };

__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }

template <bool LDS>
__device__ __forceinline__ void load_batch_level1(const BuildArgs &g, const TileDesc &td, int k, int lane, BatchRegs<LDS> &r)
{
    r.o0 = rl(td.obs, k); r.nob = rl(td.obs, k + 1) - r.o0;
    r.p0 = rl(td.pt, k); r.npt = rl(td.pt, k + 1) - r.p0;
    const long long pq = ((long long)rl(td.pair_hi, k) << 32) | (unsigned)rl(td.pair_lo, k);   // first entry | rounds << 48 (plan.h)
    r.q0 = pq & 0xFFFFFFFFFFFFll;
    r.q1 = r.q0 + (pq >> 48) * kWave;
    r.j = 0; r.ip = r.p0; r.cs = 0; r.xy = make_double2(0.0, 0.0);
    if (lane < r.nob) {
        const int o = r.o0 + lane;
        r.j = g.obs_cam[o]; r.ip = g.obs_pt[o]; r.xy = g.obs_xy[o];
        if constexpr (LDS) r.cs = g.obs_cslot[o];
    }
}

__device__ __forceinline__ void load_cam_record(const BuildArgs &g, int j, double cl[kCamLin])
{
    const double2 *src = reinterpret_cast<const double2 *>(g.cam_lin + (size_t)j * kCamLin);
#pragma unroll
    for (int k = 0; k < kCamLin / 2; ++k) { const double2 v = src[k]; cl[2 * k] = v.x; cl[2 * k + 1] = v.y; }
}

// one pair: T = (delta_ab I - G_a B_b^T) A_b with G_a = B_a V*^-1 of the pair's point.
// Rows and sv are 16-byte aligned: every operand comes in through ds_read_b128.
template <int CNP, int SR>
__device__ __forceinline__ void pair_block(const double *sR, const double *sv, int la, int lb, double T0[CNP], double T1[CNP])
{
    constexpr int OB = 2 * CNP;
    const double2 *Ba2 = reinterpret_cast<const double2 *>(sR + la * SR + OB);
    const double2 *Bb2 = reinterpret_cast<const double2 *>(sR + lb * SR + OB);
    const double2 *sv2 = reinterpret_cast<const double2 *>(sv);
    const double2 a01 = Ba2[0], a23 = Ba2[1], a45 = Ba2[2], b01 = Bb2[0], b23 = Bb2[1], b45 = Bb2[2];
    const double2 v01 = sv2[0], v23 = sv2[1], v45 = sv2[2];
    const double Ba[6] = { a01.x, a01.y, a23.x, a23.y, a45.x, a45.y }, Bb[6] = { b01.x, b01.y, b23.x, b23.y, b45.x, b45.y };
    const double i00 = v01.x, i01 = v01.y, i02 = v23.x, i11 = v23.y, i12 = v45.x, i22 = v45.y;
    double G[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        G[r][0] = fma(Ba[3 * r + 2], i02, fma(Ba[3 * r + 1], i01, Ba[3 * r] * i00));
        G[r][1] = fma(Ba[3 * r + 2], i12, fma(Ba[3 * r + 1], i11, Ba[3 * r] * i01));
        G[r][2] = fma(Ba[3 * r + 2], i22, fma(Ba[3 * r + 1], i12, Ba[3 * r] * i02));
    }
    double M00 = fma(G[0][2], Bb[2], fma(G[0][1], Bb[1], G[0][0] * Bb[0]));
    double M01 = fma(G[0][2], Bb[5], fma(G[0][1], Bb[4], G[0][0] * Bb[3]));
    double M10 = fma(G[1][2], Bb[2], fma(G[1][1], Bb[1], G[1][0] * Bb[0]));
    double M11 = fma(G[1][2], Bb[5], fma(G[1][1], Bb[4], G[1][0] * Bb[3]));
    const double dg = (la == lb) ? 1.0 : 0.0;
    M00 = dg - M00; M01 = -M01; M10 = -M10; M11 = dg - M11;
    const double2 *Ab2 = reinterpret_cast<const double2 *>(sR + lb * SR);
    double Ab[2 * CNP];
#pragma unroll
    for (int k = 0; k < CNP; ++k) { const double2 v = Ab2[k]; Ab[2 * k] = v.x; Ab[2 * k + 1] = v.y; }
#pragma unroll
    for (int c = 0; c < CNP; ++c) {
        T0[c] = fma(M01, Ab[CNP + c], M00 * Ab[c]);
        T1[c] = fma(M11, Ab[CNP + c], M10 * Ab[c]);
    }
}

// Diagnostic build only (make stamp): per-phase cycle shares through s_memtime, summed per
// workgroup into g.dbg (cdna_hip_programming.md section 7, "In-kernel stamps").  Never timed.
#ifdef MIBA_STAMPS
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        stamp_acc[i] += t_ - stamp_prev;                                                           \
        stamp_prev = t_;                                                                           \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// TABLE (only with LDS): the tile's blocks are summed in an LDS-resident table by ds_add_f64 and handed to HBM once at
// the end of the tile, for tiles whose points keep changing camera set (plan.h kTableRuns); the register scheme would
// flush to HBM almost every round there and the L2's f64 atomic rate (~120 G elements/s) would be the bound
template <int CNP, bool LDS, bool FULL, bool TABLE>
__global__ __launch_bounds__(64, (CNP <= 7 && LDS && !TABLE ? 2 : 1)) void k_build(const BuildArgs g) // two wavefronts per SIMD (98 accumulator VGPRs, <= 20 KB of LDS) up to 7 parameters; 8x8 blocks: 128 accumulator VGPRs, one
{
    if constexpr (LDS) { if (g.tile_mode[g.tile0 + blockIdx.x] != (TABLE ? 2 : 1)) return; }
    if (g.spec && g.spec[0] == 0.0) return;   // enqueued ahead of the host's decision, and the trial was not accepted
    const double mu_lin = g.spec ? g.spec[1] : g.mu;
#ifdef MIBA_STAMPS
    unsigned long long stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    // one 16-byte aligned LDS row per observation: A (2 x CNP) | B (6) | e (2).  Row stride SR doubles with
    // SR/2 odd (22 doubles = 176 B): 16 distinct rows fall into 16 distinct 16-byte bank slots, so the
    // pair phase reads its operands with conflict-free ds_read_b128.
    constexpr int OB = 2 * CNP;               // offset of B inside the row (even)
    constexpr int SR = (((OB + 8) / 2) % 2 == 1) ? OB + 8 : OB + 10;
    constexpr int SV = 10;                    // sV row: Vinv (6) h (3) pad; 80 B, 16-byte aligned
    constexpr int BLK = CNP * CNP, SC = (3 * CNP) | 1;
    __shared__ __attribute__((aligned(16))) double sR[kWave * SR];
    __shared__ __attribute__((aligned(16))) double sV[kMaxPtsBatch * SV];
    // LDS budget: <= 20 KB so that TWO tiles share a SIMD (8 workgroups per CU).  The S blocks live in
    // registers; sStage only transposes them (kStageLanes lanes at a time) into the contiguous 49-double
    // shape a global f64 atomic wants.
    constexpr int kStageLanes = 5;
    // sC: the camera slots' sums; a tile with <= kCamLds cameras (one camera set: the common case) also keeps its cameras' 192-B
    // records behind them, fetched once per tile in coalesced pieces instead of one divergent gather per observation and batch
    constexpr int kCamLds = 12;
    constexpr int SCN = LDS ? (kNCSlot * SC > kCamLds * (SC + kCamLin) ? kNCSlot * SC : kCamLds * (SC + kCamLin)) : 2;
    __shared__ double sStage[(TABLE || !LDS) ? 1 : kStageLanes * BLK];
    __shared__ __attribute__((aligned(16))) double sC[SCN];
    __shared__ double sTab[TABLE ? kNSlot * BLK : 1];
    __shared__ int sStart[kMaxPtsBatch + 1];  // first lane of every point of the batch
    __shared__ int sFix[LDS ? kNCSlot : 1];   // per camera slot: the camera is held fixed
#ifdef MIBA_OCC1   // probe build (make occ1): 24 KB of unused LDS = ONE tile per SIMD, to see what the second wavefront buys
    __shared__ double sPad[3072];
    if (g.m < 0) sPad[threadIdx.x] = 1.0;
#endif

    const int lane = threadIdx.x;
    const int tile = g.tile0 + blockIdx.x;
    const ModelCfg cfg = g.cfg;
    int nslots = 0, ncs = 0;
    if constexpr (LDS) {
        nslots = g.tile_slot[tile + 1] - g.tile_slot[tile];
        ncs = g.tile_cslot[tile + 1] - g.tile_cslot[tile];
        for (int i = lane; i < ncs * SC; i += kWave) sC[i] = 0.0;
        if constexpr (TABLE) for (int i = lane; i < nslots * BLK; i += kWave) sTab[i] = 0.0;
    }
    const bool cam_in_lds = LDS && ncs <= kCamLds;
    double *const sCam = sC + ((kCamLds * SC + 1) & ~1);
    static_assert(!LDS || ((kCamLds * SC + 1) & ~1) + kCamLds * kCamLin <= SCN + 1, "camera records fit behind the sums");
    if constexpr (LDS) {
        if (cam_in_lds) {
            for (int i = lane; i < ncs * (kCamLin / 2); i += kWave) {
                const int c = i / (kCamLin / 2), piece = i - c * (kCamLin / 2);
                const int cam = g.cslot_cam[g.tile_cslot[tile] + c];
                reinterpret_cast<double2 *>(sCam)[i] = reinterpret_cast<const double2 *>(g.cam_lin + (size_t)cam * kCamLin)[piece];
            }
        }
    }
    double cost = 0.0, pl2 = 0.0, ebinf = 0.0, maxdv = 0.0, sing = 0.0, ccost = 0.0;
    // register-resident accumulator (LDS tiles): this lane's S block follows the lane for as long as its
    // block does not change
    double acc[LDS && !TABLE ? BLK : 1];
    int cur_slot = -1;
    if constexpr (LDS && !TABLE) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) acc[i] = 0.0;
    }

    // slot -> block and camera-slot -> camera tables of the tile: one coalesced load, one entry per lane
    int my_gblk = 0, my_cam = 0;
    if constexpr (LDS) {
        my_gblk = lane < nslots ? g.slot_gblk[g.tile_slot[tile] + lane] : 0;
        my_cam = lane < ncs ? g.cslot_cam[g.tile_cslot[tile] + lane] : 0;
        if (lane < ncs) sFix[lane] = g.cam_fixed[my_cam];
    }
    // wave-collective: every lane hands its register-resident block to HBM.  kStageLanes blocks at a time go
    // through sStage so that ONE global f64 atomic instruction adds ONE contiguous 7x7 block (the full-rate
    // atomic shape); then the accumulators restart from zero.
    unsigned long long slot_written = 0;      // ordered hand-over: the slots of this tile that hold a block already
    auto flush_blocks = [&]() {
        if constexpr (LDS && !TABLE) {
            const int rr = lane / CNP, cc = lane - rr * CNP;
            for (int base = 0; base < kWave; base += kStageLanes) {
                const int q = lane - base;
                if (q >= 0 && q < kStageLanes && cur_slot >= 0) {
                    double *dst = sStage + q * BLK;
#pragma unroll
                    for (int i = 0; i < BLK; ++i) dst[i] = acc[i];
                }
                wave_lds_sync();
                for (int q2 = 0; q2 < kStageLanes && base + q2 < kWave; ++q2) {
                    const int cL = rl(cur_slot, base + q2);
                    if (cL >= 0) {
                        if (g.tile_blk) {   // ordered hand-over: this tile's own copy of the block, summed over the tiles by k_tile_reduce_blocks
                            double *dst = g.tile_blk + ((size_t)tile * kNSlot + cL) * BLK;
                            if (!((slot_written >> cL) & 1ull)) {                        // (uniform) the slot's first block: a plain store
                                if (lane < BLK) dst[lane] = sStage[q2 * BLK + lane];
                            } else if (lane < BLK) dst[lane] += sStage[q2 * BLK + lane];    // several lanes of a round held it, or a later flush
                            slot_written |= 1ull << cL;
                        } else {
                            const int gb = rl(my_gblk, cL);
                            if (lane < BLK) gadd(&g.S_up[(size_t)gb * (kCP * kCP) + rr * kCP + cc], sStage[q2 * BLK + lane]);
                        }
                    }
                }
                wave_lds_sync();
            }
#pragma unroll
            for (int i = 0; i < BLK; ++i) acc[i] = 0.0;
            cur_slot = -1;
        }
    };

    const int b_begin = g.tile_batch[tile], b_end = g.tile_batch[tile + 1];
    TileDesc td;
    {
        const int bi = min(b_begin + lane, b_end);        // lanes 0..nb hold batches b_begin..b_end
        td.obs = g.batch_obs[bi]; td.pt = g.batch_pt[bi];
        const long long pq = g.batch_pair[bi];
        td.pair_lo = (int)(unsigned)pq; td.pair_hi = (int)(pq >> 32);
    }
    BatchRegs<LDS> cur, nxt;
    double X[3];
    load_batch_level1<LDS>(g, td, 0, lane, cur);
    X[0] = g.pts[3 * (size_t)cur.ip]; X[1] = g.pts[3 * (size_t)cur.ip + 1]; X[2] = g.pts[3 * (size_t)cur.ip + 2];
    nxt = cur;
    if constexpr (LDS) wave_lds_sync();   // the camera records are in place
    STAMP(0); // prologue: LDS clear, first batch loads issued

    unsigned ent_r[LDS ? kMaxRounds : 1];
    long long ent_q0 = -1, ent_q1 = -1;
    for (int b = b_begin; b < b_end; ++b) {
        const bool have_next = b + 1 < b_end;
        // the per-point phases are chains of latency, the pair loop is issue-bound: of the two tiles of a SIMD the one in its per-point
        // phases goes first (measured on C3: build 0.824 -> 0.798 ms; alone a batch's phases 1-4 take 6.4 k cycles, beside the other
        // tile's pair loop 11 k)
        __builtin_amdgcn_s_setprio(3);
        const int nob = cur.nob, p0 = cur.p0, npt = cur.npt;
        const bool active = lane < nob;
        double A[2][CNP], B[2][3], e[2] = { 0.0, 0.0 };
        const int j = cur.j, lp = cur.ip - p0, cs = cur.cs;
        bool head = false, cam_is_fixed = false;
        double pcw = 0.0, pcd[3] = { 0.0, 0.0, 0.0 };
        // ---- phase 1 (camera records are L2-resident, 192 B x m)
        {
            double cl[kCamLin];
            if (cam_in_lds) {
                const double2 *src = reinterpret_cast<const double2 *>(sCam + cs * kCamLin);
#pragma unroll
                for (int k = 0; k < kCamLin / 2; ++k) { const double2 v = src[k]; cl[2 * k] = v.x; cl[2 * k + 1] = v.y; }
            } else load_cam_record(g, j, cl);
            if (have_next) load_batch_level1<LDS>(g, td, b + 1 - b_begin, lane, nxt);   // streams + pair entries of the next batch
            STAMP(7);
            if (active) {
                if constexpr (LDS) cam_is_fixed = sFix[cs] != 0; else cam_is_fixed = g.cam_fixed[j] != 0;
                linearize_obs<CNP, FULL>(cfg, cl, X, cur.xy, cam_is_fixed, e, A, B);
                cost += e[0] * e[0] + e[1] * e[1];
                double *row = sR + lane * SR;
#pragma unroll
                for (int t = 0; t < CNP; ++t) { row[t] = A[0][t]; row[CNP + t] = A[1][t]; }
#pragma unroll
                for (int t = 0; t < 3; ++t) { row[OB + t] = B[0][t]; row[OB + 3 + t] = B[1][t]; }
                row[OB + 6] = e[0]; row[OB + 7] = e[1];
            }
            // point boundaries without touching memory: observations of a point are consecutive lanes
            const int lp_prev = __builtin_amdgcn_update_dpp(-2, active ? lp : -2, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps -2)
            head = active && (lane == 0 || lp != lp_prev);
            if (head) {
                sStart[lp] = lane; pl2 += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
                if (FULL && g.pt_cw) { // soft point constraint (sba_levmar.c:1017-1029): needs X, which the prefetch below replaces
                    pcw = g.pt_cw[cur.ip];
#pragma unroll
                    for (int t = 0; t < 3; ++t) pcd[t] = g.pt_cv[3 * (size_t)cur.ip + t] - X[t];
                }
            }
            if (lane == 0) sStart[npt] = nob;
        }
        // the next batch's points stream from HBM: issue now, consumed in its phase 1
        if (have_next) { X[0] = g.pts[3 * (size_t)nxt.ip]; X[1] = g.pts[3 * (size_t)nxt.ip + 1]; X[2] = g.pts[3 * (size_t)nxt.ip + 2]; }
        wave_lds_sync();
        STAMP(1);
        // ---- phase 2: V_i upper triangle (00 01 02 11 12 22) and eb_i, 7 points per sweep
        for (int base = 0; base < npt; base += 7) {
            const int qd = lane / 9, comp = lane - 9 * qd, lpp = base + qd;
            if (lane < 63 && lpp < npt) {
                const int os = sStart[lpp], oe = sStart[lpp + 1];
                // V(r,c) = sum B0r*B0c + B1r*B1c ; eb(r) = sum B0r*e0 + B1r*e1 : same form, columns (c, 3+c) or (6, 7)
                const int r = comp < 3 ? 0 : (comp < 5 ? 1 : (comp < 6 ? 2 : comp - 6));
                const int c0 = comp < 3 ? comp : (comp < 5 ? comp - 2 : (comp < 6 ? 2 : 6));
                const int c1 = comp < 6 ? c0 + 3 : 7;
                double s0 = 0.0, s1 = 0.0;
                if constexpr (LDS) {
                    // a point of an LDS tile has <= kMaxViewsLds views: all operand loads issued at once instead of a loop of
                    // dependent LDS round trips (the wavefront has nothing else to do here)
                    constexpr int H = kMaxViewsLds / 2;           // two groups of five views: 20 loads in flight (more would spill)
#pragma unroll
                    for (int k0 = 0; k0 < kMaxViewsLds; k0 += H) {
                        double x0[H], x1[H], y0[H], y1[H];
#pragma unroll
                        for (int k = 0; k < H; ++k) {
                            const double *r0 = sR + min(os + k0 + k, oe - 1) * SR + OB;   // (past the point's end: the last row again, the product is dropped)
                            x0[k] = r0[r]; y0[k] = r0[c0]; x1[k] = r0[3 + r]; y1[k] = r0[c1];
                        }
#pragma unroll
                        for (int k = 0; k < H; ++k) {
                            if (os + k0 + k < oe) { if (k & 1) s1 = fma(x1[k], y1[k], fma(x0[k], y0[k], s1)); else s0 = fma(x1[k], y1[k], fma(x0[k], y0[k], s0)); }
                        }
                    }
                } else {
                    int o = os;
                    for (; o + 1 < oe; o += 2) {
                        const double *r0 = sR + o * SR + OB, *r1 = r0 + SR;
                        s0 = fma(r0[3 + r], r0[c1], fma(r0[r], r0[c0], s0));
                        s1 = fma(r1[3 + r], r1[c1], fma(r1[r], r1[c0], s1));
                    }
                    if (o < oe) { const double *r0 = sR + o * SR + OB; s0 = fma(r0[3 + r], r0[c1], fma(r0[r], r0[c0], s0)); }
                }
                sV[lpp * SV + comp] = s0 + s1;
            }
        }
        wave_lds_sync();
        STAMP(2);
        // ---- phase 3: damped inverse per point (role of sba_symat_invert_BK on a 3x3), by the head lane
        if (head) {
            double v00 = sV[lp * SV], v01 = sV[lp * SV + 1], v02 = sV[lp * SV + 2];
            double v11 = sV[lp * SV + 3], v12 = sV[lp * SV + 4], v22 = sV[lp * SV + 5];
            double eb0 = sV[lp * SV + 6], eb1 = sV[lp * SV + 7], eb2 = sV[lp * SV + 8];
            if (FULL && pcw != 0.0) {
                v00 += pcw; v11 += pcw; v22 += pcw;
                eb0 += pcw * pcd[0]; eb1 += pcw * pcd[1]; eb2 += pcw * pcd[2];
                ccost += pcw * (pcd[0] * pcd[0] + pcd[1] * pcd[1] + pcd[2] * pcd[2]);
            }
            maxdv = fmax(maxdv, fmax(v00, fmax(v11, v22)));
            ebinf = fmax(ebinf, fmax(fabs(eb0), fmax(fabs(eb1), fabs(eb2))));
            if (cfg.damp_diag) { v00 = fma(mu_lin, v00, v00); v11 = fma(mu_lin, v11, v11); v22 = fma(mu_lin, v22, v22); }
            else { v00 += mu_lin; v11 += mu_lin; v22 += mu_lin; }
            const double c00 = v11 * v22 - v12 * v12, c01 = v02 * v12 - v01 * v22, c02 = v01 * v12 - v02 * v11;
            const double det = v00 * c00 + v01 * c01 + v02 * c02;
            double i00 = 0, i01 = 0, i02 = 0, i11 = 0, i12 = 0, i22 = 0;
            if (det == 0.0 || !isfinite(det)) { if (!(cfg.points_fixed && cfg.damp_diag)) sing = 1.0; } // "singular matrix V*_i, increasing damping", :1156-1158 (motion only with PBA's damping: V = 0 on purpose)
            else {
                const double id = 1.0 / det;
                i00 = c00 * id; i01 = c01 * id; i02 = c02 * id;
                i11 = (v00 * v22 - v02 * v02) * id; i12 = (v01 * v02 - v00 * v12) * id; i22 = (v00 * v11 - v01 * v01) * id;
            }
            const double h0 = i00 * eb0 + i01 * eb1 + i02 * eb2, h1 = i01 * eb0 + i11 * eb1 + i12 * eb2,
                         h2 = i02 * eb0 + i12 * eb1 + i22 * eb2;
            double *sv = sV + lp * SV;
            sv[0] = i00; sv[1] = i01; sv[2] = i02; sv[3] = i11; sv[4] = i12; sv[5] = i22; sv[6] = h0; sv[7] = h1; sv[8] = h2;
            double *po = g.ptout + (size_t)cur.ip * kPtOut;
            po[0] = i00; po[1] = i01; po[2] = i02; po[3] = i11; po[4] = i12; po[5] = i22;
            po[6] = h0; po[7] = h1; po[8] = h2; po[9] = eb0; po[10] = eb1; po[11] = eb2;
        }
        wave_lds_sync();
        STAMP(3);
        // ---- phase 4: camera-side vectors ea_j, E_j = A^T (e - B h)  (+ diag U_j on request).
        // The pair entries of this batch are requested first; phase 4 and the other wavefront of the SIMD
        // cover their latency.
        int my_rec = 0;
        if constexpr (LDS) {
            if (cur.q0 != ent_q0 || cur.q1 != ent_q1) {   // (uniform) consecutive batches of one camera set share their entries (plan.h: pair templates)
#pragma unroll
                for (int t = 0; t < kMaxRounds; ++t) {
                    const long long q = cur.q0 + (long long)t * kWave + lane;
                    ent_r[t] = (q < cur.q1) ? g.pair[q] : kPairInvalid;
                }
                ent_q0 = cur.q0; ent_q1 = cur.q1;
            }
        } else if (active) my_rec = g.obs_rec[cur.o0 + lane];
        double er[2] = { 0.0, 0.0 };
        if (active && (!LDS || !cam_is_fixed)) {
            const double *sv = sV + lp * SV;
#pragma unroll
            for (int r = 0; r < 2; ++r) er[r] = e[r] - (B[r][0] * sv[6] + B[r][1] * sv[7] + B[r][2] * sv[8]);
            if constexpr (LDS) {
#pragma unroll
                for (int t = 0; t < CNP; ++t) {
                    const double vea = fma(A[1][t], e[1], A[0][t] * e[0]);
                    const double vE = fma(A[1][t], er[1], A[0][t] * er[0]);
                    ladd(&sC[cs * SC + t], vea); ladd(&sC[cs * SC + CNP + t], vE);
                    if (g.want_diag) ladd(&sC[cs * SC + 2 * CNP + t], fma(A[1][t], A[1][t], A[0][t] * A[0][t]));
                }
            }
        }
        STAMP(4);
        // ---- phase 5: pair products (reads sR, sV only: no barrier needed after phase 4)
        __builtin_amdgcn_s_setprio(0);
        if constexpr (LDS) {
            const int nrounds = (int)((cur.q1 - cur.q0) / kWave);
            for (int k = 0; k < nrounds; ++k) {
                unsigned ent = ent_r[0];
#pragma unroll
                for (int t = 1; t < kMaxRounds; ++t) ent = (k == t) ? ent_r[t] : ent;
                const bool valid = ent != kPairInvalid;
                const int la = ent & 63, lb = (ent >> 6) & 63, slot = (ent >> 12) & 63, plp = (ent >> 18) & 31;
                // some lane is about to start another block: everybody flushes (in signature-sorted data all
                // lanes change together, once per few hundred points)
                if constexpr (!TABLE) { if (__any(valid && cur_slot >= 0 && slot != cur_slot)) flush_blocks(); }
                if (valid) {
                    cur_slot = slot;
                    double T0[CNP], T1[CNP];
                    pair_block<CNP, SR>(sR, sV + plp * SV, la, lb, T0, T1);
                    const double2 *Aa2 = reinterpret_cast<const double2 *>(sR + la * SR);
                    double Aa[2 * CNP];
#pragma unroll
                    for (int k2 = 0; k2 < CNP; ++k2) { const double2 v = Aa2[k2]; Aa[2 * k2] = v.x; Aa[2 * k2 + 1] = v.y; }
#pragma unroll
                    for (int r = 0; r < CNP; ++r) {
                        const double a0 = Aa[r], a1 = Aa[CNP + r];
#pragma unroll
                        for (int c = 0; c < CNP; ++c) {
                            if constexpr (TABLE) ladd(&sTab[slot * BLK + r * CNP + c], fma(a1, T1[c], a0 * T0[c]));
                            else acc[r * CNP + c] = fma(a1, T1[c], fma(a0, T0[c], acc[r * CNP + c]));
                        }
                    }
                }
            }
        } else {
            // row tiles: the observation's record goes to HBM (camera-major, plan.h); k_schur_blocks does the rest.
            // A fixed camera's A is zero (linearize_obs), so its record contributes nothing there.
            if (active) {
                const double2 *sv2 = reinterpret_cast<const double2 *>(sV + lp * SV);
                const double2 v01 = sv2[0], v23 = sv2[1], v45 = sv2[2];
                const double i00 = v01.x, i01 = v01.y, i02 = v23.x, i11 = v23.y, i12 = v45.x, i22 = v45.y;
                double G[2][3];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    G[r][0] = fma(B[r][2], i02, fma(B[r][1], i01, B[r][0] * i00));
                    G[r][1] = fma(B[r][2], i12, fma(B[r][1], i11, B[r][0] * i01));
                    G[r][2] = fma(B[r][2], i22, fma(B[r][1], i12, B[r][0] * i02));
                }
                using L = RecL<CNP>;
                double out[L::SIZE];
#pragma unroll
                for (int t = 0; t < L::AP; ++t) { out[L::A0 + t] = t < CNP ? A[0][t < CNP ? t : 0] : 0.0; out[L::A1 + t] = t < CNP ? A[1][t < CNP ? t : 0] : 0.0; }
#pragma unroll
                for (int t = 0; t < 3; ++t) { out[L::B + t] = B[0][t]; out[L::B + 3 + t] = B[1][t]; out[L::G + t] = G[0][t]; out[L::G + 3 + t] = G[1][t]; }
                out[L::E] = e[0]; out[L::E + 1] = e[1]; out[L::ER] = er[0]; out[L::ER + 1] = er[1];
                double2 *dst = reinterpret_cast<double2 *>(g.rec + (size_t)my_rec * L::SIZE);
#pragma unroll
                for (int t = 0; t < L::SIZE / 2; ++t) dst[t] = make_double2(out[2 * t], out[2 * t + 1]);
            }
        }
        wave_lds_sync();
        STAMP(5);
        if (have_next) cur = nxt;
    }
    // ---- flush: S blocks registers -> HBM, camera vectors registers -> LDS slots -> HBM
    if constexpr (LDS) {
        flush_blocks();
        wave_lds_sync();
        if constexpr (TABLE) { // the table goes to HBM once, one contiguous atomic instruction per block
            const int rr = lane / CNP, cc = lane - rr * CNP;
            for (int sl = 0; sl < nslots; ++sl) {
                if (g.tile_blk) { if (lane < BLK) g.tile_blk[((size_t)tile * kNSlot + sl) * BLK + lane] = sTab[sl * BLK + lane]; }
                else {
                    const int gb = rl(my_gblk, sl);
                    if (lane < BLK) gadd(&g.S_up[(size_t)gb * (kCP * kCP) + rr * kCP + cc], sTab[sl * BLK + lane]);
                }
            }
        }
        const int which = lane / CNP, t = lane - which * CNP;
        for (int k = 0; k < ncs; ++k) {
            if (g.tile_blk) { if (lane < 3 * CNP) g.tile_cam[((size_t)tile * kNCSlot + k) * (3 * kCP) + lane] = sC[k * SC + lane]; }
            else {
                const int cam = rl(my_cam, k);
                if (lane < 3 * CNP) gadd(&g.camvec[((size_t)which * g.m + cam) * kCP + t], sC[k * SC + lane]);
            }
        }
    }
    STAMP(6); // flush
#ifdef MIBA_STAMPS
    if (lane == 0 && g.dbg)
        for (int i = 0; i < 8; ++i) atomicAdd(&g.dbg[i], stamp_acc[i]);
#endif
    cost = wave_sum(cost); pl2 = wave_sum(pl2); ccost = wave_sum(ccost);
    ebinf = wave_max(ebinf); maxdv = wave_max(maxdv); sing = wave_max(sing);
    if (lane == 0) { // one partial row per tile; k_reduce_partials folds them (no same-address atomics)
        double *pr = g.part + (size_t)tile * kPartW;
        pr[0] = cost; pr[1] = pl2; pr[2] = ccost; pr[3] = 0.0; pr[4] = ebinf; pr[5] = maxdv; pr[6] = sing; pr[7] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// k_build_mfma: the tile of ONE camera set (plan.h tile mode 3: every point of the tile is seen by the same d cameras,
// kMfmaMinViews <= d <= kMaxViewsLds) with its Schur blocks on the matrix pipe.
//
// With V*^-1 = F F^T (Cholesky factor of the damped inverse, per point) a point's contribution to block (a, b) is
//     Y_ia W_ib^T = W_ia V*^-1 W_ib^T = Z_ia Z_ib^T,      Z_ia = (A_ia^T B_ia) F_i      (cnp x 3),
// i.e. over a batch of points  S_ab -= Zt_a^T Zt_b  with  Zt[k = 3 p + c][a cnp + r] = Z_pa[r][c]:  a GEMM with K = 3 x points whose
// operands are ONE array.  v_mfma_f64_4x4x4_4b_f64 does four independent 4x4x4 products per instruction -- the four 4x4
// quarters of one 8x8-padded block (a, b) --, so a K-chunk of 4 costs one instruction per block, d (d + 1) / 2 <= 55 of them, fed by
// 2 d operand loads (ds_read_b64, one f64 per lane: the left operand of camera a serves all b, the right operand of b all a).
// U_a = sum A^T A goes the same way from At[k = 2 p + e][a cnp + r] = A_pa[e][r] (K = 2 x points, diagonal blocks only).
// Lane maps of the instruction (tools/mfma_f64_layout_probe.hip, measured): A operand lane = 16 k + 4 blk + i, B operand
// lane = 16 k + 4 blk + j, D lane = 16 i + 4 blk + j; blk = 2 ti + tj is the quarter (rows 4 ti.., columns 4 tj..).
// The f64 matrix pipe has the vector pipe's rate (tools/mfma_f64_probe.hip) -- what this buys is issue slots: 1 instruction
// instead of 16, so the wavefront's per-point phases (and the SIMD's other wavefront) overlap with the products instead of
// queueing behind 1 100 v_fma_f64 per batch; the 98 accumulator VGPRs + pair-entry bookkeeping of the vector form are gone,
// and the blocks never leave their registers before the tile ends (one camera set = no hand-over inside the tile).
// ------------------------------------------------------------------------------------------
constexpr int mfma_row_stride(int cnp, int d)   // doubles per K row of Zt / At: the d cameras' rows, room for the padded rows 4 ti + i >= cnp of
{                                               // the last camera, and a stride = 8 mod 16 doubles (four K rows of 8 parameters fall into distinct banks)
    int zs = d * cnp + (8 - cnp);
    while (zs % 16 != 8) ++zs;
    return zs;
}
constexpr int mfma_k_rows(int d) { return ((3 * (kWave / d) + 3) / 4) * 4; }   // K rows of a full batch, padded to whole chunks of 4
constexpr int mfma_big_doubles(int cnp)
{
    int best = kWave * 9;                        // phases 1-2: B (6) | e (2) per observation, stride 9
    for (int d = kMfmaMinViews; d <= kMaxViewsLds; ++d) { const int v = mfma_k_rows(d) * mfma_row_stride(cnp, d); if (v > best) best = v; }
    return best;
}

// the three parts of k_build_mfma whose shape is the number of views D: straight-line code per D (a uniform branch around every
// matrix instruction would cost as much as the instruction), the accumulators indexed statically
constexpr int mfma_q(int a, int b) { return a * kMaxViewsLds - a * (a - 1) / 2 + (b - a); }   // block (a <= b) in the enumeration of kMaxViewsLds cameras
constexpr int kMfmaNQ = kMaxViewsLds * (kMaxViewsLds + 1) / 2;

template <int D, int CNP>
__device__ __forceinline__ void mfma_u_pass(double (&acc)[kMfmaNQ], const double *sBig, int nch, int lane)
{
    constexpr int ZS = mfma_row_stride(CNP, D);
    const double *pl_ = sBig + (lane >> 4) * ZS + 4 * ((lane >> 3) & 1) + (lane & 3), *pr_ = sBig + (lane >> 4) * ZS + 4 * ((lane >> 2) & 1) + (lane & 3);
    for (int ch = 0; ch < nch; ++ch, pl_ += 4 * ZS, pr_ += 4 * ZS) {
#pragma unroll
        for (int a = 0; a < D; ++a)      // (the diagonal block's accumulator holds Z Z^T - A^T A: the right operand negated)
            acc[mfma_q(a, a)] = __builtin_amdgcn_mfma_f64_4x4x4f64(pl_[a * CNP], -pr_[a * CNP], acc[mfma_q(a, a)], 0, 0, 0);
    }
}

template <int D, int CNP>
__device__ __forceinline__ void mfma_z_pass(double (&acc)[kMfmaNQ], const double *sBig, int nch, int lane)
{
    constexpr int ZS = mfma_row_stride(CNP, D);
    const double *pl_ = sBig + (lane >> 4) * ZS + 4 * ((lane >> 3) & 1) + (lane & 3), *pr_ = sBig + (lane >> 4) * ZS + 4 * ((lane >> 2) & 1) + (lane & 3);
    for (int ch = 0; ch < nch; ++ch, pl_ += 4 * ZS, pr_ += 4 * ZS) {
        double zl[D], zr[D];
#pragma unroll
        for (int a = 0; a < D; ++a) { zl[a] = pl_[a * CNP]; zr[a] = pr_[a * CNP]; }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = a; b < D; ++b) acc[mfma_q(a, b)] = __builtin_amdgcn_mfma_f64_4x4x4f64(zl[a], zr[b], acc[mfma_q(a, b)], 0, 0, 0);
    }
}

// the tile's blocks: registers -> HBM.  D lane = 16 i + 4 blk + j: row 4 ti + i, column 4 tj + j of the block
template <int D, int CNP>
__device__ __forceinline__ void mfma_flush(const double (&acc)[kMfmaNQ], const BuildArgs &g, int tile, int my_gblk, int lane)
{
    constexpr int BLK = CNP * CNP;
    const int row = 4 * ((lane >> 3) & 1) + (lane >> 4), col = 4 * ((lane >> 2) & 1) + (lane & 3);
    const bool inside = row < CNP && col < CNP;
    int slot = 0;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a; b < D; ++b, ++slot) {
            const double val = -acc[mfma_q(a, b)];
            if (g.tile_blk) { if (inside) g.tile_blk[((size_t)tile * kNSlot + slot) * BLK + row * CNP + col] = val; }
            else {
                const int gb = rl(my_gblk, slot);
                if (inside) gadd(&g.S_up[(size_t)gb * (kCP * kCP) + row * kCP + col], val);
            }
        }
}

#define MFMA_FOR_D(d, CALL)                                                                     \
    switch (d) {                                                                                \
    case 4: { constexpr int D = 4; CALL; } break;  case 5: { constexpr int D = 5; CALL; } break;      \
    case 6: { constexpr int D = 6; CALL; } break;  case 7: { constexpr int D = 7; CALL; } break;      \
    case 8: { constexpr int D = 8; CALL; } break;  case 9: { constexpr int D = 9; CALL; } break;      \
    default: { constexpr int D = 10; CALL; } break;                                             \
    }
static_assert(kMfmaMinViews == 4 && kMaxViewsLds == 10, "MFMA_FOR_D lists the view counts");

template <int CNP, bool FULL>
__global__ __launch_bounds__(64, CNP <= 7 ? 2 : 1) void k_build_mfma(const BuildArgs g)
{
    if (g.tile_mode[g.tile0 + blockIdx.x] != 3) return;
    if (g.spec && g.spec[0] == 0.0) return;
    const double mu_lin = g.spec ? g.spec[1] : g.mu;
#ifdef MIBA_STAMPS
    unsigned long long stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    constexpr int SB = 9, SV = 10, SC = (3 * CNP) | 1;
    constexpr int kMaxD = kMaxViewsLds, kMaxPts = kWave / kMfmaMinViews;
    constexpr int NQ = kMfmaNQ;
    __shared__ __attribute__((aligned(16))) double sBig[mfma_big_doubles(CNP)];   // phases 1-2: B | e rows; then At, then Zt
    __shared__ __attribute__((aligned(16))) double sV[kMaxPts * SV];              // per point: h (3) at 6.., F (6) at 0..
    __shared__ __attribute__((aligned(16))) double sC[kMaxD * SC];
    __shared__ __attribute__((aligned(16))) double sCam[kMaxD * kCamLin];
    __shared__ int sStart[kMaxPts + 1];
    __shared__ int sFix[kMaxD];

    const int lane = threadIdx.x;
    const int tile = g.tile0 + blockIdx.x;
    const ModelCfg cfg = g.cfg;
    const int d = g.tile_cslot[tile + 1] - g.tile_cslot[tile];           // views per point = cameras of the tile
    const int nslots = g.tile_slot[tile + 1] - g.tile_slot[tile];        // = d (d + 1) / 2, slot of (a, b) = its rank in the a <= b enumeration
    int ZS = 0;
#pragma unroll
    for (int dd = kMfmaMinViews; dd <= kMaxD; ++dd) if (d == dd) ZS = mfma_row_stride(CNP, dd);
    for (int i = lane; i < d * SC; i += kWave) sC[i] = 0.0;
    for (int i = lane; i < d * (kCamLin / 2); i += kWave) {
        const int c = i / (kCamLin / 2), piece = i - c * (kCamLin / 2);
        const int cam = g.cslot_cam[g.tile_cslot[tile] + c];
        reinterpret_cast<double2 *>(sCam)[i] = reinterpret_cast<const double2 *>(g.cam_lin + (size_t)cam * kCamLin)[piece];
    }
    const int my_gblk = lane < nslots ? g.slot_gblk[g.tile_slot[tile] + lane] : 0;
    const int my_cam = lane < d ? g.cslot_cam[g.tile_cslot[tile] + lane] : 0;
    if (lane < d) sFix[lane] = g.cam_fixed[my_cam];
    double cost = 0.0, pl2 = 0.0, ebinf = 0.0, maxdv = 0.0, sing = 0.0, ccost = 0.0;
    double acc[NQ];                          // sum Z_a Z_b^T (- U_a on the diagonal) of the blocks a <= b, in the enumeration of kMaxD cameras
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc[i] = 0.0;

    const int b_begin = g.tile_batch[tile], b_end = g.tile_batch[tile + 1];
    TileDesc td;
    {
        const int bi = min(b_begin + lane, b_end);
        td.obs = g.batch_obs[bi]; td.pt = g.batch_pt[bi]; td.pair_lo = 0; td.pair_hi = 0;
    }
    BatchRegs<true> cur, nxt;
    double X[3];
    load_batch_level1<true>(g, td, 0, lane, cur);
    X[0] = g.pts[3 * (size_t)cur.ip]; X[1] = g.pts[3 * (size_t)cur.ip + 1]; X[2] = g.pts[3 * (size_t)cur.ip + 2];
    nxt = cur;
    wave_lds_sync();
    STAMP(0);

    for (int b = b_begin; b < b_end; ++b) {
        const bool have_next = b + 1 < b_end;
        const int nob = cur.nob, p0 = cur.p0, npt = cur.npt;
        const bool active = lane < nob;
        double A[2][CNP], B[2][3], e[2] = { 0.0, 0.0 };
        const int lp = cur.ip - p0, cs = cur.cs;
        bool head = false, cam_is_fixed = false;
        double pcw = 0.0, pcd[3] = { 0.0, 0.0, 0.0 };
        // ---- phase 1: residual and Jacobian of the lane's observation
        {
            double cl[kCamLin];
            const double2 *src = reinterpret_cast<const double2 *>(sCam + cs * kCamLin);
#pragma unroll
            for (int k = 0; k < kCamLin / 2; ++k) { const double2 v = src[k]; cl[2 * k] = v.x; cl[2 * k + 1] = v.y; }
            if (have_next) load_batch_level1<true>(g, td, b + 1 - b_begin, lane, nxt);
            if (active) {
                cam_is_fixed = sFix[cs] != 0;
                linearize_obs<CNP, FULL>(cfg, cl, X, cur.xy, cam_is_fixed, e, A, B);
                cost += e[0] * e[0] + e[1] * e[1];
                double *row = sBig + lane * SB;
#pragma unroll
                for (int t = 0; t < 3; ++t) { row[t] = B[0][t]; row[3 + t] = B[1][t]; }
                row[6] = e[0]; row[7] = e[1];
            } else {
#pragma unroll
                for (int t = 0; t < CNP; ++t) { A[0][t] = 0.0; A[1][t] = 0.0; }
#pragma unroll
                for (int t = 0; t < 3; ++t) { B[0][t] = 0.0; B[1][t] = 0.0; }
            }
            const int lp_prev = __builtin_amdgcn_update_dpp(-2, active ? lp : -2, 0x138, 0xF, 0xF, false);
            head = active && (lane == 0 || lp != lp_prev);
            if (head) {
                sStart[lp] = lane; pl2 += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
                if (FULL && g.pt_cw) {
                    pcw = g.pt_cw[cur.ip];
#pragma unroll
                    for (int t = 0; t < 3; ++t) pcd[t] = g.pt_cv[3 * (size_t)cur.ip + t] - X[t];
                }
            }
            if (lane == 0) sStart[npt] = nob;
        }
        if (have_next) { X[0] = g.pts[3 * (size_t)nxt.ip]; X[1] = g.pts[3 * (size_t)nxt.ip + 1]; X[2] = g.pts[3 * (size_t)nxt.ip + 2]; }
        wave_lds_sync();
        STAMP(1);
        // ---- phase 2: V_i (00 01 02 11 12 22) and eb_i, 7 points per sweep
        for (int base = 0; base < npt; base += 7) {
            const int qd = lane / 9, comp = lane - 9 * qd, lpp = base + qd;
            if (lane < 63 && lpp < npt) {
                const int os = sStart[lpp], oe = sStart[lpp + 1];
                const int r = comp < 3 ? 0 : (comp < 5 ? 1 : (comp < 6 ? 2 : comp - 6));
                const int c0 = comp < 3 ? comp : (comp < 5 ? comp - 2 : (comp < 6 ? 2 : 6));
                const int c1 = comp < 6 ? c0 + 3 : 7;
                double s0 = 0.0, s1 = 0.0;
                int o = os;
                for (; o + 1 < oe; o += 2) {
                    const double *r0 = sBig + o * SB, *r1 = r0 + SB;
                    s0 = fma(r0[3 + r], r0[c1], fma(r0[r], r0[c0], s0));
                    s1 = fma(r1[3 + r], r1[c1], fma(r1[r], r1[c0], s1));
                }
                if (o < oe) { const double *r0 = sBig + o * SB; s0 = fma(r0[3 + r], r0[c1], fma(r0[r], r0[c0], s0)); }
                sV[lpp * SV + comp] = s0 + s1;
            }
        }
        wave_lds_sync();
        STAMP(2);
        // ---- phase 3: damped inverse per point, h, and the Cholesky factor F of the inverse (V*^-1 = F F^T, F lower triangular)
        if (head) {
            double v00 = sV[lp * SV], v01 = sV[lp * SV + 1], v02 = sV[lp * SV + 2];
            double v11 = sV[lp * SV + 3], v12 = sV[lp * SV + 4], v22 = sV[lp * SV + 5];
            double eb0 = sV[lp * SV + 6], eb1 = sV[lp * SV + 7], eb2 = sV[lp * SV + 8];
            if (FULL && pcw != 0.0) {
                v00 += pcw; v11 += pcw; v22 += pcw;
                eb0 += pcw * pcd[0]; eb1 += pcw * pcd[1]; eb2 += pcw * pcd[2];
                ccost += pcw * (pcd[0] * pcd[0] + pcd[1] * pcd[1] + pcd[2] * pcd[2]);
            }
            maxdv = fmax(maxdv, fmax(v00, fmax(v11, v22)));
            ebinf = fmax(ebinf, fmax(fabs(eb0), fmax(fabs(eb1), fabs(eb2))));
            if (cfg.damp_diag) { v00 = fma(mu_lin, v00, v00); v11 = fma(mu_lin, v11, v11); v22 = fma(mu_lin, v22, v22); }
            else { v00 += mu_lin; v11 += mu_lin; v22 += mu_lin; }
            const double c00 = v11 * v22 - v12 * v12, c01 = v02 * v12 - v01 * v22, c02 = v01 * v12 - v02 * v11;
            const double det = v00 * c00 + v01 * c01 + v02 * c02;
            double i00 = 0, i01 = 0, i02 = 0, i11 = 0, i12 = 0, i22 = 0;
            const bool zero_on_purpose = cfg.points_fixed && cfg.damp_diag;
            if (det == 0.0 || !isfinite(det)) { if (!zero_on_purpose) sing = 1.0; }
            else {
                const double id = 1.0 / det;
                i00 = c00 * id; i01 = c01 * id; i02 = c02 * id;
                i11 = (v00 * v22 - v02 * v02) * id; i12 = (v01 * v02 - v00 * v12) * id; i22 = (v00 * v11 - v01 * v01) * id;
            }
            const double h0 = i00 * eb0 + i01 * eb1 + i02 * eb2, h1 = i01 * eb0 + i11 * eb1 + i12 * eb2,
                         h2 = i02 * eb0 + i12 * eb1 + i22 * eb2;
            // F: a pivot that is not positive means V* is not positive definite to working precision (V = sum B^T B is, and mu > 0):
            // reported like a singular V*_i -- "increasing damping", sba_levmar.c:1156-1158 -- unless the inverse is zero on purpose
            double f00 = 0, f10 = 0, f20 = 0, f11 = 0, f21 = 0, f22 = 0;
            if (i00 > 0.0) {
                f00 = sqrt(i00);
                const double r0 = 1.0 / f00;
                f10 = i01 * r0; f20 = i02 * r0;
                const double t11 = i11 - f10 * f10;
                if (t11 > 0.0) {
                    f11 = sqrt(t11);
                    f21 = (i12 - f20 * f10) / f11;
                    const double t22 = i22 - f20 * f20 - f21 * f21;
                    if (t22 > 0.0) f22 = sqrt(t22); else sing = 1.0;
                } else sing = 1.0;
            } else if (!(i00 == 0.0 && i11 == 0.0 && i22 == 0.0)) sing = 1.0;
            double *sv = sV + lp * SV;
            sv[0] = f00; sv[1] = f10; sv[2] = f20; sv[3] = f11; sv[4] = f21; sv[5] = f22; sv[6] = h0; sv[7] = h1; sv[8] = h2;
            double *po = g.ptout + (size_t)cur.ip * kPtOut;
            po[0] = i00; po[1] = i01; po[2] = i02; po[3] = i11; po[4] = i12; po[5] = i22;
            po[6] = h0; po[7] = h1; po[8] = h2; po[9] = eb0; po[10] = eb1; po[11] = eb2;
        }
        wave_lds_sync();
        STAMP(3);
        // ---- phase 4: camera-side vectors; At = the observation's A, transposed into K rows (the B | e rows are dead by now)
        double Z[CNP][3];
        {
            const double *sv = sV + lp * SV;
            const double2 f01 = reinterpret_cast<const double2 *>(sv)[0], f23 = reinterpret_cast<const double2 *>(sv)[1], f45 = reinterpret_cast<const double2 *>(sv)[2];
            double er[2] = { 0.0, 0.0 };
            if (active && !cam_is_fixed) {
#pragma unroll
                for (int r = 0; r < 2; ++r) er[r] = e[r] - (B[r][0] * sv[6] + B[r][1] * sv[7] + B[r][2] * sv[8]);
#pragma unroll
                for (int t = 0; t < CNP; ++t) {
                    const double vea = fma(A[1][t], e[1], A[0][t] * e[0]);
                    const double vE = fma(A[1][t], er[1], A[0][t] * er[0]);
                    ladd(&sC[cs * SC + t], vea); ladd(&sC[cs * SC + CNP + t], vE);
                    if (g.want_diag) ladd(&sC[cs * SC + 2 * CNP + t], fma(A[1][t], A[1][t], A[0][t] * A[0][t]));
                }
            }
            // Z = (A^T B) F
#pragma unroll
            for (int r = 0; r < CNP; ++r) {
                const double w0 = fma(A[1][r], B[1][0], A[0][r] * B[0][0]), w1 = fma(A[1][r], B[1][1], A[0][r] * B[0][1]),
                             w2 = fma(A[1][r], B[1][2], A[0][r] * B[0][2]);
                Z[r][0] = fma(w2, f23.x, fma(w1, f01.y, w0 * f01.x));
                Z[r][1] = fma(w2, f45.x, w1 * f23.y);
                Z[r][2] = w2 * f45.y;
            }
            if (active) {
                double *dst = sBig + (2 * lp) * ZS + cs * CNP;
#pragma unroll
                for (int t = 0; t < CNP; ++t) { dst[t] = A[0][t]; dst[ZS + t] = A[1][t]; }
            }
            // K rows beyond this batch's points, up to whole chunks, are zero (at most 3 rows)
            for (int i = 2 * npt * ZS + lane; i < ((2 * npt + 3) & ~3) * ZS; i += kWave) sBig[i] = 0.0;
        }
        wave_lds_sync();
        STAMP(4);
        // ---- phase 5a: U_a = sum At_a^T At_a on the matrix pipe, into the diagonal blocks' accumulators
        MFMA_FOR_D(d, (mfma_u_pass<D, CNP>(acc, sBig, (2 * npt + 3) >> 2, lane)));
        wave_lds_sync();
        STAMP(5);
        // ---- Zt over At
        if (active) {
            double *dst = sBig + (3 * lp) * ZS + cs * CNP;
#pragma unroll
            for (int t = 0; t < CNP; ++t) { dst[t] = Z[t][0]; dst[ZS + t] = Z[t][1]; dst[2 * ZS + t] = Z[t][2]; }
        }
        for (int i = 3 * npt * ZS + lane; i < ((3 * npt + 3) & ~3) * ZS; i += kWave) sBig[i] = 0.0;
        wave_lds_sync();
        STAMP(7);
        // ---- phase 5b: S_ab -= Zt_a^T Zt_b: per K-chunk of 4 the 2 d operands, then one instruction per block
        MFMA_FOR_D(d, (mfma_z_pass<D, CNP>(acc, sBig, (3 * npt + 3) >> 2, lane)));
        wave_lds_sync();
        STAMP(5);
        if (have_next) cur = nxt;
    }
    {
        MFMA_FOR_D(d, (mfma_flush<D, CNP>(acc, g, tile, my_gblk, lane)));
        wave_lds_sync();
        const int which = lane / CNP, t = lane - which * CNP;
        for (int k = 0; k < d; ++k) {
            if (g.tile_blk) { if (lane < 3 * CNP) g.tile_cam[((size_t)tile * kNCSlot + k) * (3 * kCP) + lane] = sC[k * SC + lane]; }
            else {
                const int cam = rl(my_cam, k);
                if (lane < 3 * CNP) gadd(&g.camvec[((size_t)which * g.m + cam) * kCP + t], sC[k * SC + lane]);
            }
        }
    }
    STAMP(6);
#ifdef MIBA_STAMPS
    if (lane == 0 && g.dbg)
        for (int i = 0; i < 8; ++i) atomicAdd(&g.dbg[i], stamp_acc[i]);
#endif
    cost = wave_sum(cost); pl2 = wave_sum(pl2); ccost = wave_sum(ccost);
    ebinf = wave_max(ebinf); maxdv = wave_max(maxdv); sing = wave_max(sing);
    if (lane == 0) {
        double *pr = g.part + (size_t)tile * kPartW;
        pr[0] = cost; pr[1] = pl2; pr[2] = ccost; pr[3] = 0.0; pr[4] = ebinf; pr[5] = maxdv; pr[6] = sing; pr[7] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// k_build_uni: the tile of ONE camera set (plan.h tile mode 4: every point of the tile is seen by the same D cameras,
// kUniMinViews <= D <= kMaxViewsLds) on the vector pipe -- what k_build does for any tile, with everything the uniform shape makes
// constant taken out of the loops:
//   * lane = point * D + camera in every batch: no per-observation camera / point / slot indices are loaded, the batch
//     descriptors are arithmetic on the tile's first point, the lane's camera record and fixed flag are per-tile constants;
//   * the lane's pair (a <= b) and block never change: no pair entries, no decode, no hand-over inside the tile; a round's
//     operand rows are lane constants + one uniform offset (2 vector adds per round instead of ~45 instructions of
//     select / decode / address arithmetic); several points per round while D (D + 1) / 2 <= 32;
//   * G = B V*^-1 per OBSERVATION (phase 4, written over the dead e of the row) instead of per pair: the pair core is
//     M = delta_ab I - G_a B_b^T (12 FMA, 6 + 6 doubles from LDS) -- 138 instead of 156 FMA and 40 instead of 46 doubles per pair
//     [the per-pair product that sba_levmar.c:1195-1268 forms as Y_ij = W_ij V*^-1 once per observation as well];
//   * phase 3 runs on every lane of the point (same instructions as on the head lane alone) straight into phase 4: V*^-1 and h
//     stay in registers, one LDS round trip and one wave synchronisation less per batch.
// The blocks leave their registers once, when the tile ends (staged through the rows' LDS, dead by then).
// ------------------------------------------------------------------------------------------
// DC: the tile's view count D as a compile-time constant (the instantiation for ten views: every tile of C2 / C3), or 0 = whatever the tile has
// (a launch of each walks the tile range; a workgroup whose tile belongs to the other leaves at once)
template <int CNP, bool FULL, int DC>
__global__ __launch_bounds__(64, CNP <= 7 ? 2 : 1) void k_build_uni(const BuildArgs g)
{
    if (g.tile_mode[g.tile0 + blockIdx.x] != 4) return;
    {
        const int t_ = g.tile0 + blockIdx.x, d_ = g.tile_cslot[t_ + 1] - g.tile_cslot[t_];
        if (DC > 0 ? d_ != DC : (g.has_uni_fixed && d_ == kUniFixedViews)) return;
    }
    if (g.spec && g.spec[0] == 0.0) return;
    const double mu_lin = g.spec ? g.spec[1] : g.mu;
#ifdef MIBA_STAMPS
    unsigned long long stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    // one 16-byte aligned LDS row per observation: A (2 x CNP) | B (6) | e (2), then G (6) over e.  Row stride SR doubles with SR / 2
    // odd: 16 distinct rows fall into 16 distinct 16-byte bank slots (conflict-free ds_read_b128 in the pair phase)
    constexpr int OB = 2 * CNP, OG = OB + 6;
    constexpr int SR = (((OB + 12) / 2) % 2 == 1) ? OB + 12 : OB + 14;
    constexpr int SV = 10, BLK = CNP * CNP, SC = (3 * CNP) | 1;
    constexpr int kMaxD = kMaxViewsLds, kMaxPts = kWave / kUniMinViews;
    constexpr int kStageBlocks = (kWave * SR) / BLK;         // blocks the rows' LDS holds when the tile hands over
    __shared__ __attribute__((aligned(16))) double sR[kWave * SR];
    __shared__ __attribute__((aligned(16))) double sV[kMaxPts * SV];
    __shared__ __attribute__((aligned(16))) double sC[kMaxD * SC];
    // the cameras' records, 26 doubles apart: at the records' own 24 (192 B = 48 banks) cameras c and c + 4 start in the same bank and every
    // ds_read_b128 of phase 1 was a 2-3-way conflict; 52 dwords apart sixteen records start in sixteen different 16-byte slots
    constexpr int SCAM = kCamLin + 2;
    __shared__ __attribute__((aligned(16))) double sCam[kMaxD * SCAM];
    // the next batch's measurement and point streams land here by LDS-DMA (global_load_lds: no destination registers held through the pair phase)
    __shared__ __attribute__((aligned(16))) double2 sXY[kWave];
    __shared__ __attribute__((aligned(16))) double sX[kMaxPts * 3];

    const int lane = threadIdx.x;
    const int tile = g.tile0 + blockIdx.x;
    const ModelCfg cfg = g.cfg;
    const int D = DC > 0 ? DC : g.tile_cslot[tile + 1] - g.tile_cslot[tile];   // views per point = cameras of the tile
    const int NB = DC > 0 ? DC * (DC + 1) / 2 : g.tile_slot[tile + 1] - g.tile_slot[tile];   // = D (D + 1) / 2, slot of (a, b) = its rank in the a <= b enumeration
    const int PB = kWave / D, PPR = kWave / NB;                            // points per batch, points per pair round
    for (int i = lane; i < D * SC; i += kWave) sC[i] = 0.0;
    for (int i = lane; i < D * (kCamLin / 2); i += kWave) {
        const int c = i / (kCamLin / 2), piece = i - c * (kCamLin / 2);
        const int cam = g.cslot_cam[g.tile_cslot[tile] + c];
        reinterpret_cast<double2 *>(sCam + c * SCAM)[piece] = reinterpret_cast<const double2 *>(g.cam_lin + (size_t)cam * kCamLin)[piece];
    }
    const int my_gblk = lane < NB ? g.slot_gblk[g.tile_slot[tile] + lane] : 0;
    const int my_cam = lane < D ? g.cslot_cam[g.tile_cslot[tile] + lane] : 0;
    // the lane as an observation: point lp of the batch, camera slot cs
    const int lp = lane / D, cs = lane - lp * D;
    const bool cam_is_fixed = g.cam_fixed[g.cslot_cam[g.tile_cslot[tile] + cs]] != 0;
    // the lane as a pair: sub-point ps of the round, block pq = (pa <= pb)
    const int ps = lane / NB, pq = lane - ps * NB;
    int pa = 0, pb = pq;
    while (pb >= D - pa) { pb -= D - pa; ++pa; }
    pb += pa;
    const bool pair_lane = ps < PPR;
    const double dg = (pa == pb) ? 1.0 : 0.0;
    double cost = 0.0, pl2 = 0.0, ebinf = 0.0, maxdv = 0.0, sing = 0.0, ccost = 0.0;
    double acc[BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i) acc[i] = 0.0;

    const int b_begin = g.tile_batch[tile], b_end = g.tile_batch[tile + 1];
    const int tp0 = g.batch_pt[b_begin], tp1 = g.batch_pt[b_end], to0 = g.batch_obs[b_begin];
    const int nbatch = b_end - b_begin;                                    // = ceil((tp1 - tp0) / PB): whole batches but the last (plan.cpp)
    // The batch's streams: its measurements (16 B per observation, lane-linear) and its points (24 B per point, consecutive in the
    // internal order: one contiguous chunk, a dword per lane) go from HBM straight into LDS, one batch ahead.
    auto stream_batch = [&](int kk) {
        int lv = lane; asm volatile("" : "+v"(lv));       // (opaque: the address arithmetic is redone at every call, not kept in registers through the loop)
        const int p1 = tp0 + kk * PB, np1 = min(PB, tp1 - p1);
        const unsigned *src = reinterpret_cast<const unsigned *>(g.pts + 3 * (size_t)p1);
        if (lv < np1 * D) glds_b128(g.obs_xy + (to0 + kk * PB * D + lv), sXY);
        if (lv < 6 * np1) glds_b32(src + lv, sX);
        if (lv + kWave < 6 * np1) glds_b32(src + kWave + lv, reinterpret_cast<unsigned *>(sX) + kWave);
    };
    double ea[CNP], Ev[CNP];                                // this lane's share of its camera's e_a and E, summed over the tile's batches
#pragma unroll
    for (int t = 0; t < CNP; ++t) { ea[t] = 0.0; Ev[t] = 0.0; }
    stream_batch(0);
    wave_lds_sync();   // the camera records are in place
    STAMP(0);

    for (int k = 0; k < nbatch; ++k) {
        __builtin_amdgcn_s_setprio(3);
        const int p0 = tp0 + k * PB, npt = min(PB, tp1 - p0);
        const bool active = lp < npt;
        const int ip = p0 + lp;
        const bool head = active && cs == 0;
        double pcw = 0.0, pcd[3] = { 0.0, 0.0, 0.0 };
        double *const row = sR + lane * SR;
        // ---- phase 1: residual and Jacobian of the lane's observation
        wait_lds_dma();                                    // the batch's streams have landed (issued a pair phase ago)
        const double2 xy = active ? sXY[lane] : make_double2(0.0, 0.0);
        double X[3] = { 0.0, 0.0, 0.0 };
        if (active) { X[0] = sX[3 * lp]; X[1] = sX[3 * lp + 1]; X[2] = sX[3 * lp + 2]; }
        {
            const double *cl = sCam + cs * SCAM;    // the camera's record is read where it is used: no 48 registers held through the phase
            if (active) {
                double A[2][CNP], B[2][3], e[2];           // (dead once they are in the row: phase 4 reads them back, the registers are the accumulators')
                linearize_obs<CNP, FULL>(cfg, cl, X, xy, cam_is_fixed, e, A, B);
                cost += e[0] * e[0] + e[1] * e[1];
#pragma unroll
                for (int t = 0; t < CNP; ++t) { row[t] = A[0][t]; row[CNP + t] = A[1][t]; }
#pragma unroll
                for (int t = 0; t < 3; ++t) { row[OB + t] = B[0][t]; row[OB + 3 + t] = B[1][t]; }
                row[OG] = e[0]; row[OG + 1] = e[1];
            }
            if (head) {
                pl2 += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
                if (FULL && g.pt_cw) { // soft point constraint (sba_levmar.c:1017-1029): needs X, which the prefetch below replaces
                    pcw = g.pt_cw[ip];
#pragma unroll
                    for (int t = 0; t < 3; ++t) pcd[t] = g.pt_cv[3 * (size_t)ip + t] - X[t];
                }
            }
        }
        wave_lds_sync();
        STAMP(1);
        // ---- phase 2: V_i upper triangle (00 01 02 11 12 22) and eb_i, lane = (point, component), 7 points per sweep; a point's
        // rows are lanes lpp D .. lpp D + D - 1.  (the sums in k_build's order: the same bits)
        for (int base = 0; base < npt; base += 7) {
            int lv = lane; asm volatile("" : "+v"(lv));       // (opaque: see the prefetch below)
            const int qd = lv / 9, comp = lv - 9 * qd, lpp = base + qd;
            if (lv < 63 && lpp < npt) {
                // V(r,c) = sum B0r*B0c + B1r*B1c ; eb(r) = sum B0r*e0 + B1r*e1 : same form, columns (c, 3+c) or (6, 7) of the B | e part.
                // The four operands of view t sit at compile-time offsets t SR from the point's first row: every load is an immediate-
                // offset ds_read, all kMaxViewsLds views issued at once (past the point's D views: rows of the next point or LDS beyond the
                // rows -- read and dropped)
                const int r = comp < 3 ? 0 : (comp < 5 ? 1 : (comp < 6 ? 2 : comp - 6));
                const int c0 = comp < 3 ? comp : (comp < 5 ? comp - 2 : (comp < 6 ? 2 : 6));
                const int c1 = comp < 6 ? c0 + 3 : 7;
                const double *rb0 = sR + lpp * D * SR + OB;
                const double *px0 = rb0 + r, *py0 = rb0 + c0, *px1 = rb0 + 3 + r, *py1 = rb0 + c1;
                double s0 = 0.0, s1 = 0.0;
                constexpr int H = kMaxViewsLds / 2;               // two groups of five views: 20 loads in flight (all forty would spill)
#pragma unroll
                for (int k0 = 0; k0 < kMaxViewsLds; k0 += H) {
                    double x0[H], x1[H], y0[H], y1[H];
#pragma unroll
                    for (int t = 0; t < H; ++t) { x0[t] = px0[(k0 + t) * SR]; y0[t] = py0[(k0 + t) * SR]; x1[t] = px1[(k0 + t) * SR]; y1[t] = py1[(k0 + t) * SR]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < H; ++t) {                 // (uniform conditions: D is the tile's)
                        if (k0 + t < D) { if ((k0 + t) & 1) s1 = fma(x1[t], y1[t], fma(x0[t], y0[t], s1)); else s0 = fma(x1[t], y1[t], fma(x0[t], y0[t], s0)); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                sV[lpp * SV + comp] = s0 + s1;
            }
        }
        wave_lds_sync();
        STAMP(2);
        // ---- phases 3 + 4 on every lane of the point: damped inverse (role of sba_symat_invert_BK on a 3x3), h, then the observation's
        // G = B V*^-1 into its row and the camera-side vectors e_a, E = A^T (e - B h) (+ diag U on request)
        if (active) {
            const double2 *sv2 = reinterpret_cast<const double2 *>(sV + lp * SV);
            const double2 q01 = sv2[0], q23 = sv2[1], q45 = sv2[2], q67 = sv2[3];
            double v00 = q01.x, v01 = q01.y, v02 = q23.x, v11 = q23.y, v12 = q45.x, v22 = q45.y;
            double eb0 = q67.x, eb1 = q67.y, eb2 = sV[lp * SV + 8];
            if (FULL && g.pt_cw) {          // (pcw, pcd live on the head lane: handed to the point's other lanes)
                const int hl = lp * D;
                pcw = __shfl(pcw, hl);
#pragma unroll
                for (int t = 0; t < 3; ++t) pcd[t] = __shfl(pcd[t], hl);
                if (pcw != 0.0) {
                    v00 += pcw; v11 += pcw; v22 += pcw;
                    eb0 += pcw * pcd[0]; eb1 += pcw * pcd[1]; eb2 += pcw * pcd[2];
                    if (head) ccost += pcw * (pcd[0] * pcd[0] + pcd[1] * pcd[1] + pcd[2] * pcd[2]);
                }
            }
            maxdv = fmax(maxdv, fmax(v00, fmax(v11, v22)));
            ebinf = fmax(ebinf, fmax(fabs(eb0), fmax(fabs(eb1), fabs(eb2))));
            if (cfg.damp_diag) { v00 = fma(mu_lin, v00, v00); v11 = fma(mu_lin, v11, v11); v22 = fma(mu_lin, v22, v22); }
            else { v00 += mu_lin; v11 += mu_lin; v22 += mu_lin; }
            const double c00 = v11 * v22 - v12 * v12, c01 = v02 * v12 - v01 * v22, c02 = v01 * v12 - v02 * v11;
            const double det = v00 * c00 + v01 * c01 + v02 * c02;
            double i00 = 0, i01 = 0, i02 = 0, i11 = 0, i12 = 0, i22 = 0;
            if (det == 0.0 || !isfinite(det)) { if (!(cfg.points_fixed && cfg.damp_diag)) sing = 1.0; } // "singular matrix V*_i, increasing damping", :1156-1158
            else {
                const double id = fast_rcp(det);
                i00 = c00 * id; i01 = c01 * id; i02 = c02 * id;
                i11 = (v00 * v22 - v02 * v02) * id; i12 = (v01 * v02 - v00 * v12) * id; i22 = (v00 * v11 - v01 * v01) * id;
            }
            const double h0 = i00 * eb0 + i01 * eb1 + i02 * eb2, h1 = i01 * eb0 + i11 * eb1 + i12 * eb2,
                         h2 = i02 * eb0 + i12 * eb1 + i22 * eb2;
            // the observation's row again: B | e, and A further down
            double B[2][3], e[2];
            {
                const double2 *rb2 = reinterpret_cast<const double2 *>(row + OB);
                const double2 b01 = rb2[0], b23 = rb2[1], b45 = rb2[2], e01 = rb2[3];
                B[0][0] = b01.x; B[0][1] = b01.y; B[0][2] = b23.x; B[1][0] = b23.y; B[1][1] = b45.x; B[1][2] = b45.y; e[0] = e01.x; e[1] = e01.y;
            }
            if (head) {
                double *po = g.ptout + (size_t)ip * kPtOut;
                po[0] = i00; po[1] = i01; po[2] = i02; po[3] = i11; po[4] = i12; po[5] = i22;
                po[6] = h0; po[7] = h1; po[8] = h2; po[9] = eb0; po[10] = eb1; po[11] = eb2;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                row[OG + 3 * r] = fma(B[r][2], i02, fma(B[r][1], i01, B[r][0] * i00));
                row[OG + 3 * r + 1] = fma(B[r][2], i12, fma(B[r][1], i11, B[r][0] * i01));
                row[OG + 3 * r + 2] = fma(B[r][2], i22, fma(B[r][1], i12, B[r][0] * i02));
            }
            if (!cam_is_fixed) {
                double er[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) er[r] = e[r] - (B[r][0] * h0 + B[r][1] * h1 + B[r][2] * h2);
                double A[2][CNP];
                {
                    const double2 *ra2 = reinterpret_cast<const double2 *>(row);
                    double Af[2 * CNP];
#pragma unroll
                    for (int t = 0; t < CNP; ++t) { const double2 v = ra2[t]; Af[2 * t] = v.x; Af[2 * t + 1] = v.y; }
#pragma unroll
                    for (int t = 0; t < CNP; ++t) { A[0][t] = Af[t]; A[1][t] = Af[CNP + t]; }
                }
                // (in registers: six lanes of a batch share a camera, and ds_add_f64 on one address serialises them -- 14 such
                // instructions per batch were a third of the kernel's LDS time)
#pragma unroll
                for (int t = 0; t < CNP; ++t) {
                    ea[t] = fma(A[1][t], e[1], fma(A[0][t], e[0], ea[t]));
                    Ev[t] = fma(A[1][t], er[1], fma(A[0][t], er[0], Ev[t]));
                    if (g.want_diag) ladd(&sC[cs * SC + 2 * CNP + t], fma(A[1][t], A[1][t], A[0][t] * A[0][t]));
                }
            }
        }
        wave_lds_sync();
        STAMP(4);
        // the next batch's streams: issued here, consumed in its phase 1 -- behind them nothing in the loop waits for memory before that
        // (a vmcnt wait right after the issue -- a register reloaded from scratch is enough -- would expose the whole HBM latency)
        if (k + 1 < nbatch) stream_batch(k + 1);
        // ---- phase 5: the lane's block += A_a^T (delta_ab I - G_a B_b^T) A_b of the round's point
        __builtin_amdgcn_s_setprio(0);
        {
            const int nrounds = (npt + PPR - 1) / PPR;
            const double *ra = sR + (ps * D + pa) * SR, *rb = sR + (ps * D + pb) * SR;
            const int step = PPR * D * SR;
            for (int r = 0; r < nrounds; ++r, ra += step, rb += step) {
                if (pair_lane && r * PPR + ps < npt) {
                    // the order of the loads is pinned (sched_barrier): all twenty operand loads in flight at once would need 80 registers
                    // beside the 98 accumulators; G_a and B_b are dead once M is formed, A_b once T is
                    const double2 *Ga2 = reinterpret_cast<const double2 *>(ra + OG), *Bb2 = reinterpret_cast<const double2 *>(rb + OB);
                    const double2 *Ab2 = reinterpret_cast<const double2 *>(rb), *Aa2 = reinterpret_cast<const double2 *>(ra);
                    const double2 g01 = Ga2[0], g23 = Ga2[1], g45 = Ga2[2], b01 = Bb2[0], b23 = Bb2[1], b45 = Bb2[2];
                    double Ab[2 * CNP], Aa[2 * CNP];
#pragma unroll
                    for (int t = 0; t < CNP; ++t) { const double2 v = Ab2[t]; Ab[2 * t] = v.x; Ab[2 * t + 1] = v.y; }
                    __builtin_amdgcn_sched_barrier(0);
                    // M = delta I - G_a B_b^T  (G rows: g01.x g01.y g23.x | g23.y g45.x g45.y; B rows alike)
                    const double M00 = fma(-g23.x, b23.x, fma(-g01.y, b01.y, fma(-g01.x, b01.x, dg)));
                    const double M01 = -fma(g23.x, b45.y, fma(g01.y, b45.x, g01.x * b23.y));
                    const double M10 = -fma(g45.y, b23.x, fma(g45.x, b01.y, g23.y * b01.x));
                    const double M11 = fma(-g45.y, b45.y, fma(-g45.x, b45.x, fma(-g23.y, b23.y, dg)));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < CNP; ++t) { const double2 v = Aa2[t]; Aa[2 * t] = v.x; Aa[2 * t + 1] = v.y; }
                    __builtin_amdgcn_sched_barrier(0);
                    double T0[CNP], T1[CNP];
#pragma unroll
                    for (int c = 0; c < CNP; ++c) {
                        T0[c] = fma(M01, Ab[CNP + c], M00 * Ab[c]);
                        T1[c] = fma(M11, Ab[CNP + c], M10 * Ab[c]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rr = 0; rr < CNP; ++rr) {
                        const double a0 = Aa[rr], a1 = Aa[CNP + rr];
#pragma unroll
                        for (int c = 0; c < CNP; ++c) acc[rr * CNP + c] = fma(a1, T1[c], fma(a0, T0[c], acc[rr * CNP + c]));
                    }
                }
            }
        }
        wave_lds_sync();
        STAMP(5);
    }
    // ---- the tile hands over: blocks registers -> the rows' LDS (dead now) -> HBM, one contiguous f64 atomic instruction per block;
    // with several points per round the copies of a block are summed in lane order first
    {
        const int rr = lane / CNP, cc = lane - rr * CNP;
        for (int c0 = 0; c0 < NB; c0 += kStageBlocks) {
            for (int s2 = 0; s2 < PPR; ++s2) {
                if (ps == s2 && pq >= c0 && pq < c0 + kStageBlocks) {
                    double *dst = sR + (pq - c0) * BLK;
                    if (s2 == 0) {
#pragma unroll
                        for (int i = 0; i < BLK; ++i) dst[i] = acc[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < BLK; ++i) dst[i] += acc[i];
                    }
                }
                wave_lds_sync();
            }
            const int c1 = min(NB, c0 + kStageBlocks);
            for (int q = c0; q < c1; ++q) {
                if (g.tile_blk) { if (lane < BLK) g.tile_blk[((size_t)tile * kNSlot + q) * BLK + lane] = sR[(q - c0) * BLK + lane]; }
                else {
                    const int gb = rl(my_gblk, q);
                    if (lane < BLK) gadd(&g.S_up[(size_t)gb * (kCP * kCP) + rr * kCP + cc], sR[(q - c0) * BLK + lane]);
                }
            }
            wave_lds_sync();
        }
        if (lane < PB * D) {                               // the lanes' shares of e_a and E: the (up to 16) lanes of a camera add up in its slot
#pragma unroll
            for (int t = 0; t < CNP; ++t) { ladd(&sC[cs * SC + t], ea[t]); ladd(&sC[cs * SC + CNP + t], Ev[t]); }
        }
        wave_lds_sync();
        const int which = lane / CNP, t = lane - which * CNP;
        for (int c = 0; c < D; ++c) {
            if (g.tile_blk) { if (lane < 3 * CNP) g.tile_cam[((size_t)tile * kNCSlot + c) * (3 * kCP) + lane] = sC[c * SC + lane]; }
            else {
                const int cam = rl(my_cam, c);
                if (lane < 3 * CNP) gadd(&g.camvec[((size_t)which * g.m + cam) * kCP + t], sC[c * SC + lane]);
            }
        }
    }
    STAMP(6);
#ifdef MIBA_STAMPS
    if (lane == 0 && g.dbg)
        for (int i = 0; i < 8; ++i) atomicAdd(&g.dbg[i], stamp_acc[i]);
#endif
    cost = wave_sum(cost); pl2 = wave_sum(pl2); ccost = wave_sum(ccost);
    ebinf = wave_max(ebinf); maxdv = wave_max(maxdv); sing = wave_max(sing);
    if (lane == 0) {
        double *pr = g.part + (size_t)tile * kPartW;
        pr[0] = cost; pr[1] = pl2; pr[2] = ccost; pr[3] = 0.0; pr[4] = ebinf; pr[5] = maxdv; pr[6] = sing; pr[7] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// Ordered hand-over (mi_ba_config.deterministic = 1): the register and table tiles leave their blocks and camera vectors in
// per-tile copies (k_build above) instead of adding them to S with f64 atomics; these two kernels sum the copies in tile order --
// one wavefront per block of S / per camera, the plan's list of (tile, slot) contributors -- so two runs give the same bits.
// They run before the row path's kernels, which add theirs on top in stream order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_reduce_blocks(int nb_up, int cnp, const int *__restrict__ ptr, const int *__restrict__ list,
                                                           const double *__restrict__ tile_blk, double *__restrict__ S_up)
{
    const int gb = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (gb >= nb_up || lane >= cnp * cnp) return;
    const int q0 = ptr[gb], q1 = ptr[gb + 1], blk = cnp * cnp;
    if (q0 == q1) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int q = q0;
    for (; q + 3 < q1; q += 4) {      // four copies in flight; the order of the sum is fixed all the same
        const int e0 = list[q], e1 = list[q + 1], e2 = list[q + 2], e3 = list[q + 3];
        const double v0 = tile_blk[(size_t)e0 * blk + lane], v1 = tile_blk[(size_t)e1 * blk + lane], v2 = tile_blk[(size_t)e2 * blk + lane],
                     v3 = tile_blk[(size_t)e3 * blk + lane];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; q < q1; ++q) s0 += tile_blk[(size_t)list[q] * blk + lane];
    const int rr = lane / cnp, cc = lane - rr * cnp;
    S_up[(size_t)gb * (kCP * kCP) + rr * kCP + cc] = (s0 + s1) + (s2 + s3);
}

// (cameras: a camera sits in ~60 tiles of C3; the groups of 3 cnp lanes a wavefront holds -- three, two with 8 parameters -- take the
// copies in turn, two in flight each, and are added up in group order)
__global__ __launch_bounds__(256) void k_tile_reduce_cams(int m, int cnp, const int *__restrict__ ptr, const int *__restrict__ list,
                                                         const double *__restrict__ tile_cam, double *__restrict__ camvec)
{
    const int cam = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, w = 3 * cnp;
    if (cam >= m) return;
    const int q0 = ptr[cam], q1 = ptr[cam + 1];
    if (q0 == q1) return;
    const int ngrp = kWave / w, grp = lane / w, el = lane - grp * w;        // 3 groups up to 7 parameters, 2 with 8
    double s0 = 0.0, s1 = 0.0;
    if (grp < ngrp) {
        int q = q0 + grp;
        for (; q + ngrp < q1; q += 2 * ngrp) { s0 += tile_cam[(size_t)list[q] * (3 * kCP) + el]; s1 += tile_cam[(size_t)list[q + ngrp] * (3 * kCP) + el]; }
        if (q < q1) s0 += tile_cam[(size_t)list[q] * (3 * kCP) + el];
    }
    const double t = s0 + s1;
    const double t1 = __shfl(t, (lane + w) & 63), t2 = ngrp > 2 ? __shfl(t, (lane + 2 * w) & 63) : 0.0;
    if (lane < w) {
        const int which = lane / cnp, k = lane - which * cnp;
        camvec[((size_t)which * m + cam) * kCP + k] = (t + t1) + t2;
    }
}

// ------------------------------------------------------------------------------------------
// Long tracks: points seen by more than 64 cameras do not fit a wavefront batch.  They take a
// workgroup each (k_long_reduce: phases 1-4 with block reductions) and leave the same observation
// records as the row tiles; their d(d+1)/2 pair blocks are k_schur_blocks' work like everybody else's.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double *sh)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <int CNP>
__global__ __launch_bounds__(256) void k_long_reduce(const LongArgs g)
{
    using L = RecL<CNP>;
    __shared__ double sh[4], sVh[9];
    if (g.spec && g.spec[0] == 0.0) return;
    const double mu_lin = g.spec ? g.spec[1] : g.mu;
    const int l = blockIdx.x, ip = g.long_pts[l], tid = threadIdx.x;
    const int os = g.pt_obs[ip], oe = g.pt_obs[ip + 1];
    const double X[3] = { g.pts[3 * (size_t)ip], g.pts[3 * (size_t)ip + 1], g.pts[3 * (size_t)ip + 2] };
    double v[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, cost = 0.0;
    for (int o = os + tid; o < oe; o += 256) {
        const int j = g.obs_cam[o];
        double A[2][CNP], B[2][3], e[2];
        linearize_obs<CNP, true>(g.cfg, g.cam_lin + (size_t)j * kCamLin, X, g.obs_xy[o], g.cam_fixed[j] != 0, e, A, B);
        cost += e[0] * e[0] + e[1] * e[1];
        v[0] += B[0][0] * B[0][0] + B[1][0] * B[1][0]; v[1] += B[0][0] * B[0][1] + B[1][0] * B[1][1];
        v[2] += B[0][0] * B[0][2] + B[1][0] * B[1][2]; v[3] += B[0][1] * B[0][1] + B[1][1] * B[1][1];
        v[4] += B[0][1] * B[0][2] + B[1][1] * B[1][2]; v[5] += B[0][2] * B[0][2] + B[1][2] * B[1][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) v[6 + r] += B[0][r] * e[0] + B[1][r] * e[1];
        double *row = g.rec + (size_t)g.obs_rec[o] * L::SIZE;
#pragma unroll
        for (int t = 0; t < L::AP; ++t) { row[L::A0 + t] = t < CNP ? A[0][t < CNP ? t : 0] : 0.0; row[L::A1 + t] = t < CNP ? A[1][t < CNP ? t : 0] : 0.0; }
#pragma unroll
        for (int t = 0; t < 3; ++t) { row[L::B + t] = B[0][t]; row[L::B + 3 + t] = B[1][t]; }
        row[L::E] = e[0]; row[L::E + 1] = e[1];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) v[c] = block_sum_256(v[c], sh);
    cost = block_sum_256(cost, sh);
    if (tid == 0) {
        double ccost = 0.0;
        if (g.pt_cw && g.pt_cw[ip] != 0.0) {
            const double w = g.pt_cw[ip];
            for (int t = 0; t < 3; ++t) {
                const double d = g.pt_cv[3 * (size_t)ip + t] - X[t];
                v[6 + t] += w * d; ccost += w * d * d;
            }
            v[0] += w; v[3] += w; v[5] += w;
        }
        double v00 = v[0], v01 = v[1], v02 = v[2], v11 = v[3], v12 = v[4], v22 = v[5];
        const double maxdv = fmax(v00, fmax(v11, v22)), ebinf = fmax(fabs(v[6]), fmax(fabs(v[7]), fabs(v[8])));
        if (g.cfg.damp_diag) { v00 = fma(mu_lin, v00, v00); v11 = fma(mu_lin, v11, v11); v22 = fma(mu_lin, v22, v22); }
        else { v00 += mu_lin; v11 += mu_lin; v22 += mu_lin; }
        const double c00 = v11 * v22 - v12 * v12, c01 = v02 * v12 - v01 * v22, c02 = v01 * v12 - v02 * v11;
        const double det = v00 * c00 + v01 * c01 + v02 * c02;
        double inv[6] = { 0, 0, 0, 0, 0, 0 }, sing = 0.0;
        if (det == 0.0 || !isfinite(det)) { if (!(g.cfg.points_fixed && g.cfg.damp_diag)) sing = 1.0; }
        else {
            const double id = 1.0 / det;
            inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
            inv[3] = (v00 * v22 - v02 * v02) * id; inv[4] = (v01 * v02 - v00 * v12) * id; inv[5] = (v00 * v11 - v01 * v01) * id;
        }
        const double h0 = inv[0] * v[6] + inv[1] * v[7] + inv[2] * v[8], h1 = inv[1] * v[6] + inv[3] * v[7] + inv[4] * v[8],
                     h2 = inv[2] * v[6] + inv[4] * v[7] + inv[5] * v[8];
        double *po = g.ptout + (size_t)ip * kPtOut;
#pragma unroll
        for (int c = 0; c < 6; ++c) { po[c] = inv[c]; sVh[c] = inv[c]; }
        po[6] = h0; po[7] = h1; po[8] = h2; po[9] = v[6]; po[10] = v[7]; po[11] = v[8];
        sVh[6] = h0; sVh[7] = h1; sVh[8] = h2;
        double *pr = g.part + (size_t)(g.part_row0 + l) * kPartW;
        pr[0] = cost; pr[1] = X[0] * X[0] + X[1] * X[1] + X[2] * X[2]; pr[2] = ccost; pr[3] = 0.0;
        pr[4] = ebinf; pr[5] = maxdv; pr[6] = sing; pr[7] = 0.0;
    }
    __syncthreads();
    const double i00 = sVh[0], i01 = sVh[1], i02 = sVh[2], i11 = sVh[3], i12 = sVh[4], i22 = sVh[5];
    for (int o = os + tid; o < oe; o += 256) {
        double *row = g.rec + (size_t)g.obs_rec[o] * L::SIZE; // written by this same thread above
        const double B0[3] = { row[L::B], row[L::B + 1], row[L::B + 2] }, B1[3] = { row[L::B + 3], row[L::B + 4], row[L::B + 5] };
        double *G = row + L::G;
        G[0] = B0[0] * i00 + B0[1] * i01 + B0[2] * i02; G[1] = B0[0] * i01 + B0[1] * i11 + B0[2] * i12;
        G[2] = B0[0] * i02 + B0[1] * i12 + B0[2] * i22; G[3] = B1[0] * i00 + B1[1] * i01 + B1[2] * i02;
        G[4] = B1[0] * i01 + B1[1] * i11 + B1[2] * i12; G[5] = B1[0] * i02 + B1[1] * i12 + B1[2] * i22;
        row[L::ER] = row[L::E] - (B0[0] * sVh[6] + B0[1] * sVh[7] + B0[2] * sVh[8]);
        row[L::ER + 1] = row[L::E + 1] - (B1[0] * sVh[6] + B1[1] * sVh[7] + B1[2] * sVh[8]);
    }
}

// ------------------------------------------------------------------------------------------
// k_schur_blocks: S_jk = delta_jk U_j - sum_i Y_ij W_ik^T and e_j = ea_j - sum_i Y_ij eb_i (sba_levmar.c:1182-1339) for the
// points of the row path, BLOCK-major: one wavefront per block of S, a lane per (observation of camera j, observation of
// camera k) pair of one point from the plan's list, CNP x CNP accumulators per lane across the whole list, one reduction over
// the lanes through LDS and one store.  No atomics: the sum has a fixed order, whatever the visibility looks like.
// DIAG: the diagonal block of a camera runs over the camera's records themselves and also leaves ea_j, E_j (and diag U_j).
// ------------------------------------------------------------------------------------------
template <int CNP, bool DIAG>
__global__ __launch_bounds__(64, (CNP <= 7 && !DIAG) ? 2 : 1) void k_schur_blocks(const SchurArgs g)
{
    using L = RecL<CNP>;
    constexpr int BLK = CNP * CNP, AP = L::AP;
    constexpr int NV = DIAG ? BLK + 3 * CNP : BLK;
    constexpr int ST = NV | 1;                       // odd row stride: the lanes' ds_write_b64 and the column reads are conflict-free
    __shared__ double sRed[32 * ST];
    if (g.spec && g.spec[0] == 0.0) return;
    const int lane = threadIdx.x;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), so XCD x takes the x-th
    // contiguous eighth of the work list -- the blocks of one camera row, which re-read that camera's records, then meet
    // in ONE L2 instead of in eight (placement is for speed only: any mapping gives the same sums)
    const int per = (int)(gridDim.x >> 3);                      // the grid is 8 x ceil(items / 8)
    const int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (w >= (DIAG ? g.ndiag : g.nwork)) return;
    long long q0, q1;
    int gb, cam = 0;
    if constexpr (DIAG) { cam = g.diag_cam[w]; gb = g.diag_blk[w]; q0 = g.rec_cam[cam]; q1 = g.rec_cam[cam + 1]; }
    else { gb = g.wk_blk[w]; q0 = g.wk_q[w]; q1 = g.wk_q[w + 1]; }
    const uint2 *rp = reinterpret_cast<const uint2 *>(g.rpair);
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    uint2 ent_next = make_uint2(0u, 0u);
    if constexpr (!DIAG) { if (q0 + lane < q1) ent_next = rp[q0 + lane]; }
    for (long long q = q0 + lane; q < q1; q += kWave) {
        size_t ra, rb;
        if constexpr (DIAG) { ra = rb = (size_t)q; }
        else { ra = ent_next.x; rb = ent_next.y; if (q + kWave < q1) ent_next = rp[q + kWave]; }
        const double2 *pa = reinterpret_cast<const double2 *>(g.rec + ra * L::SIZE);
        const double2 *pb = reinterpret_cast<const double2 *>(g.rec + rb * L::SIZE);
        double Aa[2 * AP], Ab[2 * AP], G[6], Bb[6];
#pragma unroll
        for (int t = 0; t < AP; ++t) { const double2 v = pa[t]; Aa[2 * t] = v.x; Aa[2 * t + 1] = v.y; }
#pragma unroll
        for (int t = 0; t < 3; ++t) { const double2 v = pa[L::G / 2 + t]; G[2 * t] = v.x; G[2 * t + 1] = v.y; }
#pragma unroll
        for (int t = 0; t < 3; ++t) { const double2 v = pb[L::B / 2 + t]; Bb[2 * t] = v.x; Bb[2 * t + 1] = v.y; }
        if constexpr (DIAG) {
#pragma unroll
            for (int t = 0; t < 2 * AP; ++t) Ab[t] = Aa[t];
        } else {
#pragma unroll
            for (int t = 0; t < AP; ++t) { const double2 v = pb[t]; Ab[2 * t] = v.x; Ab[2 * t + 1] = v.y; }
        }
        // M = delta_ab I - G_a B_b^T;  G = [G0 (3) | G1 (3)], B = [B0 (3) | B1 (3)]
        double M00 = fma(G[2], Bb[2], fma(G[1], Bb[1], G[0] * Bb[0])), M01 = fma(G[2], Bb[5], fma(G[1], Bb[4], G[0] * Bb[3]));
        double M10 = fma(G[5], Bb[2], fma(G[4], Bb[1], G[3] * Bb[0])), M11 = fma(G[5], Bb[5], fma(G[4], Bb[4], G[3] * Bb[3]));
        const double dg = DIAG ? 1.0 : 0.0;
        M00 = dg - M00; M01 = -M01; M10 = -M10; M11 = dg - M11;
        double T0[CNP], T1[CNP];
#pragma unroll
        for (int c = 0; c < CNP; ++c) { T0[c] = fma(M01, Ab[AP + c], M00 * Ab[c]); T1[c] = fma(M11, Ab[AP + c], M10 * Ab[c]); }
#pragma unroll
        for (int r = 0; r < CNP; ++r)
#pragma unroll
            for (int c = 0; c < CNP; ++c) acc[r * CNP + c] = fma(Aa[AP + r], T1[c], fma(Aa[r], T0[c], acc[r * CNP + c]));
        if constexpr (DIAG) {
            const double2 ee = pa[L::E / 2], rr = pa[L::ER / 2];
#pragma unroll
            for (int r = 0; r < CNP; ++r) {
                acc[BLK + r] = fma(Aa[AP + r], ee.y, fma(Aa[r], ee.x, acc[BLK + r]));
                acc[BLK + CNP + r] = fma(Aa[AP + r], rr.y, fma(Aa[r], rr.x, acc[BLK + CNP + r]));
                acc[BLK + 2 * CNP + r] = fma(Aa[AP + r], Aa[AP + r], fma(Aa[r], Aa[r], acc[BLK + 2 * CNP + r]));
            }
        }
    }
    // the lanes' partial blocks -> one: the upper half of the wavefront onto the lower through LDS, then every value of
    // the block is one lane's column sum over the <= 32 rows that hold something
    const long long npairs = q1 - q0;
    if (npairs > 32) {
        if (lane >= 32) {
#pragma unroll
            for (int i = 0; i < NV; ++i) sRed[(lane - 32) * ST + i] = acc[i];
        }
        wave_lds_sync();
        if (lane < 32) {
#pragma unroll
            for (int i = 0; i < NV; ++i) acc[i] += sRed[lane * ST + i];
        }
        wave_lds_sync();
    }
    if (lane < 32) {
#pragma unroll
        for (int i = 0; i < NV; ++i) sRed[lane * ST + i] = acc[i];
    }
    wave_lds_sync();
    const int rows = npairs < 32 ? (int)npairs : 32;
    for (int c = lane; c < NV; c += kWave) {
        double s0 = 0.0, s1 = 0.0;
        int r = 0;
        for (; r + 1 < rows; r += 2) { s0 += sRed[r * ST + c]; s1 += sRed[(r + 1) * ST + c]; }
        if (r < rows) s0 += sRed[r * ST + c];
        const double v = s0 + s1;
        if (c < BLK) {
            const int br = c / CNP, bc = c % CNP;
            if constexpr (CNP <= kCP) {
                // (the register tiles of the same linearisation -- an earlier kernel on this stream -- may have added here already)
                g.S_up[(size_t)gb * (kCP * kCP) + br * kCP + bc] += v;
            } else {
                // nine parameters: the block lands in the blocks of the two virtual cameras of each side (kernels.h ModelCfg.split):
                // (2a, 2b) parameters 0-7 x 0-7, (2a, 2b+1) x parameter 8, (2a+1, 2b) parameter 8 x, (2a+1, 2b+1) the corner
                const int *vb = (DIAG ? g.diag_vblk : g.wk_vblk) + (size_t)w * 4;
                const int hi_r = br == kCP, hi_c = bc == kCP;
                const int which = 2 * hi_r + hi_c;
                if (!(DIAG && which == 2))                                   // the diagonal item's lower corner block is its upper one mirrored
                    g.S_up[(size_t)vb[which] * (kCP * kCP) + (hi_r ? 0 : br) * kCP + (hi_c ? 0 : bc)] += v;
            }
        } else if constexpr (DIAG) {
            const int which = (c - BLK) / CNP, t = (c - BLK) % CNP;
            if (which < 2 || g.want_diag) {
                const int vcam = CNP <= kCP ? cam : 2 * cam + (t == kCP), vt = (CNP > kCP && t == kCP) ? 0 : t;
                g.camvec[((size_t)which * g.mv + vcam) * kCP + vt] += v;
            }
        }
    }
}

// The off-diagonal blocks, staged: what bounds the kernel above there is the texture addresser -- a lane fetching its own two
// records issues 22 divergent 16-byte requests per pair, and the addresser retires about one such request per cycle per CU
// (measured: TA_BUSY 79 %, 22.4 cache accesses per pair, 34 cycles per pair per CU).  Here the wavefront fetches the 64 pairs'
// records TOGETHER, four lanes per 64-byte line (one request per line: 6 per pair instead of 22), parks them in LDS and every
// lane then reads its own pair's operands from there (ds_read_b128, conflict-free: 208-byte slots).  All 24 fetches of a round
// are issued before anything is waited for: one round of memory latency per 64 pairs, as before.
template <int CNP>
__global__ __launch_bounds__(64, CNP <= 7 ? 2 : 1) void k_schur_pairs(const SchurArgs g)
{
    constexpr int BLK = CNP * CNP, ST = BLK | 1;
    constexpr int SLOT = 26;                                        // doubles per pair and side in the staging buffer: 3 lines (24) + 2 pad = 208 B
    constexpr int NBUF = (2 * kWave * SLOT > 32 * ST) ? 2 * kWave * SLOT : 32 * ST;
    __shared__ __attribute__((aligned(16))) double sBuf[NBUF];
    if (g.spec && g.spec[0] == 0.0) return;
    const int lane = threadIdx.x;
    const int per = (int)(gridDim.x >> 3);
    const int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);          // XCD-aware order, as above
    if (w >= g.nwork) return;
    const int gb = g.wk_blk[w];
    const long long q0 = g.wk_q[w], q1 = g.wk_q[w + 1];
    const uint2 *rp = reinterpret_cast<const uint2 *>(g.rpair);
    // the quad's pair of a round: consecutive quads take pairs 8 apart, so that the two quads of one 8-lane ds_write_b128 group
    // hit slots 8 x 208 B = 32 banks apart (consecutive slots would overlap in 4 banks: measured 33 % of the LDS cycles)
    const int quad = ((lane >> 2) & 1) * 8 + (lane >> 3), piece = lane & 3;
    double acc[BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i) acc[i] = 0.0;
    // the four pairs whose lines this lane helps to fetch (pair j * 16 + quad of a round), clamped inside the list; one round ahead
    uint2 ent_next[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const long long q = q0 + j * 16 + quad; ent_next[j] = rp[q < q1 ? q : q1 - 1]; }
    for (long long qb = q0; qb < q1; qb += kWave) {
        uint2 ent[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ent[j] = ent_next[j];
        if (qb + kWave < q1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const long long q = qb + kWave + j * 16 + quad; ent_next[j] = rp[q < q1 ? q : q1 - 1]; }
        }
        const bool valid = qb + lane < q1;
        // all 24 line fetches of the round at once (one round of memory latency), then through LDS to the lanes that own the pairs
        double2 st[2][12];
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int k = 0; k < 12; ++k) {                          // instruction k: line k / 4 of pairs (k % 4) * 16 + quad
                const int lj = k >> 2, j = k & 3;
                const unsigned r = side == 0 ? ent[j].x : ent[j].y;
                const int line = side == 0 ? (lj == 2 ? 3 : lj) : lj;    // a side: A (lines 0, 1), G (line 3); b side: A, B (line 2)
                st[side][k] = *reinterpret_cast<const double2 *>(g.rec + (size_t)r * kRec + line * 8 + piece * 2);
            }
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int lj = k >> 2, j = k & 3;
                *reinterpret_cast<double2 *>(sBuf + side * (kWave * SLOT) + (j * 16 + quad) * SLOT + lj * 8 + piece * 2) = st[side][k];
            }
        wave_lds_sync();
        double Aa[16], Ab[16], G[6], Bb[6];
        {
            const double2 *ma = reinterpret_cast<const double2 *>(sBuf + lane * SLOT);
            const double2 *mb = reinterpret_cast<const double2 *>(sBuf + kWave * SLOT + lane * SLOT);
#pragma unroll
            for (int t = 0; t < 8; ++t) { const double2 v = ma[t]; Aa[2 * t] = v.x; Aa[2 * t + 1] = v.y; }
#pragma unroll
            for (int t = 0; t < 3; ++t) { const double2 v = ma[8 + t]; G[2 * t] = v.x; G[2 * t + 1] = v.y; }
#pragma unroll
            for (int t = 0; t < 8; ++t) { const double2 v = mb[t]; Ab[2 * t] = v.x; Ab[2 * t + 1] = v.y; }
#pragma unroll
            for (int t = 0; t < 3; ++t) { const double2 v = mb[8 + t]; Bb[2 * t] = v.x; Bb[2 * t + 1] = v.y; }
        }
        wave_lds_sync();                                            // operands are in registers: the buffer is free for the next round
        if (valid) {
            const double M00 = -fma(G[2], Bb[2], fma(G[1], Bb[1], G[0] * Bb[0])), M01 = -fma(G[2], Bb[5], fma(G[1], Bb[4], G[0] * Bb[3]));
            const double M10 = -fma(G[5], Bb[2], fma(G[4], Bb[1], G[3] * Bb[0])), M11 = -fma(G[5], Bb[5], fma(G[4], Bb[4], G[3] * Bb[3]));
            double T0[CNP], T1[CNP];
#pragma unroll
            for (int c = 0; c < CNP; ++c) { T0[c] = fma(M01, Ab[8 + c], M00 * Ab[c]); T1[c] = fma(M11, Ab[8 + c], M10 * Ab[c]); }
#pragma unroll
            for (int r = 0; r < CNP; ++r)
#pragma unroll
                for (int c = 0; c < CNP; ++c) acc[r * CNP + c] = fma(Aa[8 + r], T1[c], fma(Aa[r], T0[c], acc[r * CNP + c]));
        }
    }
    const long long npairs = q1 - q0;
    if (npairs > 32) {
        if (lane >= 32) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) sBuf[(lane - 32) * ST + i] = acc[i];
        }
        wave_lds_sync();
        if (lane < 32) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) acc[i] += sBuf[lane * ST + i];
        }
        wave_lds_sync();
    }
    if (lane < 32) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) sBuf[lane * ST + i] = acc[i];
    }
    wave_lds_sync();
    const int rows = npairs < 32 ? (int)npairs : 32;
    if (lane < BLK) {
        double s0 = 0.0, s1 = 0.0;
        int r = 0;
        for (; r + 1 < rows; r += 2) { s0 += sBuf[r * ST + lane]; s1 += sBuf[(r + 1) * ST + lane]; }
        if (r < rows) s0 += sBuf[r * ST + lane];
        g.S_up[(size_t)gb * (kCP * kCP) + (lane / CNP) * kCP + (lane % CNP)] += s0 + s1;
    }
}

// The off-diagonal blocks whose two cameras lie in different groups of GS (four, eight) consecutive cameras, by SUPER-BLOCK (plan.h sb_*, sb8_*): a
// work item is a (row group, column group) pair = GS x GS blocks of S, its list holds one entry per point both groups see -- the point's records
// of the GS row cameras and of the GS column cameras.  A round = 64 / GS^2 entries (four, one): 32 (16) records fetched once serve 64 pairs, where
// the pair list above fetches 2 records (6 lines) for every pair -- the kernel above is bound by exactly that traffic (C4: 18.5 GB of record
// lines per launch).  Measured on C4 (859 us, 820 000 rounds of four entries): without the arithmetic 865 us, with the records always the same 32
// (L2 hits) 549 us, with neither 417 us -- the arithmetic is hidden, the floor is the round's chain ids -> record lines -> LDS -> operands at two
// wavefronts per SIMD; a register prefetch of the next round and an LDS-DMA ring two rounds ahead were both built and changed nothing, groups of
// eight (half the lines per pair, a quarter of the work items) were slower (opt-in MI_BA_SUPER=8: 900 + 450 us).  Lane = (entry of the round,
// row camera, column camera): one block per lane across the whole list as before; with four entries a round the copies of a block are summed
// through LDS at the end; a lane whose entry lacks one of its two cameras idles for the round.
template <int CNP, int GS>
__global__ __launch_bounds__(64, CNP <= 7 ? 2 : 1) void k_schur_super(const SchurArgs g)
{
    static_assert(GS == 4 || GS == 8, "groups of four or eight cameras");
    constexpr int BLK = CNP * CNP, ST = BLK | 1;
    constexpr int EPR = kWave / (GS * GS), NR = EPR * 2 * GS, NH = NR / 16;   // entries per round, records per round, fetches of 16 records
    constexpr int SLOT = 26;                                        // doubles per record in the staging buffer: 3 lines (24) + 2 pad = 208 B
    constexpr int NBUF = (NR * SLOT > 32 * ST) ? NR * SLOT : 32 * ST;
    __shared__ __attribute__((aligned(16))) double sBuf[NBUF];
    __shared__ int sGb[32];
    if (g.spec && g.spec[0] == 0.0) return;
    const int lane = threadIdx.x, w = blockIdx.x;                  // (the work items come sorted, the longest first)
    const long long q0 = (GS == 4 ? g.sb_q : g.sb8_q)[w], q1 = (GS == 4 ? g.sb_q : g.sb8_q)[w + 1];
    const int rounds = (int)((q1 - q0) / EPR);
    const int *__restrict__ ent = (GS == 4 ? g.sb_ent : g.sb8_ent) + 2 * GS * q0;
    const int *__restrict__ blk = (GS == 4 ? g.sb_blk : g.sb8_blk) + (size_t)w * GS * GS;
    // as a fetcher: record slot `quad` of each sixteen of the round (two quads of one 8-lane ds_write_b128 group 8 slots = 32 banks apart), 16-byte piece
    const int quad = ((lane >> 2) & 1) * 8 + (lane >> 3), piece = lane & 3;
    const int line2 = (quad & GS) ? 2 : 3;                         // third line of a record: G (row side, the first GS slots of an entry) or B (column side)
    // as a pair: entry s of the round, row camera ja, column camera kb
    const int s = GS == 4 ? lane >> 4 : 0, ja = GS == 4 ? (lane >> 2) & 3 : lane >> 3, kb = lane & (GS - 1);
    const int aslot = s * 2 * GS + ja, bslot = s * 2 * GS + GS + kb;
    double acc[BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i) acc[i] = 0.0;
    int nF[NH], nVa = ent[aslot], nVb = ent[bslot];
#pragma unroll
    for (int h = 0; h < NH; ++h) nF[h] = ent[h * 16 + quad];
    for (int r = 0; r < rounds; ++r) {
        int iF[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) iF[h] = nF[h];
        const int iVa = nVa, iVb = nVb;
        if (r + 1 < rounds) {
            const int *en = ent + (size_t)(r + 1) * NR;
#pragma unroll
            for (int h = 0; h < NH; ++h) nF[h] = en[h * 16 + quad];
            nVa = en[aslot]; nVb = en[bslot];
        }
        double2 st[NH][3];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const double *rp = g.rec + (size_t)(iF[h] < 0 ? 0 : iF[h]) * kRec + piece * 2;
            st[h][0] = *reinterpret_cast<const double2 *>(rp); st[h][1] = *reinterpret_cast<const double2 *>(rp + 8); st[h][2] = *reinterpret_cast<const double2 *>(rp + line2 * 8);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                *reinterpret_cast<double2 *>(sBuf + (h * 16 + quad) * SLOT + k * 8 + piece * 2) = st[h][k];
        wave_lds_sync();
        double Aa[16], Ab[16], G[6], Bb[6];
        {
            const double2 *ma = reinterpret_cast<const double2 *>(sBuf + aslot * SLOT);
            const double2 *mb = reinterpret_cast<const double2 *>(sBuf + bslot * SLOT);
#pragma unroll
            for (int t = 0; t < 8; ++t) { const double2 v = ma[t]; Aa[2 * t] = v.x; Aa[2 * t + 1] = v.y; }
#pragma unroll
            for (int t = 0; t < 3; ++t) { const double2 v = ma[8 + t]; G[2 * t] = v.x; G[2 * t + 1] = v.y; }
#pragma unroll
            for (int t = 0; t < 8; ++t) { const double2 v = mb[t]; Ab[2 * t] = v.x; Ab[2 * t + 1] = v.y; }
#pragma unroll
            for (int t = 0; t < 3; ++t) { const double2 v = mb[8 + t]; Bb[2 * t] = v.x; Bb[2 * t + 1] = v.y; }
        }
        wave_lds_sync();                                            // operands are in registers: the buffer is free for the next round
        if (iVa >= 0 && iVb >= 0) {
            const double M00 = -fma(G[2], Bb[2], fma(G[1], Bb[1], G[0] * Bb[0])), M01 = -fma(G[2], Bb[5], fma(G[1], Bb[4], G[0] * Bb[3]));
            const double M10 = -fma(G[5], Bb[2], fma(G[4], Bb[1], G[3] * Bb[0])), M11 = -fma(G[5], Bb[5], fma(G[4], Bb[4], G[3] * Bb[3]));
            double T0[CNP], T1[CNP];
#pragma unroll
            for (int c = 0; c < CNP; ++c) { T0[c] = fma(M01, Ab[8 + c], M00 * Ab[c]); T1[c] = fma(M11, Ab[8 + c], M10 * Ab[c]); }
#pragma unroll
            for (int rr = 0; rr < CNP; ++rr)
#pragma unroll
                for (int c = 0; c < CNP; ++c) acc[rr * CNP + c] = fma(Aa[8 + rr], T1[c], fma(Aa[rr], T0[c], acc[rr * CNP + c]));
        }
    }
    // nb blocks (the staging buffer's rows 0 .. nb - 1 = blocks b0 .. of the work item) onto S: element i * 64 + lane of the list, every read
    // requested before the first is waited for (block after block they were 2 nb dependent round trips at the end of every work item)
    auto flush = [&](int b0, auto nb_c) {
        constexpr int nb = decltype(nb_c)::value;
        if (lane < nb) sGb[lane] = blk[b0 + lane];
        wave_lds_sync();
        constexpr int NEL = (nb * BLK + kWave - 1) / kWave;
        double *dst[NEL];
        double old[NEL], add[NEL];
#pragma unroll
        for (int i = 0; i < NEL; ++i) {
            const int idx = i * kWave + lane, b = idx / BLK, e = idx - b * BLK;
            const int gb = idx < nb * BLK ? sGb[b] : -1;
            dst[i] = gb >= 0 ? g.S_up + (size_t)gb * (kCP * kCP) + (e / CNP) * kCP + (e % CNP) : nullptr;
            add[i] = gb >= 0 ? sBuf[b * ST + e] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NEL; ++i) old[i] = dst[i] ? *dst[i] : 0.0;
#pragma unroll
        for (int i = 0; i < NEL; ++i) if (dst[i]) *dst[i] = old[i] + add[i];
        wave_lds_sync();
    };
    if constexpr (GS == 4) {
        // the four entries' copies of a block (lanes b, b + 16, b + 32, b + 48) -> one, in a fixed order: (0 + 2) + (1 + 3)
        if (lane >= 32) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) sBuf[(lane - 32) * ST + i] = acc[i];
        }
        wave_lds_sync();
        if (lane < 32) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) acc[i] += sBuf[lane * ST + i];
        }
        wave_lds_sync();
        if (lane >= 16 && lane < 32) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) sBuf[(lane - 16) * ST + i] = acc[i];
        }
        wave_lds_sync();
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) sBuf[lane * ST + i] = acc[i] + sBuf[lane * ST + i];
        }
        wave_lds_sync();
        flush(0, std::integral_constant<int, 16>{});
    } else {
        // one block per lane: through the buffer in two halves of 32 blocks
        for (int hh = 0; hh < 2; ++hh) {
            if ((lane >> 5) == hh) {
#pragma unroll
                for (int i = 0; i < BLK; ++i) sBuf[(lane & 31) * ST + i] = acc[i];
            }
            wave_lds_sync();
            flush(hh * 32, std::integral_constant<int, 32>{});
        }
    }
}

template <int CNP>
__global__ __launch_bounds__(256) void k_long_backsub(const LongArgs g)
{
    __shared__ double sh[4], sX[3];
    if (g.scal[SC_PCG_FINAL] == 0.0) return;   // launched ahead of the host's look at the reduced solve (engine.cpp solve_phase)
    const int l = blockIdx.x, ip = g.long_pts[l], tid = threadIdx.x;
    const int os = g.pt_obs[ip], oe = g.pt_obs[ip + 1];
    const double X[3] = { g.pts[3 * (size_t)ip], g.pts[3 * (size_t)ip + 1], g.pts[3 * (size_t)ip + 2] };
    double t3[3] = { 0, 0, 0 }, dv[3] = { 0, 0, 0 };
    using L = RecL<CNP>;
    for (int o = os + tid; o < oe; o += 256) {
        const int j = g.obs_cam[o];
        const double *row = g.rec + (size_t)g.obs_rec[o] * L::SIZE; // A, B, e of this trial's k_long_reduce
        double da[9];
        load_cam_params(g.cfg, g.dA, j, da);
        double w0 = 0.0, w1 = 0.0;
#pragma unroll
        for (int t = 0; t < CNP; ++t) { w0 += row[L::A0 + t] * da[t]; w1 += row[L::A1 + t] * da[t]; }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            t3[t] += row[L::B + t] * w0 + row[L::B + 3 + t] * w1;
            if (g.cfg.damp_diag) dv[t] += row[L::B + t] * row[L::B + t] + row[L::B + 3 + t] * row[L::B + 3 + t];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) t3[c] = block_sum_256(t3[c], sh);
    if (g.cfg.damp_diag) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dv[c] = block_sum_256(dv[c], sh);
    } else dv[0] = dv[1] = dv[2] = 1.0;
    double dl = 0.0, dp2 = 0.0;
    if (tid == 0) {
        const double *po = g.ptout + (size_t)ip * kPtOut;
        const double d0 = po[6] - (po[0] * t3[0] + po[1] * t3[1] + po[2] * t3[2]);
        const double d1 = po[7] - (po[1] * t3[0] + po[3] * t3[1] + po[4] * t3[2]);
        const double d2 = po[8] - (po[2] * t3[0] + po[4] * t3[1] + po[5] * t3[2]);
        sX[0] = X[0] + d0; sX[1] = X[1] + d1; sX[2] = X[2] + d2;
        g.pts_new[3 * (size_t)ip] = sX[0]; g.pts_new[3 * (size_t)ip + 1] = sX[1]; g.pts_new[3 * (size_t)ip + 2] = sX[2];
        if (g.dB) { g.dB[3 * (size_t)ip] = d0; g.dB[3 * (size_t)ip + 1] = d1; g.dB[3 * (size_t)ip + 2] = d2; }
        dl = d0 * (g.mu * dv[0] * d0 + po[9]) + d1 * (g.mu * dv[1] * d1 + po[10]) + d2 * (g.mu * dv[2] * d2 + po[11]);
        dp2 = dv[0] * d0 * d0 + dv[1] * d1 * d1 + dv[2] * d2 * d2;
    }
    __syncthreads();
    const double Xn[3] = { sX[0], sX[1], sX[2] };
    double cn = 0.0, pct = 0.0;
    for (int o = os + tid; o < oe; o += 256) {
        const int j = g.obs_cam[o];
        const double *row = g.rec + (size_t)g.obs_rec[o] * L::SIZE;
        const double2 xy = g.obs_xy[o];
        double u, v;
        project_obs<true>(g.cfg, g.cam_lin_new + (size_t)j * kCamLin, Xn, xy, u, v);
        const double en0 = xy.x - u, en1 = xy.y - v, e0 = row[L::E], e1 = row[L::E + 1];
        cn += en0 * en0 + en1 * en1;
        pct = fmax(pct, fmax(pct_change_flag(e0, en0, g.eps5), pct_change_flag(e1, en1, g.eps5)));
    }
    cn = block_sum_256(cn, sh);
    pct = wave_max(pct);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = pct;
    __syncthreads();
    if (tid == 0) {
        double *pr = g.part + (size_t)(g.part_row0 + l) * kPartW;
        pr[0] = cn; pr[1] = dl; pr[2] = dp2; pr[3] = 0.0; pr[4] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3])); pr[5] = 0.0; pr[6] = 0.0; pr[7] = 0.0;
    }
}

// folds per-workgroup partial rows: columns [0,4) are sums, [4,8) maxima.  Two stages when there are many
// rows: gridDim.x workgroups each fold a contiguous chunk into one row of `stage`, then one workgroup
// folds those (deterministic: fixed chunking and order).
// columns 0..3 of the rows [r0, r1) summed, 4..7 maximised, by one 1024-thread workgroup; thread c < 8 ends up with column c
__device__ __forceinline__ double reduce_rows(const double *__restrict__ part, int r0, int r1, double (*sh)[kPartW])
{
    double acc[kPartW];
#pragma unroll
    for (int c = 0; c < kPartW; ++c) acc[c] = 0.0;
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const double *pr = part + (size_t)r * kPartW;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += pr[c];
#pragma unroll
        for (int c = 4; c < kPartW; ++c) acc[c] = fmax(acc[c], pr[c]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);
#pragma unroll
    for (int c = 4; c < kPartW; ++c) acc[c] = wave_max(acc[c]);
    const int w = threadIdx.x >> 6;
    __syncthreads();                                   // sh may still be read from a previous call
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < kPartW; ++c) sh[w][c] = acc[c];
    __syncthreads();
    double v = 0.0;
    if (threadIdx.x < kPartW) {
        const int c = threadIdx.x, nw = blockDim.x >> 6;
        for (int k = 0; k < nw; ++k) v = (c < 4) ? v + sh[k][c] : fmax(v, sh[k][c]);
    }
    return v;
}

__global__ __launch_bounds__(1024) void k_reduce_partials(int nrows, const double *__restrict__ part, double *__restrict__ stage,
                                                        double *__restrict__ out_sum, double *__restrict__ out_max)
{
    __shared__ double sh[16][kPartW];
    const int chunk = (nrows + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * chunk, r1 = min(nrows, r0 + chunk);
    const double v = reduce_rows(part, r0, r1, sh);
    if (threadIdx.x < kPartW) {
        const int c = threadIdx.x;
        if (gridDim.x > 1) stage[(size_t)blockIdx.x * kPartW + c] = v;
        else if (c < 4) out_sum[c] = v;
        else out_max[c - 4] = v;
    }
}

// first stage of TWO row sets in one launch (the build's partial rows and the back-substitution's): workgroups [0, G / 2) take set A, the rest set B
__global__ __launch_bounds__(1024) void k_reduce_partials2(int nA, const double *__restrict__ partA, double *__restrict__ stageA,
                                                         int nB, const double *__restrict__ partB, double *__restrict__ stageB)
{
    __shared__ double sh[16][kPartW];
    const int half = (int)gridDim.x >> 1, second = (int)blockIdx.x >= half, wg = (int)blockIdx.x - (second ? half : 0);
    const int nrows = second ? nB : nA;
    const double *part = second ? partB : partA;
    double *stage = second ? stageB : stageA;
    const int chunk = (nrows + half - 1) / half;
    const int r0 = wg * chunk, r1 = min(nrows, r0 + chunk);
    const double v = reduce_rows(part, r0, r1, sh);
    if (threadIdx.x < kPartW) stage[(size_t)wg * kPartW + threadIdx.x] = v;
}

// The end of a trial in one launch: the last stage of the back-substitution's sums, the camera-side sums k_finalize_vec and
// k_cam_update left as partial rows (nothing on the device needs them earlier), and -- on one rank, where no exchange
// comes in between -- the controller's accept test and damping update (k_lm_peek below).
struct FinalReduceArgs {
    const double *bs_part; int bs_rows; double *bs_sum, *bs_max;       // back-substitution (or its first stage's output)
    const double *fv_part; int fv_rows;                                 // k_finalize_vec  -> scal[16, 24)
    const double *cu_part; int cu_rows;                                 // k_cam_update    -> scal[24, 28)
    const double *bp_part; int bp_rows; double *bp_sum, *bp_max;       // the build's partial rows of this trial's linearisation (one rank; 0 rows: summed after the build already)
    double *scal;
    double *spec; PeekArgs pk;                                          // spec != nullptr: peek
};

__device__ __forceinline__ void lm_peek(double *__restrict__ s, const PeekArgs &k, double *__restrict__ spec);

__global__ __launch_bounds__(1024) void k_reduce_final(const FinalReduceArgs a)
{
    // three sets of rows; each thread's first row of every set is requested before anything is added up, so the three sets
    // cost one round of memory latency (they come from other XCDs' kernels: L2 misses), not three
    __shared__ double sh[4][16][kPartW];
    const int w = threadIdx.x >> 6, tid = threadIdx.x;
    const double *part[4] = { a.fv_part, a.cu_part, a.bs_part, a.bp_rows > 0 ? a.bp_part : a.bs_part };
    const int rows[4] = { a.fv_rows, a.cu_rows, a.bs_rows, a.bp_rows };
    double acc[4][kPartW];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double *pr = part[k] + (size_t)(tid < rows[k] ? tid : 0) * kPartW;
#pragma unroll
        for (int c = 0; c < kPartW; ++c) acc[k][c] = pr[c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (tid >= rows[k]) {
#pragma unroll
            for (int c = 0; c < kPartW; ++c) acc[k][c] = 0.0;
        }
        for (int r = tid + (int)blockDim.x; r < rows[k]; r += blockDim.x) {
            const double *pr = part[k] + (size_t)r * kPartW;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[k][c] += pr[c];
#pragma unroll
            for (int c = 4; c < kPartW; ++c) acc[k][c] = fmax(acc[k][c], pr[c]);
        }
        // A wave-wide f64 sum is ~20 instructions; 16 wavefronts x 32 of them were most of this kernel's 11 us.  Only the wavefronts that hold rows of
        // a set reduce it, and only the columns its producer fills (k_finalize_vec: 0-2, 4, 5; k_cam_update: 0-2) -- the others are zero by construction.
        constexpr unsigned kUsed[4] = { 0x37u, 0x07u, 0xFFu, 0xFFu };
        const bool has = w * 64 < rows[k];                     // (wave-uniform)
#pragma unroll
        for (int c = 0; c < kPartW; ++c) {
            if (!has || !((kUsed[k] >> c) & 1u)) acc[k][c] = 0.0;
            else acc[k][c] = c < 4 ? wave_sum(acc[k][c]) : wave_max(acc[k][c]);
        }
        if ((tid & 63) == 0)
#pragma unroll
            for (int c = 0; c < kPartW; ++c) sh[k][w][c] = acc[k][c];
    }
    __syncthreads();
    if (threadIdx.x < 4 * kPartW) {
        const int k = threadIdx.x / kPartW, c = threadIdx.x % kPartW, nw = blockDim.x >> 6;
        double v = 0.0;
        for (int i = 0; i < nw; ++i) v = (c < 4) ? v + sh[k][i][c] : fmax(v, sh[k][i][c]);
        if (k == 0) { if (c < 4) a.scal[SC_PL2_CAMS + c] = v; else a.scal[SC_EA_INF + c - 4] = v; }
        else if (k == 1) { if (c < 4) a.scal[SC_DL_CAMS + c] = v; }
        else if (k == 2) { if (c < 4) a.bs_sum[c] = v; else a.bs_max[c - 4] = v; }
        else if (a.bp_rows > 0) { if (c < 4) a.bp_sum[c] = v; else a.bp_max[c - 4] = v; }
    }
    if (a.spec) {
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) lm_peek(a.scal, a.pk, a.spec);
    }
}

// ------------------------------------------------------------------------------------------
// k_finalize: upper blocks -> full BSR values of S = U* - W V*^-1 W^T  (sba_levmar.c:1270-1316:
// "copy the LOWER TRIANGULAR PART of S from the upper one"), damping on the diagonal (:1138-1142),
// identity on the padding row/column and on the rows of the mcon fixed cameras.
// one thread per matrix element, 64 threads = one block.
// ------------------------------------------------------------------------------------------
struct FinalizeArgs {
    int cnp;
    double mu;
    const unsigned char *fixed, *trans, *pfix;      // pfix[j] bit r: parameter r of camera j is held constant
    const int *rowidx, *col, *src;
    const double *S_up, *cam_cw;
    const double *spec;                             // speculative launch: the damping is spec[1]
    const double *diagU;                            // PBA's damping: mu diag(U_j) on the diagonal (camvec section 2), nullptr: mu
    CamLinks lk;
};

__device__ __forceinline__ double finalized_element(const FinalizeArgs &f, int blk, int el, const double *du_now = nullptr /* diag U of this camera as just updated (k_block_inv with k_finalize_vec's work inside) */)
{
    const int r = el >> 3, c = el & 7, j = f.rowidx[blk], k = f.col[blk];
    const double *sb = f.S_up + (size_t)f.src[blk] * 64;
    double v;
    if (j == k) {
        v = 0.5 * (sb[r * 8 + c] + sb[c * 8 + r]);
        if (r == c && r < f.cnp && f.lk.row)                           // shared-intrinsics links: U_jj += sum w
            for (int q = f.lk.row[j]; q < f.lk.row[j + 1]; ++q) v += f.lk.w[(size_t)q * kCP + r];
        if (r == c) v = (r < f.cnp) ? v + (f.spec ? f.spec[1] : f.mu) * (f.diagU ? (du_now ? du_now[r] : f.diagU[j * 8 + r]) : 1.0) + f.cam_cw[j * 8 + r] : 1.0;   // damping + soft camera constraint (sba_levmar.c:959)
        if (f.fixed[j]) v = (r == c) ? 1.0 : 0.0;
    } else {
        v = f.trans[blk] ? sb[c * 8 + r] : sb[r * 8 + c];
        if (r == c && f.lk.row) { const int pq = f.lk.blk_pair[blk]; if (pq >= 0) v -= f.lk.pair_w[(size_t)pq * kCP + r]; } // U_jk = -w
        if (f.fixed[j] || f.fixed[k]) v = 0.0;
    }
    if (r >= f.cnp || c >= f.cnp || ((f.pfix[j] >> r) & 1) || ((f.pfix[k] >> c) & 1)) v = (j == k && r == c) ? 1.0 : 0.0;   // padding, constant parameters
    return v;
}

// the full matrix in HBM: only for the parity hooks (mi_ba_debug_linearize) -- the solve reads the elements straight from
// the upper blocks (k_block_inv, k_scale_blocks)
__global__ __launch_bounds__(256) void k_finalize(int nb, const FinalizeArgs f, double *__restrict__ S_full)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int blk = (int)(gid >> 6), el = (int)(gid & 63);
    if (blk >= nb) return;
    S_full[(size_t)blk * 64 + el] = finalized_element(f, blk, el);
}

// right-hand side E_j = ea_j - sum_i Y_ij eb_i (already folded: E = sum A^T (e - B h)), ||ea||_inf,
// max diag U, ||a||^2   (sba_levmar.c:1085-1088, 1124-1128)
__global__ __launch_bounds__(256) void k_finalize_vec(int m, int cnp, const unsigned char *__restrict__ fixed, const unsigned char *__restrict__ pfix,
                                                     double *__restrict__ camvec,
                                                     const double *__restrict__ cam_a, const double *__restrict__ cam_cw,
                                                     const double *__restrict__ cam_cv, const CamLinks lk, double *__restrict__ E,
                                                     double *__restrict__ part)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    double eainf = 0.0, maxdu = 0.0, pl2 = 0.0, cc = 0.0, lc = 0.0;
    if (gid < m * kCP) {
        const int j = gid >> 3, r = gid & 7;
        const bool live = (r < cnp) && !fixed[j] && !((pfix[j] >> r) & 1);
        double ea = camvec[gid], Ev = camvec[(size_t)m * kCP + gid], du = camvec[2 * (size_t)m * kCP + gid];
        if (r < cnp) {
            const double a = cam_a[gid], w = cam_cw[gid], d = cam_cv[gid] - a;
            pl2 = a * a;
            cc = w * d * d;                                   // every camera, fixed ones too (sba_levmar.c:810-825)
            double add = 0.0, dadd = 0.0;
            if (w != 0.0) { add = w * d; dadd = w; }          // U_jj += w (k_finalize), ea_j += w diff (:959-960)
            if (lk.row)                                       // links: residual -(a_j - a_k), each link seen from both ends
                for (int q = lk.row[j]; q < lk.row[j + 1]; ++q) {
                    const double wq = lk.w[(size_t)q * kCP + r], dq = cam_a[(size_t)lk.other[q] * kCP + r] - a;
                    add = fma(wq, dq, add); dadd += wq;
                    lc = fma(0.5 * wq, dq * dq, lc);
                }
            if (live && (add != 0.0 || dadd != 0.0)) {
                ea += add; Ev += add; du += dadd;
                camvec[gid] = ea;                             // k_cam_update reads ea for dL
                camvec[2 * (size_t)m * kCP + gid] = du;       // ... and diag U for PBA's damping (finalized_element) and its dL
            }
        }
        E[gid] = live ? Ev : 0.0;
        if (live) { eainf = fabs(ea); maxdu = du; }
    }
    eainf = wave_max(eainf); maxdu = wave_max(maxdu); pl2 = wave_sum(pl2); cc = wave_sum(cc); lc = wave_sum(lc);
    if ((threadIdx.x & 63) == 0) { // one partial row per wave, folded in a fixed order by k_reduce_partials
        double *pr = part + (size_t)(gid >> 6) * kPartW;
        pr[0] = pl2; pr[1] = cc; pr[2] = lc; pr[3] = 0.0; pr[4] = eainf; pr[5] = maxdu; pr[6] = 0.0; pr[7] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// block-Jacobi preconditioning, applied to the matrix once per trial instead of to a vector twice per iteration:
// with S_jj = L_j L_j^T (unrolled 8x8 Cholesky per camera; PBA: ComputeDiagonalBlockC +
// InvertSymmetricMatrix, SparseBundleCPU.cpp:1604-1734) the solve runs on  S~ = L^-1 S L^-T  (unit diagonal blocks),
// E~ = L^-1 E, and da = L^-T x~.  Same iterates as block-Jacobi PCG on S (r~ = L^-1 r, so r~.r~ = r.M^-1 r, the quantity
// the stopping rule looks at).
// ------------------------------------------------------------------------------------------
// one wavefront per camera: the 64 lanes fetch the diagonal block's 64 finalised elements at once (one round of memory
// latency instead of 36 dependent ones), lane 0 factors it from LDS and writes L^-1
// k_finalize_vec's arguments, for its work done inside k_block_inv (fv.on; the trial's common case: one launch less): a camera's parameter r on lane (r, 0)
// of the camera's wavefront, one partial row per camera
struct FvArgs { int on; double *camvec; const double *cam_a, *cam_cw, *cam_cv; double *E, *part; };
__global__ __launch_bounds__(256) void k_block_inv(int m, const int *__restrict__ bsr_diag, const FinalizeArgs f,
                                                  double *__restrict__ Linv, double *__restrict__ scal, const PcgStart ps, double *__restrict__ Lfac, const FvArgs fv)
{
    __shared__ double sb[4][64], so[4][64], sE[4][8], sDu[4][8];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.x * 4 + wv;
    const bool live = j < m;
    if (ps.bar && blockIdx.x == 0)        // the persistent solve's barrier words start every trial from zero (k_pcg_persist)
        for (int i = threadIdx.x; i < kBarWords; i += blockDim.x) ps.bar[i] = 0ull;
    if (fv.on) {   // (uniform) the right-hand side E_j, |e_a|inf, max diag U, |a|^2 and the camera-side penalties of camera j: k_finalize_vec, line by line
        double eainf = 0.0, maxdu = 0.0, pl2 = 0.0, cc = 0.0, lc = 0.0;
        const int pr_ = lane >> 3;
        if (live && (lane & 7) == 0) {
            const int gid = j * kCP + pr_;
            const bool plive = (pr_ < f.cnp) && !f.fixed[j] && !((f.pfix[j] >> pr_) & 1);
            double ea = fv.camvec[gid], Ev = fv.camvec[(size_t)m * kCP + gid], du = fv.camvec[2 * (size_t)m * kCP + gid];
            if (pr_ < f.cnp) {
                const double a = fv.cam_a[gid], w = fv.cam_cw[gid], d = fv.cam_cv[gid] - a;
                pl2 = a * a;
                cc = w * d * d;
                double add = 0.0, dadd = 0.0;
                if (w != 0.0) { add = w * d; dadd = w; }
                if (f.lk.row)
                    for (int q = f.lk.row[j]; q < f.lk.row[j + 1]; ++q) {
                        const double wq = f.lk.w[(size_t)q * kCP + pr_], dq = fv.cam_a[(size_t)f.lk.other[q] * kCP + pr_] - a;
                        add = fma(wq, dq, add); dadd += wq;
                        lc = fma(0.5 * wq, dq * dq, lc);
                    }
                if (plive && (add != 0.0 || dadd != 0.0)) {
                    ea += add; Ev += add; du += dadd;
                    fv.camvec[gid] = ea;
                    fv.camvec[2 * (size_t)m * kCP + gid] = du;
                }
            }
            const double Eo = plive ? Ev : 0.0;
            fv.E[gid] = Eo; sE[wv][pr_] = Eo; sDu[wv][pr_] = du;
            if (plive) { eainf = fabs(ea); maxdu = du; }
        }
        eainf = wave_max(eainf); maxdu = wave_max(maxdu); pl2 = wave_sum(pl2); cc = wave_sum(cc); lc = wave_sum(lc);
        if (live && lane == 0) {
            double *pr = fv.part + (size_t)j * kPartW;
            pr[0] = pl2; pr[1] = cc; pr[2] = lc; pr[3] = 0.0; pr[4] = eainf; pr[5] = maxdu; pr[6] = 0.0; pr[7] = 0.0;
        }
        wave_lds_sync();
    }
    sb[wv][lane] = live ? finalized_element(f, bsr_diag[j], lane, fv.on ? sDu[wv] : nullptr) : (lane % 9 == 0 ? 1.0 : 0.0);
    wave_lds_sync();
    // right-looking Cholesky in LDS, lane (r, c): eight steps of (pivot, column, trailing update), the lower triangle of sb
    // becomes L; then lane c solves L x = e_c for column c of L^-1.  (One lane doing all of it serially took 7 us.)
    const int r = lane >> 3, c = lane & 7;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double piv = sb[wv][k * 9];
        if (!(piv > 0.0)) { ok = false; piv = 1.0; }
        const double lkk = sqrt(piv);
        const double lrk = sb[wv][r * 8 + k] / lkk, lck = sb[wv][c * 8 + k] / lkk;   // column k, rows r and c (used when > k)
        wave_lds_sync();
        if (c == k && r >= k) sb[wv][r * 8 + k] = (r == k) ? lkk : lrk;
        if (r > k && c > k && c <= r) sb[wv][r * 8 + c] -= lrk * lck;
        wave_lds_sync();
    }
    if (live && !ok && lane == 0) atomic_max_pos(&scal[SC_NOT_PD], 1.0);
    if (lane < 8) { // column `lane` of L^-1 by forward substitution
        double col[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double v = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < i; ++q) v -= sb[wv][i * 8 + q] * col[q];
            col[i] = (i < lane) ? 0.0 : v / sb[wv][i * 9];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) so[wv][i * 8 + lane] = col[i];
    }
    wave_lds_sync();
    if (live) {
        Linv[(size_t)j * 64 + lane] = so[wv][lane];
        if (Lfac) Lfac[(size_t)j * 64 + lane] = c <= r ? sb[wv][lane] : 0.0;   // the factor itself, for the coarse space (k_pcg2_*)
        if (lane < 8) { // the start of the CG on the scaled system: r_0 = E~ = L^-1 E, x~_0 = 0 (this camera's rows)
            const double *Ej = fv.on ? sE[wv] : ps.E + (size_t)j * 8;
            double rr = 0.0;
#pragma unroll
            for (int c = 0; c < 8; ++c) rr += so[wv][lane * 8 + c] * Ej[c];
            const int idx = j * 8 + lane;
            ps.x[idx] = 0.0; ps.r0[idx] = rr; ps.p[idx] = 0.0; ps.s0[idx] = 0.0; ps.s1[idx] = 0.0;
        }
    }
}

// S~_jk = Linv_j S_jk Linv_k^T, a wavefront per block at a time, lane (r,c); the block and the two factors go through LDS.  The wavefronts of the
// (chip-filling) grid walk the blocks in strides and keep TWO blocks ahead in flight -- the indices of the one after next (wave-uniform: scalar
// loads), the element and the two factors of the next -- while they multiply the present one: a wavefront per block with nothing ahead was a
// chain of two memory round trips and three LDS exchanges per block, 4.5 (C3) to 50 (C4) of them one after the other per wavefront slot
// (C3 19.5 us, C4 184 us for 26 / 260 MB).  Diagonal blocks are the identity by construction: nothing is fetched for them.
// layout 1: the scaled matrix is stored in single precision instead (PcgArgs.S32: the CG then streams half the bytes; engine.cpp pcg_f32_)
// Round 4, second step: only the blocks on and above the diagonal are multiplied -- S~_kj is S~_jk transposed, to the last bit, so the lane that holds
// element (r, c) of block (j, k) also stores it as element (c, r) of block (k, j): half the products and half the LDS traffic (the kernel is LDS-bound:
// 16 KB of operand reads per block).  `list`: per block with k >= j eight ints { j, k, upper block, b, b of the mirror block or -1, fixed-pitch position
// of b, of the mirror, - } (engine.cpp prepare): everything a block needs to know arrives in ONE round, two blocks ahead.
__global__ __launch_bounds__(256) void k_scale_blocks(int nl, const int4 *__restrict__ list, const FinalizeArgs f, const double *__restrict__ Linv, double *__restrict__ S_pcg, int layout, int ell)
{
    __shared__ double sh[4][4][64];   // per wavefront: S, T = Linv_j S, Linv_j, Linv_k
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
    const int nw = (int)gridDim.x * 4;
    int i = (int)blockIdx.x * 4 + wv;                 // (left a per-lane value on purpose: made wave-uniform, every index load is followed by a wait for its v_readfirstlane)
    if (i >= nl) return;
    struct Idx { int j, k, src, b, bm, pe, pme; };
    struct Dat { double s, lj, lk; int fx, pj, pk; };
    auto ldidx = [&](int ii) {                        // (past the end: the last entry again, never used)
        const int q = ii < nl ? ii : nl - 1;
        const int4 u = list[2 * (size_t)q], w2 = list[2 * (size_t)q + 1];
        return Idx{ u.x, u.y, u.z, u.w, w2.x, w2.y, w2.z };
    };
    auto lddat = [&](const Idx &x) {                  // every load unconditional and independent of the others
        Dat d;
        d.s = f.S_up[(size_t)x.src * 64 + r * 8 + c];
        d.lj = Linv[(size_t)x.j * 64 + lane]; d.lk = Linv[(size_t)x.k * 64 + lane];
        d.fx = (int)f.fixed[x.j] | (int)f.fixed[x.k]; d.pj = f.pfix[x.j]; d.pk = f.pfix[x.k];
        return d;
    };
    Idx i0 = ldidx(i), i1 = ldidx(i + nw);
    Dat d0 = lddat(i0);
    for (; i < nl; i += nw) {
        const Idx i2 = ldidx(i + 2 * nw);
        const Dat d1 = lddat(i1);
        const int j = i0.j, k = i0.k;
        double o = 0.0;
        if (j != k) {                                     // (the same for all lanes)
            double v = d0.s;
            if (r == c && f.lk.row) { const int pq = f.lk.blk_pair[i0.b]; if (pq >= 0) v -= f.lk.pair_w[(size_t)pq * kCP + r]; }   // shared-intrinsics links: U_jk = -w
            // fixed cameras, padding rows / columns, constant parameters (finalized_element)
            if (d0.fx | (r >= f.cnp) | (c >= f.cnp) | ((d0.pj >> r) & 1) | ((d0.pk >> c) & 1)) v = 0.0;
            sh[wv][0][lane] = v; sh[wv][2][lane] = d0.lj; sh[wv][3][lane] = d0.lk;
            wave_lds_sync();
            double t = 0.0;                               // T = Linv_j S
#pragma unroll
            for (int a = 0; a < 8; ++a) t = fma(sh[wv][2][r * 8 + a], sh[wv][0][a * 8 + c], t);
            sh[wv][1][lane] = t;
            wave_lds_sync();
#pragma unroll
            for (int a = 0; a < 8; ++a) o = fma(sh[wv][1][r * 8 + a], sh[wv][3][c * 8 + a], o);   // T Linv_k^T
            wave_lds_sync();                              // (the next block's operands overwrite the buffers)
        }
        const double val = (j == k) ? (r == c ? 1.0 : 0.0) : o;   // the diagonal blocks are the identity by construction
        // The product's column forms (row_products): a lane of the product holds ONE column's elements of a few rows and needs one 8-byte element of
        // the vector for them.  f64 (layout 2): element (r, c) at ((r >> 1) * 8 + c) * 2 + (r & 1) -- rows 2h, 2h + 1 of column c in one 16-byte load.
        // Single precision (layout 1): at ((r >> 2) * 8 + c) * 4 + (r & 3) -- rows 4h .. 4h + 3 of column c in one 16-byte load.
        // The mirror block gets the same value as its element (c, r).
        const bool mir = i0.bm >= 0;
        if (layout == 1) {
            float *Sf = reinterpret_cast<float *>(S_pcg);
            Sf[(size_t)i0.b * 64 + (((r >> 2) * 8 + c) * 4 + (r & 3))] = (float)val;
            if (mir) Sf[(size_t)i0.bm * 64 + (((c >> 2) * 8 + r) * 4 + (c & 3))] = (float)val;
        } else if (layout == 2) {
            S_pcg[(size_t)(ell ? i0.pe : i0.b) * 64 + (((r >> 1) * 8 + c) * 2 + (r & 1))] = val;
            if (mir) S_pcg[(size_t)(ell ? i0.pme : i0.bm) * 64 + (((c >> 1) * 8 + r) * 2 + (c & 1))] = val;
        } else {
            S_pcg[(size_t)i0.b * 64 + lane] = val;
            if (mir) S_pcg[(size_t)i0.bm * 64 + c * 8 + r] = val;
        }
        i0 = i1; i1 = i2; d0 = d1;
    }
}

// ------------------------------------------------------------------------------------------
// CG on the scaled block-sparse reduced camera system (the role of sba_Axb_Chol; what PBA's SolveNormalEquationPCGX,
// SparseBundleCPU.cpp:3347-3474, does with implicit products), in the Chronopoulos-Gear arrangement (one reduction per iteration):
//   w_i = S~ r_i,  gamma_i = r_i.r_i,  delta_i = w_i.r_i        (both dot products over vectors of the same launch)
//   beta_i = gamma_i / gamma_{i-1},  alpha_i = gamma_i / (delta_i - beta_i gamma_i / alpha_{i-1})
//   p_i = r_i + beta_i p_{i-1},  s_i = w_i + beta_i s_{i-1} (= S~ p_i),  x += alpha_i p_i,  r_{i+1} = r_i - alpha_i s_i
// so an iteration needs ONE grid-wide synchronisation -- one launch -- where the textbook arrangement needs two (p.Sp
// before the update, r.r after): a launch in this chain costs ~4.2 us however little it does.  The
// product of launch i needs r_{i+1} of the neighbouring block rows, which their owners are only writing in that same
// launch: every wavefront recomputes it from the neighbours' r_i, w_i, s_{i-1} and the two scalars (two fma per element),
// which is why r, w and s alternate between two buffers.  gamma_i, delta_i live in HBM as one partial per workgroup and
// are re-summed by every wavefront in the same order, so every workgroup takes the same converged / breakdown decision
// and no kernel ever waits on the host; the decision is sticky through SC_PCG_DONE.
// ------------------------------------------------------------------------------------------
// sum of n partials, every lane of the wavefront gets the total
__device__ __forceinline__ double wave_total(const double *__restrict__ v, int n, int lane)
{
    double s = 0.0;
    for (int i0 = lane; i0 < n; i0 += 256) {          // (four loads in flight: a load and an add per trip is a chain of round trips)
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + 64 * k; t[k] = v[i < n ? i : n - 1]; if (i >= n) t[k] = 0.0; }
        s += (t[0] + t[1]) + (t[2] + t[3]);
    }
    return wave_sum(s);
}

// before iteration minit (PBA's __cg_min_iteration) only a residual at rounding level ends the solve (a block-diagonal system is
// solved exactly by the first step: going on would divide by p'S~p = 0)
constexpr double kPcgFloor = 1e-10;
__device__ __forceinline__ bool pcg_converged(const PcgArgs &a, double rz_it)
{
    const double rz0 = a.rz_hist[0];
    if (!(rz0 > 0.0)) return true;
    return sqrt(fabs(rz_it) / rz0) < (a.it >= a.minit ? a.tol : fmin(a.tol, kPcgFloor));
}

// r_0 = E~ = Linv E, x~_0 = 0 on its own (k_block_inv does this as part of a linear phase; a second solve on the same
// linearisation -- the parity hooks -- starts here).  8 cameras per 64-lane group: lane = (camera, row).
__global__ __launch_bounds__(64) void k_pcg_init(const PcgArgs a)
{
    const int j = blockIdx.x * 8 + (threadIdx.x >> 3), r = threadIdx.x & 7;
    if (j >= a.m) return;
    const double *Li = a.Linv + (size_t)j * 64 + r * 8, *Ej = a.E + (size_t)j * 8;
    double rr = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) rr += Li[c] * Ej[c];
    const int idx = j * 8 + r;
    a.x[idx] = 0.0; a.r0[idx] = rr; a.p[idx] = 0.0; a.s0[idx] = 0.0; a.s1[idx] = 0.0;
}

// the workgroup's two dot-product partials: four wavefronts, fixed order
template <int NW = 4>
__device__ __forceinline__ void publish_partials(const PcgArgs &a, int slot, double g, double d, double (*sh)[2])
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    g = wave_sum(g); d = wave_sum(d);
    if (lane == 0) { sh[wv][0] = g; sh[wv][1] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double gt = 0.0, dt = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w += 4) { gt += (sh[w][0] + sh[w + 1][0]) + (sh[w + 2][0] + sh[w + 3][0]); dt += (sh[w][1] + sh[w + 1][1]) + (sh[w + 2][1] + sh[w + 3][1]); }
        a.gp[(size_t)slot * a.nwg + blockIdx.x] = gt;
        a.dp[(size_t)slot * a.nwg + blockIdx.x] = dt;
    }
}

// products of a block row with NV vectors in one pass over its blocks q = qb, qb + stride, ... < qe.  One wave load = 16 B per
// lane = TWO 8x8 blocks (1 KB contiguous when stride = 1): the lower half-wave takes the even blocks of the row, the upper the
// odd ones; lane (r, c2) of a half multiplies elements (r, c2) and (r, c2 + 1) with the neighbour's vector elements c2, c2 + 1
// (one 16-B load).  Eight loads = 8 KB per wavefront in flight: a launch has only ~m wavefronts, and the bytes in flight set
// the rate at which a matrix that lives in the Infinity Cache or HBM streams (8-B loads, 4 KB in flight: 3.2 TB/s on C4's
// 208 MB; sixteen loads per wavefront instead of eight: slower, C4 2.70 -> 3.91 ms per solve).  The sum over the four lanes of a row by two DPP moves, the two halves by a cross-half read; the result is left
// where the callers expect it: row r's value in all lanes 8r .. 8r + 7.
template <int NV>
__device__ __forceinline__ void row_products(const PcgArgs &a, int qb, int qe, int stride, int lane, const double *const (&vec)[NV], double (&out)[NV], const double *mbox = nullptr)
{
    double acc[NV][2];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v][0] = acc[v][1] = 0.0;
    if (a.S32) {
        // The matrix in single precision (large systems: vectors, sums and the recurrences stay f64).  One wave load = 16 B per lane = FOUR
        // 8 x 8 blocks: the quarter-wave lane >> 4 takes every fourth block of the row; lane (h, c) of a quarter holds rows 4h .. 4h + 3 of
        // column c (k_scale_blocks' layout 1) and multiplies them with ONE 8-byte element of the vector.
        const float *__restrict__ Sf = a.S32;
        const int quarter = lane >> 4, li = lane & 15, cf = li & 7;
        const int st4 = 4 * stride;
        int q = qb + quarter * stride;
        double ac[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v) ac[v][0] = ac[v][1] = ac[v][2] = ac[v][3] = 0.0;
        int kn[8];                                     // column indices, one chunk ahead (see the f64 form below)
        if (q + 7 * st4 < qe) {
#pragma unroll
            for (int u = 0; u < 8; ++u) kn[u] = a.bsr_col[q + u * st4];
        }
        for (; q + 7 * st4 < qe; q += 8 * st4) {
            int k[8];
            float4 sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                k[u] = kn[u] * 8 + cf;
                sv[u] = *reinterpret_cast<const float4 *>(Sf + (size_t)(q + u * st4) * 64 + li * 4);
            }
            if (q + 15 * st4 < qe) {
#pragma unroll
                for (int u = 0; u < 8; ++u) kn[u] = a.bsr_col[q + (8 + u) * st4];
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = vec[v][k[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ac[v][0] = fma((double)sv[u].x, x[u], ac[v][0]); ac[v][1] = fma((double)sv[u].y, x[u], ac[v][1]);
                    ac[v][2] = fma((double)sv[u].z, x[u], ac[v][2]); ac[v][3] = fma((double)sv[u].w, x[u], ac[v][3]);
                }
            }
        }
        for (; q < qe; q += st4) {
            const int k = a.bsr_col[q] * 8 + cf;
            const float4 sv = *reinterpret_cast<const float4 *>(Sf + (size_t)q * 64 + li * 4);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double x = vec[v][k];
                ac[v][0] = fma((double)sv.x, x, ac[v][0]); ac[v][1] = fma((double)sv.y, x, ac[v][1]);
                ac[v][2] = fma((double)sv.z, x, ac[v][2]); ac[v][3] = fma((double)sv.w, x, ac[v][3]);
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                             // over the eight columns: the lanes h * 8 .. h * 8 + 7
                double t = ac[v][i];
                t += dpp_move<0xB1>(t); t += dpp_move<0x4E>(t); t += dpp_move<0x141>(t);
                ac[v][i] = t;
            }
            const int i = cf & 3;                                     // lane (h, c) keeps row 4h + (c & 3)
            double t = i == 0 ? ac[v][0] : i == 1 ? ac[v][1] : i == 2 ? ac[v][2] : ac[v][3];
            t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);           // the four quarters' blocks
            const int rr = lane >> 3;                                 // row r: from lane (r >> 2, r & 3)
            out[v] = __shfl(t, (rr >> 2) * 8 + (rr & 3));
        }
        return;
    }
    if (a.cm) {
        // f64 in the column form (k_scale_blocks' layout 2): the half-wave lane >> 5 takes every second block of the row; lane (h, c) of a half
        // holds rows 2h, 2h + 1 of column c (one 16-byte load) and needs ONE 8-byte element of the vector for them
        const double *__restrict__ Sv = a.S;
        const int half = lane >> 5, l5 = lane & 31, cc = l5 & 7;
        int q = qb + half * stride;
        const int st2 = 2 * stride;
        double ac[NV][2][2];
#pragma unroll
        for (int v = 0; v < NV; ++v) ac[v][0][0] = ac[v][0][1] = ac[v][1][0] = ac[v][1][1] = 0.0;
        if (mbox) {
            // the neighbours' vectors from this row's MAILBOXES (push_row): block q's 3 x 8 doubles sit at 24 q -- no column index between the
            // row pointer and the vector loads, which go out together with the matrix loads (one round of latency less per launch)
            static_assert(NV == 3 || NV == 1, "mailboxes hold r | w | s");
            for (; q + 7 * st2 < qe; q += 8 * st2) {
                double2 sv[8];
                double x[NV][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    sv[u] = *reinterpret_cast<const double2 *>(Sv + (size_t)(q + u * st2) * 64 + l5 * 2);
#pragma unroll
                    for (int v = 0; v < NV; ++v) x[v][u] = mbox[(size_t)(q + u * st2) * 24 + v * 8 + cc];
                }
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int u = 0; u < 8; ++u) { ac[v][0][u & 1] = fma(sv[u].x, x[v][u], ac[v][0][u & 1]); ac[v][1][u & 1] = fma(sv[u].y, x[v][u], ac[v][1][u & 1]); }
            }
            if (q < qe) {                                             // the rest of the row: up to seven blocks, all requested at once
                double2 sv[7];
                double x[NV][7];
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    const bool on = q + u * st2 < qe;
                    const int qq = on ? q + u * st2 : q;
                    sv[u] = *reinterpret_cast<const double2 *>(Sv + (size_t)qq * 64 + l5 * 2);
                    if (!on) sv[u] = make_double2(0.0, 0.0);
#pragma unroll
                    for (int v = 0; v < NV; ++v) x[v][u] = mbox[(size_t)qq * 24 + v * 8 + cc];
                }
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int u = 0; u < 7; ++u) { ac[v][0][u & 1] = fma(sv[u].x, x[v][u], ac[v][0][u & 1]); ac[v][1][u & 1] = fma(sv[u].y, x[v][u], ac[v][1][u & 1]); }
            }
        } else {
        int kn[8];
        if (q + 7 * st2 < qe) {
#pragma unroll
            for (int u = 0; u < 8; ++u) kn[u] = a.bsr_col[q + u * st2];
        }
        for (; q + 7 * st2 < qe; q += 8 * st2) {
            int k[8];
            double2 sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                k[u] = kn[u] * 8 + cc;
                sv[u] = *reinterpret_cast<const double2 *>(Sv + (size_t)(q + u * st2) * 64 + l5 * 2);
            }
            if (q + 15 * st2 < qe) {
#pragma unroll
                for (int u = 0; u < 8; ++u) kn[u] = a.bsr_col[q + (8 + u) * st2];
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = vec[v][k[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) { ac[v][0][u & 1] = fma(sv[u].x, x[u], ac[v][0][u & 1]); ac[v][1][u & 1] = fma(sv[u].y, x[u], ac[v][1][u & 1]); }
            }
        }
        for (; q < qe; q += st2) {
            const int k = a.bsr_col[q] * 8 + cc;
            const double2 sv = *reinterpret_cast<const double2 *>(Sv + (size_t)q * 64 + l5 * 2);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double x = vec[v][k];
                ac[v][0][0] = fma(sv.x, x, ac[v][0][0]); ac[v][1][0] = fma(sv.y, x, ac[v][1][0]);
            }
        }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double t0 = ac[v][0][0] + ac[v][0][1], t1 = ac[v][1][0] + ac[v][1][1];
            t0 += dpp_move<0xB1>(t0); t0 += dpp_move<0x4E>(t0); t0 += dpp_move<0x141>(t0);   // over the eight columns: the lanes h * 8 .. h * 8 + 7
            t1 += dpp_move<0xB1>(t1); t1 += dpp_move<0x4E>(t1); t1 += dpp_move<0x141>(t1);
            double t = (cc & 1) ? t1 : t0;                            // lane (h, c) keeps row 2h + (c & 1)
            t += __shfl_xor(t, 32);                                   // even + odd blocks
            const int rr = lane >> 3;                                 // row r: from lane (r >> 1, r & 1)
            out[v] = __shfl(t, (rr >> 1) * 8 + (rr & 1));
        }
        return;
    }
    const int half = lane >> 5, c2 = (lane & 3) * 2, el = ((lane & 31) >> 2) * 8 + c2;
    const double *__restrict__ Sv = a.S;
    int q = qb + half * stride;                       // this half's first block; then every second block of the row
    const int st2 = 2 * stride;
    // the column indices of a chunk are requested one chunk ahead: the vector loads depend on them, and a row's chunks are otherwise two
    // dependent round trips each (long rows -- C4: 239 blocks -- are chains of such round trips, not a stream)
    int kn[8];
    if (q + 7 * st2 < qe) {                           // (short rows never enter the chunked loop: nothing to request ahead)
#pragma unroll
        for (int u = 0; u < 8; ++u) kn[u] = a.bsr_col[q + u * st2];
    }
    for (; q + 7 * st2 < qe; q += 8 * st2) {          // (the halves may leave the loop one trip apart: nothing wave-wide inside)
        int k[8];
        double2 sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            k[u] = kn[u] * 8 + c2;
            sv[u] = *reinterpret_cast<const double2 *>(Sv + (size_t)(q + u * st2) * 64 + el);
        }
        if (q + 15 * st2 < qe) {                      // (uniform per half) the next trip's indices
#pragma unroll
            for (int u = 0; u < 8; ++u) kn[u] = a.bsr_col[q + (8 + u) * st2];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double2 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const double2 *>(vec[v] + k[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[v][u & 1] = fma(sv[u].y, x[u].y, fma(sv[u].x, x[u].x, acc[v][u & 1]));
        }
    }
    for (; q < qe; q += st2) {
        const int k = a.bsr_col[q] * 8 + c2;
        const double2 sv = *reinterpret_cast<const double2 *>(Sv + (size_t)q * 64 + el);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double2 x = *reinterpret_cast<const double2 *>(vec[v] + k);
            acc[v][0] = fma(sv.y, x.y, fma(sv.x, x.x, acc[v][0]));
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double t = acc[v][0] + acc[v][1];
        t += dpp_move<0xB1>(t); t += dpp_move<0x4E>(t);          // the 4 lanes of a row (quad permutes)
        t += __shfl_xor(t, 32);                                   // even + odd blocks
        out[v] = __shfl(t, (lane >> 3) * 4);                      // row r: from lanes 4r .. 4r + 3 to lanes 8r .. 8r + 7
    }
}

// Push halo (PcgArgs.mb0 / mb1): a block row leaves its new r | w | s (3 x 8 doubles) in the mailbox of every block row that multiplies with
// them -- position bsr_tq[q] of its own blocks q (the pattern is symmetric) -- so that the readers' vector loads need no column index.
// The row's `nparts` wavefronts take its blocks in turn; lane = (block of eight, element e), the element's value sits in lane 8 e.
__device__ __forceinline__ void push_row(const PcgArgs &a, double *__restrict__ mb, int q0, int q1, int part, int nparts, int lane, int tq_first, double rv, double wv, double sv)
{
    const int e = lane & 7;
    const double re = __shfl(rv, e * 8), we = __shfl(wv, e * 8), se = __shfl(sv, e * 8);
    int q = q0 + part * 8 + (lane >> 3), tq = tq_first;
    while (q < q1) {
        double *dst = mb + (size_t)tq * 24 + e;
        dst[0] = re; dst[8] = we; dst[16] = se;
        q += 8 * nparts;
        if (q < q1) tq = a.bsr_tq[q];
    }
}
// (the first mailbox position of a lane, requested early: with the row pointers)
__device__ __forceinline__ int push_first(const PcgArgs &a, int q0, int q1, int part, int lane)
{
    const int q = q0 + part * 8 + (lane >> 3);
    return q < q1 ? a.bsr_tq[q] : 0;
}

// ROWS = 4: four block rows per workgroup, WPR wavefronts each: 2 while the rows are short (C3, 37 blocks a row: the launch is
// rounds of latency, and half a row per wavefront shortens the longest one: 0.31 -> 0.25 ms per solve; four: no better), 1 for long
// rows (C4, 239 blocks a row: bound by the rate the 208 MB stream in; two: 2.68 -> 2.90 ms).  ROWS = 1 (few cameras: fewer block rows than the chip has CUs): the four
// wavefronts of a workgroup share one row.  Wavefronts that share a row take its blocks in turn and add up through LDS in a
// fixed order -- the result is in the row's first wavefront.  Returns whether this wavefront holds a row's finished products.
template <int ROWS, int NV, int WPR = 1>
__device__ __forceinline__ bool workgroup_row_products(const PcgArgs &a, int q0, int q1, const double *const (&vec)[NV], double (&out)[NV], double (*red)[NV][8], const double *mbox = nullptr)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (ROWS == 4 && WPR == 1) { row_products<NV>(a, q0, q1, 1, lane, vec, out, mbox); return true; }
    constexpr int SH = ROWS == 4 ? WPR : 4;         // wavefronts per row
    const int part = wv % SH, first = wv - part;
    row_products<NV>(a, q0 + part, q1, SH, lane, vec, out, mbox);
    if ((lane & 7) == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) red[wv][v][lane >> 3] = out[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if constexpr (SH == 2) out[v] = red[first][v][lane >> 3] + red[first + 1][v][lane >> 3];
        else out[v] = (red[first][v][lane >> 3] + red[first + 1][v][lane >> 3]) + (red[first + 2][v][lane >> 3] + red[first + 3][v][lane >> 3]);
    }
    return part == 0;
}

// w_0 = S~ r_0 and the partials of gamma_0, delta_0
template <int ROWS, int WPR = 1>
__global__ __launch_bounds__(256 * WPR) void k_pcg_start(const PcgArgs a)
{
    __shared__ double sh[4 * WPR][2];
    __shared__ double red[4 * WPR][1][8];
    const int lane = threadIdx.x & 63, j = ROWS == 4 ? blockIdx.x * 4 + (threadIdx.x >> 6) / WPR : blockIdx.x, r = lane >> 3, c = lane & 7;
    const bool live = j < a.m;
    const int jj = live ? j : 0;
    const int q0 = a.ell_w ? jj * a.ell_w : a.bsr_row[jj], q1 = !live ? q0 : a.ell_w ? q0 + a.ell_w : a.bsr_row[jj + 1];
    const double *const vec[1] = { a.r0 };
    double w[1];
    constexpr int SH = ROWS == 4 ? WPR : 4;
    const int part = (threadIdx.x >> 6) % SH;
    const int tq_first = a.mb0 ? push_first(a, q0, q1, part, lane) : 0;
    const bool mine = workgroup_row_products<ROWS, 1, WPR>(a, q0, q1, vec, w, red);      // (by column index: the mailboxes are filled below)
    double g = 0.0, d = 0.0;
    const double rr = a.r0[jj * 8 + r];
    if (live && mine && c == 0) {
        a.w0[j * 8 + r] = w[0];
        g = rr * rr; d = w[0] * rr;
    }
    if (a.mb0 && live) push_row(a, a.mb0, q0, q1, part, SH, lane, tq_first, rr, w[0], 0.0);   // what iteration 0 reads: r_0 | w_0 | s_-1 = 0
    publish_partials<4 * WPR>(a, 0, g, d, sh);
}

// ------------------------------------------------------------------------------------------
// Two-level preconditioner (kernels.h CoarseArgs): block-Jacobi -- the scaling S~ = L^-1 S L^-T -- plus an additive coarse
// correction on ng aggregates of consecutive cameras (8 coarse unknowns each: the camera parameters summed over the aggregate),
//   M~^-1 = I + Z~ Ac^-1 Z~',   Z~ = L' Z,   Ac = Z' S Z = sums of the unscaled blocks of S over pairs of aggregates.
// The slow modes of the reduced system are smooth along the camera path (profiles/r02_pcg_two_level_proto.txt: 3-6x fewer
// iterations once the damping is small).  It costs NO second grid-wide synchronisation per iteration: everything a launch
// needs about the coarse space is linear in vectors the previous launch left behind -- c_{i+1} = Z~'r_{i+1} = cr - alpha (cw + beta cs)
// from per-workgroup partial sums, y = Ac^-1 c by every workgroup for itself (nc <= 128), z = r + Z~ y on its own rows, and
// S~ z = S~ r + (S~ Z~) y with the band S~ Z~ = L^-1 (S Z) formed once per trial.
// Per trial: k_coarse_bc (per-row sums), k_coarse_assemble (fixed-order sums: the replicated solve must give the same bits on every rank), k_coarse_invert
// (Gauss-Jordan by one workgroup, the matrix in registers), after k_coarse_bc's per-row sums.
// ------------------------------------------------------------------------------------------
// Bc_j[slot] = L_j^-1 sum_{k in aggregate bc_agg[j][slot]} S_jk (= (S~ Z~)_j restricted to that aggregate) and
// U_j[slot] = L_j Bc_j[slot] (the plain sum of the row's blocks of S in that aggregate: what Z'SZ is made of).  Taken from the
// SCALED matrix the CG streams anyway, L_j^-1 S_jk = S~_jk L_k', in one pass shaped like the CG product (16 B per lane, two
// blocks per wave load, eight loads in flight; lane (r, c2) of a half holds S~[r][c2], S~[r][c2 + 1] and forms its share of
// all eight columns): one wavefront per block row.
__global__ __launch_bounds__(256) void k_coarse_bc(int m, const int *__restrict__ bsr_row, const int *__restrict__ bsr_col, const double *__restrict__ Sv,
                                                  const CoarseArgs co)
{
    // one workgroup per block row, its four wavefronts take the row's block pairs in turn (a wavefront alone walks a row of 240
    // blocks in rounds of dependent latency: blocks from HBM, then the neighbours' L from L2); then wavefront s finishes slot s
    __shared__ double sAcc[4][kBcSlots][64], sL[64], sB[kBcSlots][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.x;
    const int cams_per_agg = co.wg_per_agg * 4;
    const int half = lane >> 5, c2 = (lane & 3) * 2, r = (lane & 31) >> 2, el = r * 8 + c2;
    int H[kBcSlots];
    double acc[kBcSlots][8];
#pragma unroll
    for (int sl = 0; sl < kBcSlots; ++sl) {
        H[sl] = co.bc_agg[j * kBcSlots + sl];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[sl][c] = 0.0;
    }
    if (wv == 0) sL[lane] = co.Lfac[(size_t)j * 64 + lane];
    const int qe = bsr_row[j + 1];
    int q = bsr_row[j] + 2 * wv + half;
    constexpr int UN = 4;
    auto one = [&](int col, double2 sv) {
        const int g = col / cams_per_agg;
        const double *Lk = co.Lfac + (size_t)col * 64 + c2;
        double t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { const double2 l = *reinterpret_cast<const double2 *>(Lk + c * 8); t[c] = fma(sv.y, l.y, sv.x * l.x); }
#pragma unroll
        for (int sl = 0; sl < kBcSlots; ++sl) {
            const double on = g == H[sl] ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[sl][c] = fma(on, t[c], acc[sl][c]);
        }
    };
    for (; q + (UN - 1) * 8 < qe; q += UN * 8) {
        int col[UN];
        double2 sv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { col[u] = bsr_col[q + 8 * u]; sv[u] = *reinterpret_cast<const double2 *>(Sv + (size_t)(q + 8 * u) * 64 + el); }
#pragma unroll
        for (int u = 0; u < UN; ++u) one(col[u], sv[u]);
    }
    for (; q < qe; q += 8) one(bsr_col[q], *reinterpret_cast<const double2 *>(Sv + (size_t)q * 64 + el));
#pragma unroll
    for (int sl = 0; sl < kBcSlots; ++sl)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double t = acc[sl][c];
            t += dpp_move<0xB1>(t); t += dpp_move<0x4E>(t);      // the four lanes of row r
            t += __shfl_xor(t, 32);                               // even + odd block pairs
            if (half == 0 && (lane & 3) == 0) sAcc[wv][sl][r * 8 + c] = t;
        }
    __syncthreads();
    const int sl = wv, rr = lane >> 3, cc = lane & 7;             // (kBcSlots == 4 wavefronts)
    const double bcv = (sAcc[0][sl][lane] + sAcc[1][sl][lane]) + (sAcc[2][sl][lane] + sAcc[3][sl][lane]);
    sB[sl][lane] = bcv;
    co.Bc[((size_t)j * kBcSlots + sl) * 64 + lane] = bcv;
    wave_lds_sync();
    double u = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) u = fma(sL[rr * 8 + k], sB[sl][k * 8 + cc], u);    // U = L_j Bc
    co.U[((size_t)j * kBcSlots + sl) * 64 + lane] = u;
}

// Ac[G][H] = sum over the rows j of aggregate G of U_j[slot of H], in a fixed order (the replicated solve must give the same
// bits on every rank): one workgroup per coarse block, four wavefronts stride the rows, LDS sums them in order
__global__ __launch_bounds__(256) void k_coarse_assemble(int m, const CoarseArgs co)
{
    __shared__ double part[4][64];
    const int cb = blockIdx.x, G = cb / co.ng, Hc = cb - G * co.ng, wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
    const int cams_per_agg = co.wg_per_agg * 4, j0 = G * cams_per_agg, j1 = min(m, j0 + cams_per_agg);
    double acc = 0.0;
    for (int j = j0 + wv; j < j1; j += 4) {
        const int4 hs = *reinterpret_cast<const int4 *>(co.bc_agg + (size_t)j * kBcSlots);
        const int s = hs.x == Hc ? 0 : hs.y == Hc ? 1 : hs.z == Hc ? 2 : hs.w == Hc ? 3 : -1;
        if (s >= 0) acc += co.U[((size_t)j * kBcSlots + s) * 64 + lane];
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0) co.Ac[(size_t)(G * 8 + r) * co.nc + Hc * 8 + c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// Ac -> Ac^-1 (symmetric positive definite: Gauss-Jordan without pivoting), one workgroup, the matrix in registers (a 4x4
// tile per thread), row k and column k of each step handed round through LDS.  The matrix is first scaled to a unit diagonal
// (pivots <= 1), which lets ONE uniform rank-1 update a_ij -= c_i r_j per step do the pivot row, the pivot column and the
// pivot itself as well, by handing round r_k = p + 1 and c_k = 1 - 1/p in place of the pivot:
//   (k,k): p - (1 - 1/p)(p + 1) = 1/p,   (k,j): r_j - (1 - 1/p) r_j = r_j / p,   (i,k): a_ik - (a_ik / p)(p + 1) = -a_ik / p
// (with p <= 1 each of them keeps full relative accuracy) -- no per-element selects in the inner loop: 208 -> 40 us at 128 x 128.
// A pivot that is not positive switches the coarse correction off for this trial (Acinv = 0: plain block-Jacobi CG).
__global__ __launch_bounds__(1024) void k_coarse_invert(const CoarseArgs co)
{
    static_assert(kCoarseMax * 8 == 128, "the register tiling covers 128 x 128");
    __shared__ __attribute__((aligned(16))) double sRow[2][128], sCol[2][128];
    __shared__ double sPiv[2], sD[128];                                  // sPiv: 1 / pivot, by the thread that holds it (NaN: not positive)
    // a wavefront holds 8 x 8 tiles (32 x 32 elements): row k and column k of a step live in 4 + 4 of the 16 wavefronts, the others
    // skip the hand-over code
    const int n = co.nc, tid = threadIdx.x, wq = tid >> 6, lq = tid & 63, i0 = ((wq >> 2) * 8 + (lq >> 3)) * 4, j0 = ((wq & 3) * 8 + (lq & 7)) * 4;
    if (tid < 128) { const double d = tid < n ? co.Ac[(size_t)tid * n + tid] : 1.0; sD[tid] = d > 0.0 ? 1.0 / sqrt(d) : 0.0; }
    __syncthreads();
    double a[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int i = i0 + x, j = j0 + y;
            a[x][y] = i < n && j < n ? (i == j ? 1.0 : 0.5 * (co.Ac[(size_t)i * n + j] + co.Ac[(size_t)j * n + i]) * sD[i] * sD[j]) : 0.0;
        }
    bool fail = false;
    for (int k = 0; k < n; ++k) {
        const int b = k & 1, kr = k - i0, kc = k - j0;
        if ((unsigned)kr < 4u) {              // this thread's tile holds part of row k
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (x == kr) {
#pragma unroll
                    for (int y = 0; y < 4; ++y) sRow[b][j0 + y] = y == kc ? a[x][y] + 1.0 : a[x][y];
                    if ((unsigned)kc < 4u) {
#pragma unroll
                        for (int y = 0; y < 4; ++y) if (y == kc) sPiv[b] = (a[x][y] > 0.0 && isfinite(a[x][y])) ? 1.0 / a[x][y] : __builtin_nan("");
                    }
                }
        }
        if ((unsigned)kc < 4u) {              // ... of column k
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (y == kc) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) sCol[b][i0 + x] = x == kr ? a[x][y] - 1.0 : a[x][y];
                }
        }
        __syncthreads();                                                  // (one barrier a step: the buffers alternate)
        const double ip = sPiv[b];
        if (!(ip == ip)) { fail = true; break; }                          // uniform: every thread reads the same value
        const double2 r01 = *reinterpret_cast<const double2 *>(&sRow[b][j0]), r23 = *reinterpret_cast<const double2 *>(&sRow[b][j0 + 2]);
        const double2 c01 = *reinterpret_cast<const double2 *>(&sCol[b][i0]), c23 = *reinterpret_cast<const double2 *>(&sCol[b][i0 + 2]);
        const double rw[4] = { r01.x, r01.y, r23.x, r23.y }, cl[4] = { c01.x * ip, c01.y * ip, c23.x * ip, c23.y * ip };
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) a[x][y] = fma(-cl[x], rw[y], a[x][y]);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
            if (i0 + x < n && j0 + y < n) co.Acinv[(size_t)(i0 + x) * n + j0 + y] = fail ? 0.0 : a[x][y] * sD[i0 + x] * sD[j0 + y];
}

// The coarse part of one launch, by NT threads (tid = 0 .. NT - 1: the whole workgroup of k_pcg2_start, or the one wavefront
// k_pcg2_iter sets aside for it while the other four stream the matrix): coarse projections of up to three vectors from the
// per-workgroup partial sums (cpart[par]), combined as cr - alpha (cw + beta cs); y = Ac^-1 c for the aggregates this
// workgroup's rows reach (its own and the band's: a handful of the ng), left in sY[aggregate * 8 + r].  The caller puts a
// barrier between this and the readers of sY.
template <int NT>
__device__ __forceinline__ void coarse_solve(const PcgArgs &a, int tid, int par, int nvec, double alpha, double beta, const double *E0,
                                             double *sC /* [3][128] */, double *sY /* [128] */, int *sList /* [kCoarseMax + 1] */)
{
    const CoarseArgs &co = a.co;
    const int nc = co.nc;
    auto sync = [] { if constexpr (NT == 64) wave_lds_sync(); else __syncthreads(); };
    if (E0) {   // the first residual: c_0 = aggregate sums of the unscaled right-hand side
        for (int o = tid; o < nc; o += NT) {
            const int G = o >> 3, rr = o & 7, j0 = G * co.wg_per_agg * 4, j1 = min(a.m, j0 + co.wg_per_agg * 4);
            double s0 = 0.0, s1 = 0.0;
            int k = j0;
            for (; k + 1 < j1; k += 2) { s0 += E0[(size_t)k * 8 + rr]; s1 += E0[(size_t)(k + 1) * 8 + rr]; }
            if (k < j1) s0 += E0[(size_t)k * 8 + rr];
            sC[o] = s0 + s1;
        }
    } else {
        const double *cp = co.cpart + (size_t)par * 3 * a.nwg * 8;
        for (int o = tid; o < nvec * nc; o += NT) {
            const int v = o / nc, idx = o - v * nc, G = idx >> 3, rr = idx & 7;
            const int w0 = G * co.wg_per_agg, w1 = min(a.nwg, w0 + co.wg_per_agg);
            double s0 = 0.0, s1 = 0.0;
            int w = w0;
            for (; w + 1 < w1; w += 2) { s0 += cp[((size_t)v * a.nwg + w) * 8 + rr]; s1 += cp[((size_t)v * a.nwg + w + 1) * 8 + rr]; }
            if (w < w1) s0 += cp[((size_t)v * a.nwg + w) * 8 + rr];
            sC[v * 128 + idx] = s0 + s1;
        }
    }
    if (tid == 0) {   // the aggregates this workgroup needs y for: a short list (own first)
        int n = 0;
        const int *bc = co.bc_agg + (size_t)blockIdx.x * 4 * kBcSlots;
        const int rows = min(4, a.m - (int)blockIdx.x * 4);
        sList[1 + n++] = (int)blockIdx.x / co.wg_per_agg;
        for (int t = 0; t < rows * kBcSlots; ++t) {
            const int H = bc[t];
            bool have = H < 0;
            for (int u = 0; u < n && !have; ++u) have = sList[1 + u] == H;
            if (!have) sList[1 + n++] = H;
        }
        sList[0] = n;
    }
    sync();
    if (!E0 && nvec == 3)
        for (int o = tid; o < nc; o += NT) sC[o] = sC[o] - alpha * (sC[128 + o] + beta * sC[256 + o]);   // (only this thread reads sC[.][o] here)
    sync();
    // y_o = sum_k Acinv[o][k] c_k for the listed aggregates: 8 lanes per output (k = part, part + 8, ...: independent loads), NT / 8 outputs a pass
    const int nout = sList[0] * 8, part = tid & 7;
    for (int oi = tid >> 3; oi < ((nout + NT / 8 - 1) & ~(NT / 8 - 1)); oi += NT / 8) {    // (uniform trip count: the DPP sums below need whole rows of lanes)
        const bool on = oi < nout;
        const int o = on ? sList[1 + (oi >> 3)] * 8 + (oi & 7) : 0;
        double acc = 0.0;
        if (on)
            for (int k = part; k < nc; k += 8) acc = fma(co.Acinv[(size_t)k * nc + o], sC[k], acc);   // symmetric: column o instead of row o
        acc += dpp_move<0xB1>(acc); acc += dpp_move<0x4E>(acc); acc += dpp_move<0x141>(acc);
        if (on && part == 0) sY[o] = acc;
    }
}

// per block row: z_j = r_j + L_j' y_G, the band term (S~ Z~)_j y, and the coarse projections L_j (r | w | s)_j of the new vectors;
// lane (rr, cc), values of the 8 new components live in sV[wave][vec][8]
template <int ROWS>
__global__ __launch_bounds__(256) void k_pcg2_start(const PcgArgs a)
{
    __shared__ double sh[4][2], red[4][1][8], sC[3 * 128], sY[128], sV[4][3][8], sP[4][3][8];
    __shared__ int sList[kCoarseMax + 1];
    const CoarseArgs &co = a.co;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.x * 4 + wv, r = lane >> 3, c = lane & 7;
    const bool live = j < a.m;
    const int jj = live ? j : 0;
    const int q0 = a.bsr_row[jj], q1 = live ? a.bsr_row[jj + 1] : q0;
    // c_0 = Z~' r_0 = aggregate sums of the unscaled right-hand side E (r_0 = L^-1 E): straight from E, every workgroup for itself
    coarse_solve<256>(a, threadIdx.x, 0, 1, 0.0, 0.0, a.E, sC, sY, sList);
    const double *const vec[1] = { a.r0 };
    double w[1];
    (void)workgroup_row_products<4, 1>(a, q0, q1, vec, w, red);      // (S~ r_0)_j[r] in every lane of row r
    __syncthreads();                                                  // sY
    const int G = blockIdx.x / co.wg_per_agg;
    const double *Lj = co.Lfac + (size_t)jj * 64;
    // z = r + L' y_G ; band term
    double zt = Lj[c * 8 + r] * sY[G * 8 + c];
    double bt = 0.0;
#pragma unroll
    for (int sl = 0; sl < kBcSlots; ++sl) {
        const int H = co.bc_agg[jj * kBcSlots + sl];
        if (H >= 0) bt = fma(co.Bc[((size_t)jj * kBcSlots + sl) * 64 + lane], sY[H * 8 + c], bt);
    }
    zt += dpp_move<0xB1>(zt); zt += dpp_move<0x4E>(zt); zt += dpp_move<0x141>(zt);
    bt += dpp_move<0xB1>(bt); bt += dpp_move<0x4E>(bt); bt += dpp_move<0x141>(bt);
    double g = 0.0, d = 0.0;
    const double rr = live ? a.r0[jj * 8 + r] : 0.0;
    const double z_new = rr + zt, w_new = w[0] + bt;
    if (c == 0) { sV[wv][0][r] = live ? rr : 0.0; sV[wv][1][r] = live ? w_new : 0.0; sV[wv][2][r] = 0.0; }
    if (live && c == 0) {
        a.w0[j * 8 + r] = w_new; co.z[j * 8 + r] = z_new;
        g = rr * z_new; d = w_new * z_new;
    }
    wave_lds_sync();
    // coarse projections of the new r, w (s = 0) for the first iteration: (L_j v)_r, then summed over the workgroup's rows
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        double t = live ? Lj[r * 8 + c] * sV[wv][v][c] : 0.0;
        t += dpp_move<0xB1>(t); t += dpp_move<0x4E>(t); t += dpp_move<0x141>(t);
        if (c == 0) sP[wv][v][r] = t;
    }
    publish_partials(a, 0, g, d, sh);                                 // (its barrier orders sP)
    if (threadIdx.x < 24) {
        const int v = threadIdx.x >> 3, rr8 = threadIdx.x & 7;
        co.cpart[((size_t)0 * 3 + v) * a.nwg * 8 + (size_t)blockIdx.x * 8 + rr8] = (sP[0][v][rr8] + sP[1][v][rr8]) + (sP[2][v][rr8] + sP[3][v][rr8]);
    }
}

// iteration a.it with the coarse correction: FIVE wavefronts -- four stream their block rows of S~ as in k_pcg_iter, the fifth
// meanwhile sums the partials, forms alpha and beta and does the coarse solve (rounds of dependent loads that, done by the same
// wavefronts before or after the pass over the matrix, cost an iteration a third more: C4 63.6 against 46 us)
__global__ __launch_bounds__(320) void k_pcg2_iter(const PcgArgs a)
{
    __shared__ double sh[5][2], red[4][3][8], sC[3 * 128], sY[128], sV[4][3][8], sP[4][3][8];
    __shared__ int sList[kCoarseMax + 1];
    const CoarseArgs &co = a.co;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
    const bool cw = wv == 4;                                            // the coarse wavefront
    const int j = blockIdx.x * 4 + (cw ? 0 : wv);
    const int par = a.it & 1, prev = a.it ? a.it - 1 : 0;
    const bool live = !cw && j < a.m;
    const int jj = live ? j : 0, idx = jj * 8 + r;
    const double *__restrict__ rc = par ? a.r1 : a.r0, *__restrict__ wc = par ? a.w1 : a.w0, *__restrict__ so = par ? a.s1 : a.s0;
    const double st = a.scal[SC_PCG_STATUS], dn = a.scal[SC_PCG_DONE];
    double gs = 0.0, ds = 0.0;
    for (int i = lane; i < a.nwg; i += 64) { gs += a.gp[(size_t)par * a.nwg + i]; ds += a.dp[(size_t)par * a.nwg + i]; }
    const double g_prev = a.rz_hist[prev], g_zero = a.rz_hist[0], a_prev = a.alpha_hist[prev];
    const int q0 = a.bsr_row[jj], q1 = live ? a.bsr_row[jj + 1] : q0;
    const double r_own = rc[idx], w_own = wc[idx], s_own = so[idx], p_own = a.p[idx], x_own = a.x[idx], z_own = co.z[idx];
    if (st != 0.0 || dn != 0.0) return;
    const double *const vec[3] = { rc, wc, so };
    double S3[3] = { 0.0, 0.0, 0.0 };
    if (!cw) (void)workgroup_row_products<4, 3>(a, q0, q1, vec, S3, red);
    // every wavefront takes the same decisions from the same sums (the coarse one before its solve, the others after their rows)
    const double gamma = wave_sum(gs), delta = wave_sum(ds);           // gamma = r.z, delta = w.z (w = S~ z)
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    if (first) a.rz_hist[a.it] = gamma;
    const bool conv = a.it == 0 ? !(gamma > 0.0) : (!(g_zero > 0.0) || sqrt(fabs(gamma) / g_zero) < (a.it >= a.minit ? a.tol : fmin(a.tol, kPcgFloor)));
    if (conv) {
        if (first) a.scal[SC_PCG_DONE] = 1.0;
        return;
    }
    const double beta = a.it ? gamma / g_prev : 0.0;
    const double den = a.it ? delta - beta * gamma / a_prev : delta;
    if (!(den > 0.0) || !isfinite(den)) {
        if (first) a.scal[SC_PCG_STATUS] = 2.0;
        return;
    }
    const double alpha = gamma / den;
    // the coarse state of the NEXT residual, from what the last launch left: y = Ac^-1 (cr - alpha (cw + beta cs))
    if (cw) coarse_solve<64>(a, lane, par, 3, alpha, beta, nullptr, sC, sY, sList);
    __syncthreads();
    const int G = blockIdx.x / co.wg_per_agg;
    const double *Lj = co.Lfac + (size_t)jj * 64;
    double zt = Lj[c * 8 + r] * sY[G * 8 + c];
    double bt = 0.0;
#pragma unroll
    for (int sl = 0; sl < kBcSlots; ++sl) {
        const int H = co.bc_agg[jj * kBcSlots + sl];
        if (H >= 0) bt = fma(co.Bc[((size_t)jj * kBcSlots + sl) * 64 + lane], sY[H * 8 + c], bt);
    }
    zt += dpp_move<0xB1>(zt); zt += dpp_move<0x4E>(zt); zt += dpp_move<0x141>(zt);
    bt += dpp_move<0xB1>(bt); bt += dpp_move<0x4E>(bt); bt += dpp_move<0x141>(bt);
    double *rn = par ? a.r0 : a.r1, *wn = par ? a.w0 : a.w1, *sn = par ? a.s0 : a.s1;
    const double s_new = fma(beta, s_own, w_own), r_new = fma(-alpha, s_new, r_own), p_new = fma(beta, p_own, z_own);
    const double z_new = r_new + zt;
    const double w_new = fma(-alpha, fma(beta, S3[2], S3[1]), S3[0]) + bt;
    double g = 0.0, d = 0.0;
    if (!cw && c == 0) { sV[wv][0][r] = live ? r_new : 0.0; sV[wv][1][r] = live ? w_new : 0.0; sV[wv][2][r] = live ? s_new : 0.0; }
    if (live && c == 0) {
        sn[idx] = s_new; rn[idx] = r_new; wn[idx] = w_new; co.z[idx] = z_new;
        a.p[idx] = p_new;
        a.x[idx] = fma(alpha, p_new, x_own);
        g = r_new * z_new; d = w_new * z_new;
    }
    wave_lds_sync();
    if (!cw) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            double t = live ? Lj[r * 8 + c] * sV[wv][v][c] : 0.0;
            t += dpp_move<0xB1>(t); t += dpp_move<0x4E>(t); t += dpp_move<0x141>(t);
            if (c == 0) sP[wv][v][r] = t;
        }
    }
    publish_partials(a, par ^ 1, g, d, sh);                            // (the fifth wavefront adds zeros to a row nobody sums)
    if (threadIdx.x < 24) {
        const int v = threadIdx.x >> 3, rr8 = threadIdx.x & 7;
        co.cpart[((size_t)(par ^ 1) * 3 + v) * a.nwg * 8 + (size_t)blockIdx.x * 8 + rr8] = (sP[0][v][rr8] + sP[1][v][rr8]) + (sP[2][v][rr8] + sP[3][v][rr8]);
    }
    if (first) { a.alpha_hist[a.it] = alpha; a.scal[SC_PCG_ITERS] = (double)(a.it + 1); }
}

// rz_hist[it] = gamma_it, the sticky converged flag, and SC_PCG_FINAL when the solve is over (converged, broken down, or
// a.last: no iteration will follow); one wavefront, after a run of iterations.
__device__ __forceinline__ void pcg_tail(const PcgArgs &a)
{
    bool over = a.scal[SC_PCG_STATUS] != 0.0 || a.scal[SC_PCG_DONE] != 0.0;
    if (!over) {
        const double g = wave_total(a.gp + (size_t)(a.it & 1) * a.nwg, a.nwg, threadIdx.x);
        if (threadIdx.x == 0) a.rz_hist[a.it] = g;
        if (a.it == 0 ? !(g > 0.0) : pcg_converged(a, g)) {
            if (threadIdx.x == 0) a.scal[SC_PCG_DONE] = 1.0;
            over = true;
        }
    }
    if (threadIdx.x == 0 && (over || a.last)) a.scal[SC_PCG_FINAL] = 1.0;
}

__global__ __launch_bounds__(64) void k_pcg_tail(const PcgArgs a) { pcg_tail(a); }

// iteration a.it.  The product w_{it+1} = S~ r_{it+1} needs r_{it+1} = r_it - alpha (w_it + beta s_{it-1}) of the neighbouring block
// rows, which their owners are only writing in this same launch, and alpha, beta, which every wavefront first has to sum up:
// so the pass over the row's blocks forms S~ r_it, S~ w_it and S~ s_{it-1} separately -- its loads depend on nothing but
// the row pointers and are in flight while the partials are summed -- and the scalars combine the three at the end.
// What this launch costs is rounds of dependent memory latency (~4.2 us of it is the launch itself), so everything that
// can be requested at once is: flags, partials, histories and row pointers together, then blocks and column indices, then
// the neighbours' vectors.
// the two dot products of the launch before, summed ONCE (fixed order) for everybody: with thousands of workgroups each adding up
// all nwg partials itself, an iteration spent more time on 2 nwg^2 partial loads than on the matrix (10 000 cameras: 122 us, 42 of
// them the matrix).  One small launch between two iterations; used from kPcgTotalsFrom workgroups on (engine.cpp).
__global__ __launch_bounds__(256) void k_pcg_totals(const PcgArgs a, int slot)
{
    __shared__ double sg[256], sd[256];
    double g = 0.0, d = 0.0;
    for (int i = threadIdx.x; i < a.nwg; i += 256) { g += a.gp[(size_t)slot * a.nwg + i]; d += a.dp[(size_t)slot * a.nwg + i]; }
    sg[threadIdx.x] = g; sd[threadIdx.x] = d;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { sg[threadIdx.x] += sg[threadIdx.x + w]; sd[threadIdx.x] += sd[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.tot[slot * 2] = sg[0]; a.tot[slot * 2 + 1] = sd[0]; }
}

#ifdef MIBA_STAMPS
#define PSTAMP(i) do { if (a.stamps && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) a.stamps[(size_t)a.it * 16 + (blockIdx.x ? 8 : 0) + (i)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define PSTAMP(i)
#endif
template <int ROWS, int WPR = 1>
__global__ __launch_bounds__(256 * WPR) void k_pcg_iter(const PcgArgs a)
{
    PSTAMP(0);
    __shared__ double sh[4 * WPR][2];
    __shared__ double red[4 * WPR][3][8];
    constexpr bool kOwn = ROWS == 4 && WPR == 1;     // every wavefront works for itself: no barrier, no shared totals
    const int lane = threadIdx.x & 63, j = ROWS == 4 ? blockIdx.x * 4 + (threadIdx.x >> 6) / WPR : blockIdx.x, r = lane >> 3, c = lane & 7;
    const int par = a.it & 1, prev = a.it ? a.it - 1 : 0;
    const bool live = j < a.m;
    const int jj = live ? j : 0, idx = jj * 8 + r;
    const double *__restrict__ rc = par ? a.r1 : a.r0, *__restrict__ wc = par ? a.w1 : a.w0, *__restrict__ so = par ? a.s1 : a.s0;
    constexpr int SH = ROWS == 4 ? WPR : 4;
    const int part = (threadIdx.x >> 6) % SH;
    const double *const vec[3] = { rc, wc, so };
    // round 1
    const double st = a.scal[SC_PCG_STATUS], dn = a.scal[SC_PCG_DONE];
    // the two dot products of the last launch, one partial per workgroup: with a wavefront per block row every wavefront adds
    // them up itself; with a workgroup per row (as many partials as cameras) the first wavefront does and hands the totals to
    // the other three through LDS -- four times fewer partial loads, which at 1 700 rows were as many bytes as a fifth of S~
    __shared__ double sgd[2];
    double gs = 0.0, ds = 0.0;
    if (a.tot) { if (lane == 0) { gs = a.tot[par * 2]; ds = a.tot[par * 2 + 1]; } }      // (the wave sums below add zeros to it)
    else if (kOwn || threadIdx.x < 64)
        // (four partials per lane requested at once: a loop of one load and one add per trip was a chain of nwg / 64 round trips to another XCD's data)
        for (int i0 = lane; i0 < a.nwg; i0 += 256) {
            double gv[4], dv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = i0 + 64 * t, ic = i < a.nwg ? i : a.nwg - 1;
                gv[t] = a.gp[(size_t)par * a.nwg + ic]; dv[t] = a.dp[(size_t)par * a.nwg + ic];
                if (i >= a.nwg) { gv[t] = 0.0; dv[t] = 0.0; }
            }
            gs += (gv[0] + gv[1]) + (gv[2] + gv[3]); ds += (dv[0] + dv[1]) + (dv[2] + dv[3]);
        }
    const double g_prev = a.rz_hist[prev], g_zero = a.rz_hist[0], a_prev = a.alpha_hist[prev];
    const int q0 = a.ell_w ? jj * a.ell_w : a.bsr_row[jj], q1 = !live ? q0 : a.ell_w ? q0 + a.ell_w : a.bsr_row[jj + 1];   // (fixed pitch: nothing to wait for)
    const double r_own = rc[idx], w_own = wc[idx], s_own = so[idx], p_own = a.p[idx], x_own = a.x[idx];
    const int tq_first = a.mb0 ? push_first(a, q0, q1, part, lane) : 0;
    if (st != 0.0 || dn != 0.0) return;
    PSTAMP(1);
    // rounds 2 and 3
    double S3[3];
    if (!kOwn && threadIdx.x < 64) {                 // (the barrier inside workgroup_row_products orders this against the reads below)
        const double gsum = wave_sum(gs), dsum = wave_sum(ds);
        if (lane == 0) { sgd[0] = gsum; sgd[1] = dsum; }
    }
    const bool mine = workgroup_row_products<ROWS, 3, WPR>(a, q0, q1, vec, S3, red, a.mb0 ? (par ? a.mb1 : a.mb0) : nullptr);
    PSTAMP(2);
    const double gamma = kOwn ? wave_sum(gs) : sgd[0], delta = kOwn ? wave_sum(ds) : sgd[1];
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    if (first) a.rz_hist[a.it] = gamma;
    const bool conv = a.it == 0 ? !(gamma > 0.0) : (!(g_zero > 0.0) || sqrt(fabs(gamma) / g_zero) < (a.it >= a.minit ? a.tol : fmin(a.tol, kPcgFloor)));
    if (conv) {
        if (first) a.scal[SC_PCG_DONE] = 1.0;
        return;
    }
    const double beta = a.it ? gamma / g_prev : 0.0;
    const double den = a.it ? delta - beta * gamma / a_prev : delta;   // = p.S~p
    if (!(den > 0.0) || !isfinite(den)) { // S not positive definite along p: "not solved" (dpotrf info > 0)
        if (first) a.scal[SC_PCG_STATUS] = 2.0;
        return;
    }
    const double alpha = gamma / den;
    double *rn = par ? a.r0 : a.r1, *wn = par ? a.w0 : a.w1, *sn = par ? a.s0 : a.s1;
    double g = 0.0, d = 0.0;
    const double s_new = fma(beta, s_own, w_own), r_new = fma(-alpha, s_new, r_own), w_new = fma(-alpha, fma(beta, S3[2], S3[1]), S3[0]);
    if (live && mine && c == 0) {
        const double p_new = fma(beta, p_own, r_own);
        sn[idx] = s_new; rn[idx] = r_new; wn[idx] = w_new;
        a.p[idx] = p_new;
        a.x[idx] = fma(alpha, p_new, x_own);
        g = r_new * r_new; d = w_new * r_new;
    }
    if (a.mb0 && live) push_row(a, par ? a.mb0 : a.mb1, q0, q1, part, SH, lane, tq_first, r_new, w_new, s_new);   // (every wavefront of the row holds the sums)
    PSTAMP(3);
    publish_partials<4 * WPR>(a, par ^ 1, g, d, sh);
    if (first) { a.alpha_hist[a.it] = alpha; a.scal[SC_PCG_ITERS] = (double)(a.it + 1); }
    PSTAMP(4);
}

// ------------------------------------------------------------------------------------------
// k_pcg_persist (opt-in, MI_BA_PCG_PERSIST=1: measured SLOWER than a launch per iteration, 12.0 against 7.4 us per iteration on C3 --
// engine.cpp setup_persist): the whole reduced solve in ONE launch, for systems whose scaled matrix fits the chip's LDS (C3: 28 MB
// over 200 workgroups).  The same iteration as k_pcg_iter -- Chronopoulos-Gear on S~, the neighbours' new residuals recomputed inside the
// product, one synchronisation per iteration -- with the launch boundary replaced by a grid barrier:
//   * a workgroup owns rows_per_wg consecutive block rows; their blocks and column indices are copied into LDS once and stay;
//     the rows' own vectors (r, w, s, p, x~) live in registers for the whole solve;
//   * what the neighbours need -- r, w, s of the rows, two dot-product partials per workgroup -- goes through HBM with agent-scope
//     (sc1) stores, drained (vmcnt 0) before the workgroup arrives at the barrier, and is read with agent-scope loads after it:
//     the per-XCD L2s are not coherent with each other, a plain load may see a stale line (MI355X_MICROARCH.md, correctness
//     boundaries; tools/xcd_barrier_probe2.hip: 0 stale values in 2 000 rounds);
//   * the barrier is a two-level tree of monotonic counters (groups of `group` workgroups, the last arrival of a group reports to the
//     root, the last of all bumps a generation word everybody polls): 5.2 us per round at 256 workgroups with the exchange against
//     10 with one counter (profiles/r03_persistent_cg_probe.txt).  Every spin is BOUNDED (50 ms): a workgroup that gives up raises the
//     abort word, everybody leaves, SC_PCG_STATUS = 3 and SC_PCG_FINAL stays 0 -- the host then solves with the launched iteration
//     (engine.cpp) and stops using this kernel.  Launched with hipLaunchCooperativeKernel (all workgroups resident, checked).
// Every workgroup reads the same partials in the same order, so all take the same decisions (convergence, breakdown, the cap) in
// the same round: no workgroup ever waits at a barrier others have left.  Several ranks: the replicated solve stays bit-identical.
// ------------------------------------------------------------------------------------------
typedef unsigned long long u64;
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool persist_barrier(u64 *bar, int group, u64 round /* 1, 2, ... */)
{
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wavefront's agent-scope stores have reached the coherence point
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const int n = gridDim.x, b = blockIdx.x;
        const int grp = b / group, ng = (n + group - 1) / group, gsz = min(group, n - grp * group);
        const u64 v = __hip_atomic_fetch_add(bar + kBarGroup0 + 8 * grp, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v + 1 == (u64)gsz * round) {                         // last of its group this round
            const u64 rr = __hip_atomic_fetch_add(bar + kBarRoot, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rr + 1 == (u64)ng * round) __hip_atomic_store(bar + kBarGen, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long t0 = wall_clock64();                     // 100 MHz
        while (__hip_atomic_load(bar + kBarGen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
            if (wall_clock64() - t0 > 5000000ll || __hip_atomic_load(bar + kBarAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
                __hip_atomic_store(bar + kBarAbort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return __syncthreads_and(ok);
}

// products of one block row held in LDS (blocks tb .. te of the workgroup's copy) with the NV vectors of the workgroup's window
// (sWin[local column][vector][8], staged from HBM once per iteration).  Lane (half, r, c2) as in row_products: the lower
// half-wave takes the row's even blocks, the upper the odd ones.  The result: row r's value in all lanes 8r .. 8r + 7.
template <int NV>
__device__ __forceinline__ void row_products_lds(const double *sS, const int *sLoc, const double *sWin, int tb, int te, int lane, double (&out)[NV])
{
    const int half = lane >> 5, c2 = (lane & 3) * 2, el = ((lane & 31) >> 2) * 8 + c2;
    double acc[NV][2];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v][0] = acc[v][1] = 0.0;
    int t = tb + half;
    for (; t + 2 < te; t += 4) {
        const double2 sa = *reinterpret_cast<const double2 *>(sS + (size_t)t * 64 + el), sb = *reinterpret_cast<const double2 *>(sS + (size_t)(t + 2) * 64 + el);
        const double *xa = sWin + sLoc[t] * 24 + c2, *xb = sWin + sLoc[t + 2] * 24 + c2;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double2 va = *reinterpret_cast<const double2 *>(xa + 8 * v), vb = *reinterpret_cast<const double2 *>(xb + 8 * v);
            acc[v][0] = fma(sa.y, va.y, fma(sa.x, va.x, acc[v][0]));
            acc[v][1] = fma(sb.y, vb.y, fma(sb.x, vb.x, acc[v][1]));
        }
    }
    for (; t < te; t += 2) {
        const double2 sa = *reinterpret_cast<const double2 *>(sS + (size_t)t * 64 + el);
        const double *xa = sWin + sLoc[t] * 24 + c2;
#pragma unroll
        for (int v = 0; v < NV; ++v) { const double2 va = *reinterpret_cast<const double2 *>(xa + 8 * v); acc[v][0] = fma(sa.y, va.y, fma(sa.x, va.x, acc[v][0])); }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double tt = acc[v][0] + acc[v][1];
        tt += dpp_move<0xB1>(tt); tt += dpp_move<0x4E>(tt);
        tt += __shfl_xor(tt, 32);
        out[v] = __shfl(tt, (lane >> 3) * 4);
    }
}

__global__ __launch_bounds__(256) void k_pcg_persist(const PersistArgs g)
{
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ double sh[4][2], sgd[2];
    const PcgArgs &a = g.a;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7, nwg = gridDim.x, b = blockIdx.x;
    const int j0 = b * g.rows_per_wg, j1 = min(a.m, j0 + g.rows_per_wg);
    const int qb = a.bsr_row[j0], nblk = a.bsr_row[j1] - qb;
    const int wq0 = g.win_ptr[b], ncol = g.win_ptr[b + 1] - wq0;     // the distinct columns of this workgroup's rows (engine.cpp setup_persist)
    double *sS = lds_dyn;                                            // nblk x 64: the rows' blocks of S~
    double *sWin = sS + (size_t)nblk * 64;                           // ncol x 3 x 8: r, w, s of those columns, this iteration's
    int *sLoc = reinterpret_cast<int *>(sWin + (size_t)ncol * 24);   // nblk: window position of each block's column
    int *sWcol = sLoc + nblk;                                        // ncol: the columns
    for (int i = threadIdx.x; i < nblk * 32; i += 256) reinterpret_cast<double2 *>(sS)[i] = reinterpret_cast<const double2 *>(a.S + (size_t)qb * 64)[i];
    for (int i = threadIdx.x; i < nblk; i += 256) sLoc[i] = g.blk_loc[qb + i];
    for (int i = threadIdx.x; i < ncol; i += 256) sWcol[i] = g.win_col[wq0 + i];
    // this wavefront's rows: j0 + wv and j0 + wv + 4 (rows_per_wg <= 8)
    int jr[2], tb[2], te[2];
    bool live[2];
    double r_own[2], w_own[2], s_own[2], p_own[2], x_own[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        jr[k] = j0 + wv + 4 * k;
        live[k] = jr[k] < j1;
        const int jj = live[k] ? jr[k] : j0;
        tb[k] = a.bsr_row[jj] - qb; te[k] = live[k] ? a.bsr_row[jj + 1] - qb : tb[k];
        r_own[k] = live[k] ? a.r0[jj * 8 + r] : 0.0;
        w_own[k] = s_own[k] = p_own[k] = x_own[k] = 0.0;
    }
    __syncthreads();
    // the window: 8 consecutive doubles per (column, vector) -- 64-byte pieces, agent-scope loads, all in flight at once
    auto stage = [&](const double *v0, const double *v1, const double *v2, int nv) {
        for (int i = threadIdx.x; i < ncol * 24; i += 256) {
            const int cc = i / 24, rem = i - cc * 24, v = rem >> 3, e = rem & 7;
            if (v < nv) sWin[i] = ld_agent((v == 0 ? v0 : v == 1 ? v1 : v2) + sWcol[cc] * 8 + e);
        }
    };
    auto publish = [&](int slot, double gg, double dd) {   // the workgroup's two partials, four wavefronts in a fixed order
        gg = wave_sum(gg); dd = wave_sum(dd);
        if (lane == 0) { sh[wv][0] = gg; sh[wv][1] = dd; }
        __syncthreads();
        if (threadIdx.x == 0) {
            st_agent(a.gp + (size_t)slot * nwg + b, (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]));
            st_agent(a.dp + (size_t)slot * nwg + b, (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]));
        }
    };
    // ---- start: w_0 = S~ r_0 (r_0: k_block_inv, an earlier launch), partials of r_0.r_0 and w_0.r_0
    {
        stage(a.r0, a.r0, a.r0, 1);
        __syncthreads();
        double gg = 0.0, dd = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double w[1];
            row_products_lds<1>(sS, sLoc, sWin, tb[k], te[k], lane, w);
            if (live[k]) {
                w_own[k] = w[0];
                if (c == 0) { st_agent(a.w0 + jr[k] * 8 + r, w[0]); gg += r_own[k] * r_own[k]; dd += w[0] * r_own[k]; }
            }
        }
        publish(0, gg, dd);
    }
    u64 round = 1;
    bool alive = persist_barrier(g.bar, g.group, round);
    double g_zero = 0.0, g_prev = 0.0, a_prev = 0.0;
    int iters = 0, status = 0, done = 0;
    for (int it = 0; alive; ++it) {
        const int par = it & 1;
        const double *__restrict__ rc = par ? a.r1 : a.r0, *__restrict__ wc = par ? a.w1 : a.w0, *__restrict__ so = par ? a.s1 : a.s0;
        // one round trip: the two dot products' partials (every workgroup sums all of them in the same order) and the window
        double gs = 0.0, ds = 0.0;
        if (wv == 0)
            for (int i = lane; i < nwg; i += 64) { gs += ld_agent(a.gp + (size_t)par * nwg + i); ds += ld_agent(a.dp + (size_t)par * nwg + i); }
        stage(rc, wc, so, 3);
        if (wv == 0) {
            gs = wave_sum(gs); ds = wave_sum(ds);
            if (lane == 0) { sgd[0] = gs; sgd[1] = ds; }
        }
        __syncthreads();
        const double gamma = sgd[0], delta = sgd[1];
        if (it == 0) g_zero = gamma;
        const bool conv = it == 0 ? !(gamma > 0.0) : (!(g_zero > 0.0) || sqrt(fabs(gamma) / g_zero) < (it >= a.minit ? a.tol : fmin(a.tol, kPcgFloor)));
        if (conv) { done = 1; break; }
        if (it >= g.maxit) break;
        const double beta = it ? gamma / g_prev : 0.0;
        const double den = it ? delta - beta * gamma / a_prev : delta;   // = p.S~p
        if (!(den > 0.0) || !isfinite(den)) { status = 2; break; }
        const double alpha = gamma / den;
        double *rn = par ? a.r0 : a.r1, *wn = par ? a.w0 : a.w1, *sn = par ? a.s0 : a.s1;
        double gg = 0.0, dd = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double S3[3];
            row_products_lds<3>(sS, sLoc, sWin, tb[k], te[k], lane, S3);
            if (live[k]) {
                const double s_new = fma(beta, s_own[k], w_own[k]), r_new = fma(-alpha, s_new, r_own[k]), p_new = fma(beta, p_own[k], r_own[k]);
                const double w_new = fma(-alpha, fma(beta, S3[2], S3[1]), S3[0]);
                x_own[k] = fma(alpha, p_new, x_own[k]);
                s_own[k] = s_new; r_own[k] = r_new; p_own[k] = p_new; w_own[k] = w_new;
                if (c == 0) {
                    const int idx = jr[k] * 8 + r;
                    st_agent(sn + idx, s_new); st_agent(rn + idx, r_new); st_agent(wn + idx, w_new);
                    gg += r_new * r_new; dd += w_new * r_new;
                }
            }
        }
        publish(par ^ 1, gg, dd);          // (its barrier also orders this iteration's window reads against the next one's staging)
        g_prev = gamma; a_prev = alpha; iters = it + 1;
        alive = persist_barrier(g.bar, g.group, ++round);
    }
    // ---- the end of the solve: x~ of the rows for k_cam_update (a later launch: plain stores), the flags the host and k_backsub read
    if (alive) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (live[k] && c == 0) a.x[jr[k] * 8 + r] = x_own[k];
        // one more round: a workgroup whose bounded spin gave up in the LAST iteration's barrier (just before the generation word
        // landed) left without storing its rows -- it does not arrive here, the round times out for everybody and the solve is
        // reported unusable (status 3) instead of workgroup 0 declaring a partly stale x~ final
        alive = persist_barrier(g.bar, g.group, ++round);
    }
    if (b == 0 && threadIdx.x == 0) {
        a.scal[SC_PCG_ITERS] = (double)iters;
        if (!alive) a.scal[SC_PCG_STATUS] = 3.0;          // a barrier gave up: nothing of this solve is usable, SC_PCG_FINAL stays 0
        else {
            if (done) a.scal[SC_PCG_DONE] = 1.0;
            if (status) a.scal[SC_PCG_STATUS] = (double)status;
            a.scal[SC_PCG_FINAL] = 1.0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// camera update: a_new = a + da, new camera record, dL and ||dp||^2 camera parts
// (sba_levmar.c:1444-1447, 1524-1525)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pcg_tail(const PcgArgs &a);

// da_k[r] = (Linv_k^T x~_k)[r]: the end of the scaled solve for one camera parameter
__device__ __forceinline__ double scaled_back(const PcgArgs &pa, int k, int r)
{
    const double *Li = pa.Linv + (size_t)k * 64, *xk = pa.x + (size_t)k * 8;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) v = fma(Li[c * 8 + r], xk[c], v);
    return v;
}

// pa: the finished (or, launched ahead, perhaps unfinished: SC_PCG_FINAL) scaled solve -- this kernel also ends it: da = Linv^T x~
// goes to dA for the back-substitution, and with pa.tail the first workgroup does what k_pcg_tail does
__global__ void k_cam_update(int m, int cnp, ModelCfg cfg, const unsigned char *__restrict__ fixed, double mu, const double *__restrict__ cam_a,
                             const PcgArgs pa, double *__restrict__ dA, const double *__restrict__ camvec,
                             const double *__restrict__ R0, const double *__restrict__ f_fixed, const CamLinks lk,
                             double *__restrict__ cam_a_new, double *__restrict__ cam_lin_new, double *__restrict__ part, double *__restrict__ R0_new)
{
    // one thread per REAL camera; its parameters sit in row j of the camera-side arrays, or -- nine parameters, cfg.split -- in
    // rows 2j (parameters 0-7) and 2j + 1 (parameter 8, the rest of that row is padding held at zero)
    if (pa.tail && blockIdx.x == 0) pcg_tail(pa);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double dl = 0.0, dp2 = 0.0, lc = 0.0;
    if (j < m) {
        const int vb = cfg.split ? 2 * j : j, nrow = cfg.split ? 2 : 1;
        double an[9];
        an[8] = 0.0;
        for (int row = 0; row < nrow; ++row) {
            const int v = vb + row;
#pragma unroll
            for (int r = 0; r < kCP; ++r) {
                const double dj = scaled_back(pa, v, r);
                dA[v * kCP + r] = dj;
                const bool live = (row == 0 ? r < cnp : r == 0) && !fixed[v];
                const double d = live ? dj : 0.0;
                double a_new = cam_a[v * kCP + r] + d;
                // PBA's UpdateCamera keeps the focal length positive: f = max(f + d, 1e-10) (SparseBundleCPU.cpp:1082), in its own units
                if (cfg.pba_update && cfg.est_focal && row == 0 && r == 6 && live) a_new = fmax(a_new, 1e-10 * cfg.f_scale);
                cam_a_new[v * kCP + r] = a_new;
                const double D = cfg.damp_diag ? camvec[(2 * (size_t)(cfg.split ? 2 * m : m) + v) * kCP + r] : 1.0;   // PBA: dL = d'(mu D d + g), |d|^2 in D's metric
                dl += d * (mu * D * d + camvec[v * kCP + r]);
                dp2 += D * d * d;
                if (row == 0) an[r] = a_new; else if (r == 0) an[8] = a_new;
            }
        }
        cam_record(cfg, cnp, an, R0 + (size_t)j * 9, f_fixed ? f_fixed[j] : 1.0, cam_lin_new + (size_t)j * kCamLin);
        if (cfg.pba_update && R0_new) {
            // PBA's update of the rotation, R <- exp([d omega]x) R (:1089-1098): the trial's rotation becomes the rotation the camera is
            // linearised about, omega goes back to 0 and with it J_l(omega) = I -- the next Jacobian is PBA's JacobianOne (:1332-1455)
            double *cl = cam_lin_new + (size_t)j * kCamLin;
#pragma unroll
            for (int i = 0; i < 9; ++i) { R0_new[(size_t)j * 9 + i] = cl[i]; cl[9 + i] = (i % 4 == 0) ? 1.0 : 0.0; }
            cam_a_new[vb * kCP + 3] = 0.0; cam_a_new[vb * kCP + 4] = 0.0; cam_a_new[vb * kCP + 5] = 0.0;
        }
        if (lk.row)                                            // link penalty at the trial point (links couple parameters 0-7: row vb)
            for (int q = lk.row[vb]; q < lk.row[vb + 1]; ++q) {
                const int k = lk.other[q];
                for (int r = 0; r < (cnp < kCP ? cnp : kCP); ++r) {
                    const double ak = cam_a[k * kCP + r] + (fixed[k] ? 0.0 : scaled_back(pa, k, r)), dq = ak - an[r];
                    lc = fma(0.5 * lk.w[(size_t)q * kCP + r], dq * dq, lc);
                }
            }
    }
    dl = wave_sum(dl); dp2 = wave_sum(dp2); lc = wave_sum(lc);
    if ((threadIdx.x & 63) == 0) {
        double *pr = part + (size_t)blockIdx.x * kPartW;
        pr[0] = dl; pr[1] = dp2; pr[2] = lc; pr[3] = 0.0; pr[4] = 0.0; pr[5] = 0.0; pr[6] = 0.0; pr[7] = 0.0;
    }
}

// The same with EIGHT lanes per camera, lane = (camera, parameter) -- every camera model but the nine-parameter one (cfg.split).  A thread per camera
// was 16 wavefronts on a 1 000-camera scene, each lane walking its own 512-byte block of L^-1 (64 loads of 64 different lines per instruction) and
// then the rotation's exponential: 12.6 us between the reduced solve and the back-substitution.  Here a camera's eight lanes read its block of
// L^-1 as one contiguous piece, the new parameters meet in the camera's first lane through LDS for the record.  A partial row per wavefront = 8 cameras.
__global__ __launch_bounds__(64) void k_cam_update8(int m, int cnp, ModelCfg cfg, const unsigned char *__restrict__ fixed, double mu, const double *__restrict__ cam_a,
                                                 const PcgArgs pa, double *__restrict__ dA, const double *__restrict__ camvec,
                                                 const double *__restrict__ R0, const double *__restrict__ f_fixed, const CamLinks lk,
                                                 double *__restrict__ cam_a_new, double *__restrict__ cam_lin_new, double *__restrict__ part, double *__restrict__ R0_new)
{
    __shared__ double sA[64];
    if (pa.tail && blockIdx.x == 0) pcg_tail(pa);
    const int lane = threadIdx.x, r = lane & 7, j = blockIdx.x * 8 + (lane >> 3);
    double dl = 0.0, dp2 = 0.0, lc = 0.0;
    const bool on = j < m;
    const int jj = on ? j : 0;
    const double dj = scaled_back(pa, jj, r);
    const bool live = on && r < cnp && !fixed[jj];
    const double d = live ? dj : 0.0;
    double a_new = cam_a[jj * kCP + r] + d;
    // PBA's UpdateCamera keeps the focal length positive: f = max(f + d, 1e-10) (SparseBundleCPU.cpp:1082), in its own units
    if (cfg.pba_update && cfg.est_focal && r == 6 && live) a_new = fmax(a_new, 1e-10 * cfg.f_scale);
    const double D = cfg.damp_diag ? camvec[(2 * (size_t)m + jj) * kCP + r] : 1.0;   // PBA: dL = d'(mu D d + g), |d|^2 in D's metric
    if (on) {
        dA[j * kCP + r] = dj;
        cam_a_new[j * kCP + r] = a_new;
        dl = d * (mu * D * d + camvec[j * kCP + r]);
        dp2 = D * d * d;
    }
    sA[lane] = a_new;
    wave_lds_sync();
    if (on && r == 0) {
        double an[9];
#pragma unroll
        for (int i = 0; i < 8; ++i) an[i] = sA[lane + i];
        an[8] = 0.0;
        cam_record(cfg, cnp, an, R0 + (size_t)j * 9, f_fixed ? f_fixed[j] : 1.0, cam_lin_new + (size_t)j * kCamLin);
        if (cfg.pba_update && R0_new) {   // PBA's update of the rotation (k_cam_update above)
            double *cl = cam_lin_new + (size_t)j * kCamLin;
#pragma unroll
            for (int i = 0; i < 9; ++i) { R0_new[(size_t)j * 9 + i] = cl[i]; cl[9 + i] = (i % 4 == 0) ? 1.0 : 0.0; }
            cam_a_new[j * kCP + 3] = 0.0; cam_a_new[j * kCP + 4] = 0.0; cam_a_new[j * kCP + 5] = 0.0;
        }
        if (lk.row)                                            // link penalty at the trial point
            for (int q = lk.row[j]; q < lk.row[j + 1]; ++q) {
                const int k = lk.other[q];
                for (int rr = 0; rr < (cnp < kCP ? cnp : kCP); ++rr) {
                    const double ak = cam_a[k * kCP + rr] + (fixed[k] ? 0.0 : scaled_back(pa, k, rr)), dq = ak - an[rr];
                    lc = fma(0.5 * lk.w[(size_t)q * kCP + rr], dq * dq, lc);
                }
            }
    }
    dl = wave_sum(dl); dp2 = wave_sum(dp2); lc = wave_sum(lc);
    if (lane == 0) {
        double *pr = part + (size_t)blockIdx.x * kPartW;
        pr[0] = dl; pr[1] = dp2; pr[2] = lc; pr[3] = 0.0; pr[4] = 0.0; pr[5] = 0.0; pr[6] = 0.0; pr[7] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// k_backsub: db_i = V*^-1 (eb_i - sum_j W_ij^T da_j) = h_i - V*^-1 sum_j B^T (A da_j)
// (sba_levmar.c:1393-1433), trial point, cost at p + dp (:1468-1476), dL and ||dp||^2 point parts,
// and Bundler's max relative residual change (:1552-1561).  One wavefront per batch.
// ------------------------------------------------------------------------------------------
#ifndef MIBA_BS_WAVES
#define MIBA_BS_WAVES 3
#endif
template <int CNP, bool FULL>
__global__ __launch_bounds__(64, FULL ? 2 : MIBA_BS_WAVES) void k_backsub(const BacksubArgs g)   // three wavefronts per SIMD: 168 VGPRs (two more and the allocation granule makes it two wavefronts: 0.20 -> 0.25 ms on C3)
{
    __shared__ double sT[kWave * 3], sX[kMaxPtsBatch * 3], sD[kWave * 3];
    __shared__ int sStart[kMaxPtsBatch + 1];
    if (g.scal[SC_PCG_FINAL] == 0.0) return;   // launched ahead of the host's look at the reduced solve (engine.cpp solve_phase)
    const int lane = threadIdx.x;
    const ModelCfg cfg = g.cfg;
    double cn = 0.0, pct = 0.0, dl = 0.0, dp2 = 0.0;
    // Each workgroup walks a CONTIGUOUS chunk of batches: with signature-sorted points lane l sees the
    // same camera batch after batch, so the three per-camera gathers (record at p, record at p + dp, da)
    // stay in registers and are re-fetched only when the lane's camera changes (the kernel was bound by the
    // texture addresser before: TA_BUSY 85 %, 29 cache accesses per observation).  The observation stream
    // of the next batch is requested one batch ahead; everything a batch needs from HBM (point, ptout row
    // of the head lane) is requested at its top, so no load sits behind an LDS hand-off.
    const int chunk = (g.nbatch + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * chunk, b1 = min(g.nbatch, b0 + chunk);
    double cl[kCamLin], cln[16];
    CamDelta cd;
    int jc = -1;
    // software pipeline registers: the next batch's descriptor and observation stream
    int n_o0 = 0, n_nob = 0, n_p0 = 0, n_npt = 0, n_j = 0, n_ip = 0;
    double2 n_xy = make_double2(0.0, 0.0);
    auto load_stream = [&](int b) {
        n_o0 = g.batch_obs[b]; n_nob = g.batch_obs_end[b] - n_o0;
        n_p0 = g.batch_pt[b]; n_npt = g.batch_pt_end[b] - n_p0;
        n_j = 0; n_ip = n_p0; n_xy = make_double2(0.0, 0.0);
        if (lane < n_nob) { const int o = n_o0 + lane; n_j = g.obs_cam[o]; n_ip = g.obs_pt[o]; n_xy = g.obs_xy[o]; }
    };
    if (b0 < b1) load_stream(b0);
    for (int b = b0; b < b1; ++b) {
        const int nob = n_nob, p0 = n_p0, npt = n_npt, j = n_j, ip = n_ip;
        const double2 xy = n_xy;
        const bool active = lane < nob;
        const int lp = ip - p0;
        if (b + 1 < b1) load_stream(b + 1);
        // everything this batch needs from HBM, requested now
        const double X[3] = { g.pts[3 * (size_t)ip], g.pts[3 * (size_t)ip + 1], g.pts[3 * (size_t)ip + 2] };
        const int lp_prev = __builtin_amdgcn_update_dpp(-2, active ? lp : -2, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps -2)
        const bool head = active && (lane == 0 || lp != lp_prev);
        if (head) sStart[lp] = lane;
        if (lane == 0) sStart[npt] = nob;
        // the point side runs on lanes (point, component): four lanes per point, sixteen points per sweep.  What such a lane needs from
        // HBM -- its row of V*^-1, h_c, eb_c and X_c -- is requested here, for the first sweep (a batch has more than sixteen points only
        // when tracks are shorter than four views)
        const int qp = lane >> 2, qc = lane & 3;
        const int c3 = qc < 3 ? qc : 0;
        const int vi0 = c3 == 0 ? 0 : c3, vi1 = c3 == 0 ? 1 : (c3 == 1 ? 3 : 4), vi2 = c3 == 0 ? 2 : (c3 == 1 ? 4 : 5);   // row c of the packed inverse (00 01 02 11 12 22)
        double pv0 = 0.0, pv1 = 0.0, pv2 = 0.0, ph = 0.0, pe = 0.0, pX = 0.0;
        if (qp < npt && qc < 3) {
            const double *po = g.ptout + (size_t)(p0 + qp) * kPtOut;
            pv0 = po[vi0]; pv1 = po[vi1]; pv2 = po[vi2]; ph = po[6 + c3]; pe = po[9 + c3];
            pX = g.pts[3 * (size_t)(p0 + qp) + c3];
        }
        double e[2] = { 0.0, 0.0 };
        if (active) {
            if (j != jc) {
                const double2 *s0 = reinterpret_cast<const double2 *>(g.cam_lin + (size_t)j * kCamLin);
                const double2 *s1 = reinterpret_cast<const double2 *>(g.cam_lin_new + (size_t)j * kCamLin);
                const double2 *s2 = reinterpret_cast<const double2 *>(g.dA + (size_t)(cfg.split ? 2 * j : j) * kCP);
#pragma unroll
                for (int k = 0; k < kCamLin / 2; ++k) { const double2 v = s0[k]; cl[2 * k] = v.x; cl[2 * k + 1] = v.y; }
#pragma unroll
                for (int k = 0; k < 5; ++k) { const double2 v = s1[k]; cln[2 * k] = v.x; cln[2 * k + 1] = v.y; }       // R (9)
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double2 v = s1[9 + k]; cln[10 + 2 * k] = v.x; cln[11 + 2 * k] = v.y; } // c f k1 k2
                double da[9];
#pragma unroll
                for (int k = 0; k < kCP / 2; ++k) { const double2 v = s2[k]; da[2 * k] = v.x; da[2 * k + 1] = v.y; }
                da[8] = (CNP > kCP && cfg.split) ? g.dA[(size_t)(2 * j + 1) * kCP] : 0.0;
                cd = cam_delta(cfg, cl, da, g.cam_fixed[j] != 0, FULL);
                jc = j;
            }
            double t[3], bd[3];
            backsub_obs<CNP, FULL>(cfg, cl, cd, X, xy, e, t, bd);
#pragma unroll
            for (int c = 0; c < 3; ++c) sT[lane * 3 + c] = t[c];
            if (cfg.damp_diag) {
#pragma unroll
                for (int c = 0; c < 3; ++c) sD[lane * 3 + c] = bd[c];
            }
        }
        wave_lds_sync();
        // ---- the point's update on lanes (point, component): sum of the point's t (and diag V) from LDS, the three sums shared inside the
        // quad by DPP, one row of V*^-1 per lane.  (Until round 4 the point's head lane did all of it alone: a chain of 30 dependent
        // LDS loads and adds and the 3 x 3 product on 6 lanes of 64.)
        for (int base = 0; base < npt; base += 16) {
            const int lq = base + qp;
            const bool on = lq < npt && qc < 3;
            if (base > 0) {                                 // further sweeps: their loads could not be issued ahead
                pv0 = pv1 = pv2 = ph = pe = pX = 0.0;
                if (on) {
                    const double *po = g.ptout + (size_t)(p0 + lq) * kPtOut;
                    pv0 = po[vi0]; pv1 = po[vi1]; pv2 = po[vi2]; ph = po[6 + c3]; pe = po[9 + c3];
                    pX = g.pts[3 * (size_t)(p0 + lq) + c3];
                }
            }
            double tc0 = 0.0, tc1 = 0.0, Dc = 1.0;
            if (on) {
                const int os = sStart[lq], oe = sStart[lq + 1];
                int o = os;
                for (; o + 1 < oe; o += 2) { tc0 += sT[o * 3 + c3]; tc1 += sT[o * 3 + 3 + c3]; }
                if (o < oe) tc0 += sT[o * 3 + c3];
                if (cfg.damp_diag) {                                            // PBA: dL and |d|^2 in the metric of diag V_i
                    Dc = 0.0;
                    for (o = os; o < oe; ++o) Dc += sD[o * 3 + c3];
                }
            }
            const double tc = tc0 + tc1;
            const double t0 = dpp_move<0x00>(tc), t1 = dpp_move<0x55>(tc), t2 = dpp_move<0xAA>(tc);   // quad_perm [0,0,0,0], [1,1,1,1], [2,2,2,2]
            if (on) {
                const double d = ph - (pv0 * t0 + pv1 * t1 + pv2 * t2);
                const double Xn = pX + d;
                const size_t ix = 3 * (size_t)(p0 + lq) + c3;
                g.pts_new[ix] = Xn;
                if (g.dB) g.dB[ix] = d;
                sX[lq * 3 + c3] = Xn;
                dl += d * (g.mu * Dc * d + pe);
                dp2 += Dc * d * d;
            }
        }
        wave_lds_sync();
        if (active) {
            const double Xn[3] = { sX[lp * 3], sX[lp * 3 + 1], sX[lp * 3 + 2] };
            // project_obs on the compact new record: R = cln[0..8], c = cln[10..12], f = cln[13], k1, k2 = cln[14..15]
            double q[3];
            if (cfg.explicit_centers) {
                const double d0 = Xn[0] - cln[10], d1 = Xn[1] - cln[11], d2 = Xn[2] - cln[12];
#pragma unroll
                for (int r = 0; r < 3; ++r) q[r] = cln[3 * r] * d0 + cln[3 * r + 1] * d1 + cln[3 * r + 2] * d2;
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) q[r] = cln[3 * r] * Xn[0] + cln[3 * r + 1] * Xn[1] + cln[3 * r + 2] * Xn[2] + cln[10 + r];
            }
            const double iz = fast_rcp(q[2]), mx = q[0] * iz, my = q[1] * iz, r2 = mx * mx + my * my;
            const bool md = FULL && cfg.undistort == 2;
            const double gg = cfg.proj_sign * cln[13] * ((!FULL || md) ? 1.0 : 1.0 + cln[14] * r2 + cln[15] * r2 * r2);
            double en0 = xy.x - gg * mx, en1 = xy.y - gg * my;
            if (md) { const double kk = cln[14] * (xy.x * xy.x + xy.y * xy.y); en0 = fma(kk, xy.x, en0); en1 = fma(kk, xy.y, en1); }
            cn += en0 * en0 + en1 * en1;
            // QUIRK kept from sba_levmar.c:1554: signed comparison against eps5 (pct_change_flag)
            pct = fmax(pct, fmax(pct_change_flag(e[0], en0, g.eps5), pct_change_flag(e[1], en1, g.eps5)));
        }
        wave_lds_sync();
    }
    cn = wave_sum(cn); dl = wave_sum(dl); dp2 = wave_sum(dp2); pct = wave_max(pct);
    if (lane == 0) {
        double *pr = g.part + (size_t)blockIdx.x * kPartW;
        pr[0] = cn; pr[1] = dl; pr[2] = dp2; pr[3] = 0.0; pr[4] = pct; pr[5] = 0.0; pr[6] = 0.0; pr[7] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// k_backsub_uni: the back-substitution of a tile of ONE camera set (plan.h tile mode 4; the planner keeps those points out of
// k_backsub's batches).  Same arithmetic as k_backsub; what the uniform shape buys here is OCCUPANCY: k_backsub keeps a lane's camera --
// its record at p, its record at p + dp and its update, 33 doubles -- in registers across batches (144-160 VGPRs, three wavefronts per
// SIMD), and its batches are chains of dependent latency (LDS round trips, two reciprocal chains: ~5 600 cycles for ~1 100 of issue)
// that three wavefronts do not cover.  Here the tile's <= 10 cameras sit in an LDS table, built once per tile; a lane's state is its
// lane arithmetic and four accumulators, and five wavefronts fit a SIMD.  The measurement and point streams come by LDS-DMA one
// batch ahead (k_build_uni's), the ptout rows straight into the registers of the (point, component) lanes at the batch's top.
// ------------------------------------------------------------------------------------------
template <int CNP, bool FULL, bool DAMP, int DC /* the tile's view count as a constant, or 0: k_build_uni */>
__global__ __launch_bounds__(64, 4) void k_backsub_uni   /* (five: 96 VGPRs with two spilled -- scratch reloads among the transfers -- 0.154 ms against 0.116) */(const BacksubArgs g)
{
    const int tile = blockIdx.x;
    double *const prow = g.part + (size_t)(g.uni_row0 + tile) * kPartW;
    if (g.scal[SC_PCG_FINAL] == 0.0) return;   // launched ahead of the host's look at the reduced solve (engine.cpp solve_phase)
    const int lane = threadIdx.x;
    if (g.tile_mode[tile] != 4) {               // not a uniform tile: its points are k_backsub's; a row of zeros for the reduction
        if (lane < kPartW && (DC > 0 || !g.uni_fixed)) prow[lane] = 0.0;     // (written by one of the two instantiations)
        return;
    }
    {
        const int d_ = g.tile_cslot[tile + 1] - g.tile_cslot[tile];
        if (DC > 0 ? d_ != DC : (g.uni_fixed && d_ == kUniFixedViews)) return;      // the other instantiation's tile
    }
    constexpr int kMaxD = kMaxViewsLds, kMaxPts = kWave / kUniMinViews;
    constexpr int SCL = 18;                      // a compact camera record: R (9) - c (3) f k1 k2 at 10..15; 18 doubles apart (distinct banks)
    constexpr int SCD = 10;                      // the camera's update (CamDelta, 9 doubles)
    __shared__ __attribute__((aligned(16))) double sCl[kMaxD * SCL], sCln[kMaxD * SCL], sCd[kMaxD * SCD];
    __shared__ __attribute__((aligned(16))) double2 sXY[2][kWave];
    __shared__ __attribute__((aligned(16))) double sPX[2][kMaxPts * 3];
    __shared__ double sT[kWave * 3], sXn[kMaxPts * 3];
    __shared__ double sD[DAMP ? kWave * 3 : 1];
    __shared__ int sFix[kMaxD];
    const ModelCfg cfg = g.cfg;
    const int D = DC > 0 ? DC : g.tile_cslot[tile + 1] - g.tile_cslot[tile];
    const int PB = kWave / D;
    const int lp = lane / D, cs = lane - lp * D;
    // ---- the tile's camera table: lane c prepares camera c (record at p, record at p + dp, the update's per-camera vectors)
    if (lane < D) {
        const int j = g.cslot_cam[g.tile_cslot[tile] + lane];
        const double *cl = g.cam_lin + (size_t)j * kCamLin, *cn = g.cam_lin_new + (size_t)j * kCamLin;
        double rec[kCamLin], da[9];
#pragma unroll
        for (int k = 0; k < kCamLin; ++k) rec[k] = cl[k];
        const double *dj = g.dA + (size_t)(cfg.split ? 2 * j : j) * kCP;
#pragma unroll
        for (int k = 0; k < kCP; ++k) da[k] = dj[k];
        da[8] = (CNP > kCP && cfg.split) ? g.dA[(size_t)(2 * j + 1) * kCP] : 0.0;
        const bool fx = g.cam_fixed[j] != 0;
        const CamDelta cd = cam_delta(cfg, rec, da, fx, FULL);
        double *o = sCl + lane * SCL, *on = sCln + lane * SCL, *od = sCd + lane * SCD;
#pragma unroll
        for (int k = 0; k < 9; ++k) { o[k] = rec[k]; on[k] = cn[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { o[10 + k] = rec[18 + k]; on[10 + k] = cn[18 + k]; }
        od[0] = cd.v0[0]; od[1] = cd.v0[1]; od[2] = cd.v0[2]; od[3] = cd.u[0]; od[4] = cd.u[1]; od[5] = cd.u[2]; od[6] = cd.df; od[7] = cd.dk; od[8] = cd.dk2;
        sFix[lane] = fx;
    }
    const int b_begin = g.tile_batch[tile], b_end = g.tile_batch[tile + 1];
    const int tp0 = g.bld_batch_pt[b_begin], tp1 = g.bld_batch_pt[b_end], to0 = g.bld_batch_obs[b_begin];
    const int nbatch = b_end - b_begin;
    auto stream_batch = [&](int kk, int buf) {
        int lv = lane; asm volatile("" : "+v"(lv));
        const int p1 = tp0 + kk * PB, np1 = min(PB, tp1 - p1);
        const unsigned *src = reinterpret_cast<const unsigned *>(g.pts + 3 * (size_t)p1);
        if (lv < np1 * D) glds_b128(g.obs_xy + (to0 + kk * PB * D + lv), sXY[buf]);
        if (lv < 6 * np1) glds_b32(src + lv, sPX[buf]);
        if (lv + kWave < 6 * np1) glds_b32(src + kWave + lv, reinterpret_cast<unsigned *>(sPX[buf]) + kWave);
    };
    double cn = 0.0, pct = 0.0, dl = 0.0, dp2 = 0.0;
    // the lane as (point, component) of the point phase: four lanes per point
    const int qp = lane >> 2, qc = lane & 3;
    const int c3 = qc < 3 ? qc : 0;
    const int vi0 = c3 == 0 ? 0 : c3, vi1 = c3 == 0 ? 1 : (c3 == 1 ? 3 : 4), vi2 = c3 == 0 ? 2 : (c3 == 1 ? 4 : 5);   // row c of the packed inverse (00 01 02 11 12 22)
    // Memory waits in the loop: ONE full drain at a batch's top (the streams issued in the middle of the batch before have landed), and the
    // compiler's own wait for the five ptout loads of the point lanes, which are ordinary loads issued at the top and consumed behind the
    // first hand-off -- at a moment when no transfer is in flight, because the next batch's are issued only AFTER the point phase.
    // (LDS-DMA transfers and ordinary loads are not retired in their common issue order, so a counted s_waitcnt cannot separate them: a
    // version that waited vmcnt(5) for "the transfers, but not the five loads behind them" read stale LDS; and loads prefetched into
    // registers by inline asm were copied by the compiler BEFORE the hand-written wait.)
    stream_batch(0, 0);
    wave_lds_sync();   // the camera table is in place
    for (int k = 0; k < nbatch; ++k) {
        const int buf = k & 1;
        const int p0 = tp0 + k * PB, npt = min(PB, tp1 - p0);
        const bool active = lp < npt;
        const bool on = qp < npt && qc < 3;
        wait_lds_dma();
        double pv0 = 0.0, pv1 = 0.0, pv2 = 0.0, ph = 0.0, pe = 0.0;
        if (on) {
            const double *po = g.ptout + (size_t)(p0 + qp) * kPtOut;
            pv0 = po[vi0]; pv1 = po[vi1]; pv2 = po[vi2]; ph = po[6 + c3]; pe = po[9 + c3];
        }
        double2 xy = make_double2(0.0, 0.0);
        double e[2] = { 0.0, 0.0 };
        if (active) {
            xy = sXY[buf][lane];
            const double X[3] = { sPX[buf][3 * lp], sPX[buf][3 * lp + 1], sPX[buf][3 * lp + 2] };
            // backsub_obs wants the record in cam_lin's layout: R at 0..8, c f k1 k2 at 18..23 -- the compact one is read through a shim
            double cl[kCamLin];
            const double *o = sCl + cs * SCL;
#pragma unroll
            for (int i = 0; i < 9; ++i) cl[i] = o[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) cl[18 + i] = o[10 + i];
            CamDelta cd;
            const double *od = sCd + cs * SCD;
            cd.v0[0] = od[0]; cd.v0[1] = od[1]; cd.v0[2] = od[2]; cd.u[0] = od[3]; cd.u[1] = od[4]; cd.u[2] = od[5]; cd.df = od[6]; cd.dk = od[7]; cd.dk2 = od[8];
            double t[3], bd[3];
            backsub_obs<CNP, FULL>(cfg, cl, cd, X, xy, e, t, bd);
#pragma unroll
            for (int c = 0; c < 3; ++c) sT[lane * 3 + c] = t[c];
            if constexpr (DAMP) {
#pragma unroll
                for (int c = 0; c < 3; ++c) sD[lane * 3 + c] = bd[c];
            }
        }
        wave_lds_sync();
        // ---- the point's update on lanes (point, component): the point's D observations are lanes qp D .. qp D + D - 1
        {
            double tc0 = 0.0, tc1 = 0.0, Dc = 1.0, pX = 0.0;
            if (on) {
                const double *pt = sT + qp * D * 3 + c3;
#pragma unroll
                for (int v = 0; v < kMaxViewsLds; v += 2) {
                    if (v < D) tc0 += pt[v * 3];
                    if (v + 1 < D) tc1 += pt[v * 3 + 3];
                }
                if constexpr (DAMP) {
                    Dc = 0.0;
                    const double *pd = sD + qp * D * 3 + c3;
                    for (int v = 0; v < D; ++v) Dc += pd[v * 3];
                }
                pX = sPX[buf][3 * qp + c3];
            }
            const double tc = tc0 + tc1;
            const double t0 = dpp_move<0x00>(tc), t1 = dpp_move<0x55>(tc), t2 = dpp_move<0xAA>(tc);   // quad_perm [0,0,0,0], [1,1,1,1], [2,2,2,2]
            if (on) {
                const double d = ph - (pv0 * t0 + pv1 * t1 + pv2 * t2);
                const double Xn = pX + d;
                const size_t ix = 3 * (size_t)(p0 + qp) + c3;
                g.pts_new[ix] = Xn;
                if (g.dB) g.dB[ix] = d;
                sXn[qp * 3 + c3] = Xn;
                dl += d * (g.mu * Dc * d + pe);
                dp2 += Dc * d * d;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < nbatch) stream_batch(k + 1, buf ^ 1);   // (behind the ptout loads' wait: see above)
        __builtin_amdgcn_sched_barrier(0);
        wave_lds_sync();
        if (active) {
            const double Xn[3] = { sXn[lp * 3], sXn[lp * 3 + 1], sXn[lp * 3 + 2] };
            const double *cln = sCln + cs * SCL;
            double q[3];
            if (cfg.explicit_centers) {
                const double d0 = Xn[0] - cln[10], d1 = Xn[1] - cln[11], d2 = Xn[2] - cln[12];
#pragma unroll
                for (int r = 0; r < 3; ++r) q[r] = cln[3 * r] * d0 + cln[3 * r + 1] * d1 + cln[3 * r + 2] * d2;
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) q[r] = cln[3 * r] * Xn[0] + cln[3 * r + 1] * Xn[1] + cln[3 * r + 2] * Xn[2] + cln[10 + r];
            }
            const double iz = fast_rcp(q[2]), mx = q[0] * iz, my = q[1] * iz, r2 = mx * mx + my * my;
            const bool md = FULL && cfg.undistort == 2;
            const double gg = cfg.proj_sign * cln[13] * ((!FULL || md) ? 1.0 : 1.0 + cln[14] * r2 + cln[15] * r2 * r2);
            double en0 = xy.x - gg * mx, en1 = xy.y - gg * my;
            if (md) { const double kk = cln[14] * (xy.x * xy.x + xy.y * xy.y); en0 = fma(kk, xy.x, en0); en1 = fma(kk, xy.y, en1); }
            cn += en0 * en0 + en1 * en1;
            pct = fmax(pct, fmax(pct_change_flag(e[0], en0, g.eps5), pct_change_flag(e[1], en1, g.eps5)));
        }
        wave_lds_sync();
    }
    wait_lds_dma();
    cn = wave_sum(cn); dl = wave_sum(dl); dp2 = wave_sum(dp2); pct = wave_max(pct);
    if (lane == 0) { prow[0] = cn; prow[1] = dl; prow[2] = dp2; prow[3] = 0.0; prow[4] = pct; prow[5] = 0.0; prow[6] = 0.0; prow[7] = 0.0; }
}

// PBA's parametrisation at the start of a solve: the caller's (R0, omega) folded into R0 <- exp([omega]x) R0 (cam_lin holds it), omega <- 0
__global__ void k_cam_rebase(int m, int row_stride, double *__restrict__ cam_a, double *__restrict__ cam_lin, double *__restrict__ R0)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double *cl = cam_lin + (size_t)j * kCamLin;
    for (int i = 0; i < 9; ++i) { R0[(size_t)j * 9 + i] = cl[i]; cl[9 + i] = (i % 4 == 0) ? 1.0 : 0.0; }
    double *a = cam_a + (size_t)j * row_stride;
    a[3] = 0.0; a[4] = 0.0; a[5] = 0.0;
}

__global__ void k_rotations(int m, const double *__restrict__ cam_lin, double *__restrict__ R)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < m * 9) R[gid] = cam_lin[(size_t)(gid / 9) * kCamLin + gid % 9];
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
// the accumulation buffer [S_up | camvec | scal] back to zero before a linearisation: an ordinary launch in the engine's stream
// (hipMemsetAsync's fill kernel came with ~6 us of idle GPU in front of it in every trial: profiles/r04_trial_timeline_C3.txt)
__global__ __launch_bounds__(256) void k_zero(double2 *__restrict__ p, size_t n2, double *__restrict__ tail, int ntail)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = i; q < n2; q += stride) p[q] = make_double2(0.0, 0.0);
    if (i < (size_t)ntail) tail[i] = 0.0;
}
void launch_zero(hipStream_t st, double *p, size_t n)
{
    if (n == 0) return;
    const size_t n2 = n / 2;
    const int grid = (int)std::min<size_t>((n2 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(k_zero, dim3(grid), dim3(256), 0, st, reinterpret_cast<double2 *>(p), n2, p + 2 * n2, (int)(n - 2 * n2));
}

void launch_cam_prep(hipStream_t st, int m, int cnp, const ModelCfg &cfg, const double *cam_a, const double *R0,
                     const double *f_fixed, double *cam_lin)
{
    hipLaunchKernelGGL(k_cam_prep, dim3((m + 63) / 64), dim3(64), 0, st, m, cnp, cfg, cam_a, R0, f_fixed, cam_lin);
}

template <int CNP, bool FULL>
static void launch_build_t(hipStream_t st, bool lds_mode, int ntiles, const BuildArgs &a)
{
    if (lds_mode) {
        // both variants walk the same tile range; a workgroup whose tile belongs to the other variant leaves at once
        if (a.has_mfma) hipLaunchKernelGGL((k_build_mfma<CNP, FULL>), dim3(ntiles), dim3(64), 0, st, a);
        if (a.has_uni_fixed) hipLaunchKernelGGL((k_build_uni<CNP, FULL, kUniFixedViews>), dim3(ntiles), dim3(64), 0, st, a);
        if (a.has_uni) hipLaunchKernelGGL((k_build_uni<CNP, FULL, 0>), dim3(ntiles), dim3(64), 0, st, a);
        if (a.has_reg) hipLaunchKernelGGL((k_build<CNP, true, FULL, false>), dim3(ntiles), dim3(64), 0, st, a);
        if (a.has_table) hipLaunchKernelGGL((k_build<CNP, true, FULL, true>), dim3(ntiles), dim3(64), 0, st, a);
    } else hipLaunchKernelGGL((k_build<CNP, false, FULL, false>), dim3(ntiles), dim3(64), 0, st, a);
}

void launch_tile_reduce(hipStream_t st, int cnp, int nb_up, int m, const int *blk_ptr, const int *blk_list, const int *cam_ptr, const int *cam_list,
                        const double *tile_blk, const double *tile_cam, double *S_up, double *camvec)
{
    if (nb_up > 0) hipLaunchKernelGGL(k_tile_reduce_blocks, dim3((nb_up + 3) / 4), dim3(256), 0, st, nb_up, cnp, blk_ptr, blk_list, tile_blk, S_up);
    if (m > 0) hipLaunchKernelGGL(k_tile_reduce_cams, dim3((m + 3) / 4), dim3(256), 0, st, m, cnp, cam_ptr, cam_list, tile_cam, camvec);
}

void launch_build(hipStream_t st, int cnp, bool lds_mode, int ntiles, const BuildArgs &a)
{
    if (ntiles <= 0) return;
    // the lean instantiation unless the problem carries soft point constraints or measurement distortion
    const bool full = a.pt_cw != nullptr || a.cfg.undistort != 0;
    if (cnp == 9) { if (!lds_mode) hipLaunchKernelGGL((k_build<9, false, true, false>), dim3(ntiles), dim3(64), 0, st, a); return; }   // row tiles only (plan.h cam_split)
    if (cnp == 8) { if (full) launch_build_t<8, true>(st, lds_mode, ntiles, a); else launch_build_t<8, false>(st, lds_mode, ntiles, a); }
    else if (cnp == 7) { if (full) launch_build_t<7, true>(st, lds_mode, ntiles, a); else launch_build_t<7, false>(st, lds_mode, ntiles, a); }
    else { if (full) launch_build_t<6, true>(st, lds_mode, ntiles, a); else launch_build_t<6, false>(st, lds_mode, ntiles, a); }
}

// camera-side finish of the linearisation: right-hand side and norms, L^-1 of the diagonal blocks, the scaled matrix; the
// full unscaled matrix only when S_full != nullptr (parity hooks)
constexpr int kScaleBlocksGrid = 2048;   // workgroups of k_scale_blocks: eight wavefronts per SIMD, each walks the blocks in strides
void launch_finalize(hipStream_t st, int cnp, int m, int nb, const unsigned char *fixed, const unsigned char *pfix, double mu, const double *spec, const int *bsr_rowidx, const int *bsr_col,
                     const int *bsr_src, const unsigned char *bsr_trans, const int *bsr_diag, const double *S_up, double *S_full,
                     double *camvec, const double *cam_a, const double *cam_cw, const double *cam_cv, const CamLinks &links,
                     double *E, double *cam_part, double *stage, double *scal, double *Linv, double *S_pcg, bool reduce_now, const PcgStart &ps,
                     const CoarseArgs &co, const int *bsr_row, bool damp_diag)
{
    FinalizeArgs f;
    f.cnp = cnp; f.mu = mu; f.spec = spec; f.fixed = fixed; f.pfix = pfix; f.trans = bsr_trans; f.rowidx = bsr_rowidx; f.col = bsr_col; f.src = bsr_src;
    f.S_up = S_up; f.cam_cw = cam_cw; f.lk = links;
    f.diagU = damp_diag ? camvec + 2 * (size_t)m * kCP : nullptr;
    const long long total = (long long)nb * 64;
    // the camera-side vectors (E, the norms, the penalties): inside k_block_inv -- unless the full matrix (parity hooks) or the sums are wanted right away
    const bool fused = !S_full && !reduce_now;
    if (!fused)
        hipLaunchKernelGGL(k_finalize_vec, dim3((m * kCP + 255) / 256), dim3(256), 0, st, m, cnp, fixed, pfix, camvec, cam_a, cam_cw, cam_cv,
                           links, E, cam_part);
    if (S_full) hipLaunchKernelGGL(k_finalize, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, nb, f, S_full);
    if (reduce_now) launch_reduce_partials(st, (m * kCP + 63) / 64, cam_part, stage, scal + SC_PL2_CAMS, scal + SC_EA_INF);   // else: k_reduce_final
    const FvArgs fv{ fused ? 1 : 0, camvec, cam_a, cam_cw, cam_cv, E, cam_part };
    hipLaunchKernelGGL(k_block_inv, dim3((m + 3) / 4), dim3(256), 0, st, m, bsr_diag, f, Linv, scal, ps, co.ng ? co.Lfac : nullptr, fv);
    hipLaunchKernelGGL(k_scale_blocks, dim3(std::min((ps.nlist + 3) / 4, kScaleBlocksGrid)), dim3(256), 0, st, ps.nlist, reinterpret_cast<const int4 *>(ps.list), f, Linv, S_pcg, ps.layout, ps.layout == 2 && ps.ell ? 1 : 0);
    if (co.ng) {   // the coarse space of this trial's system: Z'SZ, its inverse, S~Z~
        static_assert(kBcSlots == 4, "k_coarse_bc: one wavefront per slot");
        hipLaunchKernelGGL(k_coarse_bc, dim3(m), dim3(256), 0, st, m, bsr_row, bsr_col, (const double *)S_pcg, co);
        hipLaunchKernelGGL(k_coarse_assemble, dim3(co.ng * co.ng), dim3(256), 0, st, m, co);
        hipLaunchKernelGGL(k_coarse_invert, dim3(1), dim3(1024), 0, st, co);
    }
}

void launch_pcg_reinit(hipStream_t st, const PcgArgs &a) { hipLaunchKernelGGL(k_pcg_init, dim3(a.ng), dim3(64), 0, st, a); }

void launch_pcg_init(hipStream_t st, const PcgArgs &a)
{
    if (a.co.ng) { hipLaunchKernelGGL(k_pcg2_start<4>, dim3(a.nwg), dim3(256), 0, st, a); return; }
    if (a.nwg == a.m) hipLaunchKernelGGL(k_pcg_start<1>, dim3(a.nwg), dim3(256), 0, st, a);
    else if (a.wpr == 2) hipLaunchKernelGGL((k_pcg_start<4, 2>), dim3(a.nwg), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(k_pcg_start<4>, dim3(a.nwg), dim3(256), 0, st, a);
    if (a.tot) hipLaunchKernelGGL(k_pcg_totals, dim3(1), dim3(256), 0, st, a, 0);
}

void launch_pcg_iter(hipStream_t st, const PcgArgs &a)
{
    if (a.co.ng) { hipLaunchKernelGGL(k_pcg2_iter, dim3(a.nwg), dim3(320), 0, st, a); return; }
    if (a.nwg == a.m) hipLaunchKernelGGL(k_pcg_iter<1>, dim3(a.nwg), dim3(256), 0, st, a);
    else if (a.wpr == 2) hipLaunchKernelGGL((k_pcg_iter<4, 2>), dim3(a.nwg), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(k_pcg_iter<4>, dim3(a.nwg), dim3(256), 0, st, a);
    if (a.tot) hipLaunchKernelGGL(k_pcg_totals, dim3(1), dim3(256), 0, st, a, (a.it & 1) ^ 1);
}


void launch_pcg_tail(hipStream_t st, const PcgArgs &a) { hipLaunchKernelGGL(k_pcg_tail, dim3(1), dim3(64), 0, st, a); }

// how many workgroups of k_pcg_persist with lds_bytes of dynamic LDS the device holds at once (0: the kernel cannot run like that)
int pcg_persist_capacity(size_t lds_bytes)
{
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_pcg_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg_persist, 256, lds_bytes) != hipSuccess || per_cu <= 0) { (void)hipGetLastError(); return 0; }
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return per_cu * cus;
}

bool launch_pcg_persist(hipStream_t st, const PersistArgs &g, int nwg, size_t lds_bytes)
{
    PersistArgs args = g;
    void *params[] = { &args };
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_pcg_persist), dim3(nwg), dim3(256), params, lds_bytes, st);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

// after a SUM all-reduce over a block of world x 8 slots in which every rank had filled only its own: the maximum over the
// ranks (the values are norms and flags, >= 0); the slots are cleared for the next use
__global__ void k_fold_max(double *__restrict__ slots, int world, double *__restrict__ out8)
{
    const int c = threadIdx.x;
    if (c >= 8) return;
    double v = 0.0;
    for (int r = 0; r < world; ++r) { v = fmax(v, slots[r * 8 + c]); slots[r * 8 + c] = 0.0; }
    out8[c] = v;
}

// The accept test and the damping update of the LM controller (engine.cpp step(); sba_levmar.c:1498-1552) on the trial's
// scalars, with the host's expressions operation for operation (no contraction: the host compares its own damping with
// spec[1] bit for bit before it uses what was built with it).
#pragma clang fp contract(off)
__device__ __forceinline__ void lm_peek(double *__restrict__ s, const PeekArgs &k, double *__restrict__ spec)
{
    const double mu = k.mu;
    const double c_cur = s[SC_CONS_CAM] + s[SC_CONS_PTS];
    const double p_eL2 = s[SC_COST] + k.c_lag + s[SC_LINK];
    const double pdp_eL2 = s[SC_COST_NEW] + c_cur + s[SC_LINK_NEW];
    const double dL = s[SC_DL_CAMS] + s[SC_DL_PTS], dF = p_eL2 - pdp_eL2;
    const bool solved = s[SC_PCG_FINAL] != 0.0 && s[SC_PCG_STATUS] == 0.0 && s[SC_NOT_PD] == 0.0 && s[SC_SINGULAR] == 0.0;
    double go = 0.0, mu_next = mu;
    if (k.variant >= 1) {
        // PBA's controller (engine.cpp step(), SparseBundleCPU.cpp:3892-3948): accept on any decrease, the damping update with its bounds --
        // the same arithmetic as the host's, so that the host finds ITS next damping in spec[1] and adopts the work enqueued ahead
        if (solved && isfinite(pdp_eL2) && dF > 0.0) {
            const double dl = dL > 0.0 ? dL : 0.001 * dF;
            double tmp = 2.0 * dF / dl - 1.0;
            tmp = 1.0 - tmp * tmp * tmp;
            mu_next = mu * ((tmp >= 0.3333333334) ? tmp : 0.3333333334);
            if (mu_next < k.mu_min) mu_next = k.mu_min; else if (k.variant == 1 && k.mu_max > 0.0 && mu_next > k.mu_max) mu_next = k.mu_max;
            go = k.last ? 0.0 : 1.0;
            if (k.stops && sqrt(s[SC_DP2_CAMS] + s[SC_DP2_PTS]) <= k.d_thr) go = 0.0;      // "quit on too small change"
        }
    } else if (solved && isfinite(pdp_eL2) && dL > 0.0 && dF > 0.0) {
        double tmp = 2.0 * dF / dL - 1.0;
        tmp = 1.0 - tmp * tmp * tmp;
        mu_next = mu * ((tmp >= 0.3333333334) ? tmp : 0.3333333334);
        go = 1.0;
        // the stop rules that end the solve with this trial (engine.cpp step(): stops 2, 3, 4, 5, 8): nothing to linearise
        // ahead then.  These only decide whether the work is worth starting; the host's own tests are the ones that count.
        if (k.last) go = 0.0;
        if (k.stops) {
            const double dp_L2 = s[SC_DP2_CAMS] + s[SC_DP2_PTS];
            const double p_L2 = (k.mode == 2 ? 0.0 : s[SC_PL2_CAMS]) + (k.mode == 1 ? 0.0 : s[SC_PL2_PTS] + k.pl2_inactive);
            if (dp_L2 <= k.eps2_sq * p_L2) go = 0.0;
            if (pdp_eL2 - 2.0 * sqrt(p_eL2 * pdp_eL2) < (k.eps4_sq - 1.0) * p_eL2) go = 0.0;
            if (k.mode == 0 && s[SC_MAX_PCT] < k.eps5 && k.itno >= 4) go = 0.0;
            if (pdp_eL2 <= k.eps3_sq) go = 0.0;
        }
    }
    spec[0] = go; spec[1] = mu_next;
    s[SC_SPEC_GO] = go; s[SC_SPEC_MU] = mu_next;
}
#pragma clang fp contract(fast)

__global__ void k_lm_peek(double *__restrict__ s, const PeekArgs k, double *__restrict__ spec)
{
    if (threadIdx.x == 0) lm_peek(s, k, spec);
}

void launch_lm_peek(hipStream_t st, double *scal, const PeekArgs &pk, double *spec) { hipLaunchKernelGGL(k_lm_peek, dim3(1), dim3(64), 0, st, scal, pk, spec); }

void launch_fold_max(hipStream_t st, double *slots, int world, double *out8) { hipLaunchKernelGGL(k_fold_max, dim3(1), dim3(64), 0, st, slots, world, out8); }

void launch_reduce_partials(hipStream_t st, int nrows, const double *part, double *stage, double *out_sum, double *out_max)
{
    if (nrows > 4096) { // two stages: kReduceStage workgroups, then one
        hipLaunchKernelGGL(k_reduce_partials, dim3(kReduceStage), dim3(1024), 0, st, nrows, part, stage, out_sum, out_max);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, st, kReduceStage, stage, stage, out_sum, out_max);
    } else
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, st, nrows, part, stage, out_sum, out_max);
}

void launch_cam_update(hipStream_t st, int m, int cnp, const ModelCfg &cfg, const unsigned char *fixed, double mu, const double *cam_a, const PcgArgs &pa,
                       double *dA, const double *camvec, const double *R0, const double *f_fixed, const CamLinks &links, double *cam_a_new,
                       double *cam_lin_new, double *cam_part, double *R0_new)
{
    if (!cfg.split) hipLaunchKernelGGL(k_cam_update8, dim3((m + 7) / 8), dim3(64), 0, st, m, cnp, cfg, fixed, mu, cam_a, pa, dA, camvec, R0, f_fixed,
                                       links, cam_a_new, cam_lin_new, cam_part, R0_new);
    else hipLaunchKernelGGL(k_cam_update, dim3((m + 63) / 64), dim3(64), 0, st, m, cnp, cfg, fixed, mu, cam_a, pa, dA, camvec, R0, f_fixed,
                            links, cam_a_new, cam_lin_new, cam_part, R0_new);
}

void launch_cam_rebase(hipStream_t st, int m, int row_stride, double *cam_a, double *cam_lin, double *R0)
{
    hipLaunchKernelGGL(k_cam_rebase, dim3((m + 63) / 64), dim3(64), 0, st, m, row_stride, cam_a, cam_lin, R0);
}

// last stage of the back-substitution's reduction (its first stage first when there are many rows) + the camera-side sums + peek
void launch_reduce_final(hipStream_t st, int bs_rows, const double *bs_part, double *stage, double *bs_sum, double *bs_max,
                         int m, int mv, const double *fv_part, const double *cu_part, double *scal, double *spec, const PeekArgs &pk,
                         int bp_rows, const double *bp_part, double *bp_sum, double *bp_max, int fv_rows)
{
    FinalReduceArgs a;
    a.bs_part = bs_part; a.bs_rows = bs_rows;
    a.bp_part = bp_part; a.bp_rows = bp_rows; a.bp_sum = bp_sum; a.bp_max = bp_max;
    double *stage2 = stage + (size_t)kReduceStage * kPartW;      // (the stage buffer holds two sets of kReduceStage rows)
    if (bs_rows > 4096 && bp_rows > 4096) {
        hipLaunchKernelGGL(k_reduce_partials2, dim3(2 * kReduceStage), dim3(1024), 0, st, bs_rows, bs_part, stage, bp_rows, bp_part, stage2);
        a.bs_part = stage; a.bs_rows = kReduceStage; a.bp_part = stage2; a.bp_rows = kReduceStage;
    } else {
        if (bs_rows > 4096) {
            hipLaunchKernelGGL(k_reduce_partials, dim3(kReduceStage), dim3(1024), 0, st, bs_rows, bs_part, stage, bs_sum, bs_max);
            a.bs_part = stage; a.bs_rows = kReduceStage;
        }
        if (bp_rows > 4096) {
            hipLaunchKernelGGL(k_reduce_partials, dim3(kReduceStage), dim3(1024), 0, st, bp_rows, bp_part, stage2, bp_sum, bp_max);
            a.bp_part = stage2; a.bp_rows = kReduceStage;
        }
    }
    a.bs_sum = bs_sum; a.bs_max = bs_max;
    a.fv_part = fv_part; a.fv_rows = fv_rows;                  // k_finalize_vec: a row per wavefront of (virtual camera, parameter) threads; inside k_block_inv: a row per camera
    a.cu_part = cu_part; a.cu_rows = mv == m ? (m + 7) / 8 : (m + 63) / 64;   // k_cam_update8: eight cameras per wavefront (k_cam_update, nine parameters: 64)
    a.scal = scal; a.spec = spec; a.pk = pk;
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, st, a);
}

template <int CNP, bool FULL>
static void launch_backsub_uni_t(hipStream_t st, const BacksubArgs &a)
{
    if (a.uni_fixed) {
        if (a.cfg.damp_diag) hipLaunchKernelGGL((k_backsub_uni<CNP, FULL, true, kUniFixedViews>), dim3(a.uni_ntile), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_backsub_uni<CNP, FULL, false, kUniFixedViews>), dim3(a.uni_ntile), dim3(64), 0, st, a);
    }
    if (a.uni_other || !a.uni_fixed) {
        if (a.cfg.damp_diag) hipLaunchKernelGGL((k_backsub_uni<CNP, FULL, true, 0>), dim3(a.uni_ntile), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_backsub_uni<CNP, FULL, false, 0>), dim3(a.uni_ntile), dim3(64), 0, st, a);
    }
}

void launch_backsub(hipStream_t st, int cnp, const BacksubArgs &a)
{
    const bool full = a.cfg.undistort != 0;
    if (a.uni_ntile > 0 && cnp <= 8) {       // the tiles of one camera set (plan.h mode 4): their own kernel, their own partial rows
        if (cnp == 8) { if (full) launch_backsub_uni_t<8, true>(st, a); else launch_backsub_uni_t<8, false>(st, a); }
        else if (cnp == 7) { if (full) launch_backsub_uni_t<7, true>(st, a); else launch_backsub_uni_t<7, false>(st, a); }
        else { if (full) launch_backsub_uni_t<6, true>(st, a); else launch_backsub_uni_t<6, false>(st, a); }
    }
    if (a.nbatch <= 0) return;
    if (cnp == 9) { hipLaunchKernelGGL((k_backsub<9, true>), dim3(a.ngrid), dim3(64), 0, st, a); return; }
    if (cnp == 8) { if (full) hipLaunchKernelGGL((k_backsub<8, true>), dim3(a.ngrid), dim3(64), 0, st, a); else hipLaunchKernelGGL((k_backsub<8, false>), dim3(a.ngrid), dim3(64), 0, st, a); }
    else if (cnp == 7) { if (full) hipLaunchKernelGGL((k_backsub<7, true>), dim3(a.ngrid), dim3(64), 0, st, a); else hipLaunchKernelGGL((k_backsub<7, false>), dim3(a.ngrid), dim3(64), 0, st, a); }
    else hipLaunchKernelGGL((k_backsub<6, false>), dim3(a.ngrid), dim3(64), 0, st, a);
}

void launch_long_build(hipStream_t st, int cnp, const LongArgs &a)
{
    if (a.nlong <= 0) return;
    if (cnp == 9) hipLaunchKernelGGL((k_long_reduce<9>), dim3(a.nlong), dim3(256), 0, st, a);
    else if (cnp == 8) hipLaunchKernelGGL((k_long_reduce<8>), dim3(a.nlong), dim3(256), 0, st, a);
    else if (cnp == 7) hipLaunchKernelGGL((k_long_reduce<7>), dim3(a.nlong), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_long_reduce<6>), dim3(a.nlong), dim3(256), 0, st, a);
}

template <int CNP>
static void launch_schur_blocks_t(hipStream_t st, const SchurArgs &a)
{
    if (a.ndiag > 0) hipLaunchKernelGGL((k_schur_blocks<CNP, true>), dim3(8 * ((a.ndiag + 7) / 8)), dim3(64), 0, st, a);
    if (a.nwork > 0) {
        static const bool direct = std::getenv("MI_BA_SCHUR_DIRECT") != nullptr;   // probes: every lane fetches its own records
        if (direct) hipLaunchKernelGGL((k_schur_blocks<CNP, false>), dim3(8 * ((a.nwork + 7) / 8)), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_schur_pairs<CNP>), dim3(8 * ((a.nwork + 7) / 8)), dim3(64), 0, st, a);
    }
    if (a.nsuper > 0) hipLaunchKernelGGL((k_schur_super<CNP, 4>), dim3(a.nsuper), dim3(64), 0, st, a);
    if (a.nsuper8 > 0) hipLaunchKernelGGL((k_schur_super<CNP, 8>), dim3(a.nsuper8), dim3(64), 0, st, a);
}

void launch_schur_blocks(hipStream_t st, int cnp, const SchurArgs &a)
{
    if (cnp == 9) {   // nine parameters: every lane fetches its own (288-byte) records
        if (a.ndiag > 0) hipLaunchKernelGGL((k_schur_blocks<9, true>), dim3(8 * ((a.ndiag + 7) / 8)), dim3(64), 0, st, a);
        if (a.nwork > 0) hipLaunchKernelGGL((k_schur_blocks<9, false>), dim3(8 * ((a.nwork + 7) / 8)), dim3(64), 0, st, a);
    } else if (cnp == 8) launch_schur_blocks_t<8>(st, a);
    else if (cnp == 7) launch_schur_blocks_t<7>(st, a);
    else launch_schur_blocks_t<6>(st, a);
}

void launch_long_backsub(hipStream_t st, int cnp, const LongArgs &a)
{
    if (a.nlong <= 0) return;
    if (cnp == 9) hipLaunchKernelGGL((k_long_backsub<9>), dim3(a.nlong), dim3(256), 0, st, a);
    else if (cnp == 8) hipLaunchKernelGGL((k_long_backsub<8>), dim3(a.nlong), dim3(256), 0, st, a);
    else if (cnp == 7) hipLaunchKernelGGL((k_long_backsub<7>), dim3(a.nlong), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_long_backsub<6>), dim3(a.nlong), dim3(256), 0, st, a);
}

void launch_rotations(hipStream_t st, int m, const double *cam_lin, double *R)
{
    hipLaunchKernelGGL(k_rotations, dim3((m * 9 + 255) / 256), dim3(256), 0, st, m, cam_lin, R);
}

} // namespace miba
