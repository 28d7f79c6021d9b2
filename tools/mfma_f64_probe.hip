// mfma_f64_probe.hip -- what the f64 matrix pipe of gfx950 is worth next to the f64 vector pipe (a probe, not product code).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_probe tools/mfma_f64_probe.hip && /tmp/mfma_f64_probe
// With one workgroup per CU (256 threads = one wave per SIMD, or 512 = two per SIMD):
//   mfma16   v_mfma_f64_16x16x4_f64 back to back, 4 independent accumulators          (2048 flop per instruction)
//   mfma4    v_mfma_f64_4x4x4_4b_f64 back to back, 4 independent accumulators         ( 512 flop per instruction)
//   valu     v_fma_f64, 16 independent chains                                          ( 128 flop per instruction)
//   both     two waves per SIMD: waves 0-3 issue mfma16, waves 4-7 issue valu -- do the two pipes overlap?
// Cycles are s_memtime ticks per instruction seen by one wave; TFLOP/s from the wall clock of the whole launch.
// The numbers decide whether recasting k_build's rank-2 pair products (phase 5) as MFMA tiles can pay.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 8192;

template <int MODE>   // 0 mfma16, 1 mfma4, 2 valu, 3 both
__global__ __launch_bounds__(512) void k_probe(double *out, unsigned long long *cyc)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = MODE == 0 || MODE == 1 || (MODE == 3 && wave < 4);
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    d4 c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = i * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        if (MODE == 1) {
            for (int it = 0; it < kIters; ++it) {
                s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
            }
        } else {
            for (int it = 0; it < kIters; ++it) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int threads, double flop_per_instr_mfma, double *out, unsigned long long *cyc)
{
    const int grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(threads), 0, 0, out, cyc);   // warm up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(threads), 0, 0, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    auto med = [&](int w0, int w1) { std::vector<double> c; for (int b = 0; b < grid; ++b) for (int w = w0; w < w1; ++w) c.push_back((double)h[b * 8 + w]); std::sort(c.begin(), c.end()); return c[c.size() / 2]; };
    if (MODE == 3) {
        const double cm = med(0, 4) / (4.0 * kIters), cv = med(4, 8) / (16.0 * kIters);
        const double fl = grid * 4.0 * kIters * (4 * 2048.0 + 16 * 128.0);
        std::printf("%-8s %d thr: mfma16 wave %.1f cyc/instr, valu wave %.2f cyc/instr, %.3f ms, %.1f TFLOP/s in all (mfma %.1f + valu %.1f)\n", name, threads, cm, cv, ms,
                    fl / (ms * 1e-3) / 1e12, grid * 4.0 * kIters * 4 * 2048.0 / (ms * 1e-3) / 1e12, grid * 4.0 * kIters * 16 * 128.0 / (ms * 1e-3) / 1e12);
    } else {
        const double per = MODE == 2 ? 16.0 : 4.0;
        const double c = med(0, waves) / (per * kIters);
        const double fl = (double)grid * waves * kIters * per * flop_per_instr_mfma;
        std::printf("%-8s %d thr: %.2f cyc/instr per wave, %.3f ms, %.1f TFLOP/s\n", name, threads, c, ms, fl / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
    for (int threads : { 256, 512 }) {
        run<0>("mfma16", threads, 2048.0, out, cyc);
        run<1>("mfma4", threads, 512.0, out, cyc);
        run<2>("valu", threads, 128.0, out, cyc);
    }
    run<3>("both", 512, 0, out, cyc);
    return 0;
}
