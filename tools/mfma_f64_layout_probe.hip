// mfma_f64_layout_probe.hip -- which lane holds which element of A, B and D in v_mfma_f64_4x4x4_4b_f64 and v_mfma_f64_16x16x4_f64
// (a probe, not product code).   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_layout tools/mfma_f64_layout_probe.hip && /tmp/mfma_layout
// One-hot inputs: A = 1 in lane la only, B = 1 in lane lb only; the lanes (and registers) of D that come out non-zero say which
// (A lane, B lane) pairs meet and where their product lands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_probe4(double *out)   // out[la][lb][lane]
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[((size_t)la * 64 + lb) * 64 + lane] = d;
        }
}

__global__ void k_probe16(double *out)  // out[la][lb][lane][4]
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double4_t c = { 0.0, 0.0, 0.0, 0.0 };
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r) out[(((size_t)la * 64 + lb) * 64 + lane) * 4 + r] = c[r];
        }
}

int main()
{
    double *d4 = nullptr, *d16 = nullptr;
    const size_t n4 = 64 * 64 * 64, n16 = n4 * 4;
    if (hipMalloc(&d4, n4 * 8) != hipSuccess || hipMalloc(&d16, n16 * 8) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(k_probe4, dim3(1), dim3(64), 0, 0, d4);
    hipLaunchKernelGGL(k_probe16, dim3(1), dim3(64), 0, 0, d16);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
    std::vector<double> h4(n4), h16(n16);
    hipMemcpy(h4.data(), d4, n4 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(h16.data(), d16, n16 * 8, hipMemcpyDeviceToHost);
    // 4x4x4_4b: for every A lane, the B lanes it meets and the D lane of the product
    std::printf("== v_mfma_f64_4x4x4_4b_f64: A lane la meets B lane lb -> D lane\n");
    for (int la = 0; la < 64; ++la) {
        std::printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h4[((size_t)la * 64 + lb) * 64 + l] != 0.0) std::printf(" B%02d->D%02d", lb, l);
        std::printf("\n");
    }
    // hypothesis check: A lane = 16*blk + 4*k + i ; B lane = 16*blk + 4*k + j ; D lane = 16*blk + 4*i + j  (and the transposed variants)
    const char *names[4] = { "A(blk,k,i) B(blk,k,j) D(blk,i,j)", "A(blk,k,i) B(blk,k,j) D(blk,j,i)", "A(blk,i,k) B(blk,j,k) D(blk,i,j)", "A(blk,i,k) B(blk,j,k) D(blk,j,i)" };
    for (int hyp = 0; hyp < 4; ++hyp) {
        int bad = 0;
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb) {
                const int ba = la >> 4, bb = lb >> 4;
                int ka, ia, kb, jb;
                if (hyp < 2) { ka = (la >> 2) & 3; ia = la & 3; kb = (lb >> 2) & 3; jb = lb & 3; }
                else { ia = (la >> 2) & 3; ka = la & 3; jb = (lb >> 2) & 3; kb = lb & 3; }
                const bool meet = ba == bb && ka == kb;
                const int dl = (hyp & 1) ? 16 * ba + 4 * jb + ia : 16 * ba + 4 * ia + jb;
                for (int l = 0; l < 64; ++l) {
                    const double want = (meet && l == dl) ? 1.0 : 0.0;
                    if (h4[((size_t)la * 64 + lb) * 64 + l] != want) ++bad;
                }
            }
        std::printf("hypothesis %d [%s]: %d mismatches\n", hyp, names[hyp], bad);
    }
    // 16x16x4: guide says A[l&15][k=l>>4], B[k=l>>4][l&15], D col = lane&15, row = (lane>>4) + 4*reg
    {
        int bad = 0;
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb) {
                const int i = la & 15, ka = la >> 4, j = lb & 15, kb = lb >> 4;
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 4; ++r) {
                        const int col = l & 15, row = (l >> 4) + 4 * r;
                        const double want = (ka == kb && row == i && col == j) ? 1.0 : 0.0;
                        if (h16[(((size_t)la * 64 + lb) * 64 + l) * 4 + r] != want) ++bad;
                    }
            }
        std::printf("16x16x4 guide layout (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4*reg col=l&15): %d mismatches\n", bad);
    }
    return 0;
}
