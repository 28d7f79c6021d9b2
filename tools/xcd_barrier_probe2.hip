// Probe 2: what would one iteration of a persistent CG cost?  Per round every workgroup publishes 2 partials and 96 doubles of "its rows"
// (3 vectors x 4 rows x 8), passes a grid barrier (flat counter, or a two-level tree: groups of G workgroups, the last arrival of a group
// reports to the root, the last of all bumps a generation word everybody polls), then reads ALL partials and a window of W neighbouring
// workgroups' rows with agent-scope loads.   hipcc --offload-arch=gfx950 -O2 -o probe2 tools/xcd_barrier_probe2.hip ; ./probe2 <n> <tree G or 0> <window>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

struct Sync { u64 *flat; u64 *grp; u64 *root; u64 *gen; int *abort_flag; };

__device__ __forceinline__ bool spin_until(u64 *p, u64 target, int *abort_flag)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > 5000000ll || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

__device__ __forceinline__ bool grid_barrier(const Sync &s, int G, u64 round /* 1.. */)
{
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const int n = gridDim.x, g = blockIdx.x;
        if (G <= 0) {
            __hip_atomic_fetch_add(s.flat, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = spin_until(s.flat, (u64)n * round, s.abort_flag);
        } else {
            const int grp = g / G, ng = (n + G - 1) / G, gsz = min(G, n - grp * G);
            const u64 v = __hip_atomic_fetch_add(s.grp + grp * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v + 1 == (u64)gsz * round) {       // last of its group this round
                const u64 r = __hip_atomic_fetch_add(s.root, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r + 1 == (u64)ng * round) __hip_atomic_store(s.gen, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ok = spin_until(s.gen, round, s.abort_flag);
        }
    }
    return __syncthreads_and(ok);
}

__global__ __launch_bounds__(256) void k_iter(Sync s, int G, double *part, double *rows, double *out, int rounds, int window)
{
    const int g = blockIdx.x, n = gridDim.x, t = threadIdx.x;
    __shared__ double ws[4];
    __shared__ double win[64 * 96];
    double acc = 0.0;
    for (int it = 0; it < rounds; ++it) {
        double *pc = part + (size_t)(it & 1) * 2 * n, *rc = rows + (size_t)(it & 1) * 96 * n;
        if (t < 2) __hip_atomic_store(pc + 2 * g + t, (double)(it + 1) * (g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < 96) __hip_atomic_store(rc + (size_t)96 * g + t, (double)(it + g + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(s, G, (u64)it + 1)) { if (t == 0) out[g] = -1.0; return; }
        double sum = 0.0;
        for (int i = t; i < 2 * n; i += blockDim.x) sum += __hip_atomic_load(pc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = t; i < window * 96; i += blockDim.x) {
            const int nb = (g + n - window / 2 + i / 96) % n;
            win[i] = __hip_atomic_load(rc + (size_t)96 * nb + i % 96, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
        if ((t & 63) == 0) ws[t >> 6] = sum;
        __syncthreads();
        double w = 0.0;
        for (int i = t; i < window * 96; i += blockDim.x) w += win[i];
        acc += ws[0] + ws[1] + ws[2] + ws[3] + 1e-300 * w;
        __syncthreads();
    }
    if (t == 0) out[g] = acc;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256, G = argc > 2 ? atoi(argv[2]) : 0, window = argc > 3 ? atoi(argv[3]) : 12, rounds = 2000;
    char *buf = nullptr;
    const size_t bytes = 65536 + sizeof(double) * (2 * 2 * n + 2 * 96 * (size_t)n);
    CHECK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocFinegrained));
    double *out = nullptr;
    CHECK(hipMalloc((void **)&out, sizeof(double) * n));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(buf, 0, bytes));
        Sync s;
        s.flat = (u64 *)buf; s.root = (u64 *)(buf + 128); s.gen = (u64 *)(buf + 256); s.abort_flag = (int *)(buf + 384); s.grp = (u64 *)(buf + 4096);
        double *part = (double *)(buf + 65536), *rows = part + 2 * 2 * n;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        int Gv = G, rv = rounds, wv = window;
        void *args[] = { &s, &Gv, &part, &rows, &out, &rv, &wv };
        CHECK(hipEventRecord(e0, 0));
        CHECK(hipLaunchCooperativeKernel((const void *)k_iter, dim3(n), dim3(256), args, 0, 0));
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> h(n);
        CHECK(hipMemcpy(h.data(), out, sizeof(double) * n, hipMemcpyDeviceToHost));
        double want = 0.0;
        for (int it = 0; it < rounds; ++it) want += 2.0 * (double)(it + 1) * n * (n + 1) / 2.0;
        int bad = 0, to = 0;
        for (int g = 0; g < n; ++g) { if (h[g] < 0) ++to; else if (h[g] != want) ++bad; }
        if (rep) std::printf("%d workgroups, %s, window %d: %.2f us per iteration, %d wrong sums, %d timeouts\n", n, G > 0 ? "tree" : "flat", window, 1000.0 * ms / rounds, bad, to);
    }
    return 0;
}
