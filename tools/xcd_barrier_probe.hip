// Probe: what does a grid-wide barrier across the 8 XCDs cost when the flag lives in fine-grained device memory and no
// L2 write-back fence is used (relaxed agent-scope atomics + s_waitcnt)?   hipcc --offload-arch=gfx950 -O2 -o probe tools/xcd_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned long long *ctr, unsigned long long target, int *abort_flag)
{
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's stores are acknowledged
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > 5000000ll || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {   // 50 ms at 100 MHz
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    ok = __syncthreads_and(ok);
    return ok;
}

// every workgroup publishes a value per round; after the barrier every workgroup sums all of them; checks visibility
__global__ __launch_bounds__(256) void k_probe(unsigned long long *ctr, int *abort_flag, double *slots, double *out, int rounds, int mode)
{
    const int g = blockIdx.x, n = gridDim.x;
    double acc = 0.0;
    for (int it = 0; it < rounds; ++it) {
        double *cur = slots + (size_t)(it & 1) * n;
        if (threadIdx.x == 0) __hip_atomic_store(cur + g, (double)(it + 1) * (g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(ctr, (unsigned long long)n * (it + 1), abort_flag)) { if (threadIdx.x == 0) out[g] = -1.0; return; }
        if (mode == 1) {   // read everybody's value
            double s = 0.0;
            for (int i = threadIdx.x; i < n; i += blockDim.x) s += __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
            __shared__ double ws[4];
            if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
            __syncthreads();
            acc += ws[0] + ws[1] + ws[2] + ws[3];
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) out[g] = acc;
}

int main(int argc, char **argv)
{
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int n = argc > 1 ? atoi(argv[1]) : ncu, rounds = 2000;
    for (int fine = 1; fine >= 0; --fine) {
        char *buf = nullptr;
        const size_t bytes = 4096 + sizeof(double) * 2 * n;
        if (fine) CHECK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocFinegrained));
        else CHECK(hipMalloc((void **)&buf, bytes));
        double *out = nullptr;
        CHECK(hipMalloc((void **)&out, sizeof(double) * n));
        for (int mode = 0; mode < 2; ++mode) {
            CHECK(hipMemset(buf, 0, bytes));
            unsigned long long *ctr = (unsigned long long *)buf;
            int *abort_flag = (int *)(buf + 64);
            double *slots = (double *)(buf + 4096);
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            void *args[] = { &ctr, &abort_flag, &slots, &out, (void *)&rounds, &mode };
            CHECK(hipEventRecord(e0, 0));
            CHECK(hipLaunchCooperativeKernel((const void *)k_probe, dim3(n), dim3(256), args, 0, 0));
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<double> h(n);
            CHECK(hipMemcpy(h.data(), out, sizeof(double) * n, hipMemcpyDeviceToHost));
            double want = 0.0;
            for (int it = 0; it < rounds; ++it) want += (double)(it + 1) * n * (n + 1) / 2.0;
            int bad = 0, timed_out = 0;
            for (int g = 0; g < n; ++g) { if (h[g] < 0) ++timed_out; else if (mode == 1 && h[g] != want) ++bad; }
            std::printf("%s memory, %d workgroups, mode %d: %.2f us per barrier round, %d stale sums, %d timeouts\n", fine ? "fine-grained" : "coarse", n, mode,
                        1000.0 * ms / rounds, bad, timed_out);
        }
        CHECK(hipFree(buf)); CHECK(hipFree(out));
    }
    return 0;
}
