// Development tool: what does one scattered dword read cost in HBM traffic?  Reads ONE dword per `stride` bytes of a 400 MB buffer
// (rotating over three buffers so that nothing is cache-resident) with the plain, the non-temporal and the sc1 load, and prints the
// time per pass.  Under `rocprofv3 --pmc FETCH_SIZE` the per-kernel bytes say how much the L2 fetched per touched line.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sector_probe tools/sector_probe.hip && /tmp/sector_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int POLICY>
__device__ __forceinline__ uint32_t load_dword(const uint32_t *p) {
    uint32_t v;
    if constexpr (POLICY == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (POLICY == 1) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// thread i reads dwords at byte offsets (i + k * threads) * stride, k = 0 .. per_thread - 1: consecutive lanes on consecutive strides
template <int POLICY>
__global__ __launch_bounds__(256) void k_strided(const uint32_t *buf, int64_t n_reads, int stride_dwords, uint32_t *sink) {
    const int64_t threads = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // eight independent loads in flight per lane
    for (; i + 7 * threads < n_reads; i += 8 * threads) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t *p = buf + (i + u * threads) * stride_dwords;
            if constexpr (POLICY == 0) v[u] = *p;
            else if constexpr (POLICY == 1) v[u] = __builtin_nontemporal_load(p);
            else asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
        }
        if constexpr (POLICY == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n_reads; i += threads) acc += buf[i * stride_dwords];
    if (acc == 0x12345u) *sink = acc;
}

int main() {
    const size_t bytes = 400u << 20;
    uint32_t *buf[3], *sink;
    for (auto &b : buf) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 1, bytes)); }
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int strides[] = {4, 16, 32, 64, 128, 256, 512};
    for (int policy = 0; policy < 3; ++policy)
        for (int stride : strides) {
            const int64_t n_reads = (int64_t)(bytes / stride);
            const int grid = 2048;
            float best = 1e9f;
            for (int it = 0; it < 9; ++it) {
                CHECK(hipEventRecord(e0));
                if (policy == 0) hipLaunchKernelGGL(k_strided<0>, dim3(grid), dim3(256), 0, 0, buf[it % 3], n_reads, stride / 4, sink);
                else if (policy == 1) hipLaunchKernelGGL(k_strided<1>, dim3(grid), dim3(256), 0, 0, buf[it % 3], n_reads, stride / 4, sink);
                else hipLaunchKernelGGL(k_strided<2>, dim3(grid), dim3(256), 0, 0, buf[it % 3], n_reads, stride / 4, sink);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 3 && ms < best) best = ms;
            }
            std::printf("policy %s stride %4d B: %8.1f us  (%.2f TB/s if 128 B per read, %.2f if 64, %.2f if 32)\n", policy == 0 ? "plain" : (policy == 1 ? "nt   " : "sc1  "),
                        stride, best * 1e3, stride >= 128 ? n_reads * 128.0 / best / 1e9 : bytes / best / 1e9 * 1.0, stride >= 64 ? n_reads * 64.0 / best / 1e9 : bytes / best / 1e9,
                        stride >= 32 ? n_reads * 32.0 / best / 1e9 : bytes / best / 1e9);
        }
    return 0;
}
