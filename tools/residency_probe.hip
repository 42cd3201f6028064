// Development tool: how many work-groups of a given shape (threads, LDS bytes, VGPR pressure aside) does an MI355X CU
// really hold at once?  Every work-group stamps its start on the 100 MHz clock and spins ~40 us; the work-groups that start
// within the first 5 us were resident together.  (The occupancy query said 2 per CU for 320 threads / 74 KB of LDS; the
// timeline of k_filter_project showed 1.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void probe(unsigned long long *stamps, int spin_ticks) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        stamps[blockIdx.x] = t0;
        lds[0] = 1;
    }
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
}

// the same with the register footprint of the real kernel: every lane keeps NV live values across the spin
template <int NV>
__global__ void probe_regs(unsigned long long *stamps, int spin_ticks, const int *src, int *sink) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    int v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = src[threadIdx.x + 64 * i];
    if (threadIdx.x == 0) {
        stamps[blockIdx.x] = t0;
        lds[0] = 1;
    }
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    int acc = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc ^= v[i];
    if (acc == 0x12345) *sink = acc;
}

template <int NV>
static void sweep_regs(unsigned long long *d, int *src, int *sink) {
    const int grid = 1024;
    std::vector<unsigned long long> h(grid);
    hipFuncSetAttribute((const void *)probe_regs<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int threads : {256, 320}) {
        for (int kb : {48, 60, 72}) {
            int occ = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe_regs<NV>, threads, (size_t)kb * 1024);
            hipMemset(d, 0, grid * sizeof(unsigned long long));
            hipLaunchKernelGGL(probe_regs<NV>, dim3(grid), dim3(threads), (size_t)kb * 1024, 0, d, 4000, src, sink);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            const unsigned long long t0 = *std::min_element(h.begin(), h.end());
            int early = 0;
            for (auto t : h) early += (t - t0) < 500;
            std::printf("live values %3d threads %3d lds %2d KB: occupancy query %d per CU, resident at once %4d (%.2f per CU)\n", NV, threads, kb, occ, early, early / 256.0);
        }
    }
}

int main() {
    {
        unsigned long long *d = nullptr;
        int *src = nullptr, *sink = nullptr;
        hipMalloc(&d, 1024 * sizeof(unsigned long long));
        hipMalloc(&src, 1 << 20);
        hipMemset(src, 0, 1 << 20);
        hipMalloc(&sink, 64);
        sweep_regs<60>(d, src, sink);
        sweep_regs<90>(d, src, sink);
        sweep_regs<110>(d, src, sink);
        sweep_regs<120>(d, src, sink);
        sweep_regs<150>(d, src, sink);
    }
    unsigned long long *d = nullptr;
    const int grid = 1024;
    hipMalloc(&d, grid * sizeof(unsigned long long));
    std::vector<unsigned long long> h(grid);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int threads : {256, 320, 384, 512}) {
        for (int kb : {16, 32, 48, 52, 56, 60, 64, 68, 72, 76, 80}) {
            int occ = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, threads, (size_t)kb * 1024);
            hipMemset(d, 0, grid * sizeof(unsigned long long));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), (size_t)kb * 1024, 0, d, 4000);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            const unsigned long long t0 = *std::min_element(h.begin(), h.end());
            int early = 0;
            for (auto t : h) early += (t - t0) < 500;
            std::printf("threads %3d lds %2d KB: occupancy query %d per CU, resident at once %4d (%.2f per CU)\n", threads, kb, occ, early, early / 256.0);
        }
    }
    return 0;
}
