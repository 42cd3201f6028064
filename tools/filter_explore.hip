// tools/filter_explore.hip -- development tool: A/B variants of the int32 range-filter kernel and pure
// streaming-read ceilings in ONE process (interleaved rounds, hipEvent timing, 3 rotating 400 MB buffers so
// the 256 MiB Infinity Cache cannot serve the reads).  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -o tools/filter_explore tools/filter_explore.hip && ./tools/filter_explore
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

extern "C" __device__ int wl_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

struct Args {
    const int32_t *data;
    int64_t n_rows;
    int64_t n_tiles;
    int32_t lo, hi;
    uint64_t *bitmap;
    uint32_t *tile_counts;
    unsigned long long *total;
};

__device__ __forceinline__ bool in_closed(int32_t x, int32_t lo, int32_t hi) {
    return ((uint32_t)x - (uint32_t)lo) <= ((uint32_t)hi - (uint32_t)lo);
}

// ---- ceilings -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_read_x4(Args a) { // 16 B / lane streaming read, XOR-reduce
    const int4 *p = (const int4 *)a.data;
    const int64_t n4 = a.n_rows / 4;
    int acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        int4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) a.tile_counts[0] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_read_x4_tile(Args a) { // wave owns U KiB contiguous, U x dwordx4 in flight
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_chunks = a.n_rows / (256 * U);
    int acc = 0;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < n_chunks; c += (int64_t)gridDim.x * 4) {
        const int4 *p = (const int4 *)(a.data + c * 256 * U) + lane;
        int4 v[U];
#pragma unroll
        for (int g = 0; g < U; ++g) v[g] = p[64 * g];
#pragma unroll
        for (int g = 0; g < U; ++g) acc ^= v[g].x ^ v[g].y ^ v[g].z ^ v[g].w;
    }
    if (acc == 0x12345678) a.tile_counts[0] = acc;
}

__global__ __launch_bounds__(256) void k_read_x1_tile(Args a) { // wave owns 4 KiB, 16 x dword in flight
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < a.n_tiles; t += (int64_t)gridDim.x * 4) {
        if ((t + 1) * 1024 > a.n_rows) continue;
        const int32_t *p = a.data + t * 1024 + lane;
        int32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[64 * j];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    if (acc == 0x12345678) a.tile_counts[0] = acc;
}

// ---- V0: row-strided dword loads, ballot == word ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_v0(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long wave_total = 0;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < a.n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row0 = tile * 1024;
        if (row0 + 1024 > a.n_rows) continue; // exploration: full tiles only
        const int32_t *p = a.data + row0 + lane;
        int32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[64 * j];
        int lo = 0, hi = 0;
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t m = __ballot(in_closed(v[j], a.lo, a.hi));
            lo = wl_i32((int)(uint32_t)m, j, lo);
            hi = wl_i32((int)(uint32_t)(m >> 32), j, hi);
            cnt += __popcll(m);
        }
        if (lane < 16) a.bitmap[tile * 16 + lane] = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
        if (lane == 0) a.tile_counts[tile] = cnt;
        wave_total += cnt;
    }
    if (lane == 0) a.total[1 + blockIdx.x * 4 + wave] = wave_total; // per-wave partial, no same-address atomics
}

// ---- V0 ablations: MODE bit0 = nt loads, bit1 = nt stores, bit2 = skip bitmap store, bit3 = skip tile_counts store
template <int MODE>
__global__ __launch_bounds__(256) void k_v0m(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long wave_total = 0;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < a.n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row0 = tile * 1024;
        if (row0 + 1024 > a.n_rows) continue;
        const int32_t *p = a.data + row0 + lane;
        int32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (MODE & 1) ? __builtin_nontemporal_load(p + 64 * j) : p[64 * j];
        int lo = 0, hi = 0;
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t m = __ballot(in_closed(v[j], a.lo, a.hi));
            lo = wl_i32((int)(uint32_t)m, j, lo);
            hi = wl_i32((int)(uint32_t)(m >> 32), j, hi);
            cnt += __popcll(m);
        }
        const uint64_t mine = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
        if (MODE & 4) {
            if (mine == 0x123456789abcdefULL && lane < 16) a.bitmap[tile * 16 + lane] = mine;
        } else if (lane < 16) {
            if (MODE & 2) __builtin_nontemporal_store(mine, a.bitmap + tile * 16 + lane);
            else a.bitmap[tile * 16 + lane] = mine;
        }
        if (!(MODE & 8) && lane == 0) a.tile_counts[tile] = cnt;
        wave_total += cnt;
    }
    if (lane == 0) a.total[1 + blockIdx.x * 4 + wave] = wave_total;
}

// ---- W waves per work-group (nt loads + nt stores): waves per CU between the 8 and 16 a 256-thread block allows
template <int W>
__global__ __launch_bounds__(W * 64) void k_fw(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long wave_total = 0;
    for (int64_t tile = (int64_t)blockIdx.x * W + wave; tile < a.n_tiles; tile += (int64_t)gridDim.x * W) {
        const int64_t row0 = tile * 1024;
        if (row0 + 1024 > a.n_rows) continue;
        const int32_t *p = a.data + row0 + lane;
        int32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
        int lo = 0, hi = 0;
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t m = __ballot(in_closed(v[j], a.lo, a.hi));
            lo = wl_i32((int)(uint32_t)m, j, lo);
            hi = wl_i32((int)(uint32_t)(m >> 32), j, hi);
            cnt += __popcll(m);
        }
        const uint64_t mine = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
        if (lane < 16) __builtin_nontemporal_store(mine, a.bitmap + tile * 16 + lane);
        wave_total += cnt;
    }
    if (lane == 0 && blockIdx.x * W + wave < 4 * 2048) a.total[1 + blockIdx.x * W + wave] = wave_total;
}
// ---- deferred bitmap: the tile lines are parked in LDS and written in one burst after the wave's last load (needs
// n_tiles / (grid * 4) <= 64), to keep HBM read/write turnarounds out of the streaming phase
__global__ __launch_bounds__(256) void k_fdef(Args a) {
    __shared__ uint64_t s_bm[4][64][16]; // 32 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long wave_total = 0;
    int k = 0;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < a.n_tiles; tile += (int64_t)gridDim.x * 4, ++k) {
        const int64_t row0 = tile * 1024;
        if (row0 + 1024 > a.n_rows) { if (lane < 16) s_bm[wave][k & 63][lane] = 0; continue; }
        const int32_t *p = a.data + row0 + lane;
        int32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
        int lo = 0, hi = 0;
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t m = __ballot(in_closed(v[j], a.lo, a.hi));
            lo = wl_i32((int)(uint32_t)m, j, lo);
            hi = wl_i32((int)(uint32_t)(m >> 32), j, hi);
            cnt += __popcll(m);
        }
        if (lane < 16) s_bm[wave][k & 63][lane] = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
        wave_total += cnt;
    }
    // burst: 4 tiles (4 x 16 lanes) per store instruction
    const int n_mine = k;
    for (int q = lane >> 4; q < n_mine; q += 4) {
        const int64_t tile = (int64_t)blockIdx.x * 4 + wave + (int64_t)q * gridDim.x * 4;
        if ((tile + 1) * 1024 <= a.n_rows) __builtin_nontemporal_store(s_bm[wave][q & 63][lane & 15], a.bitmap + tile * 16 + (lane & 15));
    }
    if (lane == 0 && blockIdx.x * 4 + wave < 4 * 2048) a.total[1 + blockIdx.x * 4 + wave] = wave_total;
}
static void l_fdef(const Args &a, int grid, hipStream_t s) { hipLaunchKernelGGL(k_fdef, dim3(grid < 384 ? 384 : grid), dim3(256), 0, s, a); }

#define LAUNCHER_W(fn, W) static void fn(const Args &a, int grid, hipStream_t s) { hipLaunchKernelGGL((k_fw<W>), dim3(grid), dim3(W * 64), 0, s, a); }
LAUNCHER_W(l_fw2, 2)
LAUNCHER_W(l_fw3, 3)
LAUNCHER_W(l_fw4, 4)
LAUNCHER_W(l_fw5, 5)
LAUNCHER_W(l_fw6, 6)
LAUNCHER_W(l_fw8, 8)

// ---- V3: nt loads + nt stores, T tiles (T x 16 dword loads) in flight per wave
template <int T>
__global__ __launch_bounds__(256) void k_v3(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long wave_total = 0;
    const int64_t n_groups = a.n_tiles / T;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t row0 = grp * 1024 * T;
        if (row0 + 1024 * T > a.n_rows) continue;
        const int32_t *p = a.data + row0 + lane;
        int32_t v[16 * T];
#pragma unroll
        for (int j = 0; j < 16 * T; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int lo = 0, hi = 0;
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint64_t m = __ballot(in_closed(v[t * 16 + j], a.lo, a.hi));
                lo = wl_i32((int)(uint32_t)m, j, lo);
                hi = wl_i32((int)(uint32_t)(m >> 32), j, hi);
                cnt += __popcll(m);
            }
            const uint64_t mine = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
            const int64_t tile = grp * T + t;
            if (lane < 16) __builtin_nontemporal_store(mine, a.bitmap + tile * 16 + lane);
            if (lane == 0) a.tile_counts[tile] = cnt;
            wave_total += cnt;
        }
    }
    if (lane == 0) a.total[1 + blockIdx.x * 4 + wave] = wave_total;
}

__global__ __launch_bounds__(256) void k_read_x1_nt(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < a.n_tiles; t += (int64_t)gridDim.x * 4) {
        if ((t + 1) * 1024 > a.n_rows) continue;
        const int32_t *p = a.data + t * 1024 + lane;
        int32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    if (acc == 0x12345678) a.tile_counts[0] = acc;
}

__global__ __launch_bounds__(256) void k_read_x4_nt(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_chunks = a.n_rows / 1024;
    int acc = 0;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < n_chunks; c += (int64_t)gridDim.x * 4) {
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i *p = (const v4i *)(a.data + c * 1024) + lane;
        v4i v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = __builtin_nontemporal_load(p + 64 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc ^= v[g].x ^ v[g].y ^ v[g].z ^ v[g].w;
    }
    if (acc == 0x12345678) a.tile_counts[0] = acc;
}

// ---- V4: nt loads/stores, each wave owns a CONTIGUOUS range of tiles (needed for wave-local dense output staging)
__global__ __launch_bounds__(256) void k_v4_contig(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t full = a.n_rows / 1024;
    const int64_t per = (full + n_waves - 1) / n_waves;
    const int64_t t0 = wid * per, t1 = (t0 + per < full) ? t0 + per : full;
    unsigned long long wave_total = 0;
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int32_t *p = a.data + tile * 1024 + lane;
        int32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
        int lo = 0, hi = 0;
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t m = __ballot(in_closed(v[j], a.lo, a.hi));
            lo = wl_i32((int)(uint32_t)m, j, lo);
            hi = wl_i32((int)(uint32_t)(m >> 32), j, hi);
            cnt += __popcll(m);
        }
        const uint64_t mine = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
        if (lane < 16) __builtin_nontemporal_store(mine, a.bitmap + tile * 16 + lane);
        wave_total += cnt;
    }
    if (lane == 0) a.total[1 + blockIdx.x * 4 + wave] = wave_total;
}

// ---- V1: dwordx4 loads (16 B/lane) + in-register transpose ------------------------------------------------
// load g covers rows g*256 + 4*lane + k.  Word (4g + (lane>>4)) bit 4*(lane&15)+k.  Each lane builds a nibble,
// shifts it to 4*(lane&7), OR-reduces over its 8-lane group with DPP -> one bitmap dword per 8 lanes.
__device__ __forceinline__ uint32_t or_reduce8(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true); // row_half_mirror
    return v;
}

template <int NT>
__global__ __launch_bounds__(256) void k_v1(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sh = 4 * (lane & 7);
    unsigned long long lane_total = 0;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < a.n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row0 = tile * 1024;
        if (row0 + 1024 > a.n_rows) continue;
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i *p = (const v4i *)(a.data + row0) + lane;
        v4i v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = NT ? __builtin_nontemporal_load(p + 64 * g) : p[64 * g];
        uint32_t r[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t nib = (in_closed(v[g].x, a.lo, a.hi) ? 1u : 0u) | (in_closed(v[g].y, a.lo, a.hi) ? 2u : 0u) |
                           (in_closed(v[g].z, a.lo, a.hi) ? 4u : 0u) | (in_closed(v[g].w, a.lo, a.hi) ? 8u : 0u);
            r[g] = or_reduce8(nib << sh);
        }
        // dword index within the tile's 32 dwords: g*8 + (lane>>3); lane (lane&7)==g' stores r[g'] for g' < 4
        const int gsel = lane & 7;
        uint32_t mine = gsel == 0 ? r[0] : gsel == 1 ? r[1] : gsel == 2 ? r[2] : r[3];
        uint32_t *out = (uint32_t *)(a.bitmap + tile * 16);
        if (gsel < 4) {
            if (NT) __builtin_nontemporal_store(mine, out + gsel * 8 + (lane >> 3));
            else out[gsel * 8 + (lane >> 3)] = mine;
            lane_total += __popc(mine);
        }
        // per-tile count: sum over the 32 storing lanes
        uint32_t c = gsel < 4 ? __popc(mine) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
        if (lane == 0) a.tile_counts[tile] = c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lane_total += __shfl_xor(lane_total, d);
    if (lane == 0) a.total[1 + blockIdx.x * 4 + wave] = lane_total;
}

// ---- V2: like V1 but 2 tiles (8 KiB) in flight per wave -------------------------------------------------
__global__ __launch_bounds__(256) void k_v2(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sh = 4 * (lane & 7);
    const int gsel = lane & 7;
    unsigned long long lane_total = 0;
    const int64_t n_pairs = a.n_tiles / 2;
    for (int64_t pair = (int64_t)blockIdx.x * 4 + wave; pair < n_pairs; pair += (int64_t)gridDim.x * 4) {
        const int64_t row0 = pair * 2048;
        if (row0 + 2048 > a.n_rows) continue;
        const int4 *p = (const int4 *)(a.data + row0) + lane;
        int4 v[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = p[64 * g];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint32_t r[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int4 x = v[t * 4 + g];
                uint32_t nib = (in_closed(x.x, a.lo, a.hi) ? 1u : 0u) | (in_closed(x.y, a.lo, a.hi) ? 2u : 0u) |
                               (in_closed(x.z, a.lo, a.hi) ? 4u : 0u) | (in_closed(x.w, a.lo, a.hi) ? 8u : 0u);
                r[g] = or_reduce8(nib << sh);
            }
            uint32_t mine = gsel == 0 ? r[0] : gsel == 1 ? r[1] : gsel == 2 ? r[2] : r[3];
            const int64_t tile = pair * 2 + t;
            uint32_t *out = (uint32_t *)(a.bitmap + tile * 16);
            uint32_t c = 0;
            if (gsel < 4) {
                out[gsel * 8 + (lane >> 3)] = mine;
                c = __popc(mine);
                lane_total += c;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
            if (lane == 0) a.tile_counts[tile] = c;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lane_total += __shfl_xor(lane_total, d);
    if (lane == 0) a.total[1 + blockIdx.x * 4 + wave] = lane_total;
}

// ---------------------------------------------------------------------------------------------------------
struct Variant {
    const char *name;
    void (*launch)(const Args &, int grid, hipStream_t);
};

#define LAUNCHER(fn, kern) static void fn(const Args &a, int grid, hipStream_t s) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, a); }
LAUNCHER(l_read_x4, k_read_x4)
LAUNCHER(l_read_x4_t4, k_read_x4_tile<4>)
LAUNCHER(l_read_x4_t8, k_read_x4_tile<8>)
LAUNCHER(l_read_x1, k_read_x1_tile)
LAUNCHER(l_v0, k_v0)
LAUNCHER(l_v1, k_v1<0>)
LAUNCHER(l_v4, k_v4_contig)
LAUNCHER(l_v1nt, k_v1<1>)
LAUNCHER(l_v3_1, k_v3<1>)
LAUNCHER(l_v3_2, k_v3<2>)
LAUNCHER(l_rx1nt, k_read_x1_nt)
LAUNCHER(l_rx4nt, k_read_x4_nt)
LAUNCHER(l_v0_ntl, k_v0m<1>)
LAUNCHER(l_v0_nts, k_v0m<2>)
LAUNCHER(l_v0_ntls, k_v0m<3>)
LAUNCHER(l_v0_nobm, k_v0m<4>)
LAUNCHER(l_v0_nobm_notc, k_v0m<12>)
LAUNCHER(l_v0_notc, k_v0m<8>)
LAUNCHER(l_v2, k_v2)

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000LL;
    const int rounds = argc > 2 ? atoi(argv[2]) : 41;
    const int NB = 3;
    CHECK(hipSetDevice(0));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    std::vector<int32_t> host((size_t)n);
    int32_t *d[NB];
    for (int b = 0; b < NB; ++b) {
        uint64_t x = 0x9E3779B97F4A7C15ULL * (b + 1);
        for (int64_t i = 0; i < n; ++i) {
            x += 0x9E3779B97F4A7C15ULL;
            uint64_t z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            host[(size_t)i] = (int32_t)(z >> 34);
        }
        CHECK(hipMalloc(&d[b], (size_t)n * 4 + 4096));
        CHECK(hipMemcpy(d[b], host.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    const int64_t n_tiles = (n + 1023) / 1024;
    uint64_t *bitmap;
    uint32_t *tile_counts;
    unsigned long long *total;
    CHECK(hipMalloc(&bitmap, (size_t)n_tiles * 16 * 8));
    CHECK(hipMalloc(&tile_counts, (size_t)n_tiles * 4));
    const size_t total_slots = 1 + 4 * (size_t)((n_tiles + 3) / 4 + 1);
    CHECK(hipMalloc(&total, 8 * total_slots));

    // reference bitmap from the last host buffer (b = NB-1) for V-correctness
    const int32_t lo = (1 << 28) + 1, hi = 3 * (1 << 28) - 1;
    const int64_t full_tiles = n / 1024;
    std::vector<uint64_t> ref((size_t)full_tiles * 16, 0);
    for (int64_t i = 0; i < full_tiles * 1024; ++i)
        if (host[(size_t)i] >= lo && host[(size_t)i] <= hi) ref[(size_t)(i >> 6)] |= 1ULL << (i & 63);

    std::vector<Variant> vars = {
        {"read_x1_nt", l_rx1nt}, {"f_x1_nt", l_v0_ntls}, {"fw2 (128 thr)", l_fw2}, {"fw3 (192 thr)", l_fw3}, {"fw4 (256 thr)", l_fw4},
        {"fw5 (320 thr)", l_fw5}, {"fw6 (384 thr)", l_fw6}, {"fw8 (512 thr)", l_fw8}, {"v_deferred_bitmap", l_fdef},
    };
    std::vector<int> grids = {256, 512, 768, 1024, 2048};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    // correctness of the filter variants
    for (auto &v : vars) {
        if (v.name[0] != 'v' || strstr(v.name, "no_")) continue;
        Args a{d[NB - 1], n, n_tiles, lo, hi, bitmap, tile_counts, total};
        CHECK(hipMemset(bitmap, 0, (size_t)n_tiles * 16 * 8));
        CHECK(hipMemset(total, 0, 8 * total_slots));
        v.launch(a, 2048, s);
        CHECK(hipStreamSynchronize(s));
        std::vector<uint64_t> got((size_t)full_tiles * 16);
        CHECK(hipMemcpy(got.data(), bitmap, got.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t = 0;
        {
            std::vector<unsigned long long> parts(1 + 4 * 2048);
            CHECK(hipMemcpy(parts.data(), total, parts.size() * 8, hipMemcpyDeviceToHost));
            for (size_t i = 1; i < parts.size(); ++i) t += parts[i];
        }
        size_t bad = 0;
        unsigned long long pc = 0;
        for (size_t i = 0; i < got.size(); ++i) { bad += got[i] != ref[i]; pc += __builtin_popcountll(ref[i]); }
        printf("check %-18s mismatching words: %zu  count %llu (ref %llu)\n", v.name, bad, t, pc);
    }

    printf("\n%-18s", "variant \\ grid");
    for (int g : grids) printf(" %10d", g);
    printf("   (us median/min; GB/s at best median)\n");
    std::vector<std::vector<std::vector<float>>> times(vars.size(), std::vector<std::vector<float>>(grids.size()));
    for (int r = 0; r < rounds + 2; ++r) {
        for (size_t vi = 0; vi < vars.size(); ++vi) {
            for (size_t gi = 0; gi < grids.size(); ++gi) {
                Args a{d[(r + vi + gi) % NB], n, n_tiles, lo, hi, bitmap, tile_counts, total};
                CHECK(hipEventRecord(e0, s));
                vars[vi].launch(a, grids[gi], s);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) times[vi][gi].push_back(ms * 1000.f);
            }
        }
    }
    for (size_t vi = 0; vi < vars.size(); ++vi) {
        printf("%-18s", vars[vi].name);
        float best = 1e30f;
        for (size_t gi = 0; gi < grids.size(); ++gi) {
            auto &t = times[vi][gi];
            std::sort(t.begin(), t.end());
            const float med = t[t.size() / 2];
            best = std::min(best, med);
            printf(" %5.1f/%4.1f", med, t[0]);
        }
        printf("   %.0f GB/s read (%.1f%% of 8 TB/s)\n", n * 4.0 / best / 1e3, n * 4.0 / best / 1e3 / 80.0);
    }
    return 0;
}
