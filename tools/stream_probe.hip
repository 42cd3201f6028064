// Development tool (round 5): what bounds the STREAMING side of k_filter_project?  The product kernel streams C3's two columns
// (int32 + int8: 5 KiB per 1024-row tile, 500 MB per 100 M rows) in 92 us with eight streaming waves per CU and one tile of loads
// in flight per wave; k_filter_tile reads the same bytes in 74 us.  This probe takes the streamers out of the kernel -- same tile
// mapping (a work-group owns spans of S * P consecutive tiles dealt round-robin, each wave a range of P tiles), same loads
// (16 row-strided global_load_dword nt + one global_load_dwordx4 nt per tile), ONE work-group per CU -- and varies what the
// product cannot vary cheaply: streaming waves per CU (S), tiles in flight per wave (D), and whether the tile goes to registers
// or straight to LDS (LDS-DMA, global_load_lds_dwordx4: no VGPRs held by a tile in flight).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kTileRows = 1024;

struct Args {
    const int32_t *a;   // int32 column
    const int8_t *b;    // int8 column
    uint64_t *bitmap;   // one 128-byte line per tile
    uint32_t *sink;
    int64_t n_tiles;
    int P;              // tiles per wave per span
    int stores;         // 0: none, 1: a 128-byte line per tile as it is done, 2: `park` lines at a time (parked in LDS, whatever the range boundaries)
    int park;           // lines per burst (<= 64)
    int work;           // extra vector instructions per tile (x 16)
};

struct Set {
    int32_t a[16];
    v4i b;
};

// asm loads: the compiler does not count them (its own counting collapses to vmcnt(0) as soon as a conditional store sits in the
// loop: one tile in flight whatever D says); the caller waits with an exact vmcnt(N) and ties the registers with touch()
template <int J = 0>
__device__ __forceinline__ void load_a(Set &s, const int32_t *p) {
    if constexpr (J < 16) {
        asm volatile("global_load_dword %0, %1, off offset:%2 nt" : "=v"(s.a[J]) : "v"(p), "n"(256 * J) : "memory");
        load_a<J + 1>(s, p);
    }
}
__device__ __forceinline__ void load_set(Set &s, const Args &g, int64_t t, int lane) {
    load_a<0>(s, g.a + t * kTileRows + lane);
    const v4i *pb = (const v4i *)(g.b + t * kTileRows) + lane;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(s.b) : "v"(pb) : "memory");
}
__device__ __forceinline__ void touch(Set &s) {
    asm volatile("" : "+v"(s.a[0]), "+v"(s.a[1]), "+v"(s.a[2]), "+v"(s.a[3]), "+v"(s.a[4]), "+v"(s.a[5]), "+v"(s.a[6]), "+v"(s.a[7]), "+v"(s.a[8]), "+v"(s.a[9]), "+v"(s.a[10]),
                 "+v"(s.a[11]), "+v"(s.a[12]), "+v"(s.a[13]), "+v"(s.a[14]), "+v"(s.a[15]), "+v"(s.b));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

__device__ __forceinline__ uint32_t consume(const Set &s, int work) {
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) x ^= (uint32_t)s.a[j];
    x ^= (uint32_t)s.b.x ^ (uint32_t)s.b.y ^ (uint32_t)s.b.z ^ (uint32_t)s.b.w;
    for (int k = 0; k < work; ++k) { // (dependent chain: the compiler cannot fold it)
#pragma unroll
        for (int j = 0; j < 16; ++j) x = (x << 1 | x >> 31) ^ (uint32_t)s.a[j];
    }
    return x;
}

// the wave's tile sequence: range by range (P consecutive tiles), a range of the next span every G * S * P tiles
struct Walk {
    int64_t t, jump;
    int j, P;
    __device__ __forceinline__ void init(int64_t first, int64_t jump_, int P_) { t = first; jump = jump_; j = 0; P = P_; }
    __device__ __forceinline__ int64_t next() {
        const int64_t r = t;
        ++t;
        if (++j == P) { j = 0; t += jump; }
        return r;
    }
};

// MODE 0: tiles in registers, D register sets.  One work-group = S waves.  ST (compile time): 0 no stores, 1 one 128-byte line
// per tile as it is done, 2 sixteen lines at a time (parked in LDS).  The wait before a tile is consumed is exact: everything
// issued after its loads -- the other D - 1 tiles' loads and, with ST = 1, the lines stored since -- stays in flight.
template <int S, int D, int ST>
__global__ __launch_bounds__(64 * S) void k_regs(const Args g) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_park[]; // [S][64][16] lines + [S][64] tile numbers
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint64_t *park = s_park + wave * 1024;
    uint32_t *ptile = (uint32_t *)(s_park + S * 1024) + wave * 64;
    Walk head, cur;
    const int64_t first = ((int64_t)blockIdx.x * S + wave) * g.P, jump = ((int64_t)gridDim.x * S - 1) * g.P;
    head.init(first, jump, g.P);
    cur.init(first, jump, g.P);
    Set s[D];
    int64_t last = 0;
    auto fetch = [&](Set &x) {
        int64_t t = head.next();
        if (t < g.n_tiles) last = t; else t = last;
        load_set(x, g, t, lane);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
        fetch(s[d]);
        // (the counted waits below assume one line stored behind every tile's loads -- from the first tile on: a count that is too
        // large lets a tile be consumed, and its registers be reused as ADDRESSES, before its loads have landed)
        if constexpr (ST == 1) __builtin_nontemporal_store(0ULL, g.bitmap + (g.n_tiles + d) * 16 + (lane & 15));
    }
    uint32_t acc = 0;
    int parked = 0;
    constexpr int kPerTile = 17 + (ST == 1 ? 1 : 0);
    for (;;) {
        bool done = false;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t t = cur.next();
            if (t >= g.n_tiles) { done = true; break; }
            wait_vm<(D - 1) * kPerTile + (ST == 1 ? 1 : 0)>();
            touch(s[d]);
            const uint32_t x = consume(s[d], g.work);
            acc ^= x;
            fetch(s[d]);
            if constexpr (ST == 1) { // (lanes 16.. repeat lanes 0..15: no branch around the store)
                __builtin_nontemporal_store((uint64_t)x * 0x9E3779B97F4A7C15ULL, g.bitmap + t * 16 + (lane & 15));
            } else if constexpr (ST == 2) { // (ranges are contiguous: the parked lines of one range go out together)
                if (lane < 16) park[parked * 16 + lane] = (uint64_t)x * 0x9E3779B97F4A7C15ULL;
                if (lane == 0) ptile[parked] = (uint32_t)t;
                ++parked;
                if (parked == g.park) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    for (int q = lane >> 4; q < parked; q += 4) __builtin_nontemporal_store(park[q * 16 + (lane & 15)], g.bitmap + (int64_t)ptile[q] * 16 + (lane & 15));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    parked = 0;
                }
            }
        }
        if (done) break;
    }
    wait_vm<0>();
#pragma unroll
    for (int d = 0; d < D; ++d) touch(s[d]);
    if constexpr (ST == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int q = lane >> 4; q < parked; q += 4) __builtin_nontemporal_store(park[q * 16 + (lane & 15)], g.bitmap + (int64_t)ptile[q] * 16 + (lane & 15));
    }
    if (acc == 0x12345u) *g.sink = acc;
}

// MODE 1: tiles straight to LDS (LDS-DMA), D buffers of 5 KiB per wave; the wave reads them back row-strided (ds_read_b32 /
// ds_read_u8: what the compares would need).  Counted waits: vmcnt((D - 1) * 5) leaves D - 1 tiles in flight.
__device__ __forceinline__ void glds_tile(const int32_t *pa, const int8_t *pb, uint32_t lds_at) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
                 "s_add_u32 m0, m0, 4096\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(pa), "v"(pb), "s"(lds_at)
                 : "memory");
}
template <int S, int D>
__global__ __launch_bounds__(64 * S) void k_dma(const Args g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_buf[]; // [S][D][5120]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t *mine = s_buf + (size_t)wave * D * 5120;
    const uint32_t mine_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)mine;
    Walk head, cur;
    const int64_t first = ((int64_t)blockIdx.x * S + wave) * g.P, jump = ((int64_t)gridDim.x * S - 1) * g.P;
    head.init(first, jump, g.P);
    cur.init(first, jump, g.P);
    int64_t last = 0;
    auto fetch = [&](int d) {
        int64_t t = head.next();
        if (t < g.n_tiles) last = t; else t = last;
        glds_tile(g.a + t * kTileRows + 4 * lane, g.b + t * kTileRows + 16 * lane, (uint32_t)__builtin_amdgcn_readfirstlane((int)(mine_at + d * 5120)));
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d);
    uint32_t acc = 0;
    for (;;) {
        bool done = false;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t t = cur.next();
            if (t >= g.n_tiles) { done = true; break; }
            wait_vm<(D - 1) * 5>();
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t *wa = (const uint32_t *)(mine + d * 5120);
            const uint8_t *wb = mine + d * 5120 + 4096;
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) x ^= wa[64 * j + lane];
#pragma unroll
            for (int j = 0; j < 16; ++j) x ^= (uint32_t)wb[64 * j + lane] << (j & 7);
            for (int k = 0; k < g.work; ++k) {
#pragma unroll
                for (int j = 0; j < 16; ++j) x = (x << 1 | x >> 31) ^ (uint32_t)j;
            }
            acc ^= x;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the buffer has been read: it may be filled again
            __builtin_amdgcn_sched_barrier(0);
            fetch(d);
            if (g.stores == 1 && lane < 16) __builtin_nontemporal_store((uint64_t)x * 0x9E3779B97F4A7C15ULL, g.bitmap + t * 16 + lane);
        }
        if (done) break;
    }
    wait_vm<0>();
    if (acc == 0x12345u) *g.sink = acc;
}

struct Case {
    const char *name;
    void (*launch)(const Args &, int grid, size_t lds);
    int S, D, mode;
};

template <int S, int D> void launch_regs(const Args &g, int grid, size_t lds) {
    (void)hipFuncSetAttribute((const void *)k_regs<S, D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_regs<S, D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_regs<S, D, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (g.stores == 0) hipLaunchKernelGGL((k_regs<S, D, 0>), dim3(grid), dim3(64 * S), lds, 0, g);
    else if (g.stores == 1) hipLaunchKernelGGL((k_regs<S, D, 1>), dim3(grid), dim3(64 * S), lds, 0, g);
    else hipLaunchKernelGGL((k_regs<S, D, 2>), dim3(grid), dim3(64 * S), lds, 0, g);
}
template <int S, int D> void launch_dma(const Args &g, int grid, size_t lds) { hipLaunchKernelGGL((k_dma<S, D>), dim3(grid), dim3(64 * S), lds, 0, g); }

int main(int argc, char **argv) {
    const int64_t n_rows = 100000000;
    const int64_t n_tiles = n_rows / kTileRows; // (full tiles only)
    const int n_sets = 2;
    int32_t *a[n_sets];
    int8_t *b[n_sets];
    uint64_t *bitmap;
    uint32_t *sink;
    for (int i = 0; i < n_sets; ++i) {
        CHECK(hipMalloc(&a[i], (size_t)n_rows * 4 + 65536));
        CHECK(hipMalloc(&b[i], (size_t)n_rows + 65536));
        CHECK(hipMemset(a[i], 1 + i, (size_t)n_rows * 4 + 65536));
        CHECK(hipMemset(b[i], 3 + i, (size_t)n_rows + 65536));
    }
    CHECK(hipMalloc(&bitmap, (size_t)(n_tiles + 16) * 128));
    CHECK(hipMalloc(&sink, 4));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int big = 64 * 1024; // (the product's rings: a work-group this large in LDS never shares its CU)
#define ALLOW(kern) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    ALLOW((k_dma<4, 4>)); ALLOW((k_dma<4, 6>)); ALLOW((k_dma<8, 2>)); ALLOW((k_dma<8, 3>)); ALLOW((k_dma<12, 2>)); ALLOW((k_dma<16, 2>)); ALLOW((k_dma<4, 3>)); ALLOW((k_dma<8, 4>));
    struct Run { const char *name; void (*launch)(const Args &, int, size_t); int grid_per_cu; size_t lds; int S, D; };
    const Run runs[] = {
        {"regs S=4  D=1 2wg/cu (k_filter_tile's shape)", launch_regs<4, 1>, 2, 4 * 8448, 4, 1},
        {"regs S=4  D=2 2wg/cu", launch_regs<4, 2>, 2, 4 * 8448, 4, 2},
        {"regs S=4  D=1 4wg/cu", launch_regs<4, 1>, 4, 4 * 8448, 4, 1},
        {"regs S=4  D=2 4wg/cu", launch_regs<4, 2>, 4, 4 * 8448, 4, 2},
        {"regs S=8  D=1 1wg/cu", launch_regs<8, 1>, 1, (size_t)90000, 8, 1},
        {"regs S=8  D=2 1wg/cu", launch_regs<8, 2>, 1, (size_t)90000, 8, 2},
        {"regs S=8  D=3 1wg/cu", launch_regs<8, 3>, 1, (size_t)90000, 8, 3},
        {"regs S=12 D=1 1wg/cu", launch_regs<12, 1>, 1, (size_t)101376, 12, 1},
        {"regs S=12 D=2 1wg/cu", launch_regs<12, 2>, 1, (size_t)101376, 12, 2},
        {"regs S=16 D=1 1wg/cu", launch_regs<16, 1>, 1, (size_t)135168, 16, 1},
        {"regs S=16 D=2 1wg/cu", launch_regs<16, 2>, 1, (size_t)135168, 16, 2},
        {"dma  S=4  D=3 1wg/cu", launch_dma<4, 3>, 1, (size_t)big + 4 * 3 * 5120, 4, 3},
        {"dma  S=4  D=4 1wg/cu", launch_dma<4, 4>, 1, (size_t)big + 4 * 4 * 5120, 4, 4},
        {"dma  S=4  D=6 1wg/cu", launch_dma<4, 6>, 1, (size_t)4 * 6 * 5120 + 32768, 4, 6},
        {"dma  S=8  D=2 1wg/cu", launch_dma<8, 2>, 1, (size_t)8 * 2 * 5120, 8, 2},
        {"dma  S=8  D=3 1wg/cu", launch_dma<8, 3>, 1, (size_t)8 * 3 * 5120, 8, 3},
        {"dma  S=8  D=4 1wg/cu", launch_dma<8, 4>, 1, (size_t)8 * 4 * 5120, 8, 4},
        {"dma  S=12 D=2 1wg/cu", launch_dma<12, 2>, 1, (size_t)12 * 2 * 5120, 12, 2},
        {"dma  S=16 D=2 1wg/cu", launch_dma<16, 2>, 1, (size_t)16 * 2 * 5120, 16, 2},
    };
    const int only = argc > 1 ? std::atoi(argv[1]) : -1;
    std::printf("%d CUs; 100 M rows of int32 + int8 = 500 MB per pass; P = 6; us = best of 12 after 4 warm-up passes (two buffer sets rotated)\n", cus);
    std::printf("%-48s %10s %10s %10s %10s %10s %10s\n", "variant", "no stores", "line/tile", "parked 6", "parked 16", "parked 48", "+256 valu");
    std::fflush(stdout);
    int idx = 0;
    for (const Run &r : runs) {
        if (only >= 0 && only != idx++) continue;
        float out[6] = {0, 0, 0, 0, 0, 0};
        for (int m = 0; m < 6; ++m) {
            Args g;
            g.bitmap = bitmap;
            g.sink = sink;
            g.n_tiles = n_tiles;
            g.P = 6;
            g.stores = m < 2 ? m : (m < 5 ? 2 : 0);
            g.park = m == 2 ? 6 : (m == 3 ? 16 : 48);
            g.work = m == 5 ? 16 : 0;
            if (m >= 2 && m <= 4 && r.name[0] == 'd') { out[m] = -1.f; continue; } // (the DMA variants have no parked form)
            float best = 1e9f;
            std::fprintf(stderr, "[%s mode %d]\n", r.name, m);
            for (int it = 0; it < 16; ++it) {
                g.a = a[it % n_sets];
                g.b = b[it % n_sets];
                CHECK(hipEventRecord(e0));
                r.launch(g, cus * r.grid_per_cu, r.lds);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipGetLastError());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 4 && ms < best) best = ms;
            }
            out[m] = best * 1e3f;
        }
        std::printf("%-48s %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f   (%.2f TB/s without stores)\n", r.name, out[0], out[1], out[2], out[3], out[4], out[5], 500e6 / out[0] / 1e6);
        std::fflush(stdout);
    }
    return 0;
}
