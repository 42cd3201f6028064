// imm3_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4), wave64.
//
// The reference's hot path is, per segment:
//   ScanOp.next      decode every used column block into a typed vector, selection = all rows
//                    (engine/.../operator/Scan.scala:28-70; DENSE_* decode is a fixed-width
//                    little-endian reinterpretation, core/.../codec/DenseCodec.scala:37-73)
//   SelectOp*.next   clear the bit of every row that fails (engine/.../operator/Select.scala:25-165)
//   ProjectOp.next   walk the set bits in ascending order and emit the SELECT-list values
//                    (engine/.../operator/Project.scala:37-64)
// On the GPU that becomes three launches over HBM-resident flat columns:
//   k_filter_*   one fused pass over all predicate columns -> selection bitmap (uint64 words, bit i of a
//                batch <-> word i>>6, bit i&63 == scala.collection.mutable.BitSet == wave64 ballot order),
//                per-tile survivor counts and the segment's selected-row count
//   k_scan       exclusive prefix of the per-tile counts (chunked)
//   k_gather     per tile: expand set bits into a dense LDS list in ascending row order, then write
//                row indices and gather the projected columns with dense, coalesced stores
// All of it is HBM-bound integer/byte work: no MFMA anywhere.
#include "imm3_internal.h"

namespace imm3 {

// ---------------------------------------------------------------------------------------------
// predicate evaluation
// ---------------------------------------------------------------------------------------------

// x in [lo, hi] (lo <= hi guaranteed by the host) with one subtract and one unsigned compare.
__device__ __forceinline__ bool in_closed(int32_t x, int32_t lo, int32_t hi) {
    return ((uint32_t)x - (uint32_t)lo) <= ((uint32_t)hi - (uint32_t)lo);
}

// SelectIteratorMatch (Select.scala:25-51): keep the row iff its `width` raw bytes equal one IN-list value.
__device__ __forceinline__ bool match_row(const ColPred &c, int64_t row) {
    const uint8_t *p = (const uint8_t *)c.data + row * (int64_t)c.width;
    bool found = false;
    if (c.match_in_args) {
        uint64_t v = 0;
        switch (c.width) {
        case 1: v = *p; break;
        case 2: v = *(const uint16_t *)p; break;
        case 4: v = *(const uint32_t *)p; break;
        case 8: v = *(const uint64_t *)p; break;
        default:
            for (int b = 0; b < c.width; ++b) v |= (uint64_t)p[b] << (8 * b);
        }
        for (int m = 0; m < c.n_match; ++m) found |= (v == c.match[m]);
    } else {
        for (int m = 0; m < c.n_match; ++m) {
            const uint8_t *q = c.match_blob + (int64_t)m * c.width;
            bool eq = true;
            for (int b = 0; b < c.width; ++b) eq &= (p[b] == q[b]);
            found |= eq;
        }
    }
    return found;
}

__device__ __forceinline__ bool eval_row(const ColPred &c, int64_t row) {
    switch (c.kind) {
    case KIND_I32: return in_closed(((const int32_t *)c.data)[row], c.lo, c.hi);
    case KIND_I8: return in_closed((int32_t)((const int8_t *)c.data)[row], c.lo, c.hi);
    default: return match_row(c, row);
    }
}

__device__ __forceinline__ uint64_t ballot64(bool p) { return (uint64_t)__ballot(p); }

// clang has no __builtin_amdgcn_writelane; bind the LLVM intrinsic directly (emits v_writelane_b32).
extern "C" __device__ int imm3_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// mask of the first `rem` bits (rem may be <= 0 or >= 64)
__device__ __forceinline__ uint64_t low_mask(int64_t rem) {
    return rem >= 64 ? ~0ULL : (rem <= 0 ? 0ULL : ((1ULL << rem) - 1ULL));
}

// Move 16 wave-uniform words into lanes 0..15 (lane j receives word j) with v_writelane.
__device__ __forceinline__ uint64_t words_to_lanes(const uint64_t (&acc)[kTileWords]) {
    int lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < kTileWords; ++j) {
        lo = imm3_writelane_i32((int)(uint32_t)acc[j], j, lo);
        hi = imm3_writelane_i32((int)(uint32_t)(acc[j] >> 32), j, hi);
    }
    return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}

// ---------------------------------------------------------------------------------------------
// k_filter_num: the hot kernel.  Numeric (DENSE_INT / DENSE_TINYINT) predicate columns only, uniform
// layout (every non-final block has rows % 64 == 0, so the batch-major bitmap is flat: word w <->
// rows [64w, 64w+64)).  One wave per 1024-row tile, grid-stride over tiles.
// Row-strided loads (lane l reads row 64j + l): the v_cmp result of load j IS bitmap word j, no
// cross-lane transpose.  16 independent loads per column are in flight per wave.
// ---------------------------------------------------------------------------------------------
// Per-workgroup survivor count -> block_partials[blockIdx.x]; k_total sums them.  (One same-address
// atomicAdd per wave costs ~12 ns serialised: 4096 of them were 40 % of the kernel.)
__device__ __forceinline__ void block_partial_store(uint32_t *block_partials, uint32_t wave_total, int lane, int wave) {
    __shared__ uint32_t s_part[kWavesPerBlock];
    if (lane == 0) s_part[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < kWavesPerBlock; ++i) t += s_part[i];
        block_partials[blockIdx.x] = t;
    }
}

__device__ __forceinline__ bool eval_num(const ColPred &c, int64_t row) {
    const int32_t x = c.kind == KIND_I32 ? ((const int32_t *)c.data)[row] : (int32_t)((const int8_t *)c.data)[row];
    return in_closed(x, c.lo, c.hi);
}

__global__ __launch_bounds__(kBlockThreads) void k_filter_num(const FilterArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned long long wave_total = 0;

    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < a.n_tiles;
         tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t row0 = tile * kTileRows;
        const bool full = row0 + kTileRows <= a.n_rows; // wave-uniform
        const int64_t w = tile * kTileWords + lane;      // lane j < 16 owns bitmap word j of the tile
        uint64_t mine = ~0ULL;
        if (a.and_existing) mine = (lane < kTileWords && w < a.n_words) ? a.bitmap[w] : 0ULL;

        if (full) {
            uint64_t acc[kTileWords]; // wave-uniform words (SGPR pairs)
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) acc[j] = ~0ULL;
            for (int ci = 0; ci < a.ncols; ++ci) {
                const ColPred &c = a.cols[ci];
                int32_t v[kTileWords];
                if (c.kind == KIND_I32) {
                    const int32_t *p = (const int32_t *)c.data + row0 + lane;
#pragma unroll
                    for (int j = 0; j < kTileWords; ++j) v[j] = p[64 * j];
                } else {
                    const int8_t *p = (const int8_t *)c.data + row0 + lane;
#pragma unroll
                    for (int j = 0; j < kTileWords; ++j) v[j] = (int32_t)p[64 * j];
                }
#pragma unroll
                for (int j = 0; j < kTileWords; ++j) acc[j] &= ballot64(in_closed(v[j], c.lo, c.hi));
            }
            mine &= words_to_lanes(acc);
        } else { // the one partial tile at the end of the segment: rolled, bounds-checked
            for (int ci = 0; ci < a.ncols; ++ci) {
                const ColPred &c = a.cols[ci];
#pragma unroll 1
                for (int j = 0; j < kTileWords; ++j) {
                    const int64_t row = row0 + 64 * j + lane;
                    const bool valid = row < a.n_rows;
                    const uint64_t m = ballot64(valid && eval_num(c, valid ? row : 0));
                    if (lane == j) mine &= m;
                }
            }
            mine &= low_mask(a.n_rows - (row0 + 64 * (int64_t)lane)); // rows past the end are not rows
        }
        if (lane >= kTileWords) mine = 0;

        uint32_t cnt = (uint32_t)__popcll(mine);
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d); // lanes 0..15 hold the tile's count
        if (lane < kTileWords && w < a.n_words) a.bitmap[w] = mine;   // 16 lanes x 8 B = one 128-B line
        if (lane == 0) {
            a.tile_counts[tile] = cnt;
            wave_total += cnt;
        }
    }
    block_partial_store(a.block_partials, (uint32_t)wave_total, lane, wave);
}

// ---------------------------------------------------------------------------------------------
// k_filter_generic: any column kind, any layout; one wave per bitmap word per iteration.
//   uniform layout (word_row_base == null): word w covers rows [64w, min(64w+64, n_rows))
//   ragged layout (arbitrary block sizes, e.g. the loader's trailing 1-row block): each
//   batch's BitSet starts on a fresh word, word w covers rows [base[w], base[w] + nvalid[w]).
// tile_counts must be zeroed before the launch.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void k_filter_generic(const FilterArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned long long wave_total = 0;
    for (int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave; w < a.n_words;
         w += (int64_t)gridDim.x * kWavesPerBlock) {
        int64_t base;
        int nv;
        if (a.word_row_base) {
            base = a.word_row_base[w];
            nv = a.word_nvalid[w];
        } else {
            base = 64 * w;
            const int64_t rem = a.n_rows - base;
            nv = rem >= 64 ? 64 : (int)rem;
        }
        const bool valid = lane < nv;
        uint64_t acc = a.and_existing ? a.bitmap[w] : ~0ULL;
        for (int ci = 0; ci < a.ncols; ++ci)
            acc &= ballot64(valid && eval_row(a.cols[ci], valid ? base + lane : base));
        acc &= low_mask(nv);
        const uint32_t cnt = (uint32_t)__popcll(acc);
        if (lane == 0) {
            a.bitmap[w] = acc;
            if (cnt) atomicAdd(&a.tile_counts[w / kTileWords], cnt);
        }
        wave_total += cnt;
    }
    block_partial_store(a.block_partials, (uint32_t)wave_total, lane, wave);
}

// ---------------------------------------------------------------------------------------------
// k_scan: exclusive prefix of tile_counts within chunks of 1024 tiles + per-chunk sums.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kChunkTiles) void k_scan(const ScanArgs a) {
    __shared__ uint32_t s_wave[kChunkTiles / 64];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int64_t tile = (int64_t)blockIdx.x * kChunkTiles + t;
    const uint32_t c = tile < a.n_tiles ? a.tile_counts[tile] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_prefix = 0;
    for (int i = 0; i < wave; ++i) wave_prefix += s_wave[i];
    incl += wave_prefix;
    if (tile < a.n_tiles) a.tile_offsets[tile] = incl - c;
    if (t == kChunkTiles - 1) a.chunk_sums[blockIdx.x] = incl;
}

// ---------------------------------------------------------------------------------------------
// k_total: one workgroup sums the filter launch's per-workgroup partials -> selected-row count of the
// segment, and the number of rows ProjectOp will emit.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_total(const TotalArgs a) {
    __shared__ unsigned long long s_wave[16];
    const int t = threadIdx.x;
    unsigned long long v = 0;
    for (int i = t; i < a.n_partials; i += 1024) v += a.block_partials[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((t & 63) == 0) s_wave[t >> 6] = v;
    __syncthreads();
    if (t == 0) {
        unsigned long long total = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) total += s_wave[i];
        *a.total = total;
        *a.n_emit = (a.limit > 0 && total > (unsigned long long)a.limit) ? (unsigned long long)a.limit : total;
    }
}

// ---------------------------------------------------------------------------------------------
// k_gather: ProjectOp.  One wave per tile.  Phase 1 expands the tile's set bits into an ascending list
// of in-tile positions in LDS (rank of a row = popcount of lower bits: v_mbcnt).  Phase 2 walks that
// list densely: lane i handles survivor i, so row-index / value stores are contiguous.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void copy_elem(const void *src, void *dst, int64_t row, uint64_t out) {
    ((T *)dst)[out] = ((const T *)src)[row];
}

__global__ __launch_bounds__(kBlockThreads) void k_gather(const GatherArgs a) {
    __shared__ uint16_t s_list[kWavesPerBlock][kTileRows];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint16_t *list = s_list[wave];
    const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t iters = (a.n_tiles + stride - 1) / stride; // same trip count for every wave of the grid

    for (int64_t it = 0; it < iters; ++it) {
        const int64_t tile = it * stride + (int64_t)blockIdx.x * kWavesPerBlock + wave;
        uint32_t cnt = 0;
        uint64_t base = 0;
        if (tile < a.n_tiles) {
            cnt = a.tile_counts[tile];
            if (cnt) {
                const int64_t chunk = tile / kChunkTiles;
                uint64_t part = 0;
                for (int64_t i = lane; i < chunk; i += 64) part += a.chunk_sums[i];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
                base = part + a.tile_offsets[tile];
                if (a.limit > 0) {
                    if (base >= (uint64_t)a.limit) cnt = 0;
                    else if (base + cnt > (uint64_t)a.limit) cnt = (uint32_t)((uint64_t)a.limit - base);
                }
            }
        }
        if (cnt) { // wave-uniform
            uint64_t word = 0;
            const int64_t w = tile * kTileWords + lane;
            if (lane < kTileWords && w < a.n_words) word = a.bitmap[w];
            uint32_t incl = (uint32_t)__popcll(word);
            const uint32_t pc = incl;
#pragma unroll
            for (int d = 1; d < kTileWords; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const uint32_t excl = incl - pc;
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)word, j);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(word >> 32), j);
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)excl, j);
                const uint64_t m = ((uint64_t)hi << 32) | lo;
                if ((m >> lane) & 1ULL) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                    list[off + rank] = (uint16_t)(j * 64 + lane);
                }
            }
        }
        __syncthreads();
        if (cnt) {
            for (uint32_t i = lane; i < cnt; i += 64) {
                const uint32_t r = list[i];
                const uint64_t out = base + i;
                if (out >= a.cap_rows) continue;
                const int64_t row = a.word_row_base
                                        ? (int64_t)a.word_row_base[tile * kTileWords + (r >> 6)] + (r & 63)
                                        : tile * kTileRows + r;
                if (a.row_index) a.row_index[out] = (uint32_t)row;
                for (int pj = 0; pj < a.n_proj; ++pj) {
                    const ProjCol &pc2 = a.proj[pj];
                    switch (pc2.width) {
                    case 4: copy_elem<uint32_t>(pc2.src, pc2.dst, row, out); break;
                    case 1: copy_elem<uint8_t>(pc2.src, pc2.dst, row, out); break;
                    case 2: copy_elem<uint16_t>(pc2.src, pc2.dst, row, out); break;
                    case 8: copy_elem<uint64_t>(pc2.src, pc2.dst, row, out); break;
                    default: {
                        const uint8_t *s = (const uint8_t *)pc2.src + row * (int64_t)pc2.width;
                        uint8_t *d = (uint8_t *)pc2.dst + out * (uint64_t)pc2.width;
                        for (int b = 0; b < pc2.width; ++b) d[b] = s[b];
                    }
                    }
                }
            }
        }
        __syncthreads(); // list is reused by the next iteration
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int clamp_grid(int64_t want, int cap) {
    if (want < 1) want = 1;
    return (int)(want > cap ? cap : want);
}

int filter_grid(const FilterArgs &a, bool generic, int grid_blocks) {
    // 256 CUs x 8 resident 256-thread workgroups; more only lengthens the partials reduction
    const int cap = grid_blocks > 0 ? (grid_blocks > kMaxFilterGrid ? kMaxFilterGrid : grid_blocks) : 2048;
    const int64_t units = generic ? a.n_words : a.n_tiles;
    return clamp_grid((units + kWavesPerBlock - 1) / kWavesPerBlock, cap);
}

void launch_filter(const FilterArgs &a, bool generic, int variant, int grid, hipStream_t s) {
    (void)variant;
    if (generic) hipLaunchKernelGGL(k_filter_generic, dim3(grid), dim3(kBlockThreads), 0, s, a);
    else hipLaunchKernelGGL(k_filter_num, dim3(grid), dim3(kBlockThreads), 0, s, a);
}

void launch_total(const TotalArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_total, dim3(1), dim3(1024), 0, s, a);
}

void launch_scan(const ScanArgs &a, hipStream_t s) {
    const int grid = (int)((a.n_tiles + kChunkTiles - 1) / kChunkTiles);
    hipLaunchKernelGGL(k_scan, dim3(grid < 1 ? 1 : grid), dim3(kChunkTiles), 0, s, a);
}

void launch_gather(const GatherArgs &a, int grid_blocks, hipStream_t s) {
    const int cap = grid_blocks > 0 ? grid_blocks : 4096;
    const int grid = clamp_grid((a.n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, cap);
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(kBlockThreads), 0, s, a);
}

} // namespace imm3
