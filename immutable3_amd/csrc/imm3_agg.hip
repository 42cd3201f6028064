// imm3_agg.hip -- group-by aggregation (count / min / max) over the selected rows of one segment: the GPU
// counterpart of ProjectAggOp.ProjectAggIterator.runAggs
// (engine/src/main/scala/immutabledb/engine/operator/ProjectAggregate.scala:115-227).
//
//   reference: for every selected row (ascending) build groupKey = group values mkString "_", look the key up
//              in a LinkedHashMap and update one Aggregator per alias (CountAggr.add, Max/MinDoubleAggr.add,
//              MaxStringAggr.add, :11-112).
//   here     : the group key is the concatenation of the group columns' raw bytes (<= 8 bytes, u64).  Every
//              work-group aggregates its spans into an LDS hash table (LDS atomics: no global contention on hot
//              groups), then flushes the table into a global open-addressing table with one atomic set per
//              (work-group, group).  k_group_collect compacts the occupied entries; the host orders them by
//              first_row, which IS the LinkedHashMap's first-seen order.
// Numeric min/max stay int32 (exact); the host converts to Double like value.toDouble.  String max compares the
// value bytes big-endian-packed into a u64 == lexicographic byte order == String.compareTo for ASCII.
#include "imm3_internal.h"
#include <hip/hip_ext.h>

namespace imm3 {

constexpr unsigned long long kEmptyKey = ~0ULL;
constexpr int kLdsSlots = 1024;     // per-work-group hash table entries (+1 for the all-ones key)
constexpr int kMaxProbes = 48;

__device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (uint32_t)k;
}

__device__ __forceinline__ unsigned long long load_le(const void *base, int64_t row, int width) {
    const uint8_t *p = (const uint8_t *)base + row * (int64_t)width;
    switch (width) {
    case 1: return *p;
    case 2: return *(const uint16_t *)p;
    case 4: return *(const uint32_t *)p;
    case 8: return *(const unsigned long long *)p;
    default: {
        unsigned long long v = 0;
        for (int b = 0; b < width; ++b) v |= (unsigned long long)p[b] << (8 * b);
        return v;
    }
    }
}

// value of one aggregate column at `row`, as the i64 the tables hold
__device__ __forceinline__ long long agg_value(const AggCol &a, int64_t row) {
    if (a.kind == AGG_COUNT) return 0;
    if (a.is_str) { // big-endian pack: integer order == byte-lexicographic order
        const uint8_t *p = (const uint8_t *)a.data + row * (int64_t)a.width;
        unsigned long long v = 0;
        for (int b = 0; b < a.width; ++b) v = (v << 8) | p[b];
        return (long long)v;
    }
    if (a.width == 4) return (long long)((const int32_t *)a.data)[row];
    return (long long)((const int8_t *)a.data)[row];
}

__device__ __forceinline__ void agg_update_global(const AggArgs &a, uint32_t g, uint32_t first, unsigned long long count, const long long *vals) {
    atomicMin(&a.first[g], first);
    atomicAdd(&a.counts[g], count);
    for (int j = 0; j < a.n_agg; ++j) {
        long long *slot = &a.vals[(size_t)g * kMaxAggs + j];
        if (a.aggs[j].kind == AGG_MIN) atomicMin(slot, vals[j]);
        else if (a.aggs[j].kind == AGG_MAX) {
            if (a.aggs[j].is_str) atomicMax((unsigned long long *)slot, (unsigned long long)vals[j]);
            else atomicMax(slot, vals[j]);
        }
    }
}

// slot of `key` in the global table (inserting it if new); capacity = mask + 1, plus one extra slot for the
// all-ones key.  Returns 0xFFFFFFFF and raises the overflow flag when the table is full.
__device__ __forceinline__ uint32_t global_slot(const AggArgs &a, unsigned long long key) {
    if (key == kEmptyKey) return a.mask + 1;
    uint32_t g = hash_key(key) & a.mask;
    for (uint32_t probes = 0; probes <= a.mask; ++probes) {
        const unsigned long long prev = atomicCAS(&a.keys[g], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) return g;
        g = (g + 1) & a.mask;
    }
    *a.overflow = 1;
    return 0xFFFFFFFFu;
}

__global__ __launch_bounds__(kBlockThreads) void k_group_agg(const AggArgs a) {
    __shared__ uint16_t s_list[kSpanWords * 64];              // 32 KiB: in-span positions of the survivors
    __shared__ unsigned long long s_keys[kLdsSlots + 1];      // 8 KiB
    __shared__ uint32_t s_first[kLdsSlots + 1];               // 4 KiB
    __shared__ uint32_t s_count[kLdsSlots + 1];               // 4 KiB
    __shared__ long long s_vals[(kLdsSlots + 1) * kMaxAggs];  // 32 KiB
    __shared__ uint32_t s_wave[kWavesPerBlock];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;

    for (int i = t; i <= kLdsSlots; i += kBlockThreads) {
        s_keys[i] = kEmptyKey;
        s_first[i] = 0xFFFFFFFFu;
        s_count[i] = 0;
        for (int j = 0; j < kMaxAggs; ++j) {
            const int kind = j < a.n_agg ? a.aggs[j].kind : AGG_COUNT;
            const bool str = j < a.n_agg && a.aggs[j].is_str;
            s_vals[i * kMaxAggs + j] = kind == AGG_MIN ? INT64_MAX : (str ? 0 : INT64_MIN);
        }
    }
    __syncthreads();

    const int64_t n_spans = (a.n_tiles + kSpanTiles - 1) / kSpanTiles;
    for (int64_t span = blockIdx.x; span < n_spans; span += gridDim.x) {
        // survivors of the span -> ascending list in LDS (same scheme as k_gather)
        const int64_t w = span * kSpanWords + t;
        uint64_t word = w < a.n_words ? a.bitmap[w] : 0ULL;
        const uint32_t pc = (uint32_t)__popcll(word);
        uint32_t incl = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t off = incl - pc, total = 0;
#pragma unroll
        for (int i = 0; i < kWavesPerBlock; ++i) {
            if (i < wave) off += s_wave[i];
            total += s_wave[i];
        }
        while (word) {
            const int b = __builtin_ctzll(word);
            s_list[off++] = (uint16_t)(t * 64 + b);
            word &= word - 1;
        }
        __syncthreads();
        for (uint32_t i = t; i < total; i += kBlockThreads) {
            const uint32_t r = s_list[i];
            const int64_t row = a.word_row_base ? (int64_t)a.word_row_base[span * kSpanWords + (r >> 6)] + (r & 63)
                                                : span * (int64_t)(kSpanWords * 64) + r;
            unsigned long long key = 0;
            for (int g = 0; g < a.n_group; ++g) key |= load_le(a.groups[g].data, row, a.groups[g].width) << (8 * a.groups[g].shift);
            long long vals[kMaxAggs];
            for (int j = 0; j < a.n_agg; ++j) vals[j] = agg_value(a.aggs[j], row);
            // LDS table
            int slot = -1;
            if (key == kEmptyKey) slot = kLdsSlots;
            else {
                uint32_t s = hash_key(key) & (kLdsSlots - 1);
                for (int probes = 0; probes < kMaxProbes; ++probes) {
                    const unsigned long long prev = atomicCAS(&s_keys[s], kEmptyKey, key);
                    if (prev == kEmptyKey || prev == key) { slot = (int)s; break; }
                    s = (s + 1) & (kLdsSlots - 1);
                }
            }
            if (slot >= 0) {
                atomicMin(&s_first[slot], (uint32_t)row);
                atomicAdd(&s_count[slot], 1u);
                for (int j = 0; j < a.n_agg; ++j) {
                    long long *v = &s_vals[slot * kMaxAggs + j];
                    if (a.aggs[j].kind == AGG_MIN) atomicMin(v, vals[j]);
                    else if (a.aggs[j].kind == AGG_MAX) {
                        if (a.aggs[j].is_str) atomicMax((unsigned long long *)v, (unsigned long long)vals[j]);
                        else atomicMax(v, vals[j]);
                    }
                }
            } else { // the work-group's table is crowded (many distinct keys): straight to the global table
                const uint32_t g = global_slot(a, key);
                if (g != 0xFFFFFFFFu) agg_update_global(a, g, (uint32_t)row, 1ULL, vals);
            }
        }
        __syncthreads(); // s_list / s_wave are reused by the next span
    }

    // flush: one atomic set per (work-group, group)
    for (int i = t; i <= kLdsSlots; i += kBlockThreads) {
        if (s_count[i] == 0) continue;
        const unsigned long long key = i == kLdsSlots ? kEmptyKey : s_keys[i];
        const uint32_t g = global_slot(a, key);
        if (g != 0xFFFFFFFFu) agg_update_global(a, g, s_first[i], (unsigned long long)s_count[i], &s_vals[i * kMaxAggs]);
    }
}

// occupied entries of the global table -> dense arrays (order irrelevant: the host sorts by first_row)
__global__ __launch_bounds__(kBlockThreads) void k_group_collect(const AggArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)a.mask + 2;
    for (int64_t base = ((int64_t)blockIdx.x * kBlockThreads + (threadIdx.x & ~63)); base < n; base += (int64_t)gridDim.x * kBlockThreads) {
        const int64_t i = base + lane;
        const bool occ = i < n && a.counts[i] != 0;
        const uint64_t m = (uint64_t)__ballot(occ);
        if (!m) continue;
        uint32_t start = 0;
        if (lane == 0) start = atomicAdd(a.n_groups, (uint32_t)__popcll(m));
        start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
        if (occ) {
            const uint32_t o = start + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
            if (o < a.out_cap) {
                a.out_keys[o] = i == (int64_t)a.mask + 1 ? kEmptyKey : a.keys[i];
                a.out_first[o] = a.first[i];
                a.out_counts[o] = a.counts[i];
                for (int j = 0; j < kMaxAggs; ++j) a.out_vals[(size_t)o * kMaxAggs + j] = a.vals[(size_t)i * kMaxAggs + j];
            }
        }
    }
}

// (re)initialise the global table
__global__ __launch_bounds__(kBlockThreads) void k_group_init(const AggArgs a) {
    const int64_t n = (int64_t)a.mask + 2;
    for (int64_t i = (int64_t)blockIdx.x * kBlockThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlockThreads) {
        a.keys[i] = kEmptyKey;
        a.first[i] = 0xFFFFFFFFu;
        a.counts[i] = 0;
        for (int j = 0; j < kMaxAggs; ++j) {
            const int kind = j < a.n_agg ? a.aggs[j].kind : AGG_COUNT;
            const bool str = j < a.n_agg && a.aggs[j].is_str;
            a.vals[(size_t)i * kMaxAggs + j] = kind == AGG_MIN ? INT64_MAX : (str ? 0 : INT64_MIN);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *a.n_groups = 0;
        *a.overflow = 0;
    }
}

void launch_group_agg(const AggArgs &a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const int64_t n = (int64_t)a.mask + 2;
    const int init_grid = (int)std::min<int64_t>((n + kBlockThreads - 1) / kBlockThreads, 2048);
    hipLaunchKernelGGL(k_group_init, dim3(init_grid), dim3(kBlockThreads), 0, s, a);
    const int64_t n_spans = (a.n_tiles + kSpanTiles - 1) / kSpanTiles;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_spans, 512)); // 80 KiB of LDS per work-group: 2 per CU
    hipExtLaunchKernelGGL(k_group_agg, dim3(grid), dim3(kBlockThreads), 0, s, ev0, ev1, 0, a);
}

void launch_group_collect(const AggArgs &a, hipStream_t s) {
    const int64_t n = (int64_t)a.mask + 2;
    const int grid = (int)std::min<int64_t>((n + kBlockThreads - 1) / kBlockThreads, 2048);
    hipLaunchKernelGGL(k_group_collect, dim3(grid), dim3(kBlockThreads), 0, s, a);
}

} // namespace imm3
