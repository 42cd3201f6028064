// imm3_agg.hip -- group-by aggregation (count / min / max) over the selected rows of one segment: the GPU
// counterpart of ProjectAggOp.ProjectAggIterator.runAggs
// (engine/src/main/scala/immutabledb/engine/operator/ProjectAggregate.scala:115-227).
//
//   reference: for every selected row (ascending) build groupKey = group values mkString "_", look the key up
//              in a LinkedHashMap and update one Aggregator per alias (CountAggr.add, Max/MinDoubleAggr.add,
//              MaxStringAggr.add, :11-112).
//   here     : the group key is the concatenation of the group columns' raw bytes (<= 8 bytes, u64).  Every
//              work-group aggregates its share of the selection bitmap into an LDS hash table (LDS atomics: no global contention on hot
//              groups), then flushes the table into a global open-addressing table with one atomic set per
//              (work-group, group).  k_group_collect compacts the occupied entries; the host orders them by
//              first_row, which IS the LinkedHashMap's first-seen order.
// Numeric min/max stay int32 (exact); the host converts to Double like value.toDouble.  String max compares the
// value bytes big-endian-packed into a u64 == lexicographic byte order == String.compareTo for ASCII.
#include "imm3_internal.h"
#include "imm3_device.h"
#include "imm3_tile.h"
#include <atomic>
#include <hip/hip_ext.h>

namespace imm3 {

constexpr unsigned long long kEmptyKey = ~0ULL;
constexpr int kLdsSlots = 1024;     // per-work-group hash table entries (+1 for the all-ones key)
constexpr int kMaxProbes = 48;

// 32-bit mixing only: v_mul_lo_u32 is quarter rate on CDNA and a 64 x 64-bit multiply is six of them -- the former
// two-multiply 64-bit finaliser cost ~200 cycles per row, a third of k_group_agg.
__device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
    uint32_t x = (uint32_t)k ^ ((uint32_t)(k >> 32) * 0x9E3779B1u);
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
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

// Where a selected row's group lives: slot >= 0 in the work-group's LDS table, or (slot < 0) index gslot of the
// global table when the LDS table is crowded (many distinct keys).
struct LdsTable {
    unsigned long long *keys;
    uint32_t *first;
    uint32_t *count;
    long long *vals;
};

struct GroupRef {
    int slot;
    uint32_t gslot;
};

// find / insert the key, and account the row itself (first-seen row, count)
__device__ __forceinline__ GroupRef agg_locate(const AggArgs &a, const LdsTable &t, uint32_t row, unsigned long long key) {
    GroupRef r{-1, 0xFFFFFFFFu};
    if (key == kEmptyKey) r.slot = kLdsSlots;
    else {
        uint32_t s = hash_key(key) & (kLdsSlots - 1);
        for (int probes = 0; probes < kMaxProbes; ++probes) {
            unsigned long long cur = t.keys[s]; // plain read first: after warm-up the key is there and no CAS is needed
            if (cur == kEmptyKey) cur = atomicCAS(&t.keys[s], kEmptyKey, key);
            if (cur == kEmptyKey || cur == key) { r.slot = (int)s; break; }
            s = (s + 1) & (kLdsSlots - 1);
        }
    }
    if (r.slot >= 0) {
        if (row < t.first[r.slot]) atomicMin(&t.first[r.slot], row);
        atomicAdd(&t.count[r.slot], 1u);
    } else {
        r.gslot = global_slot(a, key);
        if (r.gslot != 0xFFFFFFFFu) {
            atomicMin(&a.first[r.gslot], row);
            atomicAdd(&a.counts[r.gslot], 1ULL);
        }
    }
    return r;
}

// fold value v of aggregate j into the row's group
__device__ __forceinline__ void agg_fold(const AggArgs &a, const LdsTable &t, const GroupRef &r, int j, long long v) {
    const int kind = a.aggs[j].kind;
    if (kind == AGG_COUNT) return;
    const bool str = a.aggs[j].is_str;
    if (r.slot >= 0) {
        long long *p = &t.vals[r.slot * kMaxAggs + j];
        // read before the atomic: once a group's extreme is established almost every row is a no-op
        if (kind == AGG_MIN) { if (v < *p) atomicMin(p, v); }
        else if (str) { if ((unsigned long long)v > (unsigned long long)*p) atomicMax((unsigned long long *)p, (unsigned long long)v); }
        else if (v > *p) atomicMax(p, v);
    } else if (r.gslot != 0xFFFFFFFFu) {
        long long *p = &a.vals[(size_t)r.gslot * kMaxAggs + j];
        if (kind == AGG_MIN) atomicMin(p, v);
        else if (str) atomicMax((unsigned long long *)p, (unsigned long long)v);
        else atomicMax(p, v);
    }
}

__device__ __forceinline__ unsigned long long row_key(const AggArgs &a, int64_t row) {
    unsigned long long key = 0;
    for (int g = 0; g < a.n_group; ++g) key |= load_le(a.groups[g].data, row, a.groups[g].width) << (8 * a.groups[g].shift);
    return key;
}

// 8 waves share one LDS table: 3 work-groups per CU by LDS (48 KiB each) = 6 waves per SIMD.  (1024-thread groups
// were held to ONE per CU by their 72 VGPRs: 0.41 ms -> 0.355 ms for 100 M rows.)  Counters for the same run
// (rocprofv3 --pmc, per 256-row step): 251 vector, 238 scalar, 40 LDS instructions (LDS busy 0.15 ms, vector issue
// 0.16 ms, scalar issue 0.15 ms of the 0.34 ms) -- instruction-bound on all three ports; a wave-synchronous variant (idle lanes steered to a dummy slot, rare paths behind ballots)
// measured slower (0.47 ms) and was dropped.
constexpr int kAggThreads = 512;
constexpr int kAggWaves = kAggThreads / 64;

// raw little-endian values of 4 consecutive rows (row0 % 4 == 0) with ONE load of >= 4 bytes per lane.
// Sub-dword gathers are slow on this chip (measured: 0.49 ms just to load a 2-byte and a 1-byte column of 100 M rows
// with ushort / sbyte loads); dword / dwordx2 / dwordx4 loads of lane-contiguous rows stream at HBM rate.
__device__ __forceinline__ void load4_raw(const void *base, int64_t row0, int width, unsigned long long (&out)[4]) {
    const uint8_t *p = (const uint8_t *)base + row0 * (int64_t)width;
    switch (width) {
    case 1: {
        const uint32_t v = __builtin_nontemporal_load((const uint32_t *)p);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (v >> (8 * k)) & 0xFFu;
        break;
    }
    case 2: {
        const uint2 v = *(const uint2 *)p;
        out[0] = v.x & 0xFFFFu; out[1] = v.x >> 16; out[2] = v.y & 0xFFFFu; out[3] = v.y >> 16;
        break;
    }
    case 4: {
        const uint4 v = *(const uint4 *)p;
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        break;
    }
    case 8: {
        const uint4 v0 = *(const uint4 *)p, v1 = *((const uint4 *)p + 1);
        out[0] = ((unsigned long long)v0.y << 32) | v0.x; out[1] = ((unsigned long long)v0.w << 32) | v0.z;
        out[2] = ((unsigned long long)v1.y << 32) | v1.x; out[3] = ((unsigned long long)v1.w << 32) | v1.z;
        break;
    }
    default:
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = load_le(base, row0 + k, width);
    }
}

// raw little-endian value -> the i64 the tables hold (see agg_value)
__device__ __forceinline__ long long agg_from_raw(const AggCol &a, unsigned long long raw) {
    if (a.kind == AGG_COUNT) return 0;
    if (a.is_str) return (long long)(__builtin_bswap64(raw) >> (8 * (8 - a.width))); // big-endian pack
    if (a.width == 4) return (long long)(int32_t)(uint32_t)raw;
    return (long long)(int8_t)(uint8_t)raw;
}

// Uniform layout: one wave step = 4 consecutive bitmap words = 256 rows; lane l owns rows 4l .. 4l+3 of the step
// (bits = nibble l&15 of word l>>4) and loads them with one >= 4-byte load per column.  Steps without a survivor are
// skipped before any column load.  Ragged layout: lane l <-> row base(word) + l, one word per step.
__global__ __launch_bounds__(kAggThreads) void k_group_agg(const AggArgs a) {
    __shared__ unsigned long long s_keys[kLdsSlots + 1];      // 8 KiB
    __shared__ uint32_t s_first[kLdsSlots + 1];               // 4 KiB
    __shared__ uint32_t s_count[kLdsSlots + 1];               // 4 KiB
    __shared__ long long s_vals[(kLdsSlots + 1) * kMaxAggs];  // 32 KiB
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const LdsTable tab{s_keys, s_first, s_count, s_vals};

    for (int i = t; i <= kLdsSlots; i += kAggThreads) {
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

    const int64_t stride = (int64_t)gridDim.x * kAggWaves;
    if (!a.word_row_base) {
        const int64_t n_steps = (a.n_words + 3) / 4;
        for (int64_t st = (int64_t)blockIdx.x * kAggWaves + wave; st < n_steps; st += stride) {
            const int64_t w = st * 4 + (lane >> 4);
            const uint64_t word = w < a.n_words ? a.bitmap[w] : 0ULL; // 16 lanes share an address: 4 x 8 B per wave
            const uint32_t nib = (uint32_t)(word >> (4 * (lane & 15))) & 0xFu;
            if (!__ballot(nib != 0)) continue; // wave-uniform: nothing selected in these 256 rows
            const int64_t row0 = st * 256 + 4 * lane;                 // (virtual) row of the lane's first row
            const int64_t tile = st >> 2;                              // table queries address columns through the tile table
            const int64_t in_tile = (st & 3) * 256 + 4 * lane;
            // group key of the lane's 4 rows, one column at a time (keeps few registers live)
            unsigned long long key[4] = {0, 0, 0, 0};
            for (int g = 0; g < a.n_group; ++g) {
                unsigned long long raw[4];
                if (a.groups[g].tile_ptrs) load4_raw(as_global(a.groups[g].tile_ptrs[tile]), in_tile, a.groups[g].width, raw);
                else load4_raw(a.groups[g].data, row0, a.groups[g].width, raw);
#pragma unroll
                for (int k = 0; k < 4; ++k) key[k] |= raw[k] << (8 * a.groups[g].shift);
            }
            if (IMM3_ABLATED(a, 1)) { // ablation: key loads only
                asm volatile("" ::"v"((uint32_t)key[0]), "v"((uint32_t)key[3]));
                continue;
            }
            GroupRef ref[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ref[k] = GroupRef{-1, 0xFFFFFFFFu};
                if ((nib >> k) & 1u) ref[k] = agg_locate(a, tab, (uint32_t)(row0 + k), key[k]);
            }
            // then every aggregate column: one load for the 4 rows, fold, next column
            for (int q = 0; q < a.n_agg; ++q) {
                if (a.aggs[q].kind == AGG_COUNT) continue;
                unsigned long long raw[4];
                if (a.aggs[q].tile_ptrs) load4_raw(as_global(a.aggs[q].tile_ptrs[tile]), in_tile, a.aggs[q].width, raw);
                else load4_raw(a.aggs[q].data, row0, a.aggs[q].width, raw);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((nib >> k) & 1u) agg_fold(a, tab, ref[k], q, agg_from_raw(a.aggs[q], raw[k]));
            }
        }
    } else {
        for (int64_t w = (int64_t)blockIdx.x * kAggWaves + wave; w < a.n_words; w += stride) {
            const uint64_t word = a.bitmap[w];
            if (word == 0) continue; // wave-uniform
            if ((word >> lane) & 1ULL) {
                const int64_t row = (int64_t)a.word_row_base[w] + lane;
                const GroupRef ref = agg_locate(a, tab, (uint32_t)row, row_key(a, row));
                for (int q = 0; q < a.n_agg; ++q) agg_fold(a, tab, ref, q, agg_value(a.aggs[q], row));
            }
        }
    }
    __syncthreads();

    // flush: one atomic set per (work-group, group)
    for (int i = t; i <= kLdsSlots; i += kAggThreads) {
        if (s_count[i] == 0) continue;
        const unsigned long long key = i == kLdsSlots ? kEmptyKey : s_keys[i];
        const uint32_t g = global_slot(a, key);
        if (g != 0xFFFFFFFFu) agg_update_global(a, g, s_first[i], (unsigned long long)s_count[i], &s_vals[i * kMaxAggs]);
    }
}

// ---------------------------------------------------------------------------------------------
// k_group_agg_tile: the fast form -- one uniform segment, group key <= 4 bytes (one or two columns of 1 / 2 / 4
// bytes), aggregates count / min / max over int32 / int8 columns or max over strings of <= 4 bytes: keys, values and
// table entries are all 32-bit.  One wave per 1024-row tile, every column ROW-STRIDED like in the filter kernel
// (narrow columns transposed through LDS), so bitmap word j is the exec mask of register j and a selected row costs one
// probe read plus one LDS atomic per aggregate: ds_add (count), ds_min (first-seen row), ds_min / ds_max (values) --
// no guards, no 64-bit arithmetic, no per-row branches.  The general kernel above did ~250 vector + ~240 scalar
// instructions per 256 rows (hashing 64-bit keys, probing with CAS, guarded atomics on i64 values).
// A work-group whose 1024-slot table fills up raises overflow = 2 and the host re-runs the general kernel.
// ---------------------------------------------------------------------------------------------
constexpr int kFastSlots = 1024;
constexpr uint32_t kEmpty32 = 0xFFFFFFFFu;

// column `data` (width 1 / 2 / 4), tile `tile` -> v[j] = raw little-endian value of row 64j + lane, zero-extended
__device__ __forceinline__ void load_tile_rows(const void *data, int width, int64_t tile, int lane, uint8_t *xp, uint32_t (&v)[kTileWords]) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    if (width == 4) {
        const uint32_t *p = (const uint32_t *)data + tile * kTileRows + lane;
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
    } else if (width == 2) {
        const v4i_ *p = (const v4i_ *)((const uint16_t *)data + tile * kTileRows) + lane;
        const v4i_ r0 = __builtin_nontemporal_load(p), r1 = __builtin_nontemporal_load(p + 64);
        *(v4i_ *)(xp + 16 * lane) = r0;
        *(v4i_ *)(xp + 1024 + 16 * lane) = r1;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) v[j] = ((const uint16_t *)xp)[64 * j + lane];
        asm volatile("" ::: "memory");
    } else {
        const v4i_ r0 = __builtin_nontemporal_load((const v4i_ *)((const uint8_t *)data + tile * kTileRows) + lane);
        *(v4i_ *)(xp + 16 * lane) = r0;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) v[j] = ((const uint8_t *)xp)[64 * j + lane];
        asm volatile("" ::: "memory");
    }
}

// Table layout: one array per field, each striding by ONE element per slot.  Every update is an UNCONDITIONAL LDS atomic:
// guarding the first-seen row and the extremes with a read (update only when the row improves them -- the general kernel's
// trick) measured SLOWER here, 331-354 us against 263 per 100 M rows: each guard is one more dependent LDS round trip and
// one more divergent branch per word, and those, not the atomics (~30 us for the count and first-seen pair), are what
// this kernel's time is made of.  Ablation (100 M rows, group by state: count(id), max(age)): value phase 80 us, count +
// first-seen atomics 30 us, flush 11 us, key load + probe loop the rest -- 16 serialised probe round trips per tile.

__global__ __launch_bounds__(kBlockThreads) void k_group_agg_tile(const AggArgs a) {
    __shared__ uint32_t s_keys[kFastSlots + 1];
    __shared__ uint32_t s_first[kFastSlots + 1];              // first-seen (lowest) selected row
    __shared__ uint32_t s_count[kFastSlots + 1];
    __shared__ uint32_t s_vals[kMaxAggs][kFastSlots + 1];     // per aggregate
    __shared__ __attribute__((aligned(16))) uint8_t s_xp[kWavesPerBlock][2048];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    uint8_t *xp = s_xp[wave];
    for (int i = t; i <= kFastSlots; i += kBlockThreads) {
        s_keys[i] = kEmpty32;
        s_first[i] = 0xFFFFFFFFu;
        s_count[i] = 0;
        for (int j = 0; j < kMaxAggs; ++j) {
            const int kind = j < a.n_agg ? a.aggs[j].kind : AGG_COUNT;
            const bool str = j < a.n_agg && a.aggs[j].is_str;
            s_vals[j][i] = kind == AGG_MIN ? (uint32_t)INT32_MAX : (str ? 0u : (uint32_t)INT32_MIN);
        }
    }
    __syncthreads();

    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < a.n_tiles; tile += (int64_t)gridDim.x * kWavesPerBlock) {
        uint64_t m[kTileWords]; // the tile's bitmap words: wave-uniform
        uint64_t any = 0;
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) {
            m[j] = a.bitmap[tile * kTileWords + j];
            any |= m[j];
        }
        if (!any) continue; // nothing selected in these 1024 rows
        uint32_t key[kTileWords];
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) key[j] = 0;
        for (int g = 0; g < a.n_group; ++g) {
            uint32_t v[kTileWords];
            load_tile_rows(a.groups[g].data, a.groups[g].width, tile, lane, xp, v);
            const int sh = 8 * a.groups[g].shift;
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) key[j] |= v[j] << sh;
        }
        uint32_t slot[kTileWords];
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) {
            slot[j] = kFastSlots;
            if (__builtin_amdgcn_inverse_ballot_w64(m[j])) { // exec = the bitmap word: only selected rows run this
                const uint32_t k = key[j];
                const uint32_t row = (uint32_t)(tile * kTileRows + 64 * j + lane);
                uint32_t s = kFastSlots; // the all-ones key has a slot of its own (all-ones marks an empty slot)
                if (k != kEmpty32) {
                    s = (k * 0x9E3779B1u) >> 22;
                    int probes = 0;
                    for (;; ++probes) {
                        uint32_t cur = s_keys[s]; // plain read first: after warm-up the key is there
                        if (cur == kEmpty32) cur = atomicCAS(&s_keys[s], kEmpty32, k);
                        if (cur == kEmpty32 || cur == k) break;
                        if (probes == 64) { // crowded table: the host re-runs the general kernel
                            *a.overflow = 2;
                            s = kFastSlots;
                            break;
                        }
                        s = (s + 1) & (kFastSlots - 1);
                    }
                }
                slot[j] = s;
                if (!IMM3_ABLATED(a, 12)) {
                    atomicAdd(&s_count[s], 1u);
                    atomicMin(&s_first[s], row);
                }
            }
        }
        for (int q = 0; q < a.n_agg; ++q) {
            const int kind = a.aggs[q].kind;
            if (kind == AGG_COUNT || IMM3_ABLATED(a, 13)) continue;
            uint32_t v[kTileWords];
            load_tile_rows(a.aggs[q].data, a.aggs[q].width, tile, lane, xp, v);
            const int w = a.aggs[q].width;
            const bool str = a.aggs[q].is_str;
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) {
                if (__builtin_amdgcn_inverse_ballot_w64(m[j])) {
                    uint32_t *p = &s_vals[q][slot[j]];
                    if (str) { // big-endian pack: integer order == byte-lexicographic order
                        const uint32_t be = w == 4 ? __builtin_bswap32(v[j]) : (w == 2 ? (uint32_t)__builtin_bswap16((uint16_t)v[j]) : v[j]);
                        atomicMax(p, be);
                    } else {
                        const int32_t x = w == 4 ? (int32_t)v[j] : (int32_t)(int8_t)v[j];
                        if (kind == AGG_MIN) atomicMin((int32_t *)p, x);
                        else atomicMax((int32_t *)p, x);
                    }
                }
            }
        }
    }
    __syncthreads();

    // flush: one atomic set per (work-group, group), widened to what the global table holds
    if (IMM3_ABLATED(a, 14)) return;
    for (int i = t; i <= kFastSlots; i += kBlockThreads) {
        if (s_count[i] == 0) continue;
        const unsigned long long key = i == kFastSlots ? (unsigned long long)kEmpty32 : (unsigned long long)s_keys[i];
        const uint32_t g = global_slot(a, key);
        if (g == 0xFFFFFFFFu) continue;
        long long vals[kMaxAggs];
        for (int j = 0; j < kMaxAggs; ++j) {
            const bool str = j < a.n_agg && a.aggs[j].is_str;
            vals[j] = str ? (long long)(unsigned long long)s_vals[j][i] : (long long)(int32_t)s_vals[j][i];
        }
        agg_update_global(a, g, s_first[i], (unsigned long long)s_count[i], vals);
    }
}

// ---------------------------------------------------------------------------------------------
// k_group_agg_direct: group key <= 2 bytes.  The key indexes a byte map in LDS (256 B or 64 KiB) that names the group's
// dense slot (<= 254 distinct keys per work-group, assigned on first sight), so a row's slot costs ONE ds_read_u8 with no
// key compare, no probing and no branch -- and all 16 words of a tile issue their map reads back to back, then their
// count / first-seen / value atomics back to back: three LDS waits per TILE instead of two per WORD (k_group_agg_tile's
// 16 serialised probe round trips per tile were what its time was made of).  One 1024-thread work-group per CU shares the
// map and the tables (16 waves), which also keeps the flush small: groups x work-groups global atomics on a few hot
// addresses serialise at ~12 ns each.
// ---------------------------------------------------------------------------------------------
constexpr int kDirectThreads = 1024;
constexpr int kDirectWaves = kDirectThreads / 64;
constexpr int kDirectSlots = 256; // slot 255 = "not assigned yet" in the map

__global__ __launch_bounds__(kDirectThreads) void k_group_agg_direct(const AggArgs a, const int map_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_map[]; // [map_bytes] key -> slot
    __shared__ uint32_t s_slotkey[kDirectSlots];
    __shared__ uint32_t s_first[kDirectSlots];
    __shared__ uint32_t s_count[kDirectSlots];
    __shared__ uint32_t s_vals[kMaxAggs][kDirectSlots];
    __shared__ uint32_t s_nslots;
    __shared__ __attribute__((aligned(16))) uint8_t s_xp[kDirectWaves][2048];
    // LDS atomics retire ~1.4 lanes per cycle per CU whatever the addresses (measured: each atomic per row costs ~115 us per
    // 100 M rows), so they are the budget.  The first-seen row needs none after warm-up: a wave walks its tiles in ascending
    // order, so only its FIRST rows of a slot can lower the minimum -- a per-wave byte map remembers which slots it has seen
    // (plain read per row, plain write on first sight).
    __shared__ uint8_t s_seen[kDirectWaves][kDirectSlots];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    uint8_t *xp = s_xp[wave];
    uint8_t *seen = s_seen[wave];
    for (int i = t; i < kDirectWaves * kDirectSlots / 4; i += kDirectThreads) ((uint32_t *)s_seen)[i] = 0u;
    for (int i = t; i < map_bytes / 4; i += kDirectThreads) ((uint32_t *)s_map)[i] = 0xFFFFFFFFu;
    for (int i = t; i < kDirectSlots; i += kDirectThreads) {
        s_first[i] = 0xFFFFFFFFu;
        s_count[i] = 0;
        for (int j = 0; j < kMaxAggs; ++j) {
            const int kind = j < a.n_agg ? a.aggs[j].kind : AGG_COUNT;
            const bool str = j < a.n_agg && a.aggs[j].is_str;
            s_vals[j][i] = kind == AGG_MIN ? (uint32_t)INT32_MAX : (str ? 0u : (uint32_t)INT32_MIN);
        }
    }
    if (t == 0) s_nslots = 0;
    __syncthreads();

    for (int64_t tile = (int64_t)blockIdx.x * kDirectWaves + wave; tile < a.n_tiles; tile += (int64_t)gridDim.x * kDirectWaves) {
        uint64_t m[kTileWords]; // the tile's bitmap words: wave-uniform
        uint64_t any = 0;
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) {
            m[j] = a.bitmap[tile * kTileWords + j];
            any |= m[j];
        }
        if (!any) continue; // nothing selected in these 1024 rows
        uint32_t key[kTileWords];
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) key[j] = 0;
        for (int g = 0; g < a.n_group; ++g) {
            uint32_t v[kTileWords];
            load_tile_rows(a.groups[g].data, a.groups[g].width, tile, lane, xp, v);
            const int sh = 8 * a.groups[g].shift;
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) key[j] |= v[j] << sh;
        }
        // slots: 16 map reads back to back
        uint32_t sid[kTileWords];
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) {
            sid[j] = 0;
            if (__builtin_amdgcn_inverse_ballot_w64(m[j])) sid[j] = s_map[key[j]];
        }
        uint64_t bad = 0;
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) bad |= m[j] & ballot64(sid[j] == 255u);
        if (bad) { // wave-uniform, warm-up only: give every unseen key of this tile a slot, one distinct key at a time
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) { // (unrolled: a run-time index would push key[] / sid[] into scratch memory)
                uint64_t todo = m[j] & ballot64(sid[j] == 255u);
                while (todo) {
                    const int leader = __builtin_ctzll(todo);
                    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key[j], leader);
                    uint32_t id = 255u;
                    if (lane == 0) {
                        uint32_t *w32 = (uint32_t *)s_map + (k >> 2);
                        const int sh = 8 * (int)(k & 3u);
                        for (int tries = 0; tries < 64 && id == 255u; ++tries) {
                            const uint32_t cur = *(volatile uint32_t *)w32;
                            const uint32_t b = (cur >> sh) & 0xFFu;
                            if (b != 255u) { id = b; break; }
                            const uint32_t mine = atomicAdd(&s_nslots, 1u);
                            if (mine >= 255u) { *a.overflow = 2; id = 0; break; } // too many distinct keys for this form: the host re-runs the general kernel
                            const uint32_t want = cur ^ ((255u ^ mine) << sh);
                            if (atomicCAS(w32, cur, want) == cur) { id = mine; s_slotkey[mine] = k; }
                            // else: another wave changed this dword meanwhile (the slot number `mine` is simply not used)
                        }
                        if (id == 255u) { *a.overflow = 2; id = 0; }
                    }
                    id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
                    const uint64_t same = ballot64(key[j] == k) & todo;
                    if ((same >> lane) & 1ULL) sid[j] = id;
                    todo &= ~same;
                }
            }
        }
        // count: 16 atomics back to back; first-seen row: 16 plain reads of the wave's seen map, atomics only on first sight
        uint32_t was[kTileWords];
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) {
            was[j] = 1u;
            if (__builtin_amdgcn_inverse_ballot_w64(m[j])) {
                atomicAdd(&s_count[sid[j]], 1u);
                was[j] = seen[sid[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) {
            if (was[j] == 0u) { // (all lanes of this word that meet the slot for the first time: the atomic keeps the lowest row)
                atomicMin(&s_first[sid[j]], (uint32_t)(tile * kTileRows + 64 * j + lane));
                seen[sid[j]] = 1;
            }
        }
        // values
        for (int q = 0; q < a.n_agg; ++q) {
            const int kind = a.aggs[q].kind;
            if (kind == AGG_COUNT) continue;
            uint32_t late[kTileWords];
            load_tile_rows(a.aggs[q].data, a.aggs[q].width, tile, lane, xp, late);
            const int w = a.aggs[q].width;
            const bool str = a.aggs[q].is_str;
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) {
                if (__builtin_amdgcn_inverse_ballot_w64(m[j])) {
                    const uint32_t raw = late[j];
                    uint32_t *p = &s_vals[q][sid[j]];
                    if (str) { // big-endian pack: integer order == byte-lexicographic order
                        const uint32_t be = w == 4 ? __builtin_bswap32(raw) : (w == 2 ? (uint32_t)__builtin_bswap16((uint16_t)raw) : raw);
                        atomicMax(p, be);
                    } else {
                        const int32_t x = w == 4 ? (int32_t)raw : (int32_t)(int8_t)raw;
                        if (kind == AGG_MIN) atomicMin((int32_t *)p, x);
                        else atomicMax((int32_t *)p, x);
                    }
                }
            }
        }
    }
    __syncthreads();

    // flush: one atomic set per (work-group, group), widened to what the global table holds
    const uint32_t n_slots = s_nslots < 255u ? s_nslots : 255u;
    for (uint32_t i = t; i < n_slots; i += kDirectThreads) {
        if (s_count[i] == 0) continue;
        const uint32_t g = global_slot(a, (unsigned long long)s_slotkey[i]);
        if (g == 0xFFFFFFFFu) continue;
        long long vals[kMaxAggs];
        for (int j = 0; j < kMaxAggs; ++j) {
            const bool str = j < a.n_agg && a.aggs[j].is_str;
            vals[j] = str ? (long long)(unsigned long long)s_vals[j][i] : (long long)(int32_t)s_vals[j][i];
        }
        agg_update_global(a, g, s_first[i], (unsigned long long)s_count[i], vals);
    }
}

// ---------------------------------------------------------------------------------------------
// k_group_agg_lanes: group key <= 2 bytes, at most 63 distinct keys, one min / max aggregate (or two over 1-byte columns) and
// any number of counts -- NO ATOMICS on the row path.  LDS atomics retire ~1.4 lanes per cycle per CU on this chip whatever the addresses, which made two atomics
// per row (count + max) 230 us per 100 M rows in k_group_agg_direct.  Here every LANE owns a private table
// [slot][lane] in LDS, so an update is a plain ds_read + ds_write at full LDS rate that never conflicts (lane l always
// hits bank l).  What that needs:
//   * a lane takes 16 CONSECUTIVE rows of a tile (16 / 32 / 64 contiguous bytes per column, no transposition through
//     LDS; its 16 bitmap bits are one u16) and updates its entries row after row: read, modify, write -- LDS executes one
//     wave's operations in order, so the next row's read sees the write without a wait;
//   * rows that are not selected update a trash slot instead of branching;
//   * values are mapped to an "unsigned max" domain on load (sign bit flipped, complemented for MIN, strings packed
//     big-endian), so the update is always v_max_u32 and tables start at zero.  Values of <= 2 bytes share a dword with
//     the count (count low, value high: one read + one write per row); 4-byte values have an array of their own;
//   * key -> slot through a two-level byte map (first key byte -> page, page[second byte] -> slot): exact like the 64 KiB
//     map of the direct form at 1/8 of the LDS, which is what lets 9 waves' tables fit beside it.  New keys are placed
//     lock-free by the lanes that meet them (claim the map byte with a CAS, take the next slot, publish);
//   * the first-seen row: a wave remembers the slots it has met in one 64-bit scalar; while that differs from the slots
//     handed out so far (warm-up, or a key this wave has not met) its rows of new slots do an atomic min.
// Too many keys / pages raises overflow = 3 and the host re-runs k_group_agg_direct.
// ---------------------------------------------------------------------------------------------
constexpr int kLaneSlots = 128; // most per-lane table rows of any instance (template parameter NS: 64 or 128); the last row is the trash slot
constexpr int kLanePages = 30;  // second-level pages: 0 is the null page (every first byte starts there: all entries free), 1 .. 29 real
constexpr uint32_t kMapFree = 255u, kMapClaimed = 254u, kMapFull = 253u; // second-level bytes that are not a slot number
// first-level bytes: 0 = free (the null page), 1 .. 29 = page, 254 / 253 = claimed / full -- a lookup clamps those to page
// kLanePages, a second null page, so that the row path needs no compare

struct LanesShared { // fixed part of the dynamic LDS; [l2 .. first] start as all-ones, the rest as zero
    uint8_t l2[(kLanePages + 1) * 256];
    uint32_t first[kLaneSlots];
    uint8_t l1[256];
    uint32_t slotkey[kLaneSlots];
    uint32_t count[kLaneSlots];
    uint32_t val[kLaneSlots];
    uint32_t val2[kLaneSlots];
    uint32_t nslots, npages, pad[2];
};
constexpr int kLanesOnesBytes = (kLanePages + 1) * 256 + kLaneSlots * 4;
constexpr int kLanesFixedBytes = (int)((sizeof(LanesShared) + 255) / 256 * 256);
constexpr int lanes_wave_bytes(int vw, bool v2 = false, int ns = 64) {
    return v2 ? ns * 64 * 4 + ns * 4 : (vw <= 1 ? ns * 64 * 2 + ns * 4 : ns * 64 * (vw == 4 ? 2 + 4 : 4));
}

typedef int v4i_t __attribute__((ext_vector_type(4)));

// rows 16 * lane .. 16 * lane + 15 of tile `tile` of a W-byte column: W contiguous 16-byte pieces per lane
template <int W>
__device__ __forceinline__ void load_lane_rows(const void *data, int64_t tile, int lane, v4i_t (&r)[W == 4 ? 4 : (W == 2 ? 2 : 1)]) {
    constexpr int N = W == 4 ? 4 : (W == 2 ? 2 : 1);
    const v4i_t *p = (const v4i_t *)((const uint8_t *)data + tile * (int64_t)(kTileRows * W)) + lane * N;
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_nontemporal_load(p + i);
}
// A 2-byte column loaded the COALESCED way (round 4): lane l takes bytes [16 l, 16 l + 16) of the tile's first and of its second
// kilobyte -- two instructions that each cover contiguous bytes -- instead of 32 contiguous bytes per lane, which is two
// instructions that each use every other 16-byte piece (loads like that run well below the rate of contiguous ones: DESIGN
// finding 20; this kernel's loads alone took 60 us for 312 MB).  Lane pairs then swap halves (one DPP move per register): the even
// lane 2 k ends up with rows 16 k .. 16 k + 15, the odd lane 2 k + 1 with rows 512 + 16 k .. -- sixteen consecutive rows per lane as
// before, in the "pair" lane order that the 1-byte columns and the bitmap bits are loaded in directly (pair_group below).
__device__ __forceinline__ void load_lane_rows_pair2(const void *data, int64_t tile, int lane, v4i_t (&r)[2]) {
    const v4i_t *p = (const v4i_t *)((const uint8_t *)data + tile * (int64_t)(kTileRows * 2)) + lane;
    r[0] = __builtin_nontemporal_load(p);
    r[1] = __builtin_nontemporal_load(p + 64);
}
__device__ __forceinline__ void pair_swap2(v4i_t (&r)[2], bool odd) {
    v4i_t xa, xb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xa[k] = __builtin_amdgcn_update_dpp(0, r[0][k], 0xB1, 0xF, 0xF, true); // quad_perm [1, 0, 3, 2]: the pair neighbour's register
        xb[k] = __builtin_amdgcn_update_dpp(0, r[1][k], 0xB1, 0xF, 0xF, true);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a0 = r[0][k], b1 = r[1][k];
        r[0][k] = odd ? xb[k] : a0;
        r[1][k] = odd ? b1 : xa[k];
    }
}
// the 16-row group of a tile that lane l owns in pair order: even lanes the first half of the tile, odd lanes the second
__device__ __forceinline__ int pair_group(int lane) { return (lane >> 1) + 32 * (lane & 1); }

template <int W>
__device__ __forceinline__ uint32_t lane_row_value(const v4i_t (&r)[W == 4 ? 4 : (W == 2 ? 2 : 1)], int i) { // i: compile time after unrolling
    if constexpr (W == 4) return (uint32_t)r[i >> 2][i & 3];
    else if constexpr (W == 2) return ((uint32_t)r[i >> 3][(i >> 1) & 3] >> (16 * (i & 1))) & 0xFFFFu;
    else return ((uint32_t)r[0][i >> 2] >> (8 * (i & 3))) & 0xFFu;
}

// Bit i = row i of the lane's sixteen int8 rows lies in [lo, lo + span]: three vector instructions per row (subtract with byte
// select, compare, add-with-carry as shift-and-insert; imm3_tile.h: ColRegs<TK_I8>::rows4 -- the compiler's own code for the loop is
// five and a wait state).
__device__ __forceinline__ uint32_t int8_rows_mask(const v4i_t &rows, uint32_t lo, uint32_t span) {
    uint32_t m = 0;
    ColRegs<TK_I8>::rows4(m, rows[3], lo, span); // (row 15 first: each step shifts the rows so far up)
    ColRegs<TK_I8>::rows4(m, rows[2], lo, span);
    ColRegs<TK_I8>::rows4(m, rows[1], lo, span);
    ColRegs<TK_I8>::rows4(m, rows[0], lo, span);
    return m;
}

// The same with a RUN-TIME row index (the sparse walk of k_group_agg_lanes: every lane takes its own next selected row).  Registers
// cannot be indexed per lane, so the element is picked with byte permutes (v_perm_b32 selects any four bytes of a register pair)
// and a few conditional moves: 5 vector instructions for a 1-byte column, ~11 for a 2-byte one, ~19 for int32.
template <int W>
__device__ __forceinline__ uint32_t lane_row_value_rt(const v4i_t (&r)[W == 4 ? 4 : (W == 2 ? 2 : 1)], uint32_t i) {
    if constexpr (W == 1) {
        const uint32_t sel = 0x0C0C0C00u | (i & 7u); // byte i & 7 of the pair, zero-extended
        const uint32_t lo = __builtin_amdgcn_perm((uint32_t)r[0][1], (uint32_t)r[0][0], sel), hi = __builtin_amdgcn_perm((uint32_t)r[0][3], (uint32_t)r[0][2], sel);
        return (i & 8u) ? hi : lo;
    } else if constexpr (W == 2) {
        const uint32_t e = i & 3u, sel = 0x0C0C0100u + 0x0202u * e; // bytes 2 e, 2 e + 1 of the pair
        const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)r[0][1], (uint32_t)r[0][0], sel), p1 = __builtin_amdgcn_perm((uint32_t)r[0][3], (uint32_t)r[0][2], sel);
        const uint32_t p2 = __builtin_amdgcn_perm((uint32_t)r[1][1], (uint32_t)r[1][0], sel), p3 = __builtin_amdgcn_perm((uint32_t)r[1][3], (uint32_t)r[1][2], sel);
        const uint32_t a = (i & 4u) ? p1 : p0, b = (i & 4u) ? p3 : p2;
        return (i & 8u) ? b : a;
    } else {
        uint32_t q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = (i & 1u) ? (uint32_t)r[k][1] : (uint32_t)r[k][0], b = (i & 1u) ? (uint32_t)r[k][3] : (uint32_t)r[k][2];
            q[k] = (i & 2u) ? b : a;
        }
        const uint32_t a = (i & 4u) ? q[1] : q[0], b = (i & 4u) ? q[3] : q[2];
        return (i & 8u) ? b : a;
    }
}

// one byte of a map: its value, or -- when it is free -- an attempt to claim it and fill it from *counter (values >= limit
// become `full` and raise the overflow flag).  Returns `free` when the byte is being filled by someone else right now (or
// the CAS lost against a neighbour byte): ask again.  Straight-line between claim and publish, so lanes of one wave
// cannot wait on each other.
__device__ __forceinline__ uint32_t map_byte(uint8_t *map, uint32_t idx, uint32_t free, uint32_t *counter, uint32_t limit, uint32_t full, uint32_t *overflow, bool &placed) {
    uint32_t *w = (uint32_t *)map + (idx >> 2);
    const int sh = 8 * (int)(idx & 3u);
    const uint32_t cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (not `volatile`: that turns an LDS access into a flat one, which waits for every global load in flight)
    uint32_t b = (cur >> sh) & 0xFFu;
    placed = false;
    if (b == free) {
        if (atomicCAS(w, cur, cur ^ ((free ^ kMapClaimed) << sh)) != cur) return free;
        uint32_t n = atomicAdd(counter, 1u);
        if (n >= limit) {
            n = full;
            *overflow = 3;
        } else placed = true;
        atomicXor(w, (kMapClaimed ^ n) << sh);
        b = n;
    }
    return b == kMapClaimed ? free : b;
}

// slot of key k: 0 .. trash - 1, `trash` when the form is full, kMapFree: not placed yet, ask again
template <int KS>
__device__ __forceinline__ uint32_t lanes_slot(LanesShared &S, uint32_t k, uint32_t trash, uint32_t *overflow) {
    uint32_t pg = 1; // one-byte keys: page 1, no first level
    bool placed;
    if constexpr (KS != 0) {
        pg = map_byte(S.l1, k & 0xFFu, 0u, &S.npages, (uint32_t)kLanePages, kMapFull, overflow, placed); // (npages starts at 1)
        if (pg == 0u) return kMapFree;
        if (pg == kMapFull) return trash;
    }
    const uint32_t id = map_byte(S.l2, (pg << 8) | (KS == 0 ? (k & 0xFFu) : (k >> 8)), kMapFree, &S.nslots, trash, trash, overflow, placed);
    if (placed) S.slotkey[id] = k; // (read after the work-group barrier that precedes the flush)
    return id;
}

__device__ __forceinline__ uint64_t wave_or64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo |= (uint32_t)__shfl_xor((int)lo, d);
        hi |= (uint32_t)__shfl_xor((int)hi, d);
    }
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
}

// KS: key shape -- 0: one 1-byte column, 1: one 2-byte column, 2: two 1-byte columns.  VW: bytes of the aggregated column (0: counts only).
// V2: a second min / max aggregate, both over 1-byte columns (e.g. max(age), min(age)).
// NS: rows of a lane's table (64 or 128, the last one the trash slot): 63 or 127 distinct keys per work-group.
// FUSED: the select chain is evaluated HERE, on the sixteen rows a lane holds -- a.fused[] closed intervals over int8 / int32
// columns (a predicate on the value aggregate's own column reads nothing more), or no predicate at all -- instead of being read
// from the bitmap a filter launch wrote: SelectOp fused into ProjectAggOp (ProjectAggregate.scala:158-177 walks the selected
// positions of the batch it was handed).  Round 4 ran `group by state where age in (18, 30)` as filter 22 us + aggregation 82 us
// with the bitmap written and read back and `age` read twice.
template <int KS, int VW, bool V2, int NS, bool FUSED>
__global__ __launch_bounds__(1024) void k_group_agg_lanes(const AggArgs a, const int vq, const int vq2) {
    static_assert(!V2 || VW == 1, "two value aggregates: 1-byte columns only");
    static_assert(NS == 64 || NS == 128, "the map's markers (253 .. 255) must have a bit set that no slot number has");
    constexpr uint32_t kTrash = NS - 1;
    constexpr int kWords = NS / 64; // 64-bit words of a slot set
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[]; // the waves' private tables
    __shared__ LanesShared S; // (static: its addresses fold into the LDS instructions' offset fields)
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n_waves = (int)(blockDim.x >> 6);
    constexpr int kWaveBytes = lanes_wave_bytes(VW, V2, NS);
    constexpr int NV = VW == 4 ? 4 : (VW == 2 ? 2 : 1);
    // entry of one (slot, lane):  VW 0: u16 count.  VW 1: u16 = value << 8 | count, the 8-bit counts folded into a per-wave
    // u32 table every 15 tiles (a lane adds at most 16 per tile).  VW 2: u32 = value << 16 | count.  VW 4: u16 count + u32 value.
    // The 16-bit forms halve the tables, which is what doubles the waves per CU (16): the row path is latency-bound.
    // V2: u32 = value2 << 16 | value << 8 | count, counts folded like VW 1.
    constexpr bool kE16 = VW <= 1 && !V2;
    constexpr bool kPacked = VW == 2;
    uint8_t *wbase = s_dyn + wave * kWaveBytes;
    uint16_t *t16 = (uint16_t *)wbase + (lane & 31) * 2 + (lane >> 5);      // 16-bit: [slot * 64]; lanes l and l + 32 share a dword, so each half-wave hits 32 banks
    uint32_t *wcnt = (uint32_t *)(wbase + NS * 64 * (V2 ? 4 : 2));  // VW 1: [slot] counts folded so far (owned by lane = slot)
    uint32_t *tab = (uint32_t *)wbase + lane;                               // packed: [slot * 64]
    uint16_t *cnt = (uint16_t *)wbase + lane;                               // VW == 4: counts [slot * 64] ...
    uint32_t *val = (uint32_t *)(wbase + NS * 64 * 2) + lane;               // ... and values [slot * 64]
    for (int i = t; i < (int)(sizeof(LanesShared) / 4); i += (int)blockDim.x) ((uint32_t *)&S)[i] = i < kLanesOnesBytes / 4 ? 0xFFFFFFFFu : 0u; // l2 + first: ones
    for (int i = t; i < n_waves * kWaveBytes / 4; i += (int)blockDim.x) ((uint32_t *)s_dyn)[i] = 0u;
    __syncthreads();
    if (t == 0) S.npages = 1; // page 0 is the null page
    __syncthreads();
    if (IMM3_ABLATED(a, 46)) return; // (timing only: the launch and the clearing of the tables)

    uint32_t vflip = 0, vmask = 0;
    bool vstr = false;
    if constexpr (VW != 0) {
        vmask = VW == 4 ? 0xFFFFFFFFu : ((1u << (8 * VW)) - 1u);
        vstr = a.aggs[vq].is_str != 0;
        vflip = (vstr ? 0u : (1u << (8 * VW - 1))) ^ (a.aggs[vq].kind == AGG_MIN ? vmask : 0u);
    }
    uint32_t vflip2 = 0;
    bool vstr2 = false, same2 = false;
    if constexpr (V2) {
        vstr2 = a.aggs[vq2].is_str != 0;
        vflip2 = (vstr2 ? 0u : 0x80u) ^ (a.aggs[vq2].kind == AGG_MIN ? 0xFFu : 0u);
        same2 = a.aggs[vq2].data == a.aggs[vq].data; // e.g. max(age), min(age): one load serves both
    }
    const int sh0 = 8 * a.groups[0].shift, sh1 = KS == 2 ? 8 * a.groups[1].shift : 0;
    uint64_t seen[kWords] = {}; // slots this wave has met (wave-uniform)

    // Software pipeline, kDepth tiles deep: with one read + one wait per tile a wave spent its time in two dependent HBM round
    // trips per tile (bitmap, then columns: 128 of the first version's 173 us).  Every register set is re-loaded right
    // after its tile is done and used kDepth tiles later; loads past the wave's last tile re-read the last tile (no branch).
    struct TileRegs {
        uint32_t bits; // rows 16 * lane ..: bits 16 * lane .. of the tile's 1024
        v4i_t kr0[KS == 1 ? 2 : 1], kr1[1], vr[NV], vr2[1];
        v4i_t pr[1]; // FUSED: the rows of the ONE int8 predicate column that is not the value aggregate's (registers: the kernel sits at its 128)
    };
    constexpr int kDepth = VW == 2 ? 3 : 2; // (16-bit entries: 16 waves per CU, two tiles ahead suffice; VW 4: registers)
    // lane -> rows: 16 consecutive rows per lane; with an int32 value column group `lane`, otherwise group pair_group(lane), which
    // lets the 2-byte columns be loaded with contiguous instructions (load_lane_rows_pair2)
    constexpr bool kPair = VW != 4;
    const int rgroup = kPair ? pair_group(lane) : lane;
    const bool odd = (lane & 1) != 0;
    const int64_t stride = (int64_t)gridDim.x * n_waves;
    const int64_t first_tile = (int64_t)blockIdx.x * n_waves + wave;
    int own_pred = -1; // FUSED: the predicate whose int8 column is loaded for it alone (at most one: fused_args_ok)
    if constexpr (FUSED) {
        for (int f = 0; f < kMaxAggPreds; ++f)
            if (f < a.n_fused && !a.fused[f].share) own_pred = f;
    }
    auto issue = [&](TileRegs &r, int64_t tile) {
        if (tile >= a.n_tiles) tile = a.n_tiles - 1;
        if constexpr (FUSED) {
            if (own_pred >= 0) load_lane_rows<1>(a.fused[own_pred].data, tile, rgroup, r.pr); // (wave-uniform: a kernel argument)
        } else r.bits = ((const uint16_t *)a.bitmap)[tile * 64 + rgroup];
        if constexpr (KS == 1 && kPair) load_lane_rows_pair2(a.groups[0].data, tile, lane, r.kr0);
        else load_lane_rows<(KS == 1 ? 2 : 1)>(a.groups[0].data, tile, rgroup, r.kr0);
        if constexpr (KS == 2) load_lane_rows<1>(a.groups[1].data, tile, rgroup, r.kr1);
        if constexpr (VW == 2) load_lane_rows_pair2(a.aggs[vq].data, tile, lane, r.vr);
        else if constexpr (VW != 0) load_lane_rows<(VW ? VW : 1)>(a.aggs[vq].data, tile, rgroup, r.vr);
        if constexpr (V2) {
            if (!same2) load_lane_rows<1>(a.aggs[vq2].data, tile, rgroup, r.vr2); // wave-uniform
        }
    };
    // the pair swap of the 2-byte columns, once their loads have landed (a tile's registers are rewritten by the next issue)
    auto settle = [&](TileRegs &r, int64_t tile) {
        if constexpr (KS == 1 && kPair) pair_swap2(r.kr0, odd);
        if constexpr (VW == 2) pair_swap2(r.vr, odd);
        if constexpr (FUSED) { // the lane's sixteen selection bits from the rows it holds (SelectIteratorGT / LT / EQ folded: Select.scala:53-162)
            const int64_t left = a.n_rows - (tile * kTileRows + 16 * (int64_t)rgroup); // rows of the segment from this lane's first on
            uint32_t bits = left >= 16 ? 0xFFFFu : (left <= 0 ? 0u : ((1u << (uint32_t)left) - 1u));
#pragma unroll
            for (int f = 0; f < kMaxAggPreds; ++f) {
                if (f < a.n_fused) { // (wave-uniform)
                    const uint32_t lo = (uint32_t)a.fused[f].lo, span = (uint32_t)a.fused[f].hi - (uint32_t)a.fused[f].lo;
                    uint32_t ok = 0;
                    if (a.fused[f].share) { // the value aggregate's own rows
                        if constexpr (VW == 4) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) ok |= ((lane_row_value<4>(r.vr, i) - lo) <= span ? 1u : 0u) << i;
                        } else if constexpr (VW == 1) {
                            ok = int8_rows_mask(r.vr[0], lo, span);
                        }
                    } else {
                        ok = int8_rows_mask(r.pr[0], lo, span);
                    }
                    bits &= ok;
                }
            }
            r.bits = bits;
        }
    };
    int since_fold = 0; // VW 1: tiles since the 8-bit counts were folded (wave-uniform)
    auto fold_counts = [&]() { // lane = slot (NS / 64 passes): move the 64 lanes' 8-bit counts of its slot into wcnt (rotated: lanes on different banks)
#pragma unroll
        for (int p = 0; p < kWords; ++p) {
            const int slot = 64 * p + lane;
            uint32_t c = 0;
            // (eight reads, then their eight writes: with a write behind every read the compiler -- which cannot know that the rotated
            // indices never collide -- kept the 64 round trips in sequence)
            if constexpr (V2) {
                uint32_t *row = (uint32_t *)wbase + slot * 64;
#pragma unroll 1
                for (int j0 = 0; j0 < 64; j0 += 8) {
                    uint32_t e[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = row[(j0 + k + lane) & 63];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        c += e[k] & 0xFFu;
                        row[(j0 + k + lane) & 63] = e[k] & ~0xFFu;
                    }
                }
            } else {
                uint16_t *row = (uint16_t *)wbase + slot * 64;
#pragma unroll 1
                for (int j0 = 0; j0 < 64; j0 += 8) {
                    uint32_t e[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] = row[(j0 + k + lane) & 63];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        c += e[k] & 0xFFu;
                        row[(j0 + k + lane) & 63] = (uint16_t)(e[k] & 0xFF00u);
                    }
                }
            }
            wcnt[slot] += c;
        }
    };
    // one row into the lane's private entry of slot s (raw / raw2: the aggregated columns' values as loaded)
    auto update_entry = [&](const uint32_t s, uint32_t raw, uint32_t raw2) {
        uint32_t x = 0;
        if constexpr (VW != 0) {
            if (vstr) raw = VW == 4 ? __builtin_bswap32(raw) : (VW == 2 ? (uint32_t)__builtin_bswap16((uint16_t)raw) : raw); // big-endian pack: integer order == byte order
            x = (raw ^ vflip) & vmask;
        }
        if constexpr (V2) {
            const uint32_t x2 = (raw2 ^ vflip2) & 0xFFu; // (1-byte strings need no byte swap)
            const uint32_t old = tab[s * 64];
            tab[s * 64] = ((old + 1u) & 0xFFu) | max(old & 0xFF00u, x << 8) | max(old & 0xFF0000u, x2 << 16);
        } else if constexpr (kE16) {
            const uint32_t old = t16[s * 64];
            if constexpr (VW == 0) t16[s * 64] = (uint16_t)(old + 1u); // (a lane's count stays below 2^16: lanes_plan)
            else t16[s * 64] = (uint16_t)(((old + 1u) & 0xFFu) | (max(old, x << 8) & 0xFF00u));
        } else if constexpr (kPacked) {
            const uint32_t old = tab[s * 64];
            tab[s * 64] = ((old + 1u) & 0xFFFFu) | (max(old, x << 16) & 0xFFFF0000u); // (a lane's count stays below 2^16: lanes_plan)
        } else {
            const uint32_t oc = cnt[s * 64], ov = val[s * 64];
            cnt[s * 64] = (uint16_t)(oc + 1u);
            val[s * 64] = max(ov, x);
        }
    };
    // ProjectAggIterator visits SELECTED rows only (ProjectAggregate.scala:158-159).  The dense walk below touches all sixteen rows
    // of every lane (unselected ones update a trash slot: no branches) -- the same ~440 instructions per tile whether 11 % or all of
    // the rows are selected.  When no lane of the wave has more than kSparseRows selected rows the SPARSE walk takes over: every lane
    // takes its own next selected row per step (lowest set bit of its 16), the row's key and values are picked out of the loaded
    // registers with byte permutes, and the wave makes as many steps as its busiest lane has rows: ~6 at 11 % survivors, ~2 at 2 %.
    constexpr uint32_t kSparseRows = VW == 4 ? 5u : 8u;
    auto process_sparse = [&](const TileRegs &r, const int64_t tile) {
        const uint32_t row0 = (uint32_t)(tile * kTileRows + 16 * rgroup);
        uint32_t rem = r.bits;
        uint64_t here[kWords] = {}; // slots this lane met in this tile while the wave was behind
        bool was_behind = false;
        while (ballot64(rem != 0u)) { // wave-uniform
            const bool on = rem != 0u;
            const uint32_t i = (uint32_t)__builtin_ctz(rem | 0x10000u) & 15u; // this lane's next selected row (row 0 for a lane that has none left: its update goes to the trash slot)
            rem &= rem - 1u;
            uint32_t key = lane_row_value_rt<(KS == 1 ? 2 : 1)>(r.kr0, i);
            if constexpr (KS == 2) key = (key << sh0) | (lane_row_value_rt<1>(r.kr1, i) << sh1);
            uint32_t sid;
            if constexpr (KS == 0) sid = S.l2[256 + key];
            else sid = S.l2[(min((uint32_t)S.l1[key & 0xFFu], (uint32_t)kLanePages) << 8) | (key >> 8)];
            bool pend = on && sid >= (uint32_t)NS;
            if (ballot64(pend)) { // wave-uniform, warm-up only: the lanes place (or look up again) their new keys
                for (int rounds = 0; rounds < 1024 && ballot64(pend); ++rounds) {
                    if (pend) {
                        const uint32_t id = lanes_slot<KS>(S, key, kTrash, a.overflow);
                        if (id != kMapFree) {
                            sid = id;
                            pend = false;
                        }
                    }
                }
                if (ballot64(pend)) *a.overflow = 3; // (never seen: a claimed byte is published a few instructions later)
            }
            sid = (on && sid < (uint32_t)NS) ? sid : kTrash;
            {   // first-seen rows (the slot numbers seen above are below the count read here: slots are numbered before they are published)
                uint32_t ns = __hip_atomic_load(&S.nslots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ns = ns < kTrash ? ns : kTrash;
                bool behind = false;
#pragma unroll
                for (int p = 0; p < kWords; ++p) {
                    const uint32_t n = ns > 64u * p ? ns - 64u * p : 0u;
                    const uint64_t all = n >= 64u ? ~0ULL : ((1ULL << n) - 1ULL);
                    behind |= (seen[p] & all) != all;
                }
                if (behind) { // wave-uniform.  `seen` stays what it was when the tile began: a later step of this walk may bring a
                    // LOWER row of the same slot (another lane's), so every row of the tile is tested against the slots met in EARLIER tiles
                    was_behind = true;
                    const uint64_t bit = 1ULL << (sid & 63u);
                    uint64_t sw = seen[0];
                    if constexpr (kWords == 2) sw = (sid & 64u) ? seen[1] : seen[0];
                    const bool real = on && sid != kTrash;
                    if (real && !(sw & bit)) atomicMin(&S.first[sid], row0 + i);
                    if constexpr (kWords == 2) {
                        here[0] |= (real && !(sid & 64u)) ? bit : 0ULL;
                        here[1] |= (real && (sid & 64u)) ? bit : 0ULL;
                    } else here[0] |= real ? bit : 0ULL;
                }
            }
            uint32_t raw = 0, raw2 = 0;
            if constexpr (VW != 0) raw = lane_row_value_rt<(VW ? VW : 1)>(r.vr, i);
            if constexpr (V2) raw2 = same2 ? lane_row_value_rt<1>(r.vr, i) : lane_row_value_rt<1>(r.vr2, i);
            update_entry(sid, raw, raw2);
        }
        if (was_behind) { // wave-uniform
#pragma unroll
            for (int p = 0; p < kWords; ++p) seen[p] |= wave_or64(here[p]);
        }
    };
    auto process = [&](const TileRegs &r, const int64_t tile) {
        const uint32_t bits = r.bits;
        if (!ballot64(bits != 0u)) return; // nothing selected in these 1024 rows
        const auto &kr0 = r.kr0;
        const auto &kr1 = r.kr1;
        const auto &vr = r.vr;
        const auto &vr2 = r.vr2;
        if (IMM3_ABLATED(a, 44)) { // ablation: loads only
            asm volatile("" ::"v"(kr0[0]), "v"(kr0[KS == 1 ? 1 : 0]), "v"(vr[0]), "v"(vr[NV - 1]));
            return;
        }
        if (!IMM3_ABLATED(a, 45) && !ballot64((uint32_t)__popc(bits) > kSparseRows)) { // (ablation 45: always the dense walk)
            process_sparse(r, tile);
            return;
        }
        uint32_t key[16], sid[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            key[i] = lane_row_value<(KS == 1 ? 2 : 1)>(kr0, i);
            if constexpr (KS == 2) key[i] = (key[i] << sh0) | (lane_row_value<1>(kr1, i) << sh1);
        }
        // slots: all first-level reads, then all second-level reads (values >= 253: no slot yet)
        if constexpr (KS == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sid[i] = S.l2[256 + key[i]];
        } else {
            uint32_t pg[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) pg[i] = S.l1[key[i] & 0xFFu];
#pragma unroll
            for (int i = 0; i < 16; ++i) sid[i] = S.l2[(min(pg[i], (uint32_t)kLanePages) << 8) | (key[i] >> 8)];
        }
        uint32_t any = 0; // any row, selected or not, without a slot?  (valid slots < NS; the map's markers 253 .. 255 have bit 7 -- and 6 -- set)
#pragma unroll
        for (int i = 0; i < 16; ++i) any |= sid[i];
        if (ballot64((any & (0x100u - (uint32_t)NS)) != 0u)) { // wave-uniform, warm-up only: every lane places (or looks up again) its own new keys
            uint32_t pend = 0;                        // this lane's selected rows that need a slot
#pragma unroll
            for (int i = 0; i < 16; ++i) pend |= (sid[i] >= (uint32_t)NS ? 1u : 0u) << i;
            pend &= bits;
            for (int rounds = 0; rounds < 1024 && ballot64(pend != 0u); ++rounds) {
                if (pend) {
                    const int first = __builtin_ctz(pend);
                    uint32_t mine = key[0];
#pragma unroll
                    for (int i = 1; i < 16; ++i) mine = first == i ? key[i] : mine; // (a run-time index would push key[] into scratch memory)
                    const uint32_t id = lanes_slot<KS>(S, mine, kTrash, a.overflow);
                    if (id != kMapFree) {
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (key[i] == mine) {
                                sid[i] = id;
                                pend &= ~(1u << i);
                            }
                    }
                }
                // ... and every lane looks its other pending keys up again: the 64 lanes of this wave -- and the other waves of the
                // work-group -- have just placed up to one key each, which is nearly always all there are (51 states: two rounds
                // instead of the ~15 a lane needed for the distinct keys of its own sixteen rows, one map update at a time)
                if (ballot64(pend != 0u) && !IMM3_ABLATED(a, 49)) { // wave-uniform
                    asm volatile("" ::: "memory"); // (the map is read again, not taken from registers)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        uint32_t again;
                        if constexpr (KS == 0) again = S.l2[256 + key[i]];
                        else again = S.l2[(min((uint32_t)S.l1[key[i] & 0xFFu], (uint32_t)kLanePages) << 8) | (key[i] >> 8)];
                        if (((pend >> i) & 1u) && again < (uint32_t)NS) {
                            sid[i] = again;
                            pend &= ~(1u << i);
                        }
                    }
                }
            }
            if (ballot64(pend != 0u)) *a.overflow = 3; // (never seen: a claimed byte is published a few instructions later)
#pragma unroll
            for (int i = 0; i < 16; ++i) sid[i] = sid[i] < (uint32_t)NS ? sid[i] : kTrash; // keys of rows that are not selected stay unplaced
        }
        if (ballot64(bits != 0xFFFFu)) { // wave-uniform: rows that are not selected update the trash slot
#pragma unroll
            for (int i = 0; i < 16; ++i) sid[i] = ((bits >> i) & 1u) ? sid[i] : kTrash;
        }
        const uint32_t row0 = (uint32_t)(tile * kTileRows + 16 * rgroup);
        // first-seen rows
        {
            uint32_t ns = __hip_atomic_load(&S.nslots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ns = ns < kTrash ? ns : kTrash;
            bool behind = false; // has this wave met every slot handed out so far?
#pragma unroll
            for (int p = 0; p < kWords; ++p) {
                const uint32_t n = ns > 64u * p ? ns - 64u * p : 0u;
                const uint64_t all = n >= 64u ? ~0ULL : ((1ULL << n) - 1ULL);
                behind |= (seen[p] & all) != all;
            }
            if (behind) { // wave-uniform: warm-up, or a key some other wave met and this one has not yet
                uint64_t here[kWords] = {};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool on = ((bits >> i) & 1u) != 0u;
                    const uint64_t bit = 1ULL << (sid[i] & 63u);
                    uint64_t sw = seen[0];
                    if constexpr (kWords == 2) sw = (sid[i] & 64u) ? seen[1] : seen[0];
                    if (on && !(sw & bit)) atomicMin(&S.first[sid[i]], row0 + i);
                    if constexpr (kWords == 2) {
                        here[0] |= (on && !(sid[i] & 64u)) ? bit : 0ULL;
                        here[1] |= (on && (sid[i] & 64u)) ? bit : 0ULL;
                    } else here[0] |= on ? bit : 0ULL;
                }
#pragma unroll
                for (int p = 0; p < kWords; ++p) seen[p] |= wave_or64(here[p]);
            }
        }
        // updates: one row after the other, read -> modify -> write.  LDS executes one wave's operations in order, so a row
        // sees the write of the row before it without a wait in between; the 16 waves hide the round trips.  (Batches of
        // 4 or 2 reads with duplicate slots folded in registers measured slower: 109 / 100 us against 97 -- the fold is
        // vector work, which is what this kernel is short of.)
#pragma unroll
        for (int i = 0; i < (IMM3_ABLATED(a, 42) ? 0 : 16); ++i) {
            uint32_t raw = 0, raw2 = 0;
            if constexpr (VW != 0) raw = lane_row_value<(VW ? VW : 1)>(vr, i);
            if constexpr (V2) raw2 = same2 ? lane_row_value<1>(vr, i) : lane_row_value<1>(vr2, i);
            update_entry(sid[i], raw, raw2);
        }
    };
    TileRegs R[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth; ++d) issue(R[d], first_tile + d * stride);
    // (Tried: the work-group's first wave takes its first tile alone while the others wait, so that one wave places the keys all
    // sixteen would otherwise place at once -- slower, 96 -> 104 us at 100 M rows.  What a wave's first tile costs, ~18 us against
    // ~2.5 for a later one (group by state over 1 M rows: 33 us = 4 launch + clearing, 8 fold, 3 loads, 18 the tiles), is its own
    // first-seen bookkeeping -- an LDS atomicMin per row until the wave has met every slot -- not the collisions in the map.)
    for (int64_t base = first_tile; base < (IMM3_ABLATED(a, 47) ? 0 : a.n_tiles); base += kDepth * stride) { // (47, timing only: no tile at all)
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int64_t tile = base + d * stride;
            if (tile < a.n_tiles) { // wave-uniform
                settle(R[d], tile);
                process(R[d], tile);
            }
            issue(R[d], tile + kDepth * stride);
        }
        if constexpr (VW == 1) { // (with or without a second value)
            since_fold += kDepth;
            if (since_fold + kDepth > 15) { // 15 x 16 rows: the 8-bit counts are still below 256
                fold_counts();
                since_fold = 0;
            }
        }
    }
    __syncthreads();

    // fold: lane = slot (NS / 64 passes); it sums / maxes its slot over the 64 lanes' private entries (rotated so that lanes hit different banks)
    if constexpr (VW == 1) fold_counts();
#pragma unroll
    for (int p = 0; p < kWords; ++p) {
        const int slot = 64 * p + lane;
        uint32_t c = 0, m = 0, m2 = 0;
        if constexpr (VW == 1) c = wcnt[slot];
#pragma unroll 8
        for (int l = 0; l < 64; ++l) { // (eight reads in flight: rolled, this was 64 dependent LDS round trips per wave)
            const int src = (l + lane) & 63;
            if constexpr (V2) {
                const uint32_t e = ((const uint32_t *)wbase)[slot * 64 + src];
                m = max(m, (e >> 8) & 0xFFu);
                m2 = max(m2, (e >> 16) & 0xFFu);
            } else if constexpr (kE16) {
                const uint32_t e = ((const uint16_t *)wbase)[slot * 64 + src];
                if constexpr (VW == 0) c += e;
                else m = max(m, e >> 8);
            } else if constexpr (kPacked) {
                const uint32_t e = ((const uint32_t *)wbase)[slot * 64 + src];
                c += e & 0xFFFFu;
                m = max(m, e >> 16);
            } else {
                c += ((const uint16_t *)wbase)[slot * 64 + src];
                m = max(m, ((const uint32_t *)(wbase + NS * 64 * 2))[slot * 64 + src]);
            }
        }
        if ((uint32_t)slot < kTrash && c) {
            atomicAdd(&S.count[slot], c);
            if constexpr (VW != 0) atomicMax(&S.val[slot], m);
            if constexpr (V2) atomicMax(&S.val2[slot], m2);
        }
    }
    __syncthreads();
    // flush: one atomic set per (work-group, group), widened to what the global table holds
    const uint32_t n_slots = S.nslots < kTrash ? S.nslots : kTrash;
    if (IMM3_ABLATED(a, 48)) return; // (timing only: no flush to the global table)
    if ((uint32_t)t < n_slots && S.count[t]) {
        const uint32_t g = global_slot(a, (unsigned long long)S.slotkey[t]);
        if (g != 0xFFFFFFFFu) {
            long long vals[kMaxAggs] = {0, 0, 0, 0};
            if constexpr (VW != 0) {
                const uint32_t u = (S.val[t] ^ vflip) & vmask;
                vals[vq] = vstr ? (long long)(unsigned long long)u : (VW == 4 ? (long long)(int32_t)u : (VW == 2 ? (long long)(int16_t)u : (long long)(int8_t)u));
            }
            if constexpr (V2) {
                const uint32_t u2 = (S.val2[t] ^ vflip2) & 0xFFu;
                vals[vq2] = vstr2 ? (long long)(unsigned long long)u2 : (long long)(int8_t)u2;
            }
            agg_update_global(a, g, S.first[t], (unsigned long long)S.count[t], vals);
        }
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

// Can the fast form take this aggregation?  (uniform single segment; 32-bit keys and values)
bool group_agg_fast_ok(const AggArgs &a) {
    if (a.word_row_base || a.n_group > 2) return false;
    int key_bytes = 0;
    for (int g = 0; g < a.n_group; ++g) {
        const int w = a.groups[g].width;
        if (a.groups[g].tile_ptrs || (w != 1 && w != 2 && w != 4)) return false;
        key_bytes += w;
    }
    if (key_bytes > 4) return false;
    for (int q = 0; q < a.n_agg; ++q) {
        if (a.aggs[q].tile_ptrs) return false;
        if (a.aggs[q].kind == AGG_COUNT) continue;
        const int w = a.aggs[q].width;
        if (a.aggs[q].is_str ? (w != 1 && w != 2 && w != 4) : (w != 1 && w != 4)) return false;
    }
    return true;
}

// which form of k_group_agg_lanes takes this aggregation (false: none).  waves: as many as the LDS holds.
struct LanesPlan { int ks, vw, vq, vq2, v2, ns, waves, lds_bytes; };
static bool lanes_plan(const AggArgs &a, int ns, LanesPlan &p) {
    p.ns = ns;
    if (!group_agg_fast_ok(a)) return false;
    if (a.n_group == 1 && a.groups[0].width == 1) p.ks = 0;
    else if (a.n_group == 1 && a.groups[0].width == 2) p.ks = 1;
    else if (a.n_group == 2 && a.groups[0].width == 1 && a.groups[1].width == 1 && a.groups[0].shift + a.groups[1].shift == 1) p.ks = 2;
    else return false;
    p.vw = 0;
    p.vq = p.vq2 = 0;
    p.v2 = 0;
    int n_val = 0;
    for (int q = 0; q < a.n_agg; ++q)
        if (a.aggs[q].kind != AGG_COUNT) {
            if (n_val == 0) {
                p.vq = q;
                p.vw = a.aggs[q].width;
            } else p.vq2 = q;
            ++n_val;
        }
    if (n_val > 2) return false;
    if (n_val == 2) { // two min / max aggregates: both over 1-byte columns
        if (p.vw != 1 || a.aggs[p.vq2].width != 1) return false;
        p.v2 = 1;
    }
    const int per_wave = lanes_wave_bytes(p.vw, p.v2 != 0, ns);
    p.waves = std::min(16, (160 * 1024 - kLanesFixedBytes) / per_wave);
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((a.n_tiles + p.waves - 1) / p.waves, 256));
    const int64_t tiles_per_wave = (a.n_tiles + grid * p.waves - 1) / (grid * p.waves);
    if (tiles_per_wave * 16 >= 65536) return false; // a lane's u16 count of one slot must hold all its rows
    p.lds_bytes = p.waves * per_wave; // dynamic part; the maps are static
    return true;
}

// false: the device refused the kernel's dynamic LDS size (the caller falls through to the next form)
template <int KS, int VW, bool V2, int NS, bool FUSED>
static bool launch_lanes(const AggArgs &a, const LanesPlan &p, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    // The dynamic-LDS limit of a kernel function is raised once per DEVICE (a process may drive all eight GPUs, one
    // context each, from several threads): 0 = not yet, 1 = raised, 2 = refused.
    static std::atomic<int> raised[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    int st = raised[dev].load(std::memory_order_acquire);
    if (st == 0) { // (two threads may both get here: the call is idempotent)
        const hipError_t e = hipFuncSetAttribute((const void *)k_group_agg_lanes<KS, VW, V2, NS, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - kLanesFixedBytes);
        if (e != hipSuccess) (void)hipGetLastError();
        st = e == hipSuccess ? 1 : 2;
        raised[dev].store(st, std::memory_order_release);
    }
    if (st != 1) return false;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((a.n_tiles + p.waves - 1) / p.waves, 256)); // one work-group per CU
    IMM3_LAUNCH_LDS((k_group_agg_lanes<KS, VW, V2, NS, FUSED>), (unsigned)grid, p.waves * 64, (size_t)p.lds_bytes, s, ev0, ev1, a, p.vq, p.vq2);
    return true;
}

// The select chain rides in the aggregation launch when the 63-key lanes form takes the query and every predicate is a closed interval
// over an int8 / int32 column (at most kMaxAggPreds of them, or none at all): the FUSED instances exist for NS = 64 only.
static bool fused_args_ok(const AggArgs &a) {
    if (a.n_fused < 0 || a.n_fused > kMaxAggPreds || (a.n_fused == 0 && !a.fused_all)) return false;
    int own = 0;
    for (int f = 0; f < a.n_fused; ++f) {
        if (a.fused[f].width != 1 && a.fused[f].width != 4) return false;
        if (a.fused[f].share) continue;
        // a predicate column of its own: ONE int8 column (four registers per tile in flight; an int32 column would be sixteen, and
        // the kernel sits at the 128 registers its sixteen waves per CU leave it -- such a query keeps the filter launch)
        if (!a.fused[f].data || a.fused[f].width != 1 || ++own > 1) return false;
    }
    return true;
}
bool group_agg_fuses_select(const AggArgs &a) {
    LanesPlan lp;
    if (!fused_args_ok(a) || a.first_form != AGG_FORM_LANES || !lanes_plan(a, 64, lp)) return false;
    for (int f = 0; f < a.n_fused; ++f) // (a shared predicate column is the first value aggregate's, at its width)
        if (a.fused[f].share && (lp.vw != a.fused[f].width || a.aggs[lp.vq].data == nullptr)) return false;
    return true;
}

void launch_group_agg(const AggArgs &a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const int64_t n = (int64_t)a.mask + 2;
    const int init_grid = (int)std::min<int64_t>((n + kBlockThreads - 1) / kBlockThreads, 2048);
    hipLaunchKernelGGL(k_group_init, dim3(init_grid), dim3(kBlockThreads), 0, s, a);
    int key_bytes = 0;
    for (int g = 0; g < a.n_group; ++g) key_bytes += a.groups[g].width;
    LanesPlan lp;
    // The forms, fastest first; a.first_form is where the chain starts (the host raises it when a form's per-work-group
    // table overflowed on this query's keys: AggForm in imm3_internal.h).
    // private per-lane tables, no atomics: 63 keys per work-group, then 127
    if (a.first_form <= AGG_FORM_LANES_WIDE && lanes_plan(a, a.first_form == AGG_FORM_LANES_WIDE ? 128 : 64, lp)) {
        const bool fuse = group_agg_fuses_select(a); // (the host has then skipped the select launch: imm3_api.cpp, run_agg)
#define IMM3_LANES(KS, VW, V2)                                                                       \
    if (lp.ks == KS && lp.vw == VW && (lp.v2 != 0) == V2) {                                          \
        const bool ok = fuse ? launch_lanes<KS, VW, V2, 64, true>(a, lp, s, ev0, ev1)                \
                             : (lp.ns == 64 ? launch_lanes<KS, VW, V2, 64, false>(a, lp, s, ev0, ev1) \
                                            : launch_lanes<KS, VW, V2, 128, false>(a, lp, s, ev0, ev1)); \
        if (ok) return;                                                                              \
    }
        IMM3_LANES(0, 0, false) IMM3_LANES(0, 1, false) IMM3_LANES(0, 2, false) IMM3_LANES(0, 4, false) IMM3_LANES(0, 1, true)
        IMM3_LANES(1, 0, false) IMM3_LANES(1, 1, false) IMM3_LANES(1, 2, false) IMM3_LANES(1, 4, false) IMM3_LANES(1, 1, true)
        IMM3_LANES(2, 0, false) IMM3_LANES(2, 1, false) IMM3_LANES(2, 2, false) IMM3_LANES(2, 4, false) IMM3_LANES(2, 1, true)
#undef IMM3_LANES
    }
    if (a.first_form <= AGG_FORM_DIRECT && group_agg_fast_ok(a) && key_bytes <= 2) { // the key indexes a slot map directly
        const int map_bytes = key_bytes <= 1 ? 256 : 65536;
        const int64_t want = (a.n_tiles + kDirectWaves - 1) / kDirectWaves;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, 256)); // one 1024-thread work-group per CU
        IMM3_LAUNCH_LDS(k_group_agg_direct, grid, kDirectThreads, (size_t)map_bytes, s, ev0, ev1, a, map_bytes);
        return;
    }
    if (a.first_form <= AGG_FORM_TILE && group_agg_fast_ok(a)) {
        const int64_t want = (a.n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, 768)); // ~41 KiB of LDS: 3 per CU
        IMM3_LAUNCH(k_group_agg_tile, grid, kBlockThreads, s, ev0, ev1, a);
        return;
    }
    const int64_t want = ((a.n_words + 3) / 4 + kAggWaves - 1) / kAggWaves;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, 768)); // 512-thread work-groups, 48 KiB LDS: 3 per CU
    IMM3_LAUNCH(k_group_agg, grid, kAggThreads, s, ev0, ev1, a);
}

// ---- merge of group tables (imm3_comm_merge_groups): direct-indexed tables for keys of <= 2 bytes, a hash table for wider keys ----
__global__ __launch_bounds__(kBlockThreads) void k_merge_init(const MergeArgs a) {
    for (uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x; i < a.slots; i += gridDim.x * kBlockThreads) {
        a.t_counts[i] = 0ULL;
        a.t_first[i] = ~0ULL;
        if (a.t_keys) a.t_keys[i] = kEmptyKey;
        for (int j = 0; j < kMaxAggs; ++j) {
            const int kind = j < a.n_agg ? a.kinds[j] : AGG_COUNT;
            a.t_vals[(size_t)j * a.slots + i] = kind == AGG_MIN ? INT64_MAX : ((j < a.n_agg && a.is_str[j]) ? 0 : INT64_MIN);
        }
    }
    if (a.out_n && blockIdx.x == 0 && threadIdx.x == 0) *a.out_n = 0ULL;
}

// the key's slot: the key itself in a direct table; in the hash table the first slot from its hash on that is free (claimed with a
// compare-and-swap) or already the key's.  The table holds at least twice the entries that can arrive.
__device__ __forceinline__ uint32_t merge_slot(const MergeArgs &a, unsigned long long key) {
    if (!a.t_keys) return key < a.slots ? (uint32_t)key : 0xFFFFFFFFu; // (a key is narrower than its direct table)
    if (key == kEmptyKey) return a.mask + 1;
    uint32_t g = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 32) & a.mask;
    for (uint32_t probes = 0; probes <= a.mask; ++probes) {
        const unsigned long long prev = atomicCAS(&a.t_keys[g], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) return g;
        g = (g + 1) & a.mask;
    }
    return 0xFFFFFFFFu;
}
__device__ __forceinline__ void merge_update(const MergeArgs &a, uint32_t slot, unsigned long long first, unsigned long long count, const long long *vals) {
    if (slot == 0xFFFFFFFFu) return;
    atomicAdd(a.t_counts + slot, count);
    atomicMin(a.t_first + slot, first);
    for (int j = 0; j < a.n_agg; ++j) {
        const long long v = vals[j];
        long long *t = a.t_vals + (size_t)j * a.slots + slot;
        if (a.kinds[j] == AGG_MAX) {
            if (a.is_str[j]) atomicMax((unsigned long long *)t, (unsigned long long)v);
            else atomicMax(t, v);
        } else if (a.kinds[j] == AGG_MIN) atomicMin(t, v);
    }
}

// one query's dense group list (k_group_collect's output) into the table
__global__ __launch_bounds__(kBlockThreads) void k_merge_scatter(const MergeArgs a) {
    for (uint32_t g = blockIdx.x * kBlockThreads + threadIdx.x; g < a.n_groups; g += gridDim.x * kBlockThreads)
        merge_update(a, merge_slot(a, a.keys[g]), a.seg_hi | (unsigned long long)a.first[g], a.counts[g], a.vals + (size_t)g * kMaxAggs);
}
// a packed list (the ranks' lists after the all-gather; padding entries have count 0) into the table
__global__ __launch_bounds__(kBlockThreads) void k_merge_insert_list(const MergeArgs a) {
    for (uint32_t g = blockIdx.x * kBlockThreads + threadIdx.x; g < a.n_groups; g += gridDim.x * kBlockThreads) {
        const unsigned long long *w = a.list + (size_t)g * kMergeListWords;
        if (!w[2]) continue;
        merge_update(a, merge_slot(a, w[0]), w[1], w[2], (const long long *)(w + 3));
    }
}
// the occupied slots of the hash table as a packed list (order irrelevant: the host sorts by first arrival)
__global__ __launch_bounds__(kBlockThreads) void k_merge_collect_list(const MergeArgs a) {
    const int lane = threadIdx.x & 63;
    for (uint32_t base = blockIdx.x * kBlockThreads + (threadIdx.x & ~63u); base < a.slots; base += gridDim.x * kBlockThreads) {
        const uint32_t i = base + (uint32_t)lane;
        const bool occ = i < a.slots && a.t_counts[i] != 0ULL;
        const uint64_t m = (uint64_t)__ballot(occ);
        if (!m) continue;
        unsigned long long start = 0;
        if (lane == 0) start = atomicAdd(a.out_n, (unsigned long long)__popcll(m));
        start = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(start >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)start);
        if (occ) {
            const unsigned long long o = start + (unsigned long long)__popcll(m & ((1ULL << lane) - 1ULL));
            if (o < a.out_cap) {
                unsigned long long *w = a.out_list + o * kMergeListWords;
                w[0] = i == a.mask + 1 ? kEmptyKey : a.t_keys[i];
                w[1] = a.t_first[i];
                w[2] = a.t_counts[i];
                for (int j = 0; j < kMaxAggs; ++j) w[3 + j] = (unsigned long long)a.t_vals[(size_t)j * a.slots + i];
            }
        }
    }
}

void launch_merge_init(const MergeArgs &a, hipStream_t s) {
    const int grid = (int)std::min<uint32_t>((a.slots + kBlockThreads - 1) / kBlockThreads, 256u);
    hipLaunchKernelGGL(k_merge_init, dim3(grid < 1 ? 1 : grid), dim3(kBlockThreads), 0, s, a);
}
void launch_merge_scatter(const MergeArgs &a, hipStream_t s) {
    const int grid = (int)std::min<uint32_t>((a.n_groups + kBlockThreads - 1) / kBlockThreads, 256u);
    hipLaunchKernelGGL(k_merge_scatter, dim3(grid < 1 ? 1 : grid), dim3(kBlockThreads), 0, s, a);
}
void launch_merge_insert_list(const MergeArgs &a, hipStream_t s) {
    const int grid = (int)std::min<uint32_t>((a.n_groups + kBlockThreads - 1) / kBlockThreads, 256u);
    hipLaunchKernelGGL(k_merge_insert_list, dim3(grid < 1 ? 1 : grid), dim3(kBlockThreads), 0, s, a);
}
void launch_merge_collect_list(const MergeArgs &a, hipStream_t s) {
    const int grid = (int)std::min<uint32_t>((a.slots + kBlockThreads - 1) / kBlockThreads, 256u);
    hipLaunchKernelGGL(k_merge_collect_list, dim3(grid < 1 ? 1 : grid), dim3(kBlockThreads), 0, s, a);
}

void launch_group_collect(const AggArgs &a, hipStream_t s) {
    const int64_t n = (int64_t)a.mask + 2;
    const int grid = (int)std::min<int64_t>((n + kBlockThreads - 1) / kBlockThreads, 2048);
    hipLaunchKernelGGL(k_group_collect, dim3(grid), dim3(kBlockThreads), 0, s, a);
}

} // namespace imm3
