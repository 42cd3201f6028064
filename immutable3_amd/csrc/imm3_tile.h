// imm3_tile.h -- device-side pieces shared by the tile kernels (imm3_kernels.hip: k_filter_tile, k_emit, k_gather;
// imm3_project.hip: k_filter_project): a tile's columns in registers (ColRegs), the survivor-record layout (Rec), and
// the narrow-value load / store helpers.  gfx950, wave64.
#pragma once

#include "imm3_internal.h"
#include "imm3_device.h"

namespace imm3 {

constexpr int kXposeBytes = 2048; // per wave: one tile of the widest transposed kind (2-byte strings)

// LDS hand-off between the lanes of ONE wave needs no wait at all: a wave's LDS instructions execute in order, so a
// ds_read issued after a ds_write sees it.  Only the compiler must not reorder them.
__device__ __forceinline__ void lds_wave_order() { asm volatile("" ::: "memory"); }

// "every LDS read issued so far has landed", as an instruction the compiler's wait-count insertion SEES (s_waitcnt lgkmcnt(0); vmcnt
// and expcnt left alone): after it the compiler adds no wait of its own in front of the reads' uses.  Without it the sixteen
// ds_read_u8 of a transposed tile are each waited for separately -- sixteen s_waitcnt, one instruction slot each, in kernels
// that are bound by instruction issue (k_filter_project: DESIGN finding 21).  The scheduling barrier keeps the reads' uses behind it.
__device__ __forceinline__ void lds_reads_landed() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
}

// Column bytes are read through pointers TYPED as global (address space 1): a pointer whose provenance the compiler cannot see --
// one read from a tile table -- otherwise makes every access a FLAT instruction (imm3_device.h: as_global says what that costs).
#define IMM3_GLOBAL __attribute__((address_space(1)))

template <int KIND>
struct ColRegs { // TK_NONE: no column
    __device__ __forceinline__ void load(const void *, int64_t, int) {}
    __device__ __forceinline__ void touch() {}
    static constexpr int kLoads = 0;
    __device__ __forceinline__ void load_untracked(const void *, int64_t, int) {}
    __device__ __forceinline__ void keep() const {}
    __device__ __forceinline__ void eval(const TileCol &, uint64_t (&)[kTileWords], int, uint8_t *) {}
    __device__ __forceinline__ void stage(int, uint8_t *) {}
    __device__ __forceinline__ void test(const TileCol &, uint64_t (&)[kTileWords]) {}
    __device__ __forceinline__ bool row(const void *, const TileCol &, int64_t) { return true; }
    __device__ __forceinline__ uint32_t value(int) const { return 0u; }
    __device__ __forceinline__ uint32_t rowval(const void *, int64_t) const { return 0u; }
    __device__ __forceinline__ void load_lane_rows(const void *, int64_t, int) {}
    __device__ __forceinline__ void load_lane_rows_split(const void *, int64_t, int) {}
    __device__ __forceinline__ uint32_t lane_mask(const TileCol &) const { return 0xFFFFu; }
};

template <>
struct ColRegs<TK_I32> {
    int32_t v[kTileWords];
    __device__ __forceinline__ void load(const void *data, int64_t row0, int lane) {
        const IMM3_GLOBAL int32_t *p = (const IMM3_GLOBAL int32_t *)data + row0 + lane;
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
    }
    // The same loads as inline asm: the compiler does not know they are in flight and inserts NO wait of its own for them --
    // the caller does the waiting (k_filter_project: wait_tile<N>, "all but the N youngest vector-memory operations have
    // landed").  With two tiles in flight per wave the compiler's own counts were useless: its wait-count analysis merges the
    // paths that reach the tile body (range ends, the parked bitmap lines' store burst, the partial tile's loop) into "wait for
    // everything", vmcnt(0), which drains the tile that should stay in flight.  touch() then ties the registers to the point
    // behind the caller's wait (an empty asm: nothing is emitted).
    static constexpr int kLoads = kTileWords;
    template <int J = 0>
    __device__ __forceinline__ void load_untracked_from(const int32_t *p) {
        if constexpr (J < kTileWords) {
            asm volatile("global_load_dword %0, %1, off offset:%2 nt" : "=v"(v[J]) : "v"(p), "n"(256 * J) : "memory");
            load_untracked_from<J + 1>(p);
        }
    }
    __device__ __forceinline__ void load_untracked(const void *data, int64_t row0, int lane) { load_untracked_from<0>((const int32_t *)data + row0 + lane); }
    // "these registers are still this column's HERE" (a use the compiler sees, nothing emitted).  An untracked load writes its
    // destination when the data arrives; on a path where the loaded value is never used -- the tiles prefetched past the wave's last
    // one -- the compiler would hand the registers to other values at once, and the landing load would overwrite those (found as
    // a wrong COUNT next to a right bitmap: the per-lane tally had moved into such a register).  The caller waits for the loads
    // (vmcnt(0)) and keeps every set alive up to that wait.
    __device__ __forceinline__ void keep() const {
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]),
                     "v"(v[13]), "v"(v[14]), "v"(v[15]));
    }
    // "the loaded registers are needed HERE": pins the compiler's s_waitcnt for these loads to this point (see k_filter_tile)
    // (ONE statement: with one per register the compiler emits one s_waitcnt per register -- vmcnt(15), vmcnt(14), ... -- sixteen
    // instruction slots where one wait does)
    __device__ __forceinline__ void touch() {
        asm volatile(""
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]),
                       "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    }
    __device__ __forceinline__ void eval(const TileCol &c, uint64_t (&acc)[kTileWords], int, uint8_t *) {
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) acc[j] &= ballot64(in_closed(v[j], c.lo, c.hi));
    }
    // eval() in two steps (k_filter_project): stage() = the LDS traffic of a narrow column's transpose, test() = the compares
    __device__ __forceinline__ void stage(int, uint8_t *) {}
    __device__ __forceinline__ void test(const TileCol &c, uint64_t (&acc)[kTileWords]) { eval(c, acc, 0, nullptr); }
    __device__ __forceinline__ bool row(const void *data, const TileCol &c, int64_t r) { return in_closed(((const IMM3_GLOBAL int32_t *)data)[r], c.lo, c.hi); }
    __device__ __forceinline__ uint32_t value(int j) const { return (uint32_t)v[j]; }
    __device__ __forceinline__ uint32_t rowval(const void *data, int64_t r) const { return ((const IMM3_GLOBAL uint32_t *)data)[r]; }
    __device__ __forceinline__ void load_lane_rows(const void *data, int64_t row0, int lane) { load(data, row0, lane); } // (never in-lane: lane_tile())
    __device__ __forceinline__ void load_lane_rows_split(const void *data, int64_t row0, int lane) { load(data, row0, lane); }
    __device__ __forceinline__ uint32_t lane_mask(const TileCol &) const { return 0xFFFFu; }
};

template <>
struct ColRegs<TK_I8> {
    v4i raw;
    uint32_t v[kTileWords]; // after eval(): the byte of row 64j + lane, zero-extended (what a survivor record carries)
    __device__ __forceinline__ void load(const void *data, int64_t row0, int lane) {
        raw = __builtin_nontemporal_load((const IMM3_GLOBAL v4i *)((const IMM3_GLOBAL int8_t *)data + row0) + lane);
    }
    __device__ __forceinline__ void touch() { asm volatile("" : "+v"(raw)); }
    static constexpr int kLoads = 1;
    __device__ __forceinline__ void load_untracked(const void *data, int64_t row0, int lane) {
        const v4i *p = (const v4i *)((const int8_t *)data + row0) + lane;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(raw) : "v"(p) : "memory");
    }
    __device__ __forceinline__ void keep() const { asm volatile("" ::"v"(raw)); }
    __device__ __forceinline__ void stage(int lane, uint8_t *xp) {
        *(v4i *)(xp + 16 * lane) = raw; // the tile's 1024 bytes in row order
        lds_wave_order();
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) v[j] = ((const uint8_t *)xp)[64 * j + lane];
        lds_wave_order();
    }
    __device__ __forceinline__ void test(const TileCol &c, uint64_t (&acc)[kTileWords]) {
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) acc[j] &= ballot64(in_closed((int32_t)(int8_t)v[j], c.lo, c.hi)); // (sign extension folds into the subtract: SDWA)
    }
    __device__ __forceinline__ void eval(const TileCol &c, uint64_t (&acc)[kTileWords], int lane, uint8_t *xp) {
        stage(lane, xp);
        test(c, acc);
    }
    __device__ __forceinline__ bool row(const void *data, const TileCol &c, int64_t r) { return in_closed((int32_t)((const IMM3_GLOBAL int8_t *)data)[r], c.lo, c.hi); }
    __device__ __forceinline__ uint32_t value(int j) const { return v[j]; }
    __device__ __forceinline__ uint32_t rowval(const void *data, int64_t r) const { return ((const IMM3_GLOBAL uint8_t *)data)[r]; }
    // IN-LANE evaluation (k_filter_tile's narrow-only instances: lane_tile() there says why): `raw` holds rows 16 * lane .. + 15 of
    // the tile; bit k of the result = row 16 * lane + k passes.  No transpose, no ballot: sub (SDWA byte select) + compare + one
    // add-with-carry that shifts the compare's bit in -- three vector instructions per row, nothing else.
    __device__ __forceinline__ void load_lane_rows(const void *data, int64_t row0, int lane) { load(data, row0, lane); }
    // ... beside a 2-byte-string column (whose dense loads give the lane rows 8 lane .. + 7 and 512 + 8 lane .. + 7): the same rows,
    // as two 8-byte loads -- bits 0..7 / 8..15 of lane_mask() are then those two runs
    __device__ __forceinline__ void load_lane_rows_split(const void *data, int64_t row0, int lane) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        const IMM3_GLOBAL v2i *p = (const IMM3_GLOBAL v2i *)((const IMM3_GLOBAL int8_t *)data + row0) + lane;
        const v2i a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 64);
        raw[0] = a[0]; raw[1] = a[1]; raw[2] = b[0]; raw[3] = b[1];
    }
    // Four rows (one dword) per asm block: the compiler's own code for `m = 2 m + in_closed(row)` is select + shift-or per row behind
    // the subtract and the compare (4 instructions and an s_nop: a compare's mask may not be read by the very next vector
    // instructions); v_addc_co_u32 m, m, m, mask does shift and insert in one, and four rows interleaved keep every mask three
    // instructions away from its reader, so no s_nop either.
    static __device__ __forceinline__ void rows4(uint32_t &m, int32_t dword, uint32_t lo, uint32_t range) {
        uint32_t t0, t1, t2, t3;
        uint64_t s0, s1, s2, s3;
        asm("v_sub_u32_sdwa %[t0], sext(%[r]), %[lo] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
            "v_sub_u32_sdwa %[t1], sext(%[r]), %[lo] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\t"
            "v_sub_u32_sdwa %[t2], sext(%[r]), %[lo] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
            "v_sub_u32_sdwa %[t3], sext(%[r]), %[lo] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
            "v_cmp_ge_u32_e64 %[s0], %[rg], %[t0]\n\t"
            "v_cmp_ge_u32_e64 %[s1], %[rg], %[t1]\n\t"
            "v_cmp_ge_u32_e64 %[s2], %[rg], %[t2]\n\t"
            "v_cmp_ge_u32_e64 %[s3], %[rg], %[t3]\n\t"
            "v_addc_co_u32_e64 %[m], %[s0], %[m], %[m], %[s0]\n\t"
            "v_addc_co_u32_e64 %[m], %[s1], %[m], %[m], %[s1]\n\t"
            "v_addc_co_u32_e64 %[m], %[s2], %[m], %[m], %[s2]\n\t"
            "v_addc_co_u32_e64 %[m], %[s3], %[m], %[m], %[s3]"
            : [m] "+v"(m), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
            : [r] "v"(dword), [lo] "s"(lo), [rg] "s"(range));
    }
    __device__ __forceinline__ uint32_t lane_mask(const TileCol &c) const {
        const uint32_t lo = (uint32_t)c.lo, range = (uint32_t)c.hi - (uint32_t)c.lo; // in_closed(): (x - lo) <= (hi - lo), unsigned
        uint32_t m = 0;
        rows4(m, raw[3], lo, range); // (row 15 first: each step shifts the rows so far up)
        rows4(m, raw[2], lo, range);
        rows4(m, raw[1], lo, range);
        rows4(m, raw[0], lo, range);
        return m;
    }
};

template <>
struct ColRegs<TK_S2> {
    v4i raw[2];
    uint32_t v[kTileWords]; // after eval(): the two bytes of row 64j + lane, little-endian
    __device__ __forceinline__ void load(const void *data, int64_t row0, int lane) {
        const IMM3_GLOBAL v4i *p = (const IMM3_GLOBAL v4i *)((const IMM3_GLOBAL uint16_t *)data + row0) + lane;
        raw[0] = __builtin_nontemporal_load(p);
        raw[1] = __builtin_nontemporal_load(p + 64);
    }
    __device__ __forceinline__ void touch() {
        asm volatile("" : "+v"(raw[0]), "+v"(raw[1]));
    }
    static constexpr int kLoads = 2;
    __device__ __forceinline__ void load_untracked(const void *data, int64_t row0, int lane) {
        const v4i *p = (const v4i *)((const uint16_t *)data + row0) + lane;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(raw[0]) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=v"(raw[1]) : "v"(p) : "memory");
    }
    __device__ __forceinline__ void keep() const { asm volatile("" ::"v"(raw[0]), "v"(raw[1])); }
    __device__ __forceinline__ bool hit(const TileCol &c, uint32_t x) {
        bool f = false;
        for (int m = 0; m < c.n_match; ++m) f |= (x == c.match[m]);
        return f;
    }
    __device__ __forceinline__ void eval(const TileCol &c, uint64_t (&acc)[kTileWords], int lane, uint8_t *xp) {
        stage(lane, xp);
        test(c, acc);
    }
    __device__ __forceinline__ void stage(int lane, uint8_t *xp) {
        *(v4i *)(xp + 16 * lane) = raw[0]; // the tile's 2048 bytes in row order
        *(v4i *)(xp + 1024 + 16 * lane) = raw[1];
        lds_wave_order();
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) v[j] = ((const uint16_t *)xp)[64 * j + lane];
        lds_wave_order();
    }
    __device__ __forceinline__ void test(const TileCol &c, uint64_t (&acc)[kTileWords]) {
        if (c.n_match == 1) { // SelectIteratorMatch with the one-value list the SQL front end produces (SQLParser.scala:80-84)
            const uint32_t m0 = c.match[0];
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) acc[j] &= ballot64(v[j] == m0);
        } else { // IN-list outermost (the value sits in one SGPR), eight words at a time: 16 more SGPR pairs would spill
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint64_t h[kTileWords / 2];
#pragma unroll
                for (int j = 0; j < kTileWords / 2; ++j) h[j] = 0ULL;
                for (int m = 0; m < c.n_match; ++m) {
                    const uint32_t mm = c.match[m];
#pragma unroll
                    for (int j = 0; j < kTileWords / 2; ++j) h[j] |= ballot64(v[half * (kTileWords / 2) + j] == mm);
                }
#pragma unroll
                for (int j = 0; j < kTileWords / 2; ++j) acc[half * (kTileWords / 2) + j] &= h[j];
            }
        }
    }
    __device__ __forceinline__ bool row(const void *data, const TileCol &c, int64_t r) { return hit(c, ((const IMM3_GLOBAL uint16_t *)data)[r]); }
    __device__ __forceinline__ uint32_t value(int j) const { return v[j]; }
    __device__ __forceinline__ uint32_t rowval(const void *data, int64_t r) const { return ((const IMM3_GLOBAL uint16_t *)data)[r]; }
    // in-lane evaluation (see ColRegs<TK_I8>): the dense loads of load() leave the lane rows 8 lane .. + 7 (raw[0]) and
    // 512 + 8 lane .. + 7 (raw[1]); bits 0..7 / 8..15 of lane_mask() are those two runs.  (The lane's 16 CONSECUTIVE rows -- 32
    // contiguous bytes, two 16-byte loads at a 32-byte lane stride -- measured slower: each instruction touches every 128-byte line
    // of the tile, S2 over 100 M rows 35.3 -> 41.2 us.)
    __device__ __forceinline__ void load_lane_rows_split(const void *data, int64_t row0, int lane) { load(data, row0, lane); }
    // four rows (two dwords) per asm block, as ColRegs<TK_I8>::rows4: compare (SDWA word select) + add-with-carry per row
    static __device__ __forceinline__ void rows4(uint32_t &m, int32_t d_hi, int32_t d_lo, uint32_t value) {
        uint64_t s0, s1, s2, s3;
        asm("v_cmp_eq_u32_sdwa %[s0], %[a], %[v] src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_cmp_eq_u32_sdwa %[s1], %[a], %[v] src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_cmp_eq_u32_sdwa %[s2], %[b], %[v] src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "v_cmp_eq_u32_sdwa %[s3], %[b], %[v] src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_addc_co_u32_e64 %[m], %[s0], %[m], %[m], %[s0]\n\t"
            "v_addc_co_u32_e64 %[m], %[s1], %[m], %[m], %[s1]\n\t"
            "v_addc_co_u32_e64 %[m], %[s2], %[m], %[m], %[s2]\n\t"
            "v_addc_co_u32_e64 %[m], %[s3], %[m], %[m], %[s3]"
            : [m] "+v"(m), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
            : [a] "v"(d_hi), [b] "v"(d_lo), [v] "s"(value));
    }
    __device__ __forceinline__ uint32_t value_mask(uint32_t value) const {
        uint32_t m = 0;
        rows4(m, raw[1][3], raw[1][2], value); // (row 15 first)
        rows4(m, raw[1][1], raw[1][0], value);
        rows4(m, raw[0][3], raw[0][2], value);
        rows4(m, raw[0][1], raw[0][0], value);
        return m;
    }
    __device__ __forceinline__ uint32_t lane_mask(const TileCol &c) const {
        if (c.n_match == 1) return value_mask(c.match[0]); // (the one-value list the SQL front end produces)
        uint32_t m = 0;
        for (int i = 0; i < c.n_match; ++i) m |= value_mask(c.match[i]); // IN-list: one 16-bit mask per value
        return m;
    }
};

// ---- survivor records ---------------------------------------------------------------------------------------
// A projecting query whose select chain is ONE tile launch compacts, per tile, one RECORD per survivor -- its position
// in the tile and the value of every predicate column, all of which are in registers here -- through a per-wave LDS
// buffer (rank = set bits below the row: s_bcnt1 of the earlier words + v_mbcnt on the row's own word).  The buffer
// holds SEVERAL tiles; when the next tile might not fit it is written out in one piece to the wave's own ARENA -- a
// contiguous region of the staging area -- so the record stores are few, large and sequential per wave (~8-16 KiB at
// a time) instead of one ~800-byte piece per tile at an 8 KiB stride: scattered small writes in between the streaming
// loads ran at ~3.5 GB/ms (C3: 22 us for 78 MB; with the records aimed at an L2-resident region the cost vanished, so
// it is the HBM write pattern, not instruction issue).  Where each tile's records start in the arena goes into a small
// per-wave table (LDS, written once at the end).  k_emit then produces ProjectOp's rows from the records alone: no
// second look at the bitmap, and no second read of a predicate column (at 10 % selectivity a gather would touch nearly
// every 64-byte sector of the column again).  Layout: rec_layout() in imm3_internal.h.
template <int R> struct RecVec;
template <> struct RecVec<1> { typedef uint32_t type; };
template <> struct RecVec<2> { typedef uint2 type; };
template <> struct RecVec<4> { typedef uint4 type; };

template <int K0, int K1, int K2>
struct Rec {
    static constexpr int kinds[3] = {K0, K1, K2};
    static constexpr int R = rec_layout(kinds, -1).dwords;
    typedef typename RecVec<R>::type vec;
    template <int K>
    static __device__ __forceinline__ void put(uint32_t (&rec)[4], uint32_t value) {
        constexpr RecField f = rec_layout(kinds, K);
        if (kinds[K] == TK_NONE) return;
        rec[f.dword] |= value << f.shift;
    }
    static __device__ __forceinline__ vec pack(const uint32_t (&rec)[4]) {
        if constexpr (R == 1) return rec[0];
        else if constexpr (R == 2) return make_uint2(rec[0], rec[1]);
        else return make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
};

// 1-, 2- and 4-byte values are fetched as the aligned dword that contains them: sub-dword global loads are several
// times slower per instruction on this chip (see imm3_agg.hip), and neighbouring survivors share the dword anyway.
// The store truncates.
template <int W>
__device__ __forceinline__ uint32_t load_value(const void *src, int64_t idx) {
    const IMM3_GLOBAL uint32_t *g = (const IMM3_GLOBAL uint32_t *)src; // (every source of these helpers is a column in HBM)
    if constexpr (W == 4) return g[idx];
    else if constexpr (W == 2) return g[idx >> 1] >> (16 * ((uint32_t)idx & 1u));
    else return g[idx >> 2] >> (8 * ((uint32_t)idx & 3u));
}
template <int W>
__device__ __forceinline__ void store_value(void *dst, uint64_t out, uint32_t v) {
    if constexpr (W == 4) ((uint32_t *)dst)[out] = v;
    else if constexpr (W == 2) ((uint16_t *)dst)[out] = (uint16_t)v;
    else ((uint8_t *)dst)[out] = (uint8_t)v;
}
__device__ __forceinline__ uint32_t load_value_rt(const void *src, int width, int64_t idx) {
    return width == 4 ? load_value<4>(src, idx) : (width == 2 ? load_value<2>(src, idx) : load_value<1>(src, idx));
}
__device__ __forceinline__ void store_value_rt(void *dst, int width, uint64_t out, uint32_t v) {
    if (width == 4) store_value<4>(dst, out, v);
    else if (width == 2) store_value<2>(dst, out, v);
    else store_value<1>(dst, out, v);
}

// four consecutive output rows of one column, out0 % 4 == 0: one 16-, 8- or 4-byte store
__device__ __forceinline__ void store_quad(void *dst, int width, unsigned long long out0, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    if (width == 4) *(uint4 *)((uint32_t *)dst + out0) = make_uint4(v0, v1, v2, v3);
    else if (width == 2) *(uint2 *)((uint16_t *)dst + out0) = make_uint2((v0 & 0xFFFFu) | (v1 << 16), (v2 & 0xFFFFu) | (v3 << 16));
    else *(uint32_t *)((uint8_t *)dst + out0) = (v0 & 0xFFu) | ((v1 & 0xFFu) << 8) | ((v2 & 0xFFu) << 16) | (v3 << 24);
}


} // namespace imm3
