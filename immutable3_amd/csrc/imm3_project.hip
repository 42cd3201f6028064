// imm3_project.hip -- k_filter_project: ScanOp -> SelectOp* -> ProjectOp of one uniform segment in ONE pass over the
// columns (gfx950, wave64).
//
// The reference's ProjectIterator.next (engine/src/main/scala/immutabledb/engine/operator/Project.scala:37-64) is one walk:
// for every batch, for every set bit in ascending order, emit the SELECT-list values.  Rounds 1-2 needed three launches
// for it -- the filter kernel staging one record per survivor, an offsets scan, an emit kernel reading the records back
// (C3: 78 MB written and re-read, 145 us).  Here the filter kernel writes the final rows itself:
//
//   * a work-group (one per CU, eight STREAMER waves + four WRITER waves) owns SPANS of 8 * P consecutive tiles, each
//     streamer a RANGE of P consecutive tiles; spans are dealt to the work-groups round-robin.  A streamer evaluates its
//     tiles exactly as k_filter_tile does (all loads of a tile issued before the first compare, narrow columns transposed
//     through LDS, v_cmp result == bitmap word) and compacts one RECORD per survivor -- position, the tile's index in the
//     range, every predicate column -- into its ring in LDS;
//   * when a span's eight ranges are done a writer ANNOUNCES it: the span's survivor count goes into a DESCRIPTOR, the
//     last span of a round to arrive scans the round's counts, and every span reads its first output row from its own
//     descriptor (span_arrive); the writers then unpack the records into the packed output columns -- row index, the
//     predicate columns out of the record, other SELECT-list columns gathered at the record's row -- four output rows
//     per lane (16 / 8 / 4-byte stores);
//   * a range whose records outgrow the ring (dense survivors: a range predicate on a sorted key) keeps no records: the
//     writer takes its rows from the source columns again, at the set bits of the range's bitmap lines (unpack_dense).
//
// Order is deterministic: a row's output slot is the number of survivors before it, whatever the work-groups' timing.
// Every work-group of the launch must be resident (they wait on each other's descriptors): one work-group per CU.  A wait
// that does not resolve within a bounded number of polls, or a device that another launch of this kernel owns, makes the
// launch give up on the ROWS only: a status flag is raised (tagged with the run's epoch), every writer wave leaves, and every
// streamer goes on to the end of its tiles in count + bitmap mode -- no records, no ring, no wait of any kind.  The run's
// COUNT and BITMAP are therefore exact whatever happens (what the RCCL count all-reduce, the count log and every other
// device-side consumer read); the host gathers the rows of such a run from the bitmap (settle_single_pass, imm3_api.cpp).
#include "imm3_internal.h"
#include "imm3_device.h"
#include "imm3_tile.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <type_traits>

namespace imm3 {

#ifndef IMM3_PROJECT_PARK
#define IMM3_PROJECT_PARK 16
#endif
constexpr int kProjParkLines = IMM3_PROJECT_PARK; // bitmap lines a wave parks in LDS between store bursts

// the streamers' ring records (ring_layout, imm3_internal.h)
template <int K0, int K1, int K2>
struct RingRec {
    static constexpr int kinds[3] = {K0, K1, K2};
    static constexpr int R = ring_layout(kinds, -1).dwords;
    typedef typename RecVec<R>::type vec;
    template <int K>
    static __device__ __forceinline__ void put(uint32_t (&rec)[4], uint32_t value) {
        constexpr RecField f = ring_layout(kinds, K);
        if (kinds[K] == TK_NONE) return;
        rec[f.dword] |= value << f.shift;
    }
    static __device__ __forceinline__ vec pack(const uint32_t (&rec)[4]) {
        if constexpr (R == 1) return rec[0];
        else if constexpr (R == 2) return make_uint2(rec[0], rec[1]);
        else return make_uint4(rec[0], rec[1], rec[2], rec[3]);
    }
};

// polls before a look-back wait gives up: ~0.1-0.2 s (the tools' build takes a smaller cap from the fault-injection hook)
__device__ __forceinline__ uint32_t max_polls(const ProjectArgs &a) {
#ifdef IMM3_ABLATE
    if (a.max_polls) return a.max_polls;
#endif
    (void)a;
    return kProjectMaxPolls;
}

__device__ __forceinline__ unsigned long long desc_pack(uint32_t epoch, uint32_t flag, unsigned long long value) {
    return (((unsigned long long)epoch & kDescEpochMask) << kDescEpochShift) | ((unsigned long long)flag << kDescFlagShift) | (value & kDescValueMask);
}
__device__ __forceinline__ uint32_t desc_epoch(unsigned long long d) { return (uint32_t)(d >> kDescEpochShift); }
__device__ __forceinline__ uint32_t desc_flag(unsigned long long d) { return (uint32_t)(d >> kDescFlagShift) & 3u; }

// The status word's flags belong to ONE run: bits 8..31 carry the epoch of the run that raised them (imm3_internal.h).
__device__ __forceinline__ bool status_flag_set(const ProjectArgs &a, uint32_t epoch, unsigned long long flags) {
    const unsigned long long w = __hip_atomic_load(a.finish + kFinishStatus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((w >> kStatusEpochShift) & kStatusEpochMask) == ((unsigned long long)epoch & kStatusEpochMask) && (w & flags) != 0ULL;
}
// one lane.  (A compare-and-swap loop: an earlier run's flags are replaced, this run's are added to.)
__device__ __forceinline__ void status_raise(const ProjectArgs &a, uint32_t epoch, unsigned long long flag) {
    const unsigned long long tag = ((unsigned long long)epoch & kStatusEpochMask) << kStatusEpochShift;
    unsigned long long seen = __hip_atomic_load(a.finish + kFinishStatus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int tries = 0; tries < 1024; ++tries) { // (bounded: at most one writer per work-group and flag ever contends)
        const bool mine = ((seen >> kStatusEpochShift) & kStatusEpochMask) == ((unsigned long long)epoch & kStatusEpochMask);
        const unsigned long long want = mine ? (seen | flag) : (tag | flag);
        if (want == seen) return;
        if (__hip_atomic_compare_exchange_strong(a.finish + kFinishStatus, &seen, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// First output row of span s = survivors of spans 0 .. s-1.  Spans are dealt round-robin to the work-groups, so the spans
// of one ROUND (s / gridDim.x) finish at about the same time.  Every span publishes its survivor count and arrives at the
// round's counter; the LAST arrival of a round scans the round's counts -- one wave, coalesced reads of the descriptors,
// a wave prefix sum -- on top of the previous round's inclusive total, and hands every span of the round its prefix
// through the span's own descriptor; everybody else polls ONE word.  O(spans) uncached traffic per round: the first
// version had every work-group sum all earlier descriptors of its round itself, 65 536 bypassing loads on 16 cache lines
// per round, and the look-back took 30 us per round.  Run by one whole wave.  false: abandoned.
__device__ __forceinline__ unsigned long long desc_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void desc_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool desc_ready(unsigned long long d, uint32_t epoch, uint32_t flag) {
    return desc_epoch(d) == (uint32_t)(epoch & kDescEpochMask) && desc_flag(d) >= flag;
}
__device__ __forceinline__ bool desc_dead(unsigned long long d, uint32_t epoch) { return desc_epoch(d) == (uint32_t)(epoch & kDescEpochMask) && desc_flag(d) == 3u; }

// wave-uniform poll of one word until `flag` (2 = prefix known); false: the rows of this run are given up (abandoned, busy, timed out)
__device__ __forceinline__ bool desc_wait(const ProjectArgs &a, const unsigned long long *p, uint32_t epoch, uint32_t flag, unsigned long long &out) {
    const uint32_t cap = max_polls(a);
    for (uint32_t polls = 0;; ++polls) {
        const unsigned long long d = desc_load(p); // (every lane loads the same word: one request)
        if (desc_dead(d, epoch)) return false;
        if (desc_ready(d, epoch, flag)) { out = d & kDescValueMask; return true; }
        if (polls > cap) return false;
        if ((polls & 63u) == 63u && status_flag_set(a, epoch, kStatusAbandoned | kStatusBusy)) return false; // (somebody of this launch has given up: no need to wait for the time-out)
        __builtin_amdgcn_s_sleep(8);
    }
}

__device__ __forceinline__ bool span_arrive(const ProjectArgs &a, int64_t s, uint32_t epoch, unsigned long long agg, int lane) {
    const int64_t G = gridDim.x;
    const int64_t r = s / G, first = r * G;
    const int64_t n_in = a.n_spans - first < G ? a.n_spans - first : G; // spans of this round
    unsigned long long *round_total = a.round_total; // [n_rounds] inclusive total through the round
    uint32_t *round_ctr = a.round_ctr;               // [n_rounds] arrivals (zero between runs)
    if (lane == 0) desc_store(a.desc + s, desc_pack(epoch, 1u, agg));
    uint32_t prev = 0;
    if (lane == 0) prev = __hip_atomic_fetch_add(round_ctr + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
    if ((int64_t)prev == n_in - 1) { // the round's last arrival: scan it
        unsigned long long base = 0;
        bool ok = true;
        if (r > 0) ok = desc_wait(a, round_total + (r - 1), epoch, 2u, base);
        const uint32_t cap = max_polls(a);
        for (int64_t c0 = 0; ok && c0 < n_in; c0 += 256) { // 256 spans per step: four descriptors per lane, all loads in flight together
            unsigned long long d[4];
            for (uint32_t polls = 0;; ++polls) { // (the counts were stored before their arrivals were counted, but nothing orders the two: check)
                bool all = true, dead = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t i = c0 + 64 * q + lane;
                    d[q] = i < n_in ? desc_load(a.desc + first + i) : desc_pack(epoch, 1u, 0ULL);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dead = dead || desc_dead(d[q], epoch);
                    all = all && desc_ready(d[q], epoch, 1u) && desc_flag(d[q]) == 1u; // (exactly "count known": prefixes are written below, by this wave only)
                }
                if (ballot64(dead)) { ok = false; break; }
                if (ballot64(!all) == 0ULL) break;
                if (polls > cap) { ok = false; break; }
                if ((polls & 63u) == 63u && status_flag_set(a, epoch, kStatusAbandoned | kStatusBusy)) { ok = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) break;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = c0 + 64 * q + lane;
                const unsigned long long v = d[q] & kDescValueMask;
                unsigned long long incl = v;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const unsigned long long up = __shfl_up(incl, dd);
                    if (lane >= dd) incl += up;
                }
                if (i < n_in) desc_store(a.desc + first + i, desc_pack(epoch, 2u, base + incl - v)); // the span's EXCLUSIVE prefix
                base += __shfl(incl, 63);
            }
        }
        if (lane == 0) {
            if (ok) desc_store(round_total + r, desc_pack(epoch, 2u, base));
            else { // abandoned: nobody of this round (or a later one) may wait for the timeout
                for (int64_t i = 0; i < n_in; ++i) desc_store(a.desc + first + i, desc_pack(epoch, 3u, 0ULL));
                desc_store(round_total + r, desc_pack(epoch, 3u, 0ULL));
            }
        }
        if (!ok) return false;
    }
    return true;
}

// This work-group gives up on the run's ROWS (a wait timed out, another work-group said so, or the device is busy with another
// launch): tell the host (`flag`, tagged with the run's epoch), the other waves of this work-group (s_abort: the writers leave, the
// streamers go on in count + bitmap mode) and everybody who waits on this work-group's spans from s on (dead descriptors: a
// waiter sees them at its next poll instead of at its time-out).  One lane.
__device__ __forceinline__ void abandon_run(const ProjectArgs &a, int64_t s, uint32_t epoch, uint32_t *s_abort, unsigned long long flag = kStatusAbandoned) {
    status_raise(a, epoch, flag);
    for (int64_t r = s; r < a.n_spans; r += gridDim.x) desc_store(a.desc + r, desc_pack(epoch, 3u, 0ULL));
    __hip_atomic_store(s_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS words shared between a streamer and the writers: relaxed work-group-scope atomics stay ds_read / ds_write (a volatile
// access would become a flat one and drain the wave's global prefetch: DESIGN finding 19)
// (every lane reads the same word; readfirstlane tells the compiler so: what is decided on it stays scalar control flow)
__device__ __forceinline__ uint32_t lds_peek(const uint32_t *p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_poke(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the acquiring read that pairs with a release store of another wave of this work-group (after a relaxed poll has matched)
__device__ __forceinline__ uint32_t lds_peek_acquire(const uint32_t *p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// what a streamer publishes about a finished range (slot = range parity)
struct RangePub {
    uint32_t start;    // ring position of the range's first record
    uint32_t cnt;      // survivors of the range
    uint32_t dense;    // 1: the range outgrew the ring -- no records; the writer takes the rows from the columns at the bitmap's bits
    uint32_t pad;
};

// Wave specialisation.  A work-group is 12 waves, ONE work-group per CU (grid = number of CUs: every work-group is resident
// whatever the register footprint -- the occupancy query counted two 5-wave work-groups per CU where the hardware placed
// one, and work-groups that wait on each other must all be running).  Waves 0-7 are STREAMERS, two per SIMD: they only
// load, compare and compact -- no global store and no wait on another work-group ever sits in their loop (a wave that
// stores waits, at its next s_waitcnt for loads, for the stores' write acknowledgements too: vmcnt retires in order; and a
// descriptor poll issued behind 20 outstanding streaming loads waits for those first).  Waves 8-11 are WRITERS, one per
// SIMD: whichever is idle announces a finished span (span_arrive), the first one waits for the span's first output row
// and hands it to the others, and each unpacks two streamers' records into the output arrays.  Hand-off through LDS: each
// streamer compacts into a RING of records; a finished range is published (RangePub; the span's last range bumps
// s_span_ready), its writer frees it by advancing the ring's head.  A streamer runs at most kProjSlots ranges ahead of its
// writer and waits only when its ring is full of undrained ranges.
constexpr int kProjStreamers = kProjectStreamers;
constexpr int kProjWriters = kProjectWriters;
constexpr int kProjPerWriter = kProjStreamers / kProjWriters;
constexpr int kProjThreads = 64 * (kProjStreamers + kProjWriters);
constexpr int kProjRingBytes = kProjectRingBytes;
// Tiles of loads a streamer keeps in flight AHEAD of the one it works on: ONE.  Round 4 measured the streamers as latency-bound, not
// issue-bound -- without compares, without records and without writers the kernel streamed C3 in the same 92 us, 1.9 us per tile and
// wave with one tile's loads in flight per wave -- and built depth 2 properly: three register sets rotating (the tile loop unrolled by
// three: no copies), the tile loads as inline asm the compiler does not track and hand-counted s_waitcnt immediates (wait_tile),
// because the compiler's wait-count analysis merges the paths that reach the tile body into vmcnt(0), which drains the tile that
// should stay in flight.  Result: 115 us of streaming instead of 92 (C3 146 / 123 us): more bytes in flight per wave behave like
// more waves (DESIGN findings 3, 8) -- the memory system serves FEWER, burstier streams better.  The machinery stays behind the
// build switch (-DIMM3_PROJECT_DEPTH=2) with what it taught: registers decide where it is possible at all (a set is 16 registers per
// int32 column, 4 / 8 per narrow one; a wave of this 12-wave work-group has 168), and an instance that counts its own vector-memory
// operations must not spill (scratch traffic sits in the same counter; tests/test_host.py checks the build's resource remarks) and
// must keep its register sets alive until every load has landed (ColRegs::keep, imm3_tile.h).
#ifndef IMM3_PROJECT_DEPTH
#define IMM3_PROJECT_DEPTH 1
#endif
constexpr int tile_set_regs(int k) { return k == TK_I32 ? 16 : (k == TK_S2 ? 8 : (k == TK_I8 ? 4 : 0)); }
constexpr int project_depth(int k0, int k1, int k2) { return tile_set_regs(k0) + tile_set_regs(k1) + tile_set_regs(k2) <= 20 ? IMM3_PROJECT_DEPTH : 1; }
constexpr int kProjSlots = 4;   // published ranges a streamer may have waiting for its writer

// ---------------------------------------------------------------------------------------------
// writer side: the records of one range (in the streamer's LDS ring) -> output rows
// base .. base + n.  The record layout is a compile-time function of the column kinds, so the row index and every
// predicate column come out of a record with a shift; what is run-time is only WHERE a column goes (a.pred_dst[k], null =
// not in the SELECT list) and the gathered columns.  A lane owns four consecutive output rows -- a quad aligned in the
// GLOBAL output index -- and stores 16 / 8 / 4 bytes per column; the quads that straddle the range's first or last row are
// written row by row.  (The first version walked run-time column descriptors per row: 2.5 wave instructions per record and
// pass, 60 % of what the streamers execute -- the writers were the bottleneck at ~50 us per span.)
// ---------------------------------------------------------------------------------------------
// (the rows are written once and read by nobody on the device: non-temporal stores, NT = false for A/B runs)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <int W, bool NT = true>
__device__ __forceinline__ void store_quad_w(void *dst, uint32_t out0, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    if constexpr (W == 4) {
        const u32x4_t q = {v0, v1, v2, v3};
        if constexpr (NT) __builtin_nontemporal_store(q, (u32x4_t *)((uint32_t *)dst + out0));
        else *(u32x4_t *)((uint32_t *)dst + out0) = q;
    } else if constexpr (W == 2) {
        const u32x2_t q = {(v0 & 0xFFFFu) | (v1 << 16), (v2 & 0xFFFFu) | (v3 << 16)};
        if constexpr (NT) __builtin_nontemporal_store(q, (u32x2_t *)((uint16_t *)dst + out0));
        else *(u32x2_t *)((uint16_t *)dst + out0) = q;
    } else {
        const uint32_t q = (v0 & 0xFFu) | ((v1 & 0xFFu) << 8) | ((v2 & 0xFFu) << 16) | (v3 << 24);
        if constexpr (NT) __builtin_nontemporal_store(q, (uint32_t *)((uint8_t *)dst + out0));
        else *(uint32_t *)((uint8_t *)dst + out0) = q;
    }
}
constexpr int kind_width(int k) { return k == TK_I32 ? 4 : (k == TK_S2 ? 2 : 1); }

// One tile descriptor of a table query, read through a CONSTANT-address-space pointer: the descriptors were written before the
// launch and nothing writes them during it, and only so may the compiler use a scalar load (s_load, lgkmcnt) for a wave-uniform
// index.  As plain global memory it must assume that this kernel's own stores could alias them and emits vector loads -- which sit
// in vmcnt BEHIND the tile loads just issued, so the wait for the descriptor drained the streamers' prefetch (C3 over a table:
// 160 us against 120).
__device__ __forceinline__ ProjectTile load_tile_desc(const ProjectTile *base, int64_t idx) {
    typedef const __attribute__((address_space(4))) unsigned long long *cptr;
    const cptr w = (cptr)(base + idx);
    ProjectTile d;
    d.p[0] = (const void *)w[0];
    d.p[1] = (const void *)w[1];
    d.p[2] = (const void *)w[2];
    const unsigned long long r = w[3];
    d.rows = (uint32_t)r;
    d.pad = (uint32_t)(r >> 32);
    return d;
}

template <int R>
__device__ __forceinline__ void rec_words(const typename RecVec<R>::type &r, uint32_t (&w)[4]) {
    if constexpr (R == 1) { w[0] = r; w[1] = w[2] = w[3] = 0u; }
    else if constexpr (R == 2) { w[0] = r.x; w[1] = r.y; w[2] = w[3] = 0u; }
    else { w[0] = r.x; w[1] = r.y; w[2] = r.z; w[3] = r.w; }
}

// predicate column K (compile time) of a quad of records -> dst
template <int K0, int K1, int K2, int K>
__device__ __forceinline__ void store_pred_col(void *dst, const uint32_t (&rw)[4][4], const bool (&ok)[4], bool whole, uint32_t out0, bool plain = false) {
    constexpr int kinds[3] = {K0, K1, K2};
    if constexpr (kinds[K] != TK_NONE) {
        if (!dst) return; // wave-uniform: the column is not in the SELECT list
        constexpr RecField f = ring_layout(kinds, K);
        constexpr int W = kind_width(kinds[K]);
        uint32_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rw[e][f.dword] >> f.shift;
        if (whole) {
            if (plain) store_quad_w<W, false>(dst, out0, v[0], v[1], v[2], v[3]);
            else store_quad_w<W, true>(dst, out0, v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ok[e]) store_value<W>(dst, out0 + e, v[e]);
        }
    }
}

// The gathered columns of one range (NG of them, compile time: with a run-time count every load sat behind a branch of its
// own and was waited for before the next one was issued -- sixteen serialised round trips per step, 25 us per span on C4).
// Two quads per lane and step; every load of the step -- all columns, all eight rows -- is in flight before the first store.
template <int K0, int K1, int K2, int NG>
__device__ __forceinline__ void gather_range(const ProjectArgs &a, const typename RingRec<K0, K1, K2>::vec *src, uint32_t start, uint32_t cap, uint32_t n,
                                             uint32_t base, uint32_t tile0, int lane) {
    typedef RingRec<K0, K1, K2> L;
    constexpr int R = L::R;
    const uint32_t q0 = base >> 2;
    const uint32_t n_quads = ((base + n + 3) >> 2) - q0;
    const void *gsrc[NG];
    void *gdst[NG];
    int gw[NG];
#pragma unroll
    for (int c = 0; c < NG; ++c) {
        gsrc[c] = a.gather[c].src;
        gdst[c] = a.gather[c].dst;
        gw[c] = a.gather[c].width;
    }
    for (uint32_t qi = lane; qi < n_quads; qi += 128) {
        uint32_t raw[2][4][NG], sh[2][4][NG];
        bool ok[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t out0 = (q0 + qi + 64 * u) << 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t o = out0 + e;
                ok[u][e] = qi + 64 * u < n_quads && o >= base && o - base < n;
                uint32_t idx = start + (ok[u][e] ? o - base : 0u);
                if (idx >= cap) idx -= cap;
                uint32_t rw[4];
                rec_words<R>(src[idx], rw);
                const uint32_t row = tile0 * (uint32_t)kTileRows + (rw[0] >> 16); // (the record's position in its range rides on top of dword 0)
#pragma unroll
                for (int c = 0; c < NG; ++c) { // the aligned dword that holds the value
                    const uint32_t byte = row * (uint32_t)gw[c]; // (< 2^32: a segment's .dat is < 2 GiB, Segment.scala:33)
                    raw[u][e][c] = ((const uint32_t *)gsrc[c])[byte >> 2];
                    sh[u][e][c] = 8 * (byte & 3u);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t out0 = (q0 + qi + 64 * u) << 2;
            const bool whole = ok[u][0] && ok[u][3];
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                uint32_t v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = raw[u][e][c] >> sh[u][e][c];
                if (whole) store_quad(gdst[c], gw[c], out0, v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ok[u][e]) store_value_rt(gdst[c], gw[c], out0 + e, v[e]);
                }
            }
        }
    }
}

// src: the streamer's ring (start, cap = the range's position in it and its capacity)
template <int K0, int K1, int K2>
__device__ __forceinline__ void unpack_range(const ProjectArgs &a, const typename RingRec<K0, K1, K2>::vec *src, uint32_t start, uint32_t cap, uint32_t n,
                                             uint32_t base, uint32_t tile0, int lane) {
    typedef RingRec<K0, K1, K2> L;
    constexpr int R = L::R;
    const uint32_t cap_rows = a.cap_rows > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)a.cap_rows;
    if (base >= cap_rows) return; // wave-uniform: the output arrays are full (the host gathers again from the bitmap)
    if (n > cap_rows - base) n = cap_rows - base;
    if (!n) return;
    const uint32_t q0 = base >> 2;
    const uint32_t n_quads = ((base + n + 3) >> 2) - q0;
    uint32_t *row_index = a.row_index;
    void *d0 = a.pred_dst[0], *d1 = a.pred_dst[1], *d2 = a.pred_dst[2];
    for (uint32_t qi = lane; qi < n_quads; qi += 64) {
        const uint32_t out0 = (q0 + qi) << 2;
        uint32_t rw[4][4];
        bool ok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { // (a row outside the range re-reads record 0: no branch)
            const uint32_t o = out0 + e;
            ok[e] = o >= base && o - base < n;
            uint32_t idx = start + (ok[e] ? o - base : 0u);
            if (idx >= cap) idx -= cap;
            rec_words<R>(src[idx], rw[e]);
        }
        if (IMM3_ABLATE_BIT(a, 16)) { // (no stores)
            asm volatile("" ::"v"(rw[0][0]), "v"(rw[1][0]), "v"(rw[2][0]), "v"(rw[3][0]));
            continue;
        }
        const bool whole = ok[0] && ok[3]; // (the range's rows are contiguous: first and last in => all four in)
        uint32_t rowv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) rowv[e] = tile0 * (uint32_t)kTileRows + (rw[e][0] >> 16); // (the record's position in its range rides on top of dword 0)
        const bool plain = IMM3_ABLATE_BIT(a, 32); // (A/B: plain instead of non-temporal stores)
        if (whole) {
            if (plain) store_quad_w<4, false>(row_index, out0, rowv[0], rowv[1], rowv[2], rowv[3]);
            else store_quad_w<4, true>(row_index, out0, rowv[0], rowv[1], rowv[2], rowv[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ok[e]) row_index[out0 + e] = rowv[e];
        }
        store_pred_col<K0, K1, K2, 0>(d0, rw, ok, whole, out0, plain);
        store_pred_col<K0, K1, K2, 1>(d1, rw, ok, whole, out0, plain);
        store_pred_col<K0, K1, K2, 2>(d2, rw, ok, whole, out0, plain);
    }
    // ---- SELECT-list columns that are not predicate columns: gathered at the records' rows
    switch (a.n_gather) { // wave-uniform
    case 0: break;
    case 1: gather_range<K0, K1, K2, 1>(a, src, start, cap, n, base, tile0, lane); break;
    case 2: gather_range<K0, K1, K2, 2>(a, src, start, cap, n, base, tile0, lane); break;
    case 3: gather_range<K0, K1, K2, 3>(a, src, start, cap, n, base, tile0, lane); break;
    default: gather_range<K0, K1, K2, 4>(a, src, start, cap, n, base, tile0, lane); break;
    }
}

// A DENSE range: its survivors did not fit the streamer's ring (a run of rows that mostly survive), or it holds the segment's
// partial last tile.  The streamer kept the count and the bitmap lines only; the writer takes the rows from the source
// columns again, at the set bits of the range's bitmap lines: rank = set bits below the row, as the streamer's compaction
// computes it.  Per tile and column: sixteen predicated loads in flight, then sixteen predicated stores; a fully surviving
// tile is a straight copy.  The columns were read a moment ago by this CU (L2 / the memory-side cache usually still hold
// them); nothing goes through HBM twice the way spilled records did (the first version moved an outgrown range to an
// arena in HBM: written, read back, written again -- 50 M contiguous survivors took 319 us).
constexpr int kDenseWords = kTileWords / 2; // a UNIT of the dense path: half a tile, eight bitmap words

// What a writer keeps of a unit between issuing its loads and storing its rows: the loaded values and where the unit's
// rows start.  The rest (which rows, which output slots) is a few scalar and mbcnt instructions away from the bitmap
// line and is computed again at store time rather than held in registers: registers are what bounds the loads in flight.
struct DenseSet {
    uint32_t v[3][kDenseWords]; // the predicate columns' values at the lane's rows
    unsigned long long ubase;   // output slot of the unit's first survivor
};

// Four units are in flight per writer (two when more than one predicate column is projected: registers): a unit's loads
// are issued three units before its rows are stored (the register sets take turns, as the streamers' tile buffers do);
// the bitmap lines come four tiles -- a CHUNK: 64 words, one per lane -- at a time, the next chunk's fetched while this
// one's are worked on.  The loop body is STRAIGHT-LINE code: every lane loads (a row that does not survive loads the
// unit's first row) and every lane stores (a row that is not wanted stores to the wave's trash line), because with a
// branch per predicated access the compiler waits for ALL outstanding loads (s_waitcnt vmcnt(0)) before every store --
// one unit in flight whatever the source says.  (Tile by tile, with the tile's bitmap line fetched first and its stores
// behind its loads, a dense tile cost a writer three dependent round trips: 5 us; with two units and vmcnt(0): 3.5 us.)
// TABLE (table queries): a tile's columns start where its descriptor says (a range may straddle two segments), rows are virtual
// (tile * 1024 + position), and every tile's bitmap line exists in full (bits past a segment's last row are zero).
template <int K0, int K1, int K2, bool TABLE>
__device__ __forceinline__ void unpack_dense(const ProjectArgs &a, int64_t tile0, int P, unsigned long long base, int lane, void *trash) {
    const int64_t n_words = TABLE ? a.n_tiles * kTileWords : (a.n_rows + 63) / 64;
    int n_t = P;
    if ((int64_t)n_t > a.n_tiles - tile0) n_t = (int)(a.n_tiles - tile0);
    const int n_units = 2 * n_t;
    constexpr int NS = ((K0 != TK_NONE) + (K1 != TK_NONE) + (K2 != TK_NONE)) <= 1 ? 4 : 2; // units in flight
    const unsigned long long cap_rows = IMM3_ABLATE_BIT(a, 16) ? 0ULL : a.cap_rows; // (ablation: no stores)
    // a predicate column that is not in the SELECT list is loaded and stored to the trash line all the same (no branch)
    void *d0 = a.pred_dst[0] ? a.pred_dst[0] : trash, *d1 = a.pred_dst[1] ? a.pred_dst[1] : trash, *d2 = a.pred_dst[2] ? a.pred_dst[2] : trash;
    const bool t0 = !a.pred_dst[0], t1 = !a.pred_dst[1], t2 = !a.pred_dst[2];
    auto fetch_lines = [&](int c) __attribute__((always_inline)) -> unsigned long long { // lane l: word l of the four tiles of chunk c (stored by a streamer of this work-group: read past the L1)
        const int64_t word = (tile0 + 4 * c) * kTileWords + lane;
        const bool in = 4 * c < n_t && word < n_words && word < (tile0 + n_t) * kTileWords;
        const unsigned long long line = __hip_atomic_load(a.bitmap + (in ? word : tile0 * kTileWords), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return in ? line : 0ULL;
    };
    unsigned long long line_prev = 0ULL, line_cur = fetch_lines(0), line_nxt = fetch_lines(1); // the bitmap lines of the previous, this and the next chunk
    int cur_chunk = 0;
    // the bitmap words of unit t
    auto unit_words = [&](int t, uint64_t (&m)[kDenseWords]) __attribute__((always_inline)) {
        const int j = t >> 1;
        const unsigned long long line = (j >> 2) == cur_chunk ? line_cur : line_prev; // wave-uniform choice (a unit is stored up to NS units after it was issued)
        const uint32_t lo = (uint32_t)line, hi = (uint32_t)(line >> 32);
        const int f0 = (j & 3) * kTileWords + (t & 1) * kDenseWords;
        const bool live = t >= 0 && t < n_units; // (the pipeline's first and last rounds run on empty units)
#pragma unroll
        for (int w = 0; w < kDenseWords; ++w) {
            m[w] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, f0 + w) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, f0 + w);
            if (!live) m[w] = 0ULL;
        }
    };
    // the lane's row of word w is stored: set in the bitmap, and its output slot o below the capacity (rows past it: the host
    // gathers again from the bitmap)
    auto stored = [&](uint64_t m, unsigned long long done, uint32_t &o) __attribute__((always_inline)) -> bool {
        const unsigned long long o64 = done + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        o = (uint32_t)o64;
        return __builtin_amdgcn_inverse_ballot_w64(m) && o64 < cap_rows;
    };
    auto row_of = [&](int t) __attribute__((always_inline)) -> uint32_t { return (uint32_t)((tile0 + (t >> 1)) * kTileRows) + (uint32_t)((t & 1) * kDenseWords * 64) + (uint32_t)lane; };
    auto issue = [&](DenseSet &S, int t) __attribute__((always_inline)) {
        uint64_t m[kDenseWords];
        unit_words(t, m);
        S.ubase = base;
        const int tt = t < n_units ? t : 0;             // (an empty unit loads tile 0's rows: valid addresses)
        // where the unit's rows are read: rows of the segment's flat columns, or (table) rows of the unit's TILE behind its descriptor's pointers
        const void *s0 = a.cols[0].data, *s1 = a.cols[1].data, *s2 = a.cols[2].data;
        uint32_t row0 = row_of(tt), safe = row_of(tt & ~1) - (uint32_t)lane; // safe: the tile's first row
        if constexpr (TABLE) {
            const ProjectTile d = load_tile_desc(a.tile_desc, __builtin_amdgcn_readfirstlane((int)(tile0 + (tt >> 1)))); // (wave-uniform, read-only: scalar loads)
            s0 = as_global(d.p[0]);
            s1 = as_global(d.p[1]);
            s2 = as_global(d.p[2]);
            row0 = (uint32_t)((tt & 1) * kDenseWords * 64) + (uint32_t)lane;
            safe = 0u;
        }
        auto load_col = [&](auto kind, const void *src, uint32_t (&v)[kDenseWords]) __attribute__((always_inline)) {
            constexpr int K = decltype(kind)::value;
            if constexpr (K != TK_NONE) {
                unsigned long long done = S.ubase;
#pragma unroll
                for (int w = 0; w < kDenseWords; ++w) {
                    uint32_t o;
                    const bool p = stored(m[w], done, o);
                    v[w] = load_value<kind_width(K)>(src, (int64_t)(p ? row0 + 64u * (uint32_t)w : safe));
                    done += (uint32_t)__popcll(m[w]);
                }
            }
        };
        load_col(std::integral_constant<int, K0>(), s0, S.v[0]);
        load_col(std::integral_constant<int, K1>(), s1, S.v[1]);
        load_col(std::integral_constant<int, K2>(), s2, S.v[2]);
#pragma unroll
        for (int w = 0; w < kDenseWords; ++w) base += (uint32_t)__popcll(m[w]);
    };
    auto flush = [&](const DenseSet &S, int t) __attribute__((always_inline)) {
        uint64_t m[kDenseWords];
        unit_words(t, m);
        const uint32_t row0 = row_of(t);
        {
            unsigned long long done = S.ubase;
#pragma unroll
            for (int w = 0; w < kDenseWords; ++w) {
                uint32_t o;
                const bool p = stored(m[w], done, o);
                uint32_t *dst = p ? a.row_index + o : (uint32_t *)trash;
                *dst = row0 + 64u * (uint32_t)w;
                done += (uint32_t)__popcll(m[w]);
            }
        }
        auto store_col = [&](auto kind, void *dst, bool to_trash, const uint32_t (&v)[kDenseWords]) __attribute__((always_inline)) {
            constexpr int K = decltype(kind)::value;
            if constexpr (K != TK_NONE) {
                unsigned long long done = S.ubase;
#pragma unroll
                for (int w = 0; w < kDenseWords; ++w) {
                    uint32_t o;
                    const bool p = stored(m[w], done, o) && !to_trash;
                    store_value<kind_width(K)>(p ? dst : trash, p ? o : 0u, v[w]);
                    done += (uint32_t)__popcll(m[w]);
                }
            }
        };
        store_col(std::integral_constant<int, K0>(), d0, t0, S.v[0]);
        store_col(std::integral_constant<int, K1>(), d1, t1, S.v[1]);
        store_col(std::integral_constant<int, K2>(), d2, t2, S.v[2]);
    };
    // A range whose rows ALL survive -- a run of the sorted key, the usual shape of a dense range -- is a straight copy: the rows'
    // output slots are base, base + 1, ... in row order, so the row numbers are an iota and every projected predicate column is
    // n_t KiB-sized pieces moved with 16-byte loads and stores (dword-aligned: the slot of the range's first row is arbitrary;
    // a narrow column whose first output byte is not dword-aligned takes the general walk).  ~30 instructions per tile instead of
    // ~540: the general walk computes every row's slot from the bitmap line, per column.
    bool copied = false;
    {
        bool ok = (TABLE || (tile0 + n_t) * (int64_t)kTileRows <= a.n_rows) && base + (unsigned long long)n_t * kTileRows <= cap_rows; // (table: a partial tile's line has zero bits, found below)
        if (K0 != TK_NONE && !t0) ok = ok && ((base * kind_width(K0)) & 3ULL) == 0ULL;
        if (K1 != TK_NONE && !t1) ok = ok && ((base * kind_width(K1)) & 3ULL) == 0ULL;
        if (K2 != TK_NONE && !t2) ok = ok && ((base * kind_width(K2)) & 3ULL) == 0ULL;
        if (ok) { // wave-uniform
            bool hole = false;
            for (int c = 0; 4 * c < n_t; ++c) {
                const bool in = (4 * c) * kTileWords + lane < n_t * kTileWords;
                const unsigned long long line = __hip_atomic_load(a.bitmap + (tile0 + 4 * c) * kTileWords + (in ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hole |= in && line != ~0ULL;
            }
            copied = !ballot64(hole);
        }
    }
    if (copied) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t first_row = (uint32_t)(tile0 * kTileRows);
        // 16-byte pieces per lane and tile: 1024 rows x W bytes / (64 lanes x 16 bytes) = W
        constexpr int W0 = K0 == TK_NONE ? 0 : kind_width(K0), W1 = K1 == TK_NONE ? 0 : kind_width(K1), W2 = K2 == TK_NONE ? 0 : kind_width(K2);
        struct CopySet {
            u32x4 v0[W0 ? W0 : 1], v1[W1 ? W1 : 1], v2[W2 ? W2 : 1];
        };
        auto load_col = [&](auto width, const void *src, bool to_trash, int j, u32x4 *v) __attribute__((always_inline)) {
            constexpr int W = decltype(width)::value;
            if constexpr (W != 0) {
                if (to_trash) return; // (wave-uniform: the column is not in the SELECT list)
                typedef IMM3_GLOBAL u32x4 gvec; // (a global pointer by type: no flat loads through a tile descriptor's pointer)
                const gvec *from = TABLE ? (const gvec *)src : (const gvec *)((const IMM3_GLOBAL uint8_t *)src + ((int64_t)first_row + (int64_t)j * kTileRows) * W); // (table: src is the tile's own pointer)
#pragma unroll
                for (int i = 0; i < W; ++i) v[i] = __builtin_nontemporal_load(from + 64 * i + lane);
            }
        };
        auto store_col = [&](auto width, void *dst, bool to_trash, int j, const u32x4 *v) __attribute__((always_inline)) {
            constexpr int W = decltype(width)::value;
            if constexpr (W != 0) {
                if (to_trash) return;
                u32x4 *to = (u32x4 *)((uint8_t *)dst + (base + (unsigned long long)j * kTileRows) * W);
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    if (IMM3_ABLATE_BIT(a, 32)) to[64 * i + lane] = v[i]; // (A/B: plain stores)
                    else __builtin_nontemporal_store(v[i], to + 64 * i + lane);
                }
            }
        };
        auto load_tile = [&](CopySet &S, int j) __attribute__((always_inline)) { // (a tile past the range: the range's last one again -- no branch around loads)
            const int jj = j < n_t ? j : n_t - 1;
            const void *s0 = a.cols[0].data, *s1 = a.cols[1].data, *s2 = a.cols[2].data;
            if constexpr (TABLE) {
                const ProjectTile d = load_tile_desc(a.tile_desc, __builtin_amdgcn_readfirstlane((int)(tile0 + jj)));
                s0 = as_global(d.p[0]);
                s1 = as_global(d.p[1]);
                s2 = as_global(d.p[2]);
            }
            load_col(std::integral_constant<int, W0>(), s0, t0, jj, S.v0);
            load_col(std::integral_constant<int, W1>(), s1, t1, jj, S.v1);
            load_col(std::integral_constant<int, W2>(), s2, t2, jj, S.v2);
        };
        auto store_tile = [&](const CopySet &S, int j) __attribute__((always_inline)) {
            u32x4 *rows = (u32x4 *)(a.row_index + base + (unsigned long long)j * kTileRows);
            const uint32_t r = first_row + (uint32_t)(j * kTileRows) + 4u * (uint32_t)lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t r0 = r + 256u * (uint32_t)i;
                u32x4 q = {r0, r0 + 1u, r0 + 2u, r0 + 3u};
                if (IMM3_ABLATE_BIT(a, 32)) rows[64 * i + lane] = q;
                else __builtin_nontemporal_store(q, rows + 64 * i + lane);
            }
            store_col(std::integral_constant<int, W0>(), d0, t0, j, S.v0);
            store_col(std::integral_constant<int, W1>(), d1, t1, j, S.v1);
            store_col(std::integral_constant<int, W2>(), d2, t2, j, S.v2);
        };
        // Two tiles in flight: a tile's loads are issued before the stores of the tile before it.  (Four deep measured 10 us slower on
        // id > 5e7, one deep the same as two: the writers are not waiting for their loads.  After the streamers are through --
        // 60-75 us for 400 MB -- the four writers of every CU copy the run's 200 MB into 400 MB of rows alone, at ~4 TB/s read + written;
        // torch's copy kernel at full occupancy: 5.3.)  Non-temporal stores: 224 -> 212 us.
        constexpr int kCopyDepth = 2;
        CopySet S[kCopyDepth];
#pragma unroll
        for (int d = 0; d < kCopyDepth - 1; ++d) load_tile(S[d], d);
#pragma unroll 1
        for (int j = 0; j < n_t; j += kCopyDepth) {
#pragma unroll
            for (int d = 0; d < kCopyDepth; ++d) {
                load_tile(S[(d + kCopyDepth - 1) % kCopyDepth], j + d + kCopyDepth - 1);
                if (j + d < n_t) store_tile(S[d], j + d); // wave-uniform
            }
        }
        base += (unsigned long long)n_t * kTileRows;
    }
    DenseSet S[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) S[i].ubase = 0ULL;
#pragma unroll 1
    for (int t = 0; t < (copied ? 0 : n_units + NS); t += NS) { // (the first round only loads, the last one only stores)
        if ((t & 7) == 0 && t > 0) { // a new chunk (eight units): the units in flight are of the previous one
            line_prev = line_cur;
            line_cur = line_nxt;
            cur_chunk = t >> 3;
            line_nxt = fetch_lines(cur_chunk + 1);
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            flush(S[i], t - NS + i);
            issue(S[i], t + i);
        }
    }
    if (TABLE || a.n_gather == 0) return; // (table queries are planned as one launch only without gathered columns)
    // ---- SELECT-list columns that are not predicate columns (only when the tuning forces this kernel on such a query): a
    // second walk over the range, unit by unit
    unsigned long long gbase = base;
#pragma unroll 1
    for (int t = n_units - 1; t >= 0; --t) { // backwards from the range's end: base is now the slot behind its last row
        const int64_t word0 = (tile0 + (t >> 1)) * kTileWords + (t & 1) * kDenseWords;
        unsigned long long line = 0ULL;
        if (lane < kDenseWords && word0 + lane < n_words) line = __hip_atomic_load(a.bitmap + word0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t m[kDenseWords];
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kDenseWords; ++w) {
            m[w] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(line >> 32), w) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)line, w);
            total += (uint32_t)__popcll(m[w]);
        }
        gbase -= total;
        if (total == 0) continue;
        const uint32_t row0 = row_of(t);
        for (int c = 0; c < a.n_gather; ++c) {
            const void *src = a.gather[c].src;
            void *dst = a.gather[c].dst;
            const int width = a.gather[c].width;
            unsigned long long done = gbase;
#pragma unroll
            for (int w = 0; w < kDenseWords; ++w) {
                uint32_t o;
                if (stored(m[w], done, o)) store_value_rt(dst, width, o, load_value_rt(src, width, (int64_t)(row0 + 64u * (uint32_t)w)));
                done += (uint32_t)__popcll(m[w]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// streamer side: one record per survivor of a tile, compacted in ascending row order.  rank = set bits below the row in
// its word (v_mbcnt) + the survivors of the earlier words (scalar); the store is predicated by the word itself.
// ---------------------------------------------------------------------------------------------
// the record of the lane's row of word w.  posw = (64 w + lane) << 16, kept in a register per word for the whole kernel; j26 = the
// tile's index in its range << 26 (scalar).  With a narrow column at bit 0 the first dword is one v_or3 of the value as the
// transposing LDS read delivered it, posw and j26.
template <int K0, int K1, int K2>
__device__ __forceinline__ void build_record(uint32_t (&rec)[4], const ColRegs<K0> &c0, const ColRegs<K1> &c1, const ColRegs<K2> &c2, uint32_t posw, uint32_t j26, int w) {
    typedef RingRec<K0, K1, K2> L;
    rec[0] = posw | j26;
    rec[1] = rec[2] = rec[3] = 0u;
    L::template put<0>(rec, c0.value(w));
    L::template put<1>(rec, c1.value(w));
    L::template put<2>(rec, c2.value(w));
}

template <int K0, int K1, int K2, class Store>
__device__ __forceinline__ void compact_tile(const uint64_t (&acc)[kTileWords], const ColRegs<K0> &c0, const ColRegs<K1> &c1, const ColRegs<K2> &c2,
                                             const uint32_t (&posw)[kTileWords], uint32_t j26, Store store) {
    typedef RingRec<K0, K1, K2> L;
    uint32_t done = 0; // wave-uniform
#pragma unroll
    for (int w = 0; w < kTileWords; ++w) {
        const uint64_t m = acc[w];
        uint32_t rec[4];
        build_record<K0, K1, K2>(rec, c0, c1, c2, posw[w], j26, w);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, done));
        if (__builtin_amdgcn_inverse_ballot_w64(m)) store(rank, L::pack(rec)); // exec = the word itself
        done += (uint32_t)__popcll(m);
    }
}

// The same with the bookkeeping moved off the scalar unit (round 4).  The kernel is bound by instruction issue, scalar
// instructions included (DESIGN finding 21), and the form above spends, per word, s_bcnt1 + s_add on the running count, a v_mov to
// bring that count into the rank, and -- before it starts -- sixteen more s_bcnt1 + fifteen s_add for the tile's total, which the
// ring-space check needs first.  Here lanes 0..15 hold the tile's sixteen words (`mine`, made for the bitmap line anyway): their
// popcounts, a row-wide DPP prefix sum and one v_readlane give the tile's total (five vector instructions instead of thirty-one
// scalar ones), and the same scan gives every word the LDS ADDRESS of its first record (`word_at`, lane w), fetched per word with
// one v_readlane: rank = v_mbcnt on the word, address = word_at + rank * record size, store predicated by the word itself.
// "all but the wave's N youngest vector-memory operations have landed" (loads and stores count together, in issue order: vmcnt
// has six bits on gfx9, split 4 + 2 in the immediate).  The streamers' tile loads are inline asm the compiler does not track
// (imm3_tile.h: load_untracked), so this is THE wait for them; operations of the compiler's own that happen to be younger (the
// burst of parked bitmap lines) only make it wait for a little more than needed.
template <int N>
__device__ __forceinline__ void wait_tile() {
    static_assert(N >= 0 && N < 63, "vmcnt holds 0..63");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

template <int N>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t x) { // lane l of every 16-lane row: x of lane l - N of that row, 0 where there is none
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t row_inclusive_sum(uint32_t x) {
    x += dpp_row_shr<1>(x);
    x += dpp_row_shr<2>(x);
    x += dpp_row_shr<4>(x);
    x += dpp_row_shr<8>(x);
    return x;
}

// Four words' records stored with the words themselves as exec masks.  Written as one asm block because what costs here is
// instruction slots: the compiler's form of `if (lane survives) store` is s_and_saveexec + s_cbranch_execz + ... + s_or exec per
// word, with the rank / address / record arithmetic sunk into the masked region; here that arithmetic runs for all lanes in front
// of the block (it is needed by the masked lanes only, but a vector instruction costs one slot whatever its exec), and each word is
// one s_mov to exec and one LDS write; exec goes back to all ones once per four words.  (The streamers run with all 64 lanes
// active: wave-uniform control flow only.)  ds_write2_b32 takes the record's dwords from any two registers: no register pairs.
template <int R>
__device__ __forceinline__ void masked_store4(const uint64_t (&m)[4], const uint32_t (&to)[4], const uint32_t (&d)[4][4]) {
    if constexpr (R == 1) {
        asm volatile("s_mov_b64 exec, %0\n\tds_write_b32 %4, %8\n\t"
                     "s_mov_b64 exec, %1\n\tds_write_b32 %5, %9\n\t"
                     "s_mov_b64 exec, %2\n\tds_write_b32 %6, %10\n\t"
                     "s_mov_b64 exec, %3\n\tds_write_b32 %7, %11\n\t"
                     "s_mov_b64 exec, -1"
                     :
                     : "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "v"(to[0]), "v"(to[1]), "v"(to[2]), "v"(to[3]), "v"(d[0][0]), "v"(d[1][0]), "v"(d[2][0]), "v"(d[3][0])
                     : "memory");
    } else if constexpr (R == 2) {
        asm volatile("s_mov_b64 exec, %0\n\tds_write2_b32 %4, %8, %12 offset1:1\n\t"
                     "s_mov_b64 exec, %1\n\tds_write2_b32 %5, %9, %13 offset1:1\n\t"
                     "s_mov_b64 exec, %2\n\tds_write2_b32 %6, %10, %14 offset1:1\n\t"
                     "s_mov_b64 exec, %3\n\tds_write2_b32 %7, %11, %15 offset1:1\n\t"
                     "s_mov_b64 exec, -1"
                     :
                     : "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "v"(to[0]), "v"(to[1]), "v"(to[2]), "v"(to[3]), "v"(d[0][0]), "v"(d[1][0]), "v"(d[2][0]), "v"(d[3][0]),
                       "v"(d[0][1]), "v"(d[1][1]), "v"(d[2][1]), "v"(d[3][1])
                     : "memory");
    }
}
// four-dword records (two or three int32 predicate columns): one word at a time -- four words' worth of records would take twenty
// more registers than these instances have to spare
__device__ __forceinline__ void masked_store1_r4(uint64_t m, uint32_t to, const uint32_t (&d)[4]) {
    asm volatile("s_mov_b64 exec, %0\n\tds_write2_b32 %1, %2, %3 offset1:1\n\tds_write2_b32 %1, %4, %5 offset0:2 offset1:3\n\ts_mov_b64 exec, -1"
                 :
                 : "s"(m), "v"(to), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3])
                 : "memory");
}

template <int K0, int K1, int K2>
__device__ __forceinline__ void compact_tile_at(const uint64_t (&acc)[kTileWords], const ColRegs<K0> &c0, const ColRegs<K1> &c1, const ColRegs<K2> &c2,
                                                const uint32_t (&posw)[kTileWords], uint32_t j26, uint32_t word_at) {
    typedef RingRec<K0, K1, K2> L;
    constexpr int R = L::R;
    if constexpr (R == 4) {
#pragma unroll
        for (int w = 0; w < kTileWords; ++w) {
            uint32_t rec[4];
            build_record<K0, K1, K2>(rec, c0, c1, c2, posw[w], j26, w);
            const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)word_at, w);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(acc[w] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)acc[w], 0u));
            masked_store1_r4(acc[w], at + rank * 16u, rec);
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < kTileWords; g += 4) {
        uint64_t m[4];
        uint32_t to[4], d[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int w = g + e;
            m[e] = acc[w];
            uint32_t rec[4];
            build_record<K0, K1, K2>(rec, c0, c1, c2, posw[w], j26, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[e][k] = rec[k];
            const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)word_at, w); // wave-uniform: where word w's first record goes
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[e] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[e], 0u));
            to[e] = at + rank * (uint32_t)(4 * R); // (an LDS address)
        }
        masked_store4<R>(m, to, d);
    }
}

// (Tried and not kept: an unpredicated form in which every lane stores -- survivors at their rank, the others into a trash
// slot of their own behind the ring, v_cndmask on the word instead of s_and_saveexec / branch / restore -- was 5 us slower
// on C3: sixteen full-width ds_write_b64 per tile cost more than the five scalar instructions per word they save.)
// TABLE: the launch covers a tile TABLE -- every segment a GPU owns, each starting on a fresh tile of one virtual row space
// (imm3_table; Engine.scala:176-196: one result over all the per-segment pipelines) -- instead of one segment: a tile's columns
// start where its descriptor says, the last tile of every segment is partial (rolled path, its range a dense one), spans and
// descriptors run over the virtual tile space.  One launch ramp and one exposed prefix chain for the whole table instead of one
// per segment (C5 at one GPU: eight launches of ~125 us each paid 13-25 us of tail).
template <int K0, int K1, int K2, bool TABLE>
__global__ __launch_bounds__(kProjThreads) void k_filter_project(const ProjectArgs a) {
    typedef RingRec<K0, K1, K2> L;
    typedef typename L::vec vec;
    constexpr int R = L::R;
    constexpr uint32_t kCap = kProjRingBytes / (4 * R); // records a streamer's ring holds
    constexpr bool kS2 = K0 == TK_S2 || K1 == TK_S2 || K2 == TK_S2;
    constexpr bool kXpose = kS2 || K0 == TK_I8 || K1 == TK_I8 || K2 == TK_I8;
    constexpr int kTileLoads = ColRegs<K0>::kLoads + ColRegs<K1>::kLoads + ColRegs<K2>::kLoads; // vector-memory instructions per tile
    constexpr int kProjDepth = TABLE ? 1 : project_depth(K0, K1, K2);
    constexpr int kI32 = (K0 == TK_I32) + (K1 == TK_I32) + (K2 == TK_I32);
    // (the three-column instances with two wide columns have no registers to spare for compact_tile_at's four words at a time:
    // they keep the compiler's word-by-word form, wrap-around test included)
    constexpr bool kLeanCompaction = !(kI32 == 2 && K2 != TK_NONE) && !(K0 == TK_I32 && K1 == TK_I8 && K2 == TK_S2);
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kProjStreamers][kProjRingBytes];
    __shared__ __attribute__((aligned(16))) uint8_t s_xpose[kProjStreamers][kXpose ? (kS2 ? kXposeBytes : kXposeBytes / 2) : 16];
    __shared__ __attribute__((aligned(16))) uint64_t s_park[kProjStreamers][kProjParkLines * kTileWords];
    __shared__ RangePub s_pub[kProjStreamers][kProjSlots];
    __shared__ uint32_t s_drained[kProjStreamers]; // ranges the writer has unpacked
    __shared__ uint32_t s_head[kProjStreamers];    // records the writer has freed in the ring (running total)
    __shared__ unsigned long long s_prefix[kProjSlots]; // first output row of the span (slot = span mod kProjSlots), from the first writer ...
    __shared__ uint32_t s_prefix_seq;              // ... and how many spans it has resolved
    __shared__ uint32_t s_abort;
    __shared__ uint32_t s_arrive[kProjSlots];      // streamers that have published their range of the span (slot = span mod kProjSlots)
    __shared__ uint32_t s_span_ready;              // spans of this work-group whose eight ranges are all published ...
    __shared__ uint32_t s_announced;               // ... and how many of them a writer has announced to the other work-groups
    __shared__ uint32_t s_part[kProjStreamers];
    __shared__ uint32_t s_dense_ranges;            // ranges of this work-group that outgrew their ring (the host sizes P by it)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (scalar: what follows from it -- the wave's tiles, its role -- is wave-uniform control flow and SGPR address arithmetic)
    if (a.stamps && threadIdx.x == 0) a.stamps[2 * blockIdx.x] = wall_clock64();
#ifdef IMM3_ABLATE
    const unsigned long long cyc0 = clock64(); // (tools: shader cycles, for the clock the chip holds under this kernel)
#endif
    if (threadIdx.x < kProjStreamers) {
        s_drained[threadIdx.x] = 0u;
        s_head[threadIdx.x] = 0u;
    }
    if (threadIdx.x < kProjSlots) s_arrive[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        s_abort = 0u;
        s_prefix_seq = 0u;
        s_span_ready = 0u;
        s_announced = 0u;
        s_dense_ranges = 0u;
    }
    const uint32_t epoch = (uint32_t)a.finish[kFinishEpoch];
    // ONE launch of this kernel per device at a time: two of them, each holding part of the CUs and waiting for work-groups
    // of its own that the other one keeps from starting, would wait for each other until both time out.  The host chains
    // its launches with events (imm3_api.cpp); this lock is the net under graphs and under anything the host cannot see:
    // a launch that finds another one's ticket in the lock gives up at once (status bit 2) and the host answers the query
    // through the bitmap path.
    // A work-group that finds the device taken, or that finds a flag of THIS run already raised (an earlier work-group of the
    // launch found it taken: the owner may have finished since, and a late-comer that took the lock now would wait for spans
    // that will never be announced), raises the busy flag, marks its own spans dead and streams in count + bitmap mode.
    const unsigned long long ticket = ((unsigned long long)(uintptr_t)a.desc << 8) | (unsigned long long)((epoch & 0xFFu) | 1u);
    if (threadIdx.x == 0 && a.device_lock) {
        bool busy = status_flag_set(a, epoch, kStatusAbandoned | kStatusBusy);
        if (!busy) {
            unsigned long long seen = 0ULL;
            __hip_atomic_compare_exchange_strong(a.device_lock, &seen, ticket, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            busy = seen != 0ULL && seen != ticket;
        }
        if (busy) abandon_run(a, blockIdx.x, epoch, &s_abort, kStatusBusy);
    }
    __syncthreads();
    const int P = a.P;
    uint32_t lane_total = 0; // streamers, lanes 0..15: survivors in the bitmap words they produced

    if (wave < kProjStreamers) {
        // ------------------------------------------------------------------ streamer
        vec *ring = (vec *)s_ring[wave];
        const uint32_t ring_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_ring[wave]; // the ring's LDS address (wave-uniform)
        uint8_t *xp = s_xpose[wave];
        uint64_t *park = s_park[wave];
        const int64_t n_full = a.n_rows / kTileRows;
        // Prefetch: the wave's next kProjDepth full tiles are loading while it works on one.  kProjDepth + 1 register sets take
        // turns (the tile loop calls its body with the roles rotated every tile: no register copies); the prefetch HEAD walks the
        // wave's tiles in the order the loop below meets them -- range by range, span by span.
        ColRegs<K0> A0, B0, C0;
        ColRegs<K1> A1, B1, C1;
        ColRegs<K2> A2, B2, C2;
        // (64 w + lane) << 16 for every word w of a tile: the per-lane, per-word part of a record's first dword, in sixteen registers
        // for the whole kernel (the empty asm makes the values opaque: the compiler keeps them instead of recomputing one per use)
        uint32_t posw[kTileWords];
#pragma unroll
        for (int w = 0; w < kTileWords; ++w) {
            posw[w] = (uint32_t)(64 * w + lane) << 16;
            if constexpr (kI32 <= 1) asm volatile("" : "+v"(posw[w])); // (with more int32 columns the registers are needed for the tiles: one more instruction per word)
        }
        int64_t head_t = ((int64_t)blockIdx.x * kProjStreamers + wave) * P, head_last = 0;
        const int64_t head_jump = ((int64_t)gridDim.x * kProjStreamers - 1) * P; // from the end of a range to the wave's next one
        int head_j = 0;
        auto head_next = [&]() -> int64_t { // the next tile of the wave's sequence (range by range, span by span)
            const int64_t t = head_t;
            ++head_t;
            if (++head_j == P) {
                head_j = 0;
                head_t += head_jump;
            }
            return t;
        };
        // Table queries: the head's next tile and its descriptor, fetched ONE TILE AHEAD of the loads that need it (a scalar load in
        // front of every tile's loads would hold them back by its latency).  A segment's PARTIAL last tile is loaded and evaluated as
        // a whole one -- every column carries 16 KiB of readable slack behind its last row (imm3_api.cpp: kPad) -- and the bits of
        // the rows that do not exist are masked off the words before anything is counted or compacted (rows_in[] says how many
        // exist): a partial tile in the middle of the table costs sixteen scalar ANDs, not a rolled walk and a dense range.  (First
        // version: rolled path + dense range per partial tile -- 98 loader-made segments per 100 M rows took 187 us against 123.)
        // (SGPRs are what this loop is short of -- the sixteen bitmap words alone are 32 -- and every spilled one is a v_readlane per
        // use: the table's head keeps ONE descriptor, 32-bit tile numbers, and no "last valid pointer": behind the table's end it
        // reads the LAST tile's descriptor again, so that the loads the compiler sees on every path stay on readable addresses.)
        int32_t nx_t = 0;
        ProjectTile nx = {};
        auto fetch_desc = [&]() { // (load_tile_desc: a scalar load)
            nx_t = (int32_t)head_next();
            nx = load_tile_desc(a.tile_desc, nx_t < (int32_t)a.n_tiles ? nx_t : (int32_t)a.n_tiles - 1);
        };
        if constexpr (TABLE) fetch_desc();
        auto head_load = [&](ColRegs<K0> &r0, ColRegs<K1> &r1, ColRegs<K2> &r2, uint32_t &rows_loaded) -> int32_t { // -> the tile loaded (-1: none left)
            if constexpr (TABLE) {
                // (the scalar load of the descriptor AFTER this one is issued first: its latency passes while this tile's seventeen
                // vector loads are issued)
                const int32_t this_t = nx_t;
                const ProjectTile cur = nx;
                fetch_desc();
                rows_loaded = cur.rows;
                r0.load(as_global(cur.p[0]), 0, lane);
                r1.load(as_global(cur.p[1]), 0, lane);
                r2.load(as_global(cur.p[2]), 0, lane);
                return this_t < (int32_t)a.n_tiles ? this_t : -1;
            } else {
            int64_t t = head_t;
            if (IMM3_ABLATE_BIT(a, 8)) { // (timing only: all tiles dealt grid-stride -- every wave of the launch reads one contiguous window, as k_filter_tile)
                const int64_t span = t / ((int64_t)P * kProjStreamers), round = span / gridDim.x;
                t = ((round * P + head_j) * gridDim.x + blockIdx.x) * kProjStreamers + wave;
            }
            if (t < n_full) head_last = t;
            else t = head_last; // nothing left: re-read the last tile (a load the compiler sees on every path)
            if constexpr (kProjDepth == 2) { // (asm loads: waited for by wait_tile below, not by the compiler)
                r0.load_untracked(a.cols[0].data, t * kTileRows, lane);
                r1.load_untracked(a.cols[1].data, t * kTileRows, lane);
                r2.load_untracked(a.cols[2].data, t * kTileRows, lane);
            } else { // (one tile ahead: the compiler's own wait -- everything -- is the right one, and these instances may spill)
                r0.load(a.cols[0].data, t * kTileRows, lane);
                r1.load(a.cols[1].data, t * kTileRows, lane);
                r2.load(a.cols[2].data, t * kTileRows, lane);
            }
            (void)head_next();
            (void)rows_loaded;
            return (int32_t)t;
            }
        };
        int32_t tile_in[2] = {-1, -1};        // (table) the tile whose columns are in register set A / B ...
        uint32_t rows_in[2] = {kTileRows, kTileRows}; // ... and how many of its rows exist
        tile_in[0] = head_load(A0, A1, A2, rows_in[0]);
        if constexpr (kProjDepth == 2) head_load(B0, B1, B2, rows_in[1]);
        // ---- the wave's running state: its current span (s, the work-group's i-th), the ring, whether the rows have been given up
        int64_t s = blockIdx.x;
        uint32_t i = 0;
        uint32_t tail_pos = 0, tail_total = 0; // where the next record goes in the ring; records ever put there (minus those taken back by a spill)
        uint32_t head_seen = 0;                // the ring's head as last read from LDS (it only grows: a stale value is a safe one)
        // The rows of this run have been given up (s_abort: busy device, a look-back that timed out here or elsewhere): the writers
        // are gone, and this wave goes on to the end of its tiles in count + bitmap mode -- every range "dense", nothing published,
        // nothing waited for.  The count this work-group adds at the end and its bitmap lines are exact either way.
        bool abandoned = false;
        // ---- the current range: tiles t0 .. t0 + P of span s
        int64_t t0 = 0;
        int j = 0;                  // the next tile's index in the range
        uint32_t range_start = 0;   // ring position of the range's first record
        uint32_t range_cnt = 0;     // survivors of this range
        bool dense = false;         // the range keeps no records (see unpack_dense)
        int parked = 0, first_parked = 0;
        auto flush_park = [&]() { // 4 lines (4 x 16 lanes) per store instruction; the lines of consecutive tiles are contiguous
            lds_wave_sync();
            for (int q = lane >> 4; q < parked; q += 4)
                __builtin_nontemporal_store(park[q * kTileWords + (lane & 15)], a.bitmap + (t0 + first_parked + q) * kTileWords + (lane & 15));
            lds_wave_sync();
            first_parked += parked;
            parked = 0;
        };
        // the range outgrows the ring: it becomes a dense one -- what it has in the ring is given back
        auto to_dense = [&]() {
            tail_total -= range_cnt;
            tail_pos = range_start;
            dense = true;
        };
        auto begin_range = [&]() {
            if (!abandoned && lds_peek(&s_abort)) abandoned = true;
            t0 = (s * kProjStreamers + wave) * P;
            j = 0;
            range_start = tail_pos;
            range_cnt = 0;
            dense = abandoned;
            parked = 0;
            first_parked = 0;
        };
        // the range is streamed: its bitmap lines go out, and it is published for the writers (unless the rows have been given up)
        auto end_range = [&]() {
            if (parked) flush_park();
            if (abandoned) return; // (nobody takes ranges any more)
            // ---- publish the range (the slot's previous range, i - kProjSlots, must have been taken by the writer) ----
            while (!IMM3_ABLATE_BIT(a, 128) && i >= (uint32_t)kProjSlots && lds_peek(&s_drained[wave]) < i - (uint32_t)(kProjSlots - 1)) {
                if (lds_peek(&s_abort)) { abandoned = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (abandoned) return;
            if (dense) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the writer (another wave of this CU) reads the range's bitmap lines: the stores must have landed
            if (lane == 0) {
                if (dense) __hip_atomic_fetch_add(&s_dense_ranges, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                s_pub[wave][i % kProjSlots].start = range_start;
                s_pub[wave][i % kProjSlots].cnt = range_cnt;
                s_pub[wave][i % kProjSlots].dense = dense ? 1u : 0u;
            }
            // the work-group's last range of the span to finish says so: the span can be announced to the other work-groups
            const uint32_t slot = i % kProjSlots;
            uint32_t arrived = 0;
            if (lane == 0) arrived = __hip_atomic_fetch_add(&s_arrive[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
            if (arrived == (uint32_t)kProjStreamers - 1u && lane == 0) { // (behind the other seven's fetch_adds, hence behind their s_pub writes: the LDS serves a wave's operations in order)
                lds_poke(&s_arrive[slot], 0u); // (the slot's next span, i + kProjSlots, is published only after this one was drained)
                // (spans become ready in order: every streamer finishes range i before range i + 1.)  A RELEASE at work-group scope: the
                // other seven streamers' s_pub writes -- and a dense range's fence above -- reach this wave through their relaxed
                // fetch_adds on s_arrive, which is not a release sequence by the letter; the store that the writers acquire is.
                __hip_atomic_store(&s_span_ready, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef IMM3_ABLATE
                if (a.stamps && i < 6) a.stamps[2 * gridDim.x + blockIdx.x * 24 + i] = wall_clock64(); // (tools: span i streamed)
#endif
            }
        };
        // one full tile: its columns are in (c0, c1, c2); the loads of the tile kProjDepth tiles ahead go to (n0, n1, n2), the set
        // that was worked on last
        auto full_tile = [&](ColRegs<K0> &c0, ColRegs<K1> &c1, ColRegs<K2> &c2, ColRegs<K0> &n0, ColRegs<K1> &n1, ColRegs<K2> &n2, uint32_t rows_here, int32_t &next_in, uint32_t &next_rows) {
            // software pipeline (k_filter_tile, finding 11): the wait for this tile's loads sits BEFORE the next loads are issued (vmcnt
            // retires in order: what is in flight behind this tile's loads -- the next tile's, at depth 2 -- is not waited for)
            if constexpr (kProjDepth == 2) wait_tile<kTileLoads>(); // this tile's loads have landed; the next tile's stay in flight
            c0.touch();
            c1.touch();
            c2.touch();
            next_in = head_load(n0, n1, n2, next_rows);
            uint64_t acc[kTileWords]; // wave-uniform words (SGPR pairs)
#pragma unroll
            for (int w = 0; w < kTileWords; ++w) acc[w] = ~0ULL;
            if (IMM3_ABLATE_BIT(a, 64)) { // (timing only: no compares -- every word keeps 6 fixed rows, ~10 % survivors)
#pragma unroll
                for (int w = 0; w < kTileWords; ++w) acc[w] = 0x0101010100010101ULL << (w & 7);
            } else {
                // the narrow columns' LDS transposes are issued first, the int32 columns (kinds are sorted: they come first) are
                // compared while those are in flight, ONE wait, then the narrow columns' compares
                c0.stage(lane, xp);
                c1.stage(lane, xp);
                c2.stage(lane, xp);
                if constexpr (K0 == TK_I32) c0.test(a.cols[0], acc);
                if constexpr (K1 == TK_I32) c1.test(a.cols[1], acc);
                if constexpr (K2 == TK_I32) c2.test(a.cols[2], acc);
                if constexpr (kXpose) lds_reads_landed();
                if constexpr (K0 != TK_I32) c0.test(a.cols[0], acc);
                if constexpr (K1 != TK_I32) c1.test(a.cols[1], acc);
                if constexpr (K2 != TK_I32) c2.test(a.cols[2], acc);
            }
            if constexpr (TABLE) {
                if (rows_here < (uint32_t)kTileRows) { // (wave-uniform) a segment's last tile: the rows behind its end are no rows
#pragma unroll
                    for (int w = 0; w < kTileWords; ++w) acc[w] &= low_mask((int64_t)rows_here - 64 * w);
                }
            }
            uint64_t mine = words_to_lanes(acc);
            if (lane >= kTileWords) mine = 0;
            if (lane < kTileWords) park[parked * kTileWords + lane] = mine;
            const uint32_t pc = (uint32_t)__popcll(mine); // lanes 0..15: survivors of word `lane`
            lane_total += pc;
            if (++parked == kProjParkLines) flush_park();
            // the survivors' records
            const uint32_t incl = row_inclusive_sum(pc);  // lanes 0..15: survivors of words 0..lane
            uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)incl, kTileWords - 1); // wave-uniform: the tile's survivors
            if (IMM3_ABLATE_BIT(a, 4)) cnt = 0; // (no records at all)
            if (!dense && range_cnt + cnt > kCap) to_dense();
            if (dense) { // count and bitmap only
                range_cnt += cnt;
                return;
            }
            if (tail_total + cnt - head_seen > kCap) { // room in the ring?  Only undrained EARLIER ranges can be in the way: the writer frees them
                while (tail_total + cnt - (head_seen = lds_peek(&s_head[wave])) > kCap) {
                    if (lds_peek(&s_abort)) { abandoned = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (abandoned) { // (this tile's line is parked and counted: from here on counts and bitmap lines only)
                dense = true;
                return;
            }
            const uint32_t j26 = (uint32_t)j << 26; // the tile's index in its range, where the records carry it
            if (cnt == 0) {
            } else if (kLeanCompaction && tail_pos + cnt <= kCap) { // the common case: the tile's records do not wrap around the ring
                const uint32_t word_at = ring_at + (tail_pos + incl - pc) * (uint32_t)sizeof(vec); // lane w: LDS address of word w's first record
                compact_tile_at<K0, K1, K2>(acc, c0, c1, c2, posw, j26, word_at);
            } else {
                compact_tile<K0, K1, K2>(acc, c0, c1, c2, posw, j26, [&](uint32_t rank, vec v) {
                    uint32_t idx = tail_pos + rank;
                    if (idx >= kCap) idx -= kCap;
                    ring[idx] = v;
                });
            }
            range_cnt += cnt;
            tail_total += cnt;
            tail_pos += cnt;
            if (tail_pos >= kCap) tail_pos -= kCap;
        };
        // the one partial tile at the end of the segment (of every segment of a table): rolled, bounds-checked; its range is a dense one
        auto partial_tile = [&](int64_t tile) {
            flush_park();
            ++first_parked; // (this tile's line is stored below, not parked: the range's later lines -- a table's next segment -- start behind it)
            if (!dense) to_dense();
            int64_t row0 = tile * kTileRows, valid_rows = a.n_rows - row0;
            const void *d0 = a.cols[0].data, *d1 = a.cols[1].data, *d2 = a.cols[2].data;
            if constexpr (TABLE) { // (rare: a blocking scalar load)
                const ProjectTile d = load_tile_desc(a.tile_desc, tile);
                d0 = as_global(d.p[0]);
                d1 = as_global(d.p[1]);
                d2 = as_global(d.p[2]);
                row0 = 0;
                valid_rows = d.rows;
            }
            ColRegs<K0> c0;
            ColRegs<K1> c1;
            ColRegs<K2> c2;
            uint64_t mine = ~0ULL;
#pragma unroll 1
            for (int w = 0; w < kTileWords; ++w) {
                const int64_t r_in = 64 * w + lane;
                const bool valid = r_in < valid_rows;
                const int64_t r = row0 + (valid ? r_in : 0);
                bool keep = valid;
                if (valid) keep = c0.row(d0, a.cols[0], r) && c1.row(d1, a.cols[1], r) && c2.row(d2, a.cols[2], r);
                const uint64_t m = ballot64(keep);
                if (lane == w) mine &= m;
                range_cnt += (uint32_t)__popcll(m);
            }
            mine &= low_mask(valid_rows - 64 * (int64_t)lane);
            if (lane >= kTileWords) mine = 0;
            const int64_t word = tile * kTileWords + lane;
            if (lane < kTileWords && (TABLE || word < (a.n_rows + 63) / 64)) a.bitmap[word] = mine; // (table: every tile's line exists in full)
            lane_total += (uint32_t)__popcll(mine);
        };
        // One tile of the wave's sequence -- range by range, span by span; false: the wave has no tile left.  The loop below calls it
        // with the register sets in rotating roles (an explicit unroll by their number: no register copies).  A partial tile, or a
        // range that starts behind the data's end, consumes no register set; both only occur in the wave's last range, so the
        // strict alternation holds wherever it matters (a table's partial tiles are whole tiles to this loop).
        // every untracked tile load has landed, and up to here the register sets were the columns' (ColRegs::keep says why)
        auto drain_loads = [&]() {
            wait_tile<0>();
            A0.keep(); A1.keep(); A2.keep();
            B0.keep(); B1.keep(); B2.keep();
            C0.keep(); C1.keep(); C2.keep();
            __builtin_amdgcn_sched_barrier(0);
        };
        // `in` / `rows_here`: (table) the tile whose columns are in (c0, c1, c2) and its valid rows; `next_in` / `next_rows`: where the
        // same is noted for the tile loaded into (n0, n1, n2)
        auto step = [&](ColRegs<K0> &c0, ColRegs<K1> &c1, ColRegs<K2> &c2, ColRegs<K0> &n0, ColRegs<K1> &n1, ColRegs<K2> &n2, int32_t in, uint32_t rows_here, int32_t &next_in,
                        uint32_t &next_rows) -> bool {
            const int64_t tile = t0 + j;
            if constexpr (TABLE) { // (every tile of the table is loaded and evaluated as a whole one; a tile that is not the one loaded lies
                // behind the table's end, and so does everything after it: the sets' alternation no longer matters)
                if ((int32_t)tile == in) full_tile(c0, c1, c2, n0, n1, n2, rows_here, next_in, next_rows);
            } else {
                if (tile < n_full) full_tile(c0, c1, c2, n0, n1, n2, rows_here, next_in, next_rows);
                else {
                    if constexpr (kProjDepth == 2) drain_loads(); // (a step that consumes no register set: what was prefetched is dead from here on)
                    if (tile < a.n_tiles) partial_tile(tile);
                }
            }
            ++j;
            if (j < P && tile + 1 < a.n_tiles) return true; // (wave-uniform)
            end_range();
            s += gridDim.x;
            ++i;
            if (s >= a.n_spans) return false;
            begin_range();
            return true;
        };
        if (s < a.n_spans) {
            begin_range();
            if constexpr (kProjDepth == 2) {
                int32_t unused = 0;
                uint32_t unused_rows = 0;
                for (;;) { // (one segment: a step that keeps its set only occurs in the wave's last range)
                    if (!step(A0, A1, A2, C0, C1, C2, 0, kTileRows, unused, unused_rows)) break;
                    if (!step(B0, B1, B2, A0, A1, A2, 0, kTileRows, unused, unused_rows)) break;
                    if (!step(C0, C1, C2, B0, B1, B2, 0, kTileRows, unused, unused_rows)) break;
                }
            } else {
                for (;;) {
                    if (!step(A0, A1, A2, B0, B1, B2, tile_in[0], rows_in[0], tile_in[1], rows_in[1])) break;
                    if (!step(B0, B1, B2, A0, A1, A2, tile_in[1], rows_in[1], tile_in[0], rows_in[0])) break;
                }
            }
        }
        if constexpr (kProjDepth == 2) drain_loads(); // the tiles prefetched past the wave's last one are still landing
#ifdef IMM3_ABLATE
        if (a.stamps && lane == 0) atomicMax(a.stamps + 26 * gridDim.x + blockIdx.x, (unsigned long long)wall_clock64()); // (tools: this work-group's last streamer is through)
#endif
    } else {
        // ------------------------------------------------------------------ writer
        const int wr = wave - kProjStreamers; // 0: also owns the span's descriptor and its first output row
        const int w_first = wr * kProjPerWriter;
        int64_t s = blockIdx.x;
        if (IMM3_ABLATE_BIT(a, 128)) s = a.n_spans; // (timing only, with bit 4: the writers leave at once -- what their polling costs the streamers)
        // Announce the work-group's next finished span to the other work-groups: its count, the arrival at the round's
        // counter, the round's scan if it is the round's last arrival.  ANY writer does it, as soon as it looks -- before
        // each range it unpacks and while it waits: a span usually finishes streaming while the previous one is being
        // unpacked, and its first output row takes ~10 us of dependent device-scope round trips to come back.  (With one
        // fixed writer announcing span k + 1 after it had unpacked span k the chain was serial: unpack -> announce -> wait
        // -> unpack, 33 us per span on C4.  With the streamer that finishes the span announcing it, the slowest streamer
        // of the slowest work-group paid for the round's scan every round: 20 us per round on C3.)  false: abandoned.
        auto announce = [&]() -> bool {
            if (IMM3_ABLATE_BIT(a, 2)) return true;
            const uint32_t ready = lds_peek(&s_span_ready), ann = lds_peek(&s_announced);
            if (ann >= ready) return true;
            uint32_t got = 0;
            if (lane == 0) {
                uint32_t expect = ann;
                got = __hip_atomic_compare_exchange_strong(&s_announced, &expect, ann + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
            }
            if (!__builtin_amdgcn_readfirstlane((int)got)) return true; // (another writer has it)
#ifdef IMM3_ABLATE
            if (a.stamps && lane == 0 && ann < 6) a.stamps[2 * gridDim.x + blockIdx.x * 24 + 18 + ann] = wall_clock64(); // (tools: span about to be announced)
            // fault injection (imm3_ctx_inject_fault): this work-group never announces this span -- its round never completes, every
            // work-group that waits for a prefix of that round or a later one runs into its poll cap and gives the rows up
            if ((int)blockIdx.x == a.fault_wg && (int)ann == a.fault_span) return true;
#endif
            lds_wave_order();
            const unsigned long long agg = wave_sum_u64(lane < kProjStreamers ? (unsigned long long)s_pub[lane][ann % kProjSlots].cnt : 0ULL);
            const int64_t sp = (int64_t)blockIdx.x + (int64_t)ann * gridDim.x;
            if (span_arrive(a, sp, epoch, agg, lane)) return true;
            if (lane == 0) abandon_run(a, sp, epoch, &s_abort);
            return false;
        };
        for (uint32_t k = 0; s < a.n_spans && !lds_peek(&s_abort); s += gridDim.x, ++k) {
            // the ranges of span s: wait until all eight are published
            const uint32_t slot = k % kProjSlots;
            bool dead = false;
            while (lds_peek(&s_span_ready) < k + 1) { // (a range takes >= 10 us to stream: poll at ~0.5 us, the streamers need the issue slots)
                if (lds_peek(&s_abort)) { dead = true; break; }
                __builtin_amdgcn_s_sleep(16);
            }
            if (dead) break;
            (void)lds_peek_acquire(&s_span_ready); // (pairs with the release store of the streamer that finished the span: its and the other streamers' s_pub entries are visible)
            if (!announce()) break;
            lds_wave_order();
            unsigned long long before = 0; // survivors of the span before this writer's ranges
#pragma unroll 1
            for (int w = 0; w < w_first; ++w) before += s_pub[w][slot].cnt;
            unsigned long long prefix = 0;
            if (wr == 0) {
                bool ok = true;
                if (IMM3_ABLATE_BIT(a, 2)) prefix = (unsigned long long)s * (unsigned long long)(P * kProjStreamers * 128); // (no chained scan: rows land at wrong, but distinct, offsets)
                else ok = desc_wait(a, a.desc + s, epoch, 2u, prefix);
                if (!ok) {
                    if (lane == 0) abandon_run(a, s, epoch, &s_abort);
                    break;
                }
                if (lane == 0) {
                    s_prefix[slot] = prefix; // (the slot's previous span, k - kProjSlots, was read by the other writers before they drained it ...
                    lds_wave_order();
                    lds_poke(&s_prefix_seq, k + 1); // ... and this wave got here only after the streamers published k, i.e. after k - kProjSlots was drained)
                }
            } else {
                while (lds_peek(&s_prefix_seq) < k + 1) {
                    if (lds_peek(&s_abort) || !announce()) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (dead) break;
                lds_wave_order();
                prefix = s_prefix[slot];
            }
#ifdef IMM3_ABLATE
            if (a.stamps && lane == 0 && wr == 0 && k < 6) a.stamps[2 * gridDim.x + blockIdx.x * 24 + 12 + k] = wall_clock64(); // (tools: span k's first row known)
#endif
            // this writer's ranges of the span.  While it unpacks the writer outranks the streamers of its SIMD (the issue arbiter
            // prefers the oldest waves -- the streamers -- and the unpacking is on the critical path of the span's ring space).
            __builtin_amdgcn_s_setprio(2);
            unsigned long long base = prefix + before;
#pragma unroll 1
            for (int q = 0; q < kProjPerWriter && !dead; ++q) {
                if (!announce()) { dead = true; break; } // (the next span may have finished streaming meanwhile)
                const int w = w_first + q;
                const uint32_t cnt = s_pub[w][slot].cnt, start = s_pub[w][slot].start, dense = s_pub[w][slot].dense;
                if (!IMM3_ABLATE_BIT(a, 1)) {
                    const uint32_t tile0 = (uint32_t)((s * kProjStreamers + w) * P);
                    const uint32_t b32 = base > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)base;
                    if (dense) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        unpack_dense<K0, K1, K2, TABLE>(a, (int64_t)tile0, P, base, lane, a.trash + ((size_t)blockIdx.x * kProjWriters + wr) * 64);
                    } else {
                        unpack_range<K0, K1, K2>(a, (const vec *)&s_ring[w][0], start, kCap, cnt, b32, tile0, lane);
                    }
                }
                base += cnt;
                lds_wave_order();
                if (lane == 0) { // the range is free again (behind the ring reads above: a wave's LDS operations execute in order)
                    if (!dense) lds_poke(&s_head[w], lds_peek(&s_head[w]) + cnt);
                    lds_poke(&s_drained[w], k + 1);
                }
            }
            __builtin_amdgcn_s_setprio(0);
#ifdef IMM3_ABLATE
            if (a.stamps && lane == 0 && wr == kProjWriters - 1 && k < 6) a.stamps[2 * gridDim.x + blockIdx.x * 24 + 6 + k] = wall_clock64(); // (tools: span k drained)
#endif
        }
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) lane_total += __shfl_xor(lane_total, d); // lanes 0..15 -> lane 0
    if (lane == 0 && wave < kProjStreamers) s_part[wave] = lane_total;
    __syncthreads();
    if (threadIdx.x == 0) { // the launch's last work-group publishes the count (and bumps the epoch)
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < kProjStreamers; ++w) t += s_part[w];
        if (s_dense_ranges) __hip_atomic_fetch_add(a.finish + kFinishDense + 1, (unsigned long long)s_dense_ranges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // (the sum is read by the launch's last work-group, behind its arrival below)
        if (finish_add(a.finish, t)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            a.finish[kFinishDense] = __hip_atomic_exchange(a.finish + kFinishDense + 1, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the launch's last work-group: every round is over, the arrival counters start the next run at zero
            for (int64_t r = 0; r < a.n_rounds; ++r) a.round_ctr[r] = 0u;
            if (a.device_lock) { // hand the device on (only if the ticket in the lock is this launch's)
                unsigned long long mine = ticket;
                __hip_atomic_compare_exchange_strong(a.device_lock, &mine, 0ULL, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (a.stamps && threadIdx.x == 0) a.stamps[2 * blockIdx.x + 1] = wall_clock64();
#ifdef IMM3_ABLATE
    if (a.stamps && threadIdx.x == 0) a.stamps[27 * gridDim.x + blockIdx.x] = clock64() - cyc0;
#endif
}

// ---------------------------------------------------------------------------------------------
// launchers.  Two translation units share this file: imm3_project.hip holds the one-segment instances and the host-side
// helpers, imm3_project_table.hip (which defines IMM3_PROJECT_TABLE_TU and includes this file) the table instances -- sixteen
// kernels of ~17 000 instructions each per unit, compiled side by side.
// ---------------------------------------------------------------------------------------------
#define IMM3_PROJECT_KINDS(X)                                                                       \
    X(TK_NONE, TK_NONE, TK_NONE)                                                                    \
    X(TK_I32, TK_NONE, TK_NONE) X(TK_I8, TK_NONE, TK_NONE) X(TK_S2, TK_NONE, TK_NONE)               \
    X(TK_I32, TK_I32, TK_NONE) X(TK_I32, TK_I8, TK_NONE) X(TK_I8, TK_I8, TK_NONE)                   \
    X(TK_I32, TK_S2, TK_NONE) X(TK_I8, TK_S2, TK_NONE)                                              \
    X(TK_I32, TK_I32, TK_I32) X(TK_I32, TK_I32, TK_I8) X(TK_I32, TK_I8, TK_I8)                      \
    X(TK_I8, TK_I8, TK_I8) X(TK_I32, TK_I32, TK_S2) X(TK_I32, TK_I8, TK_S2) X(TK_I8, TK_I8, TK_S2)

#ifdef IMM3_PROJECT_TABLE_TU
bool launch_filter_project_table(const ProjectArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (!a.tile_desc) return false;
#define IMM3_PROJECT_CASE(k0, k1, k2)                                                               \
    if (a.kinds[0] == k0 && a.kinds[1] == k1 && a.kinds[2] == k2) {                                 \
        IMM3_LAUNCH((k_filter_project<k0, k1, k2, true>), grid, kProjThreads, s, ev0, ev1, a);     \
        return true;                                                                                \
    }
    IMM3_PROJECT_KINDS(IMM3_PROJECT_CASE)
#undef IMM3_PROJECT_CASE
    return false;
}

// one descriptor per tile: the tile's address in each of the launch's columns and its valid rows
__global__ __launch_bounds__(256) void k_project_tile_desc(const uint32_t *tile_rows, const void *const *p0, const void *const *p1, const void *const *p2, ProjectTile *out, int64_t n_tiles) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    ProjectTile d;
    d.p[0] = p0 ? p0[t] : nullptr;
    d.p[1] = p1 ? p1[t] : nullptr;
    d.p[2] = p2 ? p2[t] : nullptr;
    d.rows = tile_rows[t];
    d.pad = 0u;
    out[t] = d;
}
void launch_project_tile_desc(const uint32_t *tile_rows, const void *const *p0, const void *const *p1, const void *const *p2, ProjectTile *out, int64_t n_tiles, hipStream_t s) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(k_project_tile_desc, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, s, tile_rows, p0, p1, p2, out, n_tiles);
}
#else
bool launch_filter_project(const ProjectArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
#define IMM3_PROJECT_CASE(k0, k1, k2)                                                               \
    if (a.kinds[0] == k0 && a.kinds[1] == k1 && a.kinds[2] == k2) {                                 \
        IMM3_LAUNCH((k_filter_project<k0, k1, k2, false>), grid, kProjThreads, s, ev0, ev1, a);    \
        return true;                                                                                \
    }
    IMM3_PROJECT_KINDS(IMM3_PROJECT_CASE)
#undef IMM3_PROJECT_CASE
    return false;
}

int project_rec_dwords(const int32_t *kinds) {
    const int k[kMaxTileCols] = {kinds[0], kinds[1], kinds[2]};
    return ring_layout(k, -1).dwords;
}

// Work-groups of the launch: ONE per CU (12 waves, > 80 KB of LDS: a CU never takes a second one, and takes the first whatever
// the instance's register footprint), so every work-group of the launch is resident -- they wait on each other's descriptors.
int project_max_grid(const int32_t *kinds, int P) {
    (void)P;
    bool have = false;
#define IMM3_PROJECT_HAVE(k0, k1, k2) have = have || (kinds[0] == k0 && kinds[1] == k1 && kinds[2] == k2);
    IMM3_PROJECT_KINDS(IMM3_PROJECT_HAVE)
#undef IMM3_PROJECT_HAVE
    if (!have) return 0;
    static std::atomic<int> cus_of[kMaxDevices]; // (asked once per device: this sits in every query creation)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (dev >= 0 && dev < kMaxDevices && (cus = cus_of[dev].load(std::memory_order_relaxed)) > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (dev >= 0 && dev < kMaxDevices) cus_of[dev].store(cus, std::memory_order_relaxed);
    return cus;
}
#endif

} // namespace imm3
