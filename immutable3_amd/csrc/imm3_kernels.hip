// imm3_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4), wave64.
//
// The reference's hot path is, per segment:
//   ScanOp.next      decode every used column block into a typed vector, selection = all rows
//                    (engine/.../operator/Scan.scala:28-70; DENSE_* decode is a fixed-width
//                    little-endian reinterpretation, core/.../codec/DenseCodec.scala:37-73)
//   SelectOp*.next   clear the bit of every row that fails (engine/.../operator/Select.scala:25-165)
//   ProjectOp.next   walk the set bits in ascending order and emit the SELECT-list values
//                    (engine/.../operator/Project.scala:37-64)
// On the GPU that becomes, over HBM-resident flat columns:
//   k_filter_tile<KINDS...> / k_filter_generic
//                one fused pass over all predicate columns -> selection bitmap (uint64 words, bit i of a
//                batch <-> word i>>6, bit i&63 == scala.collection.mutable.BitSet == wave64 ballot order)
//                and per-workgroup partial counts
//   k_total      partial counts -> the segment's selected-row count
//   k_scan       per-tile counts from the bitmap + their exclusive prefix (chunked)
//   k_gather     per 16-tile span: expand set bits into a dense LDS list in ascending row order, then write
//                row indices and gather the projected columns with dense, coalesced stores
// All of it is HBM-bound integer/byte work: no MFMA anywhere.
#include "imm3_internal.h"
#include "imm3_device.h"
#include "imm3_tile.h"
#include <hip/hip_ext.h>

namespace imm3 {

// ---------------------------------------------------------------------------------------------
// k_filter_tile<K0, K1, K2>: the hot kernel.  Uniform layout (every non-final block has rows % 64 == 0,
// so the batch-major bitmap is flat: word w <-> rows [64w, 64w+64)).  One wave per 1024-row tile,
// grid-stride; column kinds are compile-time so the descriptors live in SGPRs and ALL loads of a tile
// (every column) are issued before the first compare.
//
// Every kind ends up ROW-STRIDED -- lane l holds row 64j + l of the tile in register j -- so that the v_cmp result
// of register j IS bitmap word j (a wave-uniform SGPR pair): no per-lane bit assembly, no cross-lane gather of bits,
// conjunction = s_and_b64, popcount = s_bcnt1, and the survivors' ranks come from v_mbcnt on that word.
//
//   TK_I32  DENSE_INT       16 row-strided dword loads (lane l reads row 64j + l) put it there directly.
//   TK_I8   DENSE_TINYINT   one 16-byte load per lane (rows 16l .. 16l+15: narrow values stream best as 16-byte loads),
//   TK_S2   DENSE_STRING(2) two of them; the tile is then TRANSPOSED THROUGH LDS: ds_write_b128 of the registers as
//                           loaded (the tile's bytes in row order), 16 ds_read_i8 / ds_read_u16 at row 64j + lane.
//                           17-18 LDS instructions per tile instead of ~100 vector instructions of byte extraction,
//                           bit insertion and ds_bpermute: these kernels were VALU-bound (int8: 61 % of HBM peak).
//
// Measured on MI355X (tools/filter_explore.hip, 100 M int32 rows, 61 interleaved rounds): the column is
// read once, so loads are non-temporal (65 us vs 74 us with the default cache policy; read-only ceiling
// with nt loads 59 us); dword and dwordx4 loads stream at the same rate; the best grid is 512 workgroups
// = 2 per CU = 8 waves/CU (more waves add DRAM page conflicts, fewer starve the memory pipeline).
// ---------------------------------------------------------------------------------------------
// The wave's staging state.  Records wait in the LDS buffer until arena_flush(); the flush itself is issued LATER than
// the decision to flush -- in the pipelined loop right after the next group's loads -- so that the stores' write
// acknowledgement never sits in front of a wait for loads: vmcnt retires in issue order, and the loop's one wait
// (before the prefetch) then only ever sees stores that have had a whole iteration to complete.
struct Arena {
    uint32_t buf_n = 0;   // records waiting in the LDS buffer
    uint32_t arena_n = 0; // records this wave has written to its arena
    uint32_t slot = 0;    // tiles this wave has staged
    uint32_t last = 0;    // records of the tile staged last (the guess for the next one)
};
constexpr int kArenaBufBytes = 8 * 1024; // LDS record buffer per wave: one tile at worst, ~10 tiles at 10 % selectivity
constexpr int arena_buf_bytes(int R) { return R == 1 ? 4 * 1024 : kArenaBufBytes; } // 1-dword records: a whole tile is 4 KiB, and 4 work-groups fit a CU (C4 filter 50.5 -> 47.3 us)
constexpr int kArenaSlots = kMaxArenaSlots; // tiles per wave the start table holds

template <int R>
__device__ __forceinline__ void arena_flush(const TileArgs &a, Arena &A, const uint8_t *lds, int64_t wave_id, int lane) {
    typedef typename RecVec<R>::type vec;
    if (!A.buf_n) return; // wave-uniform
    const vec *l = (const vec *)lds;
    vec *out = (vec *)a.stage_rec + wave_id * a.wave_cap + A.arena_n;
    lds_wave_order();
    if (!IMM3_ABLATED(a, 20)) // (ablation: records compacted in LDS but not stored)
        for (uint32_t i = lane; i < A.buf_n; i += 64) out[i] = l[i];
    lds_wave_order();
    A.arena_n += A.buf_n;
    A.buf_n = 0;
}

// would `n` more records overflow the buffer?
template <int R>
__device__ __forceinline__ bool arena_full(const Arena &A, uint32_t n) { return A.buf_n + n > (uint32_t)(arena_buf_bytes(R) / (4 * R)); }

template <int K0, int K1, int K2>
__device__ __forceinline__ void stage_full_tile(const TileArgs &a, int lane, const ColRegs<K0> &c0, const ColRegs<K1> &c1, const ColRegs<K2> &c2,
                                                const uint64_t (&acc)[kTileWords], uint8_t *lds, uint32_t *tstart, Arena &A, int64_t wave_id) {
    typedef Rec<K0, K1, K2> L;
    typedef typename L::vec vec;
    uint32_t cnt = 0; // wave-uniform
#pragma unroll
    for (int j = 0; j < kTileWords; ++j) cnt += (uint32_t)__popcll(acc[j]);
    if (arena_full<L::R>(A, cnt)) arena_flush<L::R>(a, A, lds, wave_id, lane); // (the pipelined loop has normally seen it coming)
    A.last = cnt;
    if (lane == 0) tstart[A.slot] = A.arena_n + A.buf_n;
    ++A.slot;
    // 4-dword records: a tile with more than 512 survivors does not fit the (empty) buffer -- it goes straight to the arena
    const bool direct = L::R * 4 * kTileRows > arena_buf_bytes(L::R) && cnt > (uint32_t)(arena_buf_bytes(L::R) / (4 * L::R)); // wave-uniform
    vec *l = direct ? (vec *)a.stage_rec + wave_id * a.wave_cap + A.arena_n : (vec *)lds + A.buf_n;
    uint32_t base = 0; // wave-uniform: this tile's records so far
#pragma unroll
    for (int j = 0; j < kTileWords; ++j) {
        const uint64_t m = acc[j];
        uint32_t rec[4] = {(uint32_t)(64 * j) | (uint32_t)lane, 0u, 0u, 0u};
        L::template put<0>(rec, c0.value(j));
        L::template put<1>(rec, c1.value(j));
        L::template put<2>(rec, c2.value(j));
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (__builtin_amdgcn_inverse_ballot_w64(m)) (l + base)[rank] = L::pack(rec); // exec = the word itself; `base` stays scalar
        base += (uint32_t)__popcll(m);
    }
    if (direct) A.arena_n += base;
    else A.buf_n += base;
}

// Narrow-only filter instances (int8 and 2-byte-string columns, no survivor records) evaluate IN THE LANE.  The transposing form
// below -- tile through LDS so that lane l holds row 64 j + l, one ballot per bitmap word, the sixteen words moved into lanes
// 0..15 with 32 v_writelane -- issues ~83 vector-ALU-pipe instructions per int8 tile plus ~70 scalar ones, and at one tile per KiB
// that, not HBM, bounds the kernel: I8 over 100 M rows ran at 330 cycles per tile and SIMD = 21 us = 67 % of 8 TB/s (round 4).  Here
// lane l keeps the 16 CONSECUTIVE rows its 16-byte load brought (16 l .. 16 l + 15), builds their 16-bit mask with three vector
// instructions per row (ColRegs::lane_mask), ANDs the columns' masks, and four neighbouring lanes put their masks together into one
// bitmap word with two DPP moves: lane 4 w owns word w.  No LDS, no ballot, no v_writelane, ~55 instructions per tile.
// With a 2-byte-string column among them the lanes keep TWO runs of 8 rows instead (8 l .. + 7 and 512 + 8 l .. + 7: what the string
// column's dense 16-byte loads bring), eight neighbouring lanes make a word, and lane 8 w owns words w and 8 + w (lane_tile() == 2).
constexpr int lane_tile(int k0, int k1, int k2, bool stage) {
    if (stage || !(k0 == TK_I8 || k0 == TK_S2) || k1 == TK_I32 || k2 == TK_I32) return 0;
    return (k0 == TK_S2 || k1 == TK_S2 || k2 == TK_S2) ? 2 : 1;
}

template <int LANE, int K>
__device__ __forceinline__ void tile_load(ColRegs<K> &c, const void *data, int64_t row0, int lane) {
    if constexpr (LANE == 1) c.load_lane_rows(data, row0, lane);
    else if constexpr (LANE == 2) c.load_lane_rows_split(data, row0, lane);
    else c.load(data, row0, lane);
}

// `earlier`: the tile's words from an earlier pass when the caller has loaded them already (pipelined loop), else null and
// they are loaded here.  A staging launch is the only pass of its chain: it never ANDs.
template <int K0, int K1, int K2, bool STAGE>
__device__ __forceinline__ uint32_t finish_full_tile(const TileArgs &a, int64_t tile, int lane, ColRegs<K0> &c0,
                                                     ColRegs<K1> &c1, ColRegs<K2> &c2, uint8_t *xp, uint8_t *lds, uint32_t *tstart, Arena &A,
                                                     int64_t wave_id, uint64_t *park = nullptr, const uint64_t *earlier = nullptr) {
    if constexpr (lane_tile(K0, K1, K2, STAGE) == 2) {
        const uint32_t m = c0.lane_mask(a.cols[0]) & c1.lane_mask(a.cols[1]) & c2.lane_mask(a.cols[2]); // byte 0: rows 8 lane .. + 7, byte 1: 512 + 8 lane .. + 7
        if (!a.bitmap) return (uint32_t)__popc(m); // count-only run (wave-uniform): every lane counts its own rows
        // eight lanes' bytes -> one word, for both runs at once: pairs (byte permute), quads, then the upper quad's half
        const uint32_t odd = (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0xF5, 0xF, 0xF, true);      // quad_perm [1,1,3,3]
        const uint32_t x = __builtin_amdgcn_perm(odd, m, 0x05010400u);                                 // even lanes: {a, a', b, b'} (a: low run, b: high run)
        const uint32_t y = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xAA, 0xF, 0xF, true);         // quad_perm [2,2,2,2]
        const uint32_t lo_a = __builtin_amdgcn_perm(y, x, 0x05040100u), lo_b = __builtin_amdgcn_perm(y, x, 0x07060302u); // lane 0 of a quad: 32 rows of each run
        const uint32_t hi_a = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo_a, 0x104, 0xF, 0xF, true);  // row_shl:4: the quad above
        const uint32_t hi_b = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo_b, 0x104, 0xF, 0xF, true);
        const bool owner = (lane & 7) == 0; // lane 8 w owns words w and 8 + w: 8 lanes x 8 B = half a line, twice
        const int64_t wl = tile * kTileWords + (lane >> 3);
        uint64_t word_a = ((uint64_t)hi_a << 32) | (uint64_t)lo_a, word_b = ((uint64_t)hi_b << 32) | (uint64_t)lo_b;
        if (a.and_existing) { // (the pipelined loop does not prefetch the earlier words for this layout: `earlier` is null)
            word_a &= owner ? a.bitmap[wl] : 0ULL;
            word_b &= owner ? a.bitmap[wl + kTileWords / 2] : 0ULL;
        }
        if (!owner) word_a = word_b = 0;
        if (owner) {
            if (park) {
                park[lane >> 3] = word_a;
                park[kTileWords / 2 + (lane >> 3)] = word_b;
            } else {
                __builtin_nontemporal_store(word_a, a.bitmap + wl);
                __builtin_nontemporal_store(word_b, a.bitmap + wl + kTileWords / 2);
            }
        }
        return (uint32_t)(__popcll(word_a) + __popcll(word_b));
    } else if constexpr (lane_tile(K0, K1, K2, STAGE) == 1) {
        const uint32_t m = c0.lane_mask(a.cols[0]) & c1.lane_mask(a.cols[1]) & c2.lane_mask(a.cols[2]); // rows 16 lane .. 16 lane + 15
        if (!a.bitmap) return (uint32_t)__popc(m); // count-only run (wave-uniform): every lane counts its own rows
        const uint32_t odd = (uint32_t)__builtin_amdgcn_mov_dpp((int)m, 0xF5, 0xF, 0xF, true);       // quad_perm [1,1,3,3]
        const uint32_t pair = m | (odd << 16);                                                           // (lanes 0 and 2 of a quad: 32 rows)
        const uint32_t high = (uint32_t)__builtin_amdgcn_mov_dpp((int)pair, 0xAA, 0xF, 0xF, true);    // quad_perm [2,2,2,2]
        const bool owner = (lane & 3) == 0; // lane 4 w owns word w: 16 lanes x 8 B = one 128-B line
        const int64_t wl = tile * kTileWords + (lane >> 2);
        uint64_t word = ((uint64_t)high << 32) | (uint64_t)pair;
        if (earlier) word &= *earlier;
        else if (a.and_existing) word &= owner ? a.bitmap[wl] : 0ULL;
        if (!owner) word = 0;
        if (owner) {
            if (park) park[lane >> 2] = word;
            else __builtin_nontemporal_store(word, a.bitmap + wl);
        }
        return (uint32_t)__popcll(word);
    }
    const int64_t w = tile * kTileWords + lane; // lane j < 16 owns bitmap word j of the tile
    uint64_t mine = ~0ULL;
    if constexpr (!STAGE) {
        if (earlier) mine = *earlier;
        else if (a.and_existing) mine = lane < kTileWords ? a.bitmap[w] : 0ULL;
    }
    uint64_t acc[kTileWords]; // wave-uniform words (SGPR pairs)
#pragma unroll
    for (int j = 0; j < kTileWords; ++j) acc[j] = ~0ULL;
    c0.eval(a.cols[0], acc, lane, xp);
    c1.eval(a.cols[1], acc, lane, xp);
    c2.eval(a.cols[2], acc, lane, xp);
    if constexpr (!STAGE) {
        if (!a.bitmap) { // count-only run (imm3_query_run_count; wave-uniform): the words never leave the scalar registers -- no
            uint32_t cnt = 0; // v_writelane, no bitmap line, 12.5 MB per 100 M rows less to store (DESIGN finding 20)
#pragma unroll
            for (int j = 0; j < kTileWords; ++j) cnt += (uint32_t)__popcll(acc[j]);
            return lane == 0 ? cnt : 0u;
        }
    }
    mine &= words_to_lanes(acc);
    if (lane >= kTileWords) mine = 0;
    if (lane < kTileWords && (!STAGE || a.bitmap)) { // 16 lanes x 8 B = one 128-B line  (a staging launch may run without a bitmap: the records carry the positions)
        if (park) park[lane] = mine; // deferred: the line waits in LDS for the wave's next store burst
        else __builtin_nontemporal_store(mine, a.bitmap + w);
    }
    if constexpr (STAGE)
        if (!IMM3_ABLATED(a, 21)) stage_full_tile<K0, K1, K2>(a, lane, c0, c1, c2, acc, lds, tstart, A, wave_id);
    return (uint32_t)__popcll(mine);
}

// A tile with fewer than 1024 valid rows (the end of a segment): rolled, bounds-checked, row-at-a-time.
// `valid_rows` rows starting at element `row0` of each column pointer.
template <int K0, int K1, int K2, bool STAGE>
__device__ __forceinline__ uint32_t partial_tile(const TileArgs &a, int64_t tile, int lane, const void *d0, const void *d1, const void *d2,
                                                 int64_t row0, int64_t valid_rows, ColRegs<K0> &c0, ColRegs<K1> &c1, ColRegs<K2> &c2,
                                                 uint8_t *lds, uint32_t *tstart, Arena &A, int64_t wave_id) {
    typedef Rec<K0, K1, K2> L;
    const int64_t w = tile * kTileWords + lane;
    uint64_t mine = ~0ULL;
    if (a.and_existing) mine = (lane < kTileWords && w < a.n_words) ? a.bitmap[w] : 0ULL;
    uint32_t base = 0;
    typename L::vec *arena_out = nullptr;
    if constexpr (STAGE) { // one tile per segment: its records go straight to the arena, behind everything buffered so far
        arena_flush<L::R>(a, A, lds, wave_id, lane);
        if (lane == 0) tstart[A.slot] = A.arena_n;
        ++A.slot;
        arena_out = (typename L::vec *)a.stage_rec + wave_id * a.wave_cap + A.arena_n;
    }
#pragma unroll 1
    for (int j = 0; j < kTileWords; ++j) {
        const int64_t i = 64 * j + lane;
        const bool valid = i < valid_rows;
        const int64_t r = row0 + (valid ? i : 0);
        bool keep = valid;
        if (valid) keep = c0.row(d0, a.cols[0], r) && c1.row(d1, a.cols[1], r) && c2.row(d2, a.cols[2], r);
        const uint64_t m = ballot64(keep);
        if (lane == j) mine &= m;
        if constexpr (STAGE) {
            uint32_t rec[4] = {(uint32_t)i, 0u, 0u, 0u};
            L::template put<0>(rec, c0.rowval(d0, r));
            L::template put<1>(rec, c1.rowval(d1, r));
            L::template put<2>(rec, c2.rowval(d2, r));
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
            if (keep) arena_out[rank] = L::pack(rec);
            base += (uint32_t)__popcll(m);
        }
    }
    if constexpr (STAGE) A.arena_n += base;
    mine &= low_mask(valid_rows - 64 * (int64_t)lane); // rows past the end are not rows
    if (lane >= kTileWords) mine = 0;
    if (lane < kTileWords && w < a.n_words && a.bitmap) a.bitmap[w] = mine;
    return (uint32_t)__popcll(mine);
}

// T = tiles per wave iteration: narrow columns take several tiles at once so that every wave keeps >= 4 KiB of
// loads in flight (8 waves/CU x 4 KiB is what saturates HBM, see the header comment).
// TABLE selects the tile-table walk (table queries) at compile time, so the single-segment kernel carries none of it.
// DEFER (compile time, like TABLE): the bitmap lines of a wave's tiles are parked in LDS and written in bursts.
// STAGE: the survivors' records are compacted and stored per tile (projecting queries).
// Waves per SIMD the register allocation must leave room for.  The staging launch over a lone 2-byte-string column runs
// 1024 work-groups = 4 waves per SIMD (128 VGPRs each); round 3 found that instance at 129 -- one SGPR-spill register too
// many -- and with it 768 of the 1024 work-groups resident and C4's filter at 62 us instead of 47.
constexpr int tile_min_waves(int k0, int k1, bool stage) { return stage && k0 == TK_S2 && k1 == TK_NONE ? 4 : 1; }

template <int K0, int K1, int K2, int T, bool TABLE, bool DEFER, bool STAGE>
__global__ __launch_bounds__(kBlockThreads, tile_min_waves(K0, K1, STAGE)) void k_filter_tile(const TileArgs a) {
    constexpr int kLane = lane_tile(K0, K1, K2, STAGE); // narrow-only, no records: evaluated in the lane (no transpose)
    constexpr bool kXpose = kLane == 0 && (K0 == TK_I8 || K0 == TK_S2 || K1 == TK_I8 || K1 == TK_S2 || K2 == TK_I8 || K2 == TK_S2);
    constexpr int kStage = STAGE ? arena_buf_bytes(Rec<K0, K1, K2>::R) : 16;
    // narrow-only kernels spend longer on a tile (LDS transpose) than its loads take to issue: the next group's loads go
    // out BEFORE the current group is evaluated.  (With an int32 column the same pipeline measured slower: DESIGN.md finding 8.)
    constexpr bool kPipe = STAGE || kLane != 0 || (kXpose && K0 != TK_I32 && K1 != TK_I32 && K2 != TK_I32);
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWavesPerBlock][kStage];
    __shared__ __attribute__((aligned(16))) uint8_t s_xpose[kWavesPerBlock][kXpose ? kXposeBytes : 16];
    __shared__ uint32_t s_tstart[kWavesPerBlock][STAGE ? kArenaSlots : 1]; // where each staged tile's records start in the wave's arena
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint8_t *lds = s_stage[wave];
    uint8_t *xp = s_xpose[wave];
    // Deferred bitmap (a.defer_lines > 0; dynamic LDS): the 128-byte bitmap lines of a wave's tiles are parked in LDS and
    // written in bursts of a.defer_lines lines (one burst at the end for 100 M rows) instead of one line per tile in between
    // the streaming loads -- HBM read/write turnarounds cost more than the 3 % of bytes the bitmap is (tools/filter_explore:
    // 67.3 -> 62.2 us).
    extern __shared__ __attribute__((aligned(16))) uint64_t s_park[]; // [kWavesPerBlock][a.defer_lines][16] when deferring
    uint64_t *park = DEFER ? s_park + (size_t)wave * a.defer_lines * kTileWords : nullptr; // (a launch without a bitmap never takes a deferring instance: launch_filter_tile)
    if (a.stamps && threadIdx.x == 0) a.stamps[2 * blockIdx.x] = wall_clock64(); // instrumented pass of bench.py only
#ifdef IMM3_ABLATE
    const unsigned long long cyc0 = clock64(); // (tools: shader cycles, for the clock the chip holds under this kernel)
#endif
    uint32_t lane_total = 0; // lanes 0..15: survivors in the words they stored
    Arena A;
    uint32_t *tstart = s_tstart[wave];
    constexpr int kRecDwords = Rec<K0, K1, K2>::R;
    const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;

    if constexpr (TABLE) { // table query: tiles come from the tile table (one partial tile per segment), one tile per step
        // deferred bitmap: the parked lines' tile numbers are kept next to them (partial tiles are stored directly)
        uint64_t *pidx = DEFER ? s_park + (size_t)kWavesPerBlock * a.defer_lines * kTileWords + (size_t)wave * a.defer_lines : nullptr;
        int parked = 0;
        auto flush = [&]() {
            lds_wave_sync();
            for (int q = lane >> 4; q < parked; q += 4)
                __builtin_nontemporal_store(park[q * kTileWords + (lane & 15)], a.bitmap + (int64_t)pidx[q] * kTileWords + (lane & 15));
            lds_wave_sync();
            parked = 0;
        };
        for (int64_t tile = wave_id; tile < a.n_tiles; tile += n_waves) {
            const uint32_t rows_here = a.tile_rows[tile];
            const void *d0 = K0 != TK_NONE ? as_global(a.tile_ptrs[0][tile]) : nullptr; // (as_global: no flat loads through a pointer read from memory)
            const void *d1 = K1 != TK_NONE ? as_global(a.tile_ptrs[1][tile]) : nullptr;
            const void *d2 = K2 != TK_NONE ? as_global(a.tile_ptrs[2][tile]) : nullptr;
            ColRegs<K0> c0;
            ColRegs<K1> c1;
            ColRegs<K2> c2;
            if (rows_here == kTileRows) {
                tile_load<kLane>(c0, d0, 0, lane);
                tile_load<kLane>(c1, d1, 0, lane);
                tile_load<kLane>(c2, d2, 0, lane);
                lane_total += finish_full_tile<K0, K1, K2, STAGE>(a, tile, lane, c0, c1, c2, xp, lds, tstart, A, wave_id, DEFER ? park + parked * kTileWords : nullptr);
                if (DEFER) {
                    if (lane == 0) pidx[parked] = (uint64_t)tile;
                    if (++parked == a.defer_lines) flush(); // wave-uniform
                }
            } else {
                lane_total += partial_tile<K0, K1, K2, STAGE>(a, tile, lane, d0, d1, d2, 0, rows_here, c0, c1, c2, lds, tstart, A, wave_id);
            }
        }
        if (DEFER && parked) flush();
        if constexpr (STAGE) {
            arena_flush<kRecDwords>(a, A, lds, wave_id, lane);
            for (uint32_t i = lane; i < A.slot; i += 64) a.tile_start[wave_id * a.max_slots + i] = tstart[i];
            if (lane == 0) a.tile_start[wave_id * a.max_slots + A.slot] = A.arena_n; // (the arena's end: the last tile's length for the offsets scan)
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) lane_total += __shfl_xor(lane_total, d); // (whichever lanes counted: 0..15, 0, every fourth, or all)
        if (a.finish) block_partial_finish(a.finish, lane_total, lane, wave);
        else block_partial_store(a.block_partials, lane_total, lane, wave);
        if (a.stamps && threadIdx.x == 0) a.stamps[2 * blockIdx.x + 1] = wall_clock64();
        return;
    } else {
    // a chunk of a limit scan (a.chunked) whose predecessors have already selected `limit` rows does nothing at all -- no load, no
    // bitmap line, no arrival at the tally (the count block already holds what the last chunk that ran published): the launch
    // costs its dispatch and nothing else.  Uniform over the whole grid (the words were written by the previous launch).
    if (a.chunked == 1 && a.finish[kFinishLimitRows] >= a.finish[3]) return; // (2: the run's first chunk -- the running words are the previous run's)
    const int64_t n_tiles_here = a.n_tiles;
    const int64_t n_full = a.n_rows / kTileRows;
    const int64_t n_groups = n_full / T;

    int parked = 0;           // lines waiting in LDS
    int64_t first_grp = wave_id; // group of the first parked line
    auto flush = [&]() {      // 4 lines (4 x 16 lanes) per store instruction
        lds_wave_sync();
        for (int q = lane >> 4; q < parked; q += 4) {
            const int64_t tile = (first_grp + (int64_t)(q / T) * n_waves) * T + (q % T);
            __builtin_nontemporal_store(park[q * kTileWords + (lane & 15)], a.bitmap + tile * kTileWords + (lane & 15));
        }
        lds_wave_sync();
        parked = 0;
    };
    ColRegs<K0> n0[T]; // kPipe: the group after the current one, already loading
    ColRegs<K1> n1[T];
    ColRegs<K2> n2[T];
    auto load_group = [&](ColRegs<K0> (&r0)[T], ColRegs<K1> (&r1)[T], ColRegs<K2> (&r2)[T], int64_t grp) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t row0 = (grp * T + t) * kTileRows;
            tile_load<kLane>(r0[t], a.cols[0].data, row0, lane);
            tile_load<kLane>(r1[t], a.cols[1].data, row0, lane);
            tile_load<kLane>(r2[t], a.cols[2].data, row0, lane);
        }
    };
    if (kPipe && wave_id < n_groups) load_group(n0, n1, n2, wave_id);
    for (int64_t grp = wave_id; grp < n_groups; grp += n_waves) {
        ColRegs<K0> c0[T];
        ColRegs<K1> c1[T];
        ColRegs<K2> c2[T];
        uint64_t earlier[T];
        if constexpr (kPipe) {
            // Software pipeline, and where its one wait sits.  vmcnt retires in ISSUE ORDER, so the wait for this group's loads
            // (issued an iteration ago) must come BEFORE the next group's loads are issued -- a wait placed after them would be
            // vmcnt(their count) and cover everything older, including the previous tile's record stores, whose write
            // acknowledgement then sits on the critical path of every tile (measured: C3 filter 82 -> 107 us).  touch() pins the
            // wait here, where the only younger operations are those stores: s_waitcnt vmcnt(2).
#pragma unroll
            for (int t = 0; t < T; ++t) {
                c0[t] = n0[t];
                c1[t] = n1[t];
                c2[t] = n2[t];
                c0[t].touch();
                c1[t].touch();
                c2[t].touch();
            }
            if constexpr (!STAGE) { // a later pass of a multi-pass chain: this group's words so far, ahead of the prefetch
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if constexpr (kLane == 2) earlier[t] = ~0ULL; // (two words per owner lane: finish_full_tile loads them)
                    else if constexpr (kLane == 1) earlier[t] = a.and_existing ? a.bitmap[(grp * T + t) * kTileWords + (lane >> 2)] : ~0ULL; // (lane 4 w owns word w)
                    else earlier[t] = (a.and_existing && lane < kTileWords) ? a.bitmap[(grp * T + t) * kTileWords + lane] : (a.and_existing ? 0ULL : ~0ULL);
                }
            }
            // unconditional (the last iteration re-reads its own group): a load the compiler sees on every path is a load its
            // s_waitcnt can count past
            load_group(n0, n1, n2, grp + n_waves < n_groups ? grp + n_waves : grp);
            if constexpr (STAGE)
                if (arena_full<kRecDwords>(A, 2 * A.last)) arena_flush<kRecDwords>(a, A, lds, wave_id, lane); // the next tile will probably not fit: the buffered records go out here, behind the prefetch
        } else {
            load_group(c0, c1, c2, grp);
        }
        if (DEFER && parked == 0) first_grp = grp;
#pragma unroll
        for (int t = 0; t < T; ++t)
            lane_total += finish_full_tile<K0, K1, K2, STAGE>(a, grp * T + t, lane, c0[t], c1[t], c2[t], xp, lds, tstart, A, wave_id, DEFER ? park + (parked + t) * kTileWords : nullptr,
                                                              (kPipe && !STAGE && kLane != 2) ? &earlier[t] : nullptr);
        if (DEFER) {
            parked += T;
            if (parked + T > a.defer_lines) flush(); // wave-uniform
        }
    }
    if (DEFER && parked) flush();
    // leftovers: fewer than T full tiles, then the one partial tile at the end of the segment
    for (int64_t tile = n_groups * T + wave_id; tile < n_tiles_here; tile += n_waves) {
        const int64_t row0 = tile * kTileRows;
        ColRegs<K0> c0;
        ColRegs<K1> c1;
        ColRegs<K2> c2;
        if (tile < n_full) {
            tile_load<kLane>(c0, a.cols[0].data, row0, lane);
            tile_load<kLane>(c1, a.cols[1].data, row0, lane);
            tile_load<kLane>(c2, a.cols[2].data, row0, lane);
            lane_total += finish_full_tile<K0, K1, K2, STAGE>(a, tile, lane, c0, c1, c2, xp, lds, tstart, A, wave_id);
        } else { // rolled, bounds-checked
            lane_total += partial_tile<K0, K1, K2, STAGE>(a, tile, lane, a.cols[0].data, a.cols[1].data, a.cols[2].data, row0, a.n_rows - row0, c0, c1, c2, lds, tstart, A, wave_id);
        }
    }
    if constexpr (STAGE) {
        arena_flush<kRecDwords>(a, A, lds, wave_id, lane);
        for (uint32_t i = lane; i < A.slot; i += 64) a.tile_start[wave_id * a.max_slots + i] = tstart[i];
        if (lane == 0) a.tile_start[wave_id * a.max_slots + A.slot] = A.arena_n; // (the arena's end: the last tile's length for the offsets scan)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lane_total += __shfl_xor(lane_total, d); // (whichever lanes counted: 0..15, 0, every fourth, or all)
    if (a.chunked) block_partial_finish_chunk(a.finish, lane_total, lane, wave, (unsigned long long)n_tiles_here, a.chunked == 2);
    else if (a.finish) block_partial_finish(a.finish, lane_total, lane, wave); // the last pass also reduces the count
    else block_partial_store(a.block_partials, lane_total, lane, wave);
    if (a.stamps && threadIdx.x == 0) a.stamps[2 * blockIdx.x + 1] = wall_clock64(); // after the barrier in the store above
#ifdef IMM3_ABLATE
    if (a.stamps && threadIdx.x == 0) a.stamps[2 * gridDim.x + blockIdx.x] = clock64() - cyc0;
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// k_filter_generic: any column kind (any string width, long IN-lists), any layout; one wave per bitmap
// word per iteration.
//   uniform layout (word_row_base == null): word w covers rows [64w, min(64w+64, n_rows))
//   ragged layout (arbitrary block sizes, e.g. the loader's trailing 1-row block): each
//   batch's BitSet starts on a fresh word, word w covers rows [base[w], base[w] + nvalid[w]).
// ---------------------------------------------------------------------------------------------

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

__global__ __launch_bounds__(kBlockThreads) void k_filter_generic(const FilterArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t wave_total = 0;
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
        if (lane == 0) a.bitmap[w] = acc;
        wave_total += cnt;
    }
    block_partial_store(a.block_partials, wave_total, lane, wave);
}

// ---------------------------------------------------------------------------------------------
// k_total: one wave sums the last filter launch's per-workgroup partials -> selected-row count of the
// segment, and the number of rows ProjectOp will emit.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_total(const TotalArgs a) {
    const int lane = threadIdx.x;
    unsigned long long v = 0;
    // 16-byte loads, several in flight: with 4096 partials (k_filter_pfor's grid) a dword-per-lane loop was 64 dependent
    // round trips (10-18 us)
    const uint4 *p4 = (const uint4 *)a.block_partials;
    const int n4 = a.n_partials >> 2;
#pragma unroll 4
    for (int i = lane; i < n4; i += 64) {
        const uint4 x = p4[i];
        v += (unsigned long long)x.x + x.y + x.z + x.w;
    }
    for (int i = (n4 << 2) + lane; i < a.n_partials; i += 64) v += a.block_partials[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) {
        *a.total = v;
        *a.n_emit = (a.limit > 0 && v > (unsigned long long)a.limit) ? (unsigned long long)a.limit : v;
        count_log_append(a.total, v); // a.total is the head of the query's {total, n_emit, status, limit, tally, log ...} block
    }
}

// ---------------------------------------------------------------------------------------------
// k_sum_counts: one wave adds the counts of the queries a rank ran in this pass (lane i reads query i's count word)
// into the word the RCCL all-reduce sends (Engine.scala:176-196: the per-segment pipelines' results meet in one place).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sum_counts(const SumCountsArgs a) {
    const int lane = threadIdx.x;
    unsigned long long v = lane < a.n ? *a.src[lane] : 0ULL;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) *a.dst = (a.accumulate ? *a.dst : 0ULL) + v;
}

void launch_sum_counts(const SumCountsArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(64), 0, s, a); }

// ---------------------------------------------------------------------------------------------
// k_scan: per-tile survivor counts straight from the bitmap (thread t popcounts the 16 words = one 128-B line
// of tile t), exclusive prefix within chunks of kChunkTiles tiles, per-chunk sums.  Reading the 12.5 MB bitmap here
// is cheaper than making the hot filter kernel store a 4-byte count per tile.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kChunkTiles) void k_scan(const ScanArgs a) {
    __shared__ uint32_t s_wave[kChunkTiles / 64];
    __shared__ uint32_t s_cnt[kChunkTiles];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * kChunkTiles;
    // counts: the chunk's bitmap read in 16-byte pieces, consecutive threads on consecutive pieces (8 threads = one tile's 128
    // bytes).  One thread per tile reading its own 128 bytes was 64 partial lines per load instruction: 8.0 us for 12.5 MB.
    const uint4 *p = (const uint4 *)(a.bitmap + tile0 * kTileWords);
    // (a limit scan that stopped early left the bitmap lines behind *scanned_tiles untouched: those tiles count as empty)
    const int64_t live_tiles = a.scanned_tiles ? ((int64_t)*a.scanned_tiles < a.n_tiles ? (int64_t)*a.scanned_tiles : a.n_tiles) : a.n_tiles;
    const int64_t here = live_tiles - tile0 < 0 ? 0 : (live_tiles - tile0 < kChunkTiles ? live_tiles - tile0 : (int64_t)kChunkTiles);
    const int64_t pieces = here * (kTileWords / 2); // the bitmap is allocated in whole tiles
    if (a.rec_tile_start) { // (block-uniform) no bitmap was stored: the tile's records in its wave's arena say how many rows survived
        const int64_t tile = tile0 + t;
        uint32_t c = 0;
        if (tile < a.n_tiles) {
            int64_t w, slot; // which wave of the staging launch took this tile, and as its how-manieth (k_emit finds it the same way)
            if (tile < a.rec_main_tiles) {
                const int64_t g = tile / a.rec_T;
                w = g % a.rec_n_waves;
                slot = (g / a.rec_n_waves) * a.rec_T + tile % a.rec_T;
            } else {
                const int64_t idx = tile - a.rec_main_tiles, n_groups = a.rec_main_tiles / a.rec_T;
                w = idx % a.rec_n_waves;
                slot = (w < n_groups ? ((n_groups - 1 - w) / a.rec_n_waves + 1) * a.rec_T : 0) + idx / a.rec_n_waves;
            }
            const uint32_t *ts = a.rec_tile_start + w * a.rec_max_slots + slot;
            c = ts[1] - ts[0];
        }
        s_cnt[t] = c;
    } else {
#pragma unroll
        for (int i = 0; i < kTileWords / 2; ++i) {
            const int64_t q = (int64_t)i * kChunkTiles + t;
            uint32_t c = 0;
            if (q < pieces) {
                const uint4 v = p[q];
                c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
            }
            c += __shfl_xor(c, 1);
            c += __shfl_xor(c, 2);
            c += __shfl_xor(c, 4);
            if ((t & 7) == 0) s_cnt[q >> 3] = c; // tile q / 8 of the chunk
        }
    }
    __syncthreads();
    const int64_t tile = tile0 + t;
    const uint32_t c = s_cnt[t];
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
    if (t == kChunkTiles - 1) {
        a.chunk_sums[blockIdx.x] = incl;
        // A projecting run takes the selected-row count from here instead of a k_total launch: the chunk sums meet in one
        // relaxed packed atomic (arrivals in the high bits, the running count in the low 40), the last chunk publishes.
        if (a.finish) finish_add(a.finish, incl);
    }
}

// ---------------------------------------------------------------------------------------------
// k_gather: ProjectOp.  One workgroup per SPAN of 16 tiles (256 bitmap words, 16384 rows).
//   A  thread t loads word t of the span; block-wide exclusive scan of the popcounts
//   B  every thread expands ITS word's set bits (ctz loop) into an ascending list of in-span positions in LDS
//   C  the list is walked densely: thread i handles survivor i, so row-index / value stores are contiguous
//      and the column gathers are ascending.
// ---------------------------------------------------------------------------------------------
// phases A and B of the gather: the span's 256 bitmap words -> survivor count, the span's first output slot, and the
// ascending list of in-span positions in LDS.  Returns the number of rows this span emits (limit / capacity clamped).
struct SpanScratch {
    uint16_t list[kSpanWords * 64]; // 32 KiB
    uint32_t wave[kWavesPerBlock];
    uint32_t toff[kSpanTiles];      // survivors of the span before each of its tiles
    unsigned long long base;
};

__device__ __forceinline__ uint32_t span_expand(const GatherArgs &a, int64_t span, SpanScratch &S, unsigned long long &base_out) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int64_t tile0 = span * kSpanTiles;
    const int64_t w = span * kSpanWords + t;
    uint64_t word = w < a.n_words ? a.bitmap[w] : 0ULL;
    const uint32_t pc = (uint32_t)__popcll(word);
    uint32_t incl = pc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) S.wave[wave] = incl;
    if (wave == 0) { // offset of the span's first survivor among the segment's survivors
        const int64_t chunk = tile0 / kChunkTiles;
        unsigned long long part = 0;
        for (int64_t i = lane; i < chunk; i += 64) part += a.chunk_sums[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
        if (lane == 0) S.base = part + a.tile_offsets[tile0];
        if (lane < kSpanTiles) S.toff[lane] = tile0 + lane < a.n_tiles ? a.tile_offsets[tile0 + lane] - a.tile_offsets[tile0] : 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t off = incl - pc;
    uint32_t total = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) {
        if (i < wave) off += S.wave[i];
        total += S.wave[i];
    }
    const unsigned long long base = S.base;
    uint32_t n_out = total;
    if (a.limit > 0) {
        if (base >= (unsigned long long)a.limit) n_out = 0;
        else if (base + total > (unsigned long long)a.limit) n_out = (uint32_t)((unsigned long long)a.limit - base);
    }
    if (base + n_out > a.cap_rows) n_out = base >= a.cap_rows ? 0u : (uint32_t)(a.cap_rows - base);
    if (n_out) { // block-uniform
        while (word) {
            const int b = __builtin_ctzll(word);
            S.list[off++] = (uint16_t)(t * 64 + b);
            word &= word - 1;
        }
    }
    __syncthreads();
    base_out = base;
    return n_out;
}

// Phase C, general form: any layout (ragged, table), any width, any number of columns; one column at a time.
__global__ __launch_bounds__(kBlockThreads) void k_gather(const GatherArgs a) {
    __shared__ SpanScratch S;
    const int t = threadIdx.x;
    const int64_t n_spans = (a.n_tiles + kSpanTiles - 1) / kSpanTiles;
    const int64_t scanned = a.scanned_tiles ? (int64_t)*a.scanned_tiles : a.n_tiles; // (a limit scan that stopped early: nothing behind it)
    for (int64_t span = blockIdx.x; span < n_spans; span += gridDim.x) { // block-uniform trip count
        const int64_t tile0 = span * kSpanTiles;
        if (tile0 >= scanned) break; // block-uniform
        unsigned long long base;
        const uint32_t n_out = span_expand(a, span, S, base);
        for (uint32_t i = t; i < n_out; i += kBlockThreads) {
            const uint32_t r = S.list[i];
            const unsigned long long out = base + i;
            int64_t row = a.word_row_base
                              ? (int64_t)a.word_row_base[span * kSpanWords + (r >> 6)] + (r & 63)
                              : span * (int64_t)(kSpanWords * 64) + r;
            if (a.row_index) a.row_index[out] = (uint32_t)row; // table queries: virtual row = tile * 1024 + position
            const int64_t tile = tile0 + (r >> 10);
            const bool staged_tile = a.tile_rows ? a.tile_rows[tile] == (uint32_t)kTileRows : tile < a.n_staged_tiles;
            for (int pj = 0; pj < a.n_proj; ++pj) {
                ProjCol pc2 = a.proj[pj];
                if (pc2.tile_ptrs) { // table query: the column of this tile's segment, position within the tile
                    pc2.src = as_global(pc2.tile_ptrs[tile]);
                    row = r & (kTileRows - 1);
                }
                if (pc2.staged && staged_tile) { // survivors' values were compacted per tile by the filter kernel
                    const int64_t sidx = tile * kTileRows + (i - S.toff[r >> 10]);
                    store_value_rt(pc2.dst, pc2.width, out, load_value_rt(pc2.staged, pc2.width, sidx));
                    continue;
                }
                if (pc2.width == 1 || pc2.width == 2 || pc2.width == 4) {
                    store_value_rt(pc2.dst, pc2.width, out, load_value_rt(pc2.src, pc2.width, row));
                } else if (pc2.width == 8) {
                    ((uint64_t *)pc2.dst)[out] = ((const uint64_t *)pc2.src)[row];
                } else {
                    const uint8_t *sp = (const uint8_t *)pc2.src + row * (int64_t)pc2.width;
                    uint8_t *d = (uint8_t *)pc2.dst + out * (uint64_t)pc2.width;
                    for (int b = 0; b < pc2.width; ++b) d[b] = sp[b];
                }
            }
        }
        __syncthreads(); // the scratch is reused by the next span
    }
}

// Phase C, the fast form: one uniform segment (no ragged map, no tile table), N4 + N2 + N1 <= 4 projected columns of 4, 2
// and 1 bytes (the launcher sorts them in that order; widths are compile time, so the walk has no branches).  Every
// thread takes kGatherUnroll survivors per step and issues EVERY load of the step -- all columns, all survivors --
// before the first store.  The general form's load -> wait -> store per column cost one full memory round trip per
// column and step, which is what that kernel's time was (C4, 3 columns: 72 us = 4.8 rounds of work-groups x 2 steps x
// 3 round trips): it was latency-serialised, not bandwidth- or instruction-bound.
constexpr int kGatherUnroll = 2;

template <int N4, int N2, int N1>
__global__ __launch_bounds__(kBlockThreads) void k_gather_plain(const GatherArgs a) {
    constexpr int NP = N4 + N2 + N1;
    __shared__ SpanScratch S;
    const int t = threadIdx.x;
    const int64_t n_spans = (a.n_tiles + kSpanTiles - 1) / kSpanTiles;
    const int64_t scanned = a.scanned_tiles ? (int64_t)*a.scanned_tiles : a.n_tiles; // (a limit scan that stopped early: nothing behind it)
    for (int64_t span = blockIdx.x; span < n_spans; span += gridDim.x) { // block-uniform trip count
        const int64_t tile0 = span * kSpanTiles;
        if (tile0 >= scanned) break; // block-uniform
        unsigned long long base;
        const uint32_t n_out = span_expand(a, span, S, base);
        for (uint32_t i0 = t; i0 < n_out; i0 += kGatherUnroll * kBlockThreads) {
            uint32_t val[kGatherUnroll][NP > 0 ? NP : 1];
            uint32_t rowv[kGatherUnroll];
#pragma unroll
            for (int u = 0; u < kGatherUnroll; ++u) { // all loads of the step (out-of-range survivors re-read survivor i0: no branch) ...
                const uint32_t i = i0 + u * kBlockThreads < n_out ? i0 + u * kBlockThreads : i0;
                const uint32_t r = S.list[i];
                const int64_t row = span * (int64_t)(kSpanWords * 64) + r;
                rowv[u] = (uint32_t)row;
                const int64_t tile = tile0 + (r >> 10);
                const bool staged_tile = tile < a.n_staged_tiles;
                const int64_t sidx = tile * kTileRows + (i - S.toff[r >> 10]);
#pragma unroll
                for (int pj = 0; pj < NP; ++pj) {
                    const bool st = a.proj[pj].staged != nullptr && staged_tile; // compacted per tile by the filter kernel
                    const void *src = st ? a.proj[pj].staged : a.proj[pj].src;
                    const int64_t idx = st ? sidx : row;
                    if (pj < N4) val[u][pj] = load_value<4>(src, idx);
                    else if (pj < N4 + N2) val[u][pj] = load_value<2>(src, idx);
                    else val[u][pj] = load_value<1>(src, idx);
                }
            }
#pragma unroll
            for (int u = 0; u < kGatherUnroll; ++u) { // ... before its first store
                const uint32_t i = i0 + u * kBlockThreads;
                if (i < n_out) {
                    const unsigned long long out = base + i;
                    if (a.row_index) a.row_index[out] = rowv[u];
#pragma unroll
                    for (int pj = 0; pj < NP; ++pj) {
                        if (pj < N4) store_value<4>(a.proj[pj].dst, out, val[u][pj]);
                        else if (pj < N4 + N2) store_value<2>(a.proj[pj].dst, out, val[u][pj]);
                        else store_value<1>(a.proj[pj].dst, out, val[u][pj]);
                    }
                }
            }
        }
        __syncthreads(); // the scratch is reused by the next span
    }
}

// ---------------------------------------------------------------------------------------------
// k_emit: ProjectOp over the survivor records the filter kernel staged (one uniform segment, unlimited projection).
// One work-group per group of kEmitTiles tiles; the group's tile offsets sit in LDS and every THREAD owns one output
// row: it finds the row's tile by binary search in LDS (log2 kEmitTiles steps), loads the row's record, takes the staged columns
// out of it, gathers the others at the record's position -- all NG gathers of all kEmitUnroll rows in flight
// together -- and stores.  No bitmap expansion, no per-tile lane waste (a tile has ~20 survivors at 2 % selectivity,
// ~100 at 10 %), every thread independent of every other.
// ---------------------------------------------------------------------------------------------
constexpr int kEmitUnroll = 2;
constexpr uint32_t kEmitQuadMinRows = kEmitTiles * kTileRows / 20; // gathers present: quads from 5 % survivors up

// One output row per lane is bound by vector-memory ISSUE, not bytes: every store instruction moves 256 B (64 B for an
// int8 column), and each of C3's three store streams cost the same ~7 us whatever its width.  So a lane owns FOUR
// consecutive output rows -- an aligned quad of the global output index -- and stores 16 / 8 / 4 bytes per column: a
// quarter of the store instructions (C3 42 -> 36 us, 50 % survivors 264 -> 168 us).  The quads that straddle the
// group's first and last row are written row by row (the neighbouring group writes the rest of them).  With gathers
// and few survivors (C4: 20 a tile) the row-per-lane form is the faster one -- the group is one dependent chain
// offsets -> record -> gather -> store, and four searches per lane lengthen it -- so the form is chosen per group.
template <int R, int NG>
__global__ __launch_bounds__(kBlockThreads) void k_emit(const EmitArgs a) {
    typedef typename RecVec<R>::type vec;
    __shared__ uint32_t s_off[kEmitTiles + 1];
    __shared__ unsigned long long s_addr[kEmitTiles]; // first record of each tile in the staging area
    __shared__ unsigned long long s_base;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int64_t n_groups = (a.n_tiles + kEmitTiles - 1) / kEmitTiles;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) { // block-uniform trip count
        const int64_t tile0 = g * kEmitTiles;
        const int64_t chunk = tile0 / kChunkTiles; // kChunkTiles % kEmitTiles == 0: a group never straddles chunks
        if (t <= kEmitTiles) {
            const int64_t tile = tile0 + t;
            s_off[t] = (tile < a.n_tiles && tile / kChunkTiles == chunk) ? a.tile_offsets[tile] : a.chunk_sums[chunk];
            if (t < kEmitTiles && tile < a.n_tiles) { // which wave of the filter launch staged this tile, and as its how-manieth
                int64_t w, slot;
                if (tile < a.main_tiles) {
                    const int64_t g = tile / a.T;
                    w = g % a.n_waves;
                    slot = (g / a.n_waves) * a.T + tile % a.T;
                } else {
                    const int64_t idx = tile - a.main_tiles, n_groups = a.main_tiles / a.T;
                    w = idx % a.n_waves;
                    slot = (w < n_groups ? ((n_groups - 1 - w) / a.n_waves + 1) * a.T : 0) + idx / a.n_waves;
                }
                s_addr[t] = (unsigned long long)(w * a.wave_cap) + a.tile_start[w * a.max_slots + slot];
            }
        } else if (t >= 128 && t < 192) { // one wave: survivors of the chunks before this one
            unsigned long long part = 0;
            for (int64_t i = lane; i < chunk; i += 64) part += a.chunk_sums[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
            if (lane == 0) s_base = part;
        }
        __syncthreads();
        const uint32_t o0 = s_off[0];
        const unsigned long long base = s_base + o0;
        uint32_t n_here = s_off[kEmitTiles] - o0;
        if (base + n_here > a.cap_rows) n_here = base >= a.cap_rows ? 0u : (uint32_t)(a.cap_rows - base); // block-uniform
        const bool quads = IMM3_ABLATED(a, 35) || (!IMM3_ABLATED(a, 34) && (NG == 0 || n_here >= kEmitQuadMinRows)); // block-uniform
        if (quads) {
            const unsigned long long q0 = base >> 2;                                     // first quad that holds a row of this group
            const uint32_t n_quads = n_here ? (uint32_t)(((base + n_here + 3) >> 2) - q0) : 0u;
            for (uint32_t qi = t; qi < n_quads; qi += kBlockThreads) {
                const unsigned long long out0 = (q0 + qi) << 2;
                vec rec[4];
                uint32_t rowv[4];
                bool ok[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { // four independent searches + record loads (a row outside the group re-reads row 0: no branch)
                    const unsigned long long o = out0 + e;
                    ok[e] = o >= base && o < base + n_here;
                    const uint32_t k = ok[e] ? (uint32_t)(o - base) : 0u;
                    int lo = 0;
#pragma unroll
                    for (int step = kEmitTiles / 2; step >= 1; step >>= 1)
                        if (s_off[lo + step] - o0 <= k) lo += step; // the LAST tile whose first survivor is <= k: the one that holds row k
                    const uint32_t j = k - (s_off[lo] - o0);
                    rec[e] = ((const vec *)a.stage)[s_addr[lo] + j];
                    rowv[e] = (uint32_t)(tile0 + lo);
                }
                uint32_t rw[4][4]; // [row][record dword]
                uint32_t gv[4][NG > 0 ? NG : 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) { // every gather of the quad before the first store
                    if constexpr (R == 1) { rw[e][0] = rec[e]; rw[e][1] = rw[e][2] = rw[e][3] = 0u; }
                    else if constexpr (R == 2) { rw[e][0] = rec[e].x; rw[e][1] = rec[e].y; rw[e][2] = rw[e][3] = 0u; }
                    else { rw[e][0] = rec[e].x; rw[e][1] = rec[e].y; rw[e][2] = rec[e].z; rw[e][3] = rec[e].w; }
                    rowv[e] = rowv[e] * (uint32_t)kTileRows + (rw[e][0] & (uint32_t)(kTileRows - 1));
#pragma unroll
                    for (int c = 0; c < NG; ++c) { // the launcher puts the gathered columns first
                        const int64_t byte = (int64_t)rowv[e] * a.cols[c].width;
                        gv[e][c] = ((const uint32_t *)a.cols[c].src)[byte >> 2] >> (8 * ((uint32_t)byte & 3u));
                    }
                }
                const bool whole = ok[0] && ok[3]; // (the group's rows are contiguous: first and last in => all four in)
                if (whole) {
                    if (a.row_index) *(uint4 *)(a.row_index + out0) = make_uint4(rowv[0], rowv[1], rowv[2], rowv[3]);
#pragma unroll
                    for (int c = 0; c < NG; ++c) store_quad(a.cols[c].dst, a.cols[c].width, out0, gv[0][c], gv[1][c], gv[2][c], gv[3][c]);
                    for (int c = NG; c < a.n_cols; ++c) {
                        const int d = a.cols[c].rec_dword, sh = a.cols[c].rec_shift, w = a.cols[c].width;
                        uint32_t v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (d == 0 ? rw[e][0] : (d == 1 ? rw[e][1] : (d == 2 ? rw[e][2] : rw[e][3]))) >> sh;
                        store_quad(a.cols[c].dst, w, out0, v[0], v[1], v[2], v[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!ok[e]) continue;
                        const unsigned long long out = out0 + e;
                        if (a.row_index) a.row_index[out] = rowv[e];
#pragma unroll
                        for (int c = 0; c < NG; ++c) store_value_rt(a.cols[c].dst, a.cols[c].width, out, gv[e][c]);
                        for (int c = NG; c < a.n_cols; ++c) {
                            const int d = a.cols[c].rec_dword;
                            const uint32_t word = d == 0 ? rw[e][0] : (d == 1 ? rw[e][1] : (d == 2 ? rw[e][2] : rw[e][3]));
                            store_value_rt(a.cols[c].dst, a.cols[c].width, out, word >> a.cols[c].rec_shift);
                        }
                    }
                }
            }
        } else {
            for (uint32_t k0 = t; k0 < n_here; k0 += kEmitUnroll * kBlockThreads) {
                vec rec[kEmitUnroll];
                uint32_t rowv[kEmitUnroll];
                uint32_t gv[kEmitUnroll][NG > 0 ? NG : 1];
#pragma unroll
                for (int u = 0; u < kEmitUnroll; ++u) { // the records of the step (an out-of-range row re-reads row k0: no branch)
                    const uint32_t k = k0 + u * kBlockThreads < n_here ? k0 + u * kBlockThreads : k0;
                    int lo = 0;
#pragma unroll
                    for (int step = kEmitTiles / 2; step >= 1; step >>= 1)
                        if (s_off[lo + step] - o0 <= k) lo += step; // the LAST tile whose first survivor is <= k: the one that holds row k
                    const int64_t tile = tile0 + lo;
                    const uint32_t j = k - (s_off[lo] - o0);
                    rec[u] = ((const vec *)a.stage)[s_addr[lo] + j];
                    rowv[u] = (uint32_t)tile; // (position added once the record is here)
                }
#pragma unroll
                for (int u = 0; u < kEmitUnroll; ++u) { // every gather of the step before the first store
                    uint32_t r0;
                    if constexpr (R == 1) r0 = rec[u];
                    else r0 = rec[u].x;
                    rowv[u] = rowv[u] * (uint32_t)kTileRows + (r0 & (uint32_t)(kTileRows - 1));
#pragma unroll
                    for (int c = 0; c < NG; ++c) { // the launcher puts the gathered columns first
                        const int64_t byte = (int64_t)rowv[u] * a.cols[c].width;
                        gv[u][c] = ((const uint32_t *)a.cols[c].src)[byte >> 2] >> (8 * ((uint32_t)byte & 3u));
                    }
                }
#pragma unroll
                for (int u = 0; u < kEmitUnroll; ++u) {
                    const uint32_t k = k0 + u * kBlockThreads;
                    if (k < n_here) {
                        const unsigned long long out = base + k;
                        if (a.row_index) a.row_index[out] = rowv[u];
#pragma unroll
                        for (int c = 0; c < NG; ++c) store_value_rt(a.cols[c].dst, a.cols[c].width, out, gv[u][c]);
                        uint32_t rw[4];
                        if constexpr (R == 1) { rw[0] = rec[u]; rw[1] = rw[2] = rw[3] = 0u; }
                        else if constexpr (R == 2) { rw[0] = rec[u].x; rw[1] = rec[u].y; rw[2] = rw[3] = 0u; }
                        else { rw[0] = rec[u].x; rw[1] = rec[u].y; rw[2] = rec[u].z; rw[3] = rec[u].w; }
                        for (int c = NG; c < a.n_cols; ++c) { // staged columns: already in the record
                            const int d = a.cols[c].rec_dword;
                            const uint32_t word = d == 0 ? rw[0] : (d == 1 ? rw[1] : (d == 2 ? rw[2] : rw[3]));
                            store_value_rt(a.cols[c].dst, a.cols[c].width, out, word >> a.cols[c].rec_shift);
                        }
                    }
                }
            }
        }
        __syncthreads(); // s_off / s_base are reused by the next group
    }
}

// ---------------------------------------------------------------------------------------------
// k_read_stream: read-only ceiling probe (same tiling, loads and grid as k_filter_tile<I32>, no compares, no stores)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void k_read_stream(const int32_t *data, int64_t n_tiles, int32_t *sink) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int32_t acc = 0;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int32_t *p = data + tile * kTileRows + lane;
        int32_t v[kTileWords];
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
#pragma unroll
        for (int j = 0; j < kTileWords; ++j) acc ^= v[j];
    }
    if (acc == 0x5A5A5A5A) *sink = acc; // keeps the loads alive; practically never taken
}

void launch_read_stream(const int32_t *data, int64_t n_tiles, int32_t *sink, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    IMM3_LAUNCH(k_read_stream, 512, kBlockThreads, s, ev0, ev1, data, n_tiles, sink);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int clamp_grid(int64_t want, int cap) {
    if (want < 1) want = 1;
    return (int)(want > cap ? cap : want);
}

int filter_grid(int64_t units, bool generic, bool any_i32, int grid_blocks, int narrow_row_bytes) {
    // tile kernel with an int32 column: 512 workgroups = 2 per CU (see k_filter_tile).  Without one (int8 / 2-byte strings only) a
    // tile is 1-2 KiB and more waves must be resident to keep as many bytes in flight: 6 work-groups per CU for one byte per row, 3
    // for two, 2 from three on (round 5, the in-lane instances, 100 M rows, two runs each on one device: I8 19.0 / 18.5 us at 1536,
    // 20.1 / 19.9 at 1024, 23.7 at 512; S2 33.7 / 34.2 at 768, 35.8 at 1536, 37.5 at 512; I8+I8 34.3 / 34.8 at 768, 40.8 at 512;
    // I8+S2 47.0 / 47.1 at 512, 51.1 at 1024, 55.7 at 1536.  Twice the tiles per iteration at half the work-groups measured the same
    // or worse everywhere).  The word-at-a-time kernel keeps 8 per CU.
    const int narrow = narrow_row_bytes <= 1 ? 1536 : (narrow_row_bytes == 2 ? 768 : 512);
    const int dflt = generic ? 2048 : (any_i32 ? 512 : narrow);
    const int cap = grid_blocks > 0 ? (grid_blocks > kMaxFilterGrid ? kMaxFilterGrid : grid_blocks) : dflt;
    return clamp_grid((units + kWavesPerBlock - 1) / kWavesPerBlock, cap);
}

// With ev0/ev1 set, hipExtLaunchKernelGGL stamps them with the kernel's own start and end, so the elapsed
// time is the kernel's duration (what rocprofv3 reports), not launch-to-launch.

#define IMM3_TILE_CASE(k0, k1, k2, T)                                                           \
    if (a.kinds[0] == k0 && a.kinds[1] == k1 && a.kinds[2] == k2) {                             \
        if (!a.bitmap && a.defer_lines) return false; /* (a deferring instance parks bitmap lines and stores them: it needs the bitmap) */ \
        else if (a.stage_rec && a.tile_rows) return false; /* (table queries do not stage) */            \
        else if (a.stage_rec && a.defer_lines) IMM3_LAUNCH_LDS((k_filter_tile<k0, k1, k2, T, false, true, true>), grid, kBlockThreads, \
                             (size_t)kWavesPerBlock * (size_t)a.defer_lines * kTileWords * sizeof(uint64_t), s, ev0, ev1, a); \
        else if (a.stage_rec) IMM3_LAUNCH((k_filter_tile<k0, k1, k2, T, false, false, true>), grid, kBlockThreads, s, ev0, ev1, a); \
        else if (a.tile_rows && a.defer_lines) IMM3_LAUNCH_LDS((k_filter_tile<k0, k1, k2, 1, true, true, false>), grid, kBlockThreads, \
                             (size_t)kWavesPerBlock * (size_t)a.defer_lines * (kTileWords + 1) * sizeof(uint64_t), s, ev0, ev1, a); \
        else if (a.tile_rows) IMM3_LAUNCH((k_filter_tile<k0, k1, k2, 1, true, false, false>), grid, kBlockThreads, s, ev0, ev1, a); \
        else if (a.defer_lines) IMM3_LAUNCH_LDS((k_filter_tile<k0, k1, k2, T, false, true, false>), grid, kBlockThreads,       \
                             (size_t)kWavesPerBlock * (size_t)a.defer_lines * kTileWords * sizeof(uint64_t), s, ev0, ev1, a); \
        else IMM3_LAUNCH((k_filter_tile<k0, k1, k2, T, false, false, false>), grid, kBlockThreads, s, ev0, ev1, a); \
        return true;                                                                            \
    }

// kinds must be sorted ascending with TK_NONE (= 3) last; at most one TK_S2 column per launch.
// T (tiles per wave iteration): 2 for the lone int8 column (1 KiB tiles), 1 elsewhere -- the pipelined loop keeps the next group's
// loads in flight under the current group's evaluation, and filter_grid() sizes the launch to the bytes per row.  (Round 5 measured
// T = 4 / 2 / 2 / 2 for I8, S2, I8+I8, I8+S2 at half the work-groups: the same or slower.)
#define IMM3_TILE_KINDS(X)                                                                          \
    X(TK_NONE, TK_NONE, TK_NONE, 1)                                                                 \
    X(TK_I32, TK_NONE, TK_NONE, 1) X(TK_I8, TK_NONE, TK_NONE, 2) X(TK_S2, TK_NONE, TK_NONE, 1) \
    X(TK_I32, TK_I32, TK_NONE, 1) X(TK_I32, TK_I8, TK_NONE, 1) X(TK_I8, TK_I8, TK_NONE, 1) \
    X(TK_I32, TK_S2, TK_NONE, 1) X(TK_I8, TK_S2, TK_NONE, 1) \
    X(TK_I32, TK_I32, TK_I32, 1) X(TK_I32, TK_I32, TK_I8, 1) X(TK_I32, TK_I8, TK_I8, 1)              \
    X(TK_I8, TK_I8, TK_I8, 2) X(TK_I32, TK_I32, TK_S2, 1) X(TK_I32, TK_I8, TK_S2, 1) X(TK_I8, TK_I8, TK_S2, 1)

bool launch_filter_tile(const TileArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    IMM3_TILE_KINDS(IMM3_TILE_CASE)
    return false;
}

// tiles per wave iteration of the instance launch_filter_tile picks for these kinds (0: no such instance)
int filter_tile_group(const int32_t *kinds) {
#define IMM3_TILE_T(k0, k1, k2, T) \
    if (kinds[0] == k0 && kinds[1] == k1 && kinds[2] == k2) return T;
    IMM3_TILE_KINDS(IMM3_TILE_T)
#undef IMM3_TILE_T
    return 0;
}

void launch_filter_generic(const FilterArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    IMM3_LAUNCH(k_filter_generic, grid, kBlockThreads, s, ev0, ev1, a);
}

void launch_total(const TotalArgs &a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    IMM3_LAUNCH(k_total, 1, 64, s, ev0, ev1, a);
}

// ---------------------------------------------------------------------------------------------
// k_limit_gather: ProjectOp with a small limit, behind a limit scan (run_select's chunks): the first `limit` survivors of the tiles
// that were scanned, in one launch instead of k_scan + k_gather (7 + 9 us for ten rows).  Work-group b takes a CONTIGUOUS piece of
// the scanned tiles (K chunks of 256 tiles, K from the scanned-tile word): it counts its survivors, publishes the count tagged with
// the run, and adds up the counts of the work-groups before it -- which were dispatched before it, so the wait cannot deadlock
// whatever is resident; it is bounded all the same (kLimitGatherMaxPolls: a device shared with a long-running kernel of another
// process can hold a lower work-group back) and a wait that runs out raises finish[kFinishLimitGaveUp] -- to know its first output row.  A work-group whose first row is already behind the limit leaves; the few
// that are not walk their tiles again (the lines are in the L2) and emit row numbers and column values row by row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kChunkTiles) void k_limit_gather(const LimitGatherArgs a) {
    __shared__ uint32_t s_wave[kChunkTiles / 64];
    __shared__ uint32_t s_chunk_total[kLimitGatherMaxChunks];
    __shared__ unsigned long long s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t scanned = (int64_t)a.finish[kFinishLimitTiles] < a.n_tiles ? (int64_t)a.finish[kFinishLimitTiles] : a.n_tiles;
    const unsigned long long tag = ((a.finish[kFinishEpoch] & 0x7FFFFFULL) << 1) | 1ULL; // (24 bits, never zero: the counts start out cleared)
    const int64_t n_chunks = (scanned + kChunkTiles - 1) / kChunkTiles;
    const int64_t K = (n_chunks + gridDim.x - 1) / gridDim.x; // (<= kLimitGatherMaxChunks: the host checks n_tiles against the grid)
    const int64_t chunk0 = (int64_t)blockIdx.x * K;
    // this work-group's tiles: counts per chunk (block-wide), and the sum
    auto tile_count = [&](int64_t tile) -> uint32_t {
        if (tile >= scanned) return 0u;
        const uint4 *p = (const uint4 *)(a.bitmap + tile * kTileWords);
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < kTileWords / 2; ++i) {
            const uint4 v = p[i];
            c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        }
        return c;
    };
    // inclusive scan of one value per thread over the work-group; returns the thread's exclusive prefix, total in `total`
    auto block_scan = [&](uint32_t c, uint32_t &total) -> uint32_t {
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        __syncthreads(); // (s_wave of the previous call has been read)
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        total = 0;
#pragma unroll
        for (int i = 0; i < kChunkTiles / 64; ++i) {
            if (i < wave) before += s_wave[i];
            total += s_wave[i];
        }
        return incl - c + before;
    };
    unsigned long long mine = 0;
    for (int64_t k = 0; k < K; ++k) {
        uint32_t total;
        (void)block_scan(tile_count((chunk0 + k) * kChunkTiles + t), total);
        if (t == 0) s_chunk_total[k] = total;
        mine += total;
    }
#ifdef IMM3_ABLATE
    const uint32_t poll_cap = a.max_polls ? a.max_polls : kLimitGatherMaxPolls;
    if (t == 0 && (int)blockIdx.x != a.fault_wg)
#else
    const uint32_t poll_cap = kLimitGatherMaxPolls;
    if (t == 0)
#endif
        __hip_atomic_store(a.wg_state + blockIdx.x, (tag << 40) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the survivors before this work-group's tiles: the counts of the work-groups before it (enough of them: once the sum has
    // reached the limit the rest does not matter)
    if (wave == 0) {
        unsigned long long base = 0;
        bool gave_up = false;
        for (int64_t b0 = 0; b0 < (int64_t)blockIdx.x && base < (unsigned long long)a.limit; b0 += 64) {
            const int64_t b = b0 + lane;
            unsigned long long v = 0;
            bool late = false;
            if (b < (int64_t)blockIdx.x) {
                // bounded, as the projection's look-back waits are (imm3_project.hip desc_wait): ~0.2 s of polls, then the rows
                // of this launch are given up and the host gathers them with k_scan + k_gather (settle_rows)
                for (uint32_t polls = 0;; ++polls) {
                    v = __hip_atomic_load(a.wg_state + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 40) == tag) break;
                    if (polls > poll_cap) { late = true; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                v = late ? 0ULL : v & ((1ULL << 40) - 1ULL);
            }
            if (__ballot(late)) { gave_up = true; break; } // (wave-uniform)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
            base += v;
        }
        if (lane == 0) {
            if (gave_up) a.finish[kFinishLimitGaveUp] = tag; // (tagged with the run: a stale word of an earlier run is not this one)
            s_base = gave_up ? ~0ULL : base;                 // (behind every limit: the work-group leaves below)
        }
    }
    __syncthreads();
    unsigned long long base = s_base;
    const unsigned long long stop = (unsigned long long)a.limit < a.cap_rows ? (unsigned long long)a.limit : a.cap_rows;
    if (base >= stop || mine == 0) return; // block-uniform
    for (int64_t k = 0; k < K && base < stop; ++k) {
        const int64_t tile = (chunk0 + k) * kChunkTiles + t;
        uint32_t total;
        const uint32_t cnt = tile_count(tile);
        unsigned long long out = base + block_scan(cnt, total);
        if (cnt && out < stop) {
            for (int w = 0; w < kTileWords && out < stop; ++w) {
                uint64_t word = a.bitmap[tile * kTileWords + w];
                while (word && out < stop) {
                    const int64_t row = tile * kTileRows + 64 * w + __builtin_ctzll(word);
                    word &= word - 1;
                    a.row_index[out] = (uint32_t)row;
                    for (int pj = 0; pj < a.n_proj; ++pj)
                        store_value_rt(a.proj[pj].dst, a.proj[pj].width, out, load_value_rt(a.proj[pj].src, a.proj[pj].width, row));
                    ++out;
                }
            }
        }
        base += total;
    }
}

void launch_limit_gather(const LimitGatherArgs &a, int grid_blocks, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    IMM3_LAUNCH(k_limit_gather, grid_blocks < 1 ? 1 : grid_blocks, kChunkTiles, s, ev0, ev1, a);
}

void launch_scan(const ScanArgs &a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const int grid = (int)((a.n_tiles + kChunkTiles - 1) / kChunkTiles);
    IMM3_LAUNCH(k_scan, grid < 1 ? 1 : grid, kChunkTiles, s, ev0, ev1, a);
}

#define IMM3_GATHER_CASE(n4, n2, n1)                                                             \
    if (c4 == n4 && c2 == n2 && c1 == n1) {                                                      \
        IMM3_LAUNCH((k_gather_plain<n4, n2, n1>), grid, kBlockThreads, s, ev0, ev1, g);          \
        return;                                                                                  \
    }

void launch_gather(const GatherArgs &a, int grid_blocks, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const int cap = grid_blocks > 0 ? grid_blocks : 1024; // 4 x 32 KiB LDS lists per CU
    const int64_t n_spans = (a.n_tiles + kSpanTiles - 1) / kSpanTiles;
    const int grid = clamp_grid(n_spans, cap);
    // the fast form: one uniform segment, <= 4 columns of 4 / 2 / 1 bytes, sorted by width (the order of the gathers is free)
    bool plain = !a.word_row_base && !a.tile_rows && a.n_proj >= 1 && a.n_proj <= 4;
    int c4 = 0, c2 = 0, c1 = 0;
    for (int j = 0; j < a.n_proj; ++j) {
        plain = plain && !a.proj[j].tile_ptrs && (a.proj[j].width == 1 || a.proj[j].width == 2 || a.proj[j].width == 4);
        c4 += a.proj[j].width == 4;
        c2 += a.proj[j].width == 2;
        c1 += a.proj[j].width == 1;
    }
    if (plain) {
        GatherArgs g = a;
        int k = 0;
        for (int wdt : {4, 2, 1})
            for (int j = 0; j < a.n_proj; ++j)
                if (a.proj[j].width == wdt) g.proj[k++] = a.proj[j];
        IMM3_GATHER_CASE(1, 0, 0) IMM3_GATHER_CASE(0, 1, 0) IMM3_GATHER_CASE(0, 0, 1)
        IMM3_GATHER_CASE(2, 0, 0) IMM3_GATHER_CASE(1, 1, 0) IMM3_GATHER_CASE(1, 0, 1) IMM3_GATHER_CASE(0, 2, 0) IMM3_GATHER_CASE(0, 1, 1) IMM3_GATHER_CASE(0, 0, 2)
        IMM3_GATHER_CASE(3, 0, 0) IMM3_GATHER_CASE(2, 1, 0) IMM3_GATHER_CASE(2, 0, 1) IMM3_GATHER_CASE(1, 2, 0) IMM3_GATHER_CASE(1, 1, 1) IMM3_GATHER_CASE(1, 0, 2)
        IMM3_GATHER_CASE(0, 3, 0) IMM3_GATHER_CASE(0, 2, 1) IMM3_GATHER_CASE(0, 1, 2) IMM3_GATHER_CASE(0, 0, 3)
        IMM3_GATHER_CASE(4, 0, 0) IMM3_GATHER_CASE(3, 1, 0) IMM3_GATHER_CASE(3, 0, 1) IMM3_GATHER_CASE(2, 2, 0) IMM3_GATHER_CASE(2, 1, 1) IMM3_GATHER_CASE(2, 0, 2)
        IMM3_GATHER_CASE(1, 3, 0) IMM3_GATHER_CASE(1, 2, 1) IMM3_GATHER_CASE(1, 1, 2) IMM3_GATHER_CASE(1, 0, 3)
        IMM3_GATHER_CASE(0, 4, 0) IMM3_GATHER_CASE(0, 3, 1) IMM3_GATHER_CASE(0, 2, 2) IMM3_GATHER_CASE(0, 1, 3) IMM3_GATHER_CASE(0, 0, 4)
    }
    IMM3_LAUNCH(k_gather, grid, kBlockThreads, s, ev0, ev1, a);
}

template <int R>
static void launch_emit_r(const EmitArgs &a, int n_gather, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    switch (n_gather) {
    case 0: IMM3_LAUNCH((k_emit<R, 0>), grid, kBlockThreads, s, ev0, ev1, a); break;
    case 1: IMM3_LAUNCH((k_emit<R, 1>), grid, kBlockThreads, s, ev0, ev1, a); break;
    case 2: IMM3_LAUNCH((k_emit<R, 2>), grid, kBlockThreads, s, ev0, ev1, a); break;
    case 3: IMM3_LAUNCH((k_emit<R, 3>), grid, kBlockThreads, s, ev0, ev1, a); break;
    default: IMM3_LAUNCH((k_emit<R, 4>), grid, kBlockThreads, s, ev0, ev1, a); break;
    }
}

// cols[0 .. n_gather) are gathered (rec_dword < 0), the rest come out of the record; n_gather <= kMaxEmitGather
void launch_emit(const EmitArgs &a, int n_gather, int grid_blocks, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const int64_t n_groups = (a.n_tiles + kEmitTiles - 1) / kEmitTiles;
    const int grid = clamp_grid(n_groups, grid_blocks > 0 ? grid_blocks : 2048);
    if (a.R == 1) launch_emit_r<1>(a, n_gather, grid, s, ev0, ev1);
    else if (a.R == 2) launch_emit_r<2>(a, n_gather, grid, s, ev0, ev1);
    else launch_emit_r<4>(a, n_gather, grid, s, ev0, ev1);
}

} // namespace imm3
