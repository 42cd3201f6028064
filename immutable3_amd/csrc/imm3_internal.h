// imm3_internal.h -- shared between the HIP kernels (imm3_kernels.hip) and the C-ABI (imm3_api.cpp).
// gfx950 (MI355X, CDNA4) only: wave64, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

// Ablation switches ("what does the kernel cost without its stores?") exist only in the tools' build of the library
// (make -C csrc ablate: -DIMM3_ABLATE -> lib/libimm3_ablate.so, loaded through IMM3_LIB_PATH).  In the shipped build
// IMM3_ABLATED is the constant false and the kernels carry none of it.
#ifdef IMM3_ABLATE
#define IMM3_ABLATED(args, value) ((args).ablate == (value))
#define IMM3_ABLATE_BIT(args, bit) (((args).ablate & (bit)) != 0) /* k_filter_project: ablate = a mask */
#else
#define IMM3_ABLATED(args, value) false
#define IMM3_ABLATE_BIT(args, bit) false
#endif

namespace imm3 {

// A wave owns one TILE = 1024 rows = 16 bitmap words = exactly one 128-byte line of the bitmap,
// so no two waves (let alone two XCDs, whose L2s are not coherent) ever write the same line.
constexpr int kTileRows = 1024;
constexpr int kTileWords = 16;
constexpr int kWavesPerBlock = 4;   // 256-thread workgroups
constexpr int kBlockThreads = 256;
constexpr int kChunkTiles = 256;    // tiles per offsets-scan chunk (one scan workgroup; 382 of them per 100 M rows: every CU has one)
constexpr int kSpanTiles = 16;      // tiles per gather workgroup (kChunkTiles % kSpanTiles == 0)
constexpr int kSpanWords = kSpanTiles * kTileWords;

constexpr int kMaxPredCols = 4;     // distinct predicate columns fused per launch (more -> extra AND pass)
constexpr int kMaxMatch = 8;        // IN-list values carried in kernel arguments
constexpr int kMaxProj = 8;         // projected columns per gather launch

enum ColKind : int32_t { KIND_I32 = 0, KIND_I8 = 1, KIND_STR = 2 };

// Canonical per-column predicate.  All SelectOp leaves on one column are folded on the host:
//   numeric: GT/LT/EQ conjunction -> one closed interval [lo, hi] (narrowing of Select.scala:65,73
//            already applied; empty intervals never reach the kernel),
//   string : Match IN-lists intersected; values whose length != width can never match and are dropped.
struct ColPred {
    const void *data;            // flat column in HBM (DENSE_* decode == little-endian reinterpretation)
    int32_t kind;                // ColKind
    int32_t width;               // bytes per value
    int32_t lo, hi;              // numeric closed interval (int8 values sign-extended)
    int32_t n_match;             // string: IN-list size
    int32_t match_in_args;       // 1: width <= 8 and n_match <= kMaxMatch -> values packed in match[]
    uint64_t match[kMaxMatch];   // value bytes packed little-endian (byte 0 = first character)
    const uint8_t *match_blob;   // otherwise: n_match * width bytes in device memory
};

// ---- tile kernel (compile-time column kinds) ----
enum TileKind : int32_t { TK_I32 = 0, TK_I8 = 1, TK_S2 = 2, TK_NONE = 3 };
constexpr int kMaxTileCols = 3;
constexpr int kDeferLines = 64;     // bitmap lines a wave parks in LDS between store bursts (32 KiB per work-group)
constexpr int kMaxTileMatch = 8;
constexpr int kMaxArenaSlots = 256; // tiles per wave of a staging launch (the wave's start table sits in LDS)

struct TileCol {
    const void *data;               // flat column in HBM, 16-byte aligned
    int32_t lo, hi;                 // TK_I32 / TK_I8: closed interval
    int32_t n_match;                // TK_S2: IN-list size (1..kMaxTileMatch)
    int32_t pad;
    uint32_t match[kMaxTileMatch];  // TK_S2: the two value bytes, little-endian
};

struct TileArgs {
    TileCol cols[kMaxTileCols];
    int32_t kinds[kMaxTileCols];    // sorted ascending, TK_NONE last (selects the template instance)
    int32_t and_existing;
    int32_t defer_lines;            // > 0: park the bitmap lines in (dynamic) LDS and store them in bursts (single segment only)
    int32_t ablate;                 // read only by the tools' build (IMM3_ABLATED below); 0 otherwise
    int64_t n_rows, n_words, n_tiles;
    uint64_t *bitmap;               // null: count-only run -- no bitmap line is stored (single-pass chains only: and_existing = 0, defer_lines = 0)
    uint32_t *block_partials;
    unsigned long long *finish;     // {total, n_emit, status, limit, tally, log, log index, log capacity}: the count is reduced in the kernel (block_partial_finish); null = k_total does
    void *stage_rec;                // survivor records (rec_layout(kinds).dwords dwords each): one arena of wave_cap records per wave, or null
    uint32_t *tile_start;           // [wave * max_slots + i]: where in its arena the wave's i-th staged tile starts
    int64_t wave_cap;               // records per arena
    int32_t max_slots;              // tiles a wave stages at most (<= kMaxArenaSlots)
    int32_t chunked;                // 1 (2: the run's first): this launch is one CHUNK of a limit scan (run_select, imm3_api.cpp): it adds to the running count
                                    // finish[kFinishLimitRows] / finish[kFinishLimitTiles], and does nothing at all once that count has reached the limit
    unsigned long long *stamps;     // diagnostics only: per work-group {start, end} of the 100 MHz device clock, or null
    // table queries (imm3_table): the tile table replaces cols[k].data / n_rows.  Tile t holds tile_rows[t] valid rows
    // (1024 except the last tile of each segment) starting at tile_ptrs[k][t] in column k.  Null for one segment.
    const uint32_t *tile_rows;
    const void *const *tile_ptrs[kMaxTileCols];
};

// Survivor record (k_filter_tile's STAGE instances -> k_emit): dword 0 carries the row's position in its tile in bits 0-9;
// the narrow predicate columns (int8: 8 bits, 2-byte string: 16 bits) are packed behind it from bit 16, a field that
// does not fit starting the next dword at bit 0; every int32 predicate column takes a dword of its own after those.
// 1, 2 or 4 dwords per record (3 is padded to 4: 16-byte LDS and global accesses).
struct RecField {
    int dword, shift, bits; // where column `col` sits
    int dwords;             // record size
};
constexpr RecField rec_layout(const int (&kinds)[kMaxTileCols], int col) {
    RecField f{0, 0, 0, 0};
    int d = 0, b = 16;
    for (int i = 0; i < kMaxTileCols; ++i) {
        const int w = kinds[i] == TK_I8 ? 8 : (kinds[i] == TK_S2 ? 16 : 0);
        if (!w) continue;
        if (b + w > 32) {
            ++d;
            b = 0;
        }
        if (i == col) {
            f.dword = d;
            f.shift = b;
            f.bits = w;
        }
        b += w;
    }
    int n = d + 1;
    for (int i = 0; i < kMaxTileCols; ++i) {
        if (kinds[i] != TK_I32) continue;
        if (i == col) {
            f.dword = n;
            f.shift = 0;
            f.bits = 32;
        }
        ++n;
    }
    f.dwords = n == 3 ? 4 : n;
    return f;
}

// Ring record of k_filter_project (imm3_project.hip): it never leaves the CU, so its layout is the streamers' to choose.  Dword 0 =
// the narrow predicate columns from bit 0, as long as they fit 16 bits | the row's position in its RANGE (tile in the range * 1024 +
// position in the tile) << 16; a narrow field that does not fit starts the next dword at bit 0; every int32 predicate column takes a
// dword of its own after those.  With the position on top, a record's first dword is ONE v_or3 in the streamer (value as loaded |
// a per-word constant register | the tile's index) and the row is ONE shift in the writer.  Same sizes as rec_layout: 1, 2 or 4 dwords.
constexpr RecField ring_layout(const int (&kinds)[kMaxTileCols], int col) {
    RecField f{0, 0, 0, 0};
    int d = 0, b = 0, room = 16;
    for (int i = 0; i < kMaxTileCols; ++i) {
        const int w = kinds[i] == TK_I8 ? 8 : (kinds[i] == TK_S2 ? 16 : 0);
        if (!w) continue;
        if (b + w > room) {
            ++d;
            b = 0;
            room = 32;
        }
        if (i == col) {
            f.dword = d;
            f.shift = b;
            f.bits = w;
        }
        b += w;
    }
    int n = d + 1;
    for (int i = 0; i < kMaxTileCols; ++i) {
        if (kinds[i] != TK_I32) continue;
        if (i == col) {
            f.dword = n;
            f.shift = 0;
            f.bits = 32;
        }
        ++n;
    }
    f.dwords = n == 3 ? 4 : n;
    return f;
}

// k_emit: ProjectOp from the staged records
constexpr int kEmitTiles = 32;      // tiles per emit work-group (kChunkTiles % kEmitTiles == 0).  32 beat 64 on C3 (34 -> 31 us) and on clustered survivors (2 % contiguous: 77 -> 40 us), lost on 50 % contiguous (167 -> 196 us); 16 lost on C4 (50 -> 60 us)
constexpr int kMaxEmitGather = 4;   // SELECT-list columns that are not predicate columns: gathered at the record's position
struct EmitCol {
    void *dst;                      // packed output, width bytes per emitted row
    const void *src;                // gathered column (flat), or null when the value comes out of the record
    int32_t width;                  // 1, 2 or 4
    int32_t rec_dword, rec_shift;   // staged column: where it sits in the record
    int32_t pad;
};
struct EmitArgs {
    const void *stage;              // records, R dwords each: one arena per wave of the filter launch that staged them
    const uint32_t *tile_start;     // [wave * max_slots + i]
    int64_t wave_cap, n_waves, main_tiles; // arena size in records; waves of the filter launch; tiles its main loop covered (the rest: leftovers)
    int32_t max_slots, T;           // T: tiles per wave iteration of that launch
    int32_t ablate, pad;            // read only by the tools' build (IMM3_ABLATED); 0 otherwise
    const uint32_t *tile_offsets;
    const uint32_t *chunk_sums;
    int64_t n_tiles;
    uint64_t cap_rows;
    uint32_t *row_index;
    EmitCol cols[kMaxProj];         // gathered columns first
    int32_t n_cols;
    int32_t R;
};
void launch_emit(const EmitArgs &a, int n_gather, int grid_blocks, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);

// ---- k_filter_project (imm3_project.hip): ScanOp -> SelectOp* -> ProjectOp of one uniform segment in ONE pass ----
// A work-group owns SPANS of kProjectStreamers * P consecutive tiles (each of its streaming waves P consecutive tiles);
// spans go to the work-groups round-robin.  The survivors' records of a wave's range wait in LDS; where they go in the output -- the
// number of survivors in all earlier tiles -- comes from a scan over per-span DESCRIPTORS, one ROUND of spans (one span per
// work-group) at a time: the round's last span to arrive scans the round's counts (imm3_project.hip, span_arrive):
//   desc[s] = epoch << 38 | flag << 36 | value     flag 1: value = survivors of span s (published when the span is done)
//                                                  flag 2: value = survivors of spans 0..s-1: the span's first output row
//                                                  flag 3: the run is being abandoned (a wait timed out)
// `epoch` (the low 26 bits of finish[kFinishEpoch], bumped by the launch's last work-group) tells this run's descriptors from
// earlier runs', so nothing is cleared between runs: a stale tag could only alias after 2^26 runs of the query in which its
// descriptor was never rewritten (round 3 carried 8 bits: 256 bumps -- every count of the query bumps the counter -- and a graph
// that keeps a larger P than the direct runs touches descriptors those never rewrite).  A segment has < 2^32 rows, so 36 value
// bits are plenty.  P is a launch argument: the host lowers it once it has seen how many rows survive (imm3_api.cpp,
// single_pass_adapt); kProjectMinP sizes the descriptor array.
constexpr int kProjectMinP = 1;
constexpr int kProjectMaxP = 64;        // tiles per wave per span (the tile's index in its range takes 6 bits of the record)
// (tools: -DIMM3_PROJECT_STREAMERS=12 -DIMM3_PROJECT_RING_KB=9 was tried -- 16 waves cap the kernel at 128 VGPRs: C3 137 us against 125)
#ifndef IMM3_PROJECT_STREAMERS
#define IMM3_PROJECT_STREAMERS 8
#endif
#ifndef IMM3_PROJECT_RING_KB
#define IMM3_PROJECT_RING_KB 14
#endif
constexpr int kProjectStreamers = IMM3_PROJECT_STREAMERS;  // waves of a work-group that stream tiles: a span is kProjectStreamers * P consecutive tiles
constexpr int kProjectWriters = 4;    // waves of a work-group that write the rows
constexpr int kProjectRingBytes = IMM3_PROJECT_RING_KB * 1024; // a streamer's LDS ring of survivor records
// finish[2], the query's status word: bit 0 malformed PFOR block (k_filter_pfor); single-pass projection: bit 1 = the run's ROWS
// are incomplete because a prefix never came (abandoned), bit 2 = ... because another launch of the kernel owned the device (busy).
// Bits 1-2 are valid for ONE run: bits 8..31 carry the low 24 bits of that run's epoch (status_flag_set / status_raise,
// imm3_project.hip), so a later launch of the query never mistakes an earlier run's flags for its own and the host never
// has to clear the word.  Whatever the flags say, a run's COUNT (finish[0], finish[1]) and BITMAP are exact: a work-group that gives up
// on the rows goes on streaming in count + bitmap mode.
constexpr int kFinishStatus = 2;
constexpr unsigned long long kStatusPfor = 1ULL, kStatusAbandoned = 2ULL, kStatusBusy = 4ULL;
constexpr int kStatusEpochShift = 8;
constexpr unsigned long long kStatusEpochMask = 0xFFFFFFULL;
constexpr int kFinishEpoch = 8;         // finish[8]: run counter of the query (descriptor epochs)
constexpr int kFinishDense = 9;         // finish[9]: single-pass projection: ranges of the last run that outgrew their LDS ring (finish[10]: the running sum)
// `limit` stops the scan (ProjectIterator.hasNext returns false at the limit and the bounded queue stalls the workers:
// Project.scala:73-80, Engine.scala:166,253-258): a projection with a limit runs its select pass as CHUNKS of growing size, each a
// launch that first looks at the rows selected so far and leaves at once when the limit has been reached.
constexpr int kFinishLimitRows = 11;    // finish[11]: rows selected by the chunks of this run so far
constexpr int kFinishLimitTiles = 12;   // finish[12]: tiles those chunks scanned (the offsets scan and the gather stop there)
constexpr int kFinishLimitGaveUp = 13;  // finish[13]: k_limit_gather's look-back ran into its poll cap in the run with this tag: gather with k_scan + k_gather
constexpr int kDescValueBits = 36, kDescFlagShift = 36, kDescEpochShift = 38;
constexpr unsigned long long kDescValueMask = (1ULL << kDescValueBits) - 1;
constexpr unsigned long long kDescEpochMask = (1ULL << (64 - kDescEpochShift)) - 1;
constexpr uint32_t kProjectMaxPolls = 1u << 17; // polls of a look-back wait (~0.1-0.2 s) before the run's rows are given up
struct ProjectTile {                // 32 bytes: one scalar load (s_load_dwordx8) per tile, issued a tile ahead of its use
    const void *p[kMaxTileCols];    // the tile's first value in each tile column of the launch (kinds order)
    uint32_t rows;                  // valid rows (kTileRows: a full tile)
    uint32_t pad;
};
struct ProjectArgs {
    TileCol cols[kMaxTileCols];
    int32_t kinds[kMaxTileCols];    // sorted ascending, TK_NONE last
    int32_t P;                      // tiles per wave per span
    int64_t n_rows, n_tiles, n_spans;
    int64_t n_rounds;               // ceil(n_spans / grid)
    uint64_t *bitmap;
    unsigned long long *finish;     // the query's {total, n_emit, status, limit, tally, log ..., epoch} block
    unsigned long long *desc;       // n_spans span descriptors
    unsigned long long *round_total; // n_rounds inclusive totals through the round ...
    uint32_t *round_ctr;            // ... and n_rounds arrival counters (zero between runs).  At fixed places whatever P is: the host changes P between runs
    uint8_t *trash;                 // one 64-byte line per writer wave: where a dense range's unwanted lanes store (keeps its loop free of branches)
    uint64_t cap_rows;              // capacity of the output arrays
    uint32_t *row_index;
    void *pred_dst[kMaxTileCols];   // where the values of predicate column k go (packed, its width per row), or null: not in the SELECT list
    EmitCol gather[kMaxEmitGather]; // the other SELECT-list columns (and second mentions of a predicate column): gathered at the row
    int32_t n_gather;
    int32_t ablate;
    unsigned long long *stamps;     // diagnostics only
    unsigned long long *device_lock; // one word per device: the ticket of the launch of this kernel that owns the device, or 0
    // fault injection (imm3_diag.h, imm3_ctx_inject_fault): read only by the tools' build (IMM3_ABLATE); the shipped kernel carries none of it
    uint32_t max_polls;             // polls before a look-back wait gives up (0: kProjectMaxPolls)
    int32_t fault_wg, fault_span;   // work-group fault_wg never announces its fault_span-th span (-1: none)
    int32_t pad2;
    // table queries (imm3_table; the TABLE instances, imm3_project_table.hip): one descriptor per tile of the table's virtual row
    // space -- where the tile starts in each of the launch's columns and how many of its 1024 rows exist (1024 except the last tile
    // of each segment) -- replaces cols[k].data / n_rows.  Null for one segment.
    const ProjectTile *tile_desc;
};
// false: no instance for these kinds.  grid <= project_max_grid(): every work-group must be resident (they wait on each other)
bool launch_filter_project(const ProjectArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
bool launch_filter_project_table(const ProjectArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1); // a.tile_desc set
// the per-query tile descriptors of a table query, from the table's per-column tile pointers (null pointer: no such column)
void launch_project_tile_desc(const uint32_t *tile_rows, const void *const *p0, const void *const *p1, const void *const *p2, ProjectTile *out, int64_t n_tiles, hipStream_t s);
int project_max_grid(const int32_t *kinds, int P);   // resident work-groups of the instance on the current device (0: none)
int project_rec_dwords(const int32_t *kinds);

struct FilterArgs {
    ColPred cols[kMaxPredCols];
    int32_t ncols;
    int32_t and_existing;        // 1: AND into the bitmap already in memory (second pass for > kMaxPredCols columns)
    int64_t n_rows;              // uniform layout: rows of the segment
    int64_t n_words;             // bitmap words (batch-major)
    int64_t n_tiles;             // ceil(n_words / 16)
    uint64_t *bitmap;
    uint32_t *block_partials;    // selected rows per workgroup of this launch (no same-address atomics:
                                 // ~12 ns each, serialised -- 4096 of them cost 40 us on MI355X)
    // ragged layout only: per bitmap word, first row and number of valid rows (0..64)
    const uint32_t *word_row_base;
    const uint8_t *word_nvalid;
};

struct TotalArgs {               // k_total: sum of the filter launch's per-workgroup partials
    const uint32_t *block_partials;
    int32_t n_partials;
    int32_t pad;
    unsigned long long *total;   // selected rows of the segment
    unsigned long long *n_emit;  // rows ProjectOp emits = limit > 0 ? min(total, limit) : total
    int64_t limit;
};

struct ScanArgs {
    const uint64_t *bitmap;      // allocated in whole tiles, zero past n_words
    uint32_t *tile_offsets;      // exclusive prefix of the per-tile survivor counts WITHIN its chunk
    uint32_t *chunk_sums;        // selected rows per chunk of kChunkTiles tiles
    int64_t n_tiles;
    unsigned long long *finish;  // the query's {total, n_emit, status, limit, tally, log ...} block when THIS kernel publishes the count, else null
    const unsigned long long *scanned_tiles; // a limit scan stopped early: tiles [0, *scanned_tiles) hold this run's bitmap lines, the rest count as empty; null: all
    // A records run that stored NO bitmap (round 5): a tile's survivors = the length of its piece of its wave's arena, read from the
    // staging launch's start table (one more entry per wave holds the arena's end) through the launch geometry, as k_emit finds it.
    const uint32_t *rec_tile_start;          // null: the counts come from the bitmap
    int64_t rec_n_waves, rec_main_tiles;
    int32_t rec_max_slots, rec_T;
};

struct ProjCol {
    const void *src;             // flat column
    void *dst;                   // packed output, width bytes per emitted row
    const void *staged;          // survivors' values staged per tile by the filter kernel (width 4 or 1), or null
    const void *const *tile_ptrs; // table queries: per-tile pointer into this column (src unused), else null
    int32_t width;
    int32_t pad;
};

struct GatherArgs {
    const uint64_t *bitmap;
    const uint32_t *tile_offsets;
    const uint32_t *chunk_sums;
    int64_t n_tiles;
    int64_t n_words;
    int64_t limit;               // <= 0: unlimited
    uint64_t cap_rows;           // capacity of the output buffers
    uint32_t *row_index;         // segment-global row of each emitted row
    ProjCol proj[kMaxProj];
    int32_t n_proj;
    int32_t pad;
    int64_t n_staged_tiles;        // tiles [0, n) have staged values (the full tiles)
    const uint32_t *word_row_base; // ragged layout, else null
    const uint32_t *tile_rows;     // table queries: valid rows per tile (staged iff 1024), else null
    const unsigned long long *scanned_tiles; // a limit scan stopped early: only spans below *scanned_tiles are walked; null: all
};

// k_limit_gather: the offsets scan and the gather of a SMALL `limit` in one launch, over the tiles a limit scan got to.
constexpr int kLimitGatherMaxChunks = 64;   // chunks of 256 tiles per work-group at most (the launch has one work-group per CU)
constexpr int64_t kLimitGatherMaxRows = 4096;
constexpr uint32_t kLimitGatherMaxPolls = 1u << 18; // look-back polls (with s_sleep 4, ~0.5 us each) before the launch gives its rows up
struct LimitGatherArgs {
    const uint64_t *bitmap;
    unsigned long long *finish;        // [kFinishLimitTiles] = tiles scanned, [kFinishEpoch] = the run's tag; [kFinishLimitGaveUp] is written
    unsigned long long *wg_state;      // [grid]: (tag << 40) | survivors of the work-group's tiles
    int64_t n_tiles;
    int64_t limit;
    uint64_t cap_rows;
    uint32_t *row_index;
    ProjCol proj[kMaxProj];
    int32_t n_proj;
    int32_t fault_wg;                  // tools' build (-DIMM3_ABLATE, imm3_ctx_inject_fault): this work-group never publishes its count (-1: none)
    uint32_t max_polls, pad;           // tools' build: the poll cap (0: kLimitGatherMaxPolls)
};

// launchers (imm3_kernels.hip)
// ---- group-by aggregation (imm3_agg.hip) ----
constexpr int kMaxGroupCols = 4;
constexpr int kMaxAggs = 4;
enum AggKind : int32_t { AGG_COUNT = 0, AGG_MIN = 1, AGG_MAX = 2 };
// The kernel forms of the aggregation, fastest first (imm3_agg.hip).  A query starts at AGG_FORM_LANES; a form whose
// per-work-group table cannot hold the query's keys raises the overflow word (3: a lanes form, 2: direct / tile) and the
// host aggregates again from the next form, remembering it in the query handle.
enum AggForm : int32_t { AGG_FORM_LANES = 0, AGG_FORM_LANES_WIDE = 1, AGG_FORM_DIRECT = 2, AGG_FORM_TILE = 3, AGG_FORM_GENERAL = 4 };
constexpr int kMaxDevices = 64; // per-device launch state (dynamic-LDS limits raised)

struct GroupCol {
    const void *data;
    const void *const *tile_ptrs; // table queries: per-tile pointer, else null
    int32_t width;
    int32_t shift;               // byte position of this column inside the u64 group key
};

struct AggCol {
    const void *data;
    const void *const *tile_ptrs; // table queries: per-tile pointer, else null
    int32_t width;
    int32_t kind;                // AggKind
    int32_t is_str;              // AGG_MAX over a string column: values compared as big-endian-packed bytes
    int32_t pad;
};

// SelectOp fused into the aggregation (k_group_agg_lanes' FUSED instances): ProjectAggIterator walks the selected positions of the
// batch it is handed (ProjectAggregate.scala:158-177); with the predicates evaluated on the rows the aggregation kernel already holds
// there is no bitmap to store and to read back, and a column that is both predicate and aggregate is read once.
constexpr int kMaxAggPreds = 2;
struct AggPred {
    const void *data;            // flat column (int8 or int32), or null: the predicate is on the value aggregate's own column (share = 1)
    int32_t width;               // 1 or 4
    int32_t lo, hi;              // closed interval (int8 values sign-extended)
    int32_t share;               // 1: the rows are the first value aggregate's, already in registers
};
struct AggArgs {
    const uint64_t *bitmap;
    int64_t n_words, n_tiles, n_rows;
    const uint32_t *word_row_base; // ragged layout, else null
    GroupCol groups[kMaxGroupCols];
    AggCol aggs[kMaxAggs];
    int32_t n_group, n_agg;
    int32_t first_form;          // AggForm: where the chain of kernel forms starts for this query
    int32_t ablate;              // read only by the tools' build (IMM3_ABLATED); 0 otherwise
    AggPred fused[kMaxAggPreds]; // n_fused > 0: the select chain, evaluated by the aggregation kernel itself (no bitmap is read)
    int32_t n_fused;
    int32_t fused_all;           // 1: no predicate at all -- every row of the segment is selected (n_fused == 0, no bitmap either)
    // global open-addressing table: mask + 1 slots, plus one for the all-ones key
    unsigned long long *keys;
    uint32_t *first;             // first (lowest) selected row of the group
    unsigned long long *counts;
    long long *vals;             // kMaxAggs per slot
    uint32_t mask;
    uint32_t out_cap;
    uint32_t *n_groups;
    uint32_t *overflow;
    // dense output of k_group_collect
    unsigned long long *out_keys;
    uint32_t *out_first;
    unsigned long long *out_counts;
    long long *out_vals;
};

void launch_group_agg(const AggArgs &a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
bool group_agg_fuses_select(const AggArgs &a); // would launch_group_agg run a form that evaluates a.fused[] itself (no select launch needed)?

// ---- cross-segment / cross-GPU merge of group tables (ProjectAggregateQueueOp, ProjectAggregateQueue.scala:9-55) ----
// Keys of <= 2 bytes index a DIRECT table (256 or 65 536 slots; one slot for no group column): every query's dense group
// list is scattered into it with 64-bit atomics (counts add, first = min of segment << 32 | row, values max / min), and the
// tables of the ranks then meet in element-wise all-reduces (imm3_comm.cpp).
struct MergeArgs {
    const unsigned long long *keys;   // one query's dense groups (k_group_collect's output)
    const uint32_t *first;
    const unsigned long long *counts;
    const long long *vals;            // kMaxAggs per group
    uint32_t n_groups;
    uint32_t slots;                   // K
    unsigned long long seg_hi;        // the query's segment index << 32
    unsigned long long *t_counts;     // [K]
    unsigned long long *t_first;      // [K], ~0 = empty
    long long *t_vals;                // [kMaxAggs][K]
    int32_t n_agg;
    int32_t kinds[kMaxAggs];          // AggKind
    int32_t is_str[kMaxAggs];         // MAX over strings: unsigned compare of the big-endian-packed bytes
    // keys wider than 2 bytes: the table is a hash table -- t_keys[slots] (open addressing, linear probing; slots = mask + 2, the
    // last slot is the one key that equals the empty marker), sized >= 2 x the entries that can arrive: it cannot fill up
    unsigned long long *t_keys;       // null: direct-indexed (slot = key)
    uint32_t mask;
    // packed lists (kMergeListWords u64 words per entry: key, first, count, kMaxAggs values; count 0 = padding): what the ranks
    // exchange with ncclAllGather and what the collect kernel writes
    const unsigned long long *list;   // k_merge_insert_list: n_groups entries
    unsigned long long *out_list;     // k_merge_collect_list: occupied slots, in no particular order ...
    unsigned long long *out_n;        // ... and how many (starts at 0)
    uint32_t out_cap;
};
constexpr int kMergeListWords = 3 + kMaxAggs;
void launch_merge_init(const MergeArgs &a, hipStream_t s);
void launch_merge_scatter(const MergeArgs &a, hipStream_t s);
void launch_merge_insert_list(const MergeArgs &a, hipStream_t s);
void launch_merge_collect_list(const MergeArgs &a, hipStream_t s);
void launch_group_collect(const AggArgs &a, hipStream_t s);

constexpr int kSubTallies = 32;                       // in-kernel count reduce: sub-tallies (finish_add, imm3_device.h)
constexpr int kFinishWords = 16 + kSubTallies * 16;   // u64 words of a query's `finish` block: header (9 used, padded to a 128-byte line) + one line per sub-tally
constexpr int kMaxFilterGrid = 4096; // capacity of block_partials
int filter_grid(int64_t units, bool generic, bool any_i32, int grid_blocks, int narrow_row_bytes = 0); // narrow_row_bytes: bytes per row of a launch without an int32 column
// ev0/ev1: optional events stamped with the kernel's own start/end (hipExtLaunchKernelGGL), else null
// ---- PFOR_INT blocks (imm3_codec.hip) ----
struct PforArgs {
    const uint8_t *data;          // the column's .dat bytes in HBM (blocks start on 4-byte boundaries)
    const uint32_t *block_off;    // n_blocks + 1 byte offsets
    const uint32_t *row_base;     // decode: n_blocks + 1 first rows (prefix sum of the blocks' value counts)
    int64_t n_blocks;
    int32_t lo, hi;               // filter: closed interval
    int32_t and_existing;
    int32_t pad;
    int64_t n_rows, n_words, n_tiles;
    uint64_t *bitmap;
    uint32_t *block_partials;
    int32_t *out;                 // decode: dense int32 column
    uint32_t *status;             // bit 0 set when a block is malformed
};
void launch_pfor_counts(const uint8_t *data, const uint32_t *block_off, int64_t n_blocks, int32_t *counts, hipStream_t s);
void launch_filter_pfor(const PforArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_pfor_decode(const PforArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);

// ---- snappy-coded blocks (imm3_snappy.hip) ----
struct SnappyArgs {
    const uint8_t *data;          // the column's .dat bytes in HBM (blocks start on any byte)
    const uint32_t *block_off;    // n_blocks + 1 byte offsets
    const uint32_t *row_base;     // n_blocks + 1 first rows
    const uint32_t *xpow8;        // 32769 entries: x^(8 n) mod the CRC-32C polynomial, reflected
    int64_t n_blocks;
    int32_t width;                // bytes per value of the decoded column
    int32_t in_cap, out_cap;      // LDS bytes for the staged block / one chunk
    int32_t pad;
    uint8_t *out;                 // dense column
    uint32_t *status;             // bit 0 set when a block is malformed or a checksum does not match
};
void launch_snappy_sizes(const uint8_t *data, const uint32_t *block_off, int64_t n_blocks, uint32_t *sizes, uint32_t *max_chunk, hipStream_t s);
void launch_snappy_decode(const SnappyArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);

// k_sum_counts: the selected-row counts of up to kMaxSumCounts queries (device words) -> one device word; the send buffer
// of the RCCL count all-reduce (imm3_comm.cpp)
constexpr int kMaxSumCounts = 32;
struct SumCountsArgs {
    const unsigned long long *src[kMaxSumCounts];
    int32_t n;
    int32_t accumulate;          // 1: add to *dst instead of overwriting it (more than kMaxSumCounts queries)
    unsigned long long *dst;
};
void launch_sum_counts(const SumCountsArgs &a, hipStream_t s);

bool launch_filter_tile(const TileArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
int filter_tile_group(const int32_t *kinds); // tiles per wave iteration of the instance for these kinds (0: none)
void launch_filter_generic(const FilterArgs &a, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_total(const TotalArgs &a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_read_stream(const int32_t *data, int64_t n_tiles, int32_t *sink, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_scan(const ScanArgs &a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_gather(const GatherArgs &a, int grid_blocks, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
void launch_limit_gather(const LimitGatherArgs &a, int grid_blocks, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);

} // namespace imm3

#ifdef __HIPCC__
#include <hip/hip_ext.h>
// Kernel launch with optional start / stop events stamped by the launch itself (hipExtLaunchKernelGGL); the plain launch
// when there are none: that is the one a stream capture (imm3_ctx_capture_begin) records.
#define IMM3_LAUNCH_LDS(kern, grid, block, lds, s, ev0, ev1, ...)                                                \
    do {                                                                                                         \
        if (ev0 || ev1) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, ev0, ev1, 0, __VA_ARGS__);  \
        else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, __VA_ARGS__);                             \
    } while (0)
#define IMM3_LAUNCH(kern, grid, block, s, ev0, ev1, ...) IMM3_LAUNCH_LDS(kern, grid, block, 0, s, ev0, ev1, __VA_ARGS__)
#endif

