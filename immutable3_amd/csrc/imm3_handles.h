// imm3_handles.h -- the handle structs behind include/imm3.h, shared by the translation units of the C ABI
// (imm3_api.cpp: planning + launches; imm3_comm.cpp: the RCCL count reduce).  Private to csrc/.
#pragma once

#include "../../include/imm3.h"
#include "imm3_internal.h"
#include "imm3_plan.h"
#include "imm3_sync.h"

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
namespace imm3 {
int fail(int code, const std::string &msg); // sets the calling thread's imm3_last_error() text (imm3_api.cpp)
}
using imm3::fail;

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(IMM3_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
struct TimingRecord {
    int32_t kernel_id;
    hipEvent_t start, stop;
};

// Lifetimes.  Every handle is reference counted: the caller's handle holds one reference, and every handle created
// FROM it holds another (segment / table / query / comm -> context; table -> its segments; query -> its segment or
// table).  imm3_*_destroy marks the handle closed and drops the caller's reference; the memory behind it goes when
// the last dependant is destroyed, so handles may be destroyed in ANY order (a JVM finalizer or a Python __del__ run
// by the garbage collector gives no order).  Work submitted through a handle whose context has been destroyed fails
// with IMM3_ERR_STATE; destroying the same handle twice while dependants keep it alive does too.
struct imm3_graph;
struct HipBlockBackend {
    static int alloc(void **p, size_t bytes) { return (int)hipMalloc(p, bytes); }
    static void free(void *p) { (void)hipFree(p); }
};
// Threading: see imm3_sync.h.  A context may be used by any number of threads at once; `gate` serialises a graph capture
// against the other threads' calls, `mu` guards the small mutable state below, the buffer pool has its own lock.
struct imm3_ctx {
    std::atomic<int> refs{1};
    std::atomic<bool> closed{false}; // imm3_ctx_destroy has run: streams, events and the buffer pool are gone
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    imm3::CaptureGate gate;
    std::mutex mu;                  // guards aux / copy creation, pinned, the timing and stamp records, graphs, d_xpow8
    hipStream_t aux = nullptr;      // count reduce of select-only runs: overlaps the next scan on `stream`
    hipStream_t copy = nullptr;     // host -> HBM staging of segments: never on the query stream, so staging overlaps queries
    std::map<void *, int> pinned;   // host ranges pinned in place for asynchronous staging, by start address, with a use count
    std::atomic<int> filter_variant{0};
    std::atomic<int> grid_blocks{0};
    std::atomic<bool> timing{false};
    std::atomic<uint32_t> timing_mask{0xFFFFFFFFu};
    std::vector<TimingRecord> pool; // pre-created event pairs (mu)
    size_t used = 0;                // (mu)
    imm3::BlockPool<HipBlockBackend> blocks; // per-query device buffers, reused in stream order
    // device-clock stamps (diagnostics): slot i = kMaxFilterGrid {start, end} pairs for the i-th tile launch (mu)
    unsigned long long *d_stamps = nullptr;
    int32_t stamp_slots = 0, stamp_used = 0;
    std::vector<int32_t> stamp_grids;
    uint32_t *d_xpow8 = nullptr;    // snappy CRC-32C check: x^(8 n) mod P for n = 0 .. 32768 (mu)
    imm3_graph *capture = nullptr;  // open stream capture (imm3_ctx_capture_begin .. _end), else null; owned by the capturing thread (gate)
    std::vector<imm3_graph *> graphs; // graphs recorded on this context that have not been destroyed yet (mu)
    // fault injection into k_filter_project (imm3_ctx_inject_fault, imm3_diag.h): read by the tools' build of the kernel only
    std::atomic<int> comms_attached{0}; // communicators created on this context and not yet destroyed: one-launch plans leave CUs for their kernels (imm3_api.cpp: single_pass_run_grid)
    std::atomic<int> fault_wg{-1}, fault_span{-1};
    std::atomic<uint32_t> fault_max_polls{0};
};

// What a run leaves in the query handle for the getters: which outputs exist and how they have to be settled.  A graph keeps the
// state each recorded query had after its (last) recorded run and puts it back at every imm3_graph_launch: a replay IS that run
// again, whatever direct runs, fallbacks or getters did to the handle in between.
struct QueryRunState {
    bool ran_select = false, ran_project = false, bitmap_valid = false, ran_single_pass = false, stage_written = false;
    bool count_pending_scan = false, has_pfor_pass = false, ran_agg = false, offsets_valid = false, select_partial = false;
    bool bitmap_lazy = false, agg_select_skipped = false;
};

// A recorded sequence of query runs (hipGraph): launching it enqueues every kernel of those runs with one call.
struct imm3_graph {
    imm3_ctx *ctx = nullptr;              // retained
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<imm3_query *> queries;    // the queries whose runs were recorded (not retained: destroying one makes the graph stale)
    std::vector<QueryRunState> states;    // per query: its state after the recorded run
    bool stale = false;
};

static inline bool is_snappy(int32_t c) { return c == IMM3_SNAPPY_INT || c == IMM3_SNAPPY_TINYINT || c == IMM3_SNAPPY_STRING; }
static inline bool is_compressed(int32_t c) { return c == IMM3_PFOR_INT || is_snappy(c); }
// the DENSE_* codec whose decoded vectors a column's values are (type dispatch of ScanOp / SelectOp / ProjectAggOp)
static inline int32_t value_codec(int32_t c) {
    if (c == IMM3_PFOR_INT || c == IMM3_SNAPPY_INT) return IMM3_DENSE_INT;
    if (c == IMM3_SNAPPY_TINYINT) return IMM3_DENSE_TINYINT;
    if (c == IMM3_SNAPPY_STRING) return IMM3_DENSE_STRING;
    return c;
}

struct SegCol {
    int32_t codec = 0, width = 0;
    int32_t vcodec = 0;                // value_codec(codec)
    uint8_t *d_data = nullptr;
    bool owned = false;
    uint64_t bytes = 0;
    std::vector<int32_t> offsets;
    // PFOR_INT (imm3_codec.hip) / snappy (imm3_snappy.hip): the blocks stay compressed in d_data
    std::vector<int32_t> block_rows;   // the rows each block declares (PFOR: its count word; snappy: uncompressed bytes / width)
    int32_t in_cap = 0, out_cap = 0;   // snappy: LDS bytes k_snappy_decode needs for the largest block / chunk
    int64_t rows = 0;
    bool tile_aligned = false;         // every block but the last holds exactly 1024 rows: block k == bitmap tile k
    uint32_t *d_block_off = nullptr;   // n_blocks + 1 byte offsets
    uint32_t *d_row_base = nullptr;    // n_blocks + 1 first rows
    uint8_t *d_dense = nullptr;        // decoded int32 column, made on first need (Project, aggregation, ragged, table)
};

// Batches of one segment as ScanOp yields them (Scan.scala:55,72; Segment.scala:159-168): block k of the FIRST used column holds
// size[k] rows, its BitSet starts at word word_off[k] of the segment's bitmap.  The same for every query whose first used column is
// the same, so it is computed once per (segment, first column) and shared (a 100 M-row segment has 97 657 batches: three vectors
// of that length per query were most of a query's creation time).  oid = k * table.blockSize is the caller's parameter: not stored.
struct SegLayout {
    std::vector<int32_t> size;
    std::vector<int64_t> word_off; // within the segment's own bitmap
    int64_t rows = 0, words = 0;
    bool ragged = false;
};

struct imm3_segment {
    std::atomic<int> refs{1};
    std::atomic<bool> closed{false};
    imm3_ctx *ctx = nullptr;
    std::vector<SegCol> cols;
    uint64_t device_bytes = 0;
    std::vector<void *> registered;    // host ranges pinned in place for an asynchronous create (unpinned by imm3_segment_wait)
    hipEvent_t ready = nullptr;        // recorded on the context's copy stream behind the last column's copy
    std::atomic<bool> ready_pending{false}; // the copies may still be in flight: consumers make their stream wait for `ready`
    std::mutex decode_mu;              // guards the lazy d_dense of PFOR_INT columns
    std::mutex layout_mu;              // guards the two caches below
    std::map<int32_t, std::shared_ptr<const SegLayout>> layouts; // by first used column
    std::map<std::pair<int32_t, int32_t>, bool> same_blocks;     // (first column, other column) -> holds the same rows in the same blocks (checked once)
    // the selectivity sample of query creation (single_pass_sample): a tile table over 8 evenly spaced chunks of 64 full tiles, one
    // pointer array per column (made on first need), so that the sample is ONE launch of the table instance of the scan+select kernel
    std::map<int32_t, void **> d_sample_ptrs; // by column
    uint32_t *d_sample_rows = nullptr;        // kSampleTiles x 1024
    int64_t sample_full_tiles = -1;           // the full tiles the table was laid out for
};

// the flat, fixed-width form of a column (what every kernel but k_filter_pfor reads)
static inline const uint8_t *col_flat(const SegCol &sc) { return is_compressed(sc.codec) ? sc.d_dense : sc.d_data; }

struct imm3_table { // all segments of one table as one scan unit: the tile table
    std::atomic<int> refs{1};
    std::atomic<bool> closed{false};
    imm3_ctx *ctx = nullptr;
    std::vector<const imm3_segment *> segs;
    std::vector<int64_t> seg_rows;     // rows per segment
    std::vector<int64_t> tile_start;   // n_segs + 1: first (virtual) tile of each segment
    int64_t n_tiles = 0, n_rows = 0;
    uint32_t *d_tile_rows = nullptr;   // valid rows per tile
    std::vector<void **> d_tile_ptrs;  // per column: device array of per-tile pointers
    // the sample a query's plan is made on (single_pass_sample, imm3_api.cpp): eight chunks of 64 tiles spread over the table
    uint32_t *d_sample_rows = nullptr;     // valid rows of the sampled tiles (null: the table is too small to sample)
    std::vector<void **> d_sample_ptrs;    // per column: the sampled tiles' pointers
    // batches of all segments (every column shares one block layout: checked at creation), built once: a query over 98
    // README-style segments otherwise spends ~0.4 ms of host time re-deriving them
    std::vector<int32_t> batch_size, batch_k; // rows; index of the batch within its segment (oid = k * table.blockSize)
    std::vector<int64_t> batch_word_off;
    std::vector<int32_t> seg_first_batch;     // n_segs + 1
    std::vector<int64_t> seg_first_word;      // n_segs + 1
};

struct FoldedPred { // all SelectOp leaves on one segment column, folded
    int32_t seg_col = 0;
    int32_t kind = 0, width = 0;
    int64_t lo = 0, hi = 0;                // numeric closed interval
    std::vector<std::string> match;        // string: surviving IN-list values (each exactly width bytes)
    uint8_t *d_blob = nullptr;             // device copy when it does not fit the kernel arguments
    bool pfor = false;                     // PFOR_INT column evaluated on its compressed blocks (k_filter_pfor)
};

struct imm3_query {
    imm3_ctx *ctx = nullptr;
    const imm3_segment *seg = nullptr;   // the segment (table queries: the first one, for the schema)
    const imm3_table *table = nullptr;   // table query: columns are addressed through the tile table
    int32_t table_block_size = 0;   // table.blockSize as given at creation (oid of a batch = index in its segment * blockSize)
    std::vector<int32_t> used;     // segment column index of each used column
    std::vector<int32_t> proj;     // index into `used`
    int64_t limit = 0;
    // layout (Scan.scala:55-60): shared with every query of the segment that has the same first used column (null: table query)
    std::shared_ptr<const SegLayout> layout;
    int64_t n_rows = 0, n_words = 0, n_tiles = 0, n_chunks = 0;
    bool ragged = false;
    bool always_false = false;
    std::vector<FoldedPred> preds;
    // device buffers
    uint64_t *d_bitmap = nullptr;
    uint32_t *d_tile_offsets = nullptr, *d_chunk_sums = nullptr, *d_block_partials = nullptr;
    unsigned long long *d_limit_state = nullptr; // k_limit_gather: per-work-group survivor counts, tagged with the run (small limits only)
    unsigned long long *d_total = nullptr, *d_n_emit = nullptr; // adjacent: d_n_emit = d_total + 1; d_total + 2 = status word
    bool has_pfor_pass = false;   // a k_filter_pfor pass may flag malformed blocks in the status word
    std::vector<unsigned long long> h_init; // host image of the whole finish block at creation (copied asynchronously: lives with the query)
    std::vector<uint32_t> h_word_row_base;  // ragged layouts: host images of the per-word row map (same reason)
    std::vector<uint8_t> h_word_nvalid;
    uint32_t *d_word_row_base = nullptr;
    uint8_t *d_word_nvalid = nullptr;
    uint32_t *d_row_index = nullptr;
    std::vector<uint8_t *> d_proj;
    uint64_t cap_rows = 0;
    bool reserved = false;
    bool ran_select = false, ran_project = false;
    bool bitmap_valid = false;     // the last run stored the selection bitmap (a count-only run does not)
    uint32_t run_syncs = 0;        // times imm3_query_run had to wait for the device (diagnostics: imm3_query_plan)
    // group-by aggregation
    bool is_agg = false;
    std::vector<int32_t> group_cols;           // index into `used`
    std::vector<imm3_aggregate> aggs;
    uint32_t agg_mask = 0;
    unsigned long long *d_akeys = nullptr, *d_acounts = nullptr, *d_okeys = nullptr, *d_ocounts = nullptr;
    uint32_t *d_afirst = nullptr, *d_ofirst = nullptr, *d_ameta = nullptr; // d_ameta: {n_groups, overflow}
    long long *d_avals = nullptr, *d_ovals = nullptr;
    uint32_t out_cap = 0;
    bool ran_agg = false;
    bool agg_fusable = false;          // the select chain is closed intervals over <= 2 dense int8 / int32 columns of one uniform segment (or empty): the aggregation kernel can evaluate it itself
    bool limit_gather_ran = false;     // the last projection was k_limit_gather's one launch: settle_rows looks at its give-up tag
    uint64_t limit_gather_gave_up = 0; // ... and how often it had to gather the rows again with k_scan + k_gather
    bool agg_select_skipped = false;   // the last run fused the select into the aggregation launch: bitmap and count do not exist until a getter asks (settle_agg_select)
    int32_t agg_first_form = imm3::AGG_FORM_LANES; // first kernel form to try: raised past the forms this query's keys overflowed
    // select-only runs: the count reduce goes to ctx->aux, fenced by these events
    hipEvent_t ev_filter_done = nullptr, ev_total_done = nullptr;
    bool total_on_aux = false;
    // survivor records (k_filter_tile STAGE instances -> k_emit): an unlimited projection over one uniform segment whose
    // select chain is ONE tile launch
    uint8_t *d_stage_rec = nullptr;
    int32_t stage_kinds[imm3::kMaxTileCols] = {imm3::TK_NONE, imm3::TK_NONE, imm3::TK_NONE}; // of the staged launch, in its column order
    int32_t stage_seg_col[imm3::kMaxTileCols] = {-1, -1, -1};                                 // segment column of each
    // the staging launch's geometry (fixed at creation: the arena layout depends on it)
    uint32_t *d_tile_start = nullptr;
    int32_t stage_grid = 0, stage_T = 1, stage_max_slots = 0;
    int64_t stage_wave_cap = 0, stage_main_tiles = 0;
    bool stage_written = false;   // the last select run filled the records
    bool bitmap_lazy = false;     // ... and stored NO bitmap (a records run of imm3_query_run: the records carry the positions, the offsets scan counts them): imm3_query_bitmap runs the select chain then
    bool force_plain_select = false; // the next run_select stages nothing (settle_lazy_bitmap)
    // single-pass projection (k_filter_project, imm3_project.hip): planned at creation for the same queries as the records
    bool single_pass = false;
    int32_t sp_P = 0, sp_grid = 0;          // tiles per wave per span; work-groups (all resident: they wait on each other)
    int64_t sp_spans = 0;                   // spans of 8 * P tiles
    int32_t sp_P_plan = 0, sp_max_grid = 0; // P as planned without knowing the selectivity (the ceiling of the adapted P); resident work-groups
    int32_t sp_P_plan_for[2] = {0, 0};      // ... for the whole chip / with a CU per XCD left to a communicator's kernels (single_pass_run_grid): the rounds of spans are quantised by the grid
    bool sp_P_fixed = false;                // P was set by the tuning hook: never adapted
    // the alternative plan of a projection with gathered columns: those columns streamed as always-true tile columns
    std::vector<FoldedPred> sp_pass;        // (once switched: part of the single-pass plan)
    bool alt_ok = false;
    int32_t alt_kinds[3] = {3, 3, 3}, alt_seg_col[3] = {-1, -1, -1};
    bool records_narrow_only = false;       // the projected predicate columns are all 1 byte wide (few survivors: the bitmap path beats the records)
    bool sp_narrow_checked = false;         // the first count has been looked at by the cost model ("leave the one launch?")
    imm3::PlanShape plan_shape;             // what the cost model (imm3_plan.h) needs to know of this query
    imm3::PlanDensity plan_density;         // where the survivors are: from the sample at creation (plan_have_density), else a default
    bool plan_have_density = false;
    bool sp_restore_pending = false;        // ... and it does: the next run sets the one launch up again
    uint64_t sp_restore_survivors = 0;
    bool sp_model_dropped = false;          // the cost model took this query off the one launch on an estimate: the first count may bring it back
    bool plan_pinned = false;               // tuning 12 at creation: the plan made there stands whatever the cost model says (tests of one plan's kernels)
    bool sp_have_stats = false;             // a run's count and dense-range tally have been seen (a reservation's estimate no longer moves P)
    size_t sp_rounds_max = 0;               // rounds at the smallest P: d_desc = {round totals, round counters, span descriptors (smallest P), trash lines}
    size_t sp_desc_off = 0;                 // byte offset of the span descriptors in d_desc's allocation
    unsigned long long *d_desc = nullptr;   // per-span descriptors of the chained scan
    size_t sp_trash_off = 0;                // byte offset of the writers' trash lines in d_desc's allocation
    imm3::ProjectTile *d_tile_desc = nullptr; // table queries: one descriptor per tile of the table for the launch's columns (k_filter_project's TABLE instances)
    bool ran_single_pass = false;           // the last run went through k_filter_project ...
    bool sp_verified = false;               // ... and its status word has been read since (rows complete, or gathered again from the bitmap)
    bool offsets_valid = false;             // d_tile_offsets / d_chunk_sums describe the last run's bitmap (an offsets scan has run since)
    bool count_log_on = false;              // imm3_query_log_counts is installed: every run logs the segment's count (a limit query then scans whole)
    bool select_partial = false;            // the last select pass was a limit scan in chunks: the bitmap and the count cover the tiles scanned until the
                                            // limit was reached (finish[kFinishLimitTiles]); imm3_query_count / _bitmap run the whole select first
    uint32_t sp_abandoned_runs = 0, sp_busy_runs = 0; // single-pass runs whose rows were gathered from the bitmap instead: a prefix never came / the device was busy (imm3_query_plan)
    bool count_pending_scan = false; // the last select run left the count to the projection's offsets scan
};


// reference counting (imm3_api.cpp)
namespace imm3 {
void ctx_retain(imm3_ctx *c);
void ctx_release(imm3_ctx *c);
int join_query_count(imm3_query *q, hipStream_t s); // make `s` wait for the query's count if it was reduced on the aux stream
int query_groups(imm3_query *q, uint32_t *n_groups);  // the aggregation's dense group list is complete in q->d_o* (synchronises)
}
