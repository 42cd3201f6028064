// imm3_planner.cpp -- which plan a projection takes and with what geometry (round 5: split off imm3_api.cpp, which keeps the C
// ABI's validation, launches and getters).  Everything here is host arithmetic plus the one sampling launch at query creation:
//   * the one-launch projection's geometry: tiles per wave and span (P) per grid, the grid a run may use while a communicator
//     is attached, the descriptor allocation (single_pass_setup, single_pass_run_grid, single_pass_adapt);
//   * the cost model's decisions between the one launch, survivor records and the bitmap path (imm3_plan.h) on an estimate --
//     the sample at creation, a reservation -- or on a run's count (single_pass_stream_columns, records_drop_if_narrow,
//     single_pass_drop_if_narrow, single_pass_restore);
//   * whether a projection with a `limit` scans in chunks (limit_scan_applies).
// The reference has no planner to mirror: its Engine builds ScanOp -> SelectOp* -> ProjectOp per segment unconditionally
// (engine/src/main/scala/immutabledb/engine/Engine.scala:158-196); these are choices between equivalent executions of that chain.
#include "../../include/imm3.h"
#include "../../include/imm3_diag.h"
#include "imm3_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "imm3_handles.h"
#include "imm3_api_internal.h"

namespace imm3 {

// `limit` stops the scan (Project.scala:73-80; Engine.scala:166,253-258): a projection with a limit whose select chain is ONE tile
// launch over one uniform segment runs that launch as chunks of growing size (run_select, imm3_api.cpp).  Not when the caller wants
// the whole segment's count anyway (`whole`: a getter settling a partial scan; a count log), not for a select-only run
// (count_in_scan false: no projection follows), not for tables, records plans or count-only runs, not under the tuning variants that
// pin the old launches (7: k_total, 14: no chunks), and not for segments the first chunk already covers.
bool limit_scan_applies(const LimitScanInputs &in) {
    if (in.whole || in.count_log_on || !in.count_in_scan || !(in.limit > 0) || !in.single_tile_pass) return false;
    if (in.table || in.records || in.skip_bitmap || in.overlap_total) return false;
    if (in.filter_variant == 7 || in.filter_variant == 14) return false;
    return in.n_tiles > kLimitFirstChunkTiles;
}

int tile_kind(const FoldedPred &fp) {
    if (fp.kind == KIND_I32) return TK_I32;
    if (fp.kind == KIND_I8) return TK_I8;
    if (fp.kind == KIND_STR && fp.width == 2 && !fp.match.empty() && fp.match.size() <= (size_t)kMaxTileMatch) return TK_S2;
    return TK_NONE;
}

// ---- single-pass projection: tiles per wave and span (P) ----
// CUs a one-launch plan leaves free while a communicator is attached to its context: ONE PER XCD.  The kernel wants every CU (one
// work-group per CU, all resident); the kernel RCCL launches for the count all-reduce of the pass before cannot share a CU with such
// a work-group (ncclDevKernel_Generic for gfx950: 512 threads x 256 vector registers and 37 664 bytes of LDS, beside 3 waves x 131
// registers per SIMD and 140 KB) -- so with all 256 CUs taken it waits until a work-group LEAVES, i.e. for the whole pass, and when
// it wins the race instead, one work-group of the pass starts late by the collective's duration and the pass ends that much later.
// Work-groups go to the XCDs round-robin BEFORE the dispatcher knows where there is room, so a free CU only helps on the XCD the
// collective's work-group is sent to: four free CUs (grid 252: XCDs 0-3 full) changed nothing, one per XCD does (tools/overlap_probe.py,
// profiles/r05_overlap.txt; the table is in DESIGN section 8).
constexpr int kXcds = 8;
constexpr int kCommReservedCUs = kXcds;
bool single_pass_reserves(const imm3_query *q) {
    return q->ctx->comms_attached.load(std::memory_order_relaxed) > 0 && q->ctx->filter_variant != 16 && q->sp_max_grid > 4 * kCommReservedCUs; // (tuning 16: no reservation, for A/B runs)
}
int32_t single_pass_run_grid(const imm3_query *q) {
    const int64_t g = single_pass_reserves(q) ? q->sp_max_grid - kCommReservedCUs : q->sp_max_grid;
    return (int32_t)std::max<int64_t>(1, std::min<int64_t>(g, q->sp_spans));
}
void single_pass_set_P(imm3_query *q, int32_t P) {
    const int64_t tiles_per_span = (int64_t)P * kProjectStreamers;
    q->sp_P = P;
    q->sp_spans = (q->n_tiles + tiles_per_span - 1) / tiles_per_span;
    q->sp_grid = single_pass_run_grid(q);
}
// The host has learnt how many rows survive (a count it fetched together with the number of ranges that outgrew their LDS
// ring, dense_ranges; or a reservation, dense_ranges < 0): later runs use a P at which a range's survivors fill about 45 %
// of a streamer's ring -- the streamers then keep compacting while the writers unpack, and no range outgrows its ring (a
// range that does is unpacked from the source columns row by row: several times slower per row than from records).
// Survivors are taken to sit in the dense ranges when there were any (a sorted key: half the table survives, all of it
// in one half); when not even two tiles of such a range fit a ring the planned P stays -- there the row-by-row path
// beats short ranges.  Measured at 100 M rows: 50 % survivors of an int8 column, evenly spread: P = 14 (the plan, made
// for ~10 %) 386 us, P = 3 164 us (three launches: 180-208); 28 % of int8 + int32: P = 6 506 us, P = 2 190 us (242);
// id > 5e7 on the sorted key: P = 7 232-269 us, P = 2 364 us, P = 1 426 us (338).  A run recorded in a graph keeps the P
// it was recorded with (the descriptors' layout does not depend on P).
// sigma: survivors per row WHERE THERE ARE SURVIVORS (the local density: what a range has to hold); `sure`: measured by a run
void single_pass_pick_P(imm3_query *q, double sigma, bool sure) {
    if (!q->single_pass || q->sp_P_fixed || !(sigma > 0.0)) return;
    const int R = project_rec_dwords(q->stage_kinds);
    const double ring_records = (double)kProjectRingBytes / (4.0 * R);
    const double per_tile = std::min(1.0, sigma) * kTileRows;
    const double fill = per_tile * q->sp_P / ring_records;
    if (!sure && fill >= 0.3 && fill <= 0.6) return; // (close enough: P does not flip between an estimate and the count)
    int64_t P = (int64_t)(0.45 * ring_records / per_tile);
    if (P < 2) P = (int64_t)(0.9 * ring_records / per_tile); // (nearly every row survives: whatever still fits)
    if (P < 2) P = q->sp_P_plan;
    P = std::min<int64_t>(P, q->sp_P_plan);
    if ((int32_t)P == q->sp_P) return;
    // The descriptors are tagged with the run counter (26 bits): leave no tag behind in places the new P does not rewrite every
    // run (stream order: after the runs so far, before the next one).  Hygiene, not correctness -- an alias would need 2^26 runs --
    // so a failed memset only means the old P stays.
    if (hipMemsetAsync(q->d_desc, 0, q->sp_trash_off, q->ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    single_pass_set_P(q, (int32_t)P);
}
void single_pass_adapt(imm3_query *q, uint64_t survivors, int64_t dense_ranges) {
    if (!q->single_pass || q->sp_P_fixed || q->n_rows <= 0 || survivors == 0) return;
    if (dense_ranges < 0 && q->sp_have_stats) return; // (a reservation says less than a run did)
    if (dense_ranges >= 0) q->sp_have_stats = true;
    double sigma = (double)survivors / (double)q->n_rows;
    const double n_ranges = (double)q->sp_spans * kProjectStreamers;
    const bool clustered = dense_ranges > 0 && (double)dense_ranges > 0.02 * n_ranges;
    if (clustered) sigma = std::min(1.0, sigma * n_ranges / (double)dense_ranges);
    single_pass_pick_P(q, sigma, dense_ranges > 0);
}

// Plans the one-launch projection for the tile columns in q->stage_kinds: P, the grid, the descriptor allocation.
// q->single_pass stays false when the kernel cannot run here (no instance, no resident work-group).
int single_pass_setup(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    const int R = project_rec_dwords(q->stage_kinds);
    int64_t tile_bytes = 0;
    for (int k = 0; k < kMaxTileCols; ++k)
        tile_bytes += (q->stage_kinds[k] == TK_I32 ? 4 : (q->stage_kinds[k] == TK_S2 ? 2 : (q->stage_kinds[k] == TK_I8 ? 1 : 0))) * (int64_t)kTileRows;
    // Tiles per wave and span (P).  Large enough that a span's prefix (a ~10 us chain through three other work-groups)
    // and its unpacking fit in the time the streamers need for the next spans; small enough that three ranges of
    // ~10 % survivors fit a streamer's LDS ring -- the streamers then never wait for a writer -- and that the last
    // round, whose prefix nothing overlaps, is short.  Within that window P is the value that fills the last round of
    // spans best (spans are dealt round-robin to one work-group per CU).  Measured on C3 (100 M rows, R = 2): P = 6-8
    // 127-133 us, P = 12 143 us, P = 4 154 us.
    int maxg = project_max_grid(q->stage_kinds, 0);
    if (ctx->grid_blocks > 0) maxg = std::min(maxg, ctx->grid_blocks.load());
    const int64_t ring_records = kProjectRingBytes / (4 * R);
    int64_t p_hi = std::max<int64_t>(4, std::min<int64_t>(16, ring_records / 3 / 85));
    if (tile_bytes > 0) p_hi = std::max<int64_t>(4, std::min<int64_t>(p_hi, (60 * 1024) / tile_bytes));
    // Spans go to the work-groups in ROUNDS of one span each, and a round takes the time of its P tiles whether all work-groups have a
    // span in it or one: the launch costs rounds x P tile times (+ a prefix chain per round), and the rounds are quantised by the
    // grid.  100 M rows on 256 CUs: P = 6 -> 7.95 rounds -> 8 x 6 = 48 tile times (47.7 is the floor); on 248 CUs (a CU per XCD left
    // to a communicator) P = 6 -> 8.2 rounds -> 9 x 6 = 54, P = 5 -> 9.85 -> 10 x 5 = 50.  So P is planned per grid: the candidate
    // (two below the ceiling the ring and the register sets allow, never under 4) with the fewest tile times, the larger P on a tie.
    auto plan_P = [&](int64_t grid) -> int64_t {
        int64_t best_P = p_hi;
        double best = 1e300;
        for (int64_t p = p_hi; p >= std::max<int64_t>(4, p_hi - 2) && grid > 0; --p) {
            const int64_t spans = (q->n_tiles + p * kProjectStreamers - 1) / (p * kProjectStreamers);
            const int64_t rounds = (spans + grid - 1) / grid;
            const double cost = (double)rounds * ((double)p + 0.35); // (+ the part of a round's prefix chain and ring hand-offs that nothing hides)
            if (cost < best - 1e-9) { best = cost; best_P = p; }
        }
        return best_P;
    };
    int64_t P = plan_P(maxg);
    const int64_t P_reserved = plan_P(maxg > 4 * kCommReservedCUs ? maxg - kCommReservedCUs : maxg);
    const bool fixed = ctx->filter_variant > 200 && ctx->filter_variant <= 200 + kProjectMaxP;
    if (fixed) P = ctx->filter_variant - 200; // tuning: variant 200 + P
    if (maxg < 1 || q->n_tiles < 1) return IMM3_OK;
    // One allocation: round totals and round counters first (at the same place whatever P a run uses), then the span
    // descriptors of the smallest P a run may use, then one 64-byte trash line per writer wave.  The plan is committed only
    // once the allocation stands (a query whose descriptors could not be allocated keeps the three launches).
    const int64_t spans_max = (q->n_tiles + kProjectMinP * kProjectStreamers - 1) / (kProjectMinP * kProjectStreamers);
    const int64_t grid_min = std::max<int64_t>(1, std::min<int64_t>(maxg > 4 * kCommReservedCUs ? maxg - kCommReservedCUs : maxg, spans_max)); // (the smallest grid a run may use: single_pass_run_grid)
    const size_t rounds_max = (size_t)((spans_max + grid_min - 1) / grid_min);
    const size_t desc_off = (rounds_max * (sizeof(unsigned long long) + sizeof(uint32_t)) + 255) / 256 * 256;
    const size_t desc_bytes = desc_off + (size_t)spans_max * sizeof(unsigned long long);
    const size_t trash_off = (desc_bytes + 255) / 256 * 256;
    void *d = nullptr;
    HIPCHK(pool_alloc(ctx, &d, trash_off + (size_t)maxg * kProjectWriters * 64));
    const hipError_t me = hipMemsetAsync(d, 0, desc_bytes, ctx->stream); // (pooled memory: another query's descriptors)
    if (me != hipSuccess) {
        pool_release(ctx, d);
        HIPCHK(me);
    }
    if (q->table && !q->d_tile_desc) { // the launch's view of the tile table: one descriptor per tile (built on the device from the table's per-column pointers)
        void *td = nullptr;
        const hipError_t te = pool_alloc(ctx, &td, (size_t)q->n_tiles * sizeof(ProjectTile));
        if (te != hipSuccess) {
            pool_release(ctx, d);
            HIPCHK(te);
        }
        const void *const *tp[kMaxTileCols] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < kMaxTileCols; ++k)
            if (q->stage_kinds[k] != TK_NONE && q->stage_seg_col[k] >= 0) tp[k] = (const void *const *)q->table->d_tile_ptrs[(size_t)q->stage_seg_col[k]];
        launch_project_tile_desc(q->table->d_tile_rows, tp[0], tp[1], tp[2], (ProjectTile *)td, q->n_tiles, ctx->stream);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            pool_release(ctx, d);
            pool_release(ctx, td);
            HIPCHK(le);
        }
        q->d_tile_desc = (ProjectTile *)td;
    }
    q->d_desc = (unsigned long long *)d;
    q->sp_rounds_max = rounds_max;
    q->sp_desc_off = desc_off;
    q->sp_trash_off = trash_off;
    q->single_pass = true;
    q->sp_P_fixed = fixed;
    q->sp_max_grid = maxg;
    q->sp_P_plan_for[0] = (int32_t)P;
    q->sp_P_plan_for[1] = fixed ? (int32_t)P : (int32_t)P_reserved;
    q->sp_P_plan = q->sp_P_plan_for[single_pass_reserves(q) ? 1 : 0];
    single_pass_set_P(q, q->sp_P_plan);
    return IMM3_OK;
}

// ---- the cost model's view of a query (imm3_plan.h) ----
// Where the survivors are, given a count (or an estimate of one): the sample taken at creation knows how they are spread (survivors
// per row where there are survivors, and how many sit in fully surviving stretches); without it they are taken as spread evenly.
PlanDensity plan_density_for(const imm3_query *q, uint64_t survivors) {
    PlanDensity d;
    d.sigma = q->n_rows > 0 ? std::min(1.0, (double)survivors / (double)q->n_rows) : 0.0;
    d.sloc = d.sigma;
    d.full = 0.0;
    if (q->plan_have_density) { // (how densely the survivors sit where they sit is the data's property: a better count does not change it)
        d.sloc = std::min(1.0, std::max(d.sigma, q->plan_density.sloc));
        d.full = q->plan_density.full;
    }
    return d;
}
// the plan a query would fall back to from the one launch, and its predicted cost: records when a predicate column is projected
// and they are predicted cheaper than the bitmap path
double plan_cost_three_launches(const imm3_query *q, const PlanDensity &d, bool records_possible, bool *use_records) {
    const double c = plan_cost('C', q->plan_shape, d);
    const double b = records_possible ? plan_cost('B', q->plan_shape, d) : 1e30;
    if (use_records) *use_records = b < c;
    return std::min(b, c);
}

// A projection with gathered SELECT-list columns was planned as three launches (records -> k_scan -> k_emit).  Now the host
// knows how many rows survive: when that is enough for a gather to touch most 128-byte lines of the column anyway, the
// column is STREAMED instead -- it joins the one-launch kernel as a tile column whose predicate every value passes, and its
// values ride in the records like a predicate column's.  (`select id, age ... where age > 18 and age < 30`, 11 % of 100 M rows:
// 123 us against 174; at 3 % the three launches win.)  Called outside a capture, before the query's row arrays exist or
// from imm3_query_reserve_rows; the records' buffers go back to the pool.
int single_pass_stream_columns(imm3_query *q, uint64_t survivors) {
    imm3_ctx *ctx = q->ctx;
    if (!q->alt_ok || q->single_pass || ctx->capture || q->n_rows <= 0) return IMM3_OK;
    if (ctx->filter_variant != 9) { // (9: streamed whatever the prediction)
        if (q->plan_pinned) return IMM3_OK;
        const PlanDensity d = plan_density_for(q, survivors);
        const double now = q->d_stage_rec ? std::min(plan_cost('B', q->plan_shape, d), plan_cost('C', q->plan_shape, d)) : plan_cost('C', q->plan_shape, d);
        if (!(plan_cost('A', q->plan_shape, d) < kPlanKeepMargin * now)) return IMM3_OK;
    }
    int32_t keep_kinds[kMaxTileCols], keep_cols[kMaxTileCols];
    for (int k = 0; k < kMaxTileCols; ++k) {
        keep_kinds[k] = q->stage_kinds[k];
        keep_cols[k] = q->stage_seg_col[k];
        q->stage_kinds[k] = q->alt_kinds[k];
        q->stage_seg_col[k] = q->alt_seg_col[k];
    }
    const int rc = single_pass_setup(q);
    if (rc || !q->single_pass) { // (cannot run here: the three launches stay)
        for (int k = 0; k < kMaxTileCols; ++k) {
            q->stage_kinds[k] = keep_kinds[k];
            q->stage_seg_col[k] = keep_cols[k];
        }
        q->alt_ok = false;
        return rc;
    }
    graphs_mark_stale(ctx, q); // (a graph that recorded the three launches points at buffers that go now)
    pool_release(ctx, q->d_stage_rec);
    pool_release(ctx, q->d_tile_start);
    q->d_stage_rec = nullptr;
    q->d_tile_start = nullptr;
    q->stage_written = false;
    q->alt_ok = false;
    single_pass_adapt(q, survivors, -1);
    return IMM3_OK;
}

// Survivor records planned, but the bitmap path is predicted cheaper for this many survivors (the staging instance of the filter
// kernel costs 10-35 us per 100 M rows more than the plain one, plus the records' bytes; it buys the emit kernel the projected
// predicate columns: age > 97 -> id, age at 1 %: 75 us with records, 65 without; id > 9e7 -> id, age: 152 / 140; state in (8
// values) -> id, state, age: 277 / 243).  The records' buffers go back to the pool.
void records_drop_if_narrow(imm3_query *q, uint64_t survivors) {
    imm3_ctx *ctx = q->ctx;
    if (q->single_pass || !q->d_stage_rec || q->plan_pinned || ctx->capture || ctx->filter_variant == 11 || ctx->filter_variant == 6 || q->n_rows <= 0) return;
    const PlanDensity d = plan_density_for(q, survivors);
    if (!(plan_cost('C', q->plan_shape, d) < kPlanKeepMargin * plan_cost('B', q->plan_shape, d))) return;
    graphs_mark_stale(ctx, q);
    pool_release(ctx, q->d_stage_rec);
    pool_release(ctx, q->d_tile_start);
    q->d_stage_rec = nullptr;
    q->d_tile_start = nullptr;
    q->stage_written = false;
    q->alt_ok = false; // (settled: three launches from the bitmap)
}

// Survivor records (k_filter_tile STAGE -> k_emit) for the tile columns in q->stage_kinds.  Every wave of the staging launch
// writes its records to its own arena, so the launch geometry is fixed here: 768 work-groups (3 per CU: an 8 KiB record
// buffer per wave), grid-stride over groups of T tiles.
int records_setup(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    const int R = rec_layout(q->stage_kinds, -1).dwords;
    const int T = filter_tile_group(q->stage_kinds);
    // 3 per CU (~52 KiB of LDS each); 4 per CU for a lone 2-byte-string column (1-dword records, 4 KiB record buffers: C4's
    // filter 50.5 -> 46.6 us).  Measured per shape: a lone int8 column at 1024 lost 8 us, an int32 column 7 us.
    const bool lone_s2 = q->stage_kinds[0] == TK_S2 && q->stage_kinds[1] == TK_NONE;
    const int cap = ctx->grid_blocks > 0 ? std::min(ctx->grid_blocks.load(), kMaxFilterGrid) : (R == 1 && lone_s2 ? 1024 : 768);
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((q->n_tiles + kWavesPerBlock - 1) / kWavesPerBlock, cap));
    const int64_t n_waves = grid * kWavesPerBlock;
    const int64_t n_groups = T > 0 ? (q->n_rows / kTileRows) / T : 0;
    const int64_t main_tiles = n_groups * T;
    const int64_t max_slots = ((n_groups + n_waves - 1) / n_waves) * T + (q->n_tiles - main_tiles + n_waves - 1) / n_waves;
    if (T > 0 && max_slots <= kMaxArenaSlots) {
        q->stage_grid = (int32_t)grid;
        q->stage_T = T;
        q->stage_max_slots = (int32_t)std::max<int64_t>(max_slots, 1) + 1; // (+ 1: behind a wave's tile starts sits its arena's end -- the last tile's length, for an offsets scan without a bitmap)
        q->stage_wave_cap = (int64_t)q->stage_max_slots * kTileRows; // (skewing the arena bases off their 128 KiB-aligned stride changed nothing)
        q->stage_main_tiles = main_tiles;
        void *d = nullptr;
        HIPCHK(pool_alloc(ctx, &d, (size_t)n_waves * (size_t)q->stage_wave_cap * 4 * (size_t)R + 256));
        q->d_stage_rec = (uint8_t *)d;
        HIPCHK(pool_alloc(ctx, &d, (size_t)n_waves * (size_t)q->stage_max_slots * sizeof(uint32_t) + 256));
        q->d_tile_start = (uint32_t *)d;
    }
    return IMM3_OK;
}

// The one launch planned, but three launches are predicted cheaper for this many survivors.  The one-launch kernel costs ~23 us +
// 0.6-0.85 us per million rows whatever the columns' widths (it is bound by instructions per row, DESIGN finding 21), and its
// writers walk stretches of mostly-surviving rows at ~1.5 us per million rows and column; the plain filter over 1- and 2-byte
// columns takes 21-45 us per 100 M rows and k_gather ~2.5 us per million survivors.  So: narrow predicate columns alone take
// filter -> offsets scan -> gather until many rows survive (select age ... where age > 98, 1 %: 89 us in one launch, 53 in three);
// small segments take it nearly always (4 M rows: 27 us against 16-20); and C3's shape takes it again above ~50 % survivors spread
// evenly (60 %: 373 us in one launch, 329 in three).  A projected string column with few survivors is better staged in records than
// gathered (state = CA -> state, 2 %: 62 us with records, 75 from the bitmap): the cheaper of the two is taken.
void single_pass_drop_if_narrow(imm3_query *q, uint64_t survivors) {
    imm3_ctx *ctx = q->ctx;
    if (!q->single_pass || !q->sp_pass.empty() || q->sp_P_fixed || q->plan_pinned || ctx->capture || ctx->filter_variant == 8 || ctx->filter_variant == 11 || q->n_rows <= 0) return;
    const PlanDensity d = plan_density_for(q, survivors);
    bool use_records = false;
    const double other = plan_cost_three_launches(q, d, ctx->filter_variant != 3 && !q->table, &use_records); // (a table has no survivor records: the bitmap path)
    if (!(other < kPlanKeepMargin * plan_cost('A', q->plan_shape, d))) return;
    graphs_mark_stale(ctx, q);
    q->single_pass = false;
    q->sp_model_dropped = true; // (the first count may bring it back: single_pass_restore)
    pool_release(ctx, q->d_desc);
    q->d_desc = nullptr;
    pool_release(ctx, q->d_tile_desc);
    q->d_tile_desc = nullptr;
    if (use_records && !q->d_stage_rec) {
        if (records_setup(q) != IMM3_OK || !q->d_tile_start) { // (no memory for the records: the bitmap path needs none)
            pool_release(ctx, q->d_stage_rec);
            q->d_stage_rec = nullptr;
            (void)hipGetLastError();
        }
    }
}

// ... and back: the one launch was left on an estimate (the sample, or the guess made for a segment too small to sample), and the
// first run's count says it is the cheaper plan after all (a small segment most of whose rows survive: 4 M rows, 60 %: 27 us in
// one launch, 37 in three; a range of the sorted key that the sample's chunks missed).
bool single_pass_restore_wanted(const imm3_query *q, uint64_t survivors) {
    const imm3_ctx *ctx = q->ctx;
    if (q->single_pass || !q->sp_model_dropped || q->plan_pinned || ctx->capture || ctx->filter_variant == 6 || ctx->filter_variant == 3 || q->n_rows <= 0) return false;
    const PlanDensity d = plan_density_for(q, survivors);
    const double now = q->d_stage_rec ? plan_cost('B', q->plan_shape, d) : plan_cost('C', q->plan_shape, d);
    return plan_cost('A', q->plan_shape, d) < kPlanKeepMargin * now;
}
// (the run that read the count finishes on the plan it started with -- its filter and offsets scan are done, the gather is the
// smaller part -- and the NEXT run takes the one launch: the switch happens at the start of that run)
int single_pass_restore(imm3_query *q, uint64_t survivors) {
    imm3_ctx *ctx = q->ctx;
    q->sp_restore_pending = false;
    if (q->single_pass || !q->sp_model_dropped || q->plan_pinned || ctx->capture) return IMM3_OK;
    const int rc = single_pass_setup(q);
    if (rc || !q->single_pass) return rc; // (cannot run here: the three launches stay)
    graphs_mark_stale(ctx, q);
    pool_release(ctx, q->d_stage_rec);
    pool_release(ctx, q->d_tile_start);
    q->d_stage_rec = nullptr;
    q->d_tile_start = nullptr;
    q->stage_written = false;
    q->sp_model_dropped = false;
    q->sp_narrow_checked = true; // (decided on a count: no second look)
    single_pass_adapt(q, survivors, -1);
    return IMM3_OK;
}

// A look at the data before the first run: the select chain's count over eight evenly spaced chunks of 64 tiles (0.5 % of
// 100 M rows, eight count-only launches of the scan+select kernel and one strided copy, inside query creation, which ends with
// a stream synchronisation anyway: + 0.05-0.15 ms on a creation of 0.35-0.6 ms; tools/first_run.py).  Most queries run ONCE (the reference's Engine plans, runs and drops a pipeline per statement),
// so what later runs learn from a count -- P, streamed SELECT-list columns -- the first run gets from the sample.  Survivors
// per row are also taken per chunk: weighted by the chunks' own survivors they give the density where the survivors ARE,
// which tells a sorted key's all-or-nothing ranges (keep the planned P: unpack_dense) from the same number of survivors
// spread evenly (shorter ranges).

// the sample's tile table for one column of the segment (cached on the segment: the sampled tiles are the segment's, not the query's)
int sample_tile_ptrs(imm3_ctx *ctx, const imm3_segment *cseg, int32_t col, int64_t n_full, void ***out, uint32_t **rows_out) {
    imm3_segment *seg = const_cast<imm3_segment *>(cseg);
    // The device allocations and copies happen OUTSIDE the segment's lock (they wait for the device: query creations on one
    // segment would queue up behind them); the lock only covers looking the tables up and publishing them.  Two creations that
    // race for the same column both build a table and the loser's is freed.
    bool have_rows = false, have_ptrs = false;
    {
        std::lock_guard<std::mutex> g(seg->layout_mu);
        if (seg->sample_full_tiles >= 0 && seg->sample_full_tiles != n_full) return fail(IMM3_ERR_STATE, "internal: the segment's sample was laid out for another row count");
        have_rows = seg->d_sample_rows != nullptr;
        have_ptrs = seg->d_sample_ptrs.find(col) != seg->d_sample_ptrs.end();
    }
    void *new_rows = nullptr, *new_ptrs = nullptr;
    if (!have_rows) {
        std::vector<uint32_t> rows((size_t)kSampleTiles, (uint32_t)kTileRows);
        HIPCHK(hipMalloc(&new_rows, rows.size() * sizeof(uint32_t)));
        const hipError_t e = hipMemcpy(new_rows, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(new_rows); HIPCHK(e); }
    }
    if (!have_ptrs) {
        const SegCol &sc = seg->cols[(size_t)col];
        std::vector<const void *> ptrs((size_t)kSampleTiles);
        for (int i = 0; i < kSampleChunks; ++i) {
            int64_t tile0 = (int64_t)((2 * i + 1) * n_full / (2 * kSampleChunks)) - kSampleChunkTiles / 2;
            tile0 = std::max<int64_t>(0, std::min<int64_t>(tile0, n_full - kSampleChunkTiles));
            for (int64_t t = 0; t < kSampleChunkTiles; ++t)
                ptrs[(size_t)(i * kSampleChunkTiles + t)] = col_flat(sc) + (size_t)(tile0 + t) * kTileRows * (size_t)sc.width;
        }
        hipError_t e = hipMalloc(&new_ptrs, ptrs.size() * sizeof(void *));
        if (e == hipSuccess) e = hipMemcpy(new_ptrs, ptrs.data(), ptrs.size() * sizeof(void *), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(new_rows); (void)hipFree(new_ptrs); HIPCHK(e); }
    }
    void *drop_rows = nullptr, *drop_ptrs = nullptr;
    {
        std::lock_guard<std::mutex> g(seg->layout_mu);
        if (new_rows) {
            if (!seg->d_sample_rows) { seg->d_sample_rows = (uint32_t *)new_rows; seg->sample_full_tiles = n_full; }
            else drop_rows = new_rows;
        }
        auto it = seg->d_sample_ptrs.find(col);
        if (new_ptrs) {
            if (it == seg->d_sample_ptrs.end()) it = seg->d_sample_ptrs.emplace(col, (void **)new_ptrs).first;
            else drop_ptrs = new_ptrs;
        }
        *out = it->second;
        *rows_out = seg->d_sample_rows;
    }
    (void)hipFree(drop_rows);
    (void)hipFree(drop_ptrs);
    (void)ctx;
    return IMM3_OK;
}

int single_pass_sample(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    const int64_t n_full = q->table ? q->n_tiles : q->n_rows / kTileRows; // (a table's sample may hold a segment's partial last tile: the kernel's rolled path)
    const bool undecided = q->single_pass || q->alt_ok || q->d_stage_rec;
    if (!undecided || q->sp_P_fixed || q->plan_pinned || ctx->filter_variant == 10 || n_full < 4096) return IMM3_OK; // (below ~4 M rows the sample costs what it saves)
    if (q->table && !q->table->d_sample_rows) return IMM3_OK;
    // ONE count-only launch of the scan+select kernel's table instance over the sample's tile table (round 3: eight launches, a
    // memset and a strided copy): 128 work-groups, one tile per wave, so that work-groups 16 i .. 16 i + 15 hold chunk i's count
    // in their partials.
    TileArgs a;
    std::memset(&a, 0, sizeof(a));
    bool any = false;
    for (int k = 0; k < kMaxTileCols; ++k) {
        a.kinds[k] = q->stage_kinds[k];
        if (a.kinds[k] == TK_NONE) continue;
        const FoldedPred *fp = nullptr;
        for (const auto &p : q->preds)
            if (p.seg_col == q->stage_seg_col[k]) fp = &p;
        if (!fp) return IMM3_OK; // (a streamed column already: nothing left to decide)
        if (is_compressed(q->seg->cols[(size_t)fp->seg_col].codec) && !q->seg->cols[(size_t)fp->seg_col].d_dense) return IMM3_OK;
        fill_tile_col(q, *fp, a.cols[k], a.kinds[k]);
        void **ptrs = nullptr;
        uint32_t *rows = nullptr;
        if (q->table) { // (made with the table: imm3_table_create)
            ptrs = q->table->d_sample_ptrs[(size_t)fp->seg_col];
            rows = q->table->d_sample_rows;
        } else {
            const int rc = sample_tile_ptrs(ctx, q->seg, fp->seg_col, n_full, &ptrs, &rows);
            if (rc) return rc;
        }
        a.tile_ptrs[k] = (const void *const *)ptrs;
        a.tile_rows = rows;
        any = true;
    }
    if (!any) return IMM3_OK; // (no predicate: every row survives, the plan for that is the dense path at the planned P)
    a.n_rows = (int64_t)kSampleTiles * kTileRows;
    a.n_words = (int64_t)kSampleTiles * kTileWords;
    a.n_tiles = kSampleTiles;
    a.bitmap = nullptr; // count-only
    a.block_partials = q->d_block_partials;
    a.finish = nullptr;
    constexpr int kGrid = kSampleTiles / kWavesPerBlock; // 128: wave w of the launch takes tile w
    if (!launch_filter_tile(a, kGrid, ctx->stream, nullptr, nullptr)) return IMM3_OK;
    HIPCHK(hipGetLastError());
    uint32_t partials[kGrid] = {0};
    HIPCHK(hipMemcpyAsync(partials, q->d_block_partials, sizeof(partials), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const double chunk_rows = (double)(kSampleChunkTiles * kTileRows);
    double sum = 0.0, sum_sq = 0.0, sum_full = 0.0;
    for (int i = 0; i < kSampleChunks; ++i) {
        double c = 0.0;
        for (int b = 0; b < kGrid / kSampleChunks; ++b) {
            const double part = (double)partials[i * (kGrid / kSampleChunks) + b];
            c += part;
            if (part >= (double)(kWavesPerBlock * kTileRows)) sum_full += part; // (a work-group's eight tiles, every row of them)
        }
        sum += c;
        sum_sq += c * c;
    }
    // (nothing in the sample: fewer than one row in 130 000 survives, or they all sit between the sample's chunks -- the plans for
    // very few survivors are compared; the first count corrects a miss)
    const double sigma = std::max(sum, 0.5) / (chunk_rows * kSampleChunks), sigma_local = sum > 0.0 ? sum_sq / (sum * chunk_rows) : sigma;
    q->plan_density.sigma = sigma;
    q->plan_density.sloc = std::max(sigma, std::min(1.0, sigma_local));
    q->plan_density.full = sum > 0.0 ? sum_full / sum : 0.0;
    q->plan_have_density = true;
    const uint64_t estimate = (uint64_t)(sigma * (double)q->n_rows);
    const int rc = single_pass_stream_columns(q, estimate);
    if (rc) return rc;
    records_drop_if_narrow(q, estimate);
    single_pass_drop_if_narrow(q, estimate);
    if (sum > 0.0) single_pass_pick_P(q, sigma_local, false);
    return IMM3_OK;
}

} // namespace imm3

using namespace imm3;

// diagnostics: the cost model's prediction for a shape (tests hold it against immutable3_amd/plan_model.py; tools print it)
extern "C" int imm3_plan_predict(int64_t n_rows, const int32_t *pred_width, const int32_t *pred_match, int32_t n_pred, const int32_t *proj_width,
                                 const int32_t *proj_is_pred, int32_t n_proj, int32_t rec_bytes, double sigma, double sloc, double full, double *out_abc) {
    if (n_pred < 0 || n_pred > kPlanMaxCols || n_proj < 0 || n_proj > kPlanMaxCols || !out_abc || (n_pred > 0 && (!pred_width || !pred_match)) ||
        (n_proj > 0 && (!proj_width || !proj_is_pred)))
        return fail(IMM3_ERR_ARG, "bad argument");
    PlanShape ps;
    ps.n_rows = n_rows;
    ps.n_pred = n_pred;
    for (int i = 0; i < n_pred; ++i) { ps.pred_width[i] = pred_width[i]; ps.pred_match[i] = pred_match[i]; }
    ps.n_proj = n_proj;
    for (int i = 0; i < n_proj; ++i) { ps.proj_width[i] = proj_width[i]; ps.proj_is_pred[i] = proj_is_pred[i] != 0; }
    ps.rec_bytes = rec_bytes;
    PlanDensity d;
    d.sigma = sigma;
    d.sloc = sloc;
    d.full = full;
    out_abc[0] = plan_cost('A', ps, d);
    out_abc[1] = plan_cost('B', ps, d);
    out_abc[2] = plan_cost('C', ps, d);
    return IMM3_OK;
}

// diagnostics: the limit-scan decision as the library makes it (a pure function of these inputs; tests/test_host.py walks it)
extern "C" int imm3_plan_limit_scan(int32_t whole, int32_t count_log_on, int32_t count_in_scan, int64_t limit, int32_t single_tile_pass, int32_t table, int32_t records,
                                    int32_t skip_bitmap, int32_t overlap_total, int32_t filter_variant, int64_t n_tiles) {
    LimitScanInputs in;
    in.whole = whole != 0;
    in.count_log_on = count_log_on != 0;
    in.count_in_scan = count_in_scan != 0;
    in.limit = limit;
    in.single_tile_pass = single_tile_pass != 0;
    in.table = table != 0;
    in.records = records != 0;
    in.skip_bitmap = skip_bitmap != 0;
    in.overlap_total = overlap_total != 0;
    in.filter_variant = filter_variant;
    in.n_tiles = n_tiles;
    return limit_scan_applies(in) ? 1 : 0;
}
