// imm3_api_internal.h -- what imm3_api.cpp (the C ABI: validation, launches, getters) and imm3_planner.cpp (plans and their
// geometry) share.  Internal to libimm3: nothing here is part of include/imm3.h.
#pragma once

#include "imm3_handles.h"

namespace imm3 {

// ---- imm3_api.cpp ----
hipError_t pool_alloc(imm3_ctx *ctx, void **out, size_t bytes);      // the context's caching allocator (stream-ordered reuse)
void pool_release(imm3_ctx *ctx, void *p);
void graphs_mark_stale(imm3_ctx *ctx, const imm3_query *q);          // recorded graphs that replay this query point at buffers that are about to move
void fill_tile_col(const imm3_query *q, const FoldedPred &fp, TileCol &c, int kind);

// ---- imm3_planner.cpp ----
constexpr int kSampleChunks = 8;                                      // the sample a plan is made on: eight chunks of 64 tiles spread over the segment / table
constexpr int64_t kSampleChunkTiles = 64;
constexpr int kSampleTiles = kSampleChunks * (int)kSampleChunkTiles;
constexpr int64_t kLimitFirstChunkTiles = 1024;                       // a limit scan's first chunk (the next ones are 8 x, 4 x, 4 x ... larger)

int tile_kind(const FoldedPred &fp);
int32_t single_pass_run_grid(const imm3_query *q);
bool single_pass_reserves(const imm3_query *q);
void single_pass_set_P(imm3_query *q, int32_t P);
void single_pass_pick_P(imm3_query *q, double sigma, bool sure);
void single_pass_adapt(imm3_query *q, uint64_t survivors, int64_t dense_ranges);
int single_pass_setup(imm3_query *q);
PlanDensity plan_density_for(const imm3_query *q, uint64_t survivors);
double plan_cost_three_launches(const imm3_query *q, const PlanDensity &d, bool records_possible, bool *use_records);
int single_pass_stream_columns(imm3_query *q, uint64_t survivors);
void records_drop_if_narrow(imm3_query *q, uint64_t survivors);
int records_setup(imm3_query *q);
void single_pass_drop_if_narrow(imm3_query *q, uint64_t survivors);
bool single_pass_restore_wanted(const imm3_query *q, uint64_t survivors);
int single_pass_restore(imm3_query *q, uint64_t survivors);
int single_pass_sample(imm3_query *q);

struct LimitScanInputs {
    bool whole = false, count_log_on = false, count_in_scan = false, single_tile_pass = false, table = false, records = false, skip_bitmap = false, overlap_total = false;
    int64_t limit = 0, n_tiles = 0;
    int filter_variant = 0;
};
bool limit_scan_applies(const LimitScanInputs &in);

} // namespace imm3
