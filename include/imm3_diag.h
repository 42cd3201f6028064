/*
 * imm3_diag.h -- measurement and tuning hooks of libimm3.so.  NOT part of the drop-in boundary (include/imm3.h): nothing
 * a GpuScanOp / GpuSelectOp / GpuProjectOp binds lives here.  bench.py, tools/ and the roofline tests use them to time
 * kernels with HIP events, to cross-check that timing against the device clock, to measure the GPU's read-only
 * streaming ceiling, and to A/B kernel variants.
 */
#ifndef IMM3_DIAG_H
#define IMM3_DIAG_H

#include "imm3.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- live kernel timing (HIP events on the context's stream) ----
 * When enabled, every kernel launch of this context is bracketed by an event pair.
 * kernel ids: 0 = scan+select, 1 = offsets scan, 2 = compact+gather, 3 = count reduce, 4 = group-by aggregation,
 *             5 = PFOR_INT / snappy column decode. */
int imm3_ctx_timing_enable(imm3_ctx *ctx, int32_t max_records);
int imm3_ctx_timing_reset(imm3_ctx *ctx);
/* Only launches whose kernel id has its bit set in `kernel_mask` are bracketed (default: all). */
int imm3_ctx_timing_mask(imm3_ctx *ctx, uint32_t kernel_mask);
/* Synchronises, then writes up to cap durations (ms) of launches of `kernel_id`, oldest first. */
int imm3_ctx_timing_collect(imm3_ctx *ctx, int32_t kernel_id, float *ms_out, int32_t cap, int32_t *n_out);

/* Cross-check of the event timing: when enabled, every tile-kernel launch also records, per work-group, the 100 MHz
 * device clock at entry and exit; collect() returns per launch (last work-group's exit - first work-group's entry) in
 * ms, oldest first.  Diagnostics only (bench.py's instrumented pass); costs two stores per work-group. */
int imm3_ctx_devclock_enable(imm3_ctx *ctx, int32_t max_launches);
int imm3_ctx_devclock_collect(imm3_ctx *ctx, float *ms_out, int32_t cap, int32_t *n_out);
/* The raw stamps of one recorded launch (100 MHz ticks): [2 b] / [2 b + 1] = entry / exit of work-group b; the tools' build of
 * the single-pass kernel adds per-range stamps behind them. */
int imm3_ctx_devclock_raw(imm3_ctx *ctx, int32_t launch, uint64_t *out, int32_t n);

/* Empirical read-only streaming ceiling of this GPU: times a kernel that only reads `bytes` (non-temporal dword loads,
 * same tiling and grid as the scan+select kernel, three rotated buffers) and returns the median GB/s over `iters`. */
int imm3_ctx_measure_read_gbps(imm3_ctx *ctx, uint64_t bytes, int32_t iters, double *gbps);

/* Tuning knobs (0 = default): filter variant, grid size in workgroups.  For experiments/bench.
 * variant 1 = word-at-a-time kernel only, 2 = count reduce on the aux stream, 3 = no survivor staging,
 * 4 = stage int32 columns only, 5 = PFOR_INT predicates read the decoded column instead of the compressed blocks,
 * 7 = reduce the count with a separate k_total launch instead of inside the filter kernel,
 * 6 = projections never use the one-launch kernel, 8 = they use it with gathered SELECT-list columns too, 9 = gathered int32
 * columns are streamed through it whatever the selectivity, 10 = no sampled selectivity estimate at query creation (the plan
 * then adapts from the first run's count on), 11 = survivor records are staged even when no predicate column is projected,
 * 12 = the plan made at creation stands whatever the cost model predicts (tests of one plan's kernels; P still adapts),
 * 14 = a `limit` query scans the whole segment in one launch instead of in chunks behind a limit-reached word (decided per run),
 * 15 = a small limit behind a limit scan takes k_scan + k_gather instead of the one fused launch (k_limit_gather),
 * 16 = one-launch projections use every CU even while a communicator whose collectives launch kernels is attached (default: one
 * CU per XCD is left to the collective's kernel), 17 = an aggregation's select chain runs as its own launch instead of inside the
 * aggregation launch, 19 = a projection through survivor records stores its bitmap in the staging launch (default: the bitmap is
 * materialised when imm3_query_bitmap asks), 200 + P = fixed tiles per range. */
int imm3_ctx_set_tuning(imm3_ctx *ctx, int32_t filter_variant, int32_t grid_blocks);

/* The projection planner's cost model (csrc/imm3_plan.h): predicted microseconds of one run's kernels under plan A (one launch), B
 * (survivor records) and C (the bitmap path) for a query shape -- predicate columns' widths and IN-list sizes, the SELECT list's
 * widths and which of its entries are predicate columns, the bytes of a survivor record -- with sigma = survivors per row, sloc =
 * survivors per row where there are survivors, full = share of the survivors in fully surviving stretches.  out_abc: 3 values. */
int imm3_plan_predict(int64_t n_rows, const int32_t *pred_width, const int32_t *pred_match, int32_t n_pred, const int32_t *proj_width,
                      const int32_t *proj_is_pred, int32_t n_proj, int32_t rec_bytes, double sigma, double sloc, double full, double *out_abc);

/* How the library planned a query (tests assert the path they mean to exercise; tools print it).
 * out[0] = 1 when an unlimited projection runs as ONE launch (k_filter_project: the filter kernel writes the rows), else 0;
 * out[1] = tiles per wave per span (P: lowered by the library once a getter has seen how many rows a run selected); out[2] = work-groups of that launch; out[3] = spans;
 * out[4] = 1 when the projection goes through survivor records in HBM (filter -> k_scan -> k_emit), else 0;
 * out[5] = dwords per survivor record; out[6] = 1 when the last run used the single-pass launch; out[7] = how often
 * imm3_query_run has waited for the device so far (an unreserved unlimited projection: once, on its first run);
 * out[8] / out[9] = single-pass runs of this query whose rows a getter had to gather from the bitmap because the launch gave up on
 * them: a look-back wait timed out (the query keeps the bitmap path from then on) / another launch of the kernel owned the device
 * (that run only); out[10] = runs whose small-limit gather (k_limit_gather) ran into its look-back poll cap, so that a getter
 * gathered the rows again with k_scan + k_gather.  n <= 11 values. */
int imm3_query_plan(const imm3_query *q, int64_t *out, int32_t n);

/* ---- fault injection into the single-pass projection kernel (k_filter_project, csrc/imm3_project.hip) ----
 * The kernel's work-groups wait on each other; what happens when such a wait does not resolve must be exercised on a device.
 * imm3_ctx_inject_fault: in every later single-pass launch of this context, work-group `work_group` never announces its
 * `span`-th span (so the round of spans it belongs to never completes), and look-back waits give up after `max_polls` polls
 * (0 = the shipped cap, ~0.1-0.2 s).  work_group = -1 switches the fault off.  Only the TOOLS' build of the library carries the
 * code (make -C immutable3_amd/csrc ablate -> lib/libimm3_ablate.so); the shipped library answers IMM3_ERR_STATE to anything
 * but "off".
 * imm3_ctx_debug_device_lock: overwrites the per-device ticket word that keeps two launches of the kernel from sharing the
 * device (0 = free) and returns what it held; with a foreign ticket in place every launch finds the device busy.  Synchronises
 * the context's stream first.  TOOLS' build only, like the fault injection (round 4 shipped it: any caller could have parked every
 * one-launch query of a device, in every context, on its fallback); the shipped library answers IMM3_ERR_STATE. */
/* imm3_comm_debug_standin (tools' build; the shipped library answers IMM3_ERR_STATE to anything but "off"): from now on every
 * imm3_comm_allreduce_count of this communicator first launches, on the communicator's stream, `work_groups` work-groups of a kernel
 * with the footprint of RCCL's all-reduce kernel (512 threads, 256 vector registers per lane, 37 664 bytes of LDS) that spin for
 * `spin_us` microseconds: what a multi-rank collective puts on the device beside the next pass's scans, measurable on one GPU
 * (tools/overlap_probe.py).  work_groups = 0 switches it off. */
/* imm3_plan_limit_scan: does a projection with a `limit` scan in chunks that stop once the limit is reached (1) or as one whole
 * select (0)?  The library's own decision (csrc/imm3_planner.cpp: limit_scan_applies) as a pure function of its inputs -- no device,
 * no handle: tests walk it. */
int imm3_plan_limit_scan(int32_t whole, int32_t count_log_on, int32_t count_in_scan, int64_t limit, int32_t single_tile_pass, int32_t table, int32_t records,
                         int32_t skip_bitmap, int32_t overlap_total, int32_t filter_variant, int64_t n_tiles);
struct imm3_comm;
int imm3_comm_debug_standin(struct imm3_comm *comm, int32_t work_groups, uint32_t spin_us);
int imm3_ctx_inject_fault(imm3_ctx *ctx, int32_t work_group, int32_t span, uint32_t max_polls);
int imm3_ctx_debug_device_lock(imm3_ctx *ctx, uint64_t value, uint64_t *previous);

#ifdef __cplusplus
}
#endif
#endif /* IMM3_DIAG_H */
