/*
 * imm3.h -- C ABI of the MI355X-native scan / filter / project path of immutable3.
 *
 * This is the drop-in boundary.  The reference (markosski/immutable3) has NO FFI: the path sits
 * behind Scala traits (engine/src/main/scala/immutabledb/engine/operator/Operator.scala:14-28)
 * and three factories wired in Engine.execute (engine/.../engine/Engine.scala:167-173):
 *     ScanOp.mkScanOp(sm, table)        engine/.../operator/Scan.scala:10-15
 *     SelectOp.mkSelectOp(col, cond)    engine/.../operator/Select.scala:5-12
 *     ProjectOp.mkProjectOp(cols,limit) engine/.../operator/Project.scala:8-15
 * The entry points below are what a JNI shim for GpuScanOp / GpuSelectOp / GpuProjectOp binds
 * (INTEGRATION.md shows the Scala + JNI side).  Plain pointers and sizes only; no torch types.
 *
 * Threading: the reference calls the path from a FixedThreadPool(cpuCount), one PipelineThread per segment
 * (Engine.scala:176-180, 247-262; SqlCli.scala:64).  The contract here (csrc/imm3_sync.h; tests/test_gpu_threads.py runs
 * it with 8 threads, tests/test_host_threads.py runs the host pieces under -fsanitize=thread):
 *   - every entry point may be called from any thread; there is no process-wide mutable state except the thread-local
 *     error text;
 *   - a CONTEXT may be used by any number of threads at once (the Scala binding gives every PipelineThread of a device
 *     the same one): its buffer pool, lazily made streams and diagnostics records are guarded inside; the work of all
 *     threads lands on the context's one stream in call order;
 *   - SEGMENTS and TABLES are immutable once created and may be read by any number of queries, threads and contexts of
 *     the same device (a query's context need not be the context that staged its segment);
 *   - ONE imm3_query / imm3_graph / imm3_comm is used by one thread at a time (a PipelineThread owns its iterator
 *     chain), and nothing else may still be inside a call on a handle that is being destroyed;
 *   - a graph capture is exclusive: imm3_ctx_capture_begin ... _end belong to ONE thread, and calls of other threads on
 *     that context wait meanwhile.
 * Worker threads that want their kernels to overlap on the device take one context each (a context = a stream) over the
 * shared segments.
 *
 * Errors: the reference throws Exception(msg) (Scan.scala:49, Select.scala:22,41,80,118,156).
 * Here every call returns an int status (0 = ok) and imm3_last_error() returns the message for
 * the calling thread; a JNI shim turns non-zero into ThrowNew(java/lang/Exception, msg).
 *
 * Citations: core/ = core/src/main/scala/immutabledb, engine/ = engine/src/main/scala/immutabledb.
 */
#ifndef IMM3_H
#define IMM3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMM3_ABI_VERSION 1

/* CodecType (core/codec/Codec.scala:21-24), in enumeration order.
 * IMM3_PFOR_INT: blocks as PFORCodecInt.encode writes them (core/codec/PFORCodec.scala:19-31: big-endian words of
 * JavaFastPFOR 0.1.10's IntegratedIntCompressor.compress + 8 zero bytes).  The reference dispatches the codec
 * (Column.scala:61, Scan.scala:37-39) but its decode throws on every block (PFORCodec.scala:43-50); this library
 * decodes the blocks on the GPU (csrc/imm3_codec.hip), a documented departure from the reference's failure. */
enum { IMM3_PFOR_INT = 0, IMM3_DENSE_INT = 1, IMM3_DENSE_TINYINT = 2, IMM3_DENSE_STRING = 3 };

/* EXTENSION (not CodecType values): columns whose blocks are what SnappyCodec.encode writes
 * (core/codec/SnappyCodec.scala:15-43: iq80 snappy 0.4 SnappyOutputStream framing -- "snappy\0", then per <= 32768
 * input bytes a flag, a 2-byte big-endian payload length, the masked CRC-32C of the input, and the stored or
 * raw-Snappy payload).  The reference defines the encoder only: `decode = ???` (SnappyCodec.scala:45), no CodecType
 * names the codec and nothing instantiates it, so a reference table cannot declare such a column; the ids sit outside
 * the enumeration's range on purpose.  Values decode to int32 / int8 / fixed-width strings exactly as the DENSE_*
 * codecs of the same type; blocks are decompressed on the GPU (csrc/imm3_snappy.hip), checksums verified. */
enum { IMM3_SNAPPY_INT = 16, IMM3_SNAPPY_TINYINT = 17, IMM3_SNAPPY_STRING = 18 };

/* SelectCondition (core/Query.scala:3-9) */
enum { IMM3_MATCH = 0, IMM3_NOTMATCH = 1, IMM3_EQ = 2, IMM3_GT = 3, IMM3_LT = 4, IMM3_NOOP = 5 };

/* status codes */
enum {
    IMM3_OK = 0,
    IMM3_ERR_UNSUPPORTED_CONDITION = 1, /* "Unsupported condition: ..."   Select.scala:22            */
    IMM3_ERR_UNSUPPORTED_VECTOR = 2,    /* "Unsupported column vector"    Select.scala:41,80,118,156 */
    IMM3_ERR_NO_CODEC = 3,              /* "No implementation for ..."    Scan.scala:49              */
    IMM3_ERR_LAYOUT = 4,                /* block tables the reference would fault on / mis-join      */
    IMM3_ERR_ARG = 5,                   /* bad handle / index / null pointer                         */
    IMM3_ERR_DEVICE = 6,                /* HIP runtime error (message carries hipGetErrorString)     */
    IMM3_ERR_STATE = 7                  /* result requested before imm3_query_run(); destroyed context */
};

typedef struct imm3_ctx imm3_ctx;         /* device + stream + scratch                                   */
typedef struct imm3_segment imm3_segment; /* one segment's columns resident in HBM (SegmentManager role) */
typedef struct imm3_table imm3_table;     /* all resident segments of one table: scanned by ONE launch    */
typedef struct imm3_query imm3_query;     /* one PipelineThread: ScanOp -> SelectOp* -> ProjectOp        */
typedef struct imm3_graph imm3_graph;     /* a recorded sequence of query runs (hipGraph)                */

/* One column of one segment as the reference stores it: `<col>_<id>.dat` bytes + the
 * `blockOffset` array of `<col>_<id>.meta` (core/storage/Segment.scala:33-58, 154-181;
 * core/storage/SegmentManager.scala:81-111) + the codec of core/Column.scala:18,57-63. */
typedef struct {
    int32_t codec;                /* IMM3_DENSE_INT / _TINYINT / _STRING / IMM3_PFOR_INT           */
    int32_t width;                /* bytes per value: 4, 1, or dtypeAttrs("size") for strings      */
    const void *dat;              /* host pointer to the (mmap'd) .dat bytes                       */
    uint64_t dat_bytes;
    const int32_t *block_offsets; /* N+1 byte offsets, first 0 (SegmentMeta.blockOffsets)          */
    int32_t n_offsets;
} imm3_column;

/* One SelectOp leaf (engine/.../operator/Select.scala:14-23).  Leaves are applied in array order,
 * which is the order PipelineThread.runOps composes them (Engine.scala:237-245); the AND/OR tag is
 * ignored there, so every tree is a conjunction. */
typedef struct {
    int32_t column;             /* index into the query's used-column list                           */
    int32_t cond;               /* IMM3_GT / IMM3_LT / IMM3_EQ / IMM3_MATCH (others -> error)         */
    double value;               /* GT/LT/EQ operand as the Query ADT carries it (core/Query.scala:6-8);
                                   narrowed per column type INSIDE the library: Int column d.toInt,
                                   TinyInt column d.toByte (Select.scala:65,73)                      */
    const uint8_t *match_bytes; /* MATCH: the IN-list values, concatenated                            */
    const int32_t *match_lens;  /* MATCH: byte length of each value                                   */
    int32_t n_match;
} imm3_select;

/* ---- library ---- */
int imm3_abi_version(void);
const char *imm3_last_error(void); /* thread-local; valid until the next failing call on this thread */
int imm3_device_count(int *count);

/* ---- context: device id + stream ---- */
/* stream: a hipStream_t to launch on, or NULL to create one (non-blocking).  NULL is also what HIP calls the legacy
 * default stream, so that stream cannot be named here: a caller that wants its own work ordered with the library's
 * passes a real stream (and enqueues on it), or synchronises through imm3_ctx_sync / imm3_query_sync.
 * Handles may be destroyed in any order: a context outlives, internally, the segments / tables / queries / comms made
 * from it, a segment the tables and queries that read it.  Calls through a handle whose context was destroyed fail
 * with IMM3_ERR_STATE. */
int imm3_ctx_create(int device, void *stream, imm3_ctx **out);
int imm3_ctx_destroy(imm3_ctx *ctx);
int imm3_ctx_sync(imm3_ctx *ctx);
int imm3_ctx_stream(imm3_ctx *ctx, void **stream_out);

/* ---- graphs: the reference's Engine starts one PipelineThread per segment for every query it executes
 * (Engine.scala:176-196); here a host that runs the same set of queries again and again -- the segments of a table, pass
 * after pass -- records their runs once and replays them with ONE call: a hipGraph of every kernel of those runs, which
 * removes the per-launch host cost and the gaps between the kernels.
 *   imm3_ctx_capture_begin(ctx); imm3_query_run(q0); imm3_query_run(q1); ...; imm3_ctx_capture_end(ctx, &g);
 *   imm3_graph_launch(g);   -- asynchronous on the context's stream, like the runs it stands for; results are read
 *                              through the queries as after imm3_query_run
 * Between begin and end the context accepts only imm3_query_run / imm3_query_run_select (nothing executes then: the
 * runs are recorded) -- every other call returns IMM3_ERR_STATE.  A recorded run must not need the host: an unlimited
 * projection has to have rows reserved (imm3_query_reserve_rows), and every recorded query must have run once before (its
 * buffers are allocated on first use).  Destroying a recorded query makes the graph stale (launch: IMM3_ERR_STATE). */
int imm3_ctx_capture_begin(imm3_ctx *ctx);
int imm3_ctx_capture_end(imm3_ctx *ctx, imm3_graph **out);
int imm3_graph_launch(imm3_graph *g);
int imm3_graph_destroy(imm3_graph *g);

/* ---- segment: what SegmentManager.getSegment(id, table, col) hands to ScanOp, for every column
 * of one segment id, staged into HBM once and kept resident (the reference keeps the mmaps for the
 * process lifetime, SegmentManager.scala:22,81-87).  The library owns the device copy; the host
 * buffers may be unmapped after the call returns. ---- */
int imm3_segment_create(imm3_ctx *ctx, const imm3_column *cols, int32_t ncols, imm3_segment **out);
/* same, but `dat` fields are DEVICE pointers that stay owned by the caller (no copy).  Each must be 16-byte aligned and
 * readable for 16 KiB past dat_bytes: the partial last tile of a column is read as a whole tile (1024 rows x width),
 * and the aggregation / gather kernels fetch the aligned dword around a narrow value. */
int imm3_segment_wrap_device(imm3_ctx *ctx, const imm3_column *cols, int32_t ncols, imm3_segment **out);
/* Asynchronous staging: returns as soon as the copies are enqueued on the context's copy stream (never the query stream:
 * staging a segment overlaps queries on the others); the host buffers must stay mapped until imm3_segment_wait()
 * returns (the reference's mmaps live as long as the SegmentManager).  Queries created on the segment order
 * themselves behind the copies by themselves. */
int imm3_segment_create_async(imm3_ctx *ctx, const imm3_column *cols, int32_t ncols, imm3_segment **out);
int imm3_segment_wait(imm3_segment *seg);
int imm3_segment_destroy(imm3_segment *seg);
int imm3_segment_bytes(const imm3_segment *seg, uint64_t *device_bytes);

enum { IMM3_AGG_COUNT = 0, IMM3_AGG_MIN = 1, IMM3_AGG_MAX = 2 };
typedef struct {
    int32_t kind;
    int32_t column;
} imm3_aggregate;

/* ---- table: every segment of one table (SegmentManager.getSegments, SegmentManager.scala:108-111) as ONE scan unit.
 * The reference fans out one PipelineThread per segment (Engine.scala:176-180) and merges on the consumer thread;
 * a README-style table (block 1024 x segment 1000) has ~1 M rows per segment, so 100 M rows are ~98 segments and a
 * per-segment launch sequence is launch-bound.  A table query runs the same fused kernels over a TILE TABLE (per
 * 1024-row tile: valid rows + one pointer per column) that spans all segments: one scan+select launch, one
 * compaction, rows / groups in ascending (segment, row) order -- exactly what Engine.execute returns.
 * Requires every segment to have the uniform layout (all columns cut into the same blocks, every non-final block
 * a multiple of 64 rows); otherwise IMM3_ERR_LAYOUT and the caller falls back to per-segment queries.
 * The segments stay owned by the caller and must outlive the table. ---- */
int imm3_table_create(imm3_ctx *ctx, const imm3_segment *const *segs, int32_t n_segs, imm3_table **out);
int imm3_table_destroy(imm3_table *t);
/* Table flavours of imm3_query_create / imm3_query_create_agg (same arguments, `seg` replaced by `table`).
 * Layout getters cover all segments in order (batches of segment 0, then segment 1, ...; oid restarts per segment);
 * imm3_query_segment_starts gives, per segment, its first batch and its first word in the bitmap. */
int imm3_query_create_table(imm3_ctx *ctx, const imm3_table *table,
                            const int32_t *used_cols, int32_t n_used,
                            const imm3_select *sels, int32_t n_sels,
                            const int32_t *proj, int32_t n_proj, int64_t limit,
                            int32_t table_block_size, imm3_query **out);
int imm3_query_create_table_agg(imm3_ctx *ctx, const imm3_table *table,
                                const int32_t *used_cols, int32_t n_used,
                                const imm3_select *sels, int32_t n_sels,
                                const int32_t *group_cols, int32_t n_group,
                                const imm3_aggregate *aggs, int32_t n_aggs,
                                int32_t table_block_size, imm3_query **out);
/* n_segments + 1 entries each (last = totals); any pointer may be NULL */
int imm3_query_segment_starts(const imm3_query *q, int32_t *n_segments, int32_t *first_batch, int64_t *first_word);
/* Row indices of a TABLE query are virtual (tile * 1024 + position); this maps them to (segment, row in segment). */
int imm3_query_locate_rows(const imm3_query *q, const uint32_t *row_index, uint64_t n, uint32_t *segment_out, uint32_t *row_out);

/* ---- query: one PipelineThread (Engine.scala:235-262) over one segment ----
 *   used_cols   indices into the segment's columns, in Engine.getColumns order (Engine.scala:85-106);
 *               the FIRST one defines the batches (Scan.scala:55,72)
 *   sels        SelectOp leaves in application order (may be empty: NoSelect)
 *   proj        indices into used_cols of the SELECT-list columns in SELECT-list order
 *               (Project.scala:55-57); n_proj == 0 -> no ProjectOp (bitmap/count only)
 *   limit       Project limit; <= 0 = unlimited (Project.scala:73-80)
 *   table_block_size  Table.blockSize, only used for oid = vecCounter * blockSize (Scan.scala:60)
 * Validation mirrors the reference: NotMatch/NoOp always fail (SelectOp.iterator, Select.scala:22);
 * a condition on the wrong vector type or an unknown codec fails iff the segment has >= 1 batch. */
int imm3_query_create(imm3_ctx *ctx, const imm3_segment *seg,
                      const int32_t *used_cols, int32_t n_used,
                      const imm3_select *sels, int32_t n_sels,
                      const int32_t *proj, int32_t n_proj, int64_t limit,
                      int32_t table_block_size, imm3_query **out);
int imm3_query_destroy(imm3_query *q);

/* ---- group-by aggregation: ProjectAggOp (engine/.../operator/ProjectAggregate.scala:115-227) over the rows the
 * SelectOps keep.  CountAggr / MinDoubleAggr / MaxDoubleAggr / MaxStringAggr (:22-112).
 *   group_cols  indices into used_cols, in the order their values are joined into the group key (the reference
 *               joins them in batch-column order with "_", :151-156); total width <= 8 bytes on this path
 *   aggs        {kind, column (index into used_cols)}; MIN/MAX on INT/TINYINT, MAX on STRING (<= 8 bytes), COUNT on any
 * Groups come back in first-seen order (the reference's LinkedHashMap order): ascending first selected row. ---- */
int imm3_query_create_agg(imm3_ctx *ctx, const imm3_segment *seg,
                          const int32_t *used_cols, int32_t n_used,
                          const imm3_select *sels, int32_t n_sels,
                          const int32_t *group_cols, int32_t n_group,
                          const imm3_aggregate *aggs, int32_t n_aggs,
                          int32_t table_block_size, imm3_query **out);
int imm3_query_group_count(imm3_query *q, uint32_t *n_groups);
/* The shape of an aggregation query's result rows: group columns, aggregates (the stride of `vals` in imm3_query_fetch_groups and
 * imm3_comm_merge_groups[_all]), bytes of the packed group key.  A binding sizes its buffers from THIS, not from what its caller
 * believes (IMM3_ERR_ARG for a query that is not an aggregation). */
int imm3_query_agg_shape(const imm3_query *q, int32_t *n_group_cols, int32_t *n_aggs, int32_t *key_bytes);
/* keys: the group columns' raw bytes packed little-endian in group_cols order; first_row: lowest selected row of
 * the group; counts: selected rows of the group; vals[g * n_aggs + j]: COUNT -> the count, MIN/MAX numeric -> the
 * int32 value sign-extended, MAX string -> the value's bytes packed big-endian.  Sorted by first_row. */
int imm3_query_fetch_groups(imm3_query *q, uint64_t *keys, uint32_t *first_row, uint64_t *counts, int64_t *vals,
                            uint32_t max_groups);

/* Pre-size the projected-row buffers so that not even the FIRST imm3_query_run() has to wait for the count.  Without it an
 * unlimited projection sizes them once: a query whose SELECT list is predicate columns only (it runs as ONE launch that
 * writes the rows itself) gets room for every row of the segment; any other one synchronises on its first run to read the
 * count and keeps the arrays (with an eighth of headroom) for every later run -- a steady-state projecting query never
 * synchronises.  A run that outgrows the arrays, reserved or not, is detected when its rows are fetched and emitted again. */
int imm3_query_reserve_rows(imm3_query *q, uint64_t rows);

/* Enqueue the whole pipeline on the context's stream.  n_proj == 0: the scan+select kernel (selection bitmap + count).
 * Unlimited projection whose SELECT list is predicate columns only (one of them an int32 column, or >= 30 % of the rows
 * surviving), one uniform segment: ONE launch -- scan + select +
 * project (csrc/imm3_project.hip: the filter kernel writes the rows in ascending order itself).  Otherwise: scan+select
 * (staging a record per survivor when a predicate column is projected), then an offsets scan and compact+gather -- unless enough rows survive for the other SELECT-list columns to be streamed
 * through that one launch as well (dense int32 columns, from 4 % survivors on; no string predicate): decided from a sample
 * counted at query creation (segments of 4 M rows and more), a reservation, or the first run's count.  Asynchronous except for the first run of an unreserved unlimited projection
 * (see above).  May be called repeatedly on the same query. */
int imm3_query_run(imm3_query *q);
/* Only the ScanOp -> SelectOp* part (selection bitmap + count). */
int imm3_query_run_select(imm3_query *q);
/* ScanOp -> SelectOp* for the count alone: `vec.selected.size` summed over the batches (what a count(*)-style consumer of
 * the pipeline reads, Engine.scala:190-196) without materialising the BitSets.  A select chain that is one fused launch then
 * stores no bitmap at all -- on a narrow column the 1/8 byte per row of bitmap is a fifth of the kernel -- and
 * imm3_query_bitmap fails with IMM3_ERR_STATE until the next full run; any other chain runs as imm3_query_run_select. */
int imm3_query_run_count(imm3_query *q);
int imm3_query_sync(imm3_query *q);
/* The selected-row count of a select-only run is reduced on the context's auxiliary stream so that it overlaps the
 * next scan.  Host getters wait for it by themselves; a DEVICE consumer of imm3_query_device_ptr(q, 1) that runs on
 * the context's main stream calls this first: it makes the main stream wait (stream-side, no host block).  After a projection
 * with a `limit` whose scan stopped early (imm3_query_run on a limit query) the count word holds the scanned prefix's count: this
 * call then runs the whole select first, like imm3_query_count and imm3_comm_allreduce_count, so that what the device consumer reads
 * behind it is the segment's count. */
int imm3_query_join_count(imm3_query *q);

/* Device-side log of the selected-row count of every later run of this query: run k (counted from this call) stores its
 * count at device_log[k] while k < capacity, from the kernel that produces the count -- no copy kernel, no host call
 * per run.  For consumers that reduce many runs' counts in one collective (bench.py's count all-reduce over RCCL).
 * device_log = NULL switches the log off. */
int imm3_query_log_counts(imm3_query *q, uint64_t *device_log, uint64_t capacity);

/* ---- results ---- */
/* Batches as ScanOp yields them (FilledColumnVectorBatch, core/DataVector.scala:24-31): */
int imm3_query_layout(const imm3_query *q, int32_t *n_batches, int64_t *total_words, int64_t *n_rows);
/* per batch: size (rows), oid, and the offset of its BitSet words in the batch-major bitmap
 * (ceil(size/64) words per batch; bit i of a batch <-> word i>>6, bit i&63 = mutable.BitSet) */
int imm3_query_batches(const imm3_query *q, int32_t *batch_size, int32_t *batch_oid, int64_t *batch_word_off);
int imm3_query_count(imm3_query *q, uint64_t *selected_rows);               /* sum of selected.size (the whole segment's, also behind a run that stopped at its limit) */
int imm3_query_bitmap(imm3_query *q, uint64_t *words_out, int64_t n_words); /* device -> host copy     */
/* Rows ProjectOp emits for this segment, in emission order (batch order, ascending position): */
int imm3_query_row_count(imm3_query *q, uint64_t *rows);
/* row_index_out: segment-global row number of each emitted row (may be NULL);
 * col_out[j]: packed values of projected column j, width bytes per row (int32 LE / int8 / raw bytes) */
int imm3_query_fetch_rows(imm3_query *q, uint32_t *row_index_out, void *const *col_out, uint64_t max_rows);

/* Device-resident results for callers that stay on the GPU (RCCL count reduce, chained kernels).
 * which: 0 = bitmap (uint64 words), 1 = total count (one uint64), 2 = row indices (uint32),
 *        3 = emitted row count (one uint64), 4 = status word (one uint64), 16+j = projected column j.
 * What a device-side consumer may rely on behind a run, without any host getter in between:
 *   - bitmap (0), count (1) and emitted row count (3) are exact after EVERY run.  In particular the one-launch projection
 *     (csrc/imm3_project.hip), whose work-groups wait on each other, degrades to "count + bitmap" when such a wait does not
 *     resolve or when another launch of that kernel owns the device: the counts that meet in imm3_comm_allreduce_count
 *     (Engine.scala:190-196) are right whatever happened to the rows;
 *   - row indices (2) and projected columns (16+j) of a one-launch projection are complete only if the status word (4) shows
 *     neither bit 1 (a prefix never came) nor bit 2 (device busy) FOR THAT RUN: bits 8..31 of the word are the run's tag,
 *     (run counter - 1) & 0xFFFFFF with the run counter at count pointer + 8 words.  The host getters (imm3_query_row_count /
 *     _fetch_rows) check this themselves and gather the rows from the bitmap when needed; a consumer that reads the row arrays
 *     on the device calls imm3_query_row_count first (it waits for the run) or checks the word.  Pointers 2 and 16+j change
 *     when the output arrays grow (a reservation that was too small): fetch them again after imm3_query_row_count;
 *   - a query with limit > 0 stops scanning once `limit` rows are selected (ProjectIterator.hasNext, Project.scala:73-80: the
 *     reference never looks at batches behind the limit either): behind imm3_query_run its bitmap (0) and count (1) cover the
 *     scanned prefix of the segment only -- the rows (2, 3, 16+j) are complete.  imm3_query_count, imm3_query_bitmap,
 *     imm3_query_run_select / _run_count, imm3_query_log_counts and imm3_comm_allreduce_count give the whole segment's: the host
 *     getters run the whole select when the last run stopped early;
 *   - the BITMAP (0) is written by imm3_query_run_select always, and by imm3_query_run whenever the plan reads it; two plans do
 *     not: an unlimited projection that goes through survivor records (the records carry the positions; imm3_query_plan out[4])
 *     and an aggregation whose select chain rides in the aggregation launch.  imm3_query_bitmap materialises it on demand (the
 *     select chain runs once more); a DEVICE consumer of pointer 0 calls imm3_query_run_select.  The COUNT (1) is exact behind
 *     every imm3_query_run -- except for such an aggregation, where imm3_query_join_count / imm3_query_count produce it. */
int imm3_query_device_ptr(imm3_query *q, int32_t which, void **ptr);

/* ---- multi-GPU: the count reduce.  Segments shard one per GPU -- segment s belongs to GPU s mod G -- and
 * every GPU runs its own PipelineThreads (Engine.scala:176-180); bitmaps, row lists and oids stay where they were
 * produced.  The ONE exchange is the selected-row count, summed over the GPUs with one 8-byte
 * ncclAllReduce(sum, ncclUint64) over RCCL / xGMI, ordered behind the scans that produce the counts.
 * RCCL is bound at run time (librccl.so.1, or the library the environment variable IMM3_RCCL_LIB names -- the test suite's
 * two-rank loopback transport); a single-GPU host never needs it.  While a communicator of more than one rank is attached to a
 * context, that context's one-launch projections leave one CU per XCD to the collective's kernel.
 *   one process per GPU : rank 0 calls imm3_comm_unique_id, the host hands the 128 bytes to every rank (any channel:
 *                         a file, a socket, a key-value store), every rank calls imm3_comm_create
 *   one process, G GPUs : imm3_comm_create_all over one context per device (the JVM host's shape: one Engine, one
 *                         GpuSegmentManager per device); collectives through imm3_comm_allreduce_count_all ---- */
#define IMM3_COMM_ID_BYTES 128
typedef struct imm3_comm imm3_comm;
int imm3_comm_unique_id(uint8_t *id_out /* IMM3_COMM_ID_BYTES */);
int imm3_comm_create(imm3_ctx *ctx, int32_t world, int32_t rank, const uint8_t *id, imm3_comm **out);
int imm3_comm_create_all(imm3_ctx *const *ctxs, int32_t n, imm3_comm **out /* n handles */);
int imm3_comm_destroy(imm3_comm *c);
int imm3_comm_info(const imm3_comm *c, int32_t *world, int32_t *rank);
/* Streams: a collective runs on the communicator's OWN stream.  It starts after everything enqueued so far on the
 * context's stream, and the context's stream does not wait for it: the next pass's scans overlap the message.
 * imm3_comm_sync blocks the host until the communicator's stream is idle; imm3_comm_join makes the context's stream
 * wait for the last collective (stream side, no host block) -- for a device consumer of the reduced words. */
int imm3_comm_sync(imm3_comm *c);
int imm3_comm_join(imm3_comm *c);
/* In-place sum of n device words over the ranks (asynchronous). */
int imm3_comm_allreduce_u64(imm3_comm *c, uint64_t *device_buf, uint64_t n);
/* The selected-row counts of this rank's queries (each already run on the communicator's context) are summed on the
 * device, then all-reduced over the ranks.  device_out: where the global count lands (a device word; NULL = a word the
 * communicator owns); host_out: if not NULL the call waits and also returns the value to the host. */
int imm3_comm_allreduce_count(imm3_comm *c, imm3_query *const *queries, int32_t n_queries, uint64_t *device_out, uint64_t *host_out);
/* Single-process flavour: comms[i] drives device i with its queries[i][0 .. n_queries[i]); one thread issues all G
 * collectives inside a group.  host_out (may be NULL) receives the global count. */
int imm3_comm_allreduce_count_all(imm3_comm *const *comms, int32_t n_comms, imm3_query *const *const *queries,
                                  const int32_t *n_queries, uint64_t *host_out);

/* ---- multi-GPU: the group-by merge.  ProjectAggregateQueueOp (engine/.../operator/ProjectAggregateQueue.scala:9-55) merges
 * the per-segment group maps by key: counts add, max / min combine, the first arrival keeps its place.  Here every rank
 * brings the aggregation queries of ITS segments (each already run; same group columns and aggregates everywhere, at least
 * one per rank), segment_index[i] = the global index of the segment queries[i] ran on, and every rank receives the merged
 * table in first-seen order -- ascending (segment, first selected row): the reference's arrival order is a race, this is
 * the order Engine.execute would produce with one thread.  Keys of <= 2 bytes travel as element-wise all-reduces of a
 * direct-indexed table (sum on counts, min on segment << 32 | row, max / min on each aggregate); wider keys are merged in a
 * device hash table per rank, exchanged as an ncclAllGather of the ranks' group lists and merged again on the device.  One rank:
 * the same merge over that rank's queries, no collective.
 *   keys / counts / vals as imm3_query_fetch_groups; first[g] = segment << 32 | first selected row of the group there.
 * Synchronous (the merged table is returned to the host). */
int imm3_comm_merge_groups(imm3_comm *c, imm3_query *const *queries, const int32_t *segment_index, int32_t n_queries,
                           uint64_t *keys, uint64_t *first, uint64_t *counts, int64_t *vals, uint32_t max_groups, uint32_t *n_groups);
/* Single-process flavour (comms from imm3_comm_create_all): comms[i] brings queries[i][0 .. n_queries[i]) with
 * segment_index[i][...]; the devices' tables are merged inside this process. */
int imm3_comm_merge_groups_all(imm3_comm *const *comms, int32_t n_comms, imm3_query *const *const *queries,
                               const int32_t *const *segment_index, const int32_t *n_queries,
                               uint64_t *keys, uint64_t *first, uint64_t *counts, int64_t *vals, uint32_t max_groups, uint32_t *n_groups);

/* ---- write side of the PFOR_INT codec (host code, no device involved) ----
 * PFORCodecInt.encode (core/codec/PFORCodec.scala:19-31), which SegmentWriter.flush applies to each block of a PFOR_INT
 * column (core/storage/Segment.scala:115-122).  imm3_pfor_encode_block: one block of n_values little-endian int32 ->
 * the bytes appended to the .dat.  imm3_pfor_encode_column: a whole column cut into blocks of block_rows values (the
 * last one shorter); offsets_out receives the n_blocks + 1 byte offsets of the .meta blockOffset table. */
uint64_t imm3_pfor_encode_bound(int32_t n_values);
int imm3_pfor_encode_block(const int32_t *values, int32_t n_values, void *out, uint64_t cap, uint64_t *bytes_out);
int imm3_pfor_encode_column(const int32_t *values, uint64_t n_values, int32_t block_rows, void *out, uint64_t cap,
                            int32_t *offsets_out, uint64_t *bytes_out);

/* Write side of the snappy block format (host code): SnappyCodec.encode of one storage block's raw value bytes. */
uint64_t imm3_snappy_encode_bound(uint64_t n_bytes);
int imm3_snappy_encode_block(const void *bytes, uint64_t n_bytes, void *out, uint64_t cap, uint64_t *bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* IMM3_H */
