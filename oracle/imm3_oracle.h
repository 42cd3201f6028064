/*
 * imm3_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never on the product path).
 *
 * A plain-C restatement of the markosski/immutable3 scan/filter/project hot path
 * (ScanOp -> SelectOp* -> ProjectOp over DENSE_INT / DENSE_TINYINT / DENSE_STRING
 * segments).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library.
 *
 * PARITY PINNING: the reference ships NO tests, fixtures or golden vectors and cannot
 * be run here (Scala 2.12 / JVM; no JDK in this image or on the GPU box).  At the
 * reference boundary parity is therefore UNPINNED BY REFERENCE TESTS.  The oracle is
 * pinned instead by (i) the hand-derived known-answer vectors of SURVEY.md Appendix B
 * (tests/test_oracle_kat.py), each traceable to a cited reference line, and (ii) a second,
 * independent numpy restatement (oracle/oracle_np.py) that must agree bit-for-bit on
 * randomised tables (tests/test_oracle_cross.py).
 *
 * All citations are path:line relative to the reference checkout
 * (core/ = core/src/main/scala/immutabledb, engine/ = engine/src/main/scala/immutabledb).
 */
#ifndef IMM3_ORACLE_H
#define IMM3_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* codec ids -- core/codec/Codec.scala:21-24 (CodecType enumeration order) */
enum { IMM3O_PFOR_INT = 0, IMM3O_DENSE_INT = 1, IMM3O_DENSE_TINYINT = 2, IMM3O_DENSE_STRING = 3 };

/* select conditions -- core/Query.scala:3-9 */
enum { IMM3O_MATCH = 0, IMM3O_NOTMATCH = 1, IMM3O_EQ = 2, IMM3O_GT = 3, IMM3O_LT = 4, IMM3O_NOOP = 5 };

/* status codes (the reference throws Exception(msg); we return a code + message) */
enum {
    IMM3O_OK = 0,
    IMM3O_ERR_UNSUPPORTED_CONDITION = 1, /* engine/engine/operator/Select.scala:22  */
    IMM3O_ERR_UNSUPPORTED_VECTOR = 2,    /* Select.scala:41,80,118,156               */
    IMM3O_ERR_NO_CODEC = 3,              /* engine/engine/operator/Scan.scala:49     */
    IMM3O_ERR_INDEX = 4,                 /* JVM ArrayIndexOutOfBounds equivalents    */
    IMM3O_ERR_ARG = 5
};

/* One used column of one segment: the mmap'd .dat bytes and the .meta blockOffset table
 * (core/storage/Segment.scala:33, 154-181; core/storage/SegmentManager.scala:81-111). */
typedef struct {
    const uint8_t *dat;
    uint64_t dat_bytes;
    const int32_t *block_offsets; /* N+1 entries, first 0 */
    int32_t n_offsets;
    int32_t codec;  /* IMM3O_DENSE_* */
    int32_t width;  /* bytes per value: 4, 1, or dtypeAttrs("size") for strings */
} imm3o_column;

/* One SelectOp leaf, in the order PipelineThread.runOps composes them
 * (engine/engine/Engine.scala:237-245: PNode(n1,n2,_) => rec(n2) o rec(n1); AND/OR tag ignored). */
typedef struct {
    int32_t column;  /* index into the used-column list (SelectOp looks the column up by name, Select.scala:60) */
    int32_t cond;    /* IMM3O_GT / LT / EQ / MATCH / ... */
    double value;    /* GT/LT/EQ operand, narrowed per column type at evaluation (Select.scala:65,73) */
    const uint8_t *match_bytes; /* MATCH: concatenated IN-list values */
    const int32_t *match_lens;  /* MATCH: byte length of each value */
    int32_t n_match;
} imm3o_select;

/* ---- scalar rules ---- */
int32_t imm3o_bytes_to_int(const uint8_t b[4]);      /* core/util/Conversions.scala:17-24 */
void imm3o_int_to_bytes(int32_t v, uint8_t out[4]);  /* core/DataType.scala:40-47         */
int32_t imm3o_d2i(double d);                         /* Scala Double.toInt  (JVM d2i)     */
int8_t imm3o_d2b(double d);                          /* Scala Double.toByte (d2i then i2b) */

/* ---- layout of the batches a ScanOp over `first` yields (Scan.scala:28-72) ---- */
/* number of batches = blockOffsets.size - 1 (Segment.scala:172-179) */
int32_t imm3o_n_batches(const imm3o_column *first);
/* Fills per-batch size (rows of the FIRST used column's block, Scan.scala:55), oid
 * (vecCounter * table.blockSize, Scan.scala:60) and the word offset of each batch's BitSet in the
 * batch-major concatenated bitmap (ceil(size/64) words per batch).  Returns total words. */
int64_t imm3o_layout(const imm3o_column *first, int32_t table_block_size,
                     int32_t *batch_size, int32_t *batch_oid, int64_t *batch_word_off);

/*
 * Runs ScanOp -> SelectOp* for ONE segment and returns the per-batch selection BitSets
 * (uint64 words, bit i of a batch <-> word i>>6, bit i&63: scala.collection.mutable.BitSet).
 *   cols[0..ncols)   used columns in Engine.getColumns order (engine/engine/Engine.scala:85-106)
 *   sels[0..nsels)   SelectOp leaves in application order
 *   words_out        batch-major bitmap, imm3o_layout() words
 *   count_out        total selected rows (sum of BitSet.size)
 *   flavour          0 = "faithful" (per-block copy, per-element chunk read into a growable
 *                        buffer + toArray, per-row BitSet.add / BitSet.remove calls: the cost
 *                        structure of DenseCodec.scala:37-73 / Scan.scala:55-57 / Select.scala:67-70),
 *                    1 = "tight"   (same results, no per-element allocation; labelled as such)
 * Returns IMM3O_OK or an error code; msg (>=128 bytes) receives the reference's exception text.
 */
int imm3o_scan_select(const imm3o_column *cols, int32_t ncols,
                      const imm3o_select *sels, int32_t nsels,
                      int32_t table_block_size, int32_t flavour,
                      uint64_t *words_out, uint64_t *count_out, char *msg);

/*
 * ProjectOp over the batches produced above (engine/engine/operator/Project.scala:37-80).
 * Emits, in batch order and ascending in-batch position, one row per set bit until `limit`
 * rows (limit > 0) have been produced.
 *   proj[0..nproj)   indices into cols[] of the SELECT-list columns, in SELECT-list order (Project.scala:55-57)
 *   out_batch/out_pos  per emitted row: batch index and in-batch position
 *   out_vals[j]      per projected column j: packed values, width[cols[proj[j]]] bytes per row
 *   cap_rows         capacity of the output arrays (rows)
 *   *would_throw     set to 1 if the reference ProjectIterator would have hit its
 *                    zero-survivor-batch bug (Project.scala:39-57) before finishing;
 *                    the oracle itself skips empty batches (operator-level parity definition).
 * Returns number of rows emitted, or -(error code).
 */
int64_t imm3o_project(const imm3o_column *cols, int32_t ncols,
                      const int32_t *proj, int32_t nproj, int64_t limit,
                      int32_t table_block_size, const uint64_t *words,
                      int32_t *out_batch, int32_t *out_pos, uint8_t *const *out_vals,
                      int64_t cap_rows, int32_t *would_throw);

/*
 * ProjectAggOp (engine/engine/operator/ProjectAggregate.scala:115-227) over the batches produced above: count / min /
 * max per group, groups in first-seen order (LinkedHashMap).  The C twin of oracle_np.project_agg.
 *   group[0..ngroup)  indices into cols[] of the group-by columns, in the order their values are joined with "_"
 *   aggs[0..naggs)    {kind: 0 count, 1 min, 2 max; column: index into cols[]}
 *   keys_out          group g's key string at keys_out + g * key_stride (NUL-terminated)
 *   count_out / num_out / str_out   [g * naggs + a]: CountAggr counter / Min-MaxDoubleAggr value (-/+Double.MaxValue when
 *                     untouched) / MaxStringAggr value at str_out + (g * naggs + a) * str_stride ("" = unset)
 * Returns the number of groups, or -(error code) (msg receives the reference's exception text).
 */
typedef struct {
    int32_t kind;
    int32_t column;
} imm3o_aggregate;
int64_t imm3o_project_agg(const imm3o_column *cols, int32_t ncols, const int32_t *group, int32_t ngroup,
                          const imm3o_aggregate *aggs, int32_t naggs, const uint64_t *words,
                          char *keys_out, int32_t key_stride, int64_t *count_out, double *num_out,
                          char *str_out, int32_t str_stride, int64_t max_groups, char *msg);

/* ---- PFOR_INT block codec (imm3_oracle_pfor.c; core/codec/PFORCodec.scala:19-31 + JavaFastPFOR 0.1.10) ----
 * The block format is defined by the reference's ENCODER; the reference's decoder is broken (PFORCodec.scala:43-50),
 * so decode is the inverse the encoder implies.  Parity unpinned (third-party library, no vectors offline). */
int64_t imm3o_pfor_encode_bound(int32_t n);
int64_t imm3o_pfor_encode_block(const int32_t *vals, int32_t n, uint8_t *out, int64_t cap);
int32_t imm3o_pfor_block_count(const uint8_t *blk, int64_t len);
int32_t imm3o_pfor_decode_block(const uint8_t *blk, int64_t len, int32_t *out, int32_t cap);

/* ---- snappy-coded blocks (imm3_oracle_snappy.c; core/codec/SnappyCodec.scala:14-43 + iq80 snappy 0.4) ----
 * The block format is what SnappyCodec.encode writes (SnappyOutputStream framing around raw Snappy); the reference
 * has no decoder (`???`) and no CodecType for it.  Raw-Snappy layer pinned against pyarrow's Google snappy. */
uint32_t imm3o_crc32c(const uint8_t *p, int64_t n);
uint32_t imm3o_crc32c_masked(const uint8_t *p, int64_t n);
int64_t imm3o_snappy_raw_bound(int64_t n);
int64_t imm3o_snappy_raw_encode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap);
int64_t imm3o_snappy_raw_uncompressed_length(const uint8_t *in, int64_t n);
int64_t imm3o_snappy_raw_decode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap);
int64_t imm3o_snappy_block_bound(int64_t n);
int64_t imm3o_snappy_block_encode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap);
int64_t imm3o_snappy_block_length(const uint8_t *blk, int64_t n);
int64_t imm3o_snappy_block_decode(const uint8_t *blk, int64_t n, uint8_t *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
