/*
 * imm3_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see imm3_oracle.h for the pinning statement:
 * parity is UNPINNED by reference tests because the reference has none; pinned by SURVEY Appendix B
 * known-answer vectors + an independent numpy restatement).
 *
 * Restates, function by function, the reference's per-segment pipeline.  Citations are
 * path:line in the reference checkout (core/ = core/src/main/scala/immutabledb,
 * engine/ = engine/src/main/scala/immutabledb).  Nothing here is copied: the reference is Scala.
 */
#include "imm3_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NOINLINE __attribute__((noinline))

/* ------------------------------------------------------------------------------------------
 * Scalar rules
 * ---------------------------------------------------------------------------------------- */

/* core/util/Conversions.scala:17-24.  Scala's `+` binds tighter than `<<`, so each line is
 * result = (result + (b & 0xFF)) << 8 : a Horner evaluation, most significant byte (index 3)
 * first, i.e. little-endian two's complement.  Int arithmetic wraps. */
int32_t imm3o_bytes_to_int(const uint8_t b[4]) {
    uint32_t result = 0;
    result = (result + (uint32_t)(b[3] & 0xFF)) << 8;
    result = (result + (uint32_t)(b[2] & 0xFF)) << 8;
    result = (result + (uint32_t)(b[1] & 0xFF)) << 8;
    result = result + (uint32_t)(b[0] & 0xFF);
    return (int32_t)result;
}

/* core/DataType.scala:40-47 (IntType.valueToBytes): arithmetic shift, mask, byte 0 = LSB. */
void imm3o_int_to_bytes(int32_t v, uint8_t out[4]) {
    out[3] = (uint8_t)((v >> 24) & 0xFF);
    out[2] = (uint8_t)((v >> 16) & 0xFF);
    out[1] = (uint8_t)((v >> 8) & 0xFF);
    out[0] = (uint8_t)(v & 0xFF);
}

/* Scala Double.toInt == JVM d2i: NaN -> 0, saturate to [INT_MIN, INT_MAX], else truncate
 * toward zero.  Used at engine/engine/operator/Select.scala:65,103,141. */
int32_t imm3o_d2i(double d) {
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t)d; /* C truncates toward zero; in range here */
}

/* Scala Double.toByte == d2i then i2b (low 8 bits, sign-extended).  Select.scala:73,111,149. */
int8_t imm3o_d2b(double d) {
    uint32_t i = (uint32_t)imm3o_d2i(d);
    return (int8_t)(uint8_t)(i & 0xFFu);
}

/* ------------------------------------------------------------------------------------------
 * scala.collection.mutable.BitSet restated over a fixed word array (bit i <-> word i>>6, bit i&63).
 * The "faithful" flavour goes through these out-of-line calls once per row like the reference
 * (Scan.scala:57 bitSet.add(x); Select.scala:68 selected.remove(i)).
 * ---------------------------------------------------------------------------------------- */
static NOINLINE void bitset_add(uint64_t *w, int32_t i) { w[i >> 6] |= (1ULL << (i & 63)); }
static NOINLINE void bitset_remove(uint64_t *w, int32_t i) { w[i >> 6] &= ~(1ULL << (i & 63)); }
static inline int bitset_contains(const uint64_t *w, int32_t i) { return (int)((w[i >> 6] >> (i & 63)) & 1ULL); }
static int64_t bitset_size(const uint64_t *w, int64_t nwords) {
    int64_t c = 0;
    for (int64_t k = 0; k < nwords; k++) c += __builtin_popcountll(w[k]);
    return c;
}

/* ------------------------------------------------------------------------------------------
 * Segment.BlockIterator (core/storage/Segment.scala:158-180)
 * next: allocate blockOffsets(k+1)-blockOffsets(k) bytes and fill them with a RELATIVE get from
 * the shared buffer, i.e. from a cursor that starts at 0 (rewind at :159) and advances by each
 * block's length -- not from blockOffsets(k) itself.  The two coincide when blockOffsets(0)==0,
 * which the writer guarantees (Segment.scala:91-92).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const imm3o_column *col;
    int32_t position; /* block index */
    uint64_t cursor;  /* byte cursor of the relative get */
} block_iter;

static void block_iter_init(block_iter *it, const imm3o_column *c) {
    it->col = c;
    it->position = 0;
    it->cursor = 0;
}
static int block_iter_has_next(const block_iter *it) { return it->position < it->col->n_offsets - 1; }
/* returns length or -1 on BufferUnderflow / negative size */
static int64_t block_iter_peek(const block_iter *it, const uint8_t **start) {
    int64_t len = (int64_t)it->col->block_offsets[it->position + 1] - (int64_t)it->col->block_offsets[it->position];
    if (len < 0) return -1;                                        /* NegativeArraySizeException */
    if (it->cursor + (uint64_t)len > it->col->dat_bytes) return -1; /* BufferUnderflowException   */
    *start = it->col->dat + it->cursor;
    return len;
}
static void block_iter_advance(block_iter *it, int64_t len) {
    it->cursor += (uint64_t)len;
    it->position += 1;
}

/* ------------------------------------------------------------------------------------------
 * DenseCodec*.decode (core/codec/DenseCodec.scala:37-45, 51-59, 65-73):
 *     val chunk = new Array[Byte](dtype.size)
 *     while (data.read(chunk) != -1) segment += dtype.bytesToValue(chunk)
 * ByteArrayInputStream.read(b) returns -1 only when NO byte is left; a short final read leaves
 * the tail of `chunk` holding the previous element's bytes (zeros for the very first), and still
 * appends a value.  => n = ceil(bytes/width) values.  (Malformed input; SURVEY A.1 rule 2.)
 * ---------------------------------------------------------------------------------------- */
static inline int64_t n_values(int64_t bytes, int32_t width) { return (bytes + width - 1) / width; }

/* growable buffer == scala.collection.mutable.Buffer (ArrayBuffer): amortised doubling append */
typedef struct {
    uint8_t *p;
    int64_t n, cap;
    int32_t width;
} gbuf;
static NOINLINE int gbuf_append(gbuf *g, const uint8_t *elem) {
    if (g->n == g->cap) {
        int64_t ncap = g->cap ? g->cap * 2 : 16;
        uint8_t *np = (uint8_t *)realloc(g->p, (size_t)(ncap * g->width));
        if (!np) return -1;
        g->p = np;
        g->cap = ncap;
    }
    memcpy(g->p + g->n * g->width, elem, (size_t)g->width);
    g->n++;
    return 0;
}

/* A decoded column vector (IntColumnVector / TinyIntColumnVector / StringColumnVector,
 * core/DataVector.scala:42-48), held as packed `width`-byte values. */
typedef struct {
    uint8_t *data; /* n * width bytes; for DENSE_INT these are native int32 after decode */
    int64_t n;
    int32_t width;
    int32_t codec;
} colvec;

/* faithful: copy the block (Segment.scala:165-167), then stream it element by element through
 * a `width`-byte chunk into a growable buffer, then toArray (one more copy). */
static NOINLINE int decode_faithful(const uint8_t *blk, int64_t len, int32_t codec, int32_t width, colvec *out) {
    uint8_t *bytes = (uint8_t *)malloc((size_t)(len > 0 ? len : 1)); /* new Array[Byte](...) */
    if (!bytes) return -1;
    memcpy(bytes, blk, (size_t)len);
    uint8_t chunk[256];
    uint8_t *chunkp = width <= 256 ? chunk : (uint8_t *)malloc((size_t)width);
    memset(chunkp, 0, (size_t)width);
    gbuf g = {0, 0, 0, codec == IMM3O_DENSE_INT ? 4 : width};
    int64_t pos = 0;
    while (pos < len) { /* read(chunk) != -1 */
        int64_t take = len - pos < width ? len - pos : width;
        memcpy(chunkp, bytes + pos, (size_t)take);
        pos += take;
        if (codec == IMM3O_DENSE_INT) {
            int32_t v = imm3o_bytes_to_int(chunkp); /* IntType.bytesToValue, DataType.scala:48 */
            if (gbuf_append(&g, (const uint8_t *)&v)) return -1;
        } else {
            /* TinyIntType.bytesToValue = bytes(0) (DataType.scala:61); StringType = new String(bytes) (:70) */
            if (gbuf_append(&g, chunkp)) return -1;
        }
    }
    out->n = g.n;
    out->width = g.width;
    out->codec = codec;
    out->data = (uint8_t *)malloc((size_t)(g.n * g.width > 0 ? g.n * g.width : 1)); /* toArray */
    if (!out->data) return -1;
    memcpy(out->data, g.p, (size_t)(g.n * g.width));
    free(g.p);
    free(bytes);
    if (chunkp != chunk) free(chunkp);
    return 0;
}

/* tight: same values, one pass, no per-element allocation. */
static int decode_tight(const uint8_t *blk, int64_t len, int32_t codec, int32_t width, colvec *out) {
    int64_t n = n_values(len, width);
    int32_t ow = codec == IMM3O_DENSE_INT ? 4 : width;
    out->n = n;
    out->width = ow;
    out->codec = codec;
    out->data = (uint8_t *)malloc((size_t)(n * ow > 0 ? n * ow : 1));
    if (!out->data) return -1;
    int64_t full = len / width;
    if (codec == IMM3O_DENSE_INT) {
        int32_t *o = (int32_t *)out->data;
        for (int64_t i = 0; i < full; i++) o[i] = imm3o_bytes_to_int(blk + 4 * i);
    } else {
        memcpy(out->data, blk, (size_t)(full * width));
    }
    if (full < n) { /* short final read: stale tail from the previous chunk (zeros if none) */
        uint8_t chunk[256];
        uint8_t *chunkp = width <= 256 ? chunk : (uint8_t *)malloc((size_t)width);
        if (full > 0) memcpy(chunkp, blk + (full - 1) * width, (size_t)width);
        else memset(chunkp, 0, (size_t)width);
        memcpy(chunkp, blk + full * width, (size_t)(len - full * width));
        if (codec == IMM3O_DENSE_INT) ((int32_t *)out->data)[full] = imm3o_bytes_to_int(chunkp);
        else memcpy(out->data + full * ow, chunkp, (size_t)width);
        if (chunkp != chunk) free(chunkp);
    }
    return 0;
}

static int codec_supported(int32_t codec) {
    /* Column.getCodec (core/Column.scala:57-63) + the match in Scan.scala:37-50.  PFOR_INT is
     * dispatched there too but its decode is broken (codec/PFORCodec.scala:43-50) and the loader
     * cannot produce it (loader/.../LoaderCli.scala:118-122): out of scope, reported as no codec. */
    return codec == IMM3O_DENSE_INT || codec == IMM3O_DENSE_TINYINT || codec == IMM3O_DENSE_STRING;
}

/* ------------------------------------------------------------------------------------------
 * Layout of the batches (Scan.scala:55-60, 72)
 * ---------------------------------------------------------------------------------------- */
int32_t imm3o_n_batches(const imm3o_column *first) { return first->n_offsets > 0 ? first->n_offsets - 1 : 0; }

int64_t imm3o_layout(const imm3o_column *first, int32_t table_block_size,
                     int32_t *batch_size, int32_t *batch_oid, int64_t *batch_word_off) {
    int32_t nb = imm3o_n_batches(first);
    int64_t w = 0;
    for (int32_t k = 0; k < nb; k++) {
        int64_t len = (int64_t)first->block_offsets[k + 1] - (int64_t)first->block_offsets[k];
        int64_t n = len > 0 ? n_values(len, first->width) : 0;
        if (batch_size) batch_size[k] = (int32_t)n;                 /* vecSize = columnVectors(0).data.size */
        if (batch_oid) batch_oid[k] = (int32_t)((uint32_t)k * (uint32_t)table_block_size); /* vecCounter * table.blockSize */
        if (batch_word_off) batch_word_off[k] = w;
        w += (n + 63) / 64;
    }
    return w;
}

/* ------------------------------------------------------------------------------------------
 * SelectOp iterators (engine/engine/operator/Select.scala:25-165).  Each clears the bit of every
 * row that FAILS; never sets one.  `size` is the batch size (first column), so a shorter
 * predicate column is an ArrayIndexOutOfBounds in the reference.
 * ---------------------------------------------------------------------------------------- */
#define SELECT_LOOP(T, EXPR, REMOVE)                          \
    do {                                                      \
        const T *data = (const T *)cv->data;                  \
        for (int32_t x = 0; x < size; x++) {                  \
            if (x >= cv->n) return IMM3O_ERR_INDEX;           \
            if (!(EXPR)) REMOVE(sel, x);                      \
        }                                                     \
    } while (0)

#define BITCLR_INLINE(w, i) ((w)[(i) >> 6] &= ~(1ULL << ((i) & 63)))

static int select_apply(const colvec *cv, const imm3o_select *s, int32_t size, uint64_t *sel, int faithful, char *msg) {
    switch (s->cond) {
    case IMM3O_GT: /* Select.scala:53-89, strict > */
        if (cv->codec == IMM3O_DENSE_INT) {
            int32_t t = imm3o_d2i(s->value); /* gt.toInt, :65 */
            if (faithful) SELECT_LOOP(int32_t, data[x] > t, bitset_remove); else SELECT_LOOP(int32_t, data[x] > t, BITCLR_INLINE);
        } else if (cv->codec == IMM3O_DENSE_TINYINT) {
            int8_t t = imm3o_d2b(s->value); /* gt.toByte, :73 */
            if (faithful) SELECT_LOOP(int8_t, data[x] > t, bitset_remove); else SELECT_LOOP(int8_t, data[x] > t, BITCLR_INLINE);
        } else goto unsupported_vector;
        return IMM3O_OK;
    case IMM3O_LT: /* Select.scala:91-127, strict < */
        if (cv->codec == IMM3O_DENSE_INT) {
            int32_t t = imm3o_d2i(s->value);
            if (faithful) SELECT_LOOP(int32_t, data[x] < t, bitset_remove); else SELECT_LOOP(int32_t, data[x] < t, BITCLR_INLINE);
        } else if (cv->codec == IMM3O_DENSE_TINYINT) {
            int8_t t = imm3o_d2b(s->value);
            if (faithful) SELECT_LOOP(int8_t, data[x] < t, bitset_remove); else SELECT_LOOP(int8_t, data[x] < t, BITCLR_INLINE);
        } else goto unsupported_vector;
        return IMM3O_OK;
    case IMM3O_EQ: /* Select.scala:129-165 */
        if (cv->codec == IMM3O_DENSE_INT) {
            int32_t t = imm3o_d2i(s->value);
            if (faithful) SELECT_LOOP(int32_t, data[x] == t, bitset_remove); else SELECT_LOOP(int32_t, data[x] == t, BITCLR_INLINE);
        } else if (cv->codec == IMM3O_DENSE_TINYINT) {
            int8_t t = imm3o_d2b(s->value);
            if (faithful) SELECT_LOOP(int8_t, data[x] == t, bitset_remove); else SELECT_LOOP(int8_t, data[x] == t, BITCLR_INLINE);
        } else goto unsupported_vector;
        return IMM3O_OK;
    case IMM3O_MATCH: /* Select.scala:25-51: !matchValues.contains(data(x)) -> remove */
        if (cv->codec != IMM3O_DENSE_STRING) goto unsupported_vector;
        for (int32_t x = 0; x < size; x++) {
            if (x >= cv->n) return IMM3O_ERR_INDEX;
            const uint8_t *v = cv->data + (int64_t)x * cv->width;
            int found = 0;
            int64_t off = 0;
            /* List.contains -> String.equals.  new String(bytes) (DataType.scala:70) is compared as
             * raw bytes: identical to String equality whenever both sides are valid UTF-8/ASCII. */
            for (int32_t m = 0; m < s->n_match && !found; m++) {
                if (s->match_lens[m] == cv->width && memcmp(v, s->match_bytes + off, (size_t)cv->width) == 0) found = 1;
                off += s->match_lens[m];
            }
            if (!found) { if (faithful) bitset_remove(sel, x); else BITCLR_INLINE(sel, x); }
        }
        return IMM3O_OK;
    default: /* NotMatch, NoOp: Select.scala:22 */
        snprintf(msg, 128, "Unsupported condition: %d", s->cond);
        return IMM3O_ERR_UNSUPPORTED_CONDITION;
    }
unsupported_vector:
    snprintf(msg, 128, "Unsupported column vector"); /* Select.scala:41,80,118,156 */
    return IMM3O_ERR_UNSUPPORTED_VECTOR;
}

/* ------------------------------------------------------------------------------------------
 * PipelineThread.run for one segment, minus the queue (engine/engine/Engine.scala:247-262):
 *   scanOp = ScanOp(sm, segIdx, table, usedColumns); fold the SelectOps over it; drain.
 * ---------------------------------------------------------------------------------------- */
int imm3o_scan_select(const imm3o_column *cols, int32_t ncols,
                      const imm3o_select *sels, int32_t nsels,
                      int32_t table_block_size, int32_t flavour,
                      uint64_t *words_out, uint64_t *count_out, char *msg) {
    (void)table_block_size;
    char local[128];
    if (!msg) msg = local;
    msg[0] = 0;
    if (ncols <= 0 || !cols) { snprintf(msg, 128, "no columns"); return IMM3O_ERR_ARG; }
    for (int32_t c = 0; c < ncols; c++) {
        if (cols[c].width <= 0) { snprintf(msg, 128, "bad width"); return IMM3O_ERR_ARG; }
    }
    for (int32_t s = 0; s < nsels; s++) {
        /* `vec.columns...filter(name == col).head` on a column that is not in the batch throws
         * NoSuchElementException; Engine.getColumns makes that impossible for real queries. */
        if (sels[s].column < 0 || sels[s].column >= ncols) { snprintf(msg, 128, "select column not among used columns"); return IMM3O_ERR_ARG; }
    }
    /* SelectOp.iterator (Select.scala:17-23) matches on the condition when the iterator chain is BUILT
     * (PipelineThread.run: runOps(...).iterator, Engine.scala:251), so NotMatch / NoOp throw even for a
     * segment without blocks.  "Unsupported column vector" and "No implementation for codec" on the other
     * hand are only reached when a batch is processed. */
    for (int32_t s = 0; s < nsels; s++) {
        int cnd = sels[s].cond;
        if (cnd != IMM3O_MATCH && cnd != IMM3O_GT && cnd != IMM3O_LT && cnd != IMM3O_EQ) {
            snprintf(msg, 128, "Unsupported condition: %d", cnd);
            return IMM3O_ERR_UNSUPPORTED_CONDITION;
        }
    }
    block_iter *its = (block_iter *)malloc(sizeof(block_iter) * (size_t)ncols);
    colvec *vecs = (colvec *)calloc((size_t)ncols, sizeof(colvec));
    for (int32_t c = 0; c < ncols; c++) block_iter_init(&its[c], &cols[c]);
    uint64_t total = 0;
    int64_t woff = 0;
    int rc = IMM3O_OK;
    const int faithful = (flavour == 0);
    while (block_iter_has_next(&its[0])) { /* hasNext = segmentIters.head.hasNext, Scan.scala:72 */
        /* ScanOp.DataVectorIterator.next, Scan.scala:28-70 */
        for (int32_t c = 0; c < ncols && rc == IMM3O_OK; c++) {
            if (!codec_supported(cols[c].codec)) { /* `case _ => throw`, Scan.scala:49 */
                snprintf(msg, 128, "No implementation for codec %d", cols[c].codec);
                rc = IMM3O_ERR_NO_CODEC;
                break;
            }
            if (!block_iter_has_next(&its[c])) { /* blockOffsets(endByteIdx) out of range */
                snprintf(msg, 128, "ArrayIndexOutOfBounds: column %d has fewer blocks", c);
                rc = IMM3O_ERR_INDEX;
                break;
            }
            const uint8_t *blk = 0;
            int64_t len = block_iter_peek(&its[c], &blk);
            if (len < 0) { snprintf(msg, 128, "BufferUnderflow: column %d block %d", c, its[c].position); rc = IMM3O_ERR_INDEX; break; }
            int drc = faithful ? decode_faithful(blk, len, cols[c].codec, cols[c].width, &vecs[c])
                               : decode_tight(blk, len, cols[c].codec, cols[c].width, &vecs[c]);
            if (drc) { snprintf(msg, 128, "out of memory"); rc = IMM3O_ERR_ARG; break; }
            block_iter_advance(&its[c], len);
        }
        if (rc != IMM3O_OK) break;
        int32_t vec_size = (int32_t)vecs[0].n; /* based on first column, Scan.scala:55 */
        int64_t nw = ((int64_t)vec_size + 63) / 64;
        uint64_t *sel = words_out + woff;
        for (int64_t k = 0; k < nw; k++) sel[k] = 0;
        if (faithful) {
            for (int32_t x = 0; x < vec_size; x++) bitset_add(sel, x); /* Scan.scala:56-57 */
        } else {
            for (int64_t k = 0; k < vec_size / 64; k++) sel[k] = ~0ULL;
            if (vec_size & 63) sel[vec_size / 64] = (1ULL << (vec_size & 63)) - 1;
        }
        for (int32_t s = 0; s < nsels && rc == IMM3O_OK; s++) {
            rc = select_apply(&vecs[sels[s].column], &sels[s], vec_size, sel, faithful, msg);
            if (rc == IMM3O_ERR_INDEX) snprintf(msg, 128, "ArrayIndexOutOfBounds: select column %d shorter than batch", sels[s].column);
        }
        for (int32_t c = 0; c < ncols; c++) { free(vecs[c].data); vecs[c].data = 0; }
        if (rc != IMM3O_OK) break;
        total += (uint64_t)bitset_size(sel, nw);
        woff += nw;
    }
    for (int32_t c = 0; c < ncols; c++) free(vecs[c].data);
    free(vecs);
    free(its);
    if (count_out) *count_out = total;
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * ProjectOp.ProjectIterator (engine/engine/operator/Project.scala:37-80)
 * ---------------------------------------------------------------------------------------- */
int64_t imm3o_project(const imm3o_column *cols, int32_t ncols,
                      const int32_t *proj, int32_t nproj, int64_t limit,
                      int32_t table_block_size, const uint64_t *words,
                      int32_t *out_batch, int32_t *out_pos, uint8_t *const *out_vals,
                      int64_t cap_rows, int32_t *would_throw) {
    (void)table_block_size;
    if (would_throw) *would_throw = 0;
    if (ncols <= 0) return -IMM3O_ERR_ARG;
    for (int32_t j = 0; j < nproj; j++)
        if (proj[j] < 0 || proj[j] >= ncols) return -IMM3O_ERR_ARG;
    block_iter *its = (block_iter *)malloc(sizeof(block_iter) * (size_t)ncols);
    colvec *vecs = (colvec *)calloc((size_t)ncols, sizeof(colvec));
    for (int32_t c = 0; c < ncols; c++) block_iter_init(&its[c], &cols[c]);
    int64_t total = 0; /* totalRecordCount */
    int64_t woff = 0;
    int32_t k = 0;
    int64_t rc = 0;
    /* hasNext: limit > 0 && totalRecordCount >= limit -> false (Project.scala:73-80); a new batch is
     * pulled only from next(), i.e. only while more rows are wanted. */
    while (block_iter_has_next(&its[0]) && !(limit > 0 && total >= limit)) {
        const uint8_t *blk0 = 0;
        int64_t len0 = block_iter_peek(&its[0], &blk0);
        if (len0 < 0) { rc = -IMM3O_ERR_INDEX; break; }
        int32_t size = (int32_t)n_values(len0, cols[0].width);
        int64_t nw = ((int64_t)size + 63) / 64;
        const uint64_t *sel = words + woff;
        int64_t nsel = bitset_size(sel, nw); /* currVec.selected.size, Project.scala:39 */
        /* decode only what is projected (tight); values are identical either way */
        int need_decode = nsel > 0;
        for (int32_t c = 0; c < ncols; c++) {
            if (!block_iter_has_next(&its[c])) { rc = -IMM3O_ERR_INDEX; break; }
            const uint8_t *blk = 0;
            int64_t len = block_iter_peek(&its[c], &blk);
            if (len < 0) { rc = -IMM3O_ERR_INDEX; break; }
            int used = 0;
            for (int32_t j = 0; j < nproj; j++) used |= (proj[j] == c);
            if (need_decode && used) {
                if (decode_tight(blk, len, cols[c].codec, cols[c].width, &vecs[c])) { rc = -IMM3O_ERR_ARG; break; }
            }
            block_iter_advance(&its[c], len);
        }
        if (rc < 0) break;
        if (nsel == 0) {
            /* Reference: next() walks currVecPos to size (Project.scala:50-53) and then indexes
             * data(size) (:55-57) -> ArrayIndexOutOfBoundsException.  Operator-level
             * parity skips the empty batch; the would-throw is reported, not replicated. */
            if (would_throw) *would_throw = 1;
        } else {
            int64_t curr = 0; /* currRecordCount */
            int32_t pos = 0;  /* currVecPos */
            while (curr < nsel && !(limit > 0 && total >= limit)) {
                while (pos < size && !bitset_contains(sel, pos)) pos++; /* skip over non selected, :50-53 */
                if (total >= cap_rows) { rc = -IMM3O_ERR_ARG; break; }
                for (int32_t j = 0; j < nproj; j++) {
                    const colvec *cv = &vecs[proj[j]];
                    if (pos >= cv->n) { rc = -IMM3O_ERR_INDEX; break; }
                    memcpy(out_vals[j] + total * cv->width, cv->data + (int64_t)pos * cv->width, (size_t)cv->width);
                }
                if (rc < 0) break;
                if (out_batch) out_batch[total] = k;
                if (out_pos) out_pos[total] = pos;
                curr++;
                total++;
                pos++;
            }
        }
        for (int32_t c = 0; c < ncols; c++) { free(vecs[c].data); vecs[c].data = 0; }
        if (rc < 0) break;
        woff += nw;
        k++;
    }
    for (int32_t c = 0; c < ncols; c++) free(vecs[c].data);
    free(vecs);
    free(its);
    return rc < 0 ? rc : total;
}

/* ===========================================================================================
 * ProjectAggOp.ProjectAggIterator.runAggs -- engine/engine/operator/ProjectAggregate.scala:115-227
 * (the C twin of oracle_np.project_agg: group-by aggregation is checked by two restatements).
 *
 *   for every batch, for every selected position ascending (:158-159):
 *       groupKey = group column values mkString "_"                          (:144, :160-164)
 *       resultMap.getOrElseUpdate(groupKey, fresh aggregators)               (:177, LinkedHashMap: first-seen order, :126)
 *       CountAggr.add            counter += 1                                (:22-24)
 *       Max/MinDoubleAggr.add    value.toDouble, started at -/+Double.MaxValue (:36-39, :49-52)
 *       MaxStringAggr.add        "" = unset, else String.compareTo           (:78-83)
 *   a String vector only takes CountAggr / MaxStringAggr ("bad aggregator for this data type", :205-209).
 * =========================================================================================== */
typedef struct agg_group {
    char *key;
    int64_t *counts;   /* per aggregate */
    double *nums;
    char *strs;        /* per aggregate, str_stride bytes, NUL-terminated ("" = unset) */
    struct agg_group *next_in_bucket;
} agg_group;

static uint64_t fnv1a(const char *s) {
    uint64_t h = 1469598103934665603ULL;
    for (; *s; ++s) { h ^= (uint8_t)*s; h *= 1099511628211ULL; }
    return h;
}

int64_t imm3o_project_agg(const imm3o_column *cols, int32_t ncols, const int32_t *group, int32_t ngroup,
                          const imm3o_aggregate *aggs, int32_t naggs, const uint64_t *words,
                          char *keys_out, int32_t key_stride, int64_t *count_out, double *num_out,
                          char *str_out, int32_t str_stride, int64_t max_groups, char *msg) {
    if (msg) msg[0] = 0;
    if (ncols <= 0 || naggs <= 0 || ngroup < 0) return -IMM3O_ERR_ARG;
    for (int32_t g = 0; g < ngroup; g++) if (group[g] < 0 || group[g] >= ncols) return -IMM3O_ERR_ARG;
    for (int32_t a = 0; a < naggs; a++) if (aggs[a].column < 0 || aggs[a].column >= ncols || aggs[a].kind < 0 || aggs[a].kind > 2) return -IMM3O_ERR_ARG;
    const int64_t n_buckets = 1 << 16;
    agg_group **buckets = (agg_group **)calloc((size_t)n_buckets, sizeof(agg_group *));
    agg_group **order = 0; /* first-seen order */
    int64_t n_groups = 0, cap_groups = 0;
    block_iter *its = (block_iter *)malloc(sizeof(block_iter) * (size_t)ncols);
    colvec *vecs = (colvec *)calloc((size_t)ncols, sizeof(colvec));
    for (int32_t c = 0; c < ncols; c++) block_iter_init(&its[c], &cols[c]);
    int64_t woff = 0, rc = 0;
    char *keybuf = (char *)malloc((size_t)key_stride + 64);
    while (block_iter_has_next(&its[0]) && rc == 0) {
        const uint8_t *blk0 = 0;
        int64_t len0 = block_iter_peek(&its[0], &blk0);
        if (len0 < 0) { rc = -IMM3O_ERR_INDEX; break; }
        const int32_t size = (int32_t)n_values(len0, cols[0].width);
        const int64_t nw = ((int64_t)size + 63) / 64;
        const uint64_t *sel = words + woff;
        for (int32_t c = 0; c < ncols; c++) {
            if (!block_iter_has_next(&its[c])) { rc = -IMM3O_ERR_INDEX; break; }
            const uint8_t *blk = 0;
            int64_t len = block_iter_peek(&its[c], &blk);
            if (len < 0) { rc = -IMM3O_ERR_INDEX; break; }
            if (decode_tight(blk, len, cols[c].codec, cols[c].width, &vecs[c])) { rc = -IMM3O_ERR_ARG; break; }
            block_iter_advance(&its[c], len);
        }
        for (int32_t pos = 0; pos < size && rc == 0; pos++) {
            if (!bitset_contains(sel, pos)) continue; /* for (currVecBatchPos <- currVecBatch.selected), ascending */
            /* groupKey: values mkString "_" */
            int32_t kl = 0;
            for (int32_t g = 0; g < ngroup && rc == 0; g++) {
                const colvec *cv = &vecs[group[g]];
                if (pos >= cv->n) { rc = -IMM3O_ERR_INDEX; break; }
                if (g) keybuf[kl++] = '_';
                const uint8_t *p = cv->data + (int64_t)pos * cv->width;
                if (cols[group[g]].codec == IMM3O_DENSE_INT) kl += snprintf(keybuf + kl, 16, "%d", imm3o_bytes_to_int(p));
                else if (cols[group[g]].codec == IMM3O_DENSE_TINYINT) kl += snprintf(keybuf + kl, 8, "%d", (int)(int8_t)p[0]);
                else { memcpy(keybuf + kl, p, (size_t)cv->width); kl += cv->width; } /* new String(bytes), DataType.scala:70 */
                if (kl + 24 + 256 > key_stride + 64 && kl >= key_stride) { rc = -IMM3O_ERR_ARG; break; }
            }
            if (rc) break;
            keybuf[kl] = 0;
            if (kl >= key_stride) { rc = -IMM3O_ERR_ARG; break; }
            const uint64_t h = fnv1a(keybuf) & (uint64_t)(n_buckets - 1);
            agg_group *G = buckets[h];
            while (G && strcmp(G->key, keybuf)) G = G->next_in_bucket;
            if (!G) { /* getOrElseUpdate(groupKey, getNewAggs(aggsMap)) */
                if (n_groups >= max_groups) { rc = -IMM3O_ERR_ARG; break; }
                G = (agg_group *)calloc(1, sizeof(agg_group));
                G->key = (char *)malloc((size_t)kl + 1);
                memcpy(G->key, keybuf, (size_t)kl + 1);
                G->counts = (int64_t *)calloc((size_t)naggs, sizeof(int64_t));
                G->nums = (double *)calloc((size_t)naggs, sizeof(double));
                G->strs = (char *)calloc((size_t)naggs, (size_t)str_stride);
                for (int32_t a = 0; a < naggs; a++) G->nums[a] = aggs[a].kind == 2 ? -DBL_MAX : DBL_MAX; /* Double.MinValue / Double.MaxValue */
                G->next_in_bucket = buckets[h];
                buckets[h] = G;
                if (n_groups == cap_groups) {
                    cap_groups = cap_groups ? 2 * cap_groups : 256;
                    order = (agg_group **)realloc(order, (size_t)cap_groups * sizeof(agg_group *));
                }
                order[n_groups++] = G;
            }
            for (int32_t a = 0; a < naggs && rc == 0; a++) {
                const int32_t c = aggs[a].column;
                const colvec *cv = &vecs[c];
                if (pos >= cv->n) { rc = -IMM3O_ERR_INDEX; break; }
                const uint8_t *p = cv->data + (int64_t)pos * cv->width;
                if (aggs[a].kind == 0) { G->counts[a] += 1; continue; } /* CountAggr.add(value) */
                if (cols[c].codec == IMM3O_DENSE_STRING) {
                    if (aggs[a].kind != 2) { if (msg) snprintf(msg, 128, "bad aggregator for this data type"); rc = -IMM3O_ERR_UNSUPPORTED_VECTOR; break; }
                    if (cv->width + 1 > str_stride) { rc = -IMM3O_ERR_ARG; break; }
                    char *cur = G->strs + (size_t)a * (size_t)str_stride;
                    /* MaxStringAggr.add: if (max == "") max = v else if (v > max) max = v  (String.compareTo: UTF-16 order;
                     * byte order for ASCII) */
                    if (cur[0] == 0 || memcmp(p, cur, (size_t)cv->width) > 0) { memcpy(cur, p, (size_t)cv->width); cur[cv->width] = 0; }
                } else {
                    const double v = cols[c].codec == IMM3O_DENSE_INT ? (double)imm3o_bytes_to_int(p) : (double)(int8_t)p[0]; /* value.toDouble */
                    if (aggs[a].kind == 2) { if (v > G->nums[a]) G->nums[a] = v; }
                    else if (v < G->nums[a]) G->nums[a] = v;
                }
            }
        }
        for (int32_t c = 0; c < ncols; c++) { free(vecs[c].data); vecs[c].data = 0; }
        woff += nw;
    }
    if (rc == 0) {
        for (int64_t g = 0; g < n_groups; g++) {
            snprintf(keys_out + g * key_stride, (size_t)key_stride, "%s", order[g]->key);
            memcpy(count_out + g * naggs, order[g]->counts, (size_t)naggs * sizeof(int64_t));
            memcpy(num_out + g * naggs, order[g]->nums, (size_t)naggs * sizeof(double));
            memcpy(str_out + g * naggs * str_stride, order[g]->strs, (size_t)naggs * (size_t)str_stride);
        }
    }
    for (int64_t g = 0; g < n_groups; g++) { free(order[g]->key); free(order[g]->counts); free(order[g]->nums); free(order[g]->strs); free(order[g]); }
    for (int32_t c = 0; c < ncols; c++) free(vecs[c].data);
    free(order); free(buckets); free(vecs); free(its); free(keybuf);
    return rc < 0 ? rc : n_groups;
}
