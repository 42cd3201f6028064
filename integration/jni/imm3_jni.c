/*
 * imm3_jni.c -- JNI shim between the reference's JVM and the C ABI of include/imm3.h.
 *
 * Binds immutabledb.gpu.Native (integration/scala/immutabledb/gpu/Native.scala).  Direct ByteBuffers are the
 * MappedByteBuffers SegmentManager already holds (core/src/main/scala/immutabledb/storage/SegmentManager.scala:81-87);
 * non-zero statuses become java.lang.Exception(msg), the reference's error convention (Scan.scala:49,
 * Select.scala:22,41,80).
 *
 * NOT COMPILED AGAINST A JDK IN THIS REPOSITORY'S PIPELINE: the build image has no JDK (no jni.h, no libjvm).  The guard
 * below makes the translation unit empty there; tests/test_host.py only SYNTAX-checks it against a minimal stand-in for
 * jni.h (tests/jni_stub/jni.h: the handful of JNIEnv entries used here) -- it has never been linked or run.  UNVERIFIED.
 * Build on a host with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       integration/jni/imm3_jni.c -Limmutable3_amd/lib -limm3 -o libimm3_jni.so
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#define IMM3_HAVE_JNI 1
#endif
#endif

#ifdef IMM3_HAVE_JNI
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "imm3.h"

static void throw_last(JNIEnv *env) {
    jclass cls = (*env)->FindClass(env, "java/lang/Exception");
    if (cls) (*env)->ThrowNew(env, cls, imm3_last_error());
}
#define CHECKED(call) do { if ((call) != IMM3_OK) { throw_last(env); goto done; } } while (0)

JNIEXPORT jlong JNICALL Java_immutabledb_gpu_Native_00024_ctxCreate(JNIEnv *env, jobject self, jint device) {
    imm3_ctx *ctx = NULL;
    if (imm3_ctx_create(device, NULL, &ctx) != IMM3_OK) throw_last(env);
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_ctxDestroy(JNIEnv *env, jobject self, jlong ctx) {
    imm3_ctx_destroy((imm3_ctx *)(intptr_t)ctx);
}

/* dats: Array[ByteBuffer] (direct, the mmaps); offsets: Array[Array[Int]] (SegmentMeta.blockOffsets) */
JNIEXPORT jlong JNICALL Java_immutabledb_gpu_Native_00024_segmentCreate(JNIEnv *env, jobject self, jlong ctx,
        jintArray codecs, jintArray widths, jobjectArray dats, jobjectArray offsets) {
    jsize n = (*env)->GetArrayLength(env, codecs);
    imm3_column *cols = (imm3_column *)calloc((size_t)n, sizeof(imm3_column));
    jint *cd = (*env)->GetIntArrayElements(env, codecs, NULL);
    jint *wd = (*env)->GetIntArrayElements(env, widths, NULL);
    jintArray *offArrs = (jintArray *)calloc((size_t)n, sizeof(jintArray));
    jint **offPtrs = (jint **)calloc((size_t)n, sizeof(jint *));
    imm3_segment *seg = NULL;
    for (jsize i = 0; i < n; i++) {
        jobject buf = (*env)->GetObjectArrayElement(env, dats, i);
        offArrs[i] = (jintArray)(*env)->GetObjectArrayElement(env, offsets, i);
        offPtrs[i] = (*env)->GetIntArrayElements(env, offArrs[i], NULL);
        cols[i].codec = cd[i];
        cols[i].width = wd[i];
        cols[i].dat = (*env)->GetDirectBufferAddress(env, buf);
        cols[i].dat_bytes = (uint64_t)(*env)->GetDirectBufferCapacity(env, buf);
        cols[i].block_offsets = (const int32_t *)offPtrs[i];
        cols[i].n_offsets = (int32_t)(*env)->GetArrayLength(env, offArrs[i]);
        (*env)->DeleteLocalRef(env, buf); /* one local ref per column otherwise: a wide table would exhaust the frame */
    }
    CHECKED(imm3_segment_create((imm3_ctx *)(intptr_t)ctx, cols, (int32_t)n, &seg));
done:
    for (jsize i = 0; i < n; i++)
        if (offPtrs[i]) {
            (*env)->ReleaseIntArrayElements(env, offArrs[i], offPtrs[i], JNI_ABORT);
            (*env)->DeleteLocalRef(env, offArrs[i]);
        }
    (*env)->ReleaseIntArrayElements(env, codecs, cd, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, widths, wd, JNI_ABORT);
    free(offPtrs); free(offArrs); free(cols);
    return (jlong)(intptr_t)seg;
}

JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_segmentDestroy(JNIEnv *env, jobject self, jlong seg) {
    imm3_segment_destroy((imm3_segment *)(intptr_t)seg);
}

/* selMatch(i): the IN-list of leaf i as Array[Array[Byte]] (String.getBytes of each value), or null */
JNIEXPORT jlong JNICALL Java_immutabledb_gpu_Native_00024_queryCreate(JNIEnv *env, jobject self, jlong ctx, jlong seg,
        jintArray usedCols, jintArray selCols, jintArray selConds, jdoubleArray selValues, jobjectArray selMatch,
        jintArray proj, jlong limit, jint blockSize) {
    jsize nUsed = (*env)->GetArrayLength(env, usedCols);
    jsize nSel = (*env)->GetArrayLength(env, selCols);
    jsize nProj = (*env)->GetArrayLength(env, proj);
    jint *used = (*env)->GetIntArrayElements(env, usedCols, NULL);
    jint *sc = (*env)->GetIntArrayElements(env, selCols, NULL);
    jint *sk = (*env)->GetIntArrayElements(env, selConds, NULL);
    jdouble *sv = (*env)->GetDoubleArrayElements(env, selValues, NULL);
    jint *pj = (*env)->GetIntArrayElements(env, proj, NULL);
    imm3_select *sels = (imm3_select *)calloc((size_t)(nSel > 0 ? nSel : 1), sizeof(imm3_select));
    uint8_t **blobs = (uint8_t **)calloc((size_t)(nSel > 0 ? nSel : 1), sizeof(uint8_t *));
    int32_t **lens = (int32_t **)calloc((size_t)(nSel > 0 ? nSel : 1), sizeof(int32_t *));
    imm3_query *q = NULL;
    for (jsize i = 0; i < nSel; i++) {
        sels[i].column = sc[i];
        sels[i].cond = sk[i];
        sels[i].value = sv[i];
        jobjectArray vals = selMatch ? (jobjectArray)(*env)->GetObjectArrayElement(env, selMatch, i) : NULL;
        if (vals) {
            jsize nv = (*env)->GetArrayLength(env, vals);
            size_t total = 0;
            lens[i] = (int32_t *)calloc((size_t)(nv > 0 ? nv : 1), sizeof(int32_t));
            for (jsize m = 0; m < nv; m++) {
                jbyteArray b = (jbyteArray)(*env)->GetObjectArrayElement(env, vals, m);
                lens[i][m] = (int32_t)(*env)->GetArrayLength(env, b);
                total += (size_t)lens[i][m];
                (*env)->DeleteLocalRef(env, b); /* (an IN-list can be long: do not accumulate local refs) */
            }
            blobs[i] = (uint8_t *)malloc(total ? total : 1);
            size_t off = 0;
            for (jsize m = 0; m < nv; m++) {
                jbyteArray b = (jbyteArray)(*env)->GetObjectArrayElement(env, vals, m);
                (*env)->GetByteArrayRegion(env, b, 0, lens[i][m], (jbyte *)(blobs[i] + off));
                off += (size_t)lens[i][m];
                (*env)->DeleteLocalRef(env, b);
            }
            sels[i].match_bytes = blobs[i];
            sels[i].match_lens = lens[i];
            sels[i].n_match = (int32_t)nv;
            (*env)->DeleteLocalRef(env, vals);
        }
    }
    CHECKED(imm3_query_create((imm3_ctx *)(intptr_t)ctx, (imm3_segment *)(intptr_t)seg, (const int32_t *)used, (int32_t)nUsed,
                              sels, (int32_t)nSel, (const int32_t *)pj, (int32_t)nProj, (int64_t)limit, blockSize, &q));
done:
    for (jsize i = 0; i < nSel; i++) { free(blobs[i]); free(lens[i]); }
    free(blobs); free(lens); free(sels);
    (*env)->ReleaseIntArrayElements(env, usedCols, used, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, selCols, sc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, selConds, sk, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, selValues, sv, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, proj, pj, JNI_ABORT);
    return (jlong)(intptr_t)q;
}

JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_queryDestroy(JNIEnv *env, jobject self, jlong q) {
    imm3_query_destroy((imm3_query *)(intptr_t)q);
}

JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_queryRun(JNIEnv *env, jobject self, jlong q) {
    if (imm3_query_run((imm3_query *)(intptr_t)q) != IMM3_OK) throw_last(env);
}

/* the count alone (selected.size summed): a single-launch select chain stores no bitmap */
JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_queryRunCount(JNIEnv *env, jobject self, jlong q) {
    if (imm3_query_run_count((imm3_query *)(intptr_t)q) != IMM3_OK) throw_last(env);
}

/* returns Array(size..., oid..., wordOffLo/Hi ...) packed as long[3 * nBatches] */
JNIEXPORT jlongArray JNICALL Java_immutabledb_gpu_Native_00024_queryBatches(JNIEnv *env, jobject self, jlong qh) {
    imm3_query *q = (imm3_query *)(intptr_t)qh;
    int32_t nb = 0; int64_t words = 0, rows = 0;
    jlongArray out = NULL;
    int32_t *size = NULL, *oid = NULL; int64_t *woff = NULL; jlong *packed = NULL;
    CHECKED(imm3_query_layout(q, &nb, &words, &rows));
    size = (int32_t *)calloc((size_t)(nb > 0 ? nb : 1), 4);
    oid = (int32_t *)calloc((size_t)(nb > 0 ? nb : 1), 4);
    woff = (int64_t *)calloc((size_t)(nb > 0 ? nb : 1), 8);
    CHECKED(imm3_query_batches(q, size, oid, woff));
    packed = (jlong *)calloc((size_t)(3 * nb > 0 ? 3 * nb : 1), sizeof(jlong));
    for (int32_t k = 0; k < nb; k++) { packed[k] = size[k]; packed[nb + k] = oid[k]; packed[2 * nb + k] = woff[k]; }
    out = (*env)->NewLongArray(env, 3 * nb);
    (*env)->SetLongArrayRegion(env, out, 0, 3 * nb, packed);
done:
    free(size); free(oid); free(woff); free(packed);
    return out;
}

/* the batch-major selection bitmap: slice [wordOff(k), +ceil(size(k)/64)) is batch k's BitSet words */
JNIEXPORT jlongArray JNICALL Java_immutabledb_gpu_Native_00024_queryBitmap(JNIEnv *env, jobject self, jlong qh) {
    imm3_query *q = (imm3_query *)(intptr_t)qh;
    int32_t nb = 0; int64_t words = 0, rows = 0;
    jlongArray out = NULL;
    uint64_t *buf = NULL;
    CHECKED(imm3_query_layout(q, &nb, &words, &rows));
    buf = (uint64_t *)malloc((size_t)(words > 0 ? words : 1) * 8);
    CHECKED(imm3_query_bitmap(q, buf, words));
    out = (*env)->NewLongArray(env, (jsize)words);
    (*env)->SetLongArrayRegion(env, out, 0, (jsize)words, (const jlong *)buf);
done:
    free(buf);
    return out;
}

JNIEXPORT jlong JNICALL Java_immutabledb_gpu_Native_00024_queryCount(JNIEnv *env, jobject self, jlong q) {
    uint64_t n = 0;
    if (imm3_query_count((imm3_query *)(intptr_t)q, &n) != IMM3_OK) throw_last(env);
    return (jlong)n;
}

JNIEXPORT jlong JNICALL Java_immutabledb_gpu_Native_00024_queryRowCount(JNIEnv *env, jobject self, jlong q) {
    uint64_t n = 0;
    if (imm3_query_row_count((imm3_query *)(intptr_t)q, &n) != IMM3_OK) throw_last(env);
    return (jlong)n;
}

/* cols: Array[ByteBuffer] (direct, capacity >= rows * width each); rowIndex: direct ByteBuffer of rows*4 bytes or null */
JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_queryFetchRows(JNIEnv *env, jobject self, jlong q,
        jobject rowIndex, jobjectArray cols, jlong maxRows) {
    jsize n = (*env)->GetArrayLength(env, cols);
    void **ptrs = (void **)calloc((size_t)(n > 0 ? n : 1), sizeof(void *));
    for (jsize j = 0; j < n; j++) {
        jobject b = (*env)->GetObjectArrayElement(env, cols, j);
        ptrs[j] = (*env)->GetDirectBufferAddress(env, b);
        (*env)->DeleteLocalRef(env, b);
    }
    uint32_t *idx = rowIndex ? (uint32_t *)(*env)->GetDirectBufferAddress(env, rowIndex) : NULL;
    if (imm3_query_fetch_rows((imm3_query *)(intptr_t)q, idx, ptrs, (uint64_t)maxRows) != IMM3_OK) throw_last(env);
    free(ptrs);
}

/* ---- multi-GPU (one JVM, G devices): one imm3_ctx per device, one communicator per context (ncclCommInitAll) ---- */
JNIEXPORT jlongArray JNICALL Java_immutabledb_gpu_Native_00024_commCreateAll(JNIEnv *env, jobject self, jlongArray ctxs) {
    jsize n = (*env)->GetArrayLength(env, ctxs);
    jlong *ch = (*env)->GetLongArrayElements(env, ctxs, NULL);
    imm3_ctx **cs = (imm3_ctx **)calloc((size_t)n, sizeof(imm3_ctx *));
    imm3_comm **comms = (imm3_comm **)calloc((size_t)n, sizeof(imm3_comm *));
    jlongArray out = NULL;
    for (jsize i = 0; i < n; i++) cs[i] = (imm3_ctx *)(intptr_t)ch[i];
    CHECKED(imm3_comm_create_all(cs, (int32_t)n, comms));
    out = (*env)->NewLongArray(env, n);
    for (jsize i = 0; i < n; i++) { jlong h = (jlong)(intptr_t)comms[i]; (*env)->SetLongArrayRegion(env, out, i, 1, &h); }
done:
    (*env)->ReleaseLongArrayElements(env, ctxs, ch, JNI_ABORT);
    free(cs); free(comms);
    return out;
}

/* ---- graphs: record the runs of a set of queries once, replay them with one call per execution ---- */
JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_captureBegin(JNIEnv *env, jobject self, jlong ctx) {
    if (imm3_ctx_capture_begin((imm3_ctx *)(intptr_t)ctx) != IMM3_OK) throw_last(env);
}

JNIEXPORT jlong JNICALL Java_immutabledb_gpu_Native_00024_captureEnd(JNIEnv *env, jobject self, jlong ctx) {
    imm3_graph *g = NULL;
    if (imm3_ctx_capture_end((imm3_ctx *)(intptr_t)ctx, &g) != IMM3_OK) throw_last(env);
    return (jlong)(intptr_t)g;
}

JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_graphLaunch(JNIEnv *env, jobject self, jlong g) {
    if (imm3_graph_launch((imm3_graph *)(intptr_t)g) != IMM3_OK) throw_last(env);
}

JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_graphDestroy(JNIEnv *env, jobject self, jlong g) {
    imm3_graph_destroy((imm3_graph *)(intptr_t)g);
}

JNIEXPORT void JNICALL Java_immutabledb_gpu_Native_00024_commDestroy(JNIEnv *env, jobject self, jlong c) {
    imm3_comm_destroy((imm3_comm *)(intptr_t)c);
}

/* queries(i): the queries device i ran in this pass (each already run); returns the selected-row count over all devices:
 * per device a sum kernel, then ONE ncclAllReduce(sum, uint64, 1) per device inside a group (the only collective of the path) */
JNIEXPORT jlong JNICALL Java_immutabledb_gpu_Native_00024_commAllreduceCountAll(JNIEnv *env, jobject self, jlongArray comms, jobjectArray queries) {
    jsize n = (*env)->GetArrayLength(env, comms);
    jlong *ch = (*env)->GetLongArrayElements(env, comms, NULL);
    imm3_comm **cs = (imm3_comm **)calloc((size_t)n, sizeof(imm3_comm *));
    imm3_query ***qs = (imm3_query ***)calloc((size_t)n, sizeof(imm3_query **));
    int32_t *nq = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    uint64_t total = 0;
    for (jsize i = 0; i < n; i++) {
        jlongArray qa = (jlongArray)(*env)->GetObjectArrayElement(env, queries, i);
        jsize m = (*env)->GetArrayLength(env, qa);
        jlong *qh = (*env)->GetLongArrayElements(env, qa, NULL);
        cs[i] = (imm3_comm *)(intptr_t)ch[i];
        nq[i] = (int32_t)m;
        qs[i] = (imm3_query **)calloc((size_t)(m > 0 ? m : 1), sizeof(imm3_query *));
        for (jsize k = 0; k < m; k++) qs[i][k] = (imm3_query *)(intptr_t)qh[k];
        (*env)->ReleaseLongArrayElements(env, qa, qh, JNI_ABORT);
        (*env)->DeleteLocalRef(env, qa);
    }
    CHECKED(imm3_comm_allreduce_count_all(cs, (int32_t)n, (imm3_query *const *const *)qs, nq, &total));
done:
    for (jsize i = 0; i < n; i++) free(qs[i]);
    (*env)->ReleaseLongArrayElements(env, comms, ch, JNI_ABORT);
    free(qs); free(cs); free(nq);
    return (jlong)total;
}

/* ProjectAggregateQueueOp across the devices of one JVM (ProjectAggregateQueue.scala:9-55): queries(i) / segments(i) = the
 * aggregation queries device i ran and the global segment index of each.  Returns the merged table flattened:
 * [n, keys[n], first[n], counts[n], vals[n * nAggs]] as longs (first = segment << 32 | first selected row; first-seen order). */
JNIEXPORT jlongArray JNICALL Java_immutabledb_gpu_Native_00024_commMergeGroupsAll(JNIEnv *env, jobject self, jlongArray comms, jobjectArray queries,
                                                                                  jobjectArray segments, jint nAggs) {
    jsize n = (*env)->GetArrayLength(env, comms);
    jlong *ch = (*env)->GetLongArrayElements(env, comms, NULL);
    imm3_comm **cs = (imm3_comm **)calloc((size_t)n, sizeof(imm3_comm *));
    imm3_query ***qs = (imm3_query ***)calloc((size_t)n, sizeof(imm3_query **));
    int32_t **ss = (int32_t **)calloc((size_t)n, sizeof(int32_t *));
    int32_t *nq = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    uint64_t *keys = NULL, *first = NULL, *counts = NULL;
    int64_t *vals = NULL;
    jlongArray out = NULL;
    uint32_t g = 0;
    for (jsize i = 0; i < n; i++) {
        jlongArray qa = (jlongArray)(*env)->GetObjectArrayElement(env, queries, i);
        jintArray sa = (jintArray)(*env)->GetObjectArrayElement(env, segments, i);
        jsize m = (*env)->GetArrayLength(env, qa);
        jlong *qh = (*env)->GetLongArrayElements(env, qa, NULL);
        jint *sh = (*env)->GetIntArrayElements(env, sa, NULL);
        cs[i] = (imm3_comm *)(intptr_t)ch[i];
        nq[i] = (int32_t)m;
        qs[i] = (imm3_query **)calloc((size_t)(m > 0 ? m : 1), sizeof(imm3_query *));
        ss[i] = (int32_t *)calloc((size_t)(m > 0 ? m : 1), sizeof(int32_t));
        for (jsize k = 0; k < m; k++) { qs[i][k] = (imm3_query *)(intptr_t)qh[k]; ss[i][k] = (int32_t)sh[k]; }
        (*env)->ReleaseLongArrayElements(env, qa, qh, JNI_ABORT);
        (*env)->ReleaseIntArrayElements(env, sa, sh, JNI_ABORT);
        (*env)->DeleteLocalRef(env, qa);
        (*env)->DeleteLocalRef(env, sa);
    }
    /* the stride of `vals` is the LIBRARY's number of aggregates (imm3_comm_merge_groups_all writes that many per group: the C ABI
     * takes no stride): sized from the query handles, and a caller that believes otherwise gets an exception, not a heap overrun */
    {
        int32_t libAggs = -1;
        for (jsize i = 0; i < n && libAggs < 0; i++)
            if (nq[i] > 0) CHECKED(imm3_query_agg_shape(qs[i][0], NULL, &libAggs, NULL));
        if (libAggs >= 0 && libAggs != (int32_t)nAggs) {
            jclass cls = (*env)->FindClass(env, "java/lang/Exception");
            if (cls) (*env)->ThrowNew(env, cls, "commMergeGroupsAll: nAggs does not match the queries' number of aggregates");
            goto done;
        }
    }
    CHECKED(imm3_comm_merge_groups_all(cs, (int32_t)n, (imm3_query *const *const *)qs, (const int32_t *const *)ss, nq, NULL, NULL, NULL, NULL, 0, &g));
    keys = (uint64_t *)calloc(g ? g : 1, sizeof(uint64_t));
    first = (uint64_t *)calloc(g ? g : 1, sizeof(uint64_t));
    counts = (uint64_t *)calloc(g ? g : 1, sizeof(uint64_t));
    vals = (int64_t *)calloc((size_t)(g ? g : 1) * (size_t)(nAggs > 0 ? nAggs : 1), sizeof(int64_t));
    CHECKED(imm3_comm_merge_groups_all(cs, (int32_t)n, (imm3_query *const *const *)qs, (const int32_t *const *)ss, nq, keys, first, counts, vals, g, &g));
    {
        const jsize total = (jsize)(1 + 3 * (size_t)g + (size_t)g * (size_t)nAggs);
        jlong head = (jlong)g;
        out = (*env)->NewLongArray(env, total);
        (*env)->SetLongArrayRegion(env, out, 0, 1, &head);
        (*env)->SetLongArrayRegion(env, out, 1, (jsize)g, (const jlong *)keys);
        (*env)->SetLongArrayRegion(env, out, 1 + (jsize)g, (jsize)g, (const jlong *)first);
        (*env)->SetLongArrayRegion(env, out, 1 + 2 * (jsize)g, (jsize)g, (const jlong *)counts);
        (*env)->SetLongArrayRegion(env, out, 1 + 3 * (jsize)g, (jsize)((size_t)g * (size_t)nAggs), (const jlong *)vals);
    }
done:
    for (jsize i = 0; i < n; i++) { free(qs[i]); free(ss[i]); }
    (*env)->ReleaseLongArrayElements(env, comms, ch, JNI_ABORT);
    free(qs); free(ss); free(cs); free(nq); free(keys); free(first); free(counts); free(vals);
    return out;
}
#endif /* IMM3_HAVE_JNI */
