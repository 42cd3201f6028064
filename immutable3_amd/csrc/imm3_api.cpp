// imm3_api.cpp -- the C ABI of include/imm3.h: host-side planning (validation that mirrors the reference's
// exceptions, predicate folding, batch layout) and launch orchestration of the HIP kernels.
// Compiled with hipcc for gfx950.  There is NO CPU fallback: without a HIP device every entry point that
// touches data fails with IMM3_ERR_DEVICE.
#include "../../include/imm3.h"
#include "../../include/imm3_diag.h"
#include "imm3_internal.h"
#include "../host/codec.hpp"

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <unordered_map>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "imm3_handles.h"
#include "imm3_api_internal.h" // (shared with the planner, imm3_planner.cpp: which plan a projection takes, and at what P / grid)

using namespace imm3;

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;

int imm3::fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

// Entry into a context (imm3_sync.h): the call passes the context's capture gate -- shared, so calls of any number of
// threads run side by side; while ANOTHER thread has a graph capture open it waits here until that capture ends.
// (Declares a scope guard: one use per function scope.)
#define CTX_LIVE_RUN(c)                                                                           \
    if (!(c)) return fail(IMM3_ERR_ARG, "ctx is null");                                           \
    imm3::GateScope imm3_gate_scope_(&(c)->gate);                                                 \
    if ((c)->closed) return fail(IMM3_ERR_STATE, "the context of this handle has been destroyed")
// every entry point but the run calls: not while THIS thread's graph capture is open (most of them synchronise or allocate)
#define CTX_LIVE(c)                                                                               \
    CTX_LIVE_RUN(c);                                                                              \
    if ((c)->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context: only imm3_query_run / imm3_query_run_select and imm3_ctx_capture_end are accepted")

// ---------------------------------------------------------------------------------------------
// context-level caching allocator (imm3_sync.h: BlockPool, one per context, thread-safe)
// ---------------------------------------------------------------------------------------------
hipError_t imm3::pool_alloc(imm3_ctx *ctx, void **out, size_t bytes) { return (hipError_t)ctx->blocks.alloc(out, bytes); }
void imm3::pool_release(imm3_ctx *ctx, void *p) { ctx->blocks.release(p); }
static void pool_drain(imm3_ctx *ctx) { ctx->blocks.drain(); }

// ---------------------------------------------------------------------------------------------
// scalar rules shared with the reference (JVM d2i / i2b): Select.scala:65,73; SURVEY Appendix A.1 rule 5
// ---------------------------------------------------------------------------------------------
static int32_t jvm_d2i(double d) {
    if (d != d) return 0;
    if (d >= 2147483647.0) return INT32_MAX;
    if (d <= -2147483648.0) return INT32_MIN;
    return (int32_t)d;
}
static int32_t jvm_d2b(double d) { return (int32_t)(int8_t)(uint8_t)((uint32_t)jvm_d2i(d) & 0xFFu); }

static const char *cond_name(int c) {
    switch (c) {
    case IMM3_MATCH: return "Match";
    case IMM3_NOTMATCH: return "NotMatch";
    case IMM3_EQ: return "EQ";
    case IMM3_GT: return "GT";
    case IMM3_LT: return "LT";
    case IMM3_NOOP: return "NoOp";
    default: return "?";
    }
}

// ---------------------------------------------------------------------------------------------
// library / context
// ---------------------------------------------------------------------------------------------
extern "C" int imm3_abi_version(void) { return IMM3_ABI_VERSION; }
extern "C" const char *imm3_last_error(void) { return g_err.c_str(); }

extern "C" int imm3_device_count(int *count) {
    if (!count) return fail(IMM3_ERR_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(IMM3_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return IMM3_OK;
}

extern "C" int imm3_ctx_create(int device, void *stream, imm3_ctx **out) {
    if (!out) return fail(IMM3_ERR_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (n <= 0) return fail(IMM3_ERR_DEVICE, "no HIP device: the immutable3 GPU path has no CPU fallback");
    if (device < 0 || device >= n) return fail(IMM3_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    std::unique_ptr<imm3_ctx> c(new imm3_ctx());
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    *out = c.release();
    return IMM3_OK;
}

void imm3::ctx_retain(imm3_ctx *c) { ref_retain(c->refs); }
void imm3::ctx_release(imm3_ctx *c) {
    if (ref_release(c->refs)) delete c;
}

// a query's device buffers are about to move (or go): every graph that recorded one of its runs points at the old ones
void imm3::graphs_mark_stale(imm3_ctx *ctx, const imm3_query *q) {
    std::lock_guard<std::mutex> g(ctx->mu);
    for (imm3_graph *gr : ctx->graphs)
        if (std::find(gr->queries.begin(), gr->queries.end(), q) != gr->queries.end()) gr->stale = true;
}

// Destroying a context with live segments / tables / queries is allowed (see imm3_handles.h, "Lifetimes"): everything the
// context owns on the device goes now, the struct itself when the last dependant is destroyed.
extern "C" int imm3_ctx_destroy(imm3_ctx *ctx) {
    if (!ctx) return IMM3_OK;
    if (ctx->closed) return fail(IMM3_ERR_STATE, "context destroyed twice");
    (void)hipSetDevice(ctx->device);
    if (ctx->capture) { // an abandoned capture: end it and drop what it recorded.  The imm3_graph was never handed to the
        hipGraph_t g = nullptr; // caller, so nobody else can free it or the context reference it holds
        (void)hipStreamEndCapture(ctx->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        delete ctx->capture;
        ctx->capture = nullptr;
        if (ctx->gate.owned_by_me()) ctx->gate.end_exclusive();
        ctx_release(ctx); // the reference imm3_ctx_capture_begin took
    }
    // (the caller's contract: no other thread is inside a call on this context or its children while it is destroyed)
    (void)hipStreamSynchronize(ctx->stream);
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (imm3_graph *g : ctx->graphs) { // the handles stay valid (imm3_graph_destroy frees them); what they recorded is gone
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        if (g->graph) (void)hipGraphDestroy(g->graph);
        g->exec = nullptr;
        g->graph = nullptr;
        g->stale = true;
    }
    ctx->graphs.clear();
    for (auto &r : ctx->pool) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    ctx->pool.clear();
    ctx->used = 0;
    ctx->timing = false;
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); ctx->aux = nullptr; }
    if (ctx->copy) { (void)hipStreamSynchronize(ctx->copy); (void)hipStreamDestroy(ctx->copy); ctx->copy = nullptr; }
    (void)hipFree(ctx->d_stamps);
    ctx->d_stamps = nullptr;
    ctx->stamp_slots = 0;
    (void)hipFree(ctx->d_xpow8);
    ctx->d_xpow8 = nullptr;
    pool_drain(ctx);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = nullptr;
    ctx->closed = true;
    ctx_release(ctx);
    return IMM3_OK;
}

extern "C" int imm3_ctx_sync(imm3_ctx *ctx) {
    CTX_LIVE(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->aux) HIPCHK(hipStreamSynchronize(ctx->aux));
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// graphs: a recorded sequence of query runs, enqueued with one call (hipGraph)
// ---------------------------------------------------------------------------------------------
extern "C" int imm3_ctx_capture_begin(imm3_ctx *ctx) {
    if (!ctx) return fail(IMM3_ERR_ARG, "ctx is null");
    if (ctx->closed) return fail(IMM3_ERR_STATE, "the context of this handle has been destroyed");
    if (ctx->gate.owned_by_me()) return fail(IMM3_ERR_STATE, "a graph capture is open on this context: only imm3_query_run / imm3_query_run_select and imm3_ctx_capture_end are accepted");
    // exclusive from here to imm3_ctx_capture_end: calls of other threads on this context wait (whatever they enqueued
    // on the capturing stream would be recorded into the graph)
    if (!ctx->gate.begin_exclusive()) return fail(IMM3_ERR_STATE, "imm3_ctx_capture_begin from inside another call on this context");
    hipError_t e = hipSetDevice(ctx->device);
    imm3_graph *g = new imm3_graph();
    g->ctx = ctx;
    if (e == hipSuccess) e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        delete g;
        (void)hipGetLastError();
        ctx->gate.end_exclusive();
        return fail(IMM3_ERR_DEVICE, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
    }
    ctx_retain(ctx);
    ctx->capture = g;
    return IMM3_OK;
}

extern "C" int imm3_ctx_capture_end(imm3_ctx *ctx, imm3_graph **out) {
    if (!out) return fail(IMM3_ERR_ARG, "null argument");
    if (!ctx) return fail(IMM3_ERR_ARG, "ctx is null");
    if (ctx->closed) return fail(IMM3_ERR_STATE, "the context of this handle has been destroyed");
    if (!ctx->gate.owned_by_me() || !ctx->capture) return fail(IMM3_ERR_STATE, "no capture is open on this context (begin and end belong to one thread)");
    imm3_graph *g = ctx->capture;
    ctx->capture = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipStreamEndCapture(ctx->stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        ctx->gate.end_exclusive();
        ctx_release(ctx);
        return fail(IMM3_ERR_DEVICE, std::string("graph capture failed: ") + hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->graphs.push_back(g);
    }
    ctx->gate.end_exclusive();
    *out = g;
    return IMM3_OK;
}

extern "C" int imm3_graph_launch(imm3_graph *g) {
    if (!g) return fail(IMM3_ERR_ARG, "graph is null");
    CTX_LIVE(g->ctx);
    {
        std::lock_guard<std::mutex> lk(g->ctx->mu);
        if (g->stale || !g->exec) return fail(IMM3_ERR_STATE, "a query recorded in this graph has been destroyed or has moved its buffers: record the graph again");
    }
    HIPCHK(hipSetDevice(g->ctx->device));
    HIPCHK(hipGraphLaunch(g->exec, g->ctx->stream));
    // A replay is the recorded runs again: every recorded query gets back the state it had after its recorded run (a getter of a
    // single-pass run then reads THIS replay's status word -- round 3 left `sp_verified` set by an earlier getter, or
    // `ran_single_pass` cleared by an earlier fallback, and a busy or abandoned replay went unnoticed).
    for (size_t i = 0; i < g->queries.size() && i < g->states.size(); ++i) {
        imm3_query *q = g->queries[i];
        const QueryRunState &st = g->states[i];
        q->ran_select = st.ran_select;
        q->ran_project = st.ran_project;
        q->bitmap_valid = st.bitmap_valid;
        q->ran_single_pass = st.ran_single_pass;
        q->stage_written = st.stage_written;
        q->bitmap_lazy = st.bitmap_lazy;
        q->agg_select_skipped = st.agg_select_skipped;
        q->count_pending_scan = st.count_pending_scan;
        q->has_pfor_pass = st.has_pfor_pass;
        q->ran_agg = st.ran_agg;
        q->offsets_valid = st.offsets_valid;
        q->select_partial = st.select_partial;
        q->sp_verified = false;
    }
    return IMM3_OK;
}

extern "C" int imm3_graph_destroy(imm3_graph *g) {
    if (!g) return IMM3_OK;
    imm3_ctx *ctx = g->ctx;
    imm3::GateScope gate(&ctx->gate);
    if (!ctx->closed) {
        if (ctx->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context");
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream); // a launch may still be running
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto &gs = ctx->graphs;
        gs.erase(std::remove(gs.begin(), gs.end(), g), gs.end());
    }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    ctx_release(ctx);
    return IMM3_OK;
}

extern "C" int imm3_ctx_stream(imm3_ctx *ctx, void **stream_out) {
    if (!stream_out) return fail(IMM3_ERR_ARG, "null argument");
    CTX_LIVE(ctx);
    *stream_out = (void *)ctx->stream;
    return IMM3_OK;
}

extern "C" int imm3_ctx_set_tuning(imm3_ctx *ctx, int32_t filter_variant, int32_t grid_blocks) {
    CTX_LIVE(ctx);
    ctx->filter_variant = filter_variant;
    ctx->grid_blocks = grid_blocks;
    return IMM3_OK;
}

// Fault injection into the single-pass projection kernel (imm3_diag.h).  The fields travel in every launch's arguments; only the
// tools' build of the kernel (-DIMM3_ABLATE) reads them.
extern "C" int imm3_ctx_inject_fault(imm3_ctx *ctx, int32_t work_group, int32_t span, uint32_t max_polls) {
    CTX_LIVE(ctx);
#ifndef IMM3_ABLATE
    if (work_group >= 0 || max_polls) return fail(IMM3_ERR_STATE, "fault injection exists only in the tools' build of the library (make -C csrc ablate)");
#endif
    ctx->fault_wg = work_group;
    ctx->fault_span = span;
    ctx->fault_max_polls = max_polls;
    return IMM3_OK;
}

#ifdef IMM3_ABLATE
static int single_pass_lock_word(int device, unsigned long long **out);
#endif

// The device's ticket word of the single-pass kernel (0 = free).  Tests put a foreign ticket there to make every launch find the
// device busy, deterministically; imm3_project.hip and run_single_pass say what the word is for.
extern "C" int imm3_ctx_debug_device_lock(imm3_ctx *ctx, uint64_t value, uint64_t *previous) {
    CTX_LIVE(ctx);
#ifndef IMM3_ABLATE
    // (the shipped library does not hand out a way to park every one-launch query of a device, in every context, on its fallback)
    (void)value;
    (void)previous;
    return fail(IMM3_ERR_STATE, "the device-lock hook exists only in the tools' build of the library (make -C csrc ablate)");
#else
    HIPCHK(hipSetDevice(ctx->device));
    unsigned long long *word = nullptr;
    const int rc = single_pass_lock_word(ctx->device, &word);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    unsigned long long old = 0, v = (unsigned long long)value;
    HIPCHK(hipMemcpy(&old, word, sizeof(old), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(word, &v, sizeof(v), hipMemcpyHostToDevice));
    if (previous) *previous = old;
    return IMM3_OK;
#endif
}

extern "C" int imm3_ctx_devclock_enable(imm3_ctx *ctx, int32_t max_launches) {
    CTX_LIVE(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipFree(ctx->d_stamps); // the stamp buffer only: the snappy CRC table and the buffer pool are not this call's to free
    ctx->d_stamps = nullptr;
    ctx->stamp_slots = 0;
    ctx->stamp_used = 0;
    ctx->stamp_grids.clear();
    if (max_launches > 0) {
        void *p = nullptr;
        HIPCHK(hipMalloc(&p, (size_t)max_launches * kMaxFilterGrid * 2 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(p, 0, (size_t)max_launches * kMaxFilterGrid * 2 * sizeof(unsigned long long)));
        HIPCHK(hipStreamSynchronize(nullptr)); // (the memset runs on the null stream and may return early; the context's stream does not wait for that stream)
        ctx->d_stamps = (unsigned long long *)p;
        ctx->stamp_slots = max_launches;
    }
    return IMM3_OK;
}

extern "C" int imm3_ctx_devclock_collect(imm3_ctx *ctx, float *ms_out, int32_t cap, int32_t *n_out) {
    if (!n_out) return fail(IMM3_ERR_ARG, "null argument");
    CTX_LIVE(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int32_t n = ctx->stamp_used;
    std::vector<unsigned long long> h((size_t)kMaxFilterGrid * 2);
    for (int32_t i = 0; i < n && i < cap; ++i) {
        const int32_t g = ctx->stamp_grids[(size_t)i];
        HIPCHK(hipMemcpy(h.data(), ctx->d_stamps + (size_t)i * kMaxFilterGrid * 2, (size_t)g * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ULL, hi = 0;
        for (int32_t b = 0; b < g; ++b) { lo = std::min(lo, h[(size_t)2 * b]); hi = std::max(hi, h[(size_t)2 * b + 1]); }
        if (ms_out) ms_out[i] = (float)((double)(hi - lo) / 100000.0); // 100 MHz ticks -> ms
    }
    *n_out = n;
    return IMM3_OK;
}

extern "C" int imm3_ctx_devclock_raw(imm3_ctx *ctx, int32_t launch, uint64_t *out, int32_t n) {
    if (!out || n < 0) return fail(IMM3_ERR_ARG, "null argument");
    CTX_LIVE(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (launch < 0 || launch >= ctx->stamp_used) return fail(IMM3_ERR_ARG, "no such launch");
    if (n > kMaxFilterGrid * 2) n = kMaxFilterGrid * 2;
    HIPCHK(hipMemcpy(out, ctx->d_stamps + (size_t)launch * kMaxFilterGrid * 2, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return IMM3_OK;
}

extern "C" int imm3_ctx_measure_read_gbps(imm3_ctx *ctx, uint64_t bytes, int32_t iters, double *gbps) {
    if (!gbps) return fail(IMM3_ERR_ARG, "null argument");
    CTX_LIVE(ctx);
    if (iters < 1) iters = 1;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n_tiles = (int64_t)(bytes / (kTileRows * 4));
    if (n_tiles < 1) return fail(IMM3_ERR_ARG, "bytes too small");
    const size_t sz = (size_t)n_tiles * kTileRows * 4;
    void *buf[3] = {nullptr, nullptr, nullptr};
    void *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    for (auto &b : buf) {
        HIPCHK(hipMalloc(&b, sz));
        HIPCHK(hipMemsetAsync(b, 0x11, sz, ctx->stream));
    }
    HIPCHK(hipMalloc(&sink, 64));
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int i = 0; i < iters + 3; ++i) {
        launch_read_stream((const int32_t *)buf[i % 3], n_tiles, (int32_t *)sink, ctx->stream, e0, e1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventSynchronize(e1));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, e0, e1));
        if (i >= 3) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    *gbps = (double)sz / (ms[ms.size() / 2] * 1e-3) / 1e9;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (auto &b : buf) (void)hipFree(b);
    (void)hipFree(sink);
    return IMM3_OK;
}

extern "C" int imm3_ctx_timing_enable(imm3_ctx *ctx, int32_t max_records) {
    CTX_LIVE(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(ctx->mu);
    while ((int32_t)ctx->pool.size() < max_records) {
        TimingRecord r{};
        HIPCHK(hipEventCreate(&r.start));
        HIPCHK(hipEventCreate(&r.stop));
        ctx->pool.push_back(r);
    }
    ctx->timing = max_records > 0;
    ctx->used = 0;
    return IMM3_OK;
}

extern "C" int imm3_ctx_timing_mask(imm3_ctx *ctx, uint32_t kernel_mask) {
    CTX_LIVE(ctx);
    ctx->timing_mask = kernel_mask;
    return IMM3_OK;
}

extern "C" int imm3_ctx_timing_reset(imm3_ctx *ctx) {
    CTX_LIVE(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->used = 0;
    return IMM3_OK;
}

extern "C" int imm3_ctx_timing_collect(imm3_ctx *ctx, int32_t kernel_id, float *ms_out, int32_t cap, int32_t *n_out) {
    if (!n_out) return fail(IMM3_ERR_ARG, "null argument");
    CTX_LIVE(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->aux) HIPCHK(hipStreamSynchronize(ctx->aux));
    std::lock_guard<std::mutex> lk(ctx->mu);
    int32_t n = 0;
    for (size_t i = 0; i < ctx->used; ++i) {
        if (ctx->pool[i].kernel_id != kernel_id) continue;
        if (n < cap && ms_out) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ctx->pool[i].start, ctx->pool[i].stop));
            ms_out[n] = ms;
        }
        ++n;
    }
    *n_out = n;
    return IMM3_OK;
}

namespace {
// Hands out one event pair per launch when timing is on; the launcher stamps it with the kernel's own
// start and end (hipExtLaunchKernelGGL), so elapsed time == kernel duration as rocprofv3 reports it.
struct LaunchTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    LaunchTimer(imm3_ctx *ctx, int32_t id) {
        if (!ctx->timing.load(std::memory_order_relaxed) || ctx->capture || !((ctx->timing_mask.load(std::memory_order_relaxed) >> id) & 1u)) return;
        std::lock_guard<std::mutex> lk(ctx->mu); // (diagnostics only: the lock is taken when timing is on)
        if (ctx->used < ctx->pool.size()) {
            TimingRecord &rec = ctx->pool[ctx->used++];
            rec.kernel_id = id;
            start = rec.start;
            stop = rec.stop;
        }
    }
};
} // namespace

// ---------------------------------------------------------------------------------------------
// segment
// ---------------------------------------------------------------------------------------------
static constexpr uint64_t kPad = 16384; // readable slack past every column: a partial last tile is read as a whole (1024 rows x <= 16 B)

// Host ranges pinned in place for asynchronous staging are counted per context: two segments staged from the same buffer
// share one registration, and the range stays pinned until the last of them has finished with it (unpinning it under a
// copy still in flight is an error the runtime reports much later, on an unrelated call).
static bool pin_range(imm3_ctx *ctx, void *p, size_t bytes) {
    std::lock_guard<std::mutex> g(ctx->mu);
    auto it = ctx->pinned.find(p);
    if (it != ctx->pinned.end()) { ++it->second; return true; }
    hipError_t re = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (re != hipSuccess) { (void)hipGetLastError(); re = hipHostRegister(p, bytes, hipHostRegisterReadOnly); }
    if (re != hipSuccess) { (void)hipGetLastError(); return false; }
    ctx->pinned[p] = 1;
    return true;
}
static void unpin_range(imm3_ctx *ctx, void *p) {
    std::lock_guard<std::mutex> g(ctx->mu);
    auto it = ctx->pinned.find(p);
    if (it == ctx->pinned.end()) return;
    if (--it->second == 0) {
        if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
        ctx->pinned.erase(it);
    }
}

// last reference gone: free the columns (hipFree waits for the device by itself) and let go of the context
static void segment_free(imm3_segment *seg) {
    if (!seg) return;
    if (seg->ctx) (void)hipSetDevice(seg->ctx->device);
    for (auto &c : seg->cols) {
        if (c.owned && c.d_data) (void)hipFree(c.d_data);
        if (c.d_block_off) (void)hipFree(c.d_block_off);
        if (c.d_row_base) (void)hipFree(c.d_row_base);
        if (c.d_dense) (void)hipFree(c.d_dense);
    }
    for (auto &kv : seg->d_sample_ptrs) (void)hipFree(kv.second);
    (void)hipFree(seg->d_sample_rows);
    if (seg->ctx) for (void *p : seg->registered) unpin_range(seg->ctx, p);
    if (seg->ready) (void)hipEventDestroy(seg->ready);
    if (seg->ctx) ctx_release(seg->ctx);
    delete seg;
}
static void segment_retain(const imm3_segment *seg) { const_cast<imm3_segment *>(seg)->refs.fetch_add(1, std::memory_order_relaxed); }
static void segment_release(const imm3_segment *cseg) {
    imm3_segment *seg = const_cast<imm3_segment *>(cseg);
    if (seg->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) segment_free(seg);
}

// PFOR_INT column: the rows of a block are not implied by its byte length; read the count word of every block
// (PFORCodec.scala:19-31 -> compressed(0) = input.length) and index the blocks for the kernels.
static int pfor_index(imm3_ctx *ctx, imm3_segment *seg, SegCol &sc) {
    if (sc.width != 4) return fail(IMM3_ERR_ARG, "width does not match codec");
    const size_t nb = sc.offsets.empty() ? 0 : sc.offsets.size() - 1;
    std::vector<uint32_t> off(nb + 1, 0u);
    for (size_t k = 0; k <= nb && !sc.offsets.empty(); ++k) {
        const int64_t o = sc.offsets[k];
        if (o < 0 || (uint64_t)o > sc.bytes || (o & 3) || (k > 0 && o < sc.offsets[k - 1]))
            return fail(IMM3_ERR_LAYOUT, "PFOR_INT block " + std::to_string(k) + ": offsets must ascend in whole 4-byte words within the segment data");
        off[k] = (uint32_t)o;
    }
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, (nb + 1) * sizeof(uint32_t)));
    sc.d_block_off = (uint32_t *)p;
    HIPCHK(hipMalloc(&p, (nb + 1) * sizeof(uint32_t)));
    sc.d_row_base = (uint32_t *)p;
    seg->device_bytes += 2 * (nb + 1) * sizeof(uint32_t);
    HIPCHK(hipMemcpyAsync(sc.d_block_off, off.data(), (nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    sc.block_rows.assign(nb, 0);
    if (nb) {
        int32_t *d_counts = nullptr;
        HIPCHK(hipMalloc(&p, nb * sizeof(int32_t)));
        d_counts = (int32_t *)p;
        launch_pfor_counts(sc.d_data, sc.d_block_off, (int64_t)nb, d_counts, ctx->stream);
        const hipError_t e1 = hipGetLastError();
        const hipError_t e2 = hipMemcpyAsync(sc.block_rows.data(), d_counts, nb * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        const hipError_t e3 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_counts);
        HIPCHK(e1);
        HIPCHK(e2);
        HIPCHK(e3);
    }
    std::vector<uint32_t> base(nb + 1, 0u);
    int64_t rows = 0;
    sc.tile_aligned = true;
    for (size_t k = 0; k < nb; ++k) {
        const int64_t n = sc.block_rows[k];
        const int64_t words = ((int64_t)off[k + 1] - (int64_t)off[k]) / 4;
        // a block of n values needs at least the count word (+ n / 32 mini-blocks may all have width 0)
        if (n < 0 || words < 1) return fail(IMM3_ERR_LAYOUT, "PFOR_INT block " + std::to_string(k) + " is malformed (no count word)");
        if (k + 1 < nb && n != kTileRows) sc.tile_aligned = false;
        if (n > kTileRows) sc.tile_aligned = false;
        base[k] = (uint32_t)rows;
        rows += n;
        if (rows > 0xFFFFFFFFLL) return fail(IMM3_ERR_LAYOUT, "segment too large");
    }
    base[nb] = (uint32_t)rows;
    sc.rows = rows;
    HIPCHK(hipMemcpyAsync(sc.d_row_base, base.data(), (nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return IMM3_OK;
}

// Snappy-coded column: a block's rows = the uncompressed bytes its chunks declare / width (k_snappy_sizes walks the
// chunk headers); also sizes the LDS windows k_snappy_decode needs.
static constexpr int32_t kSnappyLdsBudget = 64 * 1024 - 1024; // staged block + one chunk (the CRC table takes 1 KiB)

static int snappy_index(imm3_ctx *ctx, imm3_segment *seg, SegCol &sc) {
    const size_t nb = sc.offsets.empty() ? 0 : sc.offsets.size() - 1;
    std::vector<uint32_t> off(nb + 1, 0u);
    uint32_t biggest_block = 0;
    for (size_t k = 0; k <= nb && !sc.offsets.empty(); ++k) {
        const int64_t o = sc.offsets[k];
        if (o < 0 || (uint64_t)o > sc.bytes || (k > 0 && o < sc.offsets[k - 1]))
            return fail(IMM3_ERR_LAYOUT, "snappy block " + std::to_string(k) + ": offsets must ascend within the segment data");
        off[k] = (uint32_t)o;
        if (k > 0) biggest_block = std::max(biggest_block, off[k] - off[k - 1]);
    }
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, (nb + 1) * sizeof(uint32_t)));
    sc.d_block_off = (uint32_t *)p;
    HIPCHK(hipMalloc(&p, (nb + 1) * sizeof(uint32_t)));
    sc.d_row_base = (uint32_t *)p;
    seg->device_bytes += 2 * (nb + 1) * sizeof(uint32_t);
    HIPCHK(hipMemcpyAsync(sc.d_block_off, off.data(), (nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint32_t> sizes(nb, 0u);
    uint32_t max_chunk = 0;
    if (nb) {
        HIPCHK(hipMalloc(&p, (nb + 1) * sizeof(uint32_t)));
        uint32_t *d_sizes = (uint32_t *)p; // [nb] = the largest chunk
        hipError_t e = hipMemsetAsync(d_sizes + nb, 0, sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) {
            launch_snappy_sizes(sc.d_data, sc.d_block_off, (int64_t)nb, d_sizes, d_sizes + nb, ctx->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(sizes.data(), d_sizes, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&max_chunk, d_sizes + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_sizes);
        HIPCHK(e);
    }
    std::vector<uint32_t> base(nb + 1, 0u);
    sc.block_rows.assign(nb, 0);
    int64_t rows = 0;
    for (size_t k = 0; k < nb; ++k) {
        if (sizes[k] == 0xFFFFFFFFu) return fail(IMM3_ERR_LAYOUT, "snappy block " + std::to_string(k) + " is malformed (stream header, chunk header or length preamble)");
        if (sizes[k] % (uint32_t)sc.width) return fail(IMM3_ERR_LAYOUT, "snappy block " + std::to_string(k) + ": uncompressed length is not a multiple of the value width");
        sc.block_rows[k] = (int32_t)(sizes[k] / (uint32_t)sc.width);
        base[k] = (uint32_t)rows;
        rows += sc.block_rows[k];
        if (rows > 0xFFFFFFFFLL) return fail(IMM3_ERR_LAYOUT, "segment too large");
    }
    base[nb] = (uint32_t)rows;
    sc.rows = rows;
    sc.tile_aligned = false;
    sc.in_cap = (int32_t)((biggest_block + 3 + 8 + 15) & ~15u);
    sc.out_cap = (int32_t)((std::max<uint32_t>(max_chunk, 16) + 15) & ~15u);
    if (sc.in_cap + sc.out_cap > kSnappyLdsBudget)
        return fail(IMM3_ERR_LAYOUT, "snappy block of " + std::to_string(biggest_block) + " stored bytes does not fit the GPU decoder's LDS window (stored block + largest chunk <= 63 KiB)");
    HIPCHK(hipMemcpyAsync(sc.d_row_base, base.data(), (nb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return IMM3_OK;
}

// x^(8 n) mod the CRC-32C polynomial (reflected) for n = 0 .. 32768: moves a slice's CRC past the bytes after it
static int ensure_xpow8(imm3_ctx *ctx) {
    std::lock_guard<std::mutex> g(ctx->mu);
    if (ctx->d_xpow8) return IMM3_OK;
    std::vector<uint32_t> t(32769);
    uint32_t c = 0x80000000u; // x^0
    for (size_t n = 0; n < t.size(); ++n) {
        t[n] = c;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u))); // times x^8 = one zero byte
    }
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, t.size() * sizeof(uint32_t)));
    const hipError_t e = hipMemcpy(p, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p); HIPCHK(e); }
    ctx->d_xpow8 = (uint32_t *)p;
    return IMM3_OK;
}

// The decoded (dense, fixed-width) form of a PFOR_INT / snappy column, made once per segment on first need.
static int ensure_dense(imm3_ctx *ctx, const imm3_segment *cseg, int32_t col) {
    imm3_segment *seg = const_cast<imm3_segment *>(cseg);
    SegCol &sc = seg->cols[(size_t)col];
    if (!is_compressed(sc.codec)) return IMM3_OK;
    if (is_snappy(sc.codec)) {
        const int xrc = ensure_xpow8(ctx);
        if (xrc) return xrc;
    }
    std::lock_guard<std::mutex> g(seg->decode_mu);
    if (sc.d_dense) return IMM3_OK;
    void *p = nullptr;
    const size_t bytes = (size_t)sc.rows * (size_t)sc.width + kPad;
    HIPCHK(hipMalloc(&p, bytes));
    uint32_t *d_status = nullptr;
    void *ps = nullptr;
    hipError_t e = hipMalloc(&ps, sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipFree(p); HIPCHK(e); }
    d_status = (uint32_t *)ps;
    uint32_t status = 0;
    PforArgs a;
    std::memset(&a, 0, sizeof(a));
    a.data = sc.d_data;
    a.block_off = sc.d_block_off;
    a.row_base = sc.d_row_base;
    a.n_blocks = (int64_t)sc.block_rows.size();
    a.out = (int32_t *)p;
    a.status = d_status;
    e = hipMemsetAsync(d_status, 0, sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync((uint8_t *)p + (size_t)sc.rows * (size_t)sc.width, 0, kPad, ctx->stream);
    if (e == hipSuccess && a.n_blocks > 0 && sc.codec == IMM3_PFOR_INT) {
        LaunchTimer t(ctx, 5);
        const int64_t want = (a.n_blocks + kWavesPerBlock - 1) / kWavesPerBlock;
        launch_pfor_decode(a, (int)std::min<int64_t>(want, 2048), ctx->stream, t.start, t.stop);
        e = hipGetLastError();
    } else if (e == hipSuccess && a.n_blocks > 0) {
        SnappyArgs sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.data = sc.d_data;
        sa.block_off = sc.d_block_off;
        sa.row_base = sc.d_row_base;
        sa.xpow8 = ctx->d_xpow8;
        sa.n_blocks = a.n_blocks;
        sa.width = sc.width;
        sa.in_cap = sc.in_cap;
        sa.out_cap = sc.out_cap;
        sa.out = (uint8_t *)p;
        sa.status = d_status;
        LaunchTimer t(ctx, 5);
        // one wave per work-group; LDS per group decides how many are resident: ask for up to 16 per CU
        launch_snappy_decode(sa, (int)std::min<int64_t>(sa.n_blocks, 4096), ctx->stream, t.start, t.stop);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&status, d_status, sizeof(status), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_status);
    if (e != hipSuccess || status) {
        (void)hipFree(p);
        HIPCHK(e);
        return fail(IMM3_ERR_LAYOUT, sc.codec == IMM3_PFOR_INT ? "malformed PFOR_INT block (width above 32, data past the block end, or count mismatch)"
                                                              : "malformed snappy block (bad element, length mismatch or CRC-32C mismatch)");
    }
    sc.d_dense = (uint8_t *)p;
    seg->device_bytes += bytes;
    return IMM3_OK;
}

// the query stream must not touch a segment's columns before their copies have landed
static int segment_await(imm3_ctx *ctx, const imm3_segment *seg) {
    if (seg->ready_pending.load(std::memory_order_acquire)) HIPCHK(hipStreamWaitEvent(ctx->stream, seg->ready, 0));
    return IMM3_OK;
}

// Staging: the columns are copied on the context's COPY stream (pageable host memory goes over PCIe at the link rate on
// this platform -- 55-56 GB/s, the same as pinned or registered memory: tools/h2d_probe.py -- so there is nothing to gain
// from bounce buffers; what matters is that staging never occupies or synchronises the query stream).  async = false:
// returns when the host buffers have been consumed; async = true: returns at once, imm3_segment_wait() ends the
// caller's obligation to keep them mapped.
static int segment_build(imm3_ctx *ctx, const imm3_column *cols, int32_t ncols, bool wrap, imm3_segment **out, bool async = false) {
    if (!out) return fail(IMM3_ERR_ARG, "null argument");
    *out = nullptr;
    CTX_LIVE(ctx);
    if (ncols <= 0 || !cols) return fail(IMM3_ERR_ARG, "a segment needs at least one column");
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<imm3_segment, void (*)(imm3_segment *)> seg(new imm3_segment(), segment_free);
    seg->ctx = ctx;
    ctx_retain(ctx);
    seg->cols.resize((size_t)ncols);
    if (!wrap) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ctx->copy) HIPCHK(hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    }
    bool any_compressed = false;
    for (int32_t i = 0; i < ncols; ++i) any_compressed |= is_compressed(cols[i].codec);
    for (int32_t i = 0; i < ncols; ++i) {
        const imm3_column &c = cols[i];
        SegCol &s = seg->cols[(size_t)i];
        if (c.width <= 0) return fail(IMM3_ERR_ARG, "column width must be positive");
        if (c.n_offsets < 0 || (c.n_offsets > 0 && !c.block_offsets)) return fail(IMM3_ERR_ARG, "bad block offset table");
        if (c.dat_bytes > 0 && !c.dat) return fail(IMM3_ERR_ARG, "column data pointer is null");
        s.codec = c.codec;
        s.vcodec = value_codec(c.codec);
        s.width = c.width;
        s.bytes = c.dat_bytes;
        s.offsets.assign(c.block_offsets, c.block_offsets + c.n_offsets);
        if (wrap) {
            if ((uintptr_t)c.dat & 15) return fail(IMM3_ERR_ARG, "wrapped device columns must be 16-byte aligned");
            s.d_data = (uint8_t *)c.dat;
            s.owned = false;
        } else {
            void *p = nullptr;
            HIPCHK(hipMalloc(&p, c.dat_bytes + kPad));
            s.d_data = (uint8_t *)p;
            s.owned = true;
            seg->device_bytes += c.dat_bytes + kPad;
            if (c.dat_bytes && async && !any_compressed) {
                // hipMemcpyAsync from pageable memory blocks the host for the whole copy; pinned IN PLACE (~4 ms per 400 MB,
                // against 7 ms for the copy itself) it returns at once.  A range that cannot be pinned is copied the blocking way.
                if (pin_range(ctx, (void *)c.dat, c.dat_bytes)) seg->registered.push_back((void *)c.dat);
            }
            if (c.dat_bytes) HIPCHK(hipMemcpyAsync(s.d_data, c.dat, c.dat_bytes, hipMemcpyHostToDevice, ctx->copy));
            HIPCHK(hipMemsetAsync(s.d_data + c.dat_bytes, 0, kPad, ctx->copy));
        }
    }
    if (!wrap) {
        HIPCHK(hipEventCreateWithFlags(&seg->ready, hipEventDisableTiming));
        HIPCHK(hipEventRecord(seg->ready, ctx->copy));
        seg->ready_pending.store(true, std::memory_order_release);
        if (!async || any_compressed) { // host buffers may be unmapped after we return: wait for the COPY stream only
            HIPCHK(hipStreamSynchronize(ctx->copy));
            seg->ready_pending.store(false, std::memory_order_release);
        }
    }
    for (auto &sc : seg->cols) {
        if (!is_compressed(sc.codec)) continue;
        const int rc = sc.codec == IMM3_PFOR_INT ? pfor_index(ctx, seg.get(), sc) : snappy_index(ctx, seg.get(), sc);
        if (rc) return rc;
    }
    *out = seg.release();
    return IMM3_OK;
}

extern "C" int imm3_segment_create(imm3_ctx *ctx, const imm3_column *cols, int32_t ncols, imm3_segment **out) {
    return segment_build(ctx, cols, ncols, false, out);
}
extern "C" int imm3_segment_wrap_device(imm3_ctx *ctx, const imm3_column *cols, int32_t ncols, imm3_segment **out) {
    return segment_build(ctx, cols, ncols, true, out);
}
extern "C" int imm3_segment_create_async(imm3_ctx *ctx, const imm3_column *cols, int32_t ncols, imm3_segment **out) {
    return segment_build(ctx, cols, ncols, false, out, true);
}
extern "C" int imm3_segment_wait(imm3_segment *seg) {
    if (!seg) return fail(IMM3_ERR_ARG, "segment is null");
    if (seg->ready_pending.load(std::memory_order_acquire)) {
        HIPCHK(hipSetDevice(seg->ctx->device));
        HIPCHK(hipEventSynchronize(seg->ready));
        seg->ready_pending.store(false, std::memory_order_release);
    }
    for (void *p : seg->registered) unpin_range(seg->ctx, p);
    seg->registered.clear();
    return IMM3_OK;
}

extern "C" int imm3_segment_destroy(imm3_segment *seg) {
    if (!seg) return IMM3_OK;
    if (seg->closed) return fail(IMM3_ERR_STATE, "segment destroyed twice");
    imm3_ctx *ctx = seg->ctx;
    ctx_retain(ctx); // (the gate lives in the context)
    struct Unref { imm3_ctx *c; ~Unref() { ctx_release(c); } } unref{ctx};
    imm3::GateScope gate(&ctx->gate);
    if (!ctx->closed && ctx->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context");
    seg->closed = true;
    if (!seg->ctx->closed) { // (a destroyed context has already drained its streams)
        (void)hipSetDevice(seg->ctx->device);
        if (seg->ready_pending.load()) (void)hipEventSynchronize(seg->ready);
        (void)hipStreamSynchronize(seg->ctx->stream);
    }
    for (void *p : seg->registered) unpin_range(seg->ctx, p); // the caller may unmap its buffers after destroy
    seg->registered.clear();
    segment_release(seg); // tables / queries built on it keep the columns alive until they are destroyed
    return IMM3_OK;
}

extern "C" int imm3_segment_bytes(const imm3_segment *seg, uint64_t *device_bytes) {
    if (!seg || !device_bytes) return fail(IMM3_ERR_ARG, "null argument");
    *device_bytes = seg->device_bytes;
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// query planning
// ---------------------------------------------------------------------------------------------
static void table_release(const imm3_table *t);

static void query_free(imm3_query *q) {
    if (!q) return;
    imm3_ctx *ctx = q->ctx;
    if (!ctx) { delete q; return; }
    (void)hipSetDevice(ctx->device);
    // no synchronisation: every buffer goes back to the context's pool and is only ever reused in stream order
    if (ctx->aux && q->total_on_aux) (void)hipStreamSynchronize(ctx->aux);
    pool_release(ctx, q->d_bitmap);
    pool_release(ctx, q->d_tile_offsets);
    pool_release(ctx, q->d_chunk_sums);
    pool_release(ctx, q->d_block_partials);
    pool_release(ctx, q->d_limit_state);
    pool_release(ctx, q->d_total);
    pool_release(ctx, q->d_word_row_base);
    pool_release(ctx, q->d_word_nvalid);
    pool_release(ctx, q->d_row_index);
    for (auto p : q->d_proj) pool_release(ctx, p);
    for (auto &p : q->preds) pool_release(ctx, p.d_blob);
    pool_release(ctx, q->d_stage_rec);
    pool_release(ctx, q->d_tile_start);
    pool_release(ctx, q->d_desc);
    pool_release(ctx, q->d_tile_desc);
    pool_release(ctx, q->d_akeys); pool_release(ctx, q->d_acounts); pool_release(ctx, q->d_okeys); pool_release(ctx, q->d_ocounts);
    pool_release(ctx, q->d_afirst); pool_release(ctx, q->d_ofirst); pool_release(ctx, q->d_ameta);
    pool_release(ctx, q->d_avals); pool_release(ctx, q->d_ovals);
    if (q->ev_filter_done) (void)hipEventDestroy(q->ev_filter_done);
    if (q->ev_total_done) (void)hipEventDestroy(q->ev_total_done);
    // the query kept its inputs alive (imm3_handles.h, "Lifetimes")
    if (q->table) table_release(q->table);
    else if (q->seg) segment_release(q->seg);
    ctx_release(ctx);
    delete q;
}

static int ensure_row_capacity(imm3_query *q, uint64_t rows) {
    if (rows <= q->cap_rows && q->d_row_index) return IMM3_OK;
    if (rows < 1) rows = 1;
    imm3_ctx *ctx = q->ctx;
    if (q->d_row_index) graphs_mark_stale(ctx, q); // a recorded run would write through the old pointers (launch: IMM3_ERR_STATE)
    pool_release(ctx, q->d_row_index); // stream-ordered: a gather still in flight finishes before any reuse
    q->d_row_index = nullptr;
    for (auto &p : q->d_proj) {
        pool_release(ctx, p);
        p = nullptr;
    }
    void *p = nullptr;
    HIPCHK(pool_alloc(ctx, &p, rows * sizeof(uint32_t)));
    q->d_row_index = (uint32_t *)p;
    q->d_proj.assign(q->proj.size(), nullptr);
    for (size_t j = 0; j < q->proj.size(); ++j) {
        const SegCol &sc = q->seg->cols[(size_t)q->used[(size_t)q->proj[j]]];
        HIPCHK(pool_alloc(ctx, &p, rows * (uint64_t)sc.width));
        q->d_proj[j] = (uint8_t *)p;
    }
    q->cap_rows = rows;
    return IMM3_OK;
}

// Batches of ONE segment as ScanOp yields them: the FIRST used column defines them (Scan.scala:55,72); BlockIterator
// takes each block by a relative get from a rewound buffer (Segment.scala:159-168), i.e. from a running cursor.
// Every other used column must hold the same rows in the same blocks, otherwise the reference either throws
// ArrayIndexOutOfBounds (shorter) or silently joins the wrong rows (longer): refused.
// rows of block k of a column: DENSE_* = bytes / width; PFOR_INT = the count the block declares
static inline int64_t block_rows_of(const SegCol &sc, int32_t k, int64_t len) {
    return is_compressed(sc.codec) ? (int64_t)sc.block_rows[(size_t)k] : len / sc.width;
}

static int compute_layout(const imm3_segment *seg, int32_t first_col, SegLayout &L) {
    const SegCol &first = seg->cols[(size_t)first_col];
    const int32_t nb = first.offsets.empty() ? 0 : (int32_t)first.offsets.size() - 1;
    L.size.resize((size_t)nb);
    L.word_off.resize((size_t)nb);
    uint64_t cursor = 0;
    for (int32_t k = 0; k < nb; ++k) {
        const int64_t len = (int64_t)first.offsets[(size_t)k + 1] - (int64_t)first.offsets[(size_t)k];
        if (len < 0) return fail(IMM3_ERR_LAYOUT, "block " + std::to_string(k) + ": negative length (NegativeArraySizeException in the reference)");
        if (!is_compressed(first.codec) && len % first.width) return fail(IMM3_ERR_LAYOUT, "block " + std::to_string(k) + ": byte length is not a multiple of the value width (malformed segment)");
        if (cursor + (uint64_t)len > first.bytes) return fail(IMM3_ERR_LAYOUT, "block " + std::to_string(k) + ": bytes [" + std::to_string(cursor) + ", " + std::to_string(cursor + (uint64_t)len) + ") run past the segment data of " + std::to_string(first.bytes) + " bytes (BufferUnderflowException in the reference)");
        const int64_t n = block_rows_of(first, k, len);
        L.size[(size_t)k] = (int32_t)n;
        L.word_off[(size_t)k] = L.words;
        if (k < nb - 1 && (n % 64)) L.ragged = true;
        L.words += (n + 63) / 64;
        L.rows += n;
        cursor += (uint64_t)len;
    }
    if (L.rows > 0xFFFFFFFFLL) return fail(IMM3_ERR_LAYOUT, "segment too large");
    return IMM3_OK;
}

// does used column `other` hold the same rows in the same blocks as the layout's first column?
static int check_same_blocks(const imm3_segment *seg, const SegLayout &L, int32_t other, size_t i) {
    const SegCol &sc = seg->cols[(size_t)other];
    const int32_t nb = (int32_t)L.size.size();
    const int32_t nbc = sc.offsets.empty() ? 0 : (int32_t)sc.offsets.size() - 1;
    if (nbc < nb) return fail(IMM3_ERR_LAYOUT, "used column " + std::to_string(i) + " has fewer blocks than the first used column (ArrayIndexOutOfBounds in the reference)");
    uint64_t cur = 0;
    for (int32_t k = 0; k < nb; ++k) {
        const int64_t len = (int64_t)sc.offsets[(size_t)k + 1] - (int64_t)sc.offsets[(size_t)k];
        if (len < 0 || (!is_compressed(sc.codec) && len % sc.width) || block_rows_of(sc, k, len) != L.size[(size_t)k])
            return fail(IMM3_ERR_LAYOUT, "used column " + std::to_string(i) + " block " + std::to_string(k) + " does not hold the same rows as the first used column");
        if (cur + (uint64_t)len > sc.bytes) return fail(IMM3_ERR_LAYOUT, "used column " + std::to_string(i) + " block " + std::to_string(k) + " runs past the segment data");
        cur += (uint64_t)len;
    }
    return IMM3_OK;
}

// The layout for these used columns, from the segment's cache (segments are immutable once created; any thread, any context).
static int segment_layout(const imm3_segment *cseg, const std::vector<int32_t> &used, std::shared_ptr<const SegLayout> &out) {
    imm3_segment *seg = const_cast<imm3_segment *>(cseg);
    std::lock_guard<std::mutex> g(seg->layout_mu);
    auto it = seg->layouts.find(used[0]);
    if (it == seg->layouts.end()) {
        auto L = std::make_shared<SegLayout>();
        const int rc = compute_layout(seg, used[0], *L);
        if (rc) return rc; // (a malformed column is reported every time it is asked for: nothing cached)
        it = seg->layouts.emplace(used[0], std::move(L)).first;
    }
    for (size_t i = 1; i < used.size(); ++i) {
        if (used[i] == used[0]) continue;
        const auto key = std::make_pair(used[0], used[i]);
        if (seg->same_blocks.count(key)) continue;
        const int rc = check_same_blocks(seg, *it->second, used[i], i);
        if (rc) return rc;
        seg->same_blocks[key] = true;
    }
    out = it->second;
    return IMM3_OK;
}

// Can this folded predicate go through the tile kernel?
static int query_create_impl(imm3_ctx *ctx, const imm3_segment *seg, const imm3_table *table,
                                 const int32_t *used_cols, int32_t n_used,
                                 const imm3_select *sels, int32_t n_sels,
                                 const int32_t *proj, int32_t n_proj, int64_t limit,
                                 int32_t table_block_size, imm3_query **out) {
    if (!seg || !out) return fail(IMM3_ERR_ARG, "null argument");
    *out = nullptr;
    CTX_LIVE(ctx);
    if (seg->closed || (table && table->closed)) return fail(IMM3_ERR_STATE, "the segment / table has been destroyed");
    if (seg->ctx->device != ctx->device) return fail(IMM3_ERR_ARG, "segment lives on another device");
    if (n_used <= 0 || !used_cols) return fail(IMM3_ERR_ARG, "a scan needs at least one used column");
    if (n_sels < 0 || (n_sels > 0 && !sels)) return fail(IMM3_ERR_ARG, "bad select list");
    if (n_proj < 0 || (n_proj > 0 && !proj)) return fail(IMM3_ERR_ARG, "bad project list");
    const int32_t nsegcols = (int32_t)seg->cols.size();
    for (int32_t i = 0; i < n_used; ++i)
        if (used_cols[i] < 0 || used_cols[i] >= nsegcols) return fail(IMM3_ERR_ARG, "used column index out of range");
    for (int32_t i = 0; i < n_sels; ++i)
        if (sels[i].column < 0 || sels[i].column >= n_used) return fail(IMM3_ERR_ARG, "select column is not among the used columns");
    for (int32_t i = 0; i < n_proj; ++i)
        if (proj[i] < 0 || proj[i] >= n_used) return fail(IMM3_ERR_ARG, "project column is not among the used columns");
    HIPCHK(hipSetDevice(ctx->device));
    if (table) {
        for (const imm3_segment *sg : table->segs) { const int wrc = segment_await(ctx, sg); if (wrc) return wrc; }
    } else {
        const int wrc = segment_await(ctx, seg);
        if (wrc) return wrc;
    }

    // (1) SelectOp.iterator (Select.scala:17-23) rejects NotMatch / NoOp when the chain is built,
    //     whether or not the segment has any block.
    for (int32_t i = 0; i < n_sels; ++i) {
        const int c = sels[i].cond;
        if (c != IMM3_MATCH && c != IMM3_GT && c != IMM3_LT && c != IMM3_EQ)
            return fail(IMM3_ERR_UNSUPPORTED_CONDITION, std::string("Unsupported condition: ") + cond_name(c));
        if (c == IMM3_MATCH && sels[i].n_match > 0 && (!sels[i].match_bytes || !sels[i].match_lens))
            return fail(IMM3_ERR_ARG, "Match without values");
    }

    std::unique_ptr<imm3_query, void (*)(imm3_query *)> q(new imm3_query(), query_free);
    q->ctx = ctx;
    ctx_retain(ctx);
    q->seg = seg;
    q->table = table;
    if (table) const_cast<imm3_table *>(table)->refs.fetch_add(1, std::memory_order_relaxed); // (the table holds its segments)
    else segment_retain(seg);
    q->table_block_size = table_block_size;
    q->used.assign(used_cols, used_cols + n_used);
    q->proj.assign(proj, proj + n_proj);
    q->limit = limit;

    // (2) batches (segment_layout); a table query concatenates the segments' batches, each segment's bitmap starting
    //     on a fresh tile of the virtual row space
    int32_t nb = 0;
    if (!table) {
        const int lrc = segment_layout(seg, q->used, q->layout);
        if (lrc) return lrc;
        const SegLayout &L = *q->layout;
        nb = (int32_t)L.size.size();
        q->n_rows = L.rows;
        q->n_words = L.words;
        q->ragged = L.ragged;
        q->n_tiles = (q->n_words + kTileWords - 1) / kTileWords;
    } else {
        // every column of a table shares one block layout (imm3_table_create), so the batches do not depend on which
        // columns are used: they live in the table
        nb = (int32_t)table->batch_size.size();
        q->n_rows = table->n_rows;
        q->n_tiles = table->n_tiles;
        q->n_words = table->n_tiles * kTileWords; // virtual: every segment padded to whole tiles
    }
    q->n_chunks = (q->n_tiles + kChunkTiles - 1) / kChunkTiles;

    if (nb >= 1) {
        // ScanOp.next dispatches on the codec of every used column (Scan.scala:37-50) ...
        for (int32_t i = 0; i < n_used; ++i) {
            const SegCol &sc = seg->cols[(size_t)q->used[(size_t)i]];
            // PFOR_INT: the reference dispatches it too (Scan.scala:37-39) but its decode throws on every block
            // (PFORCodec.scala:43-50); here the blocks its encoder writes are decoded (imm3_codec.hip).
            // Snappy-coded columns (IMM3_SNAPPY_*) are this library's extension: the reference only has the encoder.
            if (sc.vcodec != IMM3_DENSE_INT && sc.vcodec != IMM3_DENSE_TINYINT && sc.vcodec != IMM3_DENSE_STRING)
                return fail(IMM3_ERR_NO_CODEC, "No implementation for codec " + std::to_string(sc.codec));
            if ((sc.vcodec == IMM3_DENSE_INT && sc.width != 4) || (sc.vcodec == IMM3_DENSE_TINYINT && sc.width != 1))
                return fail(IMM3_ERR_ARG, "width does not match codec");
        }
        // ... and each SelectIterator dispatches on the vector type (Select.scala:41,80,118,156).
        for (int32_t i = 0; i < n_sels; ++i) {
            const SegCol &sc = seg->cols[(size_t)q->used[(size_t)sels[i].column]];
            const bool is_str = sc.vcodec == IMM3_DENSE_STRING;
            if ((sels[i].cond == IMM3_MATCH) != is_str) return fail(IMM3_ERR_UNSUPPORTED_VECTOR, "Unsupported column vector");
        }
    }

    // (3) fold the SelectOp leaves per column.  Every leaf only clears bits (Select.scala:37,68,106,144) and
    //     runOps ignores AND/OR (Engine.scala:240), so the chain is a conjunction and order is irrelevant.
    if (nb >= 1) {
        for (int32_t i = 0; i < n_sels; ++i) {
            const int32_t sci = q->used[(size_t)sels[i].column];
            const SegCol &sc = seg->cols[(size_t)sci];
            FoldedPred *fp = nullptr;
            for (auto &p : q->preds)
                if (p.seg_col == sci) fp = &p;
            const bool fresh = !fp;
            if (fresh) {
                q->preds.emplace_back();
                fp = &q->preds.back();
                fp->seg_col = sci;
                fp->width = sc.width;
                if (sc.vcodec == IMM3_DENSE_INT) { fp->kind = KIND_I32; fp->lo = INT32_MIN; fp->hi = INT32_MAX; }
                else if (sc.vcodec == IMM3_DENSE_TINYINT) { fp->kind = KIND_I8; fp->lo = -128; fp->hi = 127; }
                else fp->kind = KIND_STR;
            }
            if (fp->kind == KIND_STR) {
                std::vector<std::string> vals;
                int64_t off = 0;
                for (int32_t m = 0; m < sels[i].n_match; ++m) {
                    const int32_t len = sels[i].match_lens[m];
                    if (len < 0) return fail(IMM3_ERR_ARG, "negative match length");
                    // String.equals can only hold for a value of exactly `width` bytes (DataType.scala:69-70)
                    if (len == sc.width) {
                        std::string v((const char *)sels[i].match_bytes + off, (size_t)len);
                        if (std::find(vals.begin(), vals.end(), v) == vals.end()) vals.push_back(v);
                    }
                    off += len;
                }
                if (fresh) fp->match = vals;
                else {
                    std::vector<std::string> both;
                    for (auto &v : fp->match)
                        if (std::find(vals.begin(), vals.end(), v) != vals.end()) both.push_back(v);
                    fp->match = both;
                }
            } else {
                const int64_t t = fp->kind == KIND_I32 ? (int64_t)jvm_d2i(sels[i].value) : (int64_t)jvm_d2b(sels[i].value);
                if (sels[i].cond == IMM3_GT) fp->lo = std::max(fp->lo, t + 1);      // strict >, Select.scala:68,76
                else if (sels[i].cond == IMM3_LT) fp->hi = std::min(fp->hi, t - 1); // strict <, Select.scala:106,114
                else { fp->lo = std::max(fp->lo, t); fp->hi = std::min(fp->hi, t); } // ==, Select.scala:144,152
            }
        }
        // PFOR_INT columns: a predicate-only column of a tile-aligned segment is evaluated on its compressed blocks
        // (k_filter_pfor); anything else reads the decoded column, made once per segment.
        for (int32_t i = 0; i < n_used; ++i) {
            const int32_t sci = q->used[(size_t)i];
            const SegCol &sc = seg->cols[(size_t)sci];
            if (!is_compressed(sc.codec)) continue;
            bool projected = false;
            for (int32_t pj : q->proj) projected |= (pj == i);
            FoldedPred *fp = nullptr;
            for (auto &p : q->preds)
                if (p.seg_col == sci) fp = &p;
            const bool fused = sc.codec == IMM3_PFOR_INT && fp && !table && !q->ragged && sc.tile_aligned && !projected &&
                               ctx->filter_variant != 1 && ctx->filter_variant != 5;
            if (fused) fp->pfor = true;
            else if (!table) { // a table decodes its PFOR_INT columns when it is created
                const int drc = ensure_dense(ctx, seg, sci);
                if (drc) return drc;
            }
        }
        for (auto &p : q->preds) {
            if (p.kind == KIND_STR ? p.match.empty() : p.lo > p.hi) q->always_false = true;
            if (p.kind == KIND_STR && (p.width > 8 || p.match.size() > (size_t)kMaxMatch) && !p.match.empty()) {
                std::string blob;
                for (auto &v : p.match) blob += v;
                void *d = nullptr;
                HIPCHK(pool_alloc(ctx, &d, blob.size()));
                p.d_blob = (uint8_t *)d;
                HIPCHK(hipMemcpyAsync(d, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
            }
        }
    }

    // (4) device buffers
    void *p = nullptr;
    const size_t words_alloc = (size_t)std::max<int64_t>(q->n_tiles * kTileWords, 1);
    HIPCHK(pool_alloc(ctx, &p, words_alloc * sizeof(uint64_t)));
    q->d_bitmap = (uint64_t *)p;
    HIPCHK(pool_alloc(ctx, &p, (size_t)std::max<int64_t>(q->n_tiles, 1) * sizeof(uint32_t)));
    q->d_tile_offsets = (uint32_t *)p;
    HIPCHK(pool_alloc(ctx, &p, (size_t)std::max<int64_t>(q->n_chunks, 1) * sizeof(uint32_t)));
    q->d_chunk_sums = (uint32_t *)p;
    HIPCHK(pool_alloc(ctx, &p, kMaxFilterGrid * sizeof(uint32_t)));
    q->d_block_partials = (uint32_t *)p;
    if (limit > 0 && limit <= kLimitGatherMaxRows && !table) { // (k_limit_gather's per-work-group counts: pooled memory, so cleared -- a run's tag is never zero)
        HIPCHK(pool_alloc(ctx, &p, 256 * sizeof(unsigned long long)));
        q->d_limit_state = (unsigned long long *)p;
        HIPCHK(hipMemsetAsync(q->d_limit_state, 0, 256 * sizeof(unsigned long long), ctx->stream));
    }
    HIPCHK(pool_alloc(ctx, &p, kFinishWords * sizeof(unsigned long long))); // {total, n_emit, status, limit, tally, log, log index, log capacity}, then the sub-tallies (imm3_device.h)
    q->d_total = (unsigned long long *)p;
    q->d_n_emit = q->d_total + 1;
    // Creation enqueues and returns: nothing below waits for the device (round 3 ended every creation with a stream
    // synchronisation, behind a 12.5 MB memset of the bitmap -- a query cost four times what running it did).  What the copies
    // read lives in the query handle.  The finish block = zeros + the limit, ONE copy; of the bitmap only the words behind
    // n_words in its last tile need to be zero (the offsets scan and the aggregation read whole tiles): every run writes all the
    // others before anything reads them.
    q->h_init.assign((size_t)kFinishWords, 0ULL);
    q->h_init[3] = (unsigned long long)limit;
    HIPCHK(hipMemcpyAsync(q->d_total, q->h_init.data(), (size_t)kFinishWords * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
    {
        const size_t first_pad = (size_t)std::min<int64_t>(std::max<int64_t>(q->n_words, 0), (int64_t)words_alloc);
        const size_t from = first_pad; // (words [n_words, words_alloc): at most one tile's worth)
        if (from < words_alloc) HIPCHK(hipMemsetAsync(q->d_bitmap + from, 0, (words_alloc - from) * sizeof(uint64_t), ctx->stream));
    }
    if (q->ragged) {
        std::vector<uint32_t> &base = q->h_word_row_base;
        std::vector<uint8_t> &nvalid = q->h_word_nvalid;
        base.assign((size_t)q->n_tiles * kTileWords, 0u);
        nvalid.assign((size_t)q->n_tiles * kTileWords, 0);
        int64_t row = 0;
        size_t w = 0;
        for (int32_t k = 0; k < nb; ++k) {
            const int64_t n = q->layout->size[(size_t)k];
            for (int64_t r = 0; r < n; r += 64) {
                base[w] = (uint32_t)(row + r);
                nvalid[w] = (uint8_t)std::min<int64_t>(64, n - r);
                ++w;
            }
            row += n;
        }
        HIPCHK(pool_alloc(ctx, &p, base.size() * sizeof(uint32_t) + 4));
        q->d_word_row_base = (uint32_t *)p;
        HIPCHK(pool_alloc(ctx, &p, nvalid.size() + 4));
        q->d_word_nvalid = (uint8_t *)p;
        if (!base.empty()) {
            HIPCHK(hipMemcpyAsync(q->d_word_row_base, base.data(), base.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(q->d_word_nvalid, nvalid.data(), nvalid.size(), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    if (n_proj > 0 && limit > 0) {
        const int rc = ensure_row_capacity(q.get(), (uint64_t)std::min<int64_t>(limit, std::max<int64_t>(q->n_rows, 1)));
        if (rc) return rc;
    }
    // Survivor records (k_filter_tile STAGE -> k_emit): an unlimited projection over one uniform segment whose select chain
    // is ONE tile launch (<= 3 predicate columns of int32 / int8 / 2-byte string, at most one string; none is also fine:
    // the record is then the position alone) and whose SELECT list is 1-, 2- and 4-byte columns, at most kMaxEmitGather of
    // them not predicate columns.
    // A table query (every segment the GPU owns as one scan unit) takes the ONE-LAUNCH plan under the same conditions when no
    // SELECT-list column has to be gathered (round 5; its TABLE instances walk the tile table); survivor records and streamed
    // gathers are one-segment plans, the bitmap path is a table's other plan.
    if (n_proj > 0 && limit <= 0 && !q->ragged && !q->always_false && nb >= 1 && q->n_rows > 0 && ctx->filter_variant != 1 &&
        ctx->filter_variant != 3 && q->preds.size() <= (size_t)kMaxTileCols && n_proj <= kMaxProj) {
        std::vector<const FoldedPred *> order;
        int n_s2 = 0;
        bool ok = true;
        for (const auto &fp : q->preds) {
            const int tk = fp.pfor ? (int)TK_NONE : tile_kind(fp);
            if (tk == TK_NONE) ok = false;
            n_s2 += tk == TK_S2;
            order.push_back(&fp);
        }
        ok = ok && n_s2 <= 1;
        std::stable_sort(order.begin(), order.end(), [](const FoldedPred *x, const FoldedPred *y) { return tile_kind(*x) < tile_kind(*y); });
        int n_gather = 0;
        int n_pred_proj = 0;       // predicate columns in the SELECT list (first mentions: their values ride in the records) ...
        bool pred_proj_wide = false; // ... and whether one of them is wider than a byte
        std::vector<int32_t> seen; // predicate columns already mentioned in the SELECT list: a second mention is gathered
        for (int32_t pj : q->proj) {
            const int32_t sci = q->used[(size_t)pj];
            const int32_t w = seg->cols[(size_t)sci].width;
            if (w != 1 && w != 2 && w != 4) ok = false;
            bool is_pred = false;
            for (const FoldedPred *fp : order) is_pred |= fp->seg_col == sci;
            if (is_pred && std::find(seen.begin(), seen.end(), sci) != seen.end()) is_pred = false;
            if (is_pred) {
                seen.push_back(sci);
                ++n_pred_proj;
                pred_proj_wide |= w > 1;
            }
            n_gather += !is_pred;
        }
        if (ok && n_gather <= kMaxEmitGather && (!table || (n_gather == 0 && ctx->filter_variant != 6))) {
            for (size_t k = 0; k < order.size(); ++k) {
                q->stage_kinds[k] = tile_kind(*order[k]);
                q->stage_seg_col[k] = order[k]->seg_col;
            }
            {   // what the cost model needs to know (imm3_plan.h)
                PlanShape &ps = q->plan_shape;
                ps = PlanShape();
                ps.n_rows = q->n_rows;
                for (const FoldedPred *fp : order) {
                    if (ps.n_pred >= kPlanMaxCols) break;
                    ps.pred_width[ps.n_pred] = fp->width;
                    ps.pred_match[ps.n_pred] = tile_kind(*fp) == TK_S2 ? (int32_t)fp->match.size() : 0;
                    ++ps.n_pred;
                }
                std::vector<int32_t> first; // predicate columns already mentioned (a second mention is gathered)
                for (int32_t pj : q->proj) {
                    if (ps.n_proj >= kPlanMaxCols) break;
                    const int32_t sci = q->used[(size_t)pj];
                    bool is_pred = false;
                    for (const FoldedPred *fp : order) is_pred |= fp->seg_col == sci;
                    if (is_pred && std::find(first.begin(), first.end(), sci) != first.end()) is_pred = false;
                    if (is_pred) first.push_back(sci);
                    ps.proj_width[ps.n_proj] = seg->cols[(size_t)sci].width;
                    ps.proj_is_pred[ps.n_proj] = is_pred;
                    ++ps.n_proj;
                }
                ps.rec_bytes = 4 * rec_layout(q->stage_kinds, -1).dwords;
                q->plan_pinned = ctx->filter_variant == 12;
            }
            // Single pass (k_filter_project): the filter kernel writes the rows itself.  Tuning variant 6 keeps the
            // three-launch form (records -> k_scan -> k_emit) for A/B runs.
            // Only when every SELECT-list column is a predicate column (its values ride in the records): gathers issued by the
            // four writer waves of a CU are latency-bound (C4 154 us against 118 us with the emit kernel's 2048 work-groups).
            if (ctx->filter_variant != 6 && (n_gather == 0 || ctx->filter_variant == 8)) {
                const int rc = single_pass_setup(q.get());
                if (rc) return rc;
            }
            if (!q->single_pass && n_gather > 0 && ctx->filter_variant != 6 && !table) {
                // The alternative the first count may switch to (single_pass_stream_columns): every gathered column of the SELECT list
                // (first mentions; dense int32 / int8) as a tile column that lets every value pass.
                std::vector<FoldedPred> pass;
                bool alt = true;
                for (int32_t pj : q->proj) {
                    const int32_t sci = q->used[(size_t)pj];
                    bool have = false;
                    for (const FoldedPred *fp : order) have |= fp->seg_col == sci;
                    for (const FoldedPred &fp : pass) have |= fp.seg_col == sci;
                    if (have) continue;
                    const SegCol &sc = seg->cols[(size_t)sci];
                    FoldedPred fp;
                    fp.seg_col = sci;
                    fp.width = sc.width;
                    if (sc.codec == IMM3_DENSE_INT && sc.width == 4) { fp.kind = KIND_I32; fp.lo = INT32_MIN; fp.hi = INT32_MAX; }
                    else if (sc.codec == IMM3_DENSE_TINYINT && sc.width == 1) { fp.kind = KIND_I8; fp.lo = -128; fp.hi = 127; }
                    else { alt = false; break; }
                    if (!col_flat(sc)) { alt = false; break; }
                    pass.push_back(fp);
                }
                if (alt && !pass.empty() && order.size() + pass.size() <= (size_t)kMaxTileCols) {
                    q->sp_pass = pass;
                    std::vector<const FoldedPred *> all(order);
                    for (const FoldedPred &fp : q->sp_pass) all.push_back(&fp);
                    std::stable_sort(all.begin(), all.end(), [](const FoldedPred *x, const FoldedPred *y) { return tile_kind(*x) < tile_kind(*y); });
                    bool any4 = false;
                    for (int k = 0; k < kMaxTileCols; ++k) {
                        q->alt_kinds[k] = (size_t)k < all.size() ? tile_kind(*all[(size_t)k]) : (int)TK_NONE;
                        q->alt_seg_col[k] = (size_t)k < all.size() ? all[(size_t)k]->seg_col : -1;
                    }
                    for (const FoldedPred &fp : q->sp_pass) any4 |= fp.width == 4;
                    bool any_s2 = false;
                    for (int k = 0; k < kMaxTileCols; ++k) any_s2 |= q->alt_kinds[k] == TK_S2;
                    // Whether they ARE streamed is the cost model's call once the survivors are known (single_pass_stream_columns: the
                    // sample at creation, a reservation or the first count).  Measured at 100 M rows (one launch / three launches):
                    // age < 10 -> id, 10 %: 117 / 123 us; 3 %: 107 / 79; 30 %: 204 / 168; 99 %: 534 / 372 -- a window around 10 %.
                    // Not with a string predicate (the 2-byte match streams at 74 us with the one-launch kernel's 8 streaming waves per
                    // CU against 47), and not for 1-byte columns alone (their gather reads every line of the column from ~3 % on and
                    // still costs 33 us at 10 %).
                    q->alt_ok = any4 && !any_s2;
                }
            }
            // Survivor records pay when the predicate columns' values are wanted: the staging instances of the filter kernel cost
            // 12 (string) to 33 us (int8) per 100 M rows more than the plain ones, and buy the emit kernel the projected predicate
            // columns.  When none is projected they buy nothing -- state in (5 values) -> age, 10 %: 120 us with records, 87 without
            // (filter -> offsets scan -> gather from the bitmap); age in (18, 30) -> id, 11 %: 167 / 122; 3 %: 106 / 79.
            q->records_narrow_only = n_pred_proj > 0 && !pred_proj_wide;
            if (q->single_pass || table || (n_gather > 0 && n_pred_proj == 0 && ctx->filter_variant != 11)) { /* no survivor records in HBM */ } else {
                const int rc = records_setup(q.get());
                if (rc) return rc;
            }
        }
    }
    {
        const int src = single_pass_sample(q.get()); // (the one synchronisation a creation may contain: segments of 4 M rows and more, undecided plans)
        if (src) return src;
        // No sample (a segment below 4 M rows -- there the sample costs what it saves): the plans are compared for one survivor in
        // ten, spread evenly; the first count corrects it.  (At 4 M rows the bitmap path wins nearly every shape: the one launch
        // starts at ~27 us, three small launches at 16-20.)
        imm3_query *qq = q.get();
        if (!qq->plan_have_density && !qq->plan_pinned && !qq->sp_P_fixed && ctx->filter_variant != 10 && qq->plan_shape.n_rows > 0 &&
            (qq->single_pass || qq->alt_ok || qq->d_stage_rec)) {
            const uint64_t guess = (uint64_t)(qq->n_rows / 10);
            const int rc2 = single_pass_stream_columns(qq, guess);
            if (rc2) return rc2;
            records_drop_if_narrow(qq, guess);
            single_pass_drop_if_narrow(qq, guess);
        }
    }
    *out = q.release();
    return IMM3_OK;
}

extern "C" int imm3_query_create(imm3_ctx *ctx, const imm3_segment *seg,
                                 const int32_t *used_cols, int32_t n_used,
                                 const imm3_select *sels, int32_t n_sels,
                                 const int32_t *proj, int32_t n_proj, int64_t limit,
                                 int32_t table_block_size, imm3_query **out) {
    return query_create_impl(ctx, seg, nullptr, used_cols, n_used, sels, n_sels, proj, n_proj, limit, table_block_size, out);
}

extern "C" int imm3_query_create_table(imm3_ctx *ctx, const imm3_table *table,
                                       const int32_t *used_cols, int32_t n_used,
                                       const imm3_select *sels, int32_t n_sels,
                                       const int32_t *proj, int32_t n_proj, int64_t limit,
                                       int32_t table_block_size, imm3_query **out) {
    if (!table || table->segs.empty()) return fail(IMM3_ERR_ARG, "table is null or empty");
    return query_create_impl(ctx, table->segs[0], table, used_cols, n_used, sels, n_sels, proj, n_proj, limit, table_block_size, out);
}

// ---------------------------------------------------------------------------------------------
// table: the tile table over all segments
// ---------------------------------------------------------------------------------------------
static void table_free(imm3_table *t) {
    if (!t) return;
    if (t->ctx) (void)hipSetDevice(t->ctx->device);
    (void)hipFree(t->d_tile_rows);
    for (auto p : t->d_tile_ptrs) (void)hipFree(p);
    (void)hipFree(t->d_sample_rows);
    for (auto p : t->d_sample_ptrs) (void)hipFree(p);
    for (auto sg : t->segs) segment_release(sg);
    if (t->ctx) ctx_release(t->ctx);
    delete t;
}
static void table_release(const imm3_table *ct) {
    imm3_table *t = const_cast<imm3_table *>(ct);
    if (t->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) table_free(t);
}

extern "C" int imm3_table_create(imm3_ctx *ctx, const imm3_segment *const *segs, int32_t n_segs, imm3_table **out) {
    if (!out) return fail(IMM3_ERR_ARG, "null argument");
    *out = nullptr;
    CTX_LIVE(ctx);
    if (n_segs <= 0 || !segs) return fail(IMM3_ERR_ARG, "a table needs at least one segment");
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<imm3_table, void (*)(imm3_table *)> t(new imm3_table(), table_free);
    t->ctx = ctx;
    ctx_retain(ctx);
    const size_t ncols = segs[0]->cols.size();
    std::vector<int32_t> all_cols(ncols);
    for (size_t c = 0; c < ncols; ++c) all_cols[c] = (int32_t)c;
    t->tile_start.push_back(0);
    for (int32_t si = 0; si < n_segs; ++si) {
        const imm3_segment *sg = segs[si];
        if (!sg || sg->ctx->device != ctx->device) return fail(IMM3_ERR_ARG, "segment is null or lives on another device");
        if (sg->closed) return fail(IMM3_ERR_STATE, "segment " + std::to_string(si) + " has been destroyed");
        if (sg->cols.size() != ncols) return fail(IMM3_ERR_ARG, "segments of one table must have the same columns");
        for (size_t c = 0; c < ncols; ++c)
            if (sg->cols[c].codec != segs[0]->cols[c].codec || sg->cols[c].width != segs[0]->cols[c].width)
                return fail(IMM3_ERR_ARG, "segments of one table must have the same column types");
        std::shared_ptr<const SegLayout> Lp;
        const int rc = segment_layout(sg, all_cols, Lp);
        if (rc) return rc;
        const SegLayout &L = *Lp;
        if (L.ragged) return fail(IMM3_ERR_LAYOUT, "segment " + std::to_string(si) + ": a non-final block is not a multiple of 64 rows (ragged layout); use per-segment queries");
        for (size_t c = 0; c < ncols; ++c) { // the tile table addresses flat columns: decode PFOR_INT ones now
            const int drc = ensure_dense(ctx, sg, (int32_t)c);
            if (drc) return drc;
        }
        t->segs.push_back(sg);
        segment_retain(sg); // the tile table points into the segment's columns
        t->seg_rows.push_back(L.rows);
        t->seg_first_batch.push_back((int32_t)t->batch_size.size());
        t->seg_first_word.push_back(t->tile_start.back() * kTileWords);
        for (size_t k = 0; k < L.size.size(); ++k) {
            t->batch_size.push_back(L.size[k]);
            t->batch_k.push_back((int32_t)k);
            t->batch_word_off.push_back(t->tile_start.back() * kTileWords + L.word_off[k]);
        }
        t->tile_start.push_back(t->tile_start.back() + (L.rows + kTileRows - 1) / kTileRows);
        t->n_rows += L.rows;
    }
    t->n_tiles = t->tile_start.back();
    t->seg_first_batch.push_back((int32_t)t->batch_size.size());
    t->seg_first_word.push_back(t->n_tiles * kTileWords);
    if (t->n_tiles * (int64_t)kTileRows > 0xFFFFFFFFLL) return fail(IMM3_ERR_LAYOUT, "table too large for 32-bit virtual row ids on one device");
    std::vector<uint32_t> rows((size_t)std::max<int64_t>(t->n_tiles, 1), 0);
    std::vector<std::vector<const void *>> ptrs(ncols, std::vector<const void *>((size_t)std::max<int64_t>(t->n_tiles, 1), nullptr));
    for (size_t si = 0; si < t->segs.size(); ++si) {
        const int64_t nt = t->tile_start[si + 1] - t->tile_start[si];
        for (int64_t k = 0; k < nt; ++k) {
            const size_t tile = (size_t)(t->tile_start[si] + k);
            rows[tile] = (uint32_t)std::min<int64_t>(kTileRows, t->seg_rows[si] - k * kTileRows);
            for (size_t c = 0; c < ncols; ++c) ptrs[c][tile] = col_flat(t->segs[si]->cols[c]) + (size_t)k * kTileRows * (size_t)t->segs[si]->cols[c].width;
        }
    }
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, rows.size() * sizeof(uint32_t)));
    t->d_tile_rows = (uint32_t *)p;
    HIPCHK(hipMemcpyAsync(t->d_tile_rows, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    t->d_tile_ptrs.assign(ncols, nullptr);
    for (size_t c = 0; c < ncols; ++c) {
        HIPCHK(hipMalloc(&p, ptrs[c].size() * sizeof(void *)));
        t->d_tile_ptrs[c] = (void **)p;
        HIPCHK(hipMemcpyAsync(t->d_tile_ptrs[c], ptrs[c].data(), ptrs[c].size() * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    }
    // the sample a query's plan is made on (single_pass_sample): eight chunks of 64 tiles spread evenly over the table
    std::vector<uint32_t> srows;
    std::vector<std::vector<const void *>> sptrs;
    if (t->n_tiles >= 4096) {
        srows.resize((size_t)kSampleTiles);
        sptrs.assign(ncols, std::vector<const void *>((size_t)kSampleTiles, nullptr));
        for (int i = 0; i < kSampleChunks; ++i) {
            int64_t tile0 = (int64_t)((2 * i + 1) * t->n_tiles / (2 * kSampleChunks)) - kSampleChunkTiles / 2;
            tile0 = std::max<int64_t>(0, std::min<int64_t>(tile0, t->n_tiles - kSampleChunkTiles));
            for (int64_t k = 0; k < kSampleChunkTiles; ++k) {
                srows[(size_t)(i * kSampleChunkTiles + k)] = rows[(size_t)(tile0 + k)];
                for (size_t c = 0; c < ncols; ++c) sptrs[c][(size_t)(i * kSampleChunkTiles + k)] = ptrs[c][(size_t)(tile0 + k)];
            }
        }
        HIPCHK(hipMalloc(&p, srows.size() * sizeof(uint32_t)));
        t->d_sample_rows = (uint32_t *)p;
        HIPCHK(hipMemcpyAsync(t->d_sample_rows, srows.data(), srows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        t->d_sample_ptrs.assign(ncols, nullptr);
        for (size_t c = 0; c < ncols; ++c) {
            HIPCHK(hipMalloc(&p, sptrs[c].size() * sizeof(void *)));
            t->d_sample_ptrs[c] = (void **)p;
            HIPCHK(hipMemcpyAsync(t->d_sample_ptrs[c], sptrs[c].data(), sptrs[c].size() * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out = t.release();
    return IMM3_OK;
}

extern "C" int imm3_table_destroy(imm3_table *t) {
    if (!t) return IMM3_OK;
    if (t->closed) return fail(IMM3_ERR_STATE, "table destroyed twice");
    imm3_ctx *ctx = t->ctx;
    ctx_retain(ctx);
    struct Unref { imm3_ctx *c; ~Unref() { ctx_release(c); } } unref{ctx};
    imm3::GateScope gate(&ctx->gate);
    if (!ctx->closed && ctx->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context");
    t->closed = true;
    if (!t->ctx->closed) {
        (void)hipSetDevice(t->ctx->device);
        (void)hipStreamSynchronize(t->ctx->stream);
    }
    table_release(t); // queries built on it keep the tile table (and its segments) alive until they are destroyed
    return IMM3_OK;
}

extern "C" int imm3_query_segment_starts(const imm3_query *q, int32_t *n_segments, int32_t *first_batch, int64_t *first_word) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    if (!q->table) { // one segment
        if (n_segments) *n_segments = 1;
        if (first_batch) { first_batch[0] = 0; first_batch[1] = (int32_t)q->layout->size.size(); }
        if (first_word) { first_word[0] = 0; first_word[1] = q->n_words; }
        return IMM3_OK;
    }
    const size_t n = q->table->segs.size();
    if (n_segments) *n_segments = (int32_t)n;
    if (first_batch) std::memcpy(first_batch, q->table->seg_first_batch.data(), (n + 1) * sizeof(int32_t));
    if (first_word) std::memcpy(first_word, q->table->seg_first_word.data(), (n + 1) * sizeof(int64_t));
    return IMM3_OK;
}

extern "C" int imm3_query_locate_rows(const imm3_query *q, const uint32_t *row_index, uint64_t n, uint32_t *segment_out, uint32_t *row_out) {
    if (!q || (n && !row_index)) return fail(IMM3_ERR_ARG, "null argument");
    for (uint64_t i = 0; i < n; ++i) {
        if (!q->table) {
            if (segment_out) segment_out[i] = 0;
            if (row_out) row_out[i] = row_index[i];
            continue;
        }
        const int64_t tile = row_index[i] >> 10;
        const auto &ts = q->table->tile_start;
        const size_t si = (size_t)(std::upper_bound(ts.begin(), ts.end(), tile) - ts.begin()) - 1;
        if (segment_out) segment_out[i] = (uint32_t)si;
        if (row_out) row_out[i] = (uint32_t)((tile - ts[si]) * kTileRows + (row_index[i] & (kTileRows - 1)));
    }
    return IMM3_OK;
}

extern "C" int imm3_query_destroy(imm3_query *q) {
    if (!q) return IMM3_OK;
    imm3_ctx *ctx = q->ctx;
    if (!ctx) { query_free(q); return IMM3_OK; }
    ctx_retain(ctx); // the gate lives in the context: keep it past query_free
    int rc = IMM3_OK;
    {
        imm3::GateScope gate(&ctx->gate);
        if (!ctx->closed && ctx->capture) rc = fail(IMM3_ERR_STATE, "a graph capture is open on this context");
        else {
            if (!ctx->closed) graphs_mark_stale(ctx, q); // a graph that recorded this query's runs points into its buffers
            query_free(q);
        }
    }
    ctx_release(ctx);
    return rc;
}

extern "C" int imm3_query_reserve_rows(imm3_query *q, uint64_t rows) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE(q->ctx);
    HIPCHK(hipSetDevice(q->ctx->device));
    const int rc = ensure_row_capacity(q, rows);
    if (rc) return rc;
    q->reserved = true;
    if (rows < (uint64_t)q->n_rows) { // (a reservation skips the first run's look at the count: it is the estimate)
        const int src = single_pass_stream_columns(q, rows);
        if (src) return src;
    }
    if (rows < (uint64_t)q->n_rows) single_pass_adapt(q, rows, -1); // (the reservation bounds the survivors)
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// execution
// ---------------------------------------------------------------------------------------------
static void fill_colpred(const imm3_query *q, const FoldedPred &fp, ColPred &cp) {
    std::memset(&cp, 0, sizeof(cp));
    const SegCol &sc = q->seg->cols[(size_t)fp.seg_col];
    cp.data = col_flat(sc);
    cp.kind = fp.kind;
    cp.width = fp.width;
    cp.lo = (int32_t)fp.lo;
    cp.hi = (int32_t)fp.hi;
    cp.n_match = (int32_t)fp.match.size();
    cp.match_in_args = (fp.kind == KIND_STR && !fp.d_blob) ? 1 : 0;
    cp.match_blob = fp.d_blob;
    if (cp.match_in_args) {
        for (size_t m = 0; m < fp.match.size(); ++m) {
            uint64_t v = 0;
            for (int b = 0; b < fp.width; ++b) v |= (uint64_t)(uint8_t)fp.match[m][(size_t)b] << (8 * b);
            cp.match[m] = v;
        }
    }
}

// join: make `s` wait for this query's count reduce on the aux stream (no-op when it ran on the main stream)
static int join_total(imm3_query *q, hipStream_t s) {
    if (q->total_on_aux) HIPCHK(hipStreamWaitEvent(s, q->ev_total_done, 0));
    return IMM3_OK;
}
static int settle_whole_select(imm3_query *q);
// (imm3_comm_allreduce_count: the word that goes into the collective is the segment's count -- a run that stopped at its limit is
// followed by the whole select here, enqueued, no host wait)
static int settle_agg_select(imm3_query *q);
int imm3::join_query_count(imm3_query *q, hipStream_t s) {
    int rc = settle_agg_select(q);
    if (!rc) rc = settle_whole_select(q);
    return rc ? rc : join_total(q, s);
}


// count_in_scan: a projection follows on the same stream; its offsets scan publishes the count (no k_total launch)
// count_only: the caller wants selected.size alone -- a chain that is ONE tile launch then stores no bitmap
// Chunks of a limit scan: they end at tiles 1024, 8192, 32 768, 131 072, ... (x 4) and at the segment's end -- four launches for
// 100 M rows.  A `limit 10` is usually met in the first megarow: the chunks behind it cost their dispatch only (1 - 4 us each, which
// is why there are few of them); a limit met at 5 % of the segment stops at 8 %; one met in the second half scans everything, as a
// whole select would.  Multiples of kChunkTiles (the offsets scan's unit).
static constexpr int64_t kLimitSecondEndTiles = 8192; // (kLimitFirstChunkTiles: imm3_api_internal.h)

// whole: never in chunks (the getters' full select)
static int run_select(imm3_query *q, bool overlap_total, bool count_in_scan = false, bool count_only = false, bool whole = false) {
    imm3_ctx *ctx = q->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    {   // the previous run's reduce may still be reading block_partials / writing total on the aux stream
        const int jrc = join_total(q, s);
        if (jrc) return jrc;
        q->total_on_aux = false;
    }
    q->offsets_valid = false; // (a new bitmap)
    q->select_partial = false;
    if (q->always_false || q->n_tiles == 0) {
        // an empty interval / empty IN-list clears every bit; nothing to read
        HIPCHK(hipMemsetAsync(q->d_total, 0, 2 * sizeof(unsigned long long), s)); // (an always-false query logs nothing)
        HIPCHK(hipMemsetAsync(q->d_bitmap, 0, (size_t)std::max<int64_t>(q->n_tiles * kTileWords, 1) * sizeof(uint64_t), s));
        q->ran_select = true;
        q->bitmap_valid = true;
        q->ran_single_pass = false;
        return IMM3_OK;
    }
    // Plan the passes.  Uniform layouts: numeric and 2-byte-string predicates go through the tile kernel, up
    // to 3 columns (at most one string) per launch; everything else -- other string widths, long IN-lists,
    // ragged layouts -- through the word-at-a-time kernel, up to 4 columns per launch.  Every pass after the
    // first ANDs into the bitmap in memory.
    std::vector<const FoldedPred *> tile_preds, generic_preds, pfor_preds;
    for (const auto &p : q->preds) {
        if (p.pfor) pfor_preds.push_back(&p);
        else if (!q->ragged && ctx->filter_variant != 1 && tile_kind(p) != TK_NONE) tile_preds.push_back(&p);
        else generic_preds.push_back(&p);
    }
    if (q->table && !generic_preds.empty()) return fail(IMM3_ERR_ARG, "table queries support int32 / int8 / 2-byte string predicates (<= 8 IN-list values); use per-segment queries");
    std::stable_sort(tile_preds.begin(), tile_preds.end(),
                     [](const FoldedPred *x, const FoldedPred *y) { return tile_kind(*x) < tile_kind(*y); });
    // Plan the tile passes first: up to 3 columns per launch, numeric kinds first (sorted), at most one 2-byte string
    // column per launch -- so two string predicates are two passes even when only two columns are filtered.  A query
    // without predicates is one tile pass with zero columns.
    std::vector<std::vector<const FoldedPred *>> tile_passes;
    while (!tile_preds.empty()) {
        std::vector<const FoldedPred *> take;
        int n_s2 = 0;
        for (size_t i = 0; i < tile_preds.size() && take.size() < (size_t)kMaxTileCols; ++i) {
            const int tk = tile_kind(*tile_preds[i]);
            if (tk == TK_S2 && n_s2 == 1) continue;
            take.push_back(tile_preds[i]);
            n_s2 += tk == TK_S2;
        }
        for (const FoldedPred *fp : take) tile_preds.erase(std::find(tile_preds.begin(), tile_preds.end(), fp));
        tile_passes.push_back(take);
    }
    if (q->preds.empty() && !q->ragged && ctx->filter_variant != 1) tile_passes.emplace_back();
    int pass = 0;
    int grid = 1;
    bool count_done = false; // the filter kernel's last work-group has written total / n_emit
    q->stage_written = false;
    q->bitmap_lazy = false;
    q->ran_single_pass = false;
    // exactly ONE launch in the whole select chain: only then may that launch publish the count (and append to the count
    // log) itself, and only then are the survivors' values staged
    const bool single_tile_pass = generic_preds.empty() && pfor_preds.empty() && tile_passes.size() == 1;
    const bool skip_bitmap = count_only && single_tile_pass && !q->table && !overlap_total && ctx->filter_variant != 7;
    q->bitmap_valid = !skip_bitmap;
    // `limit` stops the scan (Project.scala:73-80; Engine.scala:166,253-258: the reference's workers stall on the full queue once the
    // consumer has its rows): a projection with a limit whose select chain is one tile launch over one uniform segment runs that
    // launch as chunks of growing size; every chunk first looks at the rows selected so far (a device word) and leaves at once when
    // the limit has been reached -- nothing is read, no bitmap line written.  Enqueued blindly: no host wait.  Tuning variant 14: off.
    LimitScanInputs li;
    li.whole = whole;
    li.count_log_on = q->count_log_on;
    li.count_in_scan = count_in_scan;
    li.limit = q->limit;
    li.single_tile_pass = single_tile_pass;
    li.table = q->table != nullptr;
    li.records = q->d_stage_rec != nullptr;
    li.skip_bitmap = skip_bitmap;
    li.overlap_total = overlap_total;
    li.filter_variant = ctx->filter_variant;
    li.n_tiles = q->n_tiles;
    const bool chunked = limit_scan_applies(li);
    for (const auto &take : tile_passes) {
        TileArgs a;
        std::memset(&a, 0, sizeof(a));
        const int n = (int)take.size();
        for (int k = 0; k < kMaxTileCols; ++k) a.kinds[k] = TK_NONE;
        for (int k = 0; k < n; ++k) {
            const FoldedPred &fp = *take[(size_t)k];
            if (q->table) a.tile_ptrs[k] = (const void *const *)q->table->d_tile_ptrs[(size_t)fp.seg_col];
            a.kinds[k] = tile_kind(fp);
            fill_tile_col(q, fp, a.cols[k], a.kinds[k]);
        }
        if (single_tile_pass && q->d_stage_rec && !skip_bitmap && !q->force_plain_select) { // the columns are in the order the records were laid out for (same sort)
            bool same = true;
            for (int k = 0; k < kMaxTileCols; ++k) same = same && a.kinds[k] == q->stage_kinds[k] && (k >= n || take[(size_t)k]->seg_col == q->stage_seg_col[k]);
            if (!same) return fail(IMM3_ERR_ARG, "internal: staged record layout does not match the tile launch");
            a.stage_rec = q->d_stage_rec;
            a.tile_start = q->d_tile_start;
            a.wave_cap = q->stage_wave_cap;
            a.max_slots = q->stage_max_slots;
            q->stage_written = true;
        }
        a.ablate = (ctx->filter_variant >= 20 && ctx->filter_variant <= 22) ? ctx->filter_variant.load() : 0; // (tools' build only)
        a.and_existing = pass > 0;
        a.n_rows = q->n_rows;
        a.n_words = q->n_words;
        a.n_tiles = q->n_tiles;
        a.bitmap = q->d_bitmap;
        // A records run whose offsets scan follows (imm3_query_run of a projection) stores NO bitmap: the records carry the positions
        // and the scan takes the tiles' counts from the arenas (round 5: 12.5 MB of 128-byte line stores in between the streaming
        // loads, and 12.5 MB read back by k_scan -- C4 107 -> 100 us).  imm3_query_bitmap materialises it on demand.  Tuning 19: off.
        q->bitmap_lazy = q->stage_written && count_in_scan && !q->count_log_on && ctx->filter_variant != 19;
        if (q->bitmap_lazy) {
            a.bitmap = nullptr;
            q->bitmap_valid = false;
        }
        a.block_partials = q->d_block_partials;
        a.tile_rows = q->table ? q->table->d_tile_rows : nullptr; // table query: address the columns through the tile table
        bool any_i32 = false;
        int narrow_bytes = 0;
        for (int k = 0; k < kMaxTileCols; ++k) {
            any_i32 |= (a.kinds[k] == TK_I32);
            narrow_bytes += a.kinds[k] == TK_I8 ? 1 : (a.kinds[k] == TK_S2 ? 2 : 0);
        }
        bool stamped = false;
        grid = filter_grid(q->n_tiles, false, any_i32, ctx->grid_blocks, narrow_bytes); // (no column at all: the store-only kernel also likes 1536 groups, 9.9 vs 17.2 us)
        if (q->stage_written) grid = q->stage_grid; // fixed at creation: the arena layout depends on it
        // A select chain that is ONE tile pass also reduces its count in the kernel (one relaxed atomic per work-group into a
        // two-level tally, finish_add): no k_total launch.  Variant 7 = never; variant 13 = only at <= 512 work-groups (what
        // round 1 did: with a single tally the 1536 atomics of a narrow-column launch cost more than the launch they saved).
        if (single_tile_pass && !overlap_total && ctx->filter_variant != 7 && (grid <= 512 || ctx->filter_variant != 13)) {
            a.finish = q->d_total;
            count_done = true;
        }
        // bitmap lines parked in LDS and stored in bursts: no staging (whose LDS and 2048 work-groups
        // leave no room for 32 KiB more per group); tuning variant 12 switches it off
        // (64 lines = 32 KiB per work-group at <= 4 groups per CU; 16 lines = 8 KiB for the 1536-group narrow-column kernels)
        a.defer_lines = (ctx->filter_variant == 12 || q->bitmap_lazy) ? 0 : (q->stage_written ? 16 : (grid <= 1024 ? kDeferLines : 16)); // (no bitmap, no lines to park)
        if (skip_bitmap) { // count-only: the kernel instance that stores nothing (the count is reduced in the kernel)
            a.bitmap = nullptr;
            a.defer_lines = 0;
        }
        if (ctx->d_stamps) { // (diagnostics: bench.py's instrumented pass)
            std::lock_guard<std::mutex> lk(ctx->mu);
            if (ctx->d_stamps && ctx->stamp_used < ctx->stamp_slots) {
                a.stamps = ctx->d_stamps + (size_t)ctx->stamp_used * kMaxFilterGrid * 2;
                ctx->stamp_grids.push_back(grid);
                ++ctx->stamp_used;
                stamped = true;
            }
        }
        (void)stamped;
        if (chunked) { // the limit scan: this launch in chunks that end at tiles 1024, 8192, 32 768, ...; each adds to the running count and scanned-tile words
            int widths[kMaxTileCols] = {0, 0, 0};
            for (int k = 0; k < kMaxTileCols; ++k) widths[k] = a.kinds[k] == TK_I32 ? 4 : (a.kinds[k] == TK_S2 ? 2 : (a.kinds[k] == TK_I8 ? 1 : 0));
            int64_t tile0 = 0, len = kLimitFirstChunkTiles;
            while (tile0 < q->n_tiles) {
                const int64_t tiles = std::min<int64_t>(len, q->n_tiles - tile0);
                TileArgs c = a;
                for (int k = 0; k < kMaxTileCols; ++k)
                    if (c.kinds[k] != TK_NONE) c.cols[k].data = (const uint8_t *)a.cols[k].data + tile0 * kTileRows * widths[k];
                c.bitmap = a.bitmap + tile0 * kTileWords;
                c.n_rows = std::min<int64_t>(tiles * kTileRows, q->n_rows - tile0 * kTileRows);
                c.n_words = (c.n_rows + 63) / 64;
                c.n_tiles = (c.n_words + kTileWords - 1) / kTileWords;
                c.finish = q->d_total;
                c.chunked = tile0 == 0 ? 2 : 1; // (the first chunk starts the running words over)
                c.stamps = nullptr;
                const int cgrid = filter_grid(c.n_tiles, false, any_i32, ctx->grid_blocks, narrow_bytes);
                c.defer_lines = ctx->filter_variant == 12 ? 0 : (cgrid <= 1024 ? kDeferLines : 16);
                LaunchTimer t(ctx, 0);
                if (!launch_filter_tile(c, cgrid, s, t.start, t.stop)) return fail(IMM3_ERR_ARG, "internal: no tile kernel for this column combination");
                HIPCHK(hipGetLastError());
                tile0 += tiles;
                len = tile0 == kLimitFirstChunkTiles ? kLimitSecondEndTiles - tile0 : tile0 * 3; // (the next chunk ends at 4 x this one's end)
            }
            count_done = true;
            q->select_partial = true;
            ++pass;
            continue;
        }
        LaunchTimer t(ctx, 0);
        if (!launch_filter_tile(a, grid, s, t.start, t.stop)) return fail(IMM3_ERR_ARG, "internal: no tile kernel for this column combination");
        HIPCHK(hipGetLastError());
        ++pass;
    }
    // PFOR_INT passes: one compressed column per launch, decoded in LDS and compared in registers
    q->has_pfor_pass = !pfor_preds.empty();
    for (const FoldedPred *fp : pfor_preds) {
        const SegCol &sc = q->seg->cols[(size_t)fp->seg_col];
        PforArgs a;
        std::memset(&a, 0, sizeof(a));
        a.data = sc.d_data;
        a.block_off = sc.d_block_off;
        a.n_blocks = (int64_t)sc.block_rows.size();
        a.lo = (int32_t)fp->lo;
        a.hi = (int32_t)fp->hi;
        a.and_existing = pass > 0;
        a.n_rows = q->n_rows;
        a.n_words = q->n_words;
        a.n_tiles = q->n_tiles;
        a.bitmap = q->d_bitmap;
        a.block_partials = q->d_block_partials;
        a.status = (uint32_t *)(q->d_total + 2);
        // VALU/LDS-latency bound, 5 waves per SIMD resident: the finest grid balances best (measured 83 us at 2048
        // work-groups, 76 us at 4096, 100 M rows)
        grid = filter_grid(q->n_tiles, true, false, ctx->grid_blocks > 0 ? ctx->grid_blocks.load() : kMaxFilterGrid);
        {
            LaunchTimer t(ctx, 0);
            launch_filter_pfor(a, grid, s, t.start, t.stop);
        }
        HIPCHK(hipGetLastError());
        ++pass;
    }
    // generic passes
    size_t gi = 0;
    const bool need_empty_generic = q->preds.empty() && pass == 0;
    while (gi < generic_preds.size() || (need_empty_generic && pass == 0)) {
        FilterArgs a;
        std::memset(&a, 0, sizeof(a));
        const size_t take = std::min<size_t>(kMaxPredCols, generic_preds.size() - gi);
        for (size_t i = 0; i < take; ++i) fill_colpred(q, *generic_preds[gi + i], a.cols[i]);
        a.ncols = (int32_t)take;
        a.and_existing = pass > 0;
        a.n_rows = q->n_rows;
        a.n_words = q->n_words;
        a.n_tiles = q->n_tiles;
        a.bitmap = q->d_bitmap;
        a.block_partials = q->d_block_partials;
        a.word_row_base = q->d_word_row_base;
        a.word_nvalid = q->d_word_nvalid;
        grid = filter_grid(q->n_words, true, false, ctx->grid_blocks);
        {
            LaunchTimer t(ctx, 0);
            launch_filter_generic(a, grid, s, t.start, t.stop);
        }
        HIPCHK(hipGetLastError());
        gi += take;
        ++pass;
    }
    q->count_pending_scan = !count_done && count_in_scan;
    if (!count_done && !count_in_scan) { // the last pass's per-workgroup partials -> selected-row count (+ rows ProjectOp will emit)
        TotalArgs ta;
        std::memset(&ta, 0, sizeof(ta));
        ta.block_partials = q->d_block_partials;
        ta.n_partials = grid;
        ta.total = q->d_total;
        ta.n_emit = q->d_n_emit;
        ta.limit = q->limit;
        hipStream_t ts = s;
        if (overlap_total) {
            // nothing downstream on the main stream needs the count: reduce it on the aux stream so the next scan
            // starts right behind this one (saves the reduce kernel and two dependent-launch gaps per step)
            {
                std::lock_guard<std::mutex> lk(ctx->mu);
                if (!ctx->aux) HIPCHK(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
            }
            if (!q->ev_filter_done) {
                HIPCHK(hipEventCreateWithFlags(&q->ev_filter_done, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&q->ev_total_done, hipEventDisableTiming));
            }
            HIPCHK(hipEventRecord(q->ev_filter_done, s));
            HIPCHK(hipStreamWaitEvent(ctx->aux, q->ev_filter_done, 0));
            ts = ctx->aux;
        }
        {
            LaunchTimer t(ctx, 3);
            launch_total(ta, ts, t.start, t.stop);
        }
        if (overlap_total) {
            HIPCHK(hipEventRecord(q->ev_total_done, ctx->aux));
            q->total_on_aux = true;
        }
    }
    HIPCHK(hipGetLastError());
    q->ran_select = true;
    return IMM3_OK;
}

// the SELECT-list columns as the unpacking kernels take them: gathered columns first, then the ones the record carries
static int fill_emit_cols(const imm3_query *q, EmitCol *out, int &n_out) {
    std::vector<EmitCol> gathered, staged;
    for (size_t j = 0; j < q->proj.size(); ++j) {
        const int32_t sci = q->used[(size_t)q->proj[j]];
        const SegCol &sc = q->seg->cols[(size_t)sci];
        EmitCol c;
        std::memset(&c, 0, sizeof(c));
        c.dst = q->d_proj[j];
        c.width = sc.width;
        c.rec_dword = -1;
        for (int k = 0; k < kMaxTileCols; ++k)
            if (q->stage_seg_col[k] == sci) {
                const RecField f = rec_layout(q->stage_kinds, k);
                c.rec_dword = f.dword;
                c.rec_shift = f.shift;
            }
        if (c.rec_dword < 0) { c.src = col_flat(sc); gathered.push_back(c); }
        else staged.push_back(c);
    }
    int n = 0;
    for (const auto &c : gathered) out[n++] = c;
    for (const auto &c : staged) out[n++] = c;
    n_out = n;
    return (int)gathered.size();
}

void imm3::fill_tile_col(const imm3_query *q, const FoldedPred &fp, TileCol &c, int kind) {
    c.data = col_flat(q->seg->cols[(size_t)fp.seg_col]);
    c.lo = (int32_t)fp.lo;
    c.hi = (int32_t)fp.hi;
    if (kind == TK_S2) {
        c.n_match = (int32_t)fp.match.size();
        for (size_t m = 0; m < fp.match.size(); ++m)
            c.match[m] = (uint32_t)(uint8_t)fp.match[m][0] | ((uint32_t)(uint8_t)fp.match[m][1] << 8);
    }
}

// Per-device state of the single-pass projection kernel, whose work-groups wait on each other and therefore must own the
// device while they run (imm3_project.hip): launches of all contexts of a device are chained through one event -- each
// starts behind the previous one, stream side, no host wait -- and a device word holds the running launch's ticket for
// whatever the host cannot order (graph replays, other processes on the same GPU).  Internal synchronisation state, guarded
// by its mutex; it lives as long as the process.
namespace {
struct SinglePassDevice {
    std::mutex mu;
    hipEvent_t last = nullptr;            // recorded behind the last launch
    unsigned long long *d_lock = nullptr; // the ticket word
};
SinglePassDevice g_single_pass[kMaxDevices];
} // namespace

static int single_pass_device(int device, SinglePassDevice **out) {
    if (device < 0 || device >= kMaxDevices) return fail(IMM3_ERR_ARG, "device index out of range");
    SinglePassDevice &d = g_single_pass[device];
    std::lock_guard<std::mutex> lk(d.mu);
    if (!d.d_lock) {
        void *p = nullptr;
        HIPCHK(hipMalloc(&p, 64));
        HIPCHK(hipMemset(p, 0, 64));
        d.d_lock = (unsigned long long *)p;
        HIPCHK(hipEventCreateWithFlags(&d.last, hipEventDisableTiming));
    }
    *out = &d;
    return IMM3_OK;
}

#ifdef IMM3_ABLATE
static int single_pass_lock_word(int device, unsigned long long **out) {
    SinglePassDevice *dev = nullptr;
    const int rc = single_pass_device(device, &dev);
    if (rc) return rc;
    *out = dev->d_lock;
    return IMM3_OK;
}
#endif

// ScanOp -> SelectOp* -> ProjectOp in one launch (k_filter_project): bitmap, count and the projected rows
static int run_single_pass(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    {
        const int jrc = join_total(q, s);
        if (jrc) return jrc;
        q->total_on_aux = false;
    }
    // The rows are written by the filter kernel itself, so their arrays exist before the count does: the caller's
    // reservation, else room for every row of the segment (pooled: allocated once).  A reservation that turns out too
    // small is answered from the bitmap when the rows are fetched (settle_rows).
    if (!q->reserved && q->cap_rows < (uint64_t)q->n_rows) {
        const int rc = ensure_row_capacity(q, (uint64_t)q->n_rows);
        if (rc) return rc;
    }
    ProjectArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int k = 0; k < kMaxTileCols; ++k) {
        a.kinds[k] = q->stage_kinds[k];
        if (a.kinds[k] == TK_NONE) continue;
        const FoldedPred *fp = nullptr;
        for (const auto &p : q->preds)
            if (p.seg_col == q->stage_seg_col[k]) fp = &p;
        for (const auto &p : q->sp_pass) // (a streamed SELECT-list column: every value passes)
            if (p.seg_col == q->stage_seg_col[k]) fp = &p;
        if (!fp) return fail(IMM3_ERR_ARG, "internal: single-pass plan lost a predicate column");
        fill_tile_col(q, *fp, a.cols[k], a.kinds[k]);
    }
    {   // a communicator may have been attached or destroyed since the plan was made: the grid follows, and so does the planned P
        // (a P the host has lowered for dense survivors stays: it is below either plan)
        const int32_t want_plan = q->sp_P_plan_for[single_pass_reserves(q) ? 1 : 0];
        if (!q->sp_P_fixed && !ctx->capture && want_plan > 0 && want_plan != q->sp_P_plan) {
            const bool at_plan = q->sp_P == q->sp_P_plan;
            q->sp_P_plan = want_plan;
            if (at_plan || q->sp_P > want_plan) {
                if (hipMemsetAsync(q->d_desc, 0, q->sp_trash_off, s) == hipSuccess) single_pass_set_P(q, want_plan); // (hygiene, as in single_pass_pick_P)
                else (void)hipGetLastError();
            }
        }
        q->sp_grid = single_pass_run_grid(q);
    }
    a.P = q->sp_P;
    a.n_rows = q->n_rows;
    if (q->table) {
        if (!q->d_tile_desc) return fail(IMM3_ERR_STATE, "internal: table query planned as one launch without its tile descriptors");
        a.tile_desc = q->d_tile_desc;
        a.n_rows = q->n_tiles * kTileRows; // (virtual rows: what the tiles span; the kernel takes a tile's valid rows from its descriptor)
    }
    a.n_tiles = q->n_tiles;
    a.n_spans = q->sp_spans;
    a.n_rounds = (q->sp_spans + q->sp_grid - 1) / q->sp_grid;
    a.bitmap = q->d_bitmap;
    a.finish = q->d_total;
    a.desc = (unsigned long long *)((uint8_t *)q->d_desc + q->sp_desc_off);
    a.round_total = q->d_desc;
    a.round_ctr = (uint32_t *)(q->d_desc + q->sp_rounds_max);
    a.trash = (uint8_t *)q->d_desc + q->sp_trash_off;
    a.cap_rows = q->cap_rows;
    a.row_index = q->d_row_index;
    {   // SELECT-list columns: the first mention of a predicate column comes out of the records, everything else is gathered
        int ng = 0;
        for (size_t j = 0; j < q->proj.size(); ++j) {
            const int32_t sci = q->used[(size_t)q->proj[j]];
            const SegCol &sc = q->seg->cols[(size_t)sci];
            int k_pred = -1;
            for (int k = 0; k < kMaxTileCols; ++k)
                if (q->stage_seg_col[k] == sci && !a.pred_dst[k]) { k_pred = k; break; }
            if (k_pred >= 0) a.pred_dst[k_pred] = q->d_proj[j];
            else {
                if (ng >= kMaxEmitGather || q->table) return fail(IMM3_ERR_ARG, "internal: single-pass plan has too many gathered columns");
                a.gather[ng].dst = q->d_proj[j];
                a.gather[ng].src = col_flat(sc);
                a.gather[ng].width = sc.width;
                ++ng;
            }
        }
        a.n_gather = ng;
    }
    const int fv = ctx->filter_variant;
    a.ablate = (fv >= 50 && fv <= 50 + 255) ? fv - 50 : 0; // (tools' build only: a mask -- 1 no unpack, 2 no chained scan, 4 no records, 16 no output stores, 32 plain instead of non-temporal stores in the straight copy of fully surviving dense ranges)
    a.max_polls = ctx->fault_max_polls; // (tools' build only: imm3_ctx_inject_fault)
    a.fault_wg = ctx->fault_wg;
    a.fault_span = ctx->fault_span;
    if (ctx->d_stamps) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (ctx->d_stamps && ctx->stamp_used < ctx->stamp_slots) {
            a.stamps = ctx->d_stamps + (size_t)ctx->stamp_used * kMaxFilterGrid * 2;
            ctx->stamp_grids.push_back(q->sp_grid);
            ++ctx->stamp_used;
        }
    }
    SinglePassDevice *dev = nullptr;
    {
        const int drc = single_pass_device(ctx->device, &dev);
        if (drc) return drc;
    }
    a.device_lock = dev->d_lock;
    {
        std::lock_guard<std::mutex> lk(dev->mu); // (wait - launch - record is one step: the next launcher waits for THIS launch)
        const bool chained = !ctx->capture;      // (a capture cannot depend on an event recorded outside it: the device lock covers replays)
        if (chained) HIPCHK(hipStreamWaitEvent(s, dev->last, 0));
        {
            LaunchTimer t(ctx, 0);
            const bool launched = q->table ? launch_filter_project_table(a, q->sp_grid, s, t.start, t.stop) : launch_filter_project(a, q->sp_grid, s, t.start, t.stop);
            if (!launched) return fail(IMM3_ERR_ARG, "internal: no single-pass kernel for this column combination");
        }
        HIPCHK(hipGetLastError());
        if (chained) HIPCHK(hipEventRecord(dev->last, s));
    }
    q->stage_written = false;
    q->count_pending_scan = false;
    q->has_pfor_pass = false;
    q->ran_select = true;
    q->bitmap_valid = true;
    q->ran_project = true;
    q->ran_single_pass = true;
    q->sp_verified = false;
    q->offsets_valid = false;
    return IMM3_OK;
}

// ProjectOp from the survivor records the select launch staged
static int launch_emit_records(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    EmitArgs e;
    std::memset(&e, 0, sizeof(e));
    e.stage = q->d_stage_rec;
    e.tile_start = q->d_tile_start;
    e.wave_cap = q->stage_wave_cap;
    e.n_waves = (int64_t)q->stage_grid * kWavesPerBlock;
    e.main_tiles = q->stage_main_tiles;
    e.max_slots = q->stage_max_slots;
    e.T = q->stage_T;
    e.ablate = (ctx->filter_variant >= 34 && ctx->filter_variant <= 35) ? ctx->filter_variant.load() : 0; // (tools' build only)
    e.tile_offsets = q->d_tile_offsets;
    e.chunk_sums = q->d_chunk_sums;
    e.n_tiles = q->n_tiles;
    e.cap_rows = q->cap_rows;
    e.row_index = q->d_row_index;
    e.R = rec_layout(q->stage_kinds, -1).dwords;
    int n_cols = 0;
    const int n_gather = fill_emit_cols(q, e.cols, n_cols);
    e.n_cols = n_cols;
    LaunchTimer t(ctx, 2);
    launch_emit(e, n_gather, 0, ctx->stream, t.start, t.stop);
    HIPCHK(hipGetLastError());
    return IMM3_OK;
}

static int launch_project(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    hipStream_t s = ctx->stream;
    if (q->stage_written) return launch_emit_records(q);
    GatherArgs g;
    std::memset(&g, 0, sizeof(g));
    g.bitmap = q->d_bitmap;
    g.tile_offsets = q->d_tile_offsets;
    g.chunk_sums = q->d_chunk_sums;
    g.n_tiles = q->n_tiles;
    g.n_words = q->n_words;
    g.limit = q->limit;
    g.cap_rows = q->cap_rows;
    g.n_staged_tiles = 0;
    g.word_row_base = q->d_word_row_base;
    g.tile_rows = q->table ? q->table->d_tile_rows : nullptr;
    g.scanned_tiles = q->select_partial ? q->d_total + kFinishLimitTiles : nullptr;
    // more SELECT-list columns than one launch carries: gather in groups (row indices written by the first)
    size_t done = 0;
    const size_t np = q->proj.size();
    do {
        const size_t take = std::min<size_t>(kMaxProj, np - done);
        g.row_index = done == 0 ? q->d_row_index : nullptr;
        g.n_proj = (int32_t)take;
        for (size_t j = 0; j < take; ++j) {
            const SegCol &sc = q->seg->cols[(size_t)q->used[(size_t)q->proj[done + j]]];
            g.proj[j].src = col_flat(sc);
            g.proj[j].tile_ptrs = q->table ? (const void *const *)q->table->d_tile_ptrs[(size_t)q->used[(size_t)q->proj[done + j]]] : nullptr;
            g.proj[j].dst = q->d_proj[done + j];
            g.proj[j].width = sc.width;
            g.proj[j].staged = nullptr;
        }
        {
            LaunchTimer t(ctx, 2);
            launch_gather(g, 0, s, t.start, t.stop);
        }
        HIPCHK(hipGetLastError());
        done += take;
    } while (done < np);
    return IMM3_OK;
}

// A small limit behind a limit scan: the offsets scan and the gather in ONE launch over the scanned tiles (k_limit_gather).
// `select id ... limit 10`: 7 + 9 us of k_scan + k_gather -> ~5.  Tuning variant 15: off.
constexpr int kLimitGatherGrid = 256;
static bool limit_gather_applies(const imm3_query *q) {
    if (!q->select_partial || !(q->limit > 0) || q->limit > kLimitGatherMaxRows || q->table || q->d_word_row_base || q->stage_written || q->ctx->filter_variant == 15) return false;
    if (q->n_chunks > (int64_t)kLimitGatherGrid * kLimitGatherMaxChunks || q->proj.size() > (size_t)kMaxProj || !q->d_limit_state) return false;
    for (int32_t pj : q->proj) {
        const SegCol &sc = q->seg->cols[(size_t)q->used[(size_t)pj]];
        if (!col_flat(sc) || (sc.width != 1 && sc.width != 2 && sc.width != 4)) return false;
    }
    return true;
}
static int launch_limit_gather_for(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    LimitGatherArgs g;
    std::memset(&g, 0, sizeof(g));
    g.bitmap = q->d_bitmap;
    g.finish = q->d_total;
    g.wg_state = q->d_limit_state;
    g.n_tiles = q->n_tiles;
    g.limit = q->limit;
    g.cap_rows = q->cap_rows;
    g.row_index = q->d_row_index;
    g.n_proj = (int32_t)q->proj.size();
    g.fault_wg = ctx->fault_wg;        // (tools' build only: imm3_ctx_inject_fault)
    g.max_polls = ctx->fault_max_polls;
    for (size_t j = 0; j < q->proj.size(); ++j) {
        const SegCol &sc = q->seg->cols[(size_t)q->used[(size_t)q->proj[j]]];
        g.proj[j].src = col_flat(sc);
        g.proj[j].dst = q->d_proj[j];
        g.proj[j].width = sc.width;
    }
    LaunchTimer t(ctx, 2);
    launch_limit_gather(g, (int)std::min<int64_t>(kLimitGatherGrid, std::max<int64_t>(q->n_chunks, 1)), ctx->stream, t.start, t.stop);
    HIPCHK(hipGetLastError());
    return IMM3_OK;
}

static int run_project(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    hipStream_t s = ctx->stream;
    if (q->n_tiles > 0 && limit_gather_applies(q)) {
        if (!q->d_row_index) {
            const int rc = ensure_row_capacity(q, 1);
            if (rc) return rc;
        }
        const int rc = launch_limit_gather_for(q);
        if (rc) return rc;
        q->offsets_valid = false; // (no offsets scan has run on this bitmap)
        q->ran_project = true;
        q->limit_gather_ran = true;
        return IMM3_OK;
    }
    q->limit_gather_ran = false;
    if (q->n_tiles > 0) {
        ScanArgs sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.bitmap = q->d_bitmap;
        sa.tile_offsets = q->d_tile_offsets;
        sa.chunk_sums = q->d_chunk_sums;
        sa.n_tiles = q->n_tiles;
        sa.finish = q->count_pending_scan ? q->d_total : nullptr;
        sa.scanned_tiles = q->select_partial ? q->d_total + kFinishLimitTiles : nullptr;
        if (q->stage_written && q->bitmap_lazy) { // no bitmap was stored: the tiles' counts come from the records' start table
            sa.rec_tile_start = q->d_tile_start;
            sa.rec_n_waves = (int64_t)q->stage_grid * kWavesPerBlock;
            sa.rec_main_tiles = q->stage_main_tiles;
            sa.rec_max_slots = q->stage_max_slots;
            sa.rec_T = q->stage_T;
        }
        {
            LaunchTimer t(ctx, 1);
            launch_scan(sa, s, t.start, t.stop);
        }
        HIPCHK(hipGetLastError());
        q->offsets_valid = true;
    }
    if (!(q->limit > 0) && !q->reserved && !q->d_row_index) {
        // Unlimited projection, no reservation, FIRST run: the output size is the count -> one synchronisation.  The arrays get
        // an eighth of headroom and every later run of the query writes into them without asking: a steady-state projecting
        // query never synchronises (a run that outgrows them is detected when its rows are fetched, and emitted again).
        unsigned long long total = 0;
        HIPCHK(hipMemcpyAsync(&total, q->d_total, sizeof(total), hipMemcpyDeviceToHost, s));
        // ... and, for the cost model, where the survivors are: the offsets scan's per-chunk counts (256 tiles each; 1.5 KB for 100 M
        // rows) say how densely they sit where they sit and how many of them in fully surviving chunks -- the whole segment, where
        // the sample at creation saw 0.5 % of it (a range of a sorted key between two sample chunks showed it nothing)
        std::vector<uint32_t> chunk_counts;
        if (!q->plan_pinned && !q->table && q->n_chunks > 0 && q->n_chunks <= (1 << 20) && q->offsets_valid) {
            chunk_counts.resize((size_t)q->n_chunks);
            HIPCHK(hipMemcpyAsync(chunk_counts.data(), q->d_chunk_sums, chunk_counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        ++q->run_syncs;
        if (!chunk_counts.empty() && total > 0) {
            const double chunk_rows = (double)kChunkTiles * kTileRows;
            double sum = 0.0, sum_sq = 0.0, sum_full = 0.0;
            for (size_t i = 0; i < chunk_counts.size(); ++i) {
                const double c = (double)chunk_counts[i];
                sum += c;
                sum_sq += c * c;
                if (c >= chunk_rows) sum_full += c;
            }
            if (sum > 0.0) {
                q->plan_density.sigma = (double)total / (double)std::max<int64_t>(q->n_rows, 1);
                q->plan_density.sloc = std::min(1.0, std::max(q->plan_density.sigma, sum_sq / (sum * chunk_rows)));
                q->plan_density.full = sum_full / sum;
                q->plan_have_density = true;
            }
        }
        {   // enough survivors for the gathered columns to be streamed instead?  Then this run is done again as one launch
            const int src = single_pass_stream_columns(q, total);
            if (src) return src;
            if (q->single_pass) return run_single_pass(q);
            if (single_pass_restore_wanted(q, total)) {
                q->sp_restore_pending = true; // (from the next run on)
                q->sp_restore_survivors = total;
            }
            records_drop_if_narrow(q, total); // (this run's rows then come from the bitmap)
            if (!q->d_stage_rec && q->bitmap_lazy) { // ... which the staging launch did not store: the select chain runs once more, plainly (the offsets stand: same counts)
                const int prc = run_select(q, false, false, false, true);
                if (prc) return prc;
                q->offsets_valid = true;
            }
        }
        const unsigned long long want = std::min<unsigned long long>((unsigned long long)std::max<int64_t>(q->n_rows, 1), total + total / 8 + 1024);
        const int rc = ensure_row_capacity(q, want);
        if (rc) return rc;
    } else if (!q->d_row_index) {
        const int rc = ensure_row_capacity(q, 1);
        if (rc) return rc;
    }
    if (q->n_tiles > 0) {
        const int rc = launch_project(q);
        if (rc) return rc;
    }
    q->ran_project = true;
    return IMM3_OK;
}

static int run_agg(imm3_query *q);
static bool agg_run_fuses(const imm3_query *q);

// a run recorded into an open capture: nothing in it may synchronise, allocate or use a second stream
static int capture_admit(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    if (!ctx->capture) return IMM3_OK;
    if (ctx->filter_variant == 2) return fail(IMM3_ERR_STATE, "tuning variant 2 (count reduce on the aux stream) cannot be captured");
    const bool sp = q->single_pass && !q->proj.empty() && !q->always_false && q->n_tiles > 0; // (writes its rows without knowing the count)
    if (!q->proj.empty() && !(q->limit > 0) && !q->reserved && !sp && !q->d_row_index)
        return fail(IMM3_ERR_STATE, "an unlimited projection sizes its output from the count on its first run (a synchronisation): run it once, or call imm3_query_reserve_rows, before capturing it");
    if (sp && !q->reserved && q->cap_rows < (uint64_t)q->n_rows)
        return fail(IMM3_ERR_STATE, "run the query once (or reserve rows) before capturing it: its output buffers are allocated on first use");
    if (!q->proj.empty() && !q->d_row_index) return fail(IMM3_ERR_STATE, "run the query once (or reserve rows) before capturing it: its output buffers are allocated on first use");
    auto &qs = ctx->capture->queries;
    if (std::find(qs.begin(), qs.end(), q) == qs.end()) {
        qs.push_back(q);
        ctx->capture->states.emplace_back();
    }
    return IMM3_OK;
}

// the run has been recorded: what it leaves in the handle is what every replay of the graph leaves (imm3_graph_launch)
static int capture_note(imm3_query *q, int rc) {
    imm3_ctx *ctx = q->ctx;
    if (rc || !ctx->capture) return rc;
    auto &qs = ctx->capture->queries;
    const auto it = std::find(qs.begin(), qs.end(), q);
    if (it == qs.end()) return rc;
    QueryRunState &st = ctx->capture->states[(size_t)(it - qs.begin())];
    st.ran_select = q->ran_select;
    st.ran_project = q->ran_project;
    st.bitmap_valid = q->bitmap_valid;
    st.ran_single_pass = q->ran_single_pass;
    st.stage_written = q->stage_written;
    st.bitmap_lazy = q->bitmap_lazy;
    st.agg_select_skipped = q->agg_select_skipped;
    st.count_pending_scan = q->count_pending_scan;
    st.has_pfor_pass = q->has_pfor_pass;
    st.ran_agg = q->ran_agg;
    st.offsets_valid = q->offsets_valid;
    st.select_partial = q->select_partial;
    return rc;
}

extern "C" int imm3_query_run_select(imm3_query *q) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE_RUN(q->ctx);
    const int ca = capture_admit(q);
    if (ca) return ca;
    q->ran_project = false;
    return capture_note(q, run_select(q, q->ctx->filter_variant == 2));
}

extern "C" int imm3_query_run_count(imm3_query *q) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE_RUN(q->ctx);
    const int ca = capture_admit(q);
    if (ca) return ca;
    q->ran_project = false;
    return capture_note(q, run_select(q, false, false, true));
}

extern "C" int imm3_query_join_count(imm3_query *q) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE(q->ctx);
    HIPCHK(hipSetDevice(q->ctx->device));
    // (the hand-off to device-side consumers of the count word: after a limit scan that stopped early the word holds the scanned
    // prefix's count -- the whole select runs first, as for imm3_query_count and the count all-reduce)
    const int arc = settle_agg_select(q);
    if (arc) return arc;
    const int wrc = settle_whole_select(q);
    if (wrc) return wrc;
    return join_total(q, q->ctx->stream);
}

extern "C" int imm3_query_run(imm3_query *q) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE_RUN(q->ctx);
    const int ca = capture_admit(q);
    if (ca) return ca;
    q->ran_project = false;
    // Reducing the count on the aux stream (tuning variant 2) measured SLOWER on MI355X / ROCm 7.2 (75.6 vs 67.1 us
    // per step: the cross-queue event packets cost more than the two same-queue launch gaps they remove), so the
    // default keeps the reduce on the main stream.
    if (q->sp_restore_pending && !q->ctx->capture) {
        const int rrc = single_pass_restore(q, q->sp_restore_survivors);
        if (rrc) return rrc;
    }
    if (q->single_pass && !q->proj.empty() && !q->always_false && q->n_tiles > 0) return capture_note(q, run_single_pass(q));
    const bool select_only = q->proj.empty() && !q->is_agg && q->ctx->filter_variant == 2;
    int rc = IMM3_OK;
    q->agg_select_skipped = agg_run_fuses(q);
    if (q->agg_select_skipped) q->ran_select = true; // (bitmap and count on demand: settle_agg_select)
    else rc = run_select(q, select_only, !q->proj.empty() && q->n_tiles > 0 && !q->always_false && q->ctx->filter_variant != 7);
    if (rc) return rc;
    if (!q->proj.empty()) rc = run_project(q);
    if (!rc && q->is_agg) rc = run_agg(q);
    return capture_note(q, rc);
}

extern "C" int imm3_query_sync(imm3_query *q) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE(q->ctx);
    HIPCHK(hipSetDevice(q->ctx->device));
    HIPCHK(hipStreamSynchronize(q->ctx->stream));
    if (q->ctx->aux) HIPCHK(hipStreamSynchronize(q->ctx->aux));
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// results
// ---------------------------------------------------------------------------------------------
extern "C" int imm3_query_layout(const imm3_query *q, int32_t *n_batches, int64_t *total_words, int64_t *n_rows) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    if (n_batches) *n_batches = (int32_t)(q->table ? q->table->batch_size.size() : q->layout->size.size());
    if (total_words) *total_words = q->n_words;
    if (n_rows) *n_rows = q->n_rows;
    return IMM3_OK;
}

extern "C" int imm3_query_batches(const imm3_query *q, int32_t *batch_size, int32_t *batch_oid, int64_t *batch_word_off) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    if (q->table) {
        const imm3_table *t = q->table;
        const size_t nb = t->batch_size.size();
        if (batch_size && nb) std::memcpy(batch_size, t->batch_size.data(), nb * sizeof(int32_t));
        if (batch_oid)
            for (size_t k = 0; k < nb; ++k) batch_oid[k] = (int32_t)((uint32_t)t->batch_k[k] * (uint32_t)q->table_block_size); // vecCounter * table.blockSize
        if (batch_word_off && nb) std::memcpy(batch_word_off, t->batch_word_off.data(), nb * sizeof(int64_t));
        return IMM3_OK;
    }
    const SegLayout &L = *q->layout;
    const size_t nb = L.size.size();
    if (batch_size && nb) std::memcpy(batch_size, L.size.data(), nb * sizeof(int32_t));
    if (batch_oid)
        for (size_t k = 0; k < nb; ++k) batch_oid[k] = (int32_t)((uint32_t)k * (uint32_t)q->table_block_size); // vecCounter * table.blockSize (Scan.scala:60)
    if (batch_word_off && nb) std::memcpy(batch_word_off, L.word_off.data(), nb * sizeof(int64_t));
    return IMM3_OK;
}

extern "C" int imm3_query_log_counts(imm3_query *q, uint64_t *device_log, uint64_t capacity) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE(q->ctx);
    HIPCHK(hipSetDevice(q->ctx->device));
    const unsigned long long v[3] = {(unsigned long long)(uintptr_t)device_log, 0ULL, device_log ? (unsigned long long)capacity : 0ULL};
    HIPCHK(hipMemcpyAsync(q->d_total + 5, v, sizeof(v), hipMemcpyHostToDevice, q->ctx->stream));
    HIPCHK(hipStreamSynchronize(q->ctx->stream)); // `v` is a stack array; also orders the switch after earlier runs
    q->count_log_on = device_log != nullptr;
    return IMM3_OK;
}

// the offsets scan over the bitmap of the last run (for a gather from the bitmap): tile offsets and chunk sums
static int scan_offsets(imm3_query *q) {
    if (q->offsets_valid || q->n_tiles <= 0) return IMM3_OK;
    ScanArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.bitmap = q->d_bitmap;
    sa.tile_offsets = q->d_tile_offsets;
    sa.chunk_sums = q->d_chunk_sums;
    sa.n_tiles = q->n_tiles;
    sa.scanned_tiles = q->select_partial ? q->d_total + kFinishLimitTiles : nullptr;
    launch_scan(sa, q->ctx->stream, nullptr, nullptr); // (finish = null: the count is already published)
    HIPCHK(hipGetLastError());
    q->offsets_valid = true;
    return IMM3_OK;
}

// Did the query's last single-pass launch give up on its rows?  `head` = the first kFinishDense + 1 words of the finish block as
// fetched AFTER that launch: its flags are tagged with its epoch, and the launch bumped the run counter exactly once.
static unsigned long long single_pass_flags(const unsigned long long *head) {
    const unsigned long long status = head[kFinishStatus], epoch_run = head[kFinishEpoch] - 1ULL;
    if (((status >> kStatusEpochShift) & kStatusEpochMask) != (epoch_run & kStatusEpochMask)) return 0ULL; // (an earlier run's flags)
    return status & (kStatusAbandoned | kStatusBusy);
}

// A projection with a limit stops its scan when the limit is reached (run_select, chunks): the bitmap and the count then cover the
// tiles scanned so far.  The reference never sees the batches behind the limit either (Project.scala:73-80); a caller that asks for
// the segment's count or bitmap all the same gets them exact: the whole select runs now, once (the rows were emitted from the scanned
// prefix and stay what they are -- they are the first `limit` survivors either way).
// The last run was a records run that stored no bitmap (run_select: bitmap_lazy): a getter wants it -- the select chain runs once
// more, plainly.  The rows that run emitted stay what they are (the same rows); a later re-gather takes them from the bitmap.
static int settle_lazy_bitmap(imm3_query *q) {
    if (q->bitmap_valid || !q->bitmap_lazy) return IMM3_OK;
    if (q->ctx->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context");
    q->force_plain_select = true;
    const int rc = run_select(q, false, false, false, true);
    q->force_plain_select = false;
    if (rc) return rc;
    return IMM3_OK; // (offsets_valid stands: the offsets scan counted the same survivors from the records)
}

static int settle_whole_select(imm3_query *q) {
    if (!q->select_partial) return IMM3_OK;
    if (q->ctx->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context");
    const int rc = run_select(q, false, false, false, true);
    if (rc) return rc;
    q->offsets_valid = false;
    return IMM3_OK;
}

// Every getter's first step after a single-pass run: read the run's status word (with the count and the dense-range tally, one
// copy).  The count and the bitmap of a run are exact whatever the flags say (imm3_project.hip: a work-group that gives up on the
// rows goes on in count + bitmap mode); only the ROWS of a flagged run are incomplete, and they are gathered here from the bitmap
// (offsets scan + k_gather into the same arrays).  Abandoned (a prefix never came although the device was this launch's: not
// every work-group resident?): the query keeps the bitmap path from now on.  Busy (another launch of the kernel owned the device --
// also when that made other work-groups of this launch time out, flags = busy | abandoned): this run only.
static int settle_single_pass(imm3_query *q) {
    if (!q->ran_single_pass || q->sp_verified) return IMM3_OK;
    imm3_ctx *ctx = q->ctx;
    static_assert(kFinishStatus == 2 && kFinishEpoch < kFinishDense, "count, status word, run counter and dense tally are fetched together");
    unsigned long long head[kFinishDense + 1] = {0}; // {count, rows emitted, status, ..., run counter, dense ranges}
    HIPCHK(hipMemcpyAsync(head, q->d_total, sizeof(head), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    q->sp_verified = true;
    const unsigned long long flags = single_pass_flags(head);
    if (!flags) {
        if (!q->plan_have_density && head[0] > 0 && q->n_rows > 0) { // no sample: what this run saw -- ranges that outgrew their ring mean dense stretches
            const double sigma = (double)head[0] / (double)q->n_rows, n_ranges = (double)q->sp_spans * kProjectStreamers, dense = (double)head[kFinishDense];
            q->plan_density.sigma = sigma;
            q->plan_density.sloc = dense > 0.02 * n_ranges ? std::min(1.0, sigma * n_ranges / dense) : sigma;
            q->plan_density.full = q->plan_density.sloc >= 0.95 && q->plan_density.sloc > 1.5 * sigma ? 1.0 : 0.0;
            q->plan_have_density = true;
        }
        single_pass_adapt(q, head[0], (int64_t)head[kFinishDense]); // (later runs: P from the selectivity this run saw)
        if (!q->sp_narrow_checked) { // (once: the data do not change)
            q->sp_narrow_checked = true;
            single_pass_drop_if_narrow(q, head[0]);
        }
        return IMM3_OK;
    }
    if (flags & kStatusBusy) ++q->sp_busy_runs;
    else {
        ++q->sp_abandoned_runs;
        graphs_mark_stale(ctx, q); // (a recorded run would take the abandoned path again)
        q->single_pass = false;
    }
    q->ran_single_pass = false; // (the rows the getters see come from the bitmap path)
    q->stage_written = false;
    int rc = scan_offsets(q);
    if (rc) return rc;
    return q->n_tiles > 0 ? launch_project(q) : IMM3_OK;
}

extern "C" int imm3_query_count(imm3_query *q, uint64_t *selected_rows) {
    if (!q || !selected_rows) return fail(IMM3_ERR_ARG, "null argument");
    CTX_LIVE(q->ctx);
    if (!q->ran_select) return fail(IMM3_ERR_STATE, "imm3_query_run has not been called");
    HIPCHK(hipSetDevice(q->ctx->device));
    unsigned long long total = 0;
    {
        const int arc = settle_agg_select(q);
        if (arc) return arc;
        const int src = settle_single_pass(q);
        if (src) return src;
        const int wrc = settle_whole_select(q);
        if (wrc) return wrc;
        const int jrc = join_total(q, q->ctx->stream);
        if (jrc) return jrc;
    }
    unsigned long long status = 0;
    HIPCHK(hipMemcpyAsync(&total, q->d_total, sizeof(total), hipMemcpyDeviceToHost, q->ctx->stream));
    if (q->has_pfor_pass) HIPCHK(hipMemcpyAsync(&status, q->d_total + 2, sizeof(status), hipMemcpyDeviceToHost, q->ctx->stream));
    HIPCHK(hipStreamSynchronize(q->ctx->stream));
    if (status) return fail(IMM3_ERR_LAYOUT, "malformed PFOR_INT block (width above 32, data past the block end, or count mismatch)");
    *selected_rows = total;
    return IMM3_OK;
}

extern "C" int imm3_query_bitmap(imm3_query *q, uint64_t *words_out, int64_t n_words) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE(q->ctx);
    if (!q->ran_select) return fail(IMM3_ERR_STATE, "imm3_query_run has not been called");
    if (!q->bitmap_valid && q->bitmap_lazy) {
        HIPCHK(hipSetDevice(q->ctx->device));
        const int lrc = settle_lazy_bitmap(q);
        if (lrc) return lrc;
    }
    if (!q->bitmap_valid) return fail(IMM3_ERR_STATE, "the last run was count-only (imm3_query_run_count): it stored no bitmap");
    if (n_words < 0 || n_words > q->n_words) return fail(IMM3_ERR_ARG, "n_words exceeds the bitmap");
    if (n_words && !words_out) return fail(IMM3_ERR_ARG, "words_out is null");
    HIPCHK(hipSetDevice(q->ctx->device));
    {
        const int arc = settle_agg_select(q);
        if (arc) return arc;
        const int src = settle_single_pass(q);
        if (src) return src;
        const int wrc = settle_whole_select(q);
        if (wrc) return wrc;
    }
    if (n_words) HIPCHK(hipMemcpyAsync(words_out, q->d_bitmap, (size_t)n_words * sizeof(uint64_t), hipMemcpyDeviceToHost, q->ctx->stream));
    HIPCHK(hipStreamSynchronize(q->ctx->stream));
    return IMM3_OK;
}

static int settle_rows(imm3_query *q, uint64_t *rows) {
    CTX_LIVE(q->ctx);
    if (!q->ran_project) return fail(IMM3_ERR_STATE, "no projection has been run (n_proj == 0 or imm3_query_run not called)");
    HIPCHK(hipSetDevice(q->ctx->device));
    {
        const int src = settle_single_pass(q);
        if (src) return src;
    }
    unsigned long long emit = 0;
    if (q->n_tiles > 0 && q->limit_gather_ran) {
        // k_limit_gather's look-back is bounded: a launch whose wait ran out tagged finish[kFinishLimitGaveUp] with its run and
        // wrote only some of the rows -- gather them the two-launch way (one copy brings the row count, the epoch and the tag)
        unsigned long long head[kFinishLimitGaveUp + 1];
        HIPCHK(hipMemcpyAsync(head, q->d_total, sizeof(head), hipMemcpyDeviceToHost, q->ctx->stream));
        HIPCHK(hipStreamSynchronize(q->ctx->stream));
        emit = head[1];
        if (head[kFinishLimitGaveUp] == (((head[kFinishEpoch] & 0x7FFFFFULL) << 1) | 1ULL)) {
            ++q->limit_gather_gave_up;
            q->limit_gather_ran = false;
            int rc = scan_offsets(q);
            if (rc) return rc;
            rc = launch_project(q);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(q->ctx->stream));
        }
    } else if (q->n_tiles > 0) {
        HIPCHK(hipMemcpyAsync(&emit, q->d_n_emit, sizeof(emit), hipMemcpyDeviceToHost, q->ctx->stream));
        HIPCHK(hipStreamSynchronize(q->ctx->stream));
    }
    if (emit > q->cap_rows) {
        // the reservation was too small: grow and gather again (offsets are still valid; a single-pass run made none:
        // they come from the bitmap now)
        int rc = ensure_row_capacity(q, q->reserved ? emit : std::min<unsigned long long>((unsigned long long)std::max<int64_t>(q->n_rows, 1), emit + emit / 8 + 1024));
        if (rc) return rc;
        rc = scan_offsets(q); // (a single-pass run made no offsets: they come from the bitmap now)
        if (rc) return rc;
        rc = launch_project(q); // (from the records when the run staged them, from the bitmap otherwise)
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(q->ctx->stream));
    }
    *rows = emit;
    return IMM3_OK;
}

extern "C" int imm3_query_row_count(imm3_query *q, uint64_t *rows) {
    if (!q || !rows) return fail(IMM3_ERR_ARG, "null argument");
    return settle_rows(q, rows);
}

extern "C" int imm3_query_fetch_rows(imm3_query *q, uint32_t *row_index_out, void *const *col_out, uint64_t max_rows) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE(q->ctx);
    uint64_t rows = 0;
    const int rc = settle_rows(q, &rows);
    if (rc) return rc;
    const uint64_t n = std::min(rows, max_rows);
    hipStream_t s = q->ctx->stream;
    if (n) {
        if (row_index_out) HIPCHK(hipMemcpyAsync(row_index_out, q->d_row_index, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        for (size_t j = 0; j < q->proj.size(); ++j) {
            if (!col_out || !col_out[j]) continue;
            const SegCol &sc = q->seg->cols[(size_t)q->used[(size_t)q->proj[j]]];
            HIPCHK(hipMemcpyAsync(col_out[j], q->d_proj[j], n * (uint64_t)sc.width, hipMemcpyDeviceToHost, s));
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    return IMM3_OK;
}

extern "C" int imm3_query_device_ptr(imm3_query *q, int32_t which, void **ptr) {
    if (!q || !ptr) return fail(IMM3_ERR_ARG, "null argument");
    switch (which) {
    case 0: *ptr = q->d_bitmap; return IMM3_OK;
    case 1: *ptr = q->d_total; return IMM3_OK;
    case 2: *ptr = q->d_row_index; return IMM3_OK;
    case 3: *ptr = q->d_n_emit; return IMM3_OK;
    case 4: *ptr = q->d_total + kFinishStatus; return IMM3_OK; // the status word (imm3.h: which device-side consumers must look at it)
    default:
        if (which >= 16 && (size_t)(which - 16) < q->d_proj.size()) {
            *ptr = q->d_proj[(size_t)(which - 16)];
            return IMM3_OK;
        }
        return fail(IMM3_ERR_ARG, "unknown device pointer id");
    }
}

extern "C" int imm3_query_plan(const imm3_query *q, int64_t *out, int32_t n) {
    if (!q || !out) return fail(IMM3_ERR_ARG, "null argument");
    const int64_t v[11] = {q->single_pass ? 1 : 0, q->sp_P, q->sp_grid, q->sp_spans, q->d_stage_rec ? 1 : 0,
                           (q->single_pass || q->d_stage_rec) ? rec_layout(q->stage_kinds, -1).dwords : 0, q->ran_single_pass ? 1 : 0, (int64_t)q->run_syncs,
                           (int64_t)q->sp_abandoned_runs, (int64_t)q->sp_busy_runs, (int64_t)q->limit_gather_gave_up};
    for (int32_t i = 0; i < n && i < 11; ++i) out[i] = v[i];
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// group-by aggregation (ProjectAggOp)
// ---------------------------------------------------------------------------------------------
static int query_create_agg_impl(imm3_ctx *ctx, const imm3_segment *seg, const imm3_table *table,
                                 const int32_t *used_cols, int32_t n_used,
                                 const imm3_select *sels, int32_t n_sels,
                                 const int32_t *group_cols, int32_t n_group,
                                 const imm3_aggregate *aggs, int32_t n_aggs,
                                 int32_t table_block_size, imm3_query **out) {
    if (!out) return fail(IMM3_ERR_ARG, "out is null");
    *out = nullptr;
    if (n_group < 0 || n_group > kMaxGroupCols || (n_group > 0 && !group_cols)) return fail(IMM3_ERR_ARG, "0..4 group columns are supported on the GPU path");
    if (n_aggs < 1 || n_aggs > kMaxAggs || !aggs) return fail(IMM3_ERR_ARG, "1..4 aggregates are supported on the GPU path");
    imm3_query *q = nullptr;
    int rc = query_create_impl(ctx, seg, table, used_cols, n_used, sels, n_sels, nullptr, 0, 0, table_block_size, &q);
    if (rc) return rc;
    std::unique_ptr<imm3_query, void (*)(imm3_query *)> guard(q, query_free);
    int key_bytes = 0;
    for (int32_t g = 0; g < n_group; ++g) {
        if (group_cols[g] < 0 || group_cols[g] >= n_used) return fail(IMM3_ERR_ARG, "group column is not among the used columns");
        key_bytes += seg->cols[(size_t)q->used[(size_t)group_cols[g]]].width;
    }
    if (key_bytes > 8) return fail(IMM3_ERR_ARG, "group key wider than 8 bytes is not supported on the GPU path");
    const bool has_batches = table ? !table->batch_size.empty() : !q->layout->size.empty();
    for (int32_t j = 0; j < n_aggs; ++j) {
        if (aggs[j].column < 0 || aggs[j].column >= n_used) return fail(IMM3_ERR_ARG, "aggregate column is not among the used columns");
        const SegCol &sc = seg->cols[(size_t)q->used[(size_t)aggs[j].column]];
        const bool is_str = sc.vcodec == IMM3_DENSE_STRING;
        if (aggs[j].kind != IMM3_AGG_COUNT && aggs[j].kind != IMM3_AGG_MIN && aggs[j].kind != IMM3_AGG_MAX) return fail(IMM3_ERR_ARG, "Unknown Aggregate type");
        // ProjectAggregate.scala:176-220: a String vector only takes CountAggr / MaxStringAggr
        if (has_batches && is_str && aggs[j].kind == IMM3_AGG_MIN) return fail(IMM3_ERR_UNSUPPORTED_VECTOR, "bad aggregator for this data type");
        if (is_str && aggs[j].kind == IMM3_AGG_MAX && sc.width > 8) return fail(IMM3_ERR_ARG, "MAX over strings wider than 8 bytes is not supported on the GPU path");
    }
    // group / aggregate columns are read row by row: PFOR_INT ones through their decoded form
    if (!table) {
        auto need_dense = [&](int32_t used_idx) -> int {
            const int32_t sci = q->used[(size_t)used_idx];
            if (!is_compressed(seg->cols[(size_t)sci].codec)) return IMM3_OK;
            for (auto &fp : q->preds)
                if (fp.seg_col == sci) fp.pfor = false;
            return ensure_dense(ctx, seg, sci);
        };
        for (int32_t g = 0; g < n_group; ++g) { rc = need_dense(group_cols[g]); if (rc) return rc; }
        for (int32_t j = 0; j < n_aggs; ++j) { rc = need_dense(aggs[j].column); if (rc) return rc; }
    }
    q->is_agg = true;
    q->group_cols.assign(group_cols, group_cols + n_group);
    q->aggs.assign(aggs, aggs + n_aggs);
    {   // SelectOp fused into the aggregation launch (k_group_agg_lanes' FUSED instances; whether the lanes form takes the query is
        // the launcher's call at run time)
        bool ok = !table && !q->ragged && !q->always_false && has_batches && q->n_rows > 0 && q->preds.size() <= (size_t)kMaxAggPreds;
        for (const auto &fp : q->preds)
            ok = ok && !fp.pfor && (fp.kind == KIND_I8 || fp.kind == KIND_I32) && col_flat(seg->cols[(size_t)fp.seg_col]) != nullptr;
        q->agg_fusable = ok;
    }
    // table capacity: twice the number of possible groups, bounded by the rows and by 2^27 slots
    double domain = 1.0;
    for (int b = 0; b < key_bytes; ++b) domain *= 256.0;
    const double bound = std::min<double>(domain, (double)std::max<int64_t>(q->n_rows, 1));
    uint64_t slots = 1024;
    while ((double)slots < 2.0 * bound && slots < (1ULL << 27)) slots <<= 1;
    q->agg_mask = (uint32_t)(slots - 1);
    HIPCHK(hipSetDevice(ctx->device));
    void *p = nullptr;
    const size_t n = (size_t)slots + 1;
    HIPCHK(pool_alloc(ctx, &p, n * sizeof(unsigned long long))); q->d_akeys = (unsigned long long *)p;
    HIPCHK(pool_alloc(ctx, &p, n * sizeof(uint32_t))); q->d_afirst = (uint32_t *)p;
    HIPCHK(pool_alloc(ctx, &p, n * sizeof(unsigned long long))); q->d_acounts = (unsigned long long *)p;
    HIPCHK(pool_alloc(ctx, &p, n * kMaxAggs * sizeof(long long))); q->d_avals = (long long *)p;
    HIPCHK(pool_alloc(ctx, &p, 2 * sizeof(uint32_t))); q->d_ameta = (uint32_t *)p;
    *out = guard.release();
    return IMM3_OK;
}

extern "C" int imm3_query_create_agg(imm3_ctx *ctx, const imm3_segment *seg,
                                     const int32_t *used_cols, int32_t n_used,
                                     const imm3_select *sels, int32_t n_sels,
                                     const int32_t *group_cols, int32_t n_group,
                                     const imm3_aggregate *aggs, int32_t n_aggs,
                                     int32_t table_block_size, imm3_query **out) {
    return query_create_agg_impl(ctx, seg, nullptr, used_cols, n_used, sels, n_sels, group_cols, n_group, aggs, n_aggs, table_block_size, out);
}

extern "C" int imm3_query_create_table_agg(imm3_ctx *ctx, const imm3_table *table,
                                           const int32_t *used_cols, int32_t n_used,
                                           const imm3_select *sels, int32_t n_sels,
                                           const int32_t *group_cols, int32_t n_group,
                                           const imm3_aggregate *aggs, int32_t n_aggs,
                                           int32_t table_block_size, imm3_query **out) {
    if (!table || table->segs.empty()) return fail(IMM3_ERR_ARG, "table is null or empty");
    return query_create_agg_impl(ctx, table->segs[0], table, used_cols, n_used, sels, n_sels, group_cols, n_group, aggs, n_aggs, table_block_size, out);
}

static void fill_agg_args(const imm3_query *q, AggArgs &a) {
    std::memset(&a, 0, sizeof(a));
    a.bitmap = q->d_bitmap;
    a.n_words = q->n_words;
    a.n_tiles = q->n_tiles;
    a.n_rows = q->n_rows;
    a.word_row_base = q->d_word_row_base;
    int shift = 0;
    for (size_t g = 0; g < q->group_cols.size(); ++g) {
        const SegCol &sc = q->seg->cols[(size_t)q->used[(size_t)q->group_cols[g]]];
        a.groups[g].data = col_flat(sc);
        a.groups[g].tile_ptrs = q->table ? (const void *const *)q->table->d_tile_ptrs[(size_t)q->used[(size_t)q->group_cols[g]]] : nullptr;
        a.groups[g].width = sc.width;
        a.groups[g].shift = shift;
        shift += sc.width;
    }
    a.n_group = (int32_t)q->group_cols.size();
    for (size_t j = 0; j < q->aggs.size(); ++j) {
        const SegCol &sc = q->seg->cols[(size_t)q->used[(size_t)q->aggs[j].column]];
        a.aggs[j].data = col_flat(sc);
        a.aggs[j].tile_ptrs = q->table ? (const void *const *)q->table->d_tile_ptrs[(size_t)q->used[(size_t)q->aggs[j].column]] : nullptr;
        a.aggs[j].width = sc.width;
        a.aggs[j].kind = q->aggs[j].kind;
        a.aggs[j].is_str = sc.vcodec == IMM3_DENSE_STRING;
    }
    a.n_agg = (int32_t)q->aggs.size();
    if (q->agg_fusable && q->ctx->filter_variant != 17) { // (tuning 17: filter launch + aggregation launch, as before round 5)
        int first_value = -1; // the aggregate whose rows the lanes form keeps in registers: the first one that is not a count
        for (size_t j = 0; j < q->aggs.size() && first_value < 0; ++j)
            if (q->aggs[j].kind != IMM3_AGG_COUNT) first_value = (int)j;
        for (const auto &fp : q->preds) {
            AggPred &f = a.fused[a.n_fused++];
            const SegCol &sc = q->seg->cols[(size_t)fp.seg_col];
            f.data = col_flat(sc);
            f.width = sc.width;
            f.lo = (int32_t)fp.lo;
            f.hi = (int32_t)fp.hi;
            f.share = first_value >= 0 && q->used[(size_t)q->aggs[(size_t)first_value].column] == fp.seg_col ? 1 : 0;
        }
        a.fused_all = q->preds.empty() ? 1 : 0;
    }
    a.keys = q->d_akeys;
    a.first = q->d_afirst;
    a.counts = q->d_acounts;
    a.vals = q->d_avals;
    a.mask = q->agg_mask;
    a.n_groups = q->d_ameta;
    a.overflow = q->d_ameta + 1;
    a.out_cap = q->out_cap;
    a.out_keys = q->d_okeys;
    a.out_first = q->d_ofirst;
    a.out_counts = q->d_ocounts;
    a.out_vals = q->d_ovals;
}

static void agg_launch_args(const imm3_query *q, AggArgs &a) {
    fill_agg_args(q, a);
    // tuning variants 100 + AggForm start the chain at that form (tools/aggexp.py); 140 + x: ablation x of the tools' build
    const int fv = q->ctx->filter_variant;
    a.first_form = (fv >= 100 && fv <= 100 + AGG_FORM_GENERAL) ? fv - 100 : q->agg_first_form; // (agg_first_form: past the forms this query's keys overflowed)
    a.ablate = fv >= 140 ? fv - 100 : 0;
}
// will this run's aggregation launch evaluate the select chain itself?  (then no select launch precedes it)
static bool agg_run_fuses(const imm3_query *q) {
    if (!q->is_agg || !q->agg_fusable || q->count_log_on) return false; // (a count log wants every run's count on the device: the select launch produces it)
    AggArgs a;
    agg_launch_args(q, a);
    return group_agg_fuses_select(a);
}
// A getter wants the bitmap or the selected-row count of an aggregation whose last run fused the select: the select chain runs now.
static int settle_agg_select(imm3_query *q) {
    if (!q->agg_select_skipped) return IMM3_OK;
    if (q->ctx->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context");
    const int rc = run_select(q, false);
    if (rc) return rc;
    q->agg_select_skipped = false;
    return IMM3_OK;
}

static int run_agg(imm3_query *q) {
    imm3_ctx *ctx = q->ctx;
    AggArgs a;
    agg_launch_args(q, a);
    if (!q->agg_select_skipped) { a.n_fused = 0; a.fused_all = 0; } // (the select ran: the bitmap is what this launch reads)
    LaunchTimer t(ctx, 4);
    launch_group_agg(a, ctx->stream, t.start, t.stop);
    HIPCHK(hipGetLastError());
    q->ran_agg = true;
    return IMM3_OK;
}

// collect the occupied slots; grows the dense output and collects again if it was too small
static int settle_groups(imm3_query *q, uint32_t *n_groups) {
    CTX_LIVE(q->ctx);
    if (!q->is_agg || !q->ran_agg) return fail(IMM3_ERR_STATE, "no aggregation has been run");
    imm3_ctx *ctx = q->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    for (int attempt = 0; attempt < 5; ++attempt) {
        AggArgs a;
        fill_agg_args(q, a);
        HIPCHK(hipMemsetAsync(q->d_ameta, 0, sizeof(uint32_t), s)); // n_groups only; keep the overflow flag
        launch_group_collect(a, s);
        HIPCHK(hipGetLastError());
        uint32_t meta[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(meta, q->d_ameta, sizeof(meta), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (meta[1] == 2 || meta[1] == 3) { // a fast form's per-work-group table filled up: aggregate again with the next form
            AggArgs g;                       // (3: k_group_agg_lanes -> k_group_agg_direct; 2: -> the general kernel)
            fill_agg_args(q, g);
            // lanes (63 keys) -> lanes (127 keys) -> direct -> general
            q->agg_first_form = meta[1] == 3 ? (q->agg_first_form == AGG_FORM_LANES ? AGG_FORM_LANES_WIDE : AGG_FORM_DIRECT) : AGG_FORM_GENERAL;
            g.first_form = q->agg_first_form;
            {   // (the forms behind the 63-key lanes form read the bitmap: a run that fused the select has none yet)
                const int arc = settle_agg_select(q);
                if (arc) return arc;
                g.n_fused = 0;
                g.fused_all = 0;
            }
            launch_group_agg(g, s, nullptr, nullptr);
            HIPCHK(hipGetLastError());
            continue;
        }
        if (meta[1]) return fail(IMM3_ERR_LAYOUT, "more distinct groups than the aggregation table holds (2^27)");
        if (meta[0] <= q->out_cap) { *n_groups = meta[0]; return IMM3_OK; }
        if (q->d_okeys) graphs_mark_stale(ctx, q);
        pool_release(ctx, q->d_okeys); pool_release(ctx, q->d_ofirst); pool_release(ctx, q->d_ocounts); pool_release(ctx, q->d_ovals);
        q->d_okeys = nullptr; q->d_ofirst = nullptr; q->d_ocounts = nullptr; q->d_ovals = nullptr;
        void *p = nullptr;
        const size_t n = meta[0];
        HIPCHK(pool_alloc(ctx, &p, n * sizeof(unsigned long long))); q->d_okeys = (unsigned long long *)p;
        HIPCHK(pool_alloc(ctx, &p, n * sizeof(uint32_t))); q->d_ofirst = (uint32_t *)p;
        HIPCHK(pool_alloc(ctx, &p, n * sizeof(unsigned long long))); q->d_ocounts = (unsigned long long *)p;
        HIPCHK(pool_alloc(ctx, &p, n * kMaxAggs * sizeof(long long))); q->d_ovals = (long long *)p;
        q->out_cap = meta[0];
    }
    return fail(IMM3_ERR_DEVICE, "group collection did not converge");
}

int imm3::query_groups(imm3_query *q, uint32_t *n_groups) { return settle_groups(q, n_groups); }

extern "C" int imm3_query_agg_shape(const imm3_query *q, int32_t *n_group_cols, int32_t *n_aggs, int32_t *key_bytes) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    if (!q->is_agg) return fail(IMM3_ERR_ARG, "not an aggregation query");
    if (n_group_cols) *n_group_cols = (int32_t)q->group_cols.size();
    if (n_aggs) *n_aggs = (int32_t)q->aggs.size();
    if (key_bytes) {
        int32_t kb = 0;
        for (int32_t g : q->group_cols) kb += q->seg->cols[(size_t)q->used[(size_t)g]].width;
        *key_bytes = kb;
    }
    return IMM3_OK;
}

extern "C" int imm3_query_group_count(imm3_query *q, uint32_t *n_groups) {
    if (!q || !n_groups) return fail(IMM3_ERR_ARG, "null argument");
    return settle_groups(q, n_groups);
}

extern "C" int imm3_query_fetch_groups(imm3_query *q, uint64_t *keys, uint32_t *first_row, uint64_t *counts, int64_t *vals, uint32_t max_groups) {
    if (!q) return fail(IMM3_ERR_ARG, "query is null");
    CTX_LIVE(q->ctx);
    uint32_t n = 0;
    const int rc = settle_groups(q, &n);
    if (rc) return rc;
    std::vector<unsigned long long> hk(n), hc(n);
    std::vector<uint32_t> hf(n);
    std::vector<long long> hv((size_t)n * kMaxAggs);
    hipStream_t s = q->ctx->stream;
    if (n) {
        HIPCHK(hipMemcpyAsync(hk.data(), q->d_okeys, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hf.data(), q->d_ofirst, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hc.data(), q->d_ocounts, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hv.data(), q->d_ovals, (size_t)n * kMaxAggs * sizeof(long long), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return hf[x] < hf[y]; }); // first-seen order
    const size_t na = q->aggs.size();
    for (uint32_t o = 0; o < n && o < max_groups; ++o) {
        const uint32_t i = order[o];
        if (keys) keys[o] = hk[i];
        if (first_row) first_row[o] = hf[i];
        if (counts) counts[o] = hc[i];
        if (vals)
            for (size_t j = 0; j < na; ++j)
                vals[(size_t)o * na + j] = q->aggs[j].kind == IMM3_AGG_COUNT ? (int64_t)hc[i] : (int64_t)hv[(size_t)i * kMaxAggs + j];
    }
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// write side of the PFOR_INT codec (host only; host/codec.hpp)
// ---------------------------------------------------------------------------------------------
extern "C" uint64_t imm3_pfor_encode_bound(int32_t n_values) {
    return n_values < 0 ? 0 : (uint64_t)immutabledb::codec::pforEncodeBound(n_values);
}

extern "C" int imm3_pfor_encode_block(const int32_t *values, int32_t n_values, void *out, uint64_t cap, uint64_t *bytes_out) {
    if (n_values < 0 || (n_values > 0 && !values) || !out || !bytes_out) return fail(IMM3_ERR_ARG, "bad argument");
    const std::vector<uint8_t> blk = immutabledb::codec::pforEncodeBlock(values, n_values);
    if (blk.size() > cap) return fail(IMM3_ERR_ARG, "output buffer too small (see imm3_pfor_encode_bound)");
    std::memcpy(out, blk.data(), blk.size());
    *bytes_out = blk.size();
    return IMM3_OK;
}

extern "C" int imm3_pfor_encode_column(const int32_t *values, uint64_t n_values, int32_t block_rows, void *out, uint64_t cap,
                                       int32_t *offsets_out, uint64_t *bytes_out) {
    if ((n_values > 0 && !values) || block_rows <= 0 || !out || !offsets_out || !bytes_out) return fail(IMM3_ERR_ARG, "bad argument");
    uint64_t pos = 0;
    size_t k = 0;
    offsets_out[0] = 0;
    for (uint64_t r = 0; r < n_values; r += (uint64_t)block_rows) {
        const int32_t n = (int32_t)std::min<uint64_t>((uint64_t)block_rows, n_values - r);
        const std::vector<uint8_t> blk = immutabledb::codec::pforEncodeBlock(values + r, n);
        if (pos + blk.size() > cap) return fail(IMM3_ERR_ARG, "output buffer too small");
        if (pos + blk.size() > 0x7FFFFFFFULL) return fail(IMM3_ERR_LAYOUT, "segment data above 2 GiB (blockOffset is an Int, Segment.scala:33)");
        std::memcpy((uint8_t *)out + pos, blk.data(), blk.size());
        pos += blk.size();
        offsets_out[++k] = (int32_t)pos;
    }
    *bytes_out = pos;
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// write side of the snappy block format (host only; host/codec.hpp)
// ---------------------------------------------------------------------------------------------
extern "C" uint64_t imm3_snappy_encode_bound(uint64_t n_bytes) { return (uint64_t)immutabledb::codec::snappyEncodeBound((size_t)n_bytes); }

extern "C" int imm3_snappy_encode_block(const void *bytes, uint64_t n_bytes, void *out, uint64_t cap, uint64_t *bytes_out) {
    if ((n_bytes > 0 && !bytes) || !out || !bytes_out) return fail(IMM3_ERR_ARG, "bad argument");
    const std::vector<uint8_t> blk = immutabledb::codec::snappyEncodeBlock((const uint8_t *)bytes, (size_t)n_bytes);
    if (blk.size() > cap) return fail(IMM3_ERR_ARG, "output buffer too small (see imm3_snappy_encode_bound)");
    std::memcpy(out, blk.data(), blk.size());
    *bytes_out = blk.size();
    return IMM3_OK;
}
