// imm3_comm.cpp -- the one collective of the path, behind the C ABI: the selected-row counts of the per-segment
// pipelines (one PipelineThread per segment, engine/src/main/scala/immutabledb/engine/Engine.scala:176-180, whose
// results meet on the consumer thread, :190-196) summed over the GPUs with ONE 8-byte ncclAllReduce(sum, ncclUint64)
// over RCCL / xGMI.  Nothing else is exchanged: segments are sharded s mod G and bitmaps, row
// lists and oids stay on the GPU that produced them.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a host that never creates a communicator -- one GPU -- does not
// need the library, and a host that already carries it (PyTorch-ROCm ships its own copy under the same soname) gets
// that copy instead of a second one.
#include "imm3_handles.h"

#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums only; every call goes through the table below

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

using namespace imm3;

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error; // why the library could not be bound
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_bind() {
    // IMM3_RCCL_LIB: another library that exports the nine nccl* entry points bound below, tried first.  The test suite's
    // loopback transport (tests/native/loopback_rccl.cpp: two ranks of ONE process on ONE device, which RCCL itself refuses)
    // comes in this way, so that the world > 1 paths of this file run where no second GPU exists.
    const char *override_path = getenv("IMM3_RCCL_LIB");
    const char *names[] = {override_path && *override_path ? override_path : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_rccl.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.so) break;
        if (override_path && *override_path && n == names[0]) { // (an override that does not load is an error, not a silent fall-back)
            const char *e = dlerror();
            g_rccl.error = std::string("cannot load IMM3_RCCL_LIB=") + override_path + ": " + (e ? e : "?");
            return;
        }
    }
    if (!g_rccl.so) {
        const char *e = dlerror();
        g_rccl.error = std::string("cannot load librccl.so.1: ") + (e ? e : "?");
        return;
    }
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(g_rccl.so, name);
        if (!p && g_rccl.error.empty()) g_rccl.error = std::string("librccl has no symbol ") + name;
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
}

int rccl_ready() {
    std::call_once(g_rccl_once, rccl_bind);
    if (!g_rccl.error.empty()) return fail(IMM3_ERR_DEVICE, g_rccl.error);
    return IMM3_OK;
}

#define NCCLCHK(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess) return fail(IMM3_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); \
    } while (0)

} // namespace

// Streams.  The collective runs on the communicator's OWN stream, fenced by events: it starts after everything already
// enqueued on the context's stream (the scans that produce the counts, the kernel that sums them) and the context's
// stream never waits for it -- the next pass's scans start right behind the sum while the 8 bytes cross xGMI (a
// latency-bound message: tens of microseconds that would otherwise sit between two 60-200 us scans).  imm3_comm_sync
// (host) / imm3_comm_join (stream side) are the ways back.
struct imm3_comm {
    imm3_ctx *ctx = nullptr;      // the GPU this rank drives
    ncclComm_t nccl = nullptr;
    int32_t world = 1, rank = 0;
    hipStream_t stream = nullptr; // where the collectives run
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool in_flight = false;       // a collective has been enqueued since the last join / sync
    unsigned long long *d_slot = nullptr; // send/receive word of imm3_comm_allreduce_count when the caller passes no buffer
    // tools' build (imm3_comm_debug_standin): a kernel with the footprint of RCCL's, in front of every count all-reduce
    int32_t standin_wgs = 0;
    uint32_t standin_ticks = 0;   // how long each of its work-groups spins, in ticks of the 100 MHz device clock
    bool reserves = false;        // counted in ctx->comms_attached: its collectives launch kernels (more than one rank, or the stand-in)
};

// A communicator whose collectives put KERNELS on the device -- more than one rank (a one-rank all-reduce launches none), or the
// tools' stand-in -- makes the one-launch plans of its context leave a CU per XCD free (imm3_api.cpp: single_pass_run_grid).
static void comm_reserve(imm3_comm *c, bool on) {
    if (on == c->reserves) return;
    c->reserves = on;
    if (on) c->ctx->comms_attached.fetch_add(1, std::memory_order_relaxed);
    else c->ctx->comms_attached.fetch_sub(1, std::memory_order_relaxed);
}

#ifdef IMM3_ABLATE
// A stand-in for the kernel RCCL launches for the count all-reduce when there are other ranks: same footprint -- 512 threads,
// 37 664 bytes of LDS, 256 vector registers per lane, as ncclDevKernel_Generic_* is built for gfx950 in ROCm 7.2 (read from
// librccl.so's code object) -- so it takes a CU the way that kernel does: it cannot share one with a work-group of the one-launch
// projection (LDS and registers), which is what decides whether the next pass waits for the collective or the collective for
// the pass.  Spins on the device clock.  One GPU is enough to measure that (tools/overlap_probe.py).
__global__ __launch_bounds__(512) void k_comm_standin(uint32_t ticks, unsigned long long *sink) {
    __shared__ unsigned char s_fill[37664];
    asm volatile("; footprint" ::: "v255");   // (256 vector registers per lane, like the kernel it stands for)
    if (threadIdx.x < 64) s_fill[threadIdx.x] = (unsigned char)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (s_fill[threadIdx.x & 63] == 255 && sink) *sink = t0;
}
#endif

// the collective on `buf` may start once the context's stream has reached this point
static int fence_in(imm3_comm *c) {
    HIPCHK(hipEventRecord(c->ev_ready, c->ctx->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    return IMM3_OK;
}
static int fence_out(imm3_comm *c) {
    HIPCHK(hipEventRecord(c->ev_done, c->stream));
    c->in_flight = true;
    return IMM3_OK;
}

static int comm_finish(imm3_ctx *ctx, ncclComm_t nc, int32_t world, int32_t rank, imm3_comm **out) {
    std::unique_ptr<imm3_comm> c(new imm3_comm());
    c->nccl = nc;
    c->world = world;
    c->rank = rank;
    void *p = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(&p, 64);
    if (e == hipSuccess) e = hipMemset(p, 0, 64);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)g_rccl.CommDestroy(nc);
        (void)hipFree(p);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
        if (c->ev_done) (void)hipEventDestroy(c->ev_done);
        return fail(IMM3_ERR_DEVICE, std::string("communicator scratch: ") + hipGetErrorString(e));
    }
    c->d_slot = (unsigned long long *)p;
    c->ctx = ctx;
    ctx_retain(ctx);
    comm_reserve(c.get(), world > 1);
    *out = c.release();
    return IMM3_OK;
}

extern "C" int imm3_comm_debug_standin(imm3_comm *c, int32_t work_groups, uint32_t spin_us) {
    if (!c) return fail(IMM3_ERR_ARG, "comm is null");
#ifndef IMM3_ABLATE
    if (work_groups > 0) return fail(IMM3_ERR_STATE, "the collective's stand-in kernel exists only in the tools' build of the library (make -C csrc ablate)");
#endif
    if (work_groups < 0 || work_groups > 64) return fail(IMM3_ERR_ARG, "0..64 work-groups");
    c->standin_wgs = work_groups;
    c->standin_ticks = spin_us * 100u;
    comm_reserve(c, c->world > 1 || work_groups > 0);
    return IMM3_OK;
}

extern "C" int imm3_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return fail(IMM3_ERR_ARG, "id_out is null");
    const int rc = rccl_ready();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == IMM3_COMM_ID_BYTES, "IMM3_COMM_ID_BYTES must be RCCL's unique-id size");
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return IMM3_OK;
}

extern "C" int imm3_comm_create(imm3_ctx *ctx, int32_t world, int32_t rank, const uint8_t *id, imm3_comm **out) {
    if (!out) return fail(IMM3_ERR_ARG, "out is null");
    *out = nullptr;
    if (!ctx) return fail(IMM3_ERR_ARG, "ctx is null");
    if (ctx->closed) return fail(IMM3_ERR_STATE, "the context has been destroyed");
    if (world < 1 || rank < 0 || rank >= world || !id) return fail(IMM3_ERR_ARG, "bad world / rank / id");
    const int rc = rccl_ready();
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t nc = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&nc, world, uid, rank));
    return comm_finish(ctx, nc, world, rank, out);
}

extern "C" int imm3_comm_create_all(imm3_ctx *const *ctxs, int32_t n, imm3_comm **out) {
    if (!out) return fail(IMM3_ERR_ARG, "out is null");
    for (int32_t i = 0; i < n; ++i) out[i] = nullptr;
    if (n < 1 || !ctxs) return fail(IMM3_ERR_ARG, "a communicator needs at least one context");
    std::vector<int> devs((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(IMM3_ERR_ARG, "ctx is null");
        if (ctxs[i]->closed) return fail(IMM3_ERR_STATE, "a context has been destroyed");
        devs[(size_t)i] = ctxs[i]->device;
        for (int32_t j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i]) return fail(IMM3_ERR_ARG, "one context per device: two contexts share device " + std::to_string(devs[(size_t)i]));
    }
    const int rc = rccl_ready();
    if (rc) return rc;
    std::vector<ncclComm_t> ncs((size_t)n, nullptr);
    NCCLCHK(g_rccl.CommInitAll(ncs.data(), n, devs.data()));
    for (int32_t i = 0; i < n; ++i) {
        const int frc = comm_finish(ctxs[i], ncs[(size_t)i], n, i, &out[i]);
        if (frc) { // all or nothing: the communicators made so far go too (the caller only sees nulls)
            const std::string why = imm3_last_error();
            for (int32_t j = i + 1; j < n; ++j) (void)g_rccl.CommDestroy(ncs[(size_t)j]);
            for (int32_t j = 0; j < i; ++j) {
                (void)imm3_comm_destroy(out[j]);
                out[j] = nullptr;
            }
            return fail(frc, why);
        }
    }
    return IMM3_OK;
}

extern "C" int imm3_comm_destroy(imm3_comm *c) {
    if (!c) return IMM3_OK;
    (void)hipSetDevice(c->ctx->device);
    {
        imm3::GateScope gate(&c->ctx->gate);
        if (!c->ctx->closed) (void)hipStreamSynchronize(c->ctx->stream);
    }
    (void)hipStreamSynchronize(c->stream);
    if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
    (void)hipStreamDestroy(c->stream);
    (void)hipEventDestroy(c->ev_ready);
    (void)hipEventDestroy(c->ev_done);
    (void)hipFree(c->d_slot);
    comm_reserve(c, false);
    ctx_release(c->ctx);
    delete c;
    return IMM3_OK;
}

extern "C" int imm3_comm_info(const imm3_comm *c, int32_t *world, int32_t *rank) {
    if (!c) return fail(IMM3_ERR_ARG, "comm is null");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return IMM3_OK;
}

extern "C" int imm3_comm_allreduce_u64(imm3_comm *c, uint64_t *device_buf, uint64_t n) {
    if (!c || !device_buf || n == 0) return fail(IMM3_ERR_ARG, "bad argument");
    imm3::GateScope gate(&c->ctx->gate);
    if (c->ctx->closed) return fail(IMM3_ERR_STATE, "the context of this communicator has been destroyed");
    HIPCHK(hipSetDevice(c->ctx->device));
    int rc = fence_in(c);
    if (rc) return rc;
    NCCLCHK(g_rccl.AllReduce(device_buf, device_buf, (size_t)n, ncclUint64, ncclSum, c->nccl, c->stream));
    return fence_out(c);
}

extern "C" int imm3_comm_sync(imm3_comm *c) {
    if (!c) return fail(IMM3_ERR_ARG, "comm is null");
    HIPCHK(hipSetDevice(c->ctx->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->in_flight = false;
    return IMM3_OK;
}

extern "C" int imm3_comm_join(imm3_comm *c) {
    if (!c) return fail(IMM3_ERR_ARG, "comm is null");
    imm3::GateScope gate(&c->ctx->gate);
    if (c->ctx->closed) return fail(IMM3_ERR_STATE, "the context of this communicator has been destroyed");
    HIPCHK(hipSetDevice(c->ctx->device));
    if (c->in_flight) HIPCHK(hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
    return IMM3_OK;
}

// this rank's part: the counts of its queries summed into `dst` on the context's stream, behind the scans that
// produce them (same stream; a count reduced on the aux stream is joined first)
static int local_sum(imm3_comm *c, imm3_query *const *queries, int32_t n_queries, unsigned long long *dst) {
    imm3_ctx *ctx = c->ctx;
    hipStream_t s = ctx->stream;
    // the communicator's own word may still be travelling from the previous call: wait for that collective (stream side)
    if (dst == c->d_slot && c->in_flight) HIPCHK(hipStreamWaitEvent(s, c->ev_done, 0));
    if (n_queries == 0) {
        HIPCHK(hipMemsetAsync(dst, 0, sizeof(unsigned long long), s));
        return IMM3_OK;
    }
    for (int32_t base = 0; base < n_queries; base += kMaxSumCounts) {
        SumCountsArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n = std::min<int32_t>(kMaxSumCounts, n_queries - base);
        for (int32_t i = 0; i < a.n; ++i) {
            imm3_query *q = queries[base + i];
            if (!q) return fail(IMM3_ERR_ARG, "query is null");
            if (q->ctx != ctx) return fail(IMM3_ERR_ARG, "the query runs on another context than the communicator");
            if (!q->ran_select) return fail(IMM3_ERR_STATE, "imm3_query_run has not been called");
            const int jrc = join_query_count(q, s);
            if (jrc) return jrc;
            a.src[i] = q->d_total;
        }
        a.accumulate = base > 0;
        a.dst = dst;
        launch_sum_counts(a, s);
        HIPCHK(hipGetLastError());
    }
    return IMM3_OK;
}

extern "C" int imm3_comm_allreduce_count(imm3_comm *c, imm3_query *const *queries, int32_t n_queries, uint64_t *device_out,
                                         uint64_t *host_out) {
    if (!c || n_queries < 0 || (n_queries > 0 && !queries)) return fail(IMM3_ERR_ARG, "bad argument");
    imm3::GateScope gate(&c->ctx->gate);
    if (c->ctx->closed) return fail(IMM3_ERR_STATE, "the context of this communicator has been destroyed");
    HIPCHK(hipSetDevice(c->ctx->device));
    unsigned long long *dst = device_out ? (unsigned long long *)device_out : c->d_slot;
    int rc = local_sum(c, queries, n_queries, dst);
    if (rc) return rc;
    rc = fence_in(c);
    if (rc) return rc;
#ifdef IMM3_ABLATE
    if (c->standin_wgs > 0) { // (tools: what a real multi-rank all-reduce puts on the device at this point)
        hipLaunchKernelGGL(k_comm_standin, dim3((unsigned)c->standin_wgs), dim3(512), 0, c->stream, c->standin_ticks, (unsigned long long *)nullptr);
        HIPCHK(hipGetLastError());
    }
#endif
    NCCLCHK(g_rccl.AllReduce(dst, dst, 1, ncclUint64, ncclSum, c->nccl, c->stream));
    rc = fence_out(c);
    if (rc) return rc;
    if (host_out) {
        unsigned long long v = 0;
        HIPCHK(hipMemcpyAsync(&v, dst, sizeof(v), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->in_flight = false;
        *host_out = v;
    }
    return IMM3_OK;
}

// Single-process shape (one JVM, every GPU): the per-device collectives are issued by ONE thread, so they go into a
// ncclGroupStart / ncclGroupEnd section (RCCL would otherwise wait in the first call for the ranks behind it).
extern "C" int imm3_comm_allreduce_count_all(imm3_comm *const *comms, int32_t n_comms, imm3_query *const *const *queries,
                                             const int32_t *n_queries, uint64_t *host_out) {
    if (!comms || n_comms < 1 || !n_queries || !queries) return fail(IMM3_ERR_ARG, "bad argument");
    for (int32_t i = 0; i < n_comms; ++i)
        if (!comms[i] || n_queries[i] < 0 || (n_queries[i] > 0 && !queries[i])) return fail(IMM3_ERR_ARG, "bad argument");
    std::vector<std::unique_ptr<imm3::GateScope>> gates; // one context per device, each with its own capture gate
    for (int32_t i = 0; i < n_comms; ++i) gates.emplace_back(new imm3::GateScope(&comms[i]->ctx->gate));
    for (int32_t i = 0; i < n_comms; ++i) {
        if (comms[i]->ctx->closed) return fail(IMM3_ERR_STATE, "the context of a communicator has been destroyed");
        HIPCHK(hipSetDevice(comms[i]->ctx->device));
        int rc = local_sum(comms[i], queries[i], n_queries[i], comms[i]->d_slot);
        if (!rc) rc = fence_in(comms[i]);
        if (rc) return rc;
    }
    NCCLCHK(g_rccl.GroupStart());
    for (int32_t i = 0; i < n_comms; ++i) {
        imm3_comm *c = comms[i];
        const ncclResult_t r = g_rccl.AllReduce(c->d_slot, c->d_slot, 1, ncclUint64, ncclSum, c->nccl, c->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail(IMM3_ERR_DEVICE, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
        }
    }
    NCCLCHK(g_rccl.GroupEnd());
    for (int32_t i = 0; i < n_comms; ++i) {
        HIPCHK(hipSetDevice(comms[i]->ctx->device));
        const int rc = fence_out(comms[i]);
        if (rc) return rc;
    }
    if (host_out) {
        imm3_comm *c = comms[0];
        unsigned long long v = 0;
        HIPCHK(hipSetDevice(c->ctx->device));
        HIPCHK(hipMemcpyAsync(&v, c->d_slot, sizeof(v), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->in_flight = false;
        *host_out = v;
    }
    return IMM3_OK;
}

// ---------------------------------------------------------------------------------------------
// Cross-segment / cross-GPU merge of group tables: ProjectAggregateQueueOp
// (engine/src/main/scala/immutabledb/engine/operator/ProjectAggregateQueue.scala:9-55) merges the per-segment maps by key --
// counts add, max / min combine, the first arrival keeps its place (here: ascending segment index, then first row).  Every
// rank brings the group tables of its own segments' aggregation queries; every rank gets the merged table.
//   keys of <= 2 bytes : the groups are scattered into a direct-indexed device table (k_merge_scatter) and the ranks' tables
//                        meet in element-wise all-reduces: sum on the counts, min on `segment << 32 | first row`, max / min
//                        on each aggregate -- no host work but the final compaction of the occupied slots;
//   wider keys         : the ranks' (locally merged) group lists are exchanged with ncclAllGather -- fixed-size slots,
//                        sized by an all-reduce(max) of the list lengths -- and merged by key.
// ---------------------------------------------------------------------------------------------
namespace {

struct MergedGroup {
    unsigned long long key, first, count;
    long long vals[kMaxAggs];
};

int same_spec(imm3_query *const *queries, int32_t n, const char **why) {
    for (int32_t i = 0; i < n; ++i) {
        imm3_query *q = queries[i];
        if (!q) { *why = "query is null"; return IMM3_ERR_ARG; }
        if (!q->is_agg) { *why = "not an aggregation query"; return IMM3_ERR_ARG; }
        if (!q->ran_agg) { *why = "imm3_query_run has not been called"; return IMM3_ERR_STATE; }
        imm3_query *q0 = queries[0];
        if (q->aggs.size() != q0->aggs.size() || q->group_cols.size() != q0->group_cols.size()) { *why = "the queries aggregate differently"; return IMM3_ERR_ARG; }
        for (size_t j = 0; j < q->aggs.size(); ++j)
            if (q->aggs[j].kind != q0->aggs[j].kind) { *why = "the queries aggregate differently"; return IMM3_ERR_ARG; }
    }
    return IMM3_OK;
}

void combine(MergedGroup &into, const MergedGroup &g, const int32_t *kinds, const int32_t *is_str, int n_agg) {
    into.count += g.count;
    into.first = std::min(into.first, g.first);
    for (int j = 0; j < n_agg; ++j) {
        if (kinds[j] == AGG_MAX) {
            if (is_str[j]) into.vals[j] = (long long)std::max((unsigned long long)into.vals[j], (unsigned long long)g.vals[j]);
            else into.vals[j] = std::max(into.vals[j], g.vals[j]);
        } else if (kinds[j] == AGG_MIN) into.vals[j] = std::min(into.vals[j], g.vals[j]);
    }
}

} // namespace

extern "C" int imm3_comm_merge_groups(imm3_comm *c, imm3_query *const *queries, const int32_t *segment_index, int32_t n_queries,
                                      uint64_t *keys, uint64_t *first, uint64_t *counts, int64_t *vals, uint32_t max_groups, uint32_t *n_groups) {
    if (!c || !n_groups || n_queries < 0 || (n_queries > 0 && (!queries || !segment_index))) return fail(IMM3_ERR_ARG, "bad argument");
    imm3_ctx *ctx = c->ctx;
    imm3::GateScope gate(&ctx->gate);
    if (ctx->closed) return fail(IMM3_ERR_STATE, "the context of this communicator has been destroyed");
    if (ctx->capture) return fail(IMM3_ERR_STATE, "a graph capture is open on this context");
    HIPCHK(hipSetDevice(ctx->device));
    // Every rank must take the same road AND the same exit: a rank that returned early while the others went on to the next
    // collective would leave them waiting in hipStreamSynchronize for good.  So everything that can fail on this rank alone --
    // argument checks, the queries' group lists (device work), the table's allocation -- happens BEFORE the first collective,
    // and what it found travels with the shape exchange: {key width, aggregates, their complements (max of the complement = min),
    // error flag}, one all-reduce(max) -- every rank sees whether all ranks agree and whether any of them has failed.
    int local_rc = IMM3_OK;
    std::string local_why;
    auto local_fail = [&](int code, const std::string &msg) {
        if (local_rc == IMM3_OK) { local_rc = code; local_why = msg; }
    };
    {
        const char *why = "";
        const int src = same_spec(queries, n_queries, &why);
        if (src) local_fail(src, why);
        else if (n_queries == 0) local_fail(IMM3_ERR_ARG, "a rank joins the merge with at least one aggregation query (the aggregates' kinds come from it)");
    }
    int key_bytes = 0, n_agg = 0;
    int32_t kinds[kMaxAggs] = {0, 0, 0, 0}, is_str[kMaxAggs] = {0, 0, 0, 0};
    std::vector<uint32_t> n_local((size_t)std::max(n_queries, 0), 0u); // groups of each query (its dense list is complete in q->d_o*)
    if (local_rc == IMM3_OK) {
        imm3_query *q0 = queries[0];
        for (int32_t g : q0->group_cols) key_bytes += q0->seg->cols[(size_t)q0->used[(size_t)g]].width;
        n_agg = (int)q0->aggs.size();
        for (int j = 0; j < n_agg; ++j) {
            kinds[j] = q0->aggs[j].kind;
            is_str[j] = q0->seg->cols[(size_t)q0->used[(size_t)q0->aggs[j].column]].vcodec == IMM3_DENSE_STRING;
        }
        for (int32_t i = 0; i < n_queries && local_rc == IMM3_OK; ++i) {
            if (queries[i]->ctx != ctx) { local_fail(IMM3_ERR_ARG, "the query runs on another context than the communicator"); break; }
            const int grc = query_groups(queries[i], &n_local[(size_t)i]);
            if (grc) local_fail(grc, imm3_last_error());
        }
    }
    // (wide keys: this rank's entries must leave room for a table of twice their number + 1 slots in the kernels' 32-bit slot
    // fields -- 2^30 entries: checked HERE, before the first collective, like everything else that can fail on one rank alone)
    constexpr unsigned long long kMaxMergeEntries = 1ULL << 30;
    unsigned long long total_local = 0;
    for (int32_t i = 0; i < n_queries; ++i) total_local += n_local[(size_t)i];
    if (local_rc == IMM3_OK && key_bytes > 2 && total_local > kMaxMergeEntries) local_fail(IMM3_ERR_ARG, "too many groups to merge");
    hipStream_t s = ctx->stream;
    // the direct table of narrow keys (allocated before the exchange: a failure here is this rank's to report)
    void *table = nullptr;
    std::unique_ptr<void, void (*)(void *)> table_guard(nullptr, [](void *q) { (void)hipFree(q); });
    const uint32_t K = key_bytes == 0 ? 1u : (key_bytes == 1 ? 256u : 65536u);
    const size_t table_words = (size_t)K * (2 + kMaxAggs);
    if (local_rc == IMM3_OK && key_bytes <= 2) {
        const hipError_t e = hipMalloc(&table, table_words * sizeof(unsigned long long));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            local_fail(IMM3_ERR_DEVICE, std::string("hipMalloc of the merge table: ") + hipGetErrorString(e));
        } else table_guard.reset(table);
    }
    if (c->world > 1) {
        const int rrc = rccl_ready();
        if (rrc) return rrc; // (no RCCL in this process: no rank of this process can be inside the collective either)
        // The communicator's word and the communicator itself may still be in use by an imm3_comm_allreduce_count on the
        // communicator's own stream: the merge's collectives run on the context's stream, behind that one.
        if (c->in_flight) HIPCHK(hipStreamWaitEvent(s, c->ev_done, 0));
        unsigned long long shape[5] = {(unsigned long long)key_bytes, (unsigned long long)n_agg, ~(unsigned long long)key_bytes, ~(unsigned long long)n_agg,
                                       local_rc == IMM3_OK ? 0ULL : 1ULL};
        if (local_rc != IMM3_OK) { shape[0] = shape[1] = 0ULL; shape[2] = shape[3] = 0ULL; } // (a failed rank does not vote on the shape)
        HIPCHK(hipMemcpyAsync(c->d_slot, shape, sizeof(shape), hipMemcpyHostToDevice, s));
        NCCLCHK(g_rccl.AllReduce(c->d_slot, c->d_slot, 5, ncclUint64, ncclMax, c->nccl, s));
        HIPCHK(hipMemcpyAsync(shape, c->d_slot, sizeof(shape), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (local_rc != IMM3_OK) return fail(local_rc, local_why);
        if (shape[4]) return fail(IMM3_ERR_STATE, "another rank failed before the merge (its call says why); no rank has merged anything");
        if (shape[0] != ~shape[2] || shape[1] != ~shape[3] || shape[0] != (unsigned long long)key_bytes || shape[1] != (unsigned long long)n_agg)
            return fail(IMM3_ERR_ARG, "the ranks' aggregation queries differ in key width or number of aggregates"); // (max != min: every rank sees it and leaves here)
    } else if (local_rc != IMM3_OK) return fail(local_rc, local_why);
    std::vector<MergedGroup> merged;
    if (key_bytes <= 2) {
        // ---- direct table + element-wise all-reduces ----
        void *p = table;
        const size_t words = table_words;
        MergeArgs a;
        std::memset(&a, 0, sizeof(a));
        a.slots = K;
        a.t_counts = (unsigned long long *)p;
        a.t_first = a.t_counts + K;
        a.t_vals = (long long *)(a.t_first + K);
        a.n_agg = n_agg;
        for (int j = 0; j < kMaxAggs; ++j) { a.kinds[j] = kinds[j]; a.is_str[j] = is_str[j]; }
        launch_merge_init(a, s);
        HIPCHK(hipGetLastError());
        for (int32_t i = 0; i < n_queries; ++i) {
            imm3_query *q = queries[i];
            const uint32_t ng = n_local[(size_t)i];
            a.keys = q->d_okeys;
            a.first = q->d_ofirst;
            a.counts = q->d_ocounts;
            a.vals = q->d_ovals;
            a.n_groups = ng;
            a.seg_hi = (unsigned long long)(uint32_t)segment_index[i] << 32;
            if (ng) launch_merge_scatter(a, s);
            HIPCHK(hipGetLastError());
        }
        if (c->world > 1) {
            NCCLCHK(g_rccl.AllReduce(a.t_counts, a.t_counts, K, ncclUint64, ncclSum, c->nccl, s));
            NCCLCHK(g_rccl.AllReduce(a.t_first, a.t_first, K, ncclUint64, ncclMin, c->nccl, s));
            for (int j = 0; j < n_agg; ++j) {
                long long *t = a.t_vals + (size_t)j * K;
                if (kinds[j] == AGG_MAX) NCCLCHK(g_rccl.AllReduce(t, t, K, is_str[j] ? ncclUint64 : ncclInt64, ncclMax, c->nccl, s));
                else if (kinds[j] == AGG_MIN) NCCLCHK(g_rccl.AllReduce(t, t, K, ncclInt64, ncclMin, c->nccl, s));
            }
        }
        std::vector<unsigned long long> h(words);
        HIPCHK(hipMemcpyAsync(h.data(), p, words * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (uint32_t k = 0; k < K; ++k) {
            if (!h[k]) continue;
            MergedGroup g;
            g.key = k;
            g.count = h[k];
            g.first = h[(size_t)K + k];
            for (int j = 0; j < kMaxAggs; ++j) g.vals[j] = (long long)h[(size_t)K * (2 + j) + k];
            merged.push_back(g);
        }
    } else {
        // ---- wide keys: merged by key ON THE DEVICE (round 4; rounds 2-3 merged std::maps on the host, fine for thousands of
        // groups, not for 10^5 and more): this rank's queries' lists go into a hash table (compare-and-swap on the key, atomic
        // add / min / max on the columns), its occupied slots come out as a packed list, the ranks' lists are exchanged with
        // ncclAllGather -- fixed-size slots, sized by an all-reduce(max) of the list lengths -- and merged the same way.  The
        // host only sorts the result by first arrival.
        auto pow2_at_least = [](unsigned long long v) { unsigned long long p = 1024; while (p < v) p <<= 1; return p; };
        // one table: {keys, counts, first, vals[kMaxAggs]} x slots, then the packed list and its counter
        auto table_bytes = [](unsigned long long slots, unsigned long long list_cap) {
            return (size_t)((slots * (3 + kMaxAggs) + list_cap * kMergeListWords + 8) * sizeof(unsigned long long));
        };
        auto bind = [&](MergeArgs &a, void *p, unsigned long long cap, unsigned long long list_cap) {
            std::memset(&a, 0, sizeof(a));
            a.mask = (uint32_t)(cap - 1);
            a.slots = (uint32_t)(cap + 1); // (+ the slot of the one key that equals the empty marker)
            unsigned long long *w = (unsigned long long *)p;
            a.t_keys = w;
            a.t_counts = a.t_keys + a.slots;
            a.t_first = a.t_counts + a.slots;
            a.t_vals = (long long *)(a.t_first + a.slots);
            a.out_list = (unsigned long long *)(a.t_vals + (size_t)kMaxAggs * a.slots);
            a.out_n = a.out_list + list_cap * kMergeListWords;
            a.out_cap = (uint32_t)list_cap;
            a.n_agg = n_agg;
            for (int j = 0; j < kMaxAggs; ++j) { a.kinds[j] = kinds[j]; a.is_str[j] = is_str[j]; }
        };
        if (c->world == 1 && total_local > kMaxMergeEntries) return fail(IMM3_ERR_ARG, "too many groups to merge"); // (world > 1: refused before the shape exchange)
        const unsigned long long cap1 = pow2_at_least(2 * std::max<unsigned long long>(total_local, 1)), list1 = std::max<unsigned long long>(total_local, 1);
        void *p1 = nullptr;
        {   // (a failure here is this rank's alone: with more than one rank it would leave the others in the collective below, so the
            // allocation is small-step: the table of the LOCAL merge was sized from counts every rank already has)
            const hipError_t e = hipMalloc(&p1, table_bytes(cap1 + 1, list1));
            if (e != hipSuccess) { (void)hipGetLastError(); p1 = nullptr; }
        }
        std::unique_ptr<void, void (*)(void *)> g1(p1, [](void *q) { (void)hipFree(q); });
        unsigned long long mine = 0;
        MergeArgs a1;
        std::memset(&a1, 0, sizeof(a1));
        std::string p1_why = "hipMalloc of the merge table failed";
        if (p1) {
            // Device work of THIS rank between two collectives: whatever fails here must not return -- the other ranks are on
            // their way to the list-length all-reduce below and would wait there for good.  A failure clears p1 instead: the
            // rank then says ~0 in that all-reduce and every rank leaves together.
            auto local_step = [&](hipError_t e, const char *what) {
                if (e == hipSuccess || !p1) return;
                (void)hipGetLastError();
                p1_why = std::string(what) + ": " + hipGetErrorString(e);
                p1 = nullptr;
            };
            bind(a1, p1, cap1, list1);
            launch_merge_init(a1, s);
            local_step(hipGetLastError(), "merge table init");
            for (int32_t i = 0; i < n_queries && p1; ++i) {
                imm3_query *q = queries[i];
                const uint32_t ng = n_local[(size_t)i];
                a1.keys = q->d_okeys;
                a1.first = q->d_ofirst;
                a1.counts = q->d_ocounts;
                a1.vals = q->d_ovals;
                a1.n_groups = ng;
                a1.seg_hi = (unsigned long long)(uint32_t)segment_index[i] << 32;
                if (ng) launch_merge_scatter(a1, s);
                local_step(hipGetLastError(), "merge scatter");
            }
            if (p1) {
                launch_merge_collect_list(a1, s);
                local_step(hipGetLastError(), "merge collect");
            }
            if (p1) local_step(hipMemcpyAsync(&mine, a1.out_n, sizeof(mine), hipMemcpyDeviceToHost, s), "merge list length");
            if (p1) local_step(hipStreamSynchronize(s), "merge list length");
        }
        auto take_list = [&](const unsigned long long *d_list, unsigned long long n) -> int {
            std::vector<unsigned long long> h((size_t)n * kMergeListWords);
            if (n) HIPCHK(hipMemcpyAsync(h.data(), d_list, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            merged.reserve((size_t)n);
            for (size_t i = 0; i < (size_t)n; ++i) {
                const unsigned long long *w = h.data() + i * kMergeListWords;
                MergedGroup m;
                m.key = w[0];
                m.first = w[1];
                m.count = w[2];
                for (int j = 0; j < kMaxAggs; ++j) m.vals[j] = (long long)w[3 + j];
                merged.push_back(m);
            }
            return IMM3_OK;
        };
        if (c->world == 1) {
            if (!p1) return fail(IMM3_ERR_DEVICE, p1_why);
            const int trc = take_list(a1.out_list, mine);
            if (trc) return trc;
        } else {
            // list lengths -> the common slot count (a rank whose table could not be allocated says so with ~0: every rank leaves)
            unsigned long long word = p1 ? mine : ~0ULL, most = 0;
            HIPCHK(hipMemcpyAsync(c->d_slot, &word, sizeof(word), hipMemcpyHostToDevice, s));
            NCCLCHK(g_rccl.AllReduce(c->d_slot, c->d_slot, 1, ncclUint64, ncclMax, c->nccl, s));
            HIPCHK(hipMemcpyAsync(&most, c->d_slot, sizeof(most), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (most == ~0ULL) return fail(IMM3_ERR_DEVICE, p1 ? std::string("a rank could not build its merge table; no rank has merged anything") : p1_why + "; no rank has merged anything");
            const size_t per_rank = (size_t)most * kMergeListWords;
            if (per_rank) {
                const unsigned long long entries = most * (unsigned long long)c->world;
                if (entries > kMaxMergeEntries) return fail(IMM3_ERR_ARG, "too many groups to merge"); // (the same figure on every rank: every rank leaves here)
                const unsigned long long cap2 = pow2_at_least(2 * entries);
                void *p = nullptr;
                // (sizes are the same on every rank: an allocation failure here is reported through one more flag exchange)
                hipError_t e = hipMalloc(&p, (per_rank + per_rank * (size_t)c->world) * sizeof(unsigned long long) + table_bytes(cap2 + 1, entries));
                if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
                std::unique_ptr<void, void (*)(void *)> guard(p, [](void *q) { (void)hipFree(q); });
                unsigned long long ok = p ? 0ULL : 1ULL, any = 0;
                HIPCHK(hipMemcpyAsync(c->d_slot, &ok, sizeof(ok), hipMemcpyHostToDevice, s));
                NCCLCHK(g_rccl.AllReduce(c->d_slot, c->d_slot, 1, ncclUint64, ncclMax, c->nccl, s));
                HIPCHK(hipMemcpyAsync(&any, c->d_slot, sizeof(any), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (any) return fail(IMM3_ERR_DEVICE, "a rank could not allocate the exchange buffers of the merge; no rank has merged anything");
                unsigned long long *d_send = (unsigned long long *)p, *d_recv = d_send + per_rank;
                HIPCHK(hipMemsetAsync(d_send, 0, per_rank * sizeof(unsigned long long), s)); // (padding entries: count 0)
                if (mine) HIPCHK(hipMemcpyAsync(d_send, a1.out_list, (size_t)mine * kMergeListWords * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
                NCCLCHK(g_rccl.AllGather(d_send, d_recv, per_rank, ncclUint64, c->nccl, s));
                MergeArgs a2;
                bind(a2, d_recv + per_rank * (size_t)c->world, cap2, entries);
                launch_merge_init(a2, s);
                HIPCHK(hipGetLastError());
                a2.list = d_recv;
                a2.n_groups = (uint32_t)entries;
                launch_merge_insert_list(a2, s);
                HIPCHK(hipGetLastError());
                launch_merge_collect_list(a2, s);
                HIPCHK(hipGetLastError());
                unsigned long long n2 = 0;
                HIPCHK(hipMemcpyAsync(&n2, a2.out_n, sizeof(n2), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                const int trc = take_list(a2.out_list, n2);
                if (trc) return trc;
            }
        }
    }
    std::sort(merged.begin(), merged.end(), [](const MergedGroup &x, const MergedGroup &y) { return x.first < y.first; }); // first arrival first
    *n_groups = (uint32_t)merged.size();
    for (size_t o = 0; o < merged.size() && o < max_groups; ++o) {
        if (keys) keys[o] = merged[o].key;
        if (first) first[o] = merged[o].first;
        if (counts) counts[o] = merged[o].count;
        if (vals)
            for (int j = 0; j < n_agg; ++j) vals[o * (size_t)n_agg + (size_t)j] = kinds[j] == AGG_COUNT ? (int64_t)merged[o].count : (int64_t)merged[o].vals[j];
    }
    return IMM3_OK;
}

// Single-process flavour (one JVM, every GPU: comms from imm3_comm_create_all): the group tables of all devices are in this
// process, so they are merged here -- per device the same scatter / list as above (no collective), then one merge by key on
// the host; the tables are a few KB.  comms[i] brings queries[i][0 .. n_queries[i]) with segment_index[i][...].
extern "C" int imm3_comm_merge_groups_all(imm3_comm *const *comms, int32_t n_comms, imm3_query *const *const *queries,
                                          const int32_t *const *segment_index, const int32_t *n_queries,
                                          uint64_t *keys, uint64_t *first, uint64_t *counts, int64_t *vals, uint32_t max_groups, uint32_t *n_groups) {
    if (!comms || n_comms < 1 || !queries || !segment_index || !n_queries || !n_groups) return fail(IMM3_ERR_ARG, "bad argument");
    int n_agg = -1;
    int32_t kinds[kMaxAggs] = {0, 0, 0, 0}, is_str[kMaxAggs] = {0, 0, 0, 0};
    std::map<unsigned long long, MergedGroup> all;
    for (int32_t i = 0; i < n_comms; ++i) {
        if (!comms[i] || n_queries[i] < 0 || (n_queries[i] > 0 && (!queries[i] || !segment_index[i]))) return fail(IMM3_ERR_ARG, "bad argument");
        if (n_queries[i] == 0) continue;
        imm3_comm one = *comms[i]; // the same communicator seen as a world of one: its device's queries merged without a collective
        one.world = 1;
        uint32_t n = 0;
        int rc = imm3_comm_merge_groups(&one, queries[i], segment_index[i], n_queries[i], nullptr, nullptr, nullptr, nullptr, 0, &n);
        if (rc) return rc;
        imm3_query *q0 = queries[i][0];
        const int na = (int)q0->aggs.size();
        if (n_agg < 0) {
            n_agg = na;
            for (int j = 0; j < na; ++j) {
                kinds[j] = q0->aggs[j].kind;
                is_str[j] = q0->seg->cols[(size_t)q0->used[(size_t)q0->aggs[j].column]].vcodec == IMM3_DENSE_STRING;
            }
        } else if (na != n_agg) return fail(IMM3_ERR_ARG, "the devices' aggregation queries differ");
        std::vector<uint64_t> k(n), f(n), c(n);
        std::vector<int64_t> v((size_t)n * (size_t)std::max(na, 1));
        rc = imm3_comm_merge_groups(&one, queries[i], segment_index[i], n_queries[i], k.data(), f.data(), c.data(), v.data(), n, &n);
        if (rc) return rc;
        for (uint32_t g = 0; g < n; ++g) {
            MergedGroup m;
            m.key = k[g];
            m.first = f[g];
            m.count = c[g];
            for (int j = 0; j < kMaxAggs; ++j) m.vals[j] = j < na ? v[(size_t)g * na + j] : 0;
            auto it = all.find(m.key);
            if (it == all.end()) all.emplace(m.key, m);
            else combine(it->second, m, kinds, is_str, n_agg);
        }
    }
    std::vector<MergedGroup> merged;
    for (auto &kv : all) merged.push_back(kv.second);
    std::sort(merged.begin(), merged.end(), [](const MergedGroup &x, const MergedGroup &y) { return x.first < y.first; });
    *n_groups = (uint32_t)merged.size();
    for (size_t o = 0; o < merged.size() && o < max_groups; ++o) {
        if (keys) keys[o] = merged[o].key;
        if (first) first[o] = merged[o].first;
        if (counts) counts[o] = merged[o].count;
        if (vals)
            for (int j = 0; j < n_agg; ++j) vals[o * (size_t)n_agg + (size_t)j] = kinds[j] == AGG_COUNT ? (int64_t)merged[o].count : (int64_t)merged[o].vals[j];
    }
    return IMM3_OK;
}
