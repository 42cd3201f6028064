// loopback_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the nine nccl* entry points libimm3's communicator binds
// (immutable3_amd/csrc/imm3_comm.cpp), for ranks that are THREADS of one process on ONE device.  RCCL refuses two ranks on one
// device and the development boxes have one GPU, so the world > 1 paths of imm3_comm_* -- the 5-word shape exchange of the group
// merge, its list-length and allocation flags, ncclAllGather of the packed lists, the second hash table, "every rank takes the
// same exit" -- had never run.  Loaded through IMM3_RCCL_LIB by tests/test_gpu_comm_loopback.py; never shipped, never linked.
//
// Semantics: every collective waits for the caller's stream, meets the other ranks at a barrier, rank 0 reduces / gathers through
// host memory (all ranks share the device, so every rank's pointers are valid in every thread), and a second barrier lets
// everybody go on.  Correct, slow, and deliberately simple: it is the caller's protocol that is under test, not the transport.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Group {
    int world = 0, joined = 0, arrived = 0, generation = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<const void *> send;
    std::vector<void *> recv;
    std::vector<size_t> count;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct Comm {
    std::shared_ptr<Group> group;
    int rank = 0;
};
std::mutex g_mu;
std::map<std::string, std::shared_ptr<Group>> g_groups;
unsigned long long g_next_id = 1;

size_t type_size(ncclDataType_t t) { return t == ncclUint64 || t == ncclInt64 || t == ncclFloat64 ? 8 : (t == ncclUint32 || t == ncclInt32 || t == ncclFloat32 ? 4 : 1); }

template <class T>
void reduce_into(T *acc, const T *x, size_t n, ncclRedOp_t op) {
    for (size_t i = 0; i < n; ++i) acc[i] = op == ncclSum ? (T)(acc[i] + x[i]) : (op == ncclMax ? (x[i] > acc[i] ? x[i] : acc[i]) : (x[i] < acc[i] ? x[i] : acc[i]));
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof(*id));
    const unsigned long long v = g_next_id++;
    std::memcpy(id->internal, "imm3loop", 8);
    std::memcpy(id->internal + 8, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &slot = g_groups[std::string(id.internal, id.internal + 16)];
        if (!slot) {
            slot = std::make_shared<Group>();
            slot->world = nranks;
            slot->send.assign((size_t)nranks, nullptr);
            slot->recv.assign((size_t)nranks, nullptr);
            slot->count.assign((size_t)nranks, 0);
        }
        g = slot;
    }
    if (g->world != nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->group = g;
    c->rank = rank;
    *comm = (ncclComm_t)c;
    g->barrier(); // (every rank has joined, as ncclCommInitRank guarantees on return)
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; } // (one communicator per device: not a loopback shape)

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete (Comm *)comm;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    Group &g = *c->group;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    g.send[(size_t)c->rank] = sendbuff;
    g.recv[(size_t)c->rank] = recvbuff;
    g.count[(size_t)c->rank] = count;
    g.barrier();
    ncclResult_t rc = ncclSuccess;
    if (c->rank == 0) {
        const size_t bytes = count * type_size(datatype);
        std::vector<unsigned char> acc(bytes), x(bytes);
        for (int r = 0; r < g.world && rc == ncclSuccess; ++r) {
            if (g.count[(size_t)r] != count) { rc = ncclInvalidArgument; break; }
            if (hipMemcpy(r == 0 ? acc.data() : x.data(), g.send[(size_t)r], bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
            if (r == 0) continue;
            if (datatype == ncclUint64) reduce_into((unsigned long long *)acc.data(), (const unsigned long long *)x.data(), count, op);
            else if (datatype == ncclInt64) reduce_into((long long *)acc.data(), (const long long *)x.data(), count, op);
            else rc = ncclInvalidArgument;
        }
        for (int r = 0; r < g.world && rc == ncclSuccess; ++r)
            if (hipMemcpy(g.recv[(size_t)r], acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    g.barrier();
    return rc;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    Group &g = *c->group;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    g.send[(size_t)c->rank] = sendbuff;
    g.recv[(size_t)c->rank] = recvbuff;
    g.count[(size_t)c->rank] = sendcount;
    g.barrier();
    ncclResult_t rc = ncclSuccess;
    if (c->rank == 0) {
        const size_t bytes = sendcount * type_size(datatype);
        for (int r = 0; r < g.world && rc == ncclSuccess; ++r)       // destination rank
            for (int q = 0; q < g.world && rc == ncclSuccess; ++q)   // source rank
                if (bytes && hipMemcpy((unsigned char *)g.recv[(size_t)r] + (size_t)q * bytes, g.send[(size_t)q], bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        if (hipDeviceSynchronize() != hipSuccess) rc = ncclUnhandledCudaError;
    }
    g.barrier();
    return rc;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "loopback transport error"; }

} // extern "C"
