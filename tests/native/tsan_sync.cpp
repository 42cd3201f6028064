// tsan_sync.cpp -- the host-side concurrency pieces of the C ABI (immutable3_amd/csrc/imm3_sync.h: buffer pool, capture
// gate, handle reference counts, slot counter) driven by 8 threads.  Built by tests/test_host_threads.py with
// -fsanitize=thread (and once without): needs no GPU, the pool's backend is malloc / free.
// Mirrors how the reference drives the path: FixedThreadPool(cpuCount), one PipelineThread per segment
// (engine/src/main/scala/immutabledb/engine/Engine.scala:176-180,247-262).
#include "../../immutable3_amd/csrc/imm3_sync.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

using namespace imm3;

static std::atomic<long> g_live_blocks{0};
struct MallocBackend {
    static int alloc(void **p, size_t n) {
        *p = std::malloc(n);
        if (!*p) return 2;
        g_live_blocks.fetch_add(1);
        return 0;
    }
    static void free(void *p) {
        g_live_blocks.fetch_sub(1);
        std::free(p);
    }
};

static int g_fail = 0;
#define CHECK(c)                                                          \
    do {                                                                  \
        if (!(c)) {                                                       \
            std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            __atomic_store_n(&g_fail, 1, __ATOMIC_SEQ_CST);               \
        }                                                                 \
    } while (0)

// (1) the pool: a block is owned by exactly one thread between alloc and release
static void pool_worker(BlockPool<MallocBackend> *pool, int id, int iters) {
    std::mt19937 rng(1234u + (unsigned)id);
    std::vector<std::pair<unsigned char *, size_t>> mine;
    for (int i = 0; i < iters; ++i) {
        if (mine.size() < 8 && (rng() & 1)) {
            const size_t sizes[] = {64, 300, 4096, 70000, 1u << 20, (1u << 20) + 5};
            const size_t n = sizes[rng() % 6];
            void *p = nullptr;
            CHECK(pool->alloc(&p, n) == 0);
            std::memset(p, id + 1, n < 512 ? n : 512);
            mine.emplace_back((unsigned char *)p, n);
        } else if (!mine.empty()) {
            const size_t k = rng() % mine.size();
            const size_t n = mine[k].second < 512 ? mine[k].second : 512;
            for (size_t b = 0; b < n; ++b) CHECK(mine[k].first[b] == (unsigned char)(id + 1)); // nobody else wrote into it
            pool->release(mine[k].first);
            mine[k] = mine.back();
            mine.pop_back();
        }
    }
    for (auto &m : mine) pool->release(m.first);
}

// (2) the capture gate: while one thread holds it exclusively no other thread is inside a call
static std::atomic<int> g_inside{0};
static std::atomic<bool> g_stop{false};
static void gate_caller(CaptureGate *gate, int iters) {
    for (int i = 0; i < iters && !g_stop.load(); ++i) {
        GateScope outer(gate);
        g_inside.fetch_add(1);
        {
            GateScope nested(gate); // an entry point that calls another one on the same context
            g_inside.fetch_add(1);
            g_inside.fetch_sub(1);
        }
        g_inside.fetch_sub(1);
    }
}
static void gate_capturer(CaptureGate *gate, int iters) {
    for (int i = 0; i < iters; ++i) {
        CHECK(gate->begin_exclusive());
        CHECK(gate->owned_by_me() && gate->capturing());
        CHECK(g_inside.load() == 0); // everybody else waits at the gate
        {
            GateScope own(gate); // the capturing thread's own calls pass
            CHECK(g_inside.load() == 0);
        }
        CHECK(!gate->begin_exclusive()); // not twice
        gate->end_exclusive();
        std::this_thread::yield();
    }
}

// (3) reference counts: the object goes exactly once, with the last reference
struct Obj {
    std::atomic<int> refs{1};
    int payload = 42;
};
static std::atomic<int> g_deleted{0};
static void ref_worker(Obj *o, int iters) {
    for (int i = 0; i < iters; ++i) {
        ref_retain(o->refs);
        CHECK(o->payload == 42);
        if (ref_release(o->refs)) {
            g_deleted.fetch_add(1);
            delete o;
        }
    }
    if (ref_release(o->refs)) { // this thread's own reference (taken by main before the thread started)
        g_deleted.fetch_add(1);
        delete o;
    }
}

// (4) slot counter: distinct indices below the capacity
static void slot_worker(SlotCounter *sc, std::vector<std::atomic<int>> *hits) {
    for (;;) {
        const long k = sc->claim();
        if (k < 0) return;
        (*hits)[(size_t)k].fetch_add(1);
    }
}

#ifdef PLANT_RACE
static int g_racy = 0; // proves the sanitizer is live: tests/test_host_threads.py expects a data-race report from this build
static void racy_worker() {
    for (int i = 0; i < 100000; ++i) g_racy = g_racy + 1;
}
#endif

int main(int argc, char **argv) {
    const int T = 8;
#ifdef PLANT_RACE
    {
        std::thread a(racy_worker), b(racy_worker);
        a.join();
        b.join();
        std::printf("racy %d\n", g_racy);
    }
#endif
    const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
    {
        BlockPool<MallocBackend> pool;
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(pool_worker, &pool, t, iters);
        for (auto &t : th) t.join();
        CHECK(pool.cached_bytes() > 0);
        pool.drain();
        CHECK(pool.cached_bytes() == 0);
        CHECK(g_live_blocks.load() == 0); // everything handed out came back and was freed exactly once
    }
    {
        CaptureGate gate;
        std::vector<std::thread> th;
        for (int t = 0; t < T - 1; ++t) th.emplace_back(gate_caller, &gate, iters * 4);
        std::thread cap(gate_capturer, &gate, 200);
        cap.join();
        g_stop.store(true);
        for (auto &t : th) t.join();
        CHECK(g_inside.load() == 0);
    }
    {
        Obj *o = new Obj();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) {
            ref_retain(o->refs);
            th.emplace_back(ref_worker, o, iters);
        }
        if (ref_release(o->refs)) { // main's reference
            g_deleted.fetch_add(1);
            delete o;
        }
        for (auto &t : th) t.join();
        CHECK(g_deleted.load() == 1);
    }
    {
        SlotCounter sc;
        sc.reset(10000);
        std::vector<std::atomic<int>> hits(10000);
        for (auto &h : hits) h.store(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(slot_worker, &sc, &hits);
        for (auto &t : th) t.join();
        for (auto &h : hits) CHECK(h.load() == 1);
        CHECK(sc.used() == 10000);
    }
    if (g_fail) return 1;
    std::puts("ok");
    return 0;
}
