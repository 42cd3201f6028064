// imm3_sync.h -- the host-side concurrency pieces of the C ABI, free of any HIP dependency so that they build (and are
// exercised under -fsanitize=thread) on a machine without a GPU: tests/native/tsan_sync.cpp.
//
// Threading contract of include/imm3.h (the reference calls the path from a FixedThreadPool(cpuCount), one PipelineThread
// per segment: engine/src/main/scala/immutabledb/engine/Engine.scala:176-180,247-262; SqlCli.scala:64):
//   * every entry point may be called from any thread;
//   * a CONTEXT may be shared by any number of threads: its buffer pool, its timing / stamp records, its graph list and
//     its lazily made streams are guarded here; work of all threads lands on the context's one stream in call order;
//   * SEGMENTS and TABLES are immutable once created (the lazy decode of a compressed column is guarded by the segment)
//     and may be read by any number of queries, threads and contexts of the same device;
//   * ONE query / graph / comm handle is used by one thread at a time (a PipelineThread owns its iterator chain);
//   * a graph capture is EXCLUSIVE: between imm3_ctx_capture_begin and _end (same thread) calls of OTHER threads on that
//     context wait -- whatever they enqueued would otherwise be recorded into the graph.
// Worker threads that want their kernels to overlap on the device take a context each (a context = a stream) over the
// shared segments.
#pragma once

#include <atomic>
#include <cstddef>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace imm3 {

// ---------------------------------------------------------------------------------------------
// Caching allocator for per-query device buffers.  Every user of such a buffer runs on the owning context's stream, so
// a block freed by one query and handed to the next is reused in stream order: no synchronisation, no hipMalloc /
// hipFree (each ~50-100 us) on the query path once the pool is warm.  Backend: int alloc(void **, size_t) (0 = ok),
// void free(void *).
// ---------------------------------------------------------------------------------------------
inline size_t pool_bucket(size_t bytes) {
    if (bytes < 256) bytes = 256;
    if (bytes <= (1u << 20)) {
        size_t b = 256;
        while (b < bytes) b <<= 1;
        return b;
    }
    return (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
}

template <class Backend>
class BlockPool {
  public:
    static constexpr size_t kMaxCached = (size_t)16 << 30; // keep at most 16 GiB parked

    // 0 = ok, else the backend's error code
    int alloc(void **out, size_t bytes) {
        const size_t b = pool_bucket(bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = free_.find(b);
            if (it != free_.end()) {
                *out = it->second;
                free_.erase(it);
                cached_ -= b;
                return 0;
            }
        }
        void *p = nullptr;
        int e = Backend::alloc(&p, b);
        if (e != 0) { // give cached blocks back to the driver and retry once
            std::vector<void *> victims;
            {
                std::lock_guard<std::mutex> g(mu_);
                for (auto &kv : free_) {
                    size_.erase(kv.second);
                    victims.push_back(kv.second);
                }
                free_.clear();
                cached_ = 0;
            }
            for (void *v : victims) Backend::free(v);
            e = Backend::alloc(&p, b);
            if (e != 0) return e;
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            size_[p] = b;
        }
        *out = p;
        return 0;
    }

    void release(void *p) {
        if (!p) return;
        bool drop = false;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = size_.find(p);
            if (it == size_.end()) drop = true; // not ours (or the pool was drained under it)
            else if (cached_ + it->second > kMaxCached) {
                size_.erase(it);
                drop = true;
            } else {
                free_.emplace(it->second, p);
                cached_ += it->second;
            }
        }
        if (drop) Backend::free(p);
    }

    // frees every parked block and forgets the live ones (they are then freed directly when released)
    void drain() {
        std::vector<void *> victims;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto &kv : free_) victims.push_back(kv.second);
            free_.clear();
            size_.clear();
            cached_ = 0;
        }
        for (void *v : victims) Backend::free(v);
    }

    size_t cached_bytes() {
        std::lock_guard<std::mutex> g(mu_);
        return cached_;
    }

  private:
    std::mutex mu_;
    std::multimap<size_t, void *> free_;
    std::unordered_map<void *, size_t> size_;
    size_t cached_ = 0;
};

// ---------------------------------------------------------------------------------------------
// The capture gate of a context.  Ordinary calls pass it shared; imm3_ctx_capture_begin takes it exclusively and keeps it
// until imm3_ctx_capture_end (same thread), so that no other thread can enqueue on the capturing stream meanwhile.  The
// owner's own calls pass freely.  Re-entrant per thread (an entry point that calls another one on the same context).
// ---------------------------------------------------------------------------------------------
class CaptureGate {
  public:
    bool owned_by_me() const { return owner_.load(std::memory_order_acquire) == std::this_thread::get_id(); }
    bool capturing() const { return owner_.load(std::memory_order_acquire) != std::thread::id(); }

    void enter() {
        if (owned_by_me()) return;
        if (depth_of(this)++ == 0) rw_.lock_shared();
    }
    void leave() {
        if (owned_by_me()) return;
        if (--depth_of(this) == 0) {
            rw_.unlock_shared();
            forget(this);
        }
    }
    // false: this thread is inside another call on the same context (it would wait for itself)
    bool begin_exclusive() {
        if (owned_by_me() || held_by_me(this)) return false;
        rw_.lock();
        owner_.store(std::this_thread::get_id(), std::memory_order_release);
        return true;
    }
    void end_exclusive() {
        owner_.store(std::thread::id(), std::memory_order_release);
        rw_.unlock();
    }

  private:
    struct Held {
        const CaptureGate *gate;
        int depth;
    };
    static std::vector<Held> &held() {
        static thread_local std::vector<Held> h;
        return h;
    }
    static int &depth_of(const CaptureGate *g) {
        auto &h = held();
        for (auto &e : h)
            if (e.gate == g) return e.depth;
        h.push_back(Held{g, 0});
        return h.back().depth;
    }
    static bool held_by_me(const CaptureGate *g) {
        for (auto &e : held())
            if (e.gate == g && e.depth > 0) return true;
        return false;
    }
    static void forget(const CaptureGate *g) {
        auto &h = held();
        for (size_t i = 0; i < h.size(); ++i)
            if (h[i].gate == g) {
                h[i] = h.back();
                h.pop_back();
                return;
            }
    }
    std::shared_mutex rw_;
    std::atomic<std::thread::id> owner_{};
};

struct GateScope { // RAII for one entry point
    CaptureGate *g;
    explicit GateScope(CaptureGate *gate) : g(gate) { g->enter(); }
    ~GateScope() { g->leave(); }
    GateScope(const GateScope &) = delete;
    GateScope &operator=(const GateScope &) = delete;
};

// ---------------------------------------------------------------------------------------------
// Reference counts of the handles (imm3_handles.h, "Lifetimes").
// ---------------------------------------------------------------------------------------------
inline void ref_retain(std::atomic<int> &refs) { refs.fetch_add(1, std::memory_order_relaxed); }
// true: that was the last reference (the caller frees the object)
inline bool ref_release(std::atomic<int> &refs) { return refs.fetch_sub(1, std::memory_order_acq_rel) == 1; }

// ---------------------------------------------------------------------------------------------
// Slots handed out to concurrent launches (timing event pairs, device-clock stamp slots): claim() gives each caller a
// distinct index below the capacity, or -1.
// ---------------------------------------------------------------------------------------------
class SlotCounter {
  public:
    void reset(size_t capacity) {
        cap_.store(capacity, std::memory_order_relaxed);
        used_.store(0, std::memory_order_release);
    }
    long claim() {
        size_t u = used_.load(std::memory_order_relaxed);
        while (u < cap_.load(std::memory_order_relaxed))
            if (used_.compare_exchange_weak(u, u + 1, std::memory_order_acq_rel)) return (long)u;
        return -1;
    }
    size_t used() const { return used_.load(std::memory_order_acquire); }
    void rewind() { used_.store(0, std::memory_order_release); }

  private:
    std::atomic<size_t> used_{0}, cap_{0};
};

} // namespace imm3
