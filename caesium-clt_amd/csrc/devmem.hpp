// devmem.hpp -- device / pinned-host memory helpers shared by the batch pipelines (pipeline.cpp, png_pipeline.cpp)
#pragma once
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "gpu_rt.h"

namespace csh {

// ---- memory caches.  A batch of the same shape follows almost every batch (the CLI feeds groups of 1024 files), and
// hipMalloc / hipFree of ~20 GB of pools plus the pageable-memory copies cost ten times what the kernels do.  Freed device
// blocks and pinned host blocks are therefore kept (per device / process-wide) and handed to the next batch that fits.
struct BlockCache {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;   // capacity -> block
    size_t cached = 0, limit;
    bool pinned;
    explicit BlockCache(size_t lim, bool pin) : limit(lim), pinned(pin) {}
    static size_t round_up(size_t bytes) { size_t g = bytes < (1u << 20) ? 4096 : (2u << 20); return (bytes + g - 1) / g * g; }
    void *get(size_t bytes, size_t &cap) {
        bytes = round_up(bytes ? bytes : 1);
        {
            std::lock_guard<std::mutex> l(mu);
            auto it = free_blocks.lower_bound(bytes);
            if (it != free_blocks.end() && it->first <= 2 * bytes + (64u << 20)) {
                void *q = it->second; cap = it->first; cached -= cap; free_blocks.erase(it);
                return q;
            }
        }
        void *q = nullptr;
        hipError_t e = pinned ? hipHostMalloc(&q, bytes) : hipMalloc(&q, bytes);
        if (e != hipSuccess) {   // out of memory: drop everything cached and try once more
            trim(0);
            e = pinned ? hipHostMalloc(&q, bytes) : hipMalloc(&q, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        cap = bytes;
        return q;
    }
    void put(void *q, size_t cap) {
        std::lock_guard<std::mutex> l(mu);
        free_blocks.emplace(cap, q); cached += cap;
        if (cached > limit) trim_locked(limit / 2);
    }
    void trim(size_t keep) { std::lock_guard<std::mutex> l(mu); trim_locked(keep); }
    void trim_locked(size_t keep) {
        while (cached > keep && !free_blocks.empty()) {
            auto it = std::prev(free_blocks.end());
            if (pinned) (void)hipHostFree(it->second); else (void)hipFree(it->second);
            cached -= it->first; free_blocks.erase(it);
        }
    }
};
inline BlockCache &device_cache(int dev) {
    static BlockCache *caches[64] = {nullptr};
    static std::mutex mu;
    std::lock_guard<std::mutex> l(mu);
    dev = dev < 0 ? 0 : dev & 63;
    if (!caches[dev]) caches[dev] = new BlockCache(size_t(160) << 30, false);   // of the 288 GB of HBM
    return *caches[dev];
}
inline BlockCache &pinned_cache() { static BlockCache c(size_t(16) << 30, true); return c; }

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    int dev = 0;
    ~DevBuf() { release(); }
    void release() { if (p) device_cache(dev).put(p, cap); p = nullptr; n = 0; cap = 0; }
    int alloc(size_t count) {
        if (getenv("CSH_TRACE_ALLOC") && count * sizeof(T) >= (32u << 20)) fprintf(stderr, "[alloc] %.1f MB (%zu x %zu)\n", double(count * sizeof(T)) / 1048576.0, count, sizeof(T));   // what a batch holds, buffer by buffer
        release();
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        void *q = device_cache(dev).get((count ? count : 1) * sizeof(T), cap);
        if (!q) { csh_set_error("out of device memory"); return -1; }
        n = count;
        p = static_cast<T *>(q);
        return 0;
    }
    int upload(const std::vector<T> &v, hipStream_t st) {
        if (alloc(v.size())) return -1;
        if (!v.empty()) CSH_CHECK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
        return 0;
    }
    int zero(hipStream_t st) { if (n) CSH_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), st)); return 0; }
};

// growable byte pool in pinned host memory (the entropy-coded segments of a batch: uploaded by DMA straight from here)
struct PinnedBytes {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    ~PinnedBytes() { if (p) pinned_cache().put(p, cap); }
    size_t size() const { return n; }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        size_t ncap = 0;
        void *q = pinned_cache().get(std::max(want, cap * 2), ncap);
        if (!q) return false;
        if (p) pinned_cache().put(p, cap);   // nothing to move: the data is copied in by flush_copies()
        p = static_cast<uint8_t *>(q); cap = ncap;
        return true;
    }
    // data, then zero padding to a multiple of 64.  The bytes are copied later, by flush_copies(): the pool of a batch is
    // ~0.6 MB per file and one thread's memcpy would be most of the batch set-up time
    struct Copy { size_t dst; const uint8_t *src; size_t len, pad; };
    std::vector<Copy> pending;
    bool append_aligned(const uint8_t *src, size_t len) {
        size_t end = (n + len + 63) & ~size_t(63);
        if (!reserve(end)) return false;
        pending.push_back({n, src, len, end - n - len});
        n = end;
        return true;
    }
    void flush_copies() {
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next++) < pending.size();) {
                const Copy &c = pending[i];
                memcpy(p + c.dst, c.src, c.len);
                memset(p + c.dst + c.len, 0, c.pad);
            }
        };
        size_t bytes = n, nthreads = std::min<size_t>(std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())), bytes / (8u << 20) + 1);
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nthreads; t++) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        pending.clear();
    }
};

}  // namespace csh
