// png_wave.h -- "one wave = one stream" helpers shared by the PNG kernels.
// Product build: a 64-lane wave; wave-uniform values live in every lane, per-lane values are plain variables.
// Emulation build (tests only): ONE host thread plays the whole wave; per-lane values are 64-element arrays (LV<T>) and
// LFOR(j) loops over the lanes.  The logic between the two is shared line for line.
#pragma once
#include "gpu_rt.h"

namespace csp {

#ifdef CSH_EMUL
#define LFOR(j) for (int j = 0; j < 64; j++)
template <class T> struct LV { T v[64]; __device__ T &operator[](int j) { return v[j]; } __device__ const T &operator[](int j) const { return v[j]; } };
__device__ __forceinline__ static uint32_t uni(uint32_t x) { return x; }
#define CSP_WAVE_THREADS 1
#define CSP_WAVE_SYNC() ((void)0)
#define CSP_MEM_FENCE() ((void)0)
template <class T> __device__ __forceinline__ static T coherent_load(const T *p) { return *p; }
template <class T> __device__ __forceinline__ static void coherent_store(T *p, T v) { *p = v; }
#define CSP_ACQUIRE_FENCE() ((void)0)
#else
#define LFOR(j) for (int j = int(threadIdx.x & 63u), once_ = 1; once_; once_ = 0)
template <class T> struct LV { T v; __device__ T &operator[](int) { return v; } __device__ const T &operator[](int) const { return v; } };
__device__ __forceinline__ static uint32_t uni(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }
#define CSP_WAVE_THREADS 64
// LDS traffic between the lanes of one wave: order it (the wave runs in lockstep, only the compiler and the LDS queue need telling)
#define CSP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// this wave's global stores are complete and visible to its own later (cache-bypassing) loads
#define CSP_MEM_FENCE() __threadfence()
template <class T> __device__ __forceinline__ static T coherent_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ static void coherent_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// later loads of this wave see what another wave published before the value just read with coherent_load
#define CSP_ACQUIRE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#endif

template <class F>
__device__ __forceinline__ static uint64_t lballot(F pred) {
#ifdef CSH_EMUL
    uint64_t m = 0;
    for (int j = 0; j < 64; j++) m |= uint64_t(pred(j) ? 1 : 0) << j;
    return m;
#else
    return __ballot(pred(int(threadIdx.x & 63u)));
#endif
}
__device__ __forceinline__ static uint64_t lanes_below(int j) { return (1ull << j) - 1ull; }

// sum over the lanes (every lane gets it)
__device__ __forceinline__ static uint64_t lsum(const LV<uint64_t> &x) {
#ifdef CSH_EMUL
    uint64_t s = 0;
    for (int j = 0; j < 64; j++) s += x.v[j];
    return s;
#else
    uint64_t v = x.v;
    CSH_UNROLL
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(v)), o, 64)), hi = uint32_t(__shfl_xor(int(uint32_t(v >> 32)), o, 64));
        v += (uint64_t(hi) << 32) | lo;
    }
    return v;
#endif
}
// exclusive prefix sum over the lanes; total through `sum`
__device__ __forceinline__ static LV<uint32_t> lscan(const LV<uint32_t> &x, uint32_t &sum) {
    LV<uint32_t> r;
#ifdef CSH_EMUL
    uint32_t s = 0;
    for (int j = 0; j < 64; j++) { r.v[j] = s; s += x.v[j]; }
    sum = s;
#else
    // inclusive scan on the VALU's cross-lane path (row shifts, then the two row broadcasts): no LDS round trips
    uint32_t v = x.v;
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));
    sum = uint32_t(__builtin_amdgcn_readlane(int(v), 63));   // callers run it with all 64 lanes active
    r.v = v - x.v;
#endif
    return r;
}

// ---- wave-uniform LSB-first bit reader (deflate order) over bytes [base, base+len); base is 4-byte aligned
struct LeReader {
    const uint8_t *base;
    uint32_t len, wbase;
    LV<uint32_t> win, nxt;   // lane l: little-endian words wbase + l and wbase + 64 + l
    uint64_t acc;
    int nb;
    uint32_t wi;             // next word to append
    uint64_t consumed;       // bits taken so far
    __device__ __forceinline__ uint32_t loadw(uint32_t w) const {
        uint32_t b = w * 4;
        if (b + 4 <= len) return *reinterpret_cast<const uint32_t *>(base + b);
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) if (b + i < len) v |= uint32_t(base[b + i]) << (8 * i);
        return v;
    }
    __device__ __forceinline__ uint32_t word(uint32_t i) {
        if (i >= wbase + 64) { wbase += 64; LFOR(l) { win[l] = nxt[l]; nxt[l] = loadw(wbase + 64 + l); } }
#ifdef CSH_EMUL
        if (i < wbase || i - wbase >= 64) { fprintf(stderr, "LeReader: non-sequential access\n"); abort(); }
        return win.v[i - wbase];
#else
        return uint32_t(__builtin_amdgcn_readlane(int(win.v), int(i - wbase)));
#endif
    }
    // start reading at byte `at`
    __device__ __forceinline__ void seek(uint32_t at) {
        wbase = at >> 2;
        LFOR(l) { win[l] = loadw(wbase + l); nxt[l] = loadw(wbase + 64 + l); }
        acc = uint64_t(word(wbase)) | (uint64_t(word(wbase + 1)) << 32);
        nb = 64; wi = wbase + 2; consumed = uint64_t(at & ~3u) * 8u;
        skip(int(at & 3u) * 8);
    }
    __device__ __forceinline__ void begin(const uint8_t *p, uint32_t n, uint32_t at) { base = p; len = n; seek(at); }
    __device__ __forceinline__ bool overrun() const { return consumed > uint64_t(len) * 8u; }
    __device__ __forceinline__ uint32_t peek(int n) const { return uint32_t(acc) & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u)); }   // n <= 32
    __device__ __forceinline__ void skip(int n) {   // n <= 32
        acc >>= n; nb -= n; consumed += uint32_t(n);
        if (nb <= 32) { acc |= uint64_t(word(wi)) << nb; wi++; nb += 32; }
    }
    __device__ __forceinline__ uint32_t get(int n) { uint32_t v = peek(n); skip(n); return v; }
    __device__ __forceinline__ uint32_t byte_pos() const { return uint32_t(consumed >> 3); }
};

}  // namespace csp
