// gpu_rt.h -- the one place that knows whether this translation unit is the product build
// (hipcc, gfx950) or the CSH_EMUL logic-test build.
//
// Product build: plain HIP runtime.  Kernels are launched with CSH_LAUNCH on the batch's stream.
//
// CSH_EMUL build (g++, tests only -- see tests/emul/README.md): the SAME kernel sources are compiled
// as ordinary C++ and a launch becomes a sequential loop over the grid.  It exists so the bit-level
// kernel logic can be regression-tested in the GPU-less authoring container; it is never linked
// into libcaesium_hip.so and the product library has no CPU fallback.  Kernels are written so that
// this is legal: no thread reads what another thread of the same launch wrote (only atomics cross
// threads), and no kernel needs a barrier.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef CSH_EMUL
#include <hip/hip_runtime.h>
#define CSH_LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__)
#define CSH_UNROLL _Pragma("unroll")
// pin a 32-bit value in a VGPR here: a load feeding it cannot be sunk into a later branch
#define CSH_PIN(x) asm volatile("" : "+v"(x))
#define CSH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)  // keep the instruction scheduler from interleaving stages (register pressure)
// Kernels that need workgroup barriers are written as a loop over "phases":
//     CSH_SHARED int lds[...];
//     CSH_PHASE_LOOP(3) { if (phase == 0) {...; continue;} ... }
// On the GPU the loop's increment is the __syncthreads(), so `continue` is the only legal early exit of a phase
// (never `return`: every lane must reach every barrier).  The emulation runs phase p for ALL lanes of a workgroup
// before phase p+1, which is exactly the ordering the barrier gives.
#define CSH_SHARED __shared__
// per-lane state that lives across the phases of a phased kernel: plain registers here (the phase loop is one function body); the
// emulation re-enters the kernel once per phase and lane, so there it is a per-thread array indexed by the lane.  Declare it in front
// of CSH_PHASE_LOOP.
#define CSH_PERSIST(T, name, N) T name[N]
// a value every lane of the wave holds alike, moved to a scalar register (and so waited for HERE, not lazily inside a later branch)
#define CSH_UNIFORM(x) __builtin_amdgcn_readfirstlane(int(x))
#define CSH_PHASE_LOOP(NPH) for (int phase = 0; phase < (NPH); ((phase + 1 < (NPH)) ? __syncthreads() : (void)0), phase++)
#define CSH_LAUNCH_PHASED(kern, nph, grid, block, stream, ...) hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__)
// the same loop where some phase ends only synchronise a wave with itself: bit p of WAVE_MASK set = after phase p the lanes of a wave
// see each other's LDS writes (no s_barrier -- the kernel's waves must then not exchange data at that point)
#define CSH_PHASE_LOOP_MIXED(NPH, WAVE_MASK)                                                                                              \
    for (int phase = 0; phase < (NPH);                                                                                                    \
         ((phase + 1 < (NPH)) ? ((((WAVE_MASK) >> phase) & 1u) ? (__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"), __builtin_amdgcn_wave_barrier()) : __syncthreads()) : (void)0), phase++)
#else
// ------------------------------------------------------------------ emulation shims
#include <algorithm>
#include <chrono>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define CSH_UNROLL
#define CSH_PIN(x) ((void)0)
#define CSH_UNIFORM(x) int(x)
#define CSH_SCHED_FENCE() ((void)0)
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern int csh_emul_reverse;  // tests flip the (arbitrary) execution order to shake out order dependence
template <class F, class... A>
static inline void csh_emul_launch(F kern, dim3 grid, dim3 block, A... args) {
    gridDim = grid; blockDim = block;
    const bool rev = csh_emul_reverse != 0;
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bxi = 0; bxi < grid.x; bxi++) {
        unsigned bx = rev ? grid.x - 1 - bxi : bxi;
        blockIdx = dim3(bx, by, bz);
        for (unsigned tz = 0; tz < block.z; tz++) for (unsigned ty = 0; ty < block.y; ty++) for (unsigned txi = 0; txi < block.x; txi++) {
            threadIdx = dim3(rev ? block.x - 1 - txi : txi, ty, tz);
            kern(args...);
        }
    }
}
#define CSH_LAUNCH(kern, grid, block, stream, ...) csh_emul_launch(kern, dim3(grid), dim3(block), __VA_ARGS__)
#define CSH_SHARED static thread_local   // one copy per host thread: the emulated kernels of concurrent batches must not share "LDS"
#define CSH_PERSIST(T, name, N) static thread_local T name##_lanes_[1024][N]; T (&name)[N] = name##_lanes_[threadIdx.x]
extern thread_local int csh_emul_phase;
#define CSH_PHASE_LOOP(NPH) for (int phase = csh_emul_phase, once_ = 1; once_; once_ = 0)
#define CSH_PHASE_LOOP_MIXED(NPH, WAVE_MASK) CSH_PHASE_LOOP(NPH)
template <class F, class... A>
static inline void csh_emul_launch_phased(F kern, int nph, dim3 grid, dim3 block, A... args) {
    gridDim = grid; blockDim = block;
    const bool rev = csh_emul_reverse != 0;
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bxi = 0; bxi < grid.x; bxi++) {
        blockIdx = dim3(rev ? grid.x - 1 - bxi : bxi, by, bz);
        for (int ph = 0; ph < nph; ph++) {
            csh_emul_phase = ph;
            for (unsigned tz = 0; tz < block.z; tz++) for (unsigned ty = 0; ty < block.y; ty++) for (unsigned txi = 0; txi < block.x; txi++) {
                threadIdx = dim3(rev ? block.x - 1 - txi : txi, ty, tz);
                kern(args...);
            }
        }
    }
    csh_emul_phase = 0;
}
#define CSH_LAUNCH_PHASED(kern, nph, grid, block, stream, ...) csh_emul_launch_phased(kern, nph, dim3(grid), dim3(block), __VA_ARGS__)
typedef int hipError_t;
typedef int hipStream_t;
struct csh_emul_event { std::chrono::steady_clock::time_point t; };
typedef csh_emul_event *hipEvent_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline const char *hipGetErrorString(hipError_t) { return "emul"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 1; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? 0 : 1; }
static inline hipError_t hipHostFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return 0; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 0; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new csh_emul_event; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return 0;
}
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline float __fmul_rn(float a, float b) { return a * b; }   // the emulation is built with -ffp-contract=off
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline int __mul24(int a, int b) { return (int)((int64_t)((a << 8) >> 8) * ((b << 8) >> 8)); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
using std::max;
using std::min;
#endif

#define CSH_CHECK(expr)                                                                                         \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) { csh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } \
    } while (0)

void csh_set_error(const char *fmt, ...);

// A copy the host waits for, ordered on the batch's OWN stream.  A plain hipMemcpy goes through the legacy default stream: it waits for every
// kernel of every other batch of the process (the second worker's, another file type's), and holds their next launches up until it is done.
// The batches' streams are created hipStreamNonBlocking for the same reason; nothing in this library uses the default stream.
template <class Kind>
static inline hipError_t csh_copy_wait(void *dst, const void *src, size_t n, Kind kind, hipStream_t st) {
    hipError_t e = hipMemcpyAsync(dst, src, n, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
