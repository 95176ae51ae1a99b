// wave.h -- wave-level helpers for kernels written once for both builds (png_wave.h holds the basic set: LV<T> per-lane values,
// LFOR(l) "for every lane", lballot, lscan).  Product build: a 64-lane wave, cross-lane traffic on the VALU's DPP path.  Emulation build
// (tests only): one host thread plays the whole wave, LV<T> is a 64-element array.  The kernel logic between the two is shared line for line.
#pragma once
#include "png_wave.h"

namespace csh {

using csp::LV;
using csp::lballot;
using csp::lanes_below;
using csp::lscan;
using csp::uni;
using csp::coherent_load;
using csp::coherent_store;

// lane l gets the value of lane l - 1; lane 0 gets `carry`
__device__ __forceinline__ static LV<uint32_t> lprev(const LV<uint32_t> &x, uint32_t carry) {
    LV<uint32_t> r;
#ifdef CSH_EMUL
    r.v[0] = carry;
    for (int j = 1; j < 64; j++) r.v[j] = x.v[j - 1];
#else
    r.v = uint32_t(__builtin_amdgcn_update_dpp(int(carry), int(x.v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
#endif
    return r;
}
// the value of lane 63, in every lane (wave-uniform)
__device__ __forceinline__ static uint32_t llast(const LV<uint32_t> &x) {
#ifdef CSH_EMUL
    return x.v[63];
#else
    return uint32_t(__builtin_amdgcn_readlane(int(x.v), 63));
#endif
}
// the value of lane `lane` (wave-uniform), in every lane;  lset: the wave-uniform `val` into lane `lane` of x
__device__ __forceinline__ static uint32_t lget(const LV<uint32_t> &x, uint32_t lane) {
#ifdef CSH_EMUL
    return x.v[lane & 63u];
#else
    return uint32_t(__builtin_amdgcn_readlane(int(x.v), int(lane)));
#endif
}
__device__ __forceinline__ static void lset(LV<uint32_t> &x, uint32_t lane, uint32_t val) {
#ifdef CSH_EMUL
    x.v[lane & 63u] = val;
#else
    x.v = (threadIdx.x & 63u) == lane ? val : x.v;   // v_cmp + v_cndmask (v_writelane wants its lane select in m0: inline assembly for one instruction less)
#endif
}
// sum over the lanes, in every lane (all 64 lanes active)
__device__ __forceinline__ static uint32_t lsum32(const LV<uint32_t> &x) {
    uint32_t s;
    (void)lscan(x, s);
    return s;
}
// per-wave state that lives across the phases of a phased kernel (gpu_rt.h CSH_PHASE_LOOP): plain registers on the device; the emulation re-enters
// the kernel once per phase and wave-thread, so there it is a per-thread array indexed by the wave.  Declare it in front of CSH_PHASE_LOOP.
#ifdef CSH_EMUL
#define CSH_WPERSIST(T, name, N, NWAVES) static thread_local T name##_w_[NWAVES][N]; T (&name)[N] = name##_w_[threadIdx.x / CSP_WAVE_THREADS]
#else
#define CSH_WPERSIST(T, name, N, NWAVES) T name[N]
#endif
__device__ __forceinline__ static uint32_t popc64(uint64_t m) { return uint32_t(__popcll((unsigned long long)m)); }
// minimum over the lanes, in every lane
__device__ __forceinline__ static uint64_t lmin64(const LV<uint64_t> &x) {
#ifdef CSH_EMUL
    uint64_t m = x.v[0];
    for (int j = 1; j < 64; j++) m = x.v[j] < m ? x.v[j] : m;
    return m;
#else
    uint64_t v = x.v;
    CSH_UNROLL
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(v)), o, 64)), hi = uint32_t(__shfl_xor(int(uint32_t(v >> 32)), o, 64));
        const uint64_t y = (uint64_t(hi) << 32) | lo;
        v = y < v ? y : v;
    }
    return v;
#endif
}

}  // namespace csh
