// valu_rates.hip -- issue rate of the VALU instructions the transform kernels are built from (gfx950).
// Each kernel runs ITER x 16 independent instructions of one kind per wave; 8 waves per SIMD; reports wave-instructions per cycle per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define DEF_KERNEL(NAME, ASM)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, int iters, uint32_t seed) {                     \
        uint32_t r[16];                                                                                        \
        for (int i = 0; i < 16; i++) r[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;                        \
        uint32_t a = seed ^ 0x12345u, b = seed + 77u + threadIdx.x;                                            \
        for (int it = 0; it < iters; it++) {                                                                   \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                    \
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b)); \
            }                                                                                                  \
        }                                                                                                      \
        uint32_t s = 0;                                                                                        \
        for (int i = 0; i < 16; i++) s ^= r[i];                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                        \
    }

DEF_KERNEL(k_add, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(k_mul24, "v_mul_i32_i24 %0, %0, %1")
DEF_KERNEL(k_mad24, "v_mad_i32_i24 %0, %1, %2, %0")
DEF_KERNEL(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %1")
DEF_KERNEL(k_mullo32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_dot2_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(k_dot2c_i16, "v_dot2c_i32_i16 %0, %1, %2")
DEF_KERNEL(k_dot4_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
DEF_KERNEL(k_pk_mul_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEF_KERNEL(k_pk_add_i16, "v_pk_add_i16 %0, %0, %1")
DEF_KERNEL(k_pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %0")
DEF_KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF_KERNEL(k_pk_ashr_i16, "v_pk_ashrrev_i16 %0, 1, %0")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1")
DEF_KERNEL(k_med3, "v_med3_i32 %0, %0, %1, %2")
DEF_KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0")
DEF_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
DEF_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 16, %1")
DEF_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEF_KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
DEF_KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
DEF_KERNEL(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
DEF_KERNEL(k_sdwa_mul24, "v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
DEF_KERNEL(k_sat_pk_u8, "v_sat_pk_u8_i16 %0, %0")
DEF_KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
DEF_KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
DEF_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_KERNEL(k_mad_i32_i16, "v_mad_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(k_mad_u64_u32, "v_add_u32 %0, %0, %1")

// packed f32 needs 64-bit registers
__global__ void __launch_bounds__(256) k_pk_fma_f32(uint32_t *out, int iters, uint32_t seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[16];
    for (int i = 0; i < 16; i++) { r[i].x = float(seed * (threadIdx.x + 1) + i); r[i].y = float(i); }
    f2 a, b; a.x = 1.0001f; a.y = 0.9999f; b.x = float(seed); b.y = 0.5f;
    for (int it = 0; it < iters; it++) {
        _Pragma("unroll") for (int u = 0; u < 4; u++) {
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
}

typedef void (*kern_t)(uint32_t *, int, uint32_t);
struct Entry { const char *name; kern_t k; };
#define E(N) {#N, N}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    std::vector<Entry> es = {E(k_add), E(k_add3), E(k_mul24), E(k_mad24), E(k_mulhi24), E(k_mullo32), E(k_dot2_i16), E(k_dot2c_i16), E(k_dot4_i8),
        E(k_pk_mul_u16), E(k_pk_add_i16), E(k_pk_mad_i16), E(k_pk_max_i16), E(k_pk_ashr_i16), E(k_perm), E(k_cvt_pk_i16_i32), E(k_med3), E(k_ashr),
        E(k_lshl_add), E(k_lshl_or), E(k_and_or), E(k_bitop3), E(k_bfi), E(k_cvt_f32_i32), E(k_cvt_i32_f32), E(k_fma_f32), E(k_mul_f32), E(k_pk_fma_f32),
        E(k_sdwa_mul24), E(k_cndmask), E(k_max_i32), E(k_sat_pk_u8), E(k_or3), E(k_xad), E(k_mad_u32_u24), E(k_mad_i32_i16)};
    const int blocks = cus * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    uint32_t *out;
    hipMalloc(&out, size_t(blocks) * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // wave-instructions issued per CU: 8 wg * 4 waves * iters * 64
        const double winstr = 8.0 * 4.0 * iters * 64.0;
        const double cycles = best * 1e-3 * mhz * 1e6;
        printf("%-20s %8.3f ms  %6.3f wave-instr/cycle/CU  (%.2f cycles per wave-instr per SIMD)\n", e.name, best, winstr / cycles, cycles / (winstr / 4.0));
    }
    return 0;
}
