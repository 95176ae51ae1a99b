#!/usr/bin/env python3
"""Writes valu_rates2.hip: issue cost of one VALU instruction form per kernel (16 independent registers, 64 instructions per loop trip)."""
import re
forms = [
    ("add_u32", "v_add_u32 %0, %0, %1"), ("add3_u32", "v_add3_u32 %0, %0, %1, %2"), ("mul_i32_i24", "v_mul_i32_i24 %0, %0, %1"), ("mad_i32_i24", "v_mad_i32_i24 %0, %1, %2, %0"),
    ("mul_hi_u32_u24", "v_mul_hi_u32_u24 %0, %0, %1"), ("mul_lo_u32", "v_mul_lo_u32 %0, %0, %1"), ("dot2_i32_i16", "v_dot2_i32_i16 %0, %1, %2, %0"),
    ("pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1"), ("pk_add_i16", "v_pk_add_i16 %0, %0, %1"), ("perm_b32", "v_perm_b32 %0, %0, %1, %2"),
    ("cvt_pk_i16_i32", "v_cvt_pk_i16_i32 %0, %0, %1"), ("med3_i32", "v_med3_i32 %0, %0, %1, %2"), ("ashrrev_i32", "v_ashrrev_i32 %0, 3, %0"),
    ("lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1"), ("lshl_or_b32", "v_lshl_or_b32 %0, %0, 16, %1"), ("and_or_b32", "v_and_or_b32 %0, %0, %1, %2"),
    ("bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96"), ("bfi_b32", "v_bfi_b32 %0, %1, %0, %2"), ("cvt_f32_i32", "v_cvt_f32_i32 %0, %0"),
    ("cvt_i32_f32", "v_cvt_i32_f32 %0, %0"), ("fma_f32", "v_fma_f32 %0, %1, %2, %0"), ("mul_f32", "v_mul_f32 %0, %0, %1"), ("fmaak_f32", "v_fmaak_f32 %0, %0, %1, 0x4b400000"),
    ("max_i32", "v_max_i32 %0, %0, %1"), ("or3_b32", "v_or3_b32 %0, %0, %1, %2"),
    ("mul24_sdwa", "v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"),
    # name, asm (%0 = accumulator register i, %1/%2 = loop-invariant VGPRs)
    ("sub_u32", "v_sub_u32 %0, %0, %1"), ("and_b32", "v_and_b32 %0, %0, %1"), ("or_b32", "v_or_b32 %0, %0, %1"), ("xor_b32", "v_xor_b32 %0, %0, %1"),
    ("lshlrev", "v_lshlrev_b32 %0, 1, %0"), ("lshrrev", "v_lshrrev_b32 %0, 1, %0"), ("mov_b32", "v_mov_b32 %0, %1"),
    ("min_i32", "v_min_i32 %0, %0, %1"), ("max_u32", "v_max_u32 %0, %0, %1"), ("mul_u32_u24", "v_mul_u32_u24 %0, %0, %1"),
    ("add_f32", "v_add_f32 %0, %0, %1"), ("sub_f32", "v_sub_f32 %0, %0, %1"), ("max_f32", "v_max_f32 %0, %0, %1"), ("min_f32", "v_min_f32 %0, %0, %1"),
    ("fmac_f32", "v_fmac_f32 %0, %1, %2"), ("mad_f32", "v_fma_f32 %0, %0, %1, %2"),
    ("cvt_f32_u32", "v_cvt_f32_u32 %0, %0"), ("cvt_u32_f32", "v_cvt_u32_f32 %0, %0"), ("rndne_f32", "v_rndne_f32 %0, %0"), ("trunc_f32", "v_trunc_f32 %0, %0"),
    ("floor_f32", "v_floor_f32 %0, %0"), ("not_b32", "v_not_b32 %0, %0"), ("ffbh_u32", "v_ffbh_u32 %0, %0"), ("cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0"),
    ("bfe_u32", "v_bfe_u32 %0, %0, 3, 9"), ("bfe_i32", "v_bfe_i32 %0, %0, 3, 9"), ("alignbit", "v_alignbit_b32 %0, %0, %1, 7"),
    ("min3_i32", "v_min3_i32 %0, %0, %1, %2"), ("max3_i32", "v_max3_i32 %0, %0, %1, %2"), ("med3_f32", "v_med3_f32 %0, %0, %1, %2"),
    ("add_lshl", "v_add_lshl_u32 %0, %0, %1, 2"), ("cmp_gt_i32", "v_cmp_gt_i32 vcc, %0, %1"), ("cmp_class_e64", "v_cmp_lt_u32 s[10:11], %0, %1"),
    ("cndmask_vcc", "v_cndmask_b32 %0, %0, %1, vcc"), ("cndmask_e64", "v_cndmask_b32 %0, %0, %1, s[12:13]"),
    ("add_u32_sdwa", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
    ("add_i32_sdwa_sext", "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
    ("mov_sdwa_w1_keep", "v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0"),
    ("add_f32_sdwa", "v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"),
    ("or_b32_sdwa", "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"),
    ("lshl_sdwa", "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0"),
    ("add_u32_dpp", "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("mov_b32_dpp", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("pk_add_u16", "v_pk_add_u16 %0, %0, %1"), ("pk_sub_i16", "v_pk_sub_i16 %0, %0, %1"), ("pk_min_i16", "v_pk_min_i16 %0, %0, %1"),
    ("pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 1, %0"), ("add_u16", "v_add_u16 %0, %0, %1"), ("mad_u16", "v_mad_u16 %0, %0, %1, %2"),
    ("add_co", "v_add_co_u32 %0, vcc, %0, %1"), ("addc_co", "v_addc_co_u32 %0, vcc, %0, %1, vcc"),
    ("mul_hi_u32", "v_mul_hi_u32 %0, %0, %1"), ("mul_hi_i32_i24", "v_mul_hi_i32_i24 %0, %0, %1"),
    ("add_const", "v_add_u32 %0, 0x12345, %0"), ("add_sgpr", "v_add_u32 %0, s14, %0"), ("mul24_sgpr", "v_mul_i32_i24 %0, s14, %0"), ("fma_sgpr", "v_fma_f32 %0, %0, s14, %1"),
    ("mad24_sgpr", "v_mad_i32_i24 %0, %1, s14, %0"), ("mul24_const", "v_mul_i32_i24 %0, 0x2333, %0"),
    # pairs: one slow then one fast per register -- do they overlap?
    ("PAIR_mul24+add", "v_mul_i32_i24 %0, %0, %1\\n v_add_u32 %0, %0, %2"), ("PAIR_add+ashr", "v_add_u32 %0, %0, %1\\n v_ashrrev_i32 %0, 3, %0"),
    ("PAIR_mad24+sub", "v_mad_i32_i24 %0, %1, %2, %0\\n v_sub_u32 %0, %0, %2"), ("PAIR_fma+cvt", "v_fma_f32 %0, %0, %1, %2\\n v_cvt_f32_i32 %0, %0"),
]
out = ['// generated by gen_rates2.py', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>']
for name, asm in forms:
    k = "k_" + name.replace("+", "_")
    lines = []
    for i in range(16):
        lines.append(asm.replace("%0", "%" + str(i)).replace("%1", "%16").replace("%2", "%17") if False else re.sub(r"%([012])", lambda m: "%" + str({0: i, 1: 16, 2: 17}[int(m.group(1))]), asm))
    asm16 = " ".join('"' + l + '\\n"' for l in lines)
    outs = ", ".join(f'"+v"(r[{i}])' for i in range(16))
    out.append(f'''__global__ void __launch_bounds__(256) {k}(uint32_t *out, int iters, uint32_t seed) {{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = seed * (threadIdx.x + 1) + i * 0x01010101u;
    uint32_t a = seed ^ 0x12345u, b = seed + 77u + threadIdx.x;
    asm volatile("s_mov_b64 s[12:13], 0x5555\\n s_mov_b32 s14, 0x3f800001\\n v_cmp_gt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "s12", "s13", "s14", "vcc", "s10", "s11");
    for (int it = 0; it < iters; it++) {{
        _Pragma("unroll") for (int u = 0; u < 4; u++) {{
            asm volatile({asm16} : {outs} : "v"(a), "v"(b) : "vcc", "s10", "s11");
        }}
    }}
    uint32_t s = 0;
    for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}}''')
out.append('typedef void (*kern_t)(uint32_t *, int, uint32_t);\nstruct Entry { const char *name; kern_t k; int n; };')
out.append('static Entry es[] = {' + ', '.join(f'{{"{n}", k_{n.replace("+", "_")}, {2 if n.startswith("PAIR") else 1}}}' for n, _ in forms) + '};')
out.append(r'''
int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    uint32_t *out;
    (void)hipMalloc(&out, size_t(cus) * 8 * 256 * 4);
    const int iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-22s", "cycles/wave-instr/SIMD at waves per SIMD =");
    for (int wps : {1, 2, 4, 8}) printf(" %6d", wps);
    printf("\n");
    for (auto &e : es) {
        printf("%-22s", e.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
            (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double winstr_per_simd = double(wps) * iters * 64.0 * e.n;
            const double cycles = best * 1e-3 * mhz * 1e6;
            printf(" %6.2f", cycles / winstr_per_simd);
        }
        printf("\n");
    }
    return 0;
}''')
open("valu_rates2.hip", "w").write("\n".join(out))
