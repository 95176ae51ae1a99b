// k_pixel.hip -- phase 1: the pixel-domain transcode, one 8x8 block per lane, whole block in VGPRs.
//
// Replaces, for libcaesium's lossy JPEG path (reference call site /root/reference/src/compressor.rs:305;
// SURVEY.md 8a rows J2,J3,J5,J6,J7 and Appendix B.2-B.6), mozjpeg's
//   jidctint (dequantise + ISLOW IDCT + range limit), jdsample h2v2 / h2v1 fancy upsample,
//   jcsample h2v2 downsample with edge expansion, jfdctint (ISLOW FDCT) and the scalar quantiser.
// Full-resolution components go IDCT -> FDCT inside one lane (k_xform_direct); resampled components go
// k_idct_plane -> k_resample_plane (per sample) -> k_plane_fdct.
// All arithmetic is int32 exactly as the oracle's (oracle/jpeg_oracle.c); multiplies by the 13-bit DCT
// constants use the full-rate 24-bit multiplier, which is exact whenever dequantised coefficients are
// below 2^15 in magnitude (every stream made from 8-bit samples; libjpeg-turbo's SIMD IDCT has the same
// domain).  Memory: a block is 8 octets of 16 bytes; octet j of the wave's 64 blocks is one coalesced 1 KiB
// access, so a tile moves in 8 vector loads / 8 vector stores per lane; planes are written/read in row segments
// contiguous across the wave's adjacent blocks.  No LDS, no cross-lane traffic: the 2-D transforms never leave
// the lane's registers, so zig-zag <-> natural reordering is free (compile-time register naming).
#include <cmath>
#include <utility>

#include "kernels.h"

namespace csh {

// zig-zag index -> natural index, as a compile-time table: the block lives in 64 named VGPRs and every
// reordering below must resolve at compile time (a run-time register index would spill the block to scratch)
static constexpr uint8_t kZ2N[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
template <int I>
__device__ __forceinline__ static int half_of(const uint4 &v) {  // sign-extended int16 element I of the octet
    uint32_t w = (I / 2 == 0) ? v.x : (I / 2 == 1) ? v.y : (I / 2 == 2) ? v.z : v.w;
    return (I & 1) ? (int(w) >> 16) : (int(w << 16) >> 16);
}
using Oct = std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>;
// coefficient tiles and the retained DCT are streamed: read once here, written once for kernels that run many milliseconds (and 12.8 GB) later.  As
// non-temporal accesses they leave the L2 / MALL to the data that is reused (round 5: k_xform_direct 2.07 -> 1.95 ms per 1024 files, 5.9 -> 6.3 TB/s)
#ifndef CSH_EMUL
typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ static void nt_store16(void *p, const uint4 &v) { nt_u32x4 x; x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w; __builtin_nontemporal_store(x, reinterpret_cast<nt_u32x4 *>(p)); }
__device__ __forceinline__ static uint4 nt_load16(const void *p) { const nt_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(p)); uint4 v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w; return v; }
#else
__device__ __forceinline__ static void nt_store16(void *p, const uint4 &v) { *reinterpret_cast<uint4 *>(p) = v; }
__device__ __forceinline__ static uint4 nt_load16(const void *p) { return *reinterpret_cast<const uint4 *>(p); }
#endif


#define FIX_0_298 2446
#define FIX_0_390 3196
#define FIX_0_541 4433
#define FIX_0_765 6270
#define FIX_0_899 7373
#define FIX_1_175 9633
#define FIX_1_501 12299
#define FIX_1_847 15137
#define FIX_1_961 16069
#define FIX_2_053 16819
#define FIX_2_562 20995
#define FIX_3_072 25172
#define MUL(a, c) __mul24((a), (c))
#define MAD(a, c, b) (__mul24((a), (c)) + (b))   // v_mad_i32_i24

// What an instruction costs on gfx950 (tools/ubench/valu_rates*.hip, profiles/r05_valu_rates.txt): a wave64 VALU instruction issues in 2 cycles if it
// is a plain add / subtract / logic / right shift / move / f32 add, multiply or fma on VGPR or literal operands, and in 4 if it is anything else --
// every integer multiply (24- or 32-bit, v_dot2 alike), v_mad, v_add3, v_med3, min / max, v_perm, the left shift, every SDWA / DPP form, every
// packed 16-bit operation, every conversion, anything that reads an SGPR.  So the transforms below are written for FEW SLOW instructions, not for
// few instructions: 14 multiplies per 1-D transform, the rounding term of a pass folded into the multiply-add that makes the even part's z1 /
// tmp0 / tmp1 (or the odd part's z5) so that an output is one add and one right shift, and sums of three terms as two adds (v_add3 costs the
// same as two adds).  A v_dot2_i32_i16 form (16 dot products per 1-D transform) would need its inputs packed in pairs by v_perm (4 cycles a
// pair) and a 16-bit range check per pass: measured arithmetic, DESIGN.md 6 -- it does not pay.

// one 1-D inverse transform of jidctint; in: frequency order, out: sample order; DESCALE(., SH) of every output
template <int SH>
__device__ __forceinline__ static void idct1d(int &x0, int &x1, int &x2, int &x3, int &x4, int &x5, int &x6, int &x7) {
    const int R = 1 << (SH - 1);
    const int z1 = MUL(x2 + x6, FIX_0_541), tmp2 = MAD(x6, -FIX_1_847, z1), tmp3 = MAD(x2, FIX_0_765, z1);
    const int tmp0 = ((x0 + x4) << 13) + R, tmp1 = ((x0 - x4) << 13) + R;   // carries the rounding term of all eight outputs; a shift, not the 24-bit multiplier: the DC term of a 16-bit-table stream can pass 2^23
    const int t10 = tmp0 + tmp3, t13 = tmp0 - tmp3, t11 = tmp1 + tmp2, t12 = tmp1 - tmp2;
    int a0 = x7, a1 = x5, a2 = x3, a3 = x1;
    const int y1 = a0 + a3, y2 = a1 + a2;
    int y3 = a0 + a2, y4 = a1 + a3;
    const int y5 = MUL(y3 + y4, FIX_1_175);
    y3 = MAD(y3, -FIX_1_961, y5); y4 = MAD(y4, -FIX_0_390, y5);
    const int p1 = MUL(y1, -FIX_0_899), p2 = MUL(y2, -FIX_2_562);
    a0 = MAD(a0, FIX_0_298, p1) + y3; a1 = MAD(a1, FIX_2_053, p2) + y4; a2 = MAD(a2, FIX_3_072, p2) + y3; a3 = MAD(a3, FIX_1_501, p1) + y4;
    x0 = (t10 + a3) >> SH; x7 = (t10 - a3) >> SH;
    x1 = (t11 + a2) >> SH; x6 = (t11 - a2) >> SH;
    x2 = (t12 + a1) >> SH; x5 = (t12 - a1) >> SH;
    x3 = (t13 + a0) >> SH; x4 = (t13 - a0) >> SH;
}

// one 1-D forward transform of jfdctint; FIRST: row pass (<<2, descale 11), else column pass (descale 2 / 15).
// The multiply-add statement of the pass: the kernels run the packed form below (fdct1d_pk); this one is what tests/xform_block_check.cpp compares it with
// (and, through it, with the oracle), so it stays next to it.
template <bool FIRST>
__device__ __forceinline__ static void fdct1d(int &d0, int &d1, int &d2, int &d3, int &d4, int &d5, int &d6, int &d7) {
    int tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    const int SH = FIRST ? 11 : 15, R = 1 << (SH - 1);
    int o0, o4;
    if (FIRST) { o0 = (tmp10 + tmp11) * 4; o4 = (tmp10 - tmp11) * 4; }
    else { o0 = (tmp10 + tmp11 + 2) >> 2; o4 = (tmp10 - tmp11 + 2) >> 2; }
    const int z1 = MAD(tmp12 + tmp13, FIX_0_541, R);
    const int o2 = MAD(tmp13, FIX_0_765, z1) >> SH, o6 = MAD(tmp12, -FIX_1_847, z1) >> SH;
    const int y1 = tmp4 + tmp7, y2 = tmp5 + tmp6;
    int y3 = tmp4 + tmp6, y4 = tmp5 + tmp7;
    const int y5 = MAD(y3 + y4, FIX_1_175, R);   // every odd output holds exactly one of y3 / y4, so z5 carries their rounding term
    y3 = MAD(y3, -FIX_1_961, y5); y4 = MAD(y4, -FIX_0_390, y5);
    const int p1 = MUL(y1, -FIX_0_899), p2 = MUL(y2, -FIX_2_562);
    d0 = o0; d4 = o4; d2 = o2; d6 = o6;
    d7 = (MAD(tmp4, FIX_0_298, p1) + y3) >> SH; d5 = (MAD(tmp5, FIX_2_053, p2) + y4) >> SH;
    d3 = (MAD(tmp6, FIX_3_072, p2) + y3) >> SH; d1 = (MAD(tmp7, FIX_1_501, p1) + y4) >> SH;
}

// two 16-bit halves in one register: lo's low half | hi's low half << 16 (v_perm_b32: one instruction, nothing to mask)
__device__ __forceinline__ static uint32_t pack_halves(uint32_t lo, uint32_t hi) {
#ifdef CSH_EMUL
    return (lo & 0xFFFFu) | (hi << 16);
#else
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
#endif
}
__device__ __forceinline__ static uint32_t float_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

// load (8 x 16-byte octets) + dequantise + 2-D IDCT + level shift + range limit; samples (0..255) out, natural order
template <int J, int... I>
__device__ __forceinline__ static void dequant_octet(int x[64], const uint4 &v, const DevQuant &q, std::integer_sequence<int, I...>) {
    ((x[kZ2N[8 * J + I]] = MUL(half_of<I>(v), int(q.q[8 * J + I]))), ...);   // 16-bit coefficient x 16-bit table entry: the 24-bit multiplier is exact and full rate (v_mul_lo_u32 is quarter rate)
}
template <int... J>
__device__ __forceinline__ static void load_dequant(const int16_t *__restrict__ blk, const DevQuant &q, int x[64], std::integer_sequence<int, J...>) {
    const uint4 v[8] = {nt_load16(blk + CSH_OCT_STRIDE * J)...};  // all eight loads in flight before first use
    (dequant_octet<J>(x, v[J], q, Oct()), ...);
}
// CENTRED: samples come out level-shifted (-128..127: what the forward transform takes) -- one clamp instead of add + clamp, and no subtraction in front of the FDCT
template <bool CENTRED = false>
__device__ __forceinline__ static void load_idct(const int16_t *__restrict__ blk, const DevQuant &q, int x[64]) {
    load_dequant(blk, q, x, Oct());
    CSH_SCHED_FENCE();
    CSH_UNROLL
    for (int c = 0; c < 8; c++) idct1d<11>(x[c], x[8 + c], x[16 + c], x[24 + c], x[32 + c], x[40 + c], x[48 + c], x[56 + c]);
    CSH_SCHED_FENCE();
    CSH_UNROLL
    for (int r = 0; r < 8; r++) {
        idct1d<18>(x[8 * r], x[8 * r + 1], x[8 * r + 2], x[8 * r + 3], x[8 * r + 4], x[8 * r + 5], x[8 * r + 6], x[8 * r + 7]);
        CSH_UNROLL
        for (int c = 0; c < 8; c++) {
            if (CENTRED) { const int v = x[8 * r + c]; x[8 * r + c] = v < -128 ? -128 : (v > 127 ? 127 : v); }
            else { const int v = x[8 * r + c] + 128; x[8 * r + c] = v < 0 ? 0 : (v > 255 ? 255 : v); }
        }
    }
    CSH_SCHED_FENCE();
}

// replicate the last valid column / row of an edge block (decoder crop + encoder edge expansion)
__device__ __forceinline__ static void replicate_edges(int x[64], int vc, int vr) {
    // propagate the last valid sample rightwards / downwards one step at a time: pure selects between neighbouring
    // registers (a "pick column vc-1" formulation makes the compiler index the block dynamically = scratch)
    CSH_UNROLL
    for (int c = 1; c < 8; c++) {
        const bool keep = c < vc;
        CSH_UNROLL
        for (int r = 0; r < 8; r++) x[8 * r + c] = keep ? x[8 * r + c] : x[8 * r + c - 1];
    }
    CSH_UNROLL
    for (int r = 1; r < 8; r++) {
        const bool keep = r < vr;
        CSH_UNROLL
        for (int c = 0; c < 8; c++) x[8 * r + c] = keep ? x[8 * r + c] : x[8 * (r - 1) + c];
    }
}

// samples (0..255, natural order) -> level shift -> 2-D FDCT -> scalar quantise -> store as 8 x 16-byte octets
// The scalar quantiser sign(t) * ((|t| + d / 2) / d), d = 8 q, as ONE fused multiply-add in f32: t * rcp + 1.5 * 2^23, whose low
// 16 bits are the two's-complement quotient.  rcp = fl((1 / d) (1 + 2^-19)) (types.h DevQuant::rcp): the product is formed exactly, the nudge
// puts every tie t = d (k + 1/2) beyond its half (away from zero, as the integer form rounds) and moves nothing else across a half (|t| / d *
// 2^-19 < 1 / (16 d), the next quotient down is 1 / d away), and the add's round-to-nearest at 2^23 (spacing 1) is the rounding itself.
// Exhaustive over every d = 8 q, q < 65536, and |t| <= 2^15: tests/test_quant_reciprocal.py.  Three instructions a coefficient (convert,
// fma, half of a v_perm) against eight for the integer reciprocal it replaces; any 16-bit table value (round 4's form stopped at q < 2048).
template <int K>
__device__ __forceinline__ static uint32_t quant_one(const int x[64], const DevQuant &q) {
    return float_bits(__builtin_fmaf(float(x[kZ2N[K]]), q.rcp[K], 12582912.0f));   // low half = the level
}
template <int J>
__device__ __forceinline__ static void quant_store_octet(const int x[64], const DevQuant &q, int16_t *__restrict__ blk) {
    uint4 v;
    v.x = pack_halves(quant_one<8 * J + 0>(x, q), quant_one<8 * J + 1>(x, q));
    v.y = pack_halves(quant_one<8 * J + 2>(x, q), quant_one<8 * J + 3>(x, q));
    v.z = pack_halves(quant_one<8 * J + 4>(x, q), quant_one<8 * J + 5>(x, q));
    v.w = pack_halves(quant_one<8 * J + 6>(x, q), quant_one<8 * J + 7>(x, q));
    nt_store16(blk + CSH_OCT_STRIDE * J, v);
}
template <int... J>
__device__ __forceinline__ static void quant_store_all(const int x[64], const DevQuant &q, int16_t *__restrict__ blk, std::integer_sequence<int, J...>) {
    ((quant_store_octet<J>(x, q, blk), CSH_SCHED_FENCE()), ...);
}
template <int J>
__device__ __forceinline__ static void raw_store_octet(const int x[64], int16_t *__restrict__ raw) {
    uint4 v;
    v.x = pack_halves(uint32_t(x[kZ2N[8 * J + 0]]), uint32_t(x[kZ2N[8 * J + 1]]));
    v.y = pack_halves(uint32_t(x[kZ2N[8 * J + 2]]), uint32_t(x[kZ2N[8 * J + 3]]));
    v.z = pack_halves(uint32_t(x[kZ2N[8 * J + 4]]), uint32_t(x[kZ2N[8 * J + 5]]));
    v.w = pack_halves(uint32_t(x[kZ2N[8 * J + 6]]), uint32_t(x[kZ2N[8 * J + 7]]));
    *reinterpret_cast<uint4 *>(raw + CSH_RAW_OCT * J) = v;
}
template <int... J>
__device__ __forceinline__ static void raw_store_all(const int x[64], int16_t *__restrict__ raw, std::integer_sequence<int, J...>) {
    (raw_store_octet<J>(x, raw), ...);
}
// The retained DCT is block-major (kernels.h raw_index): a lane's block is one 128-byte line, and eight 16-byte stores per lane at a 128-byte
// stride are 512 partial-line requests per wave.  So the wave's 64 blocks go through LDS: every lane puts its eight pieces into its row of
// the wave's 8 KiB (piece J in slot J ^ (lane & 7): the rows' equal slots would all meet in one bank group), then lane l stores piece l & 7
// of blocks l >> 3, 8 + (l >> 3), ..: eight lanes write one whole line, a store instruction eight whole lines.  The 8 KiB are the wave's
// columns of the deringing scratch (free again by the time the forward DCT is through).  The emulation stores the pieces directly.
#ifdef CSH_EMUL
#define CSH_RAW_VIA_LDS 0
#else
#define CSH_RAW_VIA_LDS 1
#endif
template <int J>
__device__ __forceinline__ static void raw_put_octet(const int x[64], int16_t (*tr)[256], int tid) {
    uint4 v;
    v.x = pack_halves(uint32_t(x[kZ2N[8 * J + 0]]), uint32_t(x[kZ2N[8 * J + 1]]));
    v.y = pack_halves(uint32_t(x[kZ2N[8 * J + 2]]), uint32_t(x[kZ2N[8 * J + 3]]));
    v.z = pack_halves(uint32_t(x[kZ2N[8 * J + 4]]), uint32_t(x[kZ2N[8 * J + 5]]));
    v.w = pack_halves(uint32_t(x[kZ2N[8 * J + 6]]), uint32_t(x[kZ2N[8 * J + 7]]));
    const int l = tid & 63;
    *reinterpret_cast<uint4 *>(&tr[l][(tid & ~63) + 8 * (J ^ (l & 7))]) = v;
}
template <int... J>
__device__ __forceinline__ static void raw_put_all(const int x[64], int16_t (*tr)[256], int tid, std::integer_sequence<int, J...>) {
    (raw_put_octet<J>(x, tr, tid), ...);
}
// every lane of the wave comes here (lanes without a block included): tile_raw = the retained DCT of the wave's first block
__device__ __forceinline__ static void raw_copy_out(int16_t *__restrict__ tile_raw, bool has_raw, int16_t (*tr)[256], int tid) {
#if CSH_RAW_VIA_LDS
    const uint64_t mask = __ballot(has_raw);
    if (!mask) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int l = tid & 63;
    CSH_UNROLL
    for (int j = 0; j < 8; j++) {
        const int src = 8 * j + (l >> 3);
        if ((mask >> src) & 1ull) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&tr[src][(tid & ~63) + 8 * ((l & 7) ^ (src & 7))]);
            nt_store16(tile_raw + src * 64 + (l & 7) * CSH_RAW_OCT, v);
        }
    }
#else
    (void)tile_raw; (void)has_raw; (void)tr; (void)tid;
#endif
}
// mozjpeg's overshoot deringing (jcdctmgr.c preprocess_deringing + catmull_rom; on in the JCP_MAX_COMPRESSION profile libcaesium's -q runs,
// /root/reference/src/compressor.rs:415,427; [UPSTREAM-RECALL], the statement checked against: oracle/jpeg_oracle.c cso_dering_block).
// On the level-shifted samples, walked in zig-zag order as one line: every run of samples at the top of the range (>= 127) becomes a
// Catmull-Rom arc through the slopes either side of it, overshooting by at most min(31, 2 * DC quantiser, the block's headroom).
// Few blocks have such samples (and not all 64 of them): a wave whose lanes have none pays the count only; otherwise the lanes that
// need it put their block into a column of LDS and walk it there (the walk indexes the samples with run-time positions, which
// registers cannot), the others wait.  float arithmetic in the C source's order; the steps 1 / (length + 1) come from a table so that
// no device division is involved.
#define CSH_DERING_LDS int16_t (*dr_col)[256]
__device__ static const uint8_t kZ2Nrt[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
__device__ static const float kDeringStep[65] = {
    0.f, 1.f / 1.f, 1.f / 2.f, 1.f / 3.f, 1.f / 4.f, 1.f / 5.f, 1.f / 6.f, 1.f / 7.f, 1.f / 8.f, 1.f / 9.f, 1.f / 10.f, 1.f / 11.f, 1.f / 12.f, 1.f / 13.f, 1.f / 14.f,
    1.f / 15.f, 1.f / 16.f, 1.f / 17.f, 1.f / 18.f, 1.f / 19.f, 1.f / 20.f, 1.f / 21.f, 1.f / 22.f, 1.f / 23.f, 1.f / 24.f, 1.f / 25.f, 1.f / 26.f, 1.f / 27.f, 1.f / 28.f,
    1.f / 29.f, 1.f / 30.f, 1.f / 31.f, 1.f / 32.f, 1.f / 33.f, 1.f / 34.f, 1.f / 35.f, 1.f / 36.f, 1.f / 37.f, 1.f / 38.f, 1.f / 39.f, 1.f / 40.f, 1.f / 41.f, 1.f / 42.f,
    1.f / 43.f, 1.f / 44.f, 1.f / 45.f, 1.f / 46.f, 1.f / 47.f, 1.f / 48.f, 1.f / 49.f, 1.f / 50.f, 1.f / 51.f, 1.f / 52.f, 1.f / 53.f, 1.f / 54.f, 1.f / 55.f, 1.f / 56.f,
    1.f / 57.f, 1.f / 58.f, 1.f / 59.f, 1.f / 60.f, 1.f / 61.f, 1.f / 62.f, 1.f / 63.f, 1.f / 64.f};
__device__ static float dering_catmull_rom(int value1, int value2, int value3, int value4, float t, int size) {
    const int tan1 = (value3 - value1) * size, tan2 = (value4 - value2) * size;
    const float t2 = t * t, t3 = t2 * t;
    const float f1 = 2.f * t3 - 3.f * t2 + 1.f, f2 = -2.f * t3 + 3.f * t2, f3 = t3 - 2.f * t2 + t, f4 = t3 - t2;
    return float(value2) * f1 + float(tan1) * f3 + float(value3) * f2 + float(tan2) * f4;
}
__device__ static void dering_walk(int16_t *col /* sample n (natural order) at col[n * 256] */, int dc_quant, int cnt, int sum) {
    const int maxsample = 127, size = 64;
    const int over = 2 * dc_quant < 31 ? 2 * dc_quant : 31, room = (maxsample * size - sum) / cnt;
    const int maxovershoot = maxsample + (over < room ? over : room);
    auto at = [&](int n) -> int16_t & { return col[int(kZ2Nrt[n]) * 256]; };
    int n = 0;
    do {
        if (at(n) < maxsample) { n++; continue; }
        const int start = n;
        while (++n < size && at(n) >= maxsample) {}
        const int end = n;
        const float f1 = float(at(start >= 1 ? start - 1 : 0)), f2 = float(at(start >= 2 ? start - 2 : 0));
        const float l1 = float(at(end < size - 1 ? end : size - 1)), l2 = float(at(end < size - 2 ? end + 1 : size - 1));
        float fslope = f1 - f2 > float(maxsample) - f1 ? f1 - f2 : float(maxsample) - f1;
        float lslope = l1 - l2 > float(maxsample) - l1 ? l1 - l2 : float(maxsample) - l1;
        if (start == 0) fslope = lslope;
        if (end == size) lslope = fslope;
        const int length = end - start;
        const float step = kDeringStep[length + 1];
        float position = step;
        for (int i = start; i < end; i++, position += step) {
            const int tmp = int(ceilf(dering_catmull_rom(int(float(maxsample) - fslope), maxsample, maxsample, int(float(maxsample) - lslope), position, length)));
            at(i) = int16_t(tmp < maxovershoot ? tmp : maxovershoot);
        }
        n++;
    } while (n < size);
}
// ---- the forward transform on PACKED samples.  jfdctint's inputs are 9-bit (level-shifted samples, up to 158 after deringing) and its first
// pass's outputs 14-bit, so both passes take their inputs as pairs of int16 in one register: the first butterfly is a packed add and a packed
// subtract (v_pk_add_i16 / v_pk_sub_i16: two butterflies each), and every output is two v_dot2_i32_i16 -- the even and odd parts of ISLOW written as
// the exact integer dot products they are (the constants below are the sums of jfdctint's; the pre-shift sum of an output equals ISLOW's, so does
// its DESCALE).  26 / 28 instructions per 1-D pass against 42 / 44 in the multiply-add form; every VALU instruction of a mixed integer stream
// costs about four cycles on gfx950 (profiles/r05_valu_rates.txt), so the count is what matters.  No range check: the bounds hold for every input.
// A block's row r is four registers: A = (d0, d1), B = (d7, d6), C = (d2, d3), D = (d5, d4).
// UNCENTRED samples (0..255, +31 after deringing) may be transformed as they are: the level shift only moves the DC term -- + 8 * 128 * 4 behind
// the row pass, which the column pass of column 0 turns into + 8192 on coefficient (0, 0); its rounding term takes that back (BIAS0).
constexpr uint32_t PK(int lo, int hi) { return (uint32_t(lo) & 0xFFFFu) | (uint32_t(hi) << 16); }
#ifdef CSH_EMUL
__device__ __forceinline__ static uint32_t pk_add(uint32_t a, uint32_t b) { return ((a + b) & 0xFFFFu) | ((a & 0xFFFF0000u) + (b & 0xFFFF0000u)); }
__device__ __forceinline__ static uint32_t pk_sub(uint32_t a, uint32_t b) { return ((a - b) & 0xFFFFu) | ((a & 0xFFFF0000u) - (b & 0xFFFF0000u)); }
__device__ __forceinline__ static uint32_t pk_max(uint32_t a, uint32_t b) {
    const int16_t al = int16_t(a), bl = int16_t(b), ah = int16_t(a >> 16), bh = int16_t(b >> 16);
    return uint32_t(uint16_t(al > bl ? al : bl)) | (uint32_t(uint16_t(ah > bh ? ah : bh)) << 16);
}
__device__ __forceinline__ static int dot2(uint32_t a, uint32_t k, int acc) {
    return int(uint32_t(acc) + uint32_t(int(int16_t(a)) * int(int16_t(k))) + uint32_t(int(int16_t(a >> 16)) * int(int16_t(k >> 16))));
}
__device__ __forceinline__ static uint32_t pack_hi_halves(uint32_t lo, uint32_t hi) { return (lo >> 16) | (hi & 0xFFFF0000u); }
// bytes I0, I1 of w, zero-extended to two halves
template <int I0, int I1> __device__ __forceinline__ static uint32_t bytes_to_halves(uint32_t w) { return ((w >> (8 * I0)) & 255u) | (((w >> (8 * I1)) & 255u) << 16); }
#else
typedef short pk16_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ static uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk16_t, a) + __builtin_bit_cast(pk16_t, b)); }
__device__ __forceinline__ static uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk16_t, a) - __builtin_bit_cast(pk16_t, b)); }
__device__ __forceinline__ static uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk16_t, a), __builtin_bit_cast(pk16_t, b))); }
__device__ __forceinline__ static int dot2(uint32_t a, uint32_t k, int acc) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(pk16_t, a), __builtin_bit_cast(pk16_t, k), acc, false); }
__device__ __forceinline__ static uint32_t pack_hi_halves(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
template <int I0, int I1> __device__ __forceinline__ static uint32_t bytes_to_halves(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c000c00u | uint32_t(I0) | (uint32_t(I1) << 16)); }
#endif
// the odd part as a 4 x 4 matrix on (tmp4, tmp5, tmp6, tmp7): what jfdctint's z1..z5 network sums up to, per output
constexpr int kO7[4] = {FIX_0_298 - FIX_0_899 - FIX_1_961 + FIX_1_175, FIX_1_175, -FIX_1_961 + FIX_1_175, -FIX_0_899 + FIX_1_175};
constexpr int kO5[4] = {FIX_1_175, FIX_2_053 - FIX_2_562 - FIX_0_390 + FIX_1_175, -FIX_2_562 + FIX_1_175, -FIX_0_390 + FIX_1_175};
constexpr int kO3[4] = {-FIX_1_961 + FIX_1_175, -FIX_2_562 + FIX_1_175, FIX_3_072 - FIX_2_562 - FIX_1_961 + FIX_1_175, FIX_1_175};
constexpr int kO1[4] = {-FIX_0_899 + FIX_1_175, -FIX_0_390 + FIX_1_175, FIX_1_175, FIX_1_501 - FIX_0_899 - FIX_0_390 + FIX_1_175};
template <bool FIRST>
__device__ __forceinline__ static void fdct1d_pk(uint32_t A, uint32_t B, uint32_t C, uint32_t D, int R, int R0, int R4, int &o0, int &o1, int &o2, int &o3, int &o4, int &o5, int &o6, int &o7) {
    const uint32_t P = pk_add(A, B), M = pk_sub(A, B), Q = pk_add(C, D), N = pk_sub(C, D);   // (tmp0, tmp1), (tmp7, tmp6), (tmp2, tmp3), (tmp5, tmp4)
    const int SH = FIRST ? 11 : 15;
    // R: the rounding term of the outputs that are descaled by SH; R0 / R4: what out0 / out4 start from (row pass: 0; column pass: 2, and column 0 of an
    // uncentred block 2 - 32768)
    constexpr uint32_t KE0P = FIRST ? PK(4, 4) : PK(1, 1), KE0Q = KE0P, KE4P = FIRST ? PK(4, -4) : PK(1, -1), KE4Q = FIRST ? PK(-4, 4) : PK(-1, 1);
    // z1 = (tmp12 + tmp13) c4433 with tmp13 = tmp0 - tmp3, tmp12 = tmp1 - tmp2: out2 = tmp13 (c4433 + c6270) + tmp12 c4433, out6 = tmp13 c4433 + tmp12 (c4433 - c15137)
    constexpr uint32_t KE2P = PK(FIX_0_541 + FIX_0_765, FIX_0_541), KE2Q = PK(-FIX_0_541, -(FIX_0_541 + FIX_0_765));
    constexpr uint32_t KE6P = PK(FIX_0_541, FIX_0_541 - FIX_1_847), KE6Q = PK(FIX_1_847 - FIX_0_541, -FIX_0_541);
#ifdef CSH_EMUL
    o0 = dot2(P, KE0P, dot2(Q, KE0Q, R0)); o4 = dot2(P, KE4P, dot2(Q, KE4Q, R4));
    o2 = dot2(P, KE2P, dot2(Q, KE2Q, R)); o6 = dot2(P, KE6P, dot2(Q, KE6Q, R));
    o7 = dot2(M, PK(kO7[3], kO7[2]), dot2(N, PK(kO7[1], kO7[0]), R));
    o5 = dot2(M, PK(kO5[3], kO5[2]), dot2(N, PK(kO5[1], kO5[0]), R));
    o3 = dot2(M, PK(kO3[3], kO3[2]), dot2(N, PK(kO3[1], kO3[0]), R));
    o1 = dot2(M, PK(kO1[3], kO1[2]), dot2(N, PK(kO1[1], kO1[0]), R));
#else
    // the three-operand v_dot2_i32_i16 with its constants in scalar registers.  (Through the builtin the compiler picks the two-operand v_dot2c with a
    // literal, which needs a v_mov of the rounding term in front of every output: 128 more instructions per block.  One asm statement per half
    // transform: between separate statements it would put a wait state.)  The compiler's hazard recogniser does not see inside an asm statement: a DOT write
    // needs three wait states before a non-DOT VALU instruction reads it, and the shifts behind the statement read o*; `s_nop 2` at its end makes that
    // structural instead of an accident of scheduling.
    asm("v_dot2_i32_i16 %0, %5, %8, %11\n\tv_dot2_i32_i16 %1, %5, %10, %12\n\tv_dot2_i32_i16 %2, %5, %14, %7\n\tv_dot2_i32_i16 %3, %5, %16, %7\n\t"
        "v_dot2_i32_i16 %0, %4, %6, %0\n\tv_dot2_i32_i16 %1, %4, %9, %1\n\tv_dot2_i32_i16 %2, %4, %13, %2\n\tv_dot2_i32_i16 %3, %4, %15, %3\n\ts_nop 2"
        : "=&v"(o0), "=&v"(o4), "=&v"(o2), "=&v"(o6)
        : "v"(P), "v"(Q), "s"(KE0P), "v"(R), "s"(KE0Q), "s"(KE4P), "s"(KE4Q), "v"(R0), "v"(R4), "s"(KE2P), "s"(KE2Q), "s"(KE6P), "s"(KE6Q));
    asm("v_dot2_i32_i16 %0, %5, %8, %6\n\tv_dot2_i32_i16 %1, %5, %10, %6\n\tv_dot2_i32_i16 %2, %5, %12, %6\n\tv_dot2_i32_i16 %3, %5, %14, %6\n\t"
        "v_dot2_i32_i16 %0, %4, %7, %0\n\tv_dot2_i32_i16 %1, %4, %9, %1\n\tv_dot2_i32_i16 %2, %4, %11, %2\n\tv_dot2_i32_i16 %3, %4, %13, %3\n\ts_nop 2"
        : "=&v"(o7), "=&v"(o5), "=&v"(o3), "=&v"(o1)
        : "v"(M), "v"(N), "v"(R), "s"(PK(kO7[3], kO7[2])), "s"(PK(kO7[1], kO7[0])), "s"(PK(kO5[3], kO5[2])), "s"(PK(kO5[1], kO5[0])),
          "s"(PK(kO3[3], kO3[2])), "s"(PK(kO3[1], kO3[0])), "s"(PK(kO1[3], kO1[2])), "s"(PK(kO1[1], kO1[0])));
#endif
    if (!FIRST) { o0 >>= 2; o4 >>= 2; }
    o2 >>= SH; o6 >>= SH; o7 >>= SH; o5 >>= SH; o3 >>= SH; o1 >>= SH;
}
// deringing on the packed block: few blocks hold a sample at the top of the range -- a packed maximum finds them (31 instructions), the others pay nothing more
template <bool CENTRED>
__device__ __forceinline__ static void dering_block_pk(uint32_t pr[8][4], int dc_quant, CSH_DERING_LDS) {
    uint32_t m = pr[0][0];
    CSH_UNROLL
    for (int i = 1; i < 32; i++) m = pk_max(m, pr[i >> 2][i & 3]);
    const int top = CENTRED ? 127 : 255, off = CENTRED ? 0 : 128;
    const bool any = int(int16_t(m)) >= top || int(int16_t(m >> 16)) >= top;
#ifndef CSH_EMUL
    if (!__ballot(any)) return;
#endif
    if (!any) return;
    // natural order: row r = (A.lo, A.hi, C.lo, C.hi, D.hi, D.lo, B.hi, B.lo)
    int x[64];
    CSH_UNROLL
    for (int r = 0; r < 8; r++) {
        const uint32_t A = pr[r][0], B = pr[r][1], C = pr[r][2], D = pr[r][3];
        x[8 * r + 0] = int(int16_t(A)) - off; x[8 * r + 1] = int(int16_t(A >> 16)) - off; x[8 * r + 2] = int(int16_t(C)) - off; x[8 * r + 3] = int(int16_t(C >> 16)) - off;
        x[8 * r + 4] = int(int16_t(D >> 16)) - off; x[8 * r + 5] = int(int16_t(D)) - off; x[8 * r + 6] = int(int16_t(B >> 16)) - off; x[8 * r + 7] = int(int16_t(B)) - off;
    }
    int cnt = 0, sum = 0;
    CSH_UNROLL
    for (int i = 0; i < 64; i++) { sum += x[i]; cnt += x[i] >= 127 ? 1 : 0; }
    if (cnt == 64) return;
    int16_t *col = &dr_col[0][threadIdx.x];
    CSH_UNROLL
    for (int i = 0; i < 64; i++) col[i * 256] = int16_t(x[i]);
    dering_walk(col, dc_quant, cnt, sum);
    CSH_UNROLL
    for (int r = 0; r < 8; r++) {
        auto at = [&](int c) { return uint32_t(int(col[(8 * r + c) * 256]) + off); };
        pr[r][0] = pack_halves(at(0), at(1)); pr[r][1] = pack_halves(at(7), at(6)); pr[r][2] = pack_halves(at(2), at(3)); pr[r][3] = pack_halves(at(5), at(4));
    }
}
// packed rows -> [deringing] -> 2-D FDCT -> [retained DCT] -> quantise -> store
template <bool DERING, bool CENTRED>
__device__ __forceinline__ static void fdct_quant_store_pk(uint32_t pr[8][4], const DevQuant &q, int16_t *__restrict__ blk, int16_t *__restrict__ raw, CSH_DERING_LDS) {
    CSH_SCHED_FENCE();
    if (DERING) { dering_block_pk<CENTRED>(pr, int(q.q[0]), dr_col); CSH_SCHED_FENCE(); }
    int x[64];
    int R = 1 << 10, Z = 0;
    CSH_PIN(R); CSH_PIN(Z);
    CSH_UNROLL
    for (int r = 0; r < 8; r++)
        fdct1d_pk<true>(pr[r][0], pr[r][1], pr[r][2], pr[r][3], R, Z, Z, x[8 * r], x[8 * r + 1], x[8 * r + 2], x[8 * r + 3], x[8 * r + 4], x[8 * r + 5], x[8 * r + 6], x[8 * r + 7]);
    CSH_SCHED_FENCE();
    int R2 = 1 << 14, TWO = 2, RDC = 2 - 32768;   // RDC: the level shift an uncentred block did not have, taken out of coefficient (0, 0)
    CSH_PIN(R2); CSH_PIN(TWO); CSH_PIN(RDC);
    CSH_UNROLL
    for (int c = 0; c < 8; c++) {
        const uint32_t A = pack_halves(uint32_t(x[c]), uint32_t(x[8 + c])), B = pack_halves(uint32_t(x[56 + c]), uint32_t(x[48 + c]));
        const uint32_t C = pack_halves(uint32_t(x[16 + c]), uint32_t(x[24 + c])), D = pack_halves(uint32_t(x[40 + c]), uint32_t(x[32 + c]));
        fdct1d_pk<false>(A, B, C, D, R2, (c == 0 && !CENTRED) ? RDC : TWO, TWO, x[c], x[8 + c], x[16 + c], x[24 + c], x[32 + c], x[40 + c], x[48 + c], x[56 + c]);
    }
    CSH_SCHED_FENCE();
    if (raw) {   // size-targeting keeps the unquantised DCT so later tries only re-quantise; the trellis quantiser works from it
        if (CSH_RAW_VIA_LDS) raw_put_all(x, dr_col, int(threadIdx.x), Oct()); else raw_store_all(x, raw, Oct());
    }
    quant_store_all(x, q, blk, Oct());
}
// centred samples in registers (natural order) -> the packed rows
__device__ __forceinline__ static void pack_rows(const int x[64], uint32_t pr[8][4]) {
    CSH_UNROLL
    for (int r = 0; r < 8; r++) {
        pr[r][0] = pack_halves(uint32_t(x[8 * r]), uint32_t(x[8 * r + 1])); pr[r][1] = pack_halves(uint32_t(x[8 * r + 7]), uint32_t(x[8 * r + 6]));
        pr[r][2] = pack_halves(uint32_t(x[8 * r + 2]), uint32_t(x[8 * r + 3])); pr[r][3] = pack_halves(uint32_t(x[8 * r + 5]), uint32_t(x[8 * r + 4]));
    }
}
template <bool DERING, bool CENTRED = false>
__device__ __forceinline__ static void fdct_quant_store(int x[64], const DevQuant &q, int16_t *__restrict__ blk, int16_t *__restrict__ raw, CSH_DERING_LDS) {
    uint32_t pr[8][4];
    pack_rows(x, pr);
    fdct_quant_store_pk<DERING, CENTRED>(pr, q, blk, raw, dr_col);
}

// re-quantise a retained DCT block with another table (k_requant)
template <int J, int... I>
__device__ __forceinline__ static void raw_to_nat(int x[64], const uint4 &v, std::integer_sequence<int, I...>) {
    ((x[kZ2N[8 * J + I]] = half_of<I>(v)), ...);
}
template <int... J>
__device__ __forceinline__ static void requant_block(const int16_t *__restrict__ raw, const DevQuant &q, int16_t *__restrict__ blk, std::integer_sequence<int, J...>) {
    const uint4 v[8] = {*reinterpret_cast<const uint4 *>(raw + CSH_RAW_OCT * J)...};
    int x[64];
    (raw_to_nat<J>(x, v[J], Oct()), ...);
    quant_store_all(x, q, blk, Oct());
}

__device__ __forceinline__ static void store_zero_block(int16_t *__restrict__ blk) {
    uint4 z; z.x = z.y = z.z = z.w = 0;
    CSH_UNROLL
    for (int j = 0; j < 8; j++) *reinterpret_cast<uint4 *>(blk + CSH_OCT_STRIDE * j) = z;
}

// ------------------------------------------------------------------------------------------------
// mode 0: full-resolution component, IDCT -> (crop + edge expand) -> FDCT -> quantise
template <bool DERING>
__global__ void __launch_bounds__(256) k_xform_direct(const ImgDesc *__restrict__ imgs, const PlaneWork *__restrict__ work, const DevQuant *__restrict__ quant,
                                                       const int16_t *__restrict__ coef_in, int16_t *__restrict__ coef_out, int16_t *__restrict__ dct_raw, uint32_t raw_tile0) {
    CSH_SHARED int16_t s_dr[64][256];   // deringing: a column per lane; then the wave's columns carry its retained-DCT blocks to whole-line stores (raw_copy_out)
    const PlaneWork w = work[blockIdx.y];
    if (w.mode != 0) return;
    const ImgDesc &im = imgs[w.image];
    const CompGeom gi = im.in[w.comp], go = im.out[w.comp];
    int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int b = tile * 64 + lane;
    bool has_raw = false;
    do {
        if (b >= go.bw * go.bh) break;
        int by = b / go.bw, bx = b - by * go.bw;
        int16_t *dst = coef_out + coef_index(go.tile_base, b, 0);
        if (by >= go.real_bh || bx >= go.real_bw) { store_zero_block(dst); break; }
        int x[64];
        load_idct<true>(coef_in + coef_index(gi.tile_base, by * gi.bw + bx, 0), quant[CSH_UNIFORM(im.qt_in[w.comp])], x);
        int vc = gi.comp_w - bx * 8, vr = gi.comp_h - by * 8;
        if (vc < 8 || vr < 8) replicate_edges(x, vc, vr);
        fdct_quant_store<DERING, true>(x, quant[CSH_UNIFORM(im.qt_out[w.comp])], dst, dct_raw ? dct_raw + raw_index(go.tile_base - raw_tile0, b) : nullptr, s_dr);
        has_raw = dct_raw != nullptr;
    } while (0);
    if (dct_raw) raw_copy_out(dct_raw + raw_index(go.tile_base - raw_tile0, tile * 64), has_raw, s_dr, int(threadIdx.x));
}

// mode 1 producer: subsampled component, IDCT -> u8 plane (pitch real_bw*8, rows real_bh*8, edges replicated)
__global__ void __launch_bounds__(256) k_idct_plane(const ImgDesc *__restrict__ imgs, const PlaneWork *__restrict__ work, const DevQuant *__restrict__ quant,
                                                     const int16_t *__restrict__ coef_in, uint8_t *__restrict__ planes) {
    const PlaneWork w = work[blockIdx.y];
    if (w.mode == 0) return;
    const ImgDesc &im = imgs[w.image];
    const CompGeom gi = im.in[w.comp];
    int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int b = tile * 64 + lane;
    if (b >= gi.bw * gi.bh) return;
    int by = b / gi.bw, bx = b - by * gi.bw;
    if (by >= gi.real_bh || bx >= gi.real_bw) return;
    int x[64];
    load_idct(coef_in + coef_index(gi.tile_base, b, 0), quant[CSH_UNIFORM(im.qt_in[w.comp])], x);
    int vc = gi.comp_w - bx * 8, vr = gi.comp_h - by * 8;
    if (vc < 8 || vr < 8) replicate_edges(x, vc, vr);
    int pitch = gi.real_bw * 8;
    uint8_t *p = planes + im.plane_off[w.comp] + size_t(by * 8) * pitch + bx * 8;
    CSH_UNROLL
    for (int r = 0; r < 8; r++) {
        uint32_t lo = uint32_t(x[8 * r]) | (uint32_t(x[8 * r + 1]) << 8) | (uint32_t(x[8 * r + 2]) << 16) | (uint32_t(x[8 * r + 3]) << 24);
        uint32_t hi = uint32_t(x[8 * r + 4]) | (uint32_t(x[8 * r + 5]) << 8) | (uint32_t(x[8 * r + 6]) << 16) | (uint32_t(x[8 * r + 7]) << 24);
        uint2 v; v.x = lo; v.y = hi;
        *reinterpret_cast<uint2 *>(p + size_t(r) * pitch) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// resampling consumers.  Decoder side: h2v2 "fancy" upsample on the REAL plane; encoder
// side: right/bottom edge expansion + h2v2 box downsample with the 1,2,1,2 bias.
struct PlaneView { const uint8_t *p; int pitch, cw, ch; };  // cw/ch: real component size (replicas beyond are equal)

__device__ __forceinline__ static int pv(const PlaneView &v, int y, int x) {
    y = y < 0 ? 0 : (y > v.ch - 1 ? v.ch - 1 : y);
    x = x < 0 ? 0 : (x > v.cw - 1 ? v.cw - 1 : x);
    return v.p[size_t(y) * v.pitch + x];
}
// full-resolution sample (r, xx) of an h2v2-subsampled plane after fancy upsampling
__device__ static int up_h2v2(const PlaneView &v, int r, int xx) {
    int cy = r >> 1, cx = xx >> 1;
    if (v.cw <= 2) return pv(v, cy, cx);  // libjpeg falls back to replication for tiny planes
    int fy = (r & 1) ? cy + 1 : cy - 1;
    int nb = (xx & 1) ? cx + 1 : cx - 1;
    int cs = 3 * pv(v, cy, cx) + pv(v, fy, cx);
    int cn = 3 * pv(v, cy, nb) + pv(v, fy, nb);
    return (3 * cs + cn + ((xx & 1) ? 7 : 8)) >> 4;
}

// full-resolution sample (r, xx) of an h2v1-subsampled plane after fancy upsampling (jdsample h2v1_fancy_upsample;
// its first/last-column special cases equal the replicate-clamped triangle filter)
__device__ static int up_h2v1(const PlaneView &v, int r, int xx) {
    int cx = xx >> 1;
    if (v.cw <= 2) return pv(v, r, cx);
    int nb = (xx & 1) ? cx + 1 : cx - 1;
    return (3 * pv(v, r, cx) + pv(v, r, nb) + ((xx & 1) ? 2 : 1)) >> 2;
}

// one sample of the encoder-side plane.  IN: how the decoded plane relates to full resolution (0 full, 1 h2v2 -> fancy
// upsample, 2 h2v1 -> fancy upsample); OUT: the encoder's downsampling (0 none, 1 h2v2 box with bias 1,2,1,2.., 2 h2v1 box with
// bias 0,1,0,1..).  libjpeg's edge rules: full-res columns beyond W replicate column W-1, full-res rows are
// padded to the row group by replication, and rows below the last DOWNSAMPLED row replicate that row.
template <int IN>
__device__ __forceinline__ static int fullres_sample(const PlaneView &v, int r, int xx) {
    return IN == 1 ? up_h2v2(v, r, xx) : (IN == 2 ? up_h2v1(v, r, xx) : pv(v, r, xx));
}
template <int IN, int OUT>
__device__ __forceinline__ static int resample_one(const PlaneView &v, int W, int H, int out_ch, int y, int xo) {
    int ye = y < out_ch - 1 ? y : out_ch - 1;
    const int HX = OUT ? 2 : 1, VX = OUT == 1 ? 2 : 1;
    int sum = 0;
    CSH_UNROLL
    for (int dy = 0; dy < VX; dy++) {
        CSH_UNROLL
        for (int dx = 0; dx < HX; dx++) {
            int r = VX * ye + dy, xx = HX * xo + dx;
            r = r > H - 1 ? H - 1 : r;
            xx = xx > W - 1 ? W - 1 : xx;
            sum += fullres_sample<IN>(v, r, xx);
        }
    }
    if (OUT == 1) return (sum + ((xo & 1) ? 2 : 1)) >> 2;
    if (OUT == 2) return (sum + (xo & 1)) >> 1;
    return sum;
}
template <int IN, int OUT>
__device__ __forceinline__ static uint32_t resample_quad(const PlaneView &v, int W, int H, int out_ch, int y, int x0) {
    uint32_t out = 0;
    CSH_UNROLL
    for (int i = 0; i < 4; i++) out |= uint32_t(resample_one<IN, OUT>(v, W, H, out_ch, y, x0 + i)) << (8 * i);
    return out;
}

// Two samples per 32-bit register (16-bit fields; every intermediate is < 4096, nothing carries across).  Of a quad of output
// columns: E = plane columns (0, 2), O = (1, 3), L = (-1, 1), R = (2, 4): even outputs have centre E and neighbours L and O, odd
// outputs have centre O and neighbours E and R -- so the bias pattern of both filters is constant per register.
struct Quad420 { uint32_t E, O, L, R; };
__device__ __forceinline__ static Quad420 quad420_fields(uint32_t a, uint32_t b, uint32_t d) {   // dwords at columns -4, 0, +4
    Quad420 f;
    f.E = b & 0x00FF00FFu;
    f.O = (b >> 8) & 0x00FF00FFu;
    f.L = (a >> 24) | (f.O << 16);
    f.R = (f.E >> 16) | ((d & 0xFFu) << 16);
    return f;
}
// sums over the two full-resolution rows 2y, 2y+1 of the upsampled samples under a quad of outputs: A = full-resolution columns
// 2x (x = 0, 2), B = columns 2x+1 (x = 0, 2), C = columns 2x (x = 1, 3), D = columns 2x+1 (x = 1, 3)
struct Sums420 { uint32_t A, B, C, D; };
__device__ __forceinline__ static Sums420 quad420_sums(const Quad420 &p, const Quad420 &c, const Quad420 &n) {
    // vertical step of the triangle filter (jdsample h2v2_fancy_upsample): 3 * nearer row + further row, for the two output
    // rows 2y (further = y-1) and 2y+1 (further = y+1)
    const uint32_t e0 = 3u * c.E + p.E, e1 = 3u * c.E + n.E, o0 = 3u * c.O + p.O, o1 = 3u * c.O + n.O;
    const uint32_t l0 = 3u * c.L + p.L, l1 = 3u * c.L + n.L, r0 = 3u * c.R + p.R, r1 = 3u * c.R + n.R;
    const uint32_t M = 0x00FF00FFu;
    // horizontal step: (3 * centre + left + 8) >> 4 and (3 * centre + right + 7) >> 4
    auto up = [&](uint32_t ce, uint32_t nb, uint32_t bias) { return ((3u * ce + nb + bias) >> 4) & M; };
    Sums420 s;
    s.A = up(e0, l0, 0x00080008u) + up(e1, l1, 0x00080008u);
    s.B = up(e0, o0, 0x00070007u) + up(e1, o1, 0x00070007u);
    s.C = up(o0, e0, 0x00080008u) + up(o1, e1, 0x00080008u);
    s.D = up(o0, r0, 0x00070007u) + up(o1, r1, 0x00070007u);
    return s;
}
// jcsample's h2v2 box: (sum of four + {1, 2, 1, 2 ..}) >> 2; even = outputs (0, 2), odd = outputs (1, 3), 16-bit fields
__device__ __forceinline__ static void quad420_finish(const Sums420 &s, uint32_t &even, uint32_t &odd) {
    even = ((s.A + s.B + 0x00010001u) >> 2) & 0x00FF00FFu;
    odd = ((s.C + s.D + 0x00020002u) >> 2) & 0x00FF00FFu;
}

// decoded plane -> encoder-side plane, 4 samples per lane (one dword store).  Interior quads of the 4:2:0 -> 4:2:0 case
// take a vector path (3 rows x 3 dwords in, composite triangle-up o box-down in registers); everything else goes through
// the same clamped per-sample formula, so image borders, odd sizes and tiny planes need no special code.
__device__ __forceinline__ static uint32_t resample_quad_420(const PlaneView &v, int rows_alloc, int y, int x0) {
    // window columns x0-1 .. x0+4 of plane rows y-1, y, y+1 (row index clamped; columns all inside the plane)
    Quad420 f[3];
    CSH_UNROLL
    for (int j = 0; j < 3; j++) {
        int yy = y - 1 + j;
        yy = yy < 0 ? 0 : (yy > rows_alloc - 1 ? rows_alloc - 1 : yy);
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(v.p + size_t(yy) * v.pitch + x0);
        f[j] = quad420_fields(rp[-1], rp[0], rp[1]);
    }
    uint32_t even, odd;
    quad420_finish(quad420_sums(f[0], f[1], f[2]), even, odd);
    return even | (odd << 8);
}

__global__ void __launch_bounds__(256) k_resample_plane(const ImgDesc *imgs, const PlaneWork *work, const uint8_t *planes, uint8_t *oplanes) {
    const PlaneWork w = work[blockIdx.y];
    if (w.mode == 0 || w.mode == 10) return;
    const ImgDesc &im = imgs[w.image];
    const CompGeom gi = im.src[w.comp], go = im.out[w.comp];
    const int pitch_o = go.real_bw * 8, rows_o = go.real_bh * 8;
    int q = blockIdx.x * blockDim.x + threadIdx.x;  // quad index
    if (q >= (pitch_o >> 2) * rows_o) return;
    int y = q / (pitch_o >> 2), x0 = (q - y * (pitch_o >> 2)) * 4;
    PlaneView v;
    v.p = planes + im.splane_off[w.comp]; v.pitch = gi.real_bw * 8; v.cw = gi.comp_w; v.ch = gi.comp_h;
    uint32_t out = 0;
    // vector path <=> no full-resolution clamp under the quad and its column neighbours lie inside the decoded plane
    const int W = im.enc_w, H = im.enc_h, och = go.comp_h;
    const int in_kind = (w.mode - 1) / 3, out_kind = (w.mode - 1) % 3;   // PlaneWork.mode = 1 + 3*in + out
    if (in_kind == 1 && out_kind == 1 && x0 >= 4 && x0 + 8 <= v.pitch && 2 * (x0 + 3) + 1 <= W - 1 && 2 * y + 1 <= H - 1 && gi.comp_w > 2) {
        out = resample_quad_420(v, gi.real_bh * 8, y, x0);
    } else {
        switch (w.mode) {
        case 1: out = resample_quad<0, 0>(v, W, H, och, y, x0); break;   // resize path: full-resolution plane, edge expansion only
        case 2: out = resample_quad<0, 1>(v, W, H, och, y, x0); break;
        case 3: out = resample_quad<0, 2>(v, W, H, och, y, x0); break;
        case 4: out = resample_quad<1, 0>(v, W, H, och, y, x0); break;
        case 5: out = resample_quad<1, 1>(v, W, H, och, y, x0); break;
        case 6: out = resample_quad<1, 2>(v, W, H, och, y, x0); break;
        case 7: out = resample_quad<2, 0>(v, W, H, och, y, x0); break;
        case 8: out = resample_quad<2, 1>(v, W, H, och, y, x0); break;
        default: out = resample_quad<2, 2>(v, W, H, och, y, x0); break;
        }
    }
    *reinterpret_cast<uint32_t *>(oplanes + im.oplane_off[w.comp] + size_t(y) * pitch_o + x0) = out;
}

// encoder-side plane -> FDCT -> quantise, one block per lane
template <bool DERING>
__global__ void __launch_bounds__(256) k_plane_fdct(const ImgDesc *__restrict__ imgs, const PlaneWork *__restrict__ work, const DevQuant *__restrict__ quant, const uint8_t *__restrict__ oplanes,
                                                     int16_t *__restrict__ coef_out, int16_t *__restrict__ dct_raw, uint32_t raw_tile0) {
    CSH_SHARED int16_t s_dr[64][256];   // deringing: a column per lane; then the wave's columns carry its retained-DCT blocks to whole-line stores (raw_copy_out)
    const PlaneWork w = work[blockIdx.y];
    if (w.mode == 0 || w.mode == 10) return;
    const ImgDesc &im = imgs[w.image];
    const CompGeom go = im.out[w.comp];
    int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int b = tile * 64 + lane;
    bool has_raw = false;
    do {
    if (b >= go.bw * go.bh) break;
    int by = b / go.bw, bx = b - by * go.bw;
    int16_t *dst = coef_out + coef_index(go.tile_base, b, 0);
    if (by >= go.real_bh || bx >= go.real_bw) { store_zero_block(dst); break; }
    const int pitch = go.real_bw * 8;
    const uint8_t *p = oplanes + im.oplane_off[w.comp] + size_t(by * 8) * pitch + bx * 8;
    uint32_t pr[8][4];
    CSH_UNROLL
    for (int r = 0; r < 8; r++) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p + size_t(r) * pitch);
        pr[r][0] = bytes_to_halves<0, 1>(v.x); pr[r][2] = bytes_to_halves<2, 3>(v.x); pr[r][1] = bytes_to_halves<3, 2>(v.y); pr[r][3] = bytes_to_halves<1, 0>(v.y);
    }
    fdct_quant_store_pk<DERING, false>(pr, quant[CSH_UNIFORM(im.qt_out[w.comp])], dst, dct_raw ? dct_raw + raw_index(go.tile_base - raw_tile0, b) : nullptr, s_dr);
    has_raw = dct_raw != nullptr;
    } while (0);
    if (dct_raw) raw_copy_out(dct_raw + raw_index(go.tile_base - raw_tile0, tile * 64), has_raw, s_dr, int(threadIdx.x));
}

// sixteen bytes of a plane row from a 4-byte aligned address (x0 - 4): one global_load_dwordx4
struct Row16 { uint32_t a, b, c, d; };
__device__ __forceinline__ static Row16 load_row16(const uint8_t *p) {
    Row16 r;
#ifdef CSH_EMUL
    __builtin_memcpy(&r, p, 16);
#else
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p);
    r.a = v.x; r.b = v.y; r.c = v.z; r.d = v.w;
#endif
    return r;
}
// The camera case in one pass: 4:2:0 in, 4:2:0 out, no resize (PlaneWork.mode 10).  One lane per OUTPUT block: a 10 x 10 window of
// the decoded plane (4 dwords x 10 rows) -> fancy upsample o box downsample in packed registers -> FDCT -> quantise.  The
// encoder-side plane is never written.  libjpeg's edge rules are applied where they bite:
//   rows:    an odd H makes full-resolution row 2y+1 of the last output row a copy of row 2y -- the window row below it is
//            loaded from plane row y-1 instead; output rows below the last one replicate it;
//   columns: full-resolution columns beyond W-1 replicate column W-1 -- the per-column sums are patched in the block that
//            holds plane column (W-1)/2 (always the last block column).
template <bool DERING>
__global__ void __launch_bounds__(256) k_resample_fdct_420(const ImgDesc *__restrict__ imgs, const PlaneWork *__restrict__ work, const DevQuant *__restrict__ quant, const uint8_t *__restrict__ planes,
                                                            int16_t *__restrict__ coef_out, int16_t *__restrict__ dct_raw, uint32_t raw_tile0) {
    CSH_SHARED int16_t s_dr[64][256];   // deringing: a column per lane; then the wave's columns carry its retained-DCT blocks to whole-line stores (raw_copy_out)
    const PlaneWork w = work[blockIdx.y];
    if (w.mode != 10) return;
    const ImgDesc &im = imgs[w.image];
    const CompGeom gi = im.src[w.comp], go = im.out[w.comp];
    int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int b = tile * 64 + lane;
    bool has_raw = false;
    do {
    if (b >= go.bw * go.bh) break;
    int by = b / go.bw, bx = b - by * go.bw;
    int16_t *dst = coef_out + coef_index(go.tile_base, b, 0);
    if (by >= go.real_bh || bx >= go.real_bw) { store_zero_block(dst); break; }
    const int pitch = gi.real_bw * 8, rows_alloc = gi.real_bh * 8;
    const uint8_t *pl = planes + im.splane_off[w.comp];
    const int x0 = bx * 8, y0 = by * 8, W = im.enc_w, H = im.enc_h, och = go.comp_h;
    const bool first = bx == 0, last = x0 + 12 > pitch, hodd = (H & 1) != 0;
    const int kt = ((W - 1) >> 1) - x0;        // block-relative plane column that holds full-resolution column W-1 (< 8: patch)
    // which 16-bit fields of A/B/C/D are beyond W-1, per quad (lo field = outputs 0 / 1, hi field = outputs 2 / 3)
    uint32_t mA[2], mB[2], mC[2], mD[2];
    const bool wodd = (W & 1) != 0;
    CSH_UNROLL
    for (int q = 0; q < 2; q++) {
        auto even_gone = [&](int k) { return k > kt ? 0xFFFFu : 0u; };                  // 2k > W-1
        auto odd_gone = [&](int k) { return (k > kt || (wodd && k == kt)) ? 0xFFFFu : 0u; };   // 2k+1 > W-1
        mA[q] = even_gone(4 * q) | (even_gone(4 * q + 2) << 16);
        mB[q] = odd_gone(4 * q) | (odd_gone(4 * q + 2) << 16);
        mC[q] = even_gone(4 * q + 1) | (even_gone(4 * q + 3) << 16);
        mD[q] = odd_gone(4 * q + 1) | (odd_gone(4 * q + 3) << 16);
    }
    // the whole 10 x 16-byte window first (one wait for memory, not ten): row j = plane row y0 - 1 + j, bytes x0 - 4 .. x0 + 11.  The first
    // block of a row reads four bytes in front of its row and the last four bytes behind it (the neighbouring rows' ends; the plane pool has 64
    // bytes of slack at either end, pipeline.cpp) and replaces them with its own edge sample.
    Row16 rows[10];
    CSH_UNROLL
    for (int j = 0; j < 10; j++) {
        int yy = y0 - 1 + j;
        if (hodd && yy == och) yy = och - 2;
        yy = yy < 0 ? 0 : (yy > rows_alloc - 1 ? rows_alloc - 1 : yy);
        rows[j] = load_row16(pl + size_t(yy) * pitch + x0 - 4);
    }
    // does any lane of the wave sit on the picture's right edge with columns to replicate?  (1080p: none -- W is even and ends with its block)
    const bool patch = kt < 8 && ((mA[0] | mB[0] | mC[0] | mD[0] | mA[1] | mB[1] | mC[1] | mD[1]) != 0u);
#ifdef CSH_EMUL
    const bool any_patch = patch;
#else
    const bool any_patch = __ballot(patch) != 0ull;
#endif
    uint32_t pr[8][4];
    Quad420 win[3][2];
    auto row_fields = [&](int j, Quad420 out[2]) {
        const uint32_t bq = rows[j].b, cq = rows[j].c;
        const uint32_t aq = first ? bq << 24 : rows[j].a;     // plane column -1 = column 0
        const uint32_t dq = last ? cq >> 24 : rows[j].d;      // plane column `pitch` = column pitch-1
        out[0] = quad420_fields(aq, bq, cq);
        out[1] = quad420_fields(bq, cq, dq);
    };
    row_fields(0, win[0]);
    row_fields(1, win[1]);
    CSH_UNROLL
    for (int r = 0; r < 8; r++) {
        Quad420 *p = win[r % 3], *c = win[(r + 1) % 3], *n = win[(r + 2) % 3];
        row_fields(r + 2, n);
        Sums420 s[2] = {quad420_sums(p[0], c[0], n[0]), quad420_sums(p[1], c[1], n[1])};
        if (any_patch && patch) {
            // the sum at full-resolution column W-1: plane column kt, the 2x+1 sample when W is even
            const int odd_col = kt & 1;
            const Sums420 &sq = (kt >> 2) ? s[1] : s[0];   // (selects, not a run-time index: that would put the sums in scratch memory)
            const uint32_t sqA = (kt >> 2) ? s[1].A : s[0].A, sqB = (kt >> 2) ? s[1].B : s[0].B, sqC = (kt >> 2) ? s[1].C : s[0].C, sqD = (kt >> 2) ? s[1].D : s[0].D;
            (void)sq;
            uint32_t sl = wodd ? (odd_col ? sqC : sqA) : (odd_col ? sqD : sqB);
            sl = (kt & 2) ? sl >> 16 : sl & 0xFFFFu;
            sl |= sl << 16;
            CSH_UNROLL
            for (int qq = 0; qq < 2; qq++) {
                s[qq].A = (s[qq].A & ~mA[qq]) | (sl & mA[qq]);
                s[qq].B = (s[qq].B & ~mB[qq]) | (sl & mB[qq]);
                s[qq].C = (s[qq].C & ~mC[qq]) | (sl & mC[qq]);
                s[qq].D = (s[qq].D & ~mD[qq]) | (sl & mD[qq]);
            }
        }
        // the row as the forward transform takes it (fdct1d_pk): the fields (d0, d2) (d1, d3) (d4, d6) (d5, d7) re-paired, never unpacked
        uint32_t e0, o0, e1, o1;
        quad420_finish(s[0], e0, o0);
        quad420_finish(s[1], e1, o1);
        pr[r][0] = pack_halves(e0, o0); pr[r][2] = pack_hi_halves(e0, o0); pr[r][1] = pack_hi_halves(o1, e1); pr[r][3] = pack_halves(o1, e1);
    }
    const int ylast = och - 1 - y0;   // rows below the last downsampled row replicate it
    if (ylast < 7) {
        CSH_UNROLL
        for (int r = 1; r < 8; r++)
            if (r > ylast) {
                CSH_UNROLL
                for (int cc = 0; cc < 4; cc++) pr[r][cc] = pr[r - 1][cc];
            }
    }
    fdct_quant_store_pk<DERING, false>(pr, quant[CSH_UNIFORM(im.qt_out[w.comp])], dst, dct_raw ? dct_raw + raw_index(go.tile_base - raw_tile0, b) : nullptr, s_dr);
    has_raw = dct_raw != nullptr;
    } while (0);
    if (dct_raw) raw_copy_out(dct_raw + raw_index(go.tile_base - raw_tile0, tile * 64), has_raw, s_dr, int(threadIdx.x));
}

// size-targeting: re-quantise every retained DCT block with the image's CURRENT output table (one block per lane)
__global__ void __launch_bounds__(256) k_requant(const ImgDesc *__restrict__ imgs, const PlaneWork *__restrict__ work, const DevQuant *__restrict__ quant, const int16_t *__restrict__ dct_raw, uint32_t raw_tile0,
                                                  int16_t *__restrict__ coef_out) {
    const PlaneWork w = work[blockIdx.y];
    const ImgDesc &im = imgs[w.image];
    const CompGeom go = im.out[w.comp];
    int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int b = tile * 64 + lane;
    if (b >= go.bw * go.bh) return;
    int by = b / go.bw, bx = b - by * go.bw;
    if (by >= go.real_bh || bx >= go.real_bw) return;   // dummy blocks: k_fix_dummy
    requant_block(dct_raw + raw_index(go.tile_base - raw_tile0, b), quant[CSH_UNIFORM(im.qt_out[w.comp])], coef_out + coef_index(go.tile_base, b, 0), Oct());
}

// dummy blocks (exist only to complete an MCU): zero AC, DC copied per libjpeg's jccoefct rule
__global__ void k_fix_dummy(const ImgDesc *imgs, int nimg, int16_t *coef_out) {
    int i = blockIdx.y;
    if (i >= nimg) return;
    const ImgDesc &im = imgs[i];
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int c = 0; c < im.ncomp; c++) {
        const CompGeom &g = im.out[c];
        int ndummy_cols = g.bw - g.real_bw, ndummy_rows = g.bh - g.real_bh;
        int n_right = ndummy_cols * g.real_bh, n_bottom = ndummy_rows * g.bw;
        if (t < n_right) {
            int by = t / ndummy_cols, bx = g.real_bw + t % ndummy_cols;
            coef_out[coef_index(g.tile_base, by * g.bw + bx, 0)] = coef_out[coef_index(g.tile_base, by * g.bw + g.real_bw - 1, 0)];
        } else if (t - n_right < n_bottom) {
            int u = t - n_right;
            int by = g.real_bh + u / g.bw, bx = u % g.bw;
            int sx = (bx / g.h) * g.h + g.h - 1;
            if (sx > g.real_bw - 1) sx = g.real_bw - 1;
            coef_out[coef_index(g.tile_base, by * g.bw + bx, 0)] = coef_out[coef_index(g.tile_base, (g.real_bh - 1) * g.bw + sx, 0)];
        }
    }
}

static dim3 tile_grid(int max_tiles, int nwork) { return dim3((max_tiles + 3) / 4, nwork); }

void launch_xform_direct(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                         const int16_t *coef_in, int16_t *coef_out, int16_t *dct_raw, uint32_t raw_tile0, bool dering) {
    if (!nwork) return;
    if (dering) CSH_LAUNCH(k_xform_direct<true>, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, coef_in, coef_out, dct_raw, raw_tile0);
    else CSH_LAUNCH(k_xform_direct<false>, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, coef_in, coef_out, dct_raw, raw_tile0);
}
void launch_idct_plane(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                       const int16_t *coef_in, uint8_t *planes) {
    if (nwork) CSH_LAUNCH(k_idct_plane, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, coef_in, planes);
}
void launch_resample_plane(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, uint32_t max_quads, const uint8_t *planes, uint8_t *oplanes) {
    if (nwork && max_quads) CSH_LAUNCH(k_resample_plane, dim3((max_quads + 255) / 256, nwork), dim3(256), st, imgs, work, planes, oplanes);
}
void launch_plane_fdct(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                       const uint8_t *oplanes, int16_t *coef_out, int16_t *dct_raw, uint32_t raw_tile0, bool dering) {
    if (!nwork) return;
    if (dering) CSH_LAUNCH(k_plane_fdct<true>, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, oplanes, coef_out, dct_raw, raw_tile0);
    else CSH_LAUNCH(k_plane_fdct<false>, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, oplanes, coef_out, dct_raw, raw_tile0);
}
void launch_resample_fdct_420(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                              const uint8_t *planes, int16_t *coef_out, int16_t *dct_raw, uint32_t raw_tile0, bool dering) {
    if (!nwork) return;
    if (dering) CSH_LAUNCH(k_resample_fdct_420<true>, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, planes, coef_out, dct_raw, raw_tile0);
    else CSH_LAUNCH(k_resample_fdct_420<false>, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, planes, coef_out, dct_raw, raw_tile0);
}
void launch_requant(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant, const int16_t *dct_raw,
                    uint32_t raw_tile0, int16_t *coef_out) {
    if (nwork) CSH_LAUNCH(k_requant, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, dct_raw, raw_tile0, coef_out);
}
void launch_fix_dummy(hipStream_t st, const ImgDesc *imgs, int nimg, int max_blocks, int16_t *coef_out) {
    if (nimg && max_blocks) CSH_LAUNCH(k_fix_dummy, dim3((max_blocks + 255) / 256, nimg), dim3(256), st, imgs, nimg, coef_out);
}

}  // namespace csh
