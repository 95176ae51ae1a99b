// k_pixel.hip -- phase 1: the pixel-domain transcode, one 8x8 block per lane, whole block in VGPRs.
//
// Replaces, for libcaesium's lossy JPEG path (reference call site /root/reference/src/compressor.rs:305;
// SURVEY.md 8a rows J2,J3,J5,J6,J7 and Appendix B.2-B.6), mozjpeg's
//   jidctint (dequantise + ISLOW IDCT + range limit), jdsample h2v2 fancy upsample,
//   jcsample h2v2 downsample with edge expansion, jfdctint (ISLOW FDCT) and the scalar quantiser.
// All arithmetic is int32 exactly as the oracle's (oracle/jpeg_oracle.c); multiplies by the 13-bit DCT
// constants use the full-rate 24-bit multiplier, which is exact whenever dequantised coefficients are
// below 2^15 in magnitude (every stream made from 8-bit samples; libjpeg-turbo's SIMD IDCT has the same
// domain).  Memory: tile rows are 128-byte coalesced (2 B per lane per row), planes are written/read in
// row segments contiguous across the wave's adjacent blocks.  No LDS, no cross-lane traffic: the 2-D
// transforms never leave the lane's registers, so zig-zag <-> natural reordering is free.
#include "kernels.h"

namespace csh {

// natural index -> zig-zag index (inverse of T.81 Figure A.6)
__device__ static const uint8_t kN2Z[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                            41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                            46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

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
#define DESC(x, n) (((x) + (1 << ((n)-1))) >> (n))

// one 1-D inverse transform of jidctint; in: frequency order, out: sample order
__device__ __forceinline__ static void idct1d(int &x0, int &x1, int &x2, int &x3, int &x4, int &x5, int &x6, int &x7, const int sh) {
    int z1 = MUL(x2 + x6, FIX_0_541), tmp2 = z1 - MUL(x6, FIX_1_847), tmp3 = z1 + MUL(x2, FIX_0_765);
    int tmp0 = (x0 + x4) * 8192, tmp1 = (x0 - x4) * 8192;
    int t10 = tmp0 + tmp3, t13 = tmp0 - tmp3, t11 = tmp1 + tmp2, t12 = tmp1 - tmp2;
    int a0 = x7, a1 = x5, a2 = x3, a3 = x1;
    int y1 = a0 + a3, y2 = a1 + a2, y3 = a0 + a2, y4 = a1 + a3, y5 = MUL(y3 + y4, FIX_1_175);
    a0 = MUL(a0, FIX_0_298); a1 = MUL(a1, FIX_2_053); a2 = MUL(a2, FIX_3_072); a3 = MUL(a3, FIX_1_501);
    y1 = MUL(y1, -FIX_0_899); y2 = MUL(y2, -FIX_2_562); y3 = MUL(y3, -FIX_1_961) + y5; y4 = MUL(y4, -FIX_0_390) + y5;
    a0 += y1 + y3; a1 += y2 + y4; a2 += y2 + y3; a3 += y1 + y4;
    x0 = DESC(t10 + a3, sh); x7 = DESC(t10 - a3, sh);
    x1 = DESC(t11 + a2, sh); x6 = DESC(t11 - a2, sh);
    x2 = DESC(t12 + a1, sh); x5 = DESC(t12 - a1, sh);
    x3 = DESC(t13 + a0, sh); x4 = DESC(t13 - a0, sh);
}

// one 1-D forward transform of jfdctint; FIRST: row pass (<<2, descale 11), else column pass (descale 2 / 15)
template <bool FIRST>
__device__ __forceinline__ static void fdct1d(int &d0, int &d1, int &d2, int &d3, int &d4, int &d5, int &d6, int &d7) {
    int tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
    int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    const int SH = FIRST ? 11 : 15;
    int o0, o4;
    if (FIRST) { o0 = (tmp10 + tmp11) * 4; o4 = (tmp10 - tmp11) * 4; }
    else { o0 = DESC(tmp10 + tmp11, 2); o4 = DESC(tmp10 - tmp11, 2); }
    int z1 = MUL(tmp12 + tmp13, FIX_0_541);
    int o2 = DESC(z1 + MUL(tmp13, FIX_0_765), SH), o6 = DESC(z1 - MUL(tmp12, FIX_1_847), SH);
    int y1 = tmp4 + tmp7, y2 = tmp5 + tmp6, y3 = tmp4 + tmp6, y4 = tmp5 + tmp7, y5 = MUL(y3 + y4, FIX_1_175);
    tmp4 = MUL(tmp4, FIX_0_298); tmp5 = MUL(tmp5, FIX_2_053); tmp6 = MUL(tmp6, FIX_3_072); tmp7 = MUL(tmp7, FIX_1_501);
    y1 = MUL(y1, -FIX_0_899); y2 = MUL(y2, -FIX_2_562); y3 = MUL(y3, -FIX_1_961) + y5; y4 = MUL(y4, -FIX_0_390) + y5;
    d0 = o0; d4 = o4; d2 = o2; d6 = o6;
    d7 = DESC(tmp4 + y1 + y3, SH); d5 = DESC(tmp5 + y2 + y4, SH);
    d3 = DESC(tmp6 + y2 + y3, SH); d1 = DESC(tmp7 + y1 + y4, SH);
}

// load + dequantise + 2-D IDCT + level shift + range limit; x[] natural order in, samples (0..255) out
__device__ __forceinline__ static void load_idct(const int16_t *__restrict__ blk, const DevQuant &q, int x[64]) {
    CSH_UNROLL
    for (int n = 0; n < 64; n++) { int k = kN2Z[n]; x[n] = int(blk[k << 6]) * int(q.q[k]); }
    CSH_UNROLL
    for (int c = 0; c < 8; c++) idct1d(x[c], x[8 + c], x[16 + c], x[24 + c], x[32 + c], x[40 + c], x[48 + c], x[56 + c], 11);
    CSH_UNROLL
    for (int r = 0; r < 8; r++) idct1d(x[8 * r], x[8 * r + 1], x[8 * r + 2], x[8 * r + 3], x[8 * r + 4], x[8 * r + 5], x[8 * r + 6], x[8 * r + 7], 18);
    CSH_UNROLL
    for (int n = 0; n < 64; n++) { int v = x[n] + 128; x[n] = v < 0 ? 0 : (v > 255 ? 255 : v); }
}

// replicate the last valid column / row of an edge block (decoder crop + encoder edge expansion)
__device__ __forceinline__ static void replicate_edges(int x[64], int vc, int vr) {
    if (vc < 8) {
        CSH_UNROLL
        for (int r = 0; r < 8; r++) {
            int last = x[8 * r];
            CSH_UNROLL
            for (int c = 1; c < 8; c++) last = (c < vc) ? x[8 * r + c] : last;
            CSH_UNROLL
            for (int c = 1; c < 8; c++) x[8 * r + c] = (c < vc) ? x[8 * r + c] : last;
        }
    }
    if (vr < 8) {
        CSH_UNROLL
        for (int c = 0; c < 8; c++) {
            int last = x[c];
            CSH_UNROLL
            for (int r = 1; r < 8; r++) last = (r < vr) ? x[8 * r + c] : last;
            CSH_UNROLL
            for (int r = 1; r < 8; r++) x[8 * r + c] = (r < vr) ? x[8 * r + c] : last;
        }
    }
}

// samples (0..255, natural order) -> level shift -> 2-D FDCT -> scalar quantise -> store zig-zag rows
__device__ __forceinline__ static void fdct_quant_store(int x[64], const DevQuant &q, int16_t *__restrict__ blk) {
    CSH_UNROLL
    for (int n = 0; n < 64; n++) x[n] -= 128;
    CSH_UNROLL
    for (int r = 0; r < 8; r++) fdct1d<true>(x[8 * r], x[8 * r + 1], x[8 * r + 2], x[8 * r + 3], x[8 * r + 4], x[8 * r + 5], x[8 * r + 6], x[8 * r + 7]);
    CSH_UNROLL
    for (int c = 0; c < 8; c++) fdct1d<false>(x[c], x[8 + c], x[16 + c], x[24 + c], x[32 + c], x[40 + c], x[48 + c], x[56 + c]);
    CSH_UNROLL
    for (int n = 0; n < 64; n++) {
        int k = kN2Z[n];
        int d = q.div[k];
        int t = x[n], a = t < 0 ? -t : t;
        a += d >> 1;
        // exact a/d: float estimate (a < 2^24) with one correction step
        int qv = int(float(a) * q.rcp[k]);
        int r = a - qv * d;
        qv += (r >= d) ? 1 : 0;
        qv -= (r < 0) ? 1 : 0;
        blk[k << 6] = int16_t(t < 0 ? -qv : qv);
    }
}

__device__ __forceinline__ static void store_zero_block(int16_t *__restrict__ blk) {
    CSH_UNROLL
    for (int k = 0; k < 64; k++) blk[k << 6] = 0;
}

// ------------------------------------------------------------------------------------------------
// mode 0: full-resolution component, IDCT -> (crop + edge expand) -> FDCT -> quantise
__global__ void __launch_bounds__(256) k_xform_direct(const ImgDesc *imgs, const PlaneWork *work, const DevQuant *quant,
                                                       const int16_t *coef_in, int16_t *coef_out) {
    const PlaneWork w = work[blockIdx.y];
    if (w.mode != 0) return;
    const ImgDesc &im = imgs[w.image];
    const CompGeom gi = im.in[w.comp], go = im.out[w.comp];
    int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int b = tile * 64 + lane;
    if (b >= go.bw * go.bh) return;
    int by = b / go.bw, bx = b - by * go.bw;
    int16_t *dst = coef_out + coef_index(go.tile_base, b, 0);
    if (by >= go.real_bh || bx >= go.real_bw) { store_zero_block(dst); return; }
    int x[64];
    load_idct(coef_in + coef_index(gi.tile_base, by * gi.bw + bx, 0), quant[im.qt_in[w.comp]], x);
    int vc = gi.comp_w - bx * 8, vr = gi.comp_h - by * 8;
    if (vc < 8 || vr < 8) replicate_edges(x, vc, vr);
    fdct_quant_store(x, quant[im.qt_out[w.comp]], dst);
}

// mode 1 producer: subsampled component, IDCT -> u8 plane (pitch real_bw*8, rows real_bh*8, edges replicated)
__global__ void __launch_bounds__(256) k_idct_plane(const ImgDesc *imgs, const PlaneWork *work, const DevQuant *quant,
                                                     const int16_t *coef_in, uint8_t *planes) {
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
    load_idct(coef_in + coef_index(gi.tile_base, b, 0), quant[im.qt_in[w.comp]], x);
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

// generic (edge-block) path: out(y,x) = box of the four full-res samples under it, with libjpeg's clamps
template <int MODE>
__device__ static void resample_block_slow(const PlaneView &v, int W, int H, int out_ch, int by, int bx, int x[64]) {
    for (int Y = 0; Y < 8; Y++) {
        int y = by * 8 + Y;
        int ye = y < out_ch - 1 ? y : out_ch - 1;  // rows below the last downsampled row replicate it
        for (int X = 0; X < 8; X++) {
            int xo = bx * 8 + X, sum = 0;
            for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                    int r = 2 * ye + dy, xx = 2 * xo + dx;
                    r = r > H - 1 ? H - 1 : r;
                    xx = xx > W - 1 ? W - 1 : xx;
                    sum += (MODE == 2) ? up_h2v2(v, r, xx) : (MODE == 4 ? up_h2v1(v, r, xx) : pv(v, r, xx));
                }
            x[8 * Y + X] = (sum + ((xo & 1) ? 2 : 1)) >> 2;
        }
    }
}

__device__ __forceinline__ static int byte_of(const uint4 &q, int i) {  // i compile-time after unrolling
    uint32_t w = i < 4 ? q.x : (i < 8 ? q.y : (i < 12 ? q.z : q.w));
    return int((w >> (8 * (i & 3))) & 255u);
}

// mode 2 interior path: 10x10 window of the subsampled plane -> composite fancy-up + box-down
__device__ __forceinline__ static void resample_block_420(const PlaneView &v, int rows_alloc, int by, int bx, int x[64]) {
    // window row j <-> plane row by*8-1+j ; window col i <-> plane col bx*8-1+i
    int c0[10], c1[10], c2[10];  // three consecutive window rows
    int cs0[10], cs1[10];
    const int xoff = bx > 0 ? bx * 8 - 4 : 0;
    const int sh = bx > 0 ? 3 : -1;  // window col i sits at byte (i + sh) of the 16-byte load
    auto load_row = [&](int j, int out[10]) {
        int y = by * 8 - 1 + j;
        y = y < 0 ? 0 : (y > rows_alloc - 1 ? rows_alloc - 1 : y);
        const uint8_t *rp = v.p + size_t(y) * v.pitch + xoff;
        uint4 q;
        q.x = reinterpret_cast<const uint32_t *>(rp)[0]; q.y = reinterpret_cast<const uint32_t *>(rp)[1];
        q.z = reinterpret_cast<const uint32_t *>(rp)[2]; q.w = reinterpret_cast<const uint32_t *>(rp)[3];
        if (bx > 0) {
            CSH_UNROLL
            for (int i = 0; i < 10; i++) out[i] = byte_of(q, i + 3);
        } else {
            out[0] = byte_of(q, 0);
            CSH_UNROLL
            for (int i = 1; i < 10; i++) out[i] = byte_of(q, i - 1);
        }
        if (bx * 8 + 8 > v.pitch - 1) out[9] = out[8];  // right neighbour outside the plane: replicate
    };
    (void)sh;
    load_row(0, c0); load_row(1, c1);
    CSH_UNROLL
    for (int Y = 0; Y < 8; Y++) {
        load_row(Y + 2, c2);
        CSH_UNROLL
        for (int i = 0; i < 10; i++) { cs0[i] = 3 * c1[i] + c0[i]; cs1[i] = 3 * c1[i] + c2[i]; }
        CSH_UNROLL
        for (int X = 0; X < 8; X++) {
            int i = X + 1;
            int u00 = (3 * cs0[i] + cs0[i - 1] + 8) >> 4, u01 = (3 * cs0[i] + cs0[i + 1] + 7) >> 4;
            int u10 = (3 * cs1[i] + cs1[i - 1] + 8) >> 4, u11 = (3 * cs1[i] + cs1[i + 1] + 7) >> 4;
            x[8 * Y + X] = (u00 + u01 + u10 + u11 + ((X & 1) ? 2 : 1)) >> 2;
        }
        CSH_UNROLL
        for (int i = 0; i < 10; i++) { c0[i] = c1[i]; c1[i] = c2[i]; }
    }
}

// mode 3 interior path: 16x16 full-resolution samples -> h2v2 box
__device__ __forceinline__ static void resample_block_box(const PlaneView &v, int by, int bx, int x[64]) {
    CSH_UNROLL
    for (int Y = 0; Y < 8; Y++) {
        const uint8_t *r0 = v.p + size_t(by * 16 + 2 * Y) * v.pitch + bx * 16;
        uint4 a = *reinterpret_cast<const uint4 *>(r0), b = *reinterpret_cast<const uint4 *>(r0 + v.pitch);
        CSH_UNROLL
        for (int X = 0; X < 8; X++)
            x[8 * Y + X] = (byte_of(a, 2 * X) + byte_of(a, 2 * X + 1) + byte_of(b, 2 * X) + byte_of(b, 2 * X + 1) + ((X & 1) ? 2 : 1)) >> 2;
    }
}

__global__ void __launch_bounds__(256) k_resample_fdct(const ImgDesc *imgs, const PlaneWork *work, const DevQuant *quant,
                                                        const uint8_t *planes, int16_t *coef_out) {
    const PlaneWork w = work[blockIdx.y];
    if (w.mode == 0) return;
    const ImgDesc &im = imgs[w.image];
    const CompGeom gi = im.in[w.comp], go = im.out[w.comp];
    int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int b = tile * 64 + lane;
    if (b >= go.bw * go.bh) return;
    int by = b / go.bw, bx = b - by * go.bw;
    int16_t *dst = coef_out + coef_index(go.tile_base, b, 0);
    if (by >= go.real_bh || bx >= go.real_bw) { store_zero_block(dst); return; }
    PlaneView v;
    v.p = planes + im.plane_off[w.comp]; v.pitch = gi.real_bw * 8; v.cw = gi.comp_w; v.ch = gi.comp_h;
    int x[64];
    const int W = im.width, H = im.height;
    // interior <=> none of the 16x16 full-resolution samples under this block is clamped
    bool interior = (16 * bx + 15 <= W - 1) && (16 * by + 15 <= H - 1);
    if (w.mode == 2) {
        if (interior && gi.comp_w > 2) resample_block_420(v, gi.real_bh * 8, by, bx, x);
        else resample_block_slow<2>(v, W, H, go.comp_h, by, bx, x);
    } else if (w.mode == 3) {
        if (interior && ((v.pitch & 15) == 0)) resample_block_box(v, by, bx, x);
        else resample_block_slow<3>(v, W, H, go.comp_h, by, bx, x);
    } else resample_block_slow<4>(v, W, H, go.comp_h, by, bx, x);  // 4:2:2 source: generic path only (rare input)
    fdct_quant_store(x, quant[im.qt_out[w.comp]], dst);
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
                         const int16_t *coef_in, int16_t *coef_out) {
    if (nwork) CSH_LAUNCH(k_xform_direct, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, coef_in, coef_out);
}
void launch_idct_plane(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                       const int16_t *coef_in, uint8_t *planes) {
    if (nwork) CSH_LAUNCH(k_idct_plane, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, coef_in, planes);
}
void launch_resample_fdct(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                          const uint8_t *planes, int16_t *coef_out) {
    if (nwork) CSH_LAUNCH(k_resample_fdct, tile_grid(max_tiles, nwork), dim3(256), st, imgs, work, quant, planes, coef_out);
}
void launch_fix_dummy(hipStream_t st, const ImgDesc *imgs, int nimg, int max_blocks, int16_t *coef_out) {
    if (nimg && max_blocks) CSH_LAUNCH(k_fix_dummy, dim3((max_blocks + 255) / 256, nimg), dim3(256), st, imgs, nimg, coef_out);
}

}  // namespace csh
