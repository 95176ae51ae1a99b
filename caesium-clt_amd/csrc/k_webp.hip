// k_webp.hip -- the lossy WebP row on the device; statement: oracle/webp_oracle.c.
//   k_webp_yuv   W1: RGB -> YUV 4:2:0 planes padded to whole macroblocks, one lane per sample; libwebp's import (chroma averaged in gamma-0.80 linear light)
//   k_webp_mb    W2: prediction, transforms, quantisation, reconstruction.  DC prediction needs the reconstructed
//                neighbours, so the macroblocks of an image form a chain: ONE WAVE PER IMAGE walks them in raster order and
//                its lanes are the blocks of the macroblock (0..15 luma, 16..19 U, 20..23 V); the 16 luma DCs meet by
//                v_readlane for the Walsh-Hadamard transform, which every lane repeats for itself
//   k_webp_stats / k_webp_probs   the frame's coefficient probabilities: every adaptive decision of the token walk is
//                counted first (all blocks of all macroblocks in parallel: contexts come from masks), and an entry of the
//                probability table is replaced when coding with the counted frequency pays for announcing it
//   k_webp_code  W3: the boolean entropy coder is one serial chain per partition: one wave per (image, partition) runs it on
//                its uniform side (lane 0 stores) -- the header partition and up to eight token partitions (macroblock rows
//                interleaved), independent of each other because their contexts come from masks stored with the levels;
//                k_webp_assemble puts the pieces and the RIFF / frame headers in place
// Parallelism is across the files of the batch, as in the reference's par_iter; what one wave does is a latency.
#include "../../include/vp8_tables.h"
#include "png_wave.h"
#include "webp_kernels.h"
#include "devmem.hpp"
#include "kernels.h"
#include "wave.h"

namespace csw {
using namespace csp;   // LFOR / LV / lsum / coherent_load (png_wave.h)

__device__ __forceinline__ static int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

// libwebp's import (oracle: cso_webp_rgb_to_yuv, pinned against WebPPictureImportRGB): chroma from the 2x2 block's mean in gamma-0.80 linear light
__global__ void __launch_bounds__(256) k_webp_yuv(const WebpImg *imgs, const uint8_t *rgb, uint8_t *work) {
    const WebpImg &im = imgs[blockIdx.y];
    const int w = int(im.width), h = int(im.height), ys = int(im.mbw) * 16, cs = int(im.mbw) * 8, nc = int(im.ncomp), cw = (w + 1) >> 1, ch = (h + 1) >> 1;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint8_t *src = rgb + im.rgb_off;
    auto px = [&](int yy, int xx, int &r, int &g, int &b) {
        const uint8_t *p = src + (size_t(yy < h ? yy : h - 1) * w + (xx < w ? xx : w - 1)) * nc;
        r = p[0]; g = nc >= 3 ? p[1] : p[0]; b = nc >= 3 ? p[2] : p[0];   // nc 2 / 4: the last sample is alpha, which the ALPH chunk carries (png_pipeline.cpp)
    };
    if (i < uint32_t(ys) * im.mbh * 16) {
        const int y = int(i / uint32_t(ys)), x = int(i - uint32_t(y) * ys);
        int r, g, b;
        px(y, x, r, g, b);
        work[im.y_off + i] = uint8_t((16839 * r + 33059 * g + 6420 * b + (16 << 16) + (1 << 15)) >> 16);
    }
    if (i < uint32_t(cs) * im.mbh * 8) {
        const int y = int(i / uint32_t(cs)), x = int(i - uint32_t(y) * cs);
        const int cx = x < cw ? x : cw - 1, cy = y < ch ? y : ch - 1;   // beyond the picture: the plane's last sample again
        // the two tables (578 bytes) stay in the first-level cache; twelve gathers per chroma sample next to twelve bytes from HBM
        int r = 0, g = 0, b = 0;
        for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) { int r1, g1, b1; px(2 * cy + dy, 2 * cx + dx, r1, g1, b1); r += kVp8GammaToLinear[r1]; g += kVp8GammaToLinear[g1]; b += kVp8GammaToLinear[b1]; }
        auto back = [&](int s) { const int pos = s >> 9, f = s & 511; return (int(kVp8LinearToGamma[pos + 1]) * f + int(kVp8LinearToGamma[pos]) * (512 - f) + 64) >> 7; };
        r = back(r); g = back(g); b = back(b);
        work[im.u_off + i] = uint8_t(clip8((-9719 * r - 19081 * g + 28800 * b + (128 << 18) + (1 << 17)) >> 18));
        work[im.v_off + i] = uint8_t(clip8((28800 * r - 24116 * g - 4684 * b + (128 << 18) + (1 << 17)) >> 18));
    }
}

// ---- transforms (oracle: fdct4 / fwht / iwht / idct4_add)
__device__ __forceinline__ static void fdct4(const int (&d)[16], int (&out)[16]) {   // d = src - pred, row-major
    int tmp[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = d[4 * i] + d[4 * i + 3], a1 = d[4 * i + 1] + d[4 * i + 2], a2 = d[4 * i + 1] - d[4 * i + 2], a3 = d[4 * i] - d[4 * i + 3];
        tmp[0 + i * 4] = (a0 + a1) * 8;
        tmp[1 + i * 4] = (a2 * 2217 + a3 * 5352 + 1812) >> 9;
        tmp[2 + i * 4] = (a0 - a1) * 8;
        tmp[3 + i * 4] = (a3 * 2217 - a2 * 5352 + 937) >> 9;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = tmp[0 + i] + tmp[12 + i], a1 = tmp[4 + i] + tmp[8 + i], a2 = tmp[4 + i] - tmp[8 + i], a3 = tmp[0 + i] - tmp[12 + i];
        out[0 + i] = (a0 + a1 + 7) >> 4;
        out[4 + i] = ((a2 * 2217 + a3 * 5352 + 12000) >> 16) + (a3 != 0);
        out[8 + i] = (a0 - a1 + 7) >> 4;
        out[12 + i] = (a3 * 2217 - a2 * 5352 + 51000) >> 16;
    }
}
__device__ __forceinline__ static void fwht(const int (&dc)[16], int (&out)[16]) {
    int tmp[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = dc[i * 4 + 0] + dc[i * 4 + 2], a1 = dc[i * 4 + 1] + dc[i * 4 + 3], a2 = dc[i * 4 + 1] - dc[i * 4 + 3], a3 = dc[i * 4 + 0] - dc[i * 4 + 2];
        tmp[0 + i * 4] = a0 + a1; tmp[1 + i * 4] = a3 + a2; tmp[2 + i * 4] = a3 - a2; tmp[3 + i * 4] = a0 - a1;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = tmp[0 + i] + tmp[8 + i], a1 = tmp[4 + i] + tmp[12 + i], a2 = tmp[4 + i] - tmp[12 + i], a3 = tmp[0 + i] - tmp[8 + i];
        out[0 + i] = (a0 + a1) >> 1; out[4 + i] = (a3 + a2) >> 1; out[8 + i] = (a3 - a2) >> 1; out[12 + i] = (a0 - a1) >> 1;
    }
}
__device__ __forceinline__ static void iwht(const int (&in)[16], int (&dc)[16]) {
    int tmp[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = in[0 + i] + in[12 + i], a1 = in[4 + i] + in[8 + i], a2 = in[4 + i] - in[8 + i], a3 = in[0 + i] - in[12 + i];
        tmp[0 + i] = a0 + a1; tmp[8 + i] = a0 - a1; tmp[4 + i] = a3 + a2; tmp[12 + i] = a3 - a2;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int d = tmp[0 + i * 4] + 3, a0 = d + tmp[3 + i * 4], a1 = tmp[1 + i * 4] + tmp[2 + i * 4], a2 = tmp[1 + i * 4] - tmp[2 + i * 4], a3 = d - tmp[3 + i * 4];
        dc[i * 4 + 0] = (a0 + a1) >> 3; dc[i * 4 + 1] = (a3 + a2) >> 3; dc[i * 4 + 2] = (a0 - a1) >> 3; dc[i * 4 + 3] = (a3 - a2) >> 3;
    }
}
__device__ __forceinline__ static int mul1(int a) { return ((a * 20091) >> 16) + a; }
__device__ __forceinline__ static int mul2(int a) { return (a * 35468) >> 16; }
__device__ __forceinline__ static void idct4_add(const int (&in)[16], const int (&pred)[16], int (&px)[16]) {   // px: reconstructed 4x4, row-major
    int tmp[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a = in[0 + i] + in[8 + i], b = in[0 + i] - in[8 + i];
        const int c = mul2(in[4 + i]) - mul1(in[12 + i]), d = mul1(in[4 + i]) + mul2(in[12 + i]);
        tmp[0 + i * 4] = a + d; tmp[1 + i * 4] = b + c; tmp[2 + i * 4] = b - c; tmp[3 + i * 4] = a - d;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int dc = tmp[0 + i] + 4, a = dc + tmp[8 + i], b = dc - tmp[8 + i];
        const int c = mul2(tmp[4 + i]) - mul1(tmp[12 + i]), d = mul1(tmp[4 + i]) + mul2(tmp[12 + i]);
        px[i * 4 + 0] = clip8(pred[i * 4 + 0] + ((a + d) >> 3)); px[i * 4 + 1] = clip8(pred[i * 4 + 1] + ((b + c) >> 3));
        px[i * 4 + 2] = clip8(pred[i * 4 + 2] + ((b - c) >> 3)); px[i * 4 + 3] = clip8(pred[i * 4 + 3] + ((a - d) >> 3));
    }
}
// bias / 256 of a step: libwebp's rounding offsets.  The division is a multiplication: numerators stay below 2^16 (|coefficient| < 2^15 for every
// transform here) and steps below 2^9, where (n * (2^32 / q + 1)) >> 32 is exactly n / q
__device__ __forceinline__ static uint32_t quant_recip(int q) { return uint32_t(0xFFFFFFFFu / uint32_t(q)) + 1u; }   // q >= 4, never a power-of-two edge case: 2^32 / q rounds down either way
__device__ __forceinline__ static int quant(int c, int q, int bias, uint32_t recip) {
    int a = c < 0 ? -c : c;
    a = int((uint64_t(uint32_t(a + ((q * bias) >> 8))) * recip) >> 32);
    if (a > 2047) a = 2047;
    return c < 0 ? -a : a;
}

// ---- the macroblock record: 25 blocks x 16 levels (Y2, 16 luma, 4 U, 4 V; scan order) + an info block
//   I[0], I[1]  which blocks have anything to code: bit 0 the Y2 flag as the macroblock to the RIGHT sees it, 1..16 luma, 17..24 chroma, bit 25 the
//               Y2 flag as the macroblock BELOW sees it (an i4x4 macroblock has no Y2 block and hands its neighbours' flags on: the two differ)
//   I[2] luma mode (0 DC, 1 V, 2 H, 3 TM; 4 = i4x4), I[3] chroma mode, I[4..19] the sixteen sub-block modes (i16: what the mode counts as in its
//   neighbours' sub-block contexts) -- the token coder's contexts and the header's modes, looked up without a serial pass
enum { MB_INFO = 400 };
static_assert(WEBP_MB_REC >= MB_INFO + 20, "macroblock record");
__device__ __forceinline__ static uint32_t nz_mask(const int16_t *L) { return uint32_t(uint16_t(L[MB_INFO])) | (uint32_t(uint16_t(L[MB_INFO + 1])) << 16); }

// minimum of a key over each row of 16 lanes, in every lane of the row
__device__ __forceinline__ static LV<uint32_t> lrowmin(const LV<uint32_t> &x) {
    LV<uint32_t> r;
#ifdef CSH_EMUL
    for (int g = 0; g < 4; g++) {
        uint32_t m = 0xFFFFFFFFu;
        for (int k = 0; k < 16; k++) m = x.v[g * 16 + k] < m ? x.v[g * 16 + k] : m;
        for (int k = 0; k < 16; k++) r.v[g * 16 + k] = m;
    }
#else
    uint32_t v = x.v, o;
    o = uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0xB1, 0xf, 0xf, false)); v = o < v ? o : v;    // quad_perm [1,0,3,2]
    o = uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x4E, 0xf, 0xf, false)); v = o < v ? o : v;    // quad_perm [2,3,0,1]
    o = uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x141, 0xf, 0xf, false)); v = o < v ? o : v;   // row_half_mirror
    o = uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x140, 0xf, 0xf, false)); v = o < v ? o : v;   // row_mirror
    r.v = v;
#endif
    return r;
}

// One wave per macroblock; one launch per skewed diagonal of the macroblock grid (mx + 2 my = diag): a macroblock predicts from the
// reconstruction of its left, upper, upper-left and -- the 4 x 4 modes of its right column -- upper-RIGHT neighbours, which all lie on earlier
// diagonals; the macroblocks of a diagonal -- of every picture of the batch -- are independent.  (The first version walked a picture's
// macroblocks in raster order with one wave: 16.7 us each, 83 ms for 256 pictures whatever else the chip had to do.)
// Luma (oracle: cso_webp_encode_yuv): i16x16 first; when that leaves AC levels to code the macroblock is coded i4x4 instead -- sixteen
// sub-blocks, each predicted from the reconstruction so far, ten modes tried by ten lanes (two sub-blocks at a time: sub-block (bx, by) only
// needs (bx - 1, by), (bx, by - 1) and (bx + 1, by - 1), so bx + 2 by = step walks the sixteen in ten steps).
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_webp_mb(const WebpImg *imgs, uint8_t *work, int16_t *levels, int diag) {
    CSH_SHARED uint32_t s_cbw[17 * 8];      // the macroblock's luma with its edges: 17 rows of 32 bytes; row 0 = the row above, byte 3 = the column
                                            // to the left, bytes 4..19 the macroblock, 20..23 the four samples above-right
    CSH_SHARED uint32_t s_srcw[64];         // luma source, 16 x 16
    CSH_SHARED uint16_t s_taps[128];
    CSH_SHARED uint8_t s_bm[16], s_nz[16], s_tm[4], s_lm[4], s_e[32];   // s_e: the edge line L K J I X A B C D E F G H of the (up to) two sub-blocks of a step
    uint8_t *s_cb = reinterpret_cast<uint8_t *>(s_cbw), *s_src = reinterpret_cast<uint8_t *>(s_srcw);
    const WebpImg im = imgs[blockIdx.y];
    const int mbw = int(im.mbw), mbh = int(im.mbh), ys = mbw * 16, cs = mbw * 8, qi = im.qi;
    const int y1dc = kVp8DcQ[qi], y1ac = kVp8AcQ[qi], y2dc = kVp8DcQ[qi] * 2, uvac = kVp8AcQ[qi];
    int y2ac = kVp8AcQ[qi] * 155 / 100; if (y2ac < 8) y2ac = 8;
    int uvdc = kVp8DcQ[qi]; if (uvdc > 132) uvdc = 132;
    const uint32_t r_y1dc = quant_recip(y1dc), r_y1ac = quant_recip(y1ac), r_y2dc = quant_recip(y2dc), r_y2ac = quant_recip(y2ac), r_uvdc = quant_recip(uvdc), r_uvac = quant_recip(uvac);
    const uint8_t *sy = work + im.y_off, *su = work + im.u_off, *sv = work + im.v_off;
    uint8_t *ry = work + im.ry_off, *ru = work + im.ru_off, *rv = work + im.rv_off;
    const int my = int(blockIdx.x), mx = diag - 2 * my;
    if (my >= mbh || mx < 0 || mx >= mbw) return;
    int16_t *L = levels + im.lev_off + (size_t(my) * mbw + mx) * WEBP_MB_REC, *I = L + MB_INFO;
    const int16_t *Ltop = L - size_t(mbw) * WEBP_MB_REC, *Lleft = L - WEBP_MB_REC;
    // the three DC predictions: lanes 0..31 gather the luma edge, 32..47 the U edge, 48..63 the V edge; one packed sum
    LV<uint64_t> edge;
    LFOR(l) {
        uint64_t v = 0;
        if (l < 16) { if (my) v = coherent_load(ry + size_t(my * 16 - 1) * ys + mx * 16 + l); }
        else if (l < 32) { if (mx) v = coherent_load(ry + size_t(my * 16 + (l - 16)) * ys + mx * 16 - 1); }
        else {
            const uint8_t *r = l < 48 ? ru : rv;
            const int k = (l - 32) & 15;
            if (k < 8) { if (my) v = coherent_load(r + size_t(my * 8 - 1) * cs + mx * 8 + k); }
            else if (mx) v = coherent_load(r + size_t(my * 8 + (k - 8)) * cs + mx * 8 - 1);
            v <<= l < 48 ? 16 : 32;
        }
        edge[l] = v;
        s_taps[l] = kVp8Pred4Taps[l]; s_taps[64 + l] = kVp8Pred4Taps[64 + l];
    }
    const uint64_t sums = lsum(edge);
    const int both = (mx && my) ? 1 : 0, any = (mx || my) ? 1 : 0;
    const int sY = int(sums & 0xFFFFu), sU = int((sums >> 16) & 0xFFFFu), sV = int((sums >> 32) & 0xFFFFu);
    const int dcY = !any ? 128 : both ? (sY + 16) >> 5 : (sY + 8) >> 4;
    const int dcU = !any ? 128 : both ? (sU + 8) >> 4 : (sU + 4) >> 3;
    const int dcV = !any ? 128 : both ? (sV + 8) >> 4 : (sV + 4) >> 3;
    // every block lane: its 4x4 source samples and, when both neighbours exist, the macroblock edge it predicts from
    LV<int> dc0;
    int coef[16];   // this lane's block (emulation: kept per lane in coefs[])
    int pred[16];   // its prediction (emulation: preds[])
#ifdef CSH_EMUL
    int coefs[24][16], preds[24][16], srcs[24][16], tops[24][4], lefts[24][4], corners[24];
#endif
    int src16[16], top4[4], left4[4], corner = 0;
    const int nmodes = (mx && my) ? 4 : 1;
    LFOR(l) if (l < 24) {
        const bool luma = l < 16;
        const int b = luma ? l : (l - 16) & 3, bx = luma ? b & 3 : b & 1, by = luma ? b >> 2 : b >> 1;
        const int stride = luma ? ys : cs, n0 = luma ? 16 : 8;
        const uint8_t *s = (luma ? sy : (l < 20 ? su : sv)) + size_t(my * n0 + by * 4) * stride + mx * n0 + bx * 4;
        const uint8_t *r = (luma ? ry : (l < 20 ? ru : rv)) + size_t(my * n0) * stride + mx * n0;   // the macroblock's corner in the reconstruction
        CSH_UNROLL
        for (int rr = 0; rr < 4; rr++) {
            const uint32_t w4 = *reinterpret_cast<const uint32_t *>(s + size_t(rr) * stride);
            if (luma) s_srcw[(by * 4 + rr) * 4 + bx] = w4;
            CSH_UNROLL
            for (int c = 0; c < 4; c++) src16[rr * 4 + c] = int((w4 >> (8 * c)) & 255u);
        }
        CSH_UNROLL
        for (int k = 0; k < 4; k++) { top4[k] = 0; left4[k] = 0; }
        corner = 0;
        if (nmodes == 4) {
            const uint32_t t4 = coherent_load(reinterpret_cast<const uint32_t *>(r - stride + bx * 4));
            CSH_UNROLL
            for (int k = 0; k < 4; k++) { top4[k] = int((t4 >> (8 * k)) & 255u); left4[k] = coherent_load(r + size_t(by * 4 + k) * stride - 1); }
            corner = coherent_load(r - stride - 1);
        }
#ifdef CSH_EMUL
        for (int k = 0; k < 16; k++) srcs[l][k] = src16[k];
        for (int k = 0; k < 4; k++) { tops[l][k] = top4[k]; lefts[l][k] = left4[k]; }
        corners[l] = corner;
#endif
    }
    // the mode of the luma block and the shared mode of the two chroma blocks: least sum of |DCT coefficients| of the residual
    int ymode = 0, cmode = 0;
    if (nmodes == 4) {
        uint64_t best_y = ~0ull, best_c = ~0ull;
        for (int m = 0; m < 4; m++) {
            LV<uint64_t> cost;
            LFOR(l) {
                cost[l] = 0;
                if (l < 24) {
#ifdef CSH_EMUL
                    for (int k = 0; k < 16; k++) src16[k] = srcs[l][k];
                    for (int k = 0; k < 4; k++) { top4[k] = tops[l][k]; left4[k] = lefts[l][k]; }
                    corner = corners[l];
#endif
                    const int flat = l < 16 ? dcY : (l < 20 ? dcU : dcV);
                    int d[16], c[16];
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) {
                        const int x = k & 3, y = k >> 2;
                        const int pv = m == 0 ? flat : m == 1 ? top4[x] : m == 2 ? left4[y] : clip8(top4[x] + left4[y] - corner);
                        d[k] = src16[k] - pv;
                    }
                    fdct4(d, c);
                    uint64_t sum = 0;
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) sum += uint64_t(c[k] < 0 ? -c[k] : c[k]);
                    cost[l] = l < 16 ? sum : sum << 32;
                }
            }
            const uint64_t tot = lsum(cost), cy = tot & 0xFFFFFFFFull, cc = tot >> 32;
            if (cy < best_y) { best_y = cy; ymode = m; }
            if (cc < best_c) { best_c = cc; cmode = m; }
        }
    }
    // residual against the chosen prediction, forward DCT; does any luma AC coefficient survive the quantiser?
    LV<int> acl;
    LFOR(l) {
        dc0[l] = 0; acl[l] = 0;
        if (l < 24) {
#ifdef CSH_EMUL
            for (int k = 0; k < 16; k++) src16[k] = srcs[l][k];
            for (int k = 0; k < 4; k++) { top4[k] = tops[l][k]; left4[k] = lefts[l][k]; }
            corner = corners[l];
#endif
            const int m = l < 16 ? ymode : cmode, flat = l < 16 ? dcY : (l < 20 ? dcU : dcV);
            int d[16];
            CSH_UNROLL
            for (int k = 0; k < 16; k++) {
                const int x = k & 3, y = k >> 2;
                pred[k] = m == 0 ? flat : m == 1 ? top4[x] : m == 2 ? left4[y] : clip8(top4[x] + left4[y] - corner);
                d[k] = src16[k] - pred[k];
            }
            fdct4(d, coef);
            dc0[l] = coef[0];
            if (l < 16) {
                int big = 0;
                CSH_UNROLL
                for (int k = 1; k < 16; k++) big |= ((coef[k] < 0 ? -coef[k] : coef[k]) + ((y1ac * 110) >> 8) >= y1ac) ? 1 : 0;   // <=> its level is not 0
                acl[l] = big;
            }
#ifdef CSH_EMUL
            for (int k = 0; k < 16; k++) { coefs[l][k] = coef[k]; preds[l][k] = pred[k]; }
#endif
        }
    }
    const bool use4 = lballot([&](int l) { return acl[l] != 0; }) != 0;
    // the 16 luma DCs to everyone; Walsh-Hadamard, quantise, and back: each luma lane takes its own DC out of the result
    int dcs[16], y2[16], dq[16], lv2[16];
    CSH_UNROLL
    for (int k = 0; k < 16; k++) {
#ifdef CSH_EMUL
        dcs[k] = dc0.v[k];
#else
        dcs[k] = __builtin_amdgcn_readlane(dc0.v, k);
#endif
    }
    int y2any = 0;
    if (!use4) {
        fwht(dcs, y2);
        CSH_UNROLL
        for (int n = 0; n < 16; n++) { const int k = kVp8Zigzag[n], q = k ? y2ac : y2dc; lv2[n] = quant(y2[k], q, k ? 108 : 96, k ? r_y2ac : r_y2dc); dq[k] = lv2[n] * q; y2any |= lv2[n]; }
        iwht(dq, dcs);
    } else {
        CSH_UNROLL
        for (int n = 0; n < 16; n++) lv2[n] = 0;
    }
    LV<int> nzl;
    LFOR(l) {
        nzl[l] = 0;
        if (l == 0) { CSH_UNROLL for (int n = 0; n < 16; n++) L[n] = int16_t(lv2[n]); }
        if (l < 24 && !(use4 && l < 16)) {
#ifdef CSH_EMUL
            for (int k = 0; k < 16; k++) { coef[k] = coefs[l][k]; pred[k] = preds[l][k]; }
#endif
            const bool luma = l < 16;
            const int b = luma ? l : (l - 16) & 3;
            int c[16], px[16], lv[16];
            if (luma) {   // only when every AC level is 0: the block is its share of the Y2 block
                int mine = 0;
                CSH_UNROLL
                for (int k = 0; k < 16; k++) mine = b == k ? dcs[k] : mine;
                CSH_UNROLL
                for (int n = 0; n < 16; n++) { lv[n] = 0; c[n] = 0; }
                c[0] = mine;
            } else {
                CSH_UNROLL
                for (int n = 0; n < 16; n++) { const int k = kVp8Zigzag[n], q = k ? uvac : uvdc; lv[n] = quant(coef[k], q, k ? 115 : 110, k ? r_uvac : r_uvdc); c[k] = lv[n] * q; }
            }
            idct4_add(c, pred, px);
            uint8_t *r = luma ? ry + size_t(my * 16 + (b >> 2) * 4) * ys + mx * 16 + (b & 3) * 4
                              : (l < 20 ? ru : rv) + size_t(my * 8 + (b >> 1) * 4) * cs + mx * 8 + (b & 1) * 4;
            const int stride = luma ? ys : cs;
            CSH_UNROLL
            for (int rr = 0; rr < 4; rr++)
                *reinterpret_cast<uint32_t *>(r + size_t(rr) * stride) = uint32_t(px[rr * 4]) | (uint32_t(px[rr * 4 + 1]) << 8) | (uint32_t(px[rr * 4 + 2]) << 16) | (uint32_t(px[rr * 4 + 3]) << 24);
            int16_t *o = L + (luma ? 1 + b : 17 + (l - 16)) * 16;
            CSH_UNROLL
            for (int n = 0; n < 16; n++) o[n] = int16_t(lv[n]);
            int any = 0;
            CSH_UNROLL
            for (int n = 0; n < 16; n++) any |= lv[n];
            nzl[l] = any != 0;
        }
    }
    uint32_t luma_nz = 0;
    if (use4) {
        // ---- i4x4.  The luma context: the row above (127 above the frame; its corner 129 on the left frame edge below the first row; the four
        // samples above-right come from the next macroblock of the row above, or repeat the last one at the right frame edge) and the column
        // to the left (129 outside) -- the decoder's rules
        LFOR(l) {
            if (l < 24) {
                const int x = l - 4;   // byte l of row 0 is sample x of the row above
                int v = 127;
                if (x >= -1 && my > 0) {
                    const uint8_t *top = ry + size_t(my * 16 - 1) * ys + mx * 16;
                    if (x < 0) v = mx > 0 ? int(coherent_load(top - 1)) : 129;
                    else if (x < 16) v = coherent_load(top + x);
                    else v = mx + 1 < mbw ? int(coherent_load(top + x)) : int(coherent_load(top + 15));
                }
                s_cb[l] = uint8_t(v);
            } else if (l >= 32 && l < 48) {
                const int y = l - 32;
                s_cb[(y + 1) * 32 + 3] = mx > 0 ? coherent_load(ry + size_t(my * 16 + y) * ys + mx * 16 - 1) : uint8_t(129);
            } else if (l >= 48 && l < 52) {
                s_tm[l - 48] = my > 0 ? uint8_t(Ltop[MB_INFO + 4 + 12 + (l - 48)]) : uint8_t(0);
            } else if (l >= 52 && l < 56) {
                s_lm[l - 52] = mx > 0 ? uint8_t(Lleft[MB_INFO + 4 + (l - 52) * 4 + 3]) : uint8_t(0);
            }
        }
        CSP_WAVE_SYNC();
        for (int t = 0; t < 10; t++) {
            const int by0 = t <= 3 ? 0 : (t - 2) >> 1;
            LV<uint32_t> key;
            int px[16], lv[16];
#ifdef CSH_EMUL
            int pxs[64][16], lvs[64][16];
#endif
            LFOR(l) {
                const int grp = l >> 4, i = l & 15, by = by0 + grp, bx = t - 2 * by;
                if (grp < 2 && i < 13 && by <= 3 && bx >= 0 && bx <= 3) {
                    const uint8_t *d = s_cb + (by * 4 + 1) * 32 + 4 + bx * 4;   // the sub-block's first sample
                    s_e[grp * 16 + i] = i < 4 ? d[(3 - i) * 32 - 1] : i == 4 ? d[-32 - 1] : i < 9 ? d[-32 + (i - 5)] : bx == 3 ? s_cb[20 + (i - 9)] : d[-32 + 4 + (i - 9)];
                }
            }
            CSP_WAVE_SYNC();
            LFOR(l) {
                key[l] = 0xFFFFFFFFu;
                const int grp = l >> 4, m = l & 15, by = by0 + grp, bx = t - 2 * by;
                if (grp < 2 && m < 10 && by <= 3 && bx >= 0 && bx <= 3) {
                    const int k = by * 4 + bx;
                    const uint8_t *e = s_e + grp * 16;
                    const int tmode = by ? s_bm[k - 4] : s_tm[bx], lmode = bx ? s_bm[k - 1] : s_lm[by];
                    int p4[16], dd[16], c[16];
                    if (m == 0) {
                        const int v = (e[5] + e[6] + e[7] + e[8] + e[3] + e[2] + e[1] + e[0] + 4) >> 3;
                        CSH_UNROLL
                        for (int i = 0; i < 16; i++) p4[i] = v;
                    } else if (m == 1) {
                        int ee[9];
                        CSH_UNROLL
                        for (int i = 0; i < 9; i++) ee[i] = e[i];
                        CSH_UNROLL
                        for (int i = 0; i < 16; i++) p4[i] = clip8(ee[3 - (i >> 2)] + ee[5 + (i & 3)] - ee[4]);
                    } else {
                        // the directional modes: four taps per sample out of the edge line (vp8_tables.h)
                        CSH_UNROLL
                        for (int i = 0; i < 16; i++) {
                            const uint32_t tp = s_taps[(m - 2) * 16 + i];
                            p4[i] = (int(e[tp & 15u]) + int(e[(tp >> 4) & 15u]) + int(e[(tp >> 8) & 15u]) + int(e[tp >> 12]) + 2) >> 2;
                        }
                    }
                    CSH_UNROLL
                    for (int i = 0; i < 16; i++) dd[i] = int(s_src[(by * 4 + (i >> 2)) * 16 + bx * 4 + (i & 3)]) - p4[i];
                    fdct4(dd, c);
                    uint32_t satd = 0;
                    CSH_UNROLL
                    for (int i = 0; i < 16; i++) satd += uint32_t(c[i] < 0 ? -c[i] : c[i]);
                    const uint32_t sc = satd * 16u + ((4u * uint32_t(y1ac) * uint32_t(kVp8BModeCost[(tmode * 10 + lmode) * 10 + m])) >> 8);
                    key[l] = (sc << 4) | uint32_t(m);
                    // every candidate goes on to its levels and reconstruction: the lanes run together anyway, and the winner has them at hand
                    int cq[16];
                    CSH_UNROLL
                    for (int n = 0; n < 16; n++) { const int z = kVp8Zigzag[n], q = z ? y1ac : y1dc; lv[n] = quant(c[z], q, z ? 110 : 96, z ? r_y1ac : r_y1dc); cq[z] = lv[n] * q; }
                    idct4_add(cq, p4, px);
#ifdef CSH_EMUL
                    for (int i = 0; i < 16; i++) { pxs[l][i] = px[i]; lvs[l][i] = lv[i]; }
#endif
                }
            }
            const LV<uint32_t> best = lrowmin(key);
            LFOR(l) {
                const int grp = l >> 4, m = l & 15, by = by0 + grp, bx = t - 2 * by;
                if (key[l] != 0xFFFFFFFFu && key[l] == best[l]) {
#ifdef CSH_EMUL
                    for (int i = 0; i < 16; i++) { px[i] = pxs[l][i]; lv[i] = lvs[l][i]; }
#endif
                    const int k = by * 4 + bx;
                    CSH_UNROLL
                    for (int rr = 0; rr < 4; rr++)
                        s_cbw[(by * 4 + rr + 1) * 8 + 1 + bx] = uint32_t(px[rr * 4]) | (uint32_t(px[rr * 4 + 1]) << 8) | (uint32_t(px[rr * 4 + 2]) << 16) | (uint32_t(px[rr * 4 + 3]) << 24);
                    int16_t *o = L + (1 + k) * 16;
                    int any = 0;
                    CSH_UNROLL
                    for (int n = 0; n < 16; n++) { o[n] = int16_t(lv[n]); any |= lv[n]; }
                    s_bm[k] = uint8_t(m);
                    s_nz[k] = any ? 1 : 0;
                }
            }
            CSP_WAVE_SYNC();
        }
        // the macroblock's reconstruction to the plane, a word per lane
        LFOR(l) *reinterpret_cast<uint32_t *>(ry + size_t(my * 16 + (l >> 2)) * ys + mx * 16 + (l & 3) * 4) = s_cbw[((l >> 2) + 1) * 8 + 1 + (l & 3)];
        luma_nz = uint32_t(lballot([&](int l) { return l < 16 && s_nz[l] != 0; }));
    } else
        luma_nz = uint32_t(lballot([&](int l) { return l < 16 && nzl[l] != 0; }));
    // the info block
    {
        const uint32_t chroma_nz = uint32_t(lballot([&](int l) { return l >= 16 && l < 24 && nzl[l] != 0; }) >> 16);
        const uint32_t left_y2 = mx > 0 ? nz_mask(Lleft) & 1u : 0u, top_y2 = my > 0 ? (nz_mask(Ltop) >> 25) & 1u : 0u;
        const uint32_t mask = (use4 ? left_y2 : (y2any ? 1u : 0u)) | ((luma_nz & 0xFFFFu) << 1) | ((chroma_nz & 0xFFu) << 17) | ((use4 ? top_y2 : (y2any ? 1u : 0u)) << 25);
        const int as_b = ymode == 0 ? 0 : ymode == 1 ? 2 : ymode == 2 ? 3 : 1;
        LFOR(l) {
            if (l == 0) { I[0] = int16_t(mask & 0xFFFFu); I[1] = int16_t(mask >> 16); I[2] = int16_t(use4 ? 4 : ymode); I[3] = int16_t(cmode); }
            if (l < 16) I[4 + l] = int16_t(use4 ? int(s_bm[l]) : as_b);
        }
    }
}

// ---- W3: boolean entropy coder (oracle: boolenc) and the token walk (oracle: put_coeffs)
#ifdef CSH_EMUL
#define LANE0 if (true)
#else
#define LANE0 if ((threadIdx.x & 63u) == 0)
#endif
// The coder's state is wave-uniform.  VEC = false keeps it in scalar registers; VEC = true keeps range and value in VECTOR registers (every lane the same
// numbers): with eight token waves per picture and a thousand pictures the token kernel runs at three quarters of the chip's SCALAR issue rate (12 k scalar
// instructions per macroblock against 1.7 k vector ones, profiles/r02_pmc_sq_webp_batch256.txt), so the arithmetic of a decision -- split, the two updates, the
// renormalisation, which needs no branch: the shift is 0 when none is due -- moves to the idle vector unit and only the bit count and the byte output stay scalar.
template <bool VEC>
struct BoolEncT {
    uint8_t *buf;
    uint32_t pos, cap;
    int32_t range, value;
    int run, nb_bits;
    bool overflow;
    __device__ __forceinline__ static int32_t u(int32_t v) { return int32_t(csp::uni(uint32_t(v))); }
    __device__ __forceinline__ static int32_t vzero() {   // a zero the compiler takes for a per-lane value
#ifdef CSH_EMUL
        return 0;
#else
        int32_t z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
#endif
    }
    __device__ __forceinline__ void init(uint8_t *b, uint32_t c) {
        buf = b; pos = 0; cap = c; run = 0; nb_bits = -8; overflow = false;
        if (VEC) { range = 254 + vzero(); value = vzero(); } else { range = 254; value = 0; }
    }
    __device__ __forceinline__ void flush_bits() {
        const int s = 8 + nb_bits;
        const int32_t bits = u(value >> s);
        value = VEC ? value - (bits << s) : u(value - (bits << s));
        nb_bits = u(nb_bits - 8);
        if ((bits & 0xff) != 0xff) {
            if (pos + uint32_t(run) + 1 > cap) { overflow = true; run = 0; return; }
            LANE0 {
                if ((bits & 0x100) && pos > 0) buf[pos - 1]++;
                const uint8_t v = (bits & 0x100) ? 0x00 : 0xff;
                for (int k = 0; k < run; k++) buf[pos + uint32_t(k)] = v;
                buf[pos + uint32_t(run)] = uint8_t(bits & 0xff);
            }
            pos = uint32_t(u(int32_t(pos + uint32_t(run) + 1))); run = 0;
        } else
            run = u(run + 1);
    }
    __device__ __forceinline__ void put(int bit, int prob) {
        bit = u(bit); prob = u(prob);
        if (VEC) {
            const int32_t split = (range * prob) >> 8, m = -int32_t(bit != 0);
            value += (split + 1) & m;
            range = split + ((range - 2 * split - 1) & m);               // bit ? range - split - 1 : split
            const int shift = __clz(uint32_t(range + 1)) - 24;          // 0 for range >= 127: no renormalisation due
            range = ((range + 1) << shift) - 1;
            value <<= shift;
            nb_bits = u(nb_bits + u(shift));
            if (nb_bits > 0) flush_bits();
            return;
        }
        const int32_t split = (range * prob) >> 8;
        if (bit) { value += split + 1; range -= split + 1; } else range = split;
        range = u(range); value = u(value);
        if (range < 127) {
            const int shift = __clz(uint32_t(range + 1)) - 24;
            range = u(((range + 1) << shift) - 1);
            value = u(value << shift);
            nb_bits = u(nb_bits + shift);
            if (nb_bits > 0) flush_bits();
        }
    }
    __device__ __forceinline__ void bits(uint32_t v, int n) { while (n--) put(int((v >> n) & 1u), 128); }
    __device__ __forceinline__ void finish() { bits(0, 9 - nb_bits); nb_bits = 0; flush_bits(); }
};
typedef BoolEncT<false> BoolEnc;      // the header partition
typedef BoolEncT<true> BoolEncTok;    // the token partitions
// the token walk of one block, either coding (CodeSink: the frame's probabilities) or only counting what it would code
// (StatSink), which is how the frame's probabilities are chosen (oracle: put_coeffs / tsink)
// the coder's wave is one serial chain, and what makes it slow is waiting for memory once per decision: the frame's probabilities
// therefore sit in LDS, and a block's sixteen levels arrive with ONE load (lane n holds level n; the walk reads them with v_readlane)
struct CodeSink {
    BoolEncTok &e;
    const uint8_t *probs;   // LDS
    LV<int> lvl;
    __device__ __forceinline__ void begin(const int16_t *lv) { LFOR(l) lvl[l] = l < 16 ? int(lv[l]) : 0; }
    __device__ __forceinline__ int lev(const int16_t *, int i) const {
#ifdef CSH_EMUL
        return lvl.v[i];
#else
        return __builtin_amdgcn_readlane(lvl.v, i);
#endif
    }
    __device__ __forceinline__ int last_nonzero(const int16_t *, int first) const {
        const uint64_t nz = lballot([&](int l) { return l >= first && l < 16 && lvl[l] != 0; });
        return nz ? 63 - __builtin_clzll(nz) : -1;
    }
    __device__ __forceinline__ void ad(int bit, int idx) { e.put(bit, probs[idx]); }
    __device__ __forceinline__ void fx(int bit, int prob) { e.put(bit, prob); }
};
struct StatSink {
    uint32_t *cnt;   // [1056][2] in LDS
    uint32_t nd;     // decisions of this block (adaptive and fixed-probability ones): what its stretch of the decision stream will hold
    __device__ __forceinline__ void begin(const int16_t *) {}
    __device__ __forceinline__ int lev(const int16_t *lv, int i) const { return lv[i]; }   // lanes = blocks here: every lane reads its own block
    __device__ __forceinline__ int last_nonzero(const int16_t *lv, int first) const { int last = -1; for (int i = first; i < 16; i++) if (lv[i]) last = i; return last; }
    __device__ __forceinline__ void ad(int bit, int idx) { atomicAdd(&cnt[2 * idx + (bit ? 1 : 0)], 1u); nd++; }
    __device__ __forceinline__ void fx(int, int) { nd++; }
};
// the same walk writing its decisions down: (bit, probability) pairs, two bytes each, in the order the coder takes them -- the probability is resolved here
// (the frame's table is final by now), so the coder behind it needs neither the levels nor the tables (k_webp_bool)
struct WriteSink {
    const uint8_t *probs;   // LDS
    uint16_t *out;
    __device__ __forceinline__ void begin(const int16_t *) {}
    __device__ __forceinline__ int lev(const int16_t *lv, int i) const { return lv[i]; }
    __device__ __forceinline__ int last_nonzero(const int16_t *lv, int first) const { int last = -1; for (int i = first; i < 16; i++) if (lv[i]) last = i; return last; }
    __device__ __forceinline__ void ad(int bit, int idx) { *out++ = uint16_t((bit ? 1u : 0u) | (uint32_t(probs[idx]) << 1)); }
    __device__ __forceinline__ void fx(int bit, int prob) { *out++ = uint16_t((bit ? 1u : 0u) | (uint32_t(prob) << 1)); }
};
template <class S>
__device__ static int put_coeffs(S &e, int type, int ctx, const int16_t *lv, int first) {
    e.begin(lv);
    const int last = e.last_nonzero(lv, first);
    int n = first;
    int p = ((type * 8 + kVp8Bands[n]) * 3 + ctx) * 11;
    if (last < 0) { e.ad(0, p + 0); return 0; }
    e.ad(1, p + 0);
    while (n < 16) {
        const int c = e.lev(lv, n++);
        const int sign = c < 0;
        int v = sign ? -c : c;
        if (!v) { e.ad(0, p + 1); p = ((type * 8 + kVp8Bands[n]) * 3 + 0) * 11; continue; }
        e.ad(1, p + 1);
        if (v == 1) { e.ad(0, p + 2); p = ((type * 8 + kVp8Bands[n]) * 3 + 1) * 11; }
        else {
            e.ad(1, p + 2);
            if (v <= 4) { e.ad(0, p + 3); if (v == 2) e.ad(0, p + 4); else { e.ad(1, p + 4); e.ad(v == 4, p + 5); } }
            else if (v <= 10) {
                e.ad(1, p + 3); e.ad(0, p + 6);
                if (v <= 6) { e.ad(0, p + 7); e.fx(v == 6, 159); }
                else { e.ad(1, p + 7); e.fx(v >= 9, 165); e.fx(!(v & 1), 145); }
            } else {
                int mask; const uint8_t *tab;
                e.ad(1, p + 3); e.ad(1, p + 6);
                if (v < 3 + (8 << 1)) { e.ad(0, p + 8); e.ad(0, p + 9); v -= 3 + (8 << 0); mask = 1 << 2; tab = kVp8Cat3; }
                else if (v < 3 + (8 << 2)) { e.ad(0, p + 8); e.ad(1, p + 9); v -= 3 + (8 << 1); mask = 1 << 3; tab = kVp8Cat4; }
                else if (v < 3 + (8 << 3)) { e.ad(1, p + 8); e.ad(0, p + 10); v -= 3 + (8 << 2); mask = 1 << 4; tab = kVp8Cat5; }
                else { e.ad(1, p + 8); e.ad(1, p + 10); v -= 3 + (8 << 3); mask = 1 << 10; tab = kVp8Cat6; }
                while (mask) { e.fx(!!(v & mask), *tab++); mask >>= 1; }
            }
            p = ((type * 8 + kVp8Bands[n]) * 3 + 2) * 11;
        }
        e.fx(sign, 128);
        if (n == 16) return 1;
        if (n > last) { e.ad(0, p + 0); return 1; }
        e.ad(1, p + 0);
    }
    return 1;
}

// One wave per (image, partition): y = 0 the header partition (frame header fields and the macroblock modes), y = 1..8 the
// token partitions (macroblock row r belongs to partition r mod P).  Contexts come from the non-zero masks k_webp_mb left
// with the levels, so no partition waits for another.  Every partition goes to its own slice of a scratch region; the
// sizes decide where k_webp_assemble puts them.
__device__ __forceinline__ static int webp_parts(int mbh) { return mbh >= 8 ? 8 : mbh >= 4 ? 4 : mbh >= 2 ? 2 : 1; }
__device__ __forceinline__ static uint32_t webp_hdr_cap(const WebpImg &im) { return 2048u + ((im.out_cap - 4096u) >> 4); }   // a sixteenth of the file's room (48 bytes per macroblock to begin with): the frame header and the modes, sixteen of them in an i4x4 macroblock; grows with out_cap when a run is repeated
__device__ __forceinline__ static uint32_t webp_part_cap(const WebpImg &im) { return (im.out_cap - 128u - webp_hdr_cap(im)) / uint32_t(webp_parts(int(im.mbh))); }
// block k of a macroblock (0 the Y2 block, 1..16 luma, 17..24 chroma): coefficient type, first coded position, and the context
// "how many of the blocks above / to the left have something to code" out of the three masks
__device__ __forceinline__ static void block_info(int k, uint32_t cur, uint32_t top, uint32_t left, bool i4, int &type, int &first, int &ctx) {
    if (k == 0) { type = 1; first = 0; ctx = int(((top >> 25) & 1u) + (left & 1u)); return; }
    if (k <= 16) {
        const int b = k - 1, bx = b & 3, by = b >> 2;
        const uint32_t t1 = by ? (cur >> (1 + (by - 1) * 4 + bx)) & 1u : (top >> (13 + bx)) & 1u;
        const uint32_t l1 = bx ? (cur >> (by * 4 + bx)) & 1u : (left >> (4 + by * 4)) & 1u;
        type = i4 ? 3 : 0; first = i4 ? 0 : 1; ctx = int(t1 + l1);   // an i4x4 macroblock's luma blocks carry their own DC
        return;
    }
    const int b = k - 17, pl = b >> 2, bx = b & 1, by = (b >> 1) & 1, b0 = 17 + pl * 4;
    const uint32_t t1 = by ? (cur >> (b0 + bx)) & 1u : (top >> (b0 + 2 + bx)) & 1u;
    const uint32_t l1 = bx ? (cur >> (b0 + by * 2)) & 1u : (left >> (b0 + by * 2 + 1)) & 1u;
    type = 2; first = 0; ctx = int(t1 + l1);
}
// counts of every adaptive decision of the token walk: one wave per macroblock row, lanes 0..24 = the blocks of one macroblock
// at a time (the contexts come from the masks, so all blocks of all macroblocks are independent); LDS counters, then one
// atomic per non-zero counter into the image's totals
enum { WEBP_NPROB = 4 * 8 * 3 * 11 };
// where macroblock (mx, my) stands among the macroblocks of its picture in CHAIN order: the rows of token partition 0 first (rows 0, P, 2P ..), then
// partition 1's -- the order in which the decision stream holds them
__device__ __forceinline__ static uint32_t webp_chain_index(int mbw, int mbh, int mx, int my) {
    const int P = webp_parts(mbh), p = my % P;
    int rows_before = 0;
    for (int q = 0; q < p; q++) rows_before += (mbh - q + P - 1) / P;
    return uint32_t((rows_before + my / P) * mbw + mx);
}
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_webp_stats(const WebpImg *imgs, const int16_t *levels, uint32_t *stats, const uint64_t *mb_base, uint32_t *mb_cnt, uint16_t *blk_cnt) {
    CSH_SHARED uint32_t cnt[2 * WEBP_NPROB];
    const WebpImg im = imgs[blockIdx.y];
    const int mbw = int(im.mbw), my = int(blockIdx.x);
    if (my >= int(im.mbh)) return;
    LFOR(l) for (int i = l; i < 2 * WEBP_NPROB; i += 64) cnt[i] = 0;
    CSP_WAVE_SYNC();
    const int16_t *row = levels + im.lev_off + size_t(my) * mbw * WEBP_MB_REC;
    for (int mx0 = 0; mx0 < mbw; mx0 += 2) {   // two macroblocks to a step: lanes 0..24 the blocks of one, lanes 32..56 those of the next
        LV<uint32_t> nd;
        LFOR(l) {
            const int mx = mx0 + (l >> 5), k = l & 31;
            nd[l] = 0;
            if (mx < mbw && k < 25) {
                const int16_t *L = row + size_t(mx) * WEBP_MB_REC;
                const bool i4 = L[MB_INFO + 2] == 4;
                if (!(i4 && k == 0)) {   // no Y2 block in an i4x4 macroblock
                    const uint32_t cur = nz_mask(L), top = my ? nz_mask(L - size_t(mbw) * WEBP_MB_REC) : 0u, left = mx ? nz_mask(L - WEBP_MB_REC) : 0u;
                    int type, first, ctx;
                    block_info(k, cur, top, left, i4, type, first, ctx);
                    StatSink sink{cnt, 0u};
                    put_coeffs(sink, type, ctx, L + k * 16, first);
                    nd[l] = sink.nd;
                }
            }
        }
        // decisions per block and per macroblock, in chain order: the write pass (k_webp_decisions) places every block's stretch from them
        if (!mb_cnt) continue;   // (no room for the decision streams: the partitions will be coded as chains)
        uint32_t total;
        const LV<uint32_t> ex = lscan(nd, total);
        const uint32_t lower = csh::lget(ex, 32);
        LFOR(l) {
            const int mx = mx0 + (l >> 5), k = l & 31;
            if (mx < mbw) {
                const uint64_t at = mb_base[blockIdx.y] + webp_chain_index(mbw, int(im.mbh), mx, my);
                blk_cnt[at * 32 + uint32_t(k)] = uint16_t(nd[l]);
                if (k == 0) mb_cnt[at] = (l >> 5) ? total - lower : lower;
            }
        }
    }
    CSP_WAVE_SYNC();
    uint32_t *dst = stats + size_t(im.image) * 2 * WEBP_NPROB;
    LFOR(l) for (int i = l; i < 2 * WEBP_NPROB; i += 64) if (cnt[i]) atomicAdd(&dst[i], cnt[i]);
}
// the frame's coefficient probabilities (oracle: bool_cost / choose_probs): one lane per entry
__device__ __forceinline__ static uint32_t bool_cost(int p) {
    const int l = 31 - __clz(uint32_t(p));
    return uint32_t(256 * (8 - l)) - (((uint32_t(p) << 8) >> l) - 256u);
}
__global__ void __launch_bounds__(256) k_webp_probs(const WebpImg *imgs, const uint32_t *stats, uint8_t *probs, uint8_t *update) {
    const WebpImg &im = imgs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= WEBP_NPROB) return;
    const uint64_t n0 = stats[(size_t(im.image) * WEBP_NPROB + i) * 2], n1 = stats[(size_t(im.image) * WEBP_NPROB + i) * 2 + 1], total = n0 + n1;
    const int oldp = kVp8CoefProbs[i], up = kVp8CoefUpdateProbs[i];
    int np = total ? int(255 - n1 * 255 / total) : 255;
    if (np < 1) np = 1;
    const uint64_t old_cost = n0 * bool_cost(oldp) + n1 * bool_cost(256 - oldp) + bool_cost(up);
    const uint64_t new_cost = n0 * bool_cost(np) + n1 * bool_cost(256 - np) + bool_cost(256 - up) + 8 * 256;
    const bool use = new_cost < old_cost;
    update[size_t(im.image) * WEBP_NPROB + i] = use ? 1 : 0;
    probs[size_t(im.image) * WEBP_NPROB + i] = uint8_t(use ? np : oldp);
}

// the header partition (frame header, the frame's coefficient probabilities, every macroblock's modes) is a function of its own, not inlined: next to it
// in one body the token walk's coder state no longer fitted the scalar registers, and the eight token waves of a picture ran 3.4 x slower
__device__ __attribute__((noinline)) static void code_header(const WebpImg &im, const int16_t *lev, const uint8_t *probs, const uint8_t *update, const uint8_t *s_bmp, uint8_t *base, uint32_t *size_out) {
    const int mbw = int(im.mbw), mbh = int(im.mbh), nparts = webp_parts(mbh);
    BoolEnc e;
        e.init(base, webp_hdr_cap(im));
        e.bits(0, 1); e.bits(0, 1); e.bits(0, 1);           // colour space, clamping, no segmentation
        e.bits(1, 1); e.bits(0, 6); e.bits(0, 3);           // simple filter at level 0 (off), sharpness
        e.bits(0, 1);                                       // no filter deltas
        e.bits(uint32_t(nparts == 8 ? 3 : nparts == 4 ? 2 : nparts == 2 ? 1 : 0), 2);
        e.bits(uint32_t(im.qi), 7);
        for (int i = 0; i < 5; i++) e.bits(0, 1);           // no quantiser deltas
        e.bits(0, 1);                                       // refresh_entropy_probs
        for (int i = 0; i < WEBP_NPROB; i++) { e.put(update[i], kVp8CoefUpdateProbs[i]); if (update[i]) e.bits(probs[i], 8); }   // the frame's coefficient probabilities
        e.bits(0, 1);                                       // no skip flags
        // the modes: this wave is one serial chain over every macroblock of the picture, so nothing in it may wait for HBM per decision.  A macroblock's
        // info block and the neighbours' modes it needs arrive with one load, a macroblock ahead (lanes 0..19 its info, 32..35 the sub-block modes above,
        // 40..43 those to the left); the sub-block mode probabilities sit in LDS and come a row (nine) at a time
        auto info_of = [&](int i) {
            LV<int> v;
            const int my = i / mbw, mx = i - my * mbw;
            const int16_t *I = lev + size_t(i) * WEBP_MB_REC + MB_INFO, *It = I - size_t(mbw) * WEBP_MB_REC, *Il = I - WEBP_MB_REC;
            LFOR(l) {
                int x = 0;
                if (i < mbw * mbh) {
                    if (l < 20) x = I[l];
                    else if (l >= 32 && l < 36) x = my ? int(It[4 + 12 + (l - 32)]) : 0;
                    else if (l >= 40 && l < 44) x = mx ? int(Il[4 + (l - 40) * 4 + 3]) : 0;
                }
                v[l] = x;
            }
            return v;
        };
        auto rd = [](const LV<int> &v, int i) {
#ifdef CSH_EMUL
            return v.v[i];
#else
            return __builtin_amdgcn_readlane(v.v, i);
#endif
        };
        LV<int> inf = info_of(0);
        for (int i = 0; i < mbw * mbh; i++) {
            const LV<int> nxt = info_of(i + 1);
            const int ym = rd(inf, 2), cm = rd(inf, 3);
            if (ym == 4) {
                e.put(0, 145);                                                                    // i4x4: sixteen sub-block modes, each after the modes above and to the left
                for (int k = 0; k < 16; k++) {
                    const int bx = k & 3, by = k >> 2, m = rd(inf, 4 + k);
                    const int tmode = by ? rd(inf, 4 + k - 4) : rd(inf, 32 + bx), lmode = bx ? rd(inf, 4 + k - 1) : rd(inf, 40 + by);
                    LV<int> prv;
                    LFOR(l) prv[l] = l < 9 ? int(s_bmp[(tmode * 10 + lmode) * 9 + l]) : 0;
                    // the key-frame sub-block mode tree (RFC 6386 11.2; oracle: bmode_path)
                    if (m == 0) e.put(0, rd(prv, 0));
                    else {
                        e.put(1, rd(prv, 0));
                        if (m == 1) e.put(0, rd(prv, 1));
                        else {
                            e.put(1, rd(prv, 1));
                            if (m == 2) e.put(0, rd(prv, 2));
                            else {
                                e.put(1, rd(prv, 2));
                                if (m <= 5) { e.put(0, rd(prv, 3)); if (m == 3) e.put(0, rd(prv, 4)); else { e.put(1, rd(prv, 4)); e.put(m == 5, rd(prv, 5)); } }
                                else { e.put(1, rd(prv, 3)); if (m == 6) e.put(0, rd(prv, 6)); else { e.put(1, rd(prv, 6)); if (m == 7) e.put(0, rd(prv, 7)); else { e.put(1, rd(prv, 7)); e.put(m == 9, rd(prv, 8)); } } }
                            }
                        }
                    }
                }
            } else {
                e.put(1, 145);                                                                    // i16x16
                if (ym >= 2) { e.put(1, 156); e.put(ym == 3, 128); } else { e.put(0, 156); e.put(ym == 1, 163); }   // (H | TM) : (DC | V)
            }
            if (!cm) e.put(0, 142); else { e.put(1, 142); if (cm == 1) e.put(0, 114); else { e.put(1, 114); e.put(cm == 3, 183); } }
            inf = nxt;
        }
    e.finish();
    LANE0 *size_out = e.overflow ? 0xFFFFFFFFu : e.pos;
}
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_webp_code(const WebpImg *imgs, const int16_t *levels, const uint8_t *probs_all, const uint8_t *update_all, uint8_t *scratch,
                                                                 uint32_t *part_size, const uint32_t *status) {
    const WebpImg im = imgs[blockIdx.x];
    if (status[im.image]) return;
    const int mbw = int(im.mbw), mbh = int(im.mbh), nparts = webp_parts(mbh), part = int(blockIdx.y) - 1;
    if (part >= nparts) return;
    const int16_t *lev = levels + im.lev_off;
    const uint8_t *probs_g = probs_all + size_t(im.image) * WEBP_NPROB, *update = update_all + size_t(im.image) * WEBP_NPROB;
    CSH_SHARED uint8_t s_probs[WEBP_NPROB], s_bmp[900];
    LFOR(l) for (int i = l; i < WEBP_NPROB; i += 64) s_probs[i] = probs_g[i];
    if (part < 0) { LFOR(l) for (int i = l; i < 900; i += 64) s_bmp[i] = kVp8BModeProbs[i]; }
    CSP_WAVE_SYNC();
    const uint8_t *probs = s_probs;
    uint8_t *base = scratch + im.out_off;
    if (part < 0) { code_header(im, lev, probs, update, s_bmp, base, &part_size[size_t(im.image) * 9]); return; }
    BoolEncTok e;
    {
        e.init(base + webp_hdr_cap(im) + uint32_t(part) * webp_part_cap(im), webp_part_cap(im));
        for (int my = part; my < mbh && !e.overflow; my += nparts)
            for (int mx = 0; mx < mbw; mx++) {
                const int16_t *L = lev + (size_t(my) * mbw + mx) * WEBP_MB_REC;
                const uint32_t cur = nz_mask(L), top = my ? nz_mask(L - size_t(mbw) * WEBP_MB_REC) : 0u, left = mx ? nz_mask(L - WEBP_MB_REC) : 0u;
                const bool i4 = L[MB_INFO + 2] == 4;
                CodeSink sink{e, probs, {}};
                for (int k = i4 ? 1 : 0; k < 25; k++) {
                    int type, first, ctx;
                    block_info(k, cur, top, left, i4, type, first, ctx);
                    put_coeffs(sink, type, ctx, L + k * 16, first);
                }
            }
    }
    e.finish();
    LANE0 part_size[size_t(im.image) * 9 + blockIdx.y] = e.overflow ? 0xFFFFFFFFu : e.pos;
}
// the file: RIFF / WEBP / "VP8 " headers, frame tag, start code, dimensions, header partition, the sizes of all token partitions
// but the last, the partitions
__global__ void __launch_bounds__(256) k_webp_assemble(const WebpImg *imgs, const uint8_t *scratch, const uint32_t *part_size, uint8_t *out, uint32_t *img_size, uint32_t *status) {
    const WebpImg im = imgs[blockIdx.x];
    if (status[im.image]) return;
    const int nparts = webp_parts(int(im.mbh));
    const uint32_t *ps = part_size + size_t(im.image) * 9;
    bool bad = false;
    uint32_t tok = 0;
    for (int k = 0; k <= nparts; k++) { if (ps[k] == 0xFFFFFFFFu) bad = true; else if (k) tok += ps[k]; }
    const uint32_t p0 = ps[0], vp8 = 10u + p0 + 3u * uint32_t(nparts - 1) + tok, total = 20u + vp8 + (vp8 & 1u);
    if (bad || total > im.out_cap) { if (threadIdx.x == 0) status[im.image] = 20200; return; }
    uint8_t *o = out + im.out_off;
    const uint8_t *base = scratch + im.out_off;
    if (threadIdx.x == 0) {
        const uint8_t hdr[20] = {'R', 'I', 'F', 'F', uint8_t(total - 8), uint8_t((total - 8) >> 8), uint8_t((total - 8) >> 16), uint8_t((total - 8) >> 24), 'W', 'E', 'B', 'P',
                                 'V', 'P', '8', ' ', uint8_t(vp8), uint8_t(vp8 >> 8), uint8_t(vp8 >> 16), uint8_t(vp8 >> 24)};
        for (int k = 0; k < 20; k++) o[k] = hdr[k];
        const uint32_t tag = (p0 << 5) | (1u << 4);   // key frame, version 0, shown
        o[20] = uint8_t(tag); o[21] = uint8_t(tag >> 8); o[22] = uint8_t(tag >> 16);
        o[23] = 0x9D; o[24] = 0x01; o[25] = 0x2A;
        o[26] = uint8_t(im.width); o[27] = uint8_t(im.width >> 8); o[28] = uint8_t(im.height); o[29] = uint8_t(im.height >> 8);
        for (int p = 0; p + 1 < nparts; p++) { uint8_t *z = o + 30 + p0 + 3 * p; z[0] = uint8_t(ps[1 + p]); z[1] = uint8_t(ps[1 + p] >> 8); z[2] = uint8_t(ps[1 + p] >> 16); }
        if (vp8 & 1u) o[20 + vp8] = 0;
        img_size[im.image] = total;
    }
    for (uint32_t i = threadIdx.x; i < p0; i += blockDim.x) o[30 + i] = base[i];
    uint32_t at = 30u + p0 + 3u * uint32_t(nparts - 1);
    for (int p = 0; p < nparts; p++) {
        const uint8_t *src = base + webp_hdr_cap(im) + uint32_t(p) * webp_part_cap(im);
        for (uint32_t i = threadIdx.x; i < ps[1 + p]; i += blockDim.x) o[at + i] = src[i];
        at += ps[1 + p];
    }
}

// ---- the token partitions as decision streams (round 4).  The boolean coder is a serial chain per partition, and a wave that walks one chain on its
// uniform side costs the chip's SCALAR issue rate: eight waves per picture, ~8 k scalar instructions per macroblock -- 137 ms per 1024 pictures, half of the
// JPEG -> WebP path.  What is serial, though, is only the arithmetic coder; WHICH decisions it takes (the token tree over the levels, the contexts out of the
// masks, the probabilities of the frame) is known for every block at once.  So:
//   k_webp_stats      (the walk that counts the frame's statistics) also counts every block's decisions;
//   an exclusive scan over the macroblocks in chain order gives every macroblock its place in the stream;
//   k_webp_decisions  the same walk, lanes = the blocks of a macroblock, writes (bit, probability) pairs -- two bytes a decision;
//   k_webp_bool       ONE LANE per partition runs the coder over its stretch of pairs: no tree, no table, no levels -- 64 chains to a wave on the vector unit;
//   k_webp_hdr        the header partition the same way (fields, probability updates, every macroblock's modes): its chain is a ninth lane per picture.
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_webp_decisions(const WebpImg *imgs, const int16_t *levels, const uint8_t *probs_all, const uint64_t *mb_base, const uint64_t *mb_off,
                                                                      const uint16_t *blk_cnt, uint16_t *stream, const uint32_t *status) {
    CSH_SHARED uint8_t s_probs[WEBP_NPROB];
    const WebpImg im = imgs[blockIdx.y];
    const int mbw = int(im.mbw), my = int(blockIdx.x);
    if (my >= int(im.mbh) || status[im.image]) return;
    const uint8_t *probs_g = probs_all + size_t(im.image) * WEBP_NPROB;
    LFOR(l) for (int i = l; i < WEBP_NPROB; i += 64) s_probs[i] = probs_g[i];
    CSP_WAVE_SYNC();
    const int16_t *row = levels + im.lev_off + size_t(my) * mbw * WEBP_MB_REC;
    for (int mx0 = 0; mx0 < mbw; mx0 += 2) {   // two macroblocks to a step, as in k_webp_stats
        LV<uint32_t> nd;
        LV<uint64_t> at;
        LFOR(l) {
            const int mx = mx0 + (l >> 5), k = l & 31;
            at[l] = mx < mbw ? mb_base[blockIdx.y] + webp_chain_index(mbw, int(im.mbh), mx, my) : 0ull;
            nd[l] = mx < mbw ? uint32_t(blk_cnt[at[l] * 32 + uint32_t(k)]) : 0u;
        }
        uint32_t total;
        const LV<uint32_t> ex = lscan(nd, total);
        const uint32_t lower = csh::lget(ex, 32);
        LFOR(l) {
            const int mx = mx0 + (l >> 5), k = l & 31;
            if (mx < mbw && k < 25) {
                const int16_t *L = row + size_t(mx) * WEBP_MB_REC;
                const bool i4 = L[MB_INFO + 2] == 4;
                if (!(i4 && k == 0)) {
                    const uint32_t cur = nz_mask(L), top = my ? nz_mask(L - size_t(mbw) * WEBP_MB_REC) : 0u, left = mx ? nz_mask(L - WEBP_MB_REC) : 0u;
                    int type, first, ctx;
                    block_info(k, cur, top, left, i4, type, first, ctx);
                    WriteSink sink{s_probs, stream + mb_off[at[l]] + (ex[l] - ((l >> 5) ? lower : 0u))};
                    put_coeffs(sink, type, ctx, L + k * 16, first);
                }
            }
        }
    }
}
// ---- the header partition the same way: its chain is the frame header's fields, the frame's probability updates, then every macroblock's modes in raster
// order.  One lane per ITEM (item 0: the fields and the updates; item 1 + i: macroblock i) counts its decisions or writes them: the modes' contexts are the
// neighbours' modes, which k_webp_mb left with the levels.  (oracle: write_frame_header / the mode part of cso_webp_encode_yuv; the wave form: code_header)
struct HdrSink {
    uint16_t *out;   // nullptr: count only
    uint32_t n;
    __device__ __forceinline__ void put(int bit, int prob) { if (out) *out++ = uint16_t((bit ? 1u : 0u) | (uint32_t(prob) << 1)); n++; }
    __device__ __forceinline__ void bits(uint32_t v, int nb) { while (nb--) put(int((v >> nb) & 1u), 128); }
};
__device__ static void hdr_item(HdrSink &e, const WebpImg &im, const int16_t *lev, const uint8_t *probs, const uint8_t *update, uint32_t item) {
    const int mbw = int(im.mbw), mbh = int(im.mbh), nparts = webp_parts(mbh);
    if (item == 0) {
        e.bits(0, 1); e.bits(0, 1); e.bits(0, 1);           // colour space, clamping, no segmentation
        e.bits(1, 1); e.bits(0, 6); e.bits(0, 3);           // simple filter at level 0 (off), sharpness
        e.bits(0, 1);                                       // no filter deltas
        e.bits(uint32_t(nparts == 8 ? 3 : nparts == 4 ? 2 : nparts == 2 ? 1 : 0), 2);
        e.bits(uint32_t(im.qi), 7);
        for (int i = 0; i < 5; i++) e.bits(0, 1);           // no quantiser deltas
        e.bits(0, 1);                                       // refresh_entropy_probs
        for (int i = 0; i < WEBP_NPROB; i++) { e.put(update[i], kVp8CoefUpdateProbs[i]); if (update[i]) e.bits(probs[i], 8); }   // the frame's coefficient probabilities
        e.bits(0, 1);                                       // no skip flags
        return;
    }
    const int i = int(item) - 1, my = i / mbw, mx = i - my * mbw;
    const int16_t *I = lev + size_t(i) * WEBP_MB_REC + MB_INFO, *It = I - size_t(mbw) * WEBP_MB_REC, *Il = I - WEBP_MB_REC;
    const int ym = I[2], cm = I[3];
    if (ym == 4) {
        e.put(0, 145);                                                                    // i4x4: sixteen sub-block modes, each after the modes above and to the left
        for (int k = 0; k < 16; k++) {
            const int bx = k & 3, by = k >> 2, m = I[4 + k];
            const int tmode = by ? int(I[4 + k - 4]) : (my ? int(It[4 + 12 + bx]) : 0), lmode = bx ? int(I[4 + k - 1]) : (mx ? int(Il[4 + by * 4 + 3]) : 0);
            const uint8_t *prv = kVp8BModeProbs + (tmode * 10 + lmode) * 9;
            // the key-frame sub-block mode tree (RFC 6386 11.2; oracle: bmode_path)
            if (m == 0) e.put(0, prv[0]);
            else {
                e.put(1, prv[0]);
                if (m == 1) e.put(0, prv[1]);
                else {
                    e.put(1, prv[1]);
                    if (m == 2) e.put(0, prv[2]);
                    else {
                        e.put(1, prv[2]);
                        if (m <= 5) { e.put(0, prv[3]); if (m == 3) e.put(0, prv[4]); else { e.put(1, prv[4]); e.put(m == 5, prv[5]); } }
                        else { e.put(1, prv[3]); if (m == 6) e.put(0, prv[6]); else { e.put(1, prv[6]); if (m == 7) e.put(0, prv[7]); else { e.put(1, prv[7]); e.put(m == 9, prv[8]); } } }
                    }
                }
            }
        }
    } else {
        e.put(1, 145);                                                                    // i16x16
        if (ym >= 2) { e.put(1, 156); e.put(ym == 3, 128); } else { e.put(0, 156); e.put(ym == 1, 163); }   // (H | TM) : (DC | V)
    }
    if (!cm) e.put(0, 142); else { e.put(1, 142); if (cm == 1) e.put(0, 114); else { e.put(1, 114); e.put(cm == 3, 183); } }
}
// hdr_base[image]: the picture's first header item among all counted things (behind the macroblocks in chain order); WRITE: cnt is the scan's output
template <bool WRITE>
__global__ void __launch_bounds__(256) k_webp_hdr(const WebpImg *imgs, const int16_t *levels, const uint8_t *probs_all, const uint8_t *update_all, const uint64_t *hdr_base, uint32_t *cnt,
                                                  const uint64_t *off, uint16_t *stream, const uint32_t *status) {
    const WebpImg &im = imgs[blockIdx.y];
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item > im.mbw * im.mbh) return;
    if (WRITE && status[im.image]) return;
    const uint64_t at = hdr_base[blockIdx.y] + item;
    HdrSink e{WRITE ? stream + off[at] : nullptr, 0u};
    hdr_item(e, im, levels + im.lev_off, probs_all + size_t(im.image) * WEBP_NPROB, update_all + size_t(im.image) * WEBP_NPROB, item);
    if (!WRITE) cnt[at] = e.n;
}

// one lane = one chain (token partition `part` of picture `image`: the decisions of nmb macroblocks from chain-order macroblock `first` on)
struct WebpChain { uint32_t image, part; uint64_t first, nmb; };   // part 0xFFFFFFFF: the header partition (first / nmb: its items)
struct BoolEncLane {   // the boolean coder (oracle: boolenc) with per-lane state.  Sixty-four of these run side by side in a wave, every lane at its own place in its own
                       // chain, and whatever one lane branches into, the whole wave executes.  So: a decision is straight-line arithmetic (the renormalisation
                       // shifts by 0 when none is due); the bits a decision completes stay in a 64-bit accumulator and leave on a FIXED schedule (flush() after
                       // every fourth decision: every lane emits its zero to four complete bytes then -- the same digits the byte-at-a-time coder produces, a
                       // carry that it would have patched into a written byte is simply added before the byte is written); and the output never READS (a carry
                       // that does reach a written byte goes into the last one, which the lane still holds in a register)
    uint8_t *buf;
    uint32_t pos, cap;
    int32_t range;
    uint64_t value;
    int run, nb_bits;
    uint32_t last;
    bool overflow;
    __device__ __forceinline__ void init(uint8_t *b, uint32_t c) { buf = b; pos = 0; cap = c; run = 0; nb_bits = -8; overflow = false; range = 254; value = 0; last = 0; }
    __device__ __forceinline__ void flush() {
        while (nb_bits > 0) {
            const int s = 8 + nb_bits;
            const uint32_t bits = uint32_t(value >> s);   // eight bits and a carry
            value -= uint64_t(bits) << s;
            nb_bits -= 8;
            if ((bits & 0xffu) != 0xffu) {
                if (pos + uint32_t(run) + 1 > cap) { overflow = true; run = 0; nb_bits = -8; value = 0; return; }
                if ((bits & 0x100u) && pos > 0) { last = (last + 1u) & 0xffu; buf[pos - 1] = uint8_t(last); }   // (the byte in front of a run of 0xff is never 0xff itself)
                const uint8_t v = (bits & 0x100u) ? 0x00 : 0xff;
                for (int k = 0; k < run; k++) buf[pos + uint32_t(k)] = v;
                last = bits & 0xffu;
                buf[pos + uint32_t(run)] = uint8_t(last);
                pos += uint32_t(run) + 1; run = 0;
            } else
                run++;
        }
    }
    __device__ __forceinline__ void put(int bit, int prob) {   // at most four of these between two flush(): 8 + 8 + 4 x 7 + 1 bits fit the accumulator many times over
        const int32_t split = (range * prob) >> 8, m = -int32_t(bit != 0);
        value += uint64_t(uint32_t((split + 1) & m));
        range = split + ((range - 2 * split - 1) & m);               // bit ? range - split - 1 : split
        const int shift = __clz(uint32_t(range + 1)) - 24;          // 0 for range >= 127: no renormalisation due
        range = ((range + 1) << shift) - 1;
        value <<= shift;
        nb_bits += shift;
    }
    __device__ __forceinline__ void finish() {
        flush();
        for (int n = 9 - nb_bits; n > 0; n--) { put(0, 128); flush(); }
        nb_bits = 0;
        {   // the last byte, as the byte-at-a-time coder flushes it (nb_bits = 0: s = 8)
            const int s = 8;
            const uint32_t bits = uint32_t(value >> s);
            value -= uint64_t(bits) << s;
            nb_bits = -8;
            if ((bits & 0xffu) != 0xffu) {
                if (pos + uint32_t(run) + 1 > cap) { overflow = true; run = 0; return; }
                if ((bits & 0x100u) && pos > 0) { last = (last + 1u) & 0xffu; buf[pos - 1] = uint8_t(last); }
                const uint8_t v = (bits & 0x100u) ? 0x00 : 0xff;
                for (int k = 0; k < run; k++) buf[pos + uint32_t(k)] = v;
                last = bits & 0xffu;
                buf[pos + uint32_t(run)] = uint8_t(last);
                pos += uint32_t(run) + 1; run = 0;
            } else
                run++;
        }
    }
};
__global__ void __launch_bounds__(64) k_webp_bool(const WebpImg *imgs, const WebpChain *chains, uint32_t nchains, const uint64_t *mb_off, const uint16_t *stream, uint8_t *scratch,
                                                  uint32_t *part_size, const uint32_t *status) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    const WebpChain ch = chains[c];
    const WebpImg &im = imgs[ch.image];   // (chains are listed per WebpImg entry)
    if (status[im.image]) return;
    BoolEncLane e;
    const bool header = ch.part == 0xFFFFFFFFu;
    if (header) e.init(scratch + im.out_off, webp_hdr_cap(im)); else e.init(scratch + im.out_off + webp_hdr_cap(im) + ch.part * webp_part_cap(im), webp_part_cap(im));
    const uint64_t d0 = mb_off[ch.first], d1 = mb_off[ch.first + ch.nmb];
    // eight decisions to a 16-byte load, the next load in flight while these are coded (a lane's loads are its own: nothing hides their latency but this)
    uint64_t d = d0;
    for (; d < d1 && (d & 7u) && !e.overflow; d++) { const uint32_t v = stream[d]; e.put(int(v & 1u), int(v >> 1)); e.flush(); }
    if (d + 8 <= d1 && !e.overflow) {
        uint4 nxt = *reinterpret_cast<const uint4 *>(stream + d);
        for (; d + 8 <= d1 && !e.overflow; d += 8) {
            const uint4 cur = nxt;
            if (d + 16 <= d1) nxt = *reinterpret_cast<const uint4 *>(stream + d + 8);
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
            CSH_UNROLL
            for (int k = 0; k < 4; k++) {
                e.put(int(w[k] & 1u), int((w[k] >> 1) & 0x7FFFu));
                e.put(int((w[k] >> 16) & 1u), int(w[k] >> 17));
                if (k & 1) e.flush();
            }
        }
    }
    for (; d < d1 && !e.overflow; d++) { const uint32_t v = stream[d]; e.put(int(v & 1u), int(v >> 1)); e.flush(); }
    e.finish();
    part_size[size_t(im.image) * 9 + (header ? 0u : 1u + ch.part)] = e.overflow ? 0xFFFFFFFFu : e.pos;
}

void launch_webp_yuv(hipStream_t st, const WebpImg *imgs, int nimg, uint32_t max_luma, const uint8_t *rgb, uint8_t *work) {
    if (nimg && max_luma) CSH_LAUNCH(k_webp_yuv, dim3((max_luma + 255) / 256, nimg), dim3(256), st, imgs, rgb, work);
}
void launch_webp_mb(hipStream_t st, const WebpImg *imgs, int nimg, uint32_t max_mbw, uint32_t max_mbh, uint8_t *work, int16_t *levels) {   // one launch per skewed diagonal mx + 2 my
    if (!nimg || !max_mbw || !max_mbh) return;
    for (uint32_t d = 0; d + 2 < max_mbw + 2 * max_mbh; d++) CSH_LAUNCH(k_webp_mb, dim3(max_mbh, unsigned(nimg)), dim3(CSP_WAVE_THREADS), st, imgs, work, levels, int(d));
}
void launch_webp_code(hipStream_t st, const WebpImg *imgs, const WebpImg *himgs, int nimg, uint32_t max_mbh, const int16_t *levels, uint32_t *stats, uint8_t *probs, uint8_t *update,
                      uint8_t *scratch, uint32_t *part_size, uint8_t *out, uint32_t *img_size, uint32_t *status) {
    if (!nimg) return;
    const char *sw = getenv("CSH_WEBP_CHAINS");   // "1": every token partition on a wave of its own (k_webp_code), as before round 4
    const bool streams = !(sw && !strcmp(sw, "1"));
    // the macroblocks of the batch in chain order (picture by picture, inside a picture partition by partition) and the chains themselves
    std::vector<uint64_t> base(size_t(nimg) + 1), hbase(size_t(nimg) + 1);
    std::vector<WebpChain> chains;
    uint64_t nmb = 0;
    uint32_t max_items = 0;
    for (int i = 0; i < nimg; i++) {
        const uint32_t mbw = himgs[i].mbw, mbh = himgs[i].mbh, P = uint32_t(mbh >= 8 ? 8 : mbh >= 4 ? 4 : mbh >= 2 ? 2 : 1);   // webp_parts
        base[size_t(i)] = nmb;
        uint64_t first = nmb;
        for (uint32_t p = 0; p < P; p++) { const uint64_t n = uint64_t((mbh - p + P - 1) / P) * mbw; chains.push_back(WebpChain{uint32_t(i), p, first, n}); first += n; }
        nmb += uint64_t(mbw) * mbh;
        max_items = std::max(max_items, mbw * mbh + 1u);
    }
    base[size_t(nimg)] = nmb;
    // behind them the header partitions' items: per picture the fields + updates, then its macroblocks in raster order
    uint64_t nall = nmb;
    for (int i = 0; i < nimg; i++) {
        const uint64_t n = uint64_t(himgs[i].mbw) * himgs[i].mbh + 1;
        hbase[size_t(i)] = nall;
        if (streams) chains.push_back(WebpChain{uint32_t(i), 0xFFFFFFFFu, nall, n});
        nall += n;
    }
    hbase[size_t(nimg)] = nall;
    csh::DevBuf<uint64_t> d_base, d_hbase, d_off;
    csh::DevBuf<uint32_t> d_cnt;
    csh::DevBuf<uint16_t> d_blk, d_stream;
    csh::DevBuf<WebpChain> d_chains;
    csh::DevBuf<uint8_t> d_tmp;
    const size_t tmp_bytes = csh::exclusive_scan_tmp_bytes(nall);
    const bool room = streams && !(d_base.upload(base, st) || d_hbase.upload(hbase, st) || d_chains.upload(chains, st) || d_cnt.alloc(nall + 1) || d_off.alloc(nall + 2) || d_blk.alloc((nmb + 1) * 32) ||
                                   d_tmp.alloc(tmp_bytes + 64));
    CSH_LAUNCH(k_webp_stats, dim3(max_mbh, nimg), dim3(CSP_WAVE_THREADS), st, imgs, levels, stats, room ? d_base.p : nullptr, room ? d_cnt.p : nullptr, room ? d_blk.p : nullptr);
    CSH_LAUNCH(k_webp_probs, dim3((WEBP_NPROB + 255) / 256, nimg), dim3(256), st, imgs, stats, probs, update);
    if (!room) CSH_LAUNCH(k_webp_code, dim3(nimg, 9), dim3(CSP_WAVE_THREADS), st, imgs, levels, probs, update, scratch, part_size, status);
    else {
        const dim3 items((max_items + 255) / 256, unsigned(nimg));
        CSH_LAUNCH(k_webp_hdr<false>, items, dim3(256), st, imgs, levels, probs, update, d_hbase.p, d_cnt.p, d_off.p, d_stream.p, status);
        csh::launch_exclusive_scan(st, d_cnt.p, d_off.p, nall, d_tmp.p, tmp_bytes + 64);
        uint64_t total = 0;
        if (csh_copy_wait(&total, d_off.p + nall, sizeof total, hipMemcpyDeviceToHost, st) == hipSuccess && !d_stream.alloc(size_t(total) + 64)) {
            CSH_LAUNCH(k_webp_decisions, dim3(max_mbh, nimg), dim3(CSP_WAVE_THREADS), st, imgs, levels, probs, d_base.p, d_off.p, d_blk.p, d_stream.p, status);
            CSH_LAUNCH(k_webp_hdr<true>, items, dim3(256), st, imgs, levels, probs, update, d_hbase.p, d_cnt.p, d_off.p, d_stream.p, status);
            const uint32_t nchains = uint32_t(chains.size());
            CSH_LAUNCH(k_webp_bool, dim3((nchains + 63) / 64), dim3(64), st, imgs, d_chains.p, nchains, d_off.p, d_stream.p, scratch, part_size, status);
        } else   // no room for the pairs (two bytes a decision): the partitions as chains, a wave each -- the same files, later
            CSH_LAUNCH(k_webp_code, dim3(nimg, 9), dim3(CSP_WAVE_THREADS), st, imgs, levels, probs, update, scratch, part_size, status);
    }
    CSH_LAUNCH(k_webp_assemble, dim3(nimg), dim3(256), st, imgs, scratch, part_size, out, img_size, status);
    (void)hipStreamSynchronize(st);
}

}  // namespace csw
