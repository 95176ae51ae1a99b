// vp8enc_dev.h -- what the kernels of the lossy WebP encoder share (k_vp8enc.hip: analysis, segments, the macroblock loop, statistics;
// k_webp.hip: RGB -> YUV and the coder back end).  Statement: oracle/vp8enc_oracle.c = libwebp's encoder at its defaults.
#pragma once
#include "../../include/vp8_cost_tables.h"
#include "../../include/vp8_tables.h"
#include "png_wave.h"
#include "wave.h"
#include "webp_kernels.h"

namespace csw {
using namespace csp;   // LFOR / LV / lballot / lscan (png_wave.h)

enum { VP8_NSLOT = 4 * 8 * 3 * 11, VP8_MAXLV = 67 };

// ---- the macroblock record: 25 blocks x 16 levels (Y2, 16 luma, 4 U, 4 V; scan order) + an info block of 32 int16
//   I[0], I[1]  which blocks have anything to code: bit 0 the Y2 flag as the macroblock to the RIGHT sees it, 1..16 luma, 17..24 chroma, bit 25 the
//               Y2 flag as the macroblock BELOW sees it (an i4x4 macroblock has no Y2 block and hands its neighbours' flags on: the two differ)
//   I[2] luma mode in libwebp's numbering (0 DC, 1 TM, 2 V, 3 H; 4 = i4x4), I[3] chroma mode (same numbering), I[4..19] the sixteen sub-block modes
//   (i16: the mode itself, which is what it counts as in its neighbours' sub-block contexts), I[20] susceptibility (analysis), I[21] segment,
//   I[22], I[23] the chroma DC errors handed down / to the right ([U, V] x 2 signed bytes each)
enum { MB_INFO = 400, MB_ALPHA = MB_INFO + 20, MB_SEG = MB_INFO + 21, MB_DERR_TOP = MB_INFO + 22, MB_DERR_LEFT = MB_INFO + 24 };
static_assert(WEBP_MB_REC >= MB_INFO + 28, "macroblock record");
__device__ __forceinline__ static uint32_t nz_mask(const int16_t *L) { return uint32_t(uint16_t(L[MB_INFO])) | (uint32_t(uint16_t(L[MB_INFO + 1])) << 16); }

struct Vp8SegDev {
    int32_t quant, fstrength, alpha, beta, max_edge, min_disto;
    int32_t lambda_i16, lambda_i4, lambda_uv, lambda_mode, tlambda, pad;
    int32_t q[3][2], iq[3][2], bias[3][2], zth[3][2];   // [Y1, Y2, UV][DC, AC]
};
// one per picture, in device memory
struct Vp8FrameDev {
    uint32_t hist[256];                 // susceptibility histogram (k_vp8_analyse)
    int32_t alpha_sum, uv_alpha_sum;
    int32_t nseg, update_map, seg_probs[3], base_quant, dq_uv_dc, dq_uv_ac, filter_level, dirty, diffuse, pad0;
    uint8_t alpha_seg[256];             // susceptibility -> segment
    Vp8SegDev seg[4];
    uint32_t stats[VP8_NSLOT];          // hi 16: events, lo 16: ones -- libwebp's proba_t with its halving
    uint8_t coeffs[VP8_NSLOT];          // at the end of the walk: the frame's probabilities
};
__device__ __forceinline__ static int vp8_slot(int t, int b, int c) { return ((t * 8 + b) * 3 + c) * 11; }
__device__ __forceinline__ static int vp8_bitcost(int bit, int p) { return kVp8EntropyCost[bit ? 255 - p : p]; }

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

// the token walk of one block (oracle: record_block; RFC 6386 13).  S::ad(bit, slot) an adaptive decision, S::ad10 the second category bit of the two big
// categories (libwebp counts it in the statistics of slot - 1: the coder still uses the slot's own probability), S::fx(bit, prob) a fixed-probability one
template <class S>
__device__ static int put_coeffs(S &e, int type, int ctx, const int16_t *lv, int first) {
    int last = -1;
    for (int i = first; i < 16; i++) if (lv[i]) last = i;
    int n = first;
    int p = vp8_slot(type, kVp8Bands[n], ctx);
    if (last < 0) { e.ad(0, p + 0); return 0; }
    e.ad(1, p + 0);
    while (n < 16) {
        const int c = lv[n++];
        const int sign = c < 0;
        int v = sign ? -c : c;
        if (!v) { e.ad(0, p + 1); p = vp8_slot(type, kVp8Bands[n], 0); continue; }
        e.ad(1, p + 1);
        if (v == 1) { e.ad(0, p + 2); p = vp8_slot(type, kVp8Bands[n], 1); }
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
                else if (v < 3 + (8 << 3)) { e.ad(1, p + 8); e.ad10(0, p + 10); v -= 3 + (8 << 2); mask = 1 << 4; tab = kVp8Cat5; }
                else { e.ad(1, p + 8); e.ad10(1, p + 10); v -= 3 + (8 << 3); mask = 1 << 10; tab = kVp8Cat6; }
                while (mask) { e.fx(!!(v & mask), *tab++); mask >>= 1; }
            }
            p = vp8_slot(type, kVp8Bands[n], 2);
        }
        e.fx(sign, 128);
        if (n == 16) return 1;
        if (n > last) { e.ad(0, p + 0); return 1; }
        e.ad(1, p + 0);
    }
    return 1;
}

}  // namespace csw
