// k_vp8l_enc.hip -- lossless WebP OUTPUT (webp.lossless: libcaesium's webp::compress encodes with libwebp's lossless coder,
// /root/reference/src/compressor.rs:427-429 sets it under --lossless; call sites compressor.rs:289-305).  A VP8L stream per picture, made
// of the format's tools that are data-parallel: subtract-green, the spatial predictor (best of the 14 modes per 16 x 16 block by the sum
// of absolute residuals -- every mode predicts from ORIGINAL neighbours, so all pixels are independent), and one group of optimal prefix
// codes limited to 15 bits (png_codes.h, shared with the DEFLATE coder).  No backward references, colour cache or meta prefix image:
// what libwebp's coder adds on top is a few per cent on photographs and is a serial search; the bytes differ from libwebp's either way
// ("parity unpinned"), the invariant that is pinned is the format's: libwebp decodes the file to exactly the source pixels
// (tests/test_webp_lossless*.py), and so does this repo's own decoder (vp8l_dec.h).
//   k_vp8l_residuals  one workgroup per block: mode choice, residual ARGB
//   k_vp8l_hist       symbol counts of the four channels (an opaque source's alpha residuals are all one value: a code without bits)
//   k_vp8l_pack       one wave per picture: code lengths, canonical codes, the headers, then every pixel's three codes through the LDS
//                     bit window straight to their place, RIFF framing
#include "webp_kernels.h"
#include "png_codes.h"
#include "vp8l_dec.h"

namespace csw {

using csp::LV;

__device__ __forceinline__ static uint32_t sg_pixel(const Vp8lImg &im, uint32_t x, uint32_t y) {   // ARGB after subtract-green
    // channels: 1 grey, 2 grey + alpha, 3 RGB, 4 RGBA; VP8L_ALPHA_OF + 2 / + 4: the ALPHA sample of such a picture taken as a grey picture (the ALPH
    // chunk of a lossy file is a VP8L stream whose green channel is the alpha plane)
    const uint32_t pick = im.channels >= VP8L_ALPHA_OF ? im.channels - VP8L_ALPHA_OF : 0u, nc = pick ? pick : im.channels;
    const uint8_t *p = im.rgb + (uint64_t(y) * im.width + x) * nc;
    if (pick) return 0xFF000000u | (uint32_t(p[nc - 1]) << 8);
    if (nc <= 2) return (nc == 2 ? uint32_t(p[1]) << 24 : 0xFF000000u) | (uint32_t(p[0]) << 8);   // grey: red - green = blue - green = 0
    const uint32_t r = p[0], g = p[1], b = p[2];
    return (nc == 4 ? uint32_t(p[3]) << 24 : 0xFF000000u) | (((r - g) & 255u) << 16) | (g << 8) | ((b - g) & 255u);
}
__device__ __forceinline__ static uint32_t lsub(uint32_t a, uint32_t b) {   // per-channel a - b mod 256
    return (((a | 0x00FF00FFu) - (b & 0xFF00FF00u)) & 0xFF00FF00u) | (((a | 0xFF00FF00u) - (b & 0x00FF00FFu)) & 0x00FF00FFu);
}
__device__ __forceinline__ static uint32_t res_cost(uint32_t r) {   // sum over the channels of the residual's distance from 0 (mod 256)
    uint32_t c = 0;
    for (int s = 0; s < 32; s += 8) { const uint32_t v = (r >> s) & 255u; c += v < 128u ? v : 256u - v; }
    return c;
}

// the predictor of pixel (x, y): libwebp's frame rules (first pixel: black; first row: left; first column: top), else the block's mode.
// The top-right neighbour of a row's last pixel is the first pixel of the row itself (the decoder reads one past the row above).
__device__ __forceinline__ static uint32_t predict_at(const Vp8lImg &im, uint32_t x, uint32_t y, int mode) {
    if (y == 0) return x == 0 ? 0xFF000000u : sg_pixel(im, x - 1, 0);
    if (x == 0) return sg_pixel(im, 0, y - 1);
    const uint32_t L = sg_pixel(im, x - 1, y), T = sg_pixel(im, x, y - 1), TL = sg_pixel(im, x - 1, y - 1);
    const uint32_t TR = x + 1 < im.width ? sg_pixel(im, x + 1, y - 1) : sg_pixel(im, 0, y);
    return lpredict_vals(mode, L, T, TR, TL);
}

__global__ void __launch_bounds__(256) k_vp8l_residuals(const Vp8lImg *imgs, uint32_t *work, uint8_t *modes) {
    CSH_SHARED uint32_t s_cost[16];
    CSH_SHARED uint32_t s_mode;
    const Vp8lImg &im = imgs[blockIdx.y];
    const uint32_t blk = blockIdx.x;
    const uint32_t by = blk / im.bw, bx = blk - by * im.bw;
    const uint32_t tid = threadIdx.x, x = bx * 16u + (tid & 15u), y = by * 16u + (tid >> 4);
    const bool inside = blk < im.bw * im.bh && x < im.width && y < im.height;
    CSH_PHASE_LOOP(4) {
        if (blk >= im.bw * im.bh) continue;
        if (phase == 0) { if (tid < 16) s_cost[tid] = 0; continue; }
        if (phase == 1) {
            if (inside && x && y) {   // the frame's first row and column are predicted the same way whatever the mode
                const uint32_t me = sg_pixel(im, x, y);
                const uint32_t L = sg_pixel(im, x - 1, y), T = sg_pixel(im, x, y - 1), TL = sg_pixel(im, x - 1, y - 1);
                const uint32_t TR = x + 1 < im.width ? sg_pixel(im, x + 1, y - 1) : sg_pixel(im, 0, y);
                for (int m = 0; m < 14; m++) atomicAdd(&s_cost[m], res_cost(lsub(me, lpredict_vals(m, L, T, TR, TL))));
            }
            continue;
        }
        if (phase == 2) {
            if (tid == 0) {
                uint32_t best = 0;
                for (uint32_t m = 1; m < 14; m++) if (s_cost[m] < s_cost[best]) best = m;
                s_mode = best; modes[im.mode_off + blk] = uint8_t(best);
            }
            continue;
        }
        if (inside) work[im.res_off + uint64_t(y) * im.width + x] = lsub(sg_pixel(im, x, y), predict_at(im, x, y, int(s_mode)));
    }
}

__global__ void __launch_bounds__(256) k_vp8l_hist(const Vp8lImg *imgs, const uint32_t *work, uint32_t *hist) {
    CSH_SHARED uint32_t h[4 * 256];
    const Vp8lImg &im = imgs[blockIdx.y];
    const uint64_t npx = uint64_t(im.width) * im.height, i0 = uint64_t(blockIdx.x) * 4096u;
    CSH_PHASE_LOOP(3) {
        if (i0 >= npx) continue;
        if (phase == 0) { for (uint32_t i = threadIdx.x; i < 1024; i += 256) h[i] = 0; continue; }
        if (phase == 1) {
            for (uint32_t k = threadIdx.x; k < 4096; k += 256) {
                const uint64_t i = i0 + k;
                if (i >= npx) break;
                const uint32_t v = work[im.res_off + i];
                atomicAdd(&h[(v >> 8) & 255u], 1u); atomicAdd(&h[256 + ((v >> 16) & 255u)], 1u); atomicAdd(&h[512 + (v & 255u)], 1u); atomicAdd(&h[768 + (v >> 24)], 1u);
            }
            continue;
        }
        for (uint32_t i = threadIdx.x; i < 1024; i += 256) if (h[i]) atomicAdd(&hist[uint64_t(blockIdx.y) * 1024u + i], h[i]);
    }
}

// ---- one wave per picture
struct PackLds {
    uint8_t len[5][288];       // green (alphabet 280), red, blue, modes (alphabet 280), alpha
    uint16_t code[5][288];
    uint32_t mh[288];          // histogram of the modes, as a green alphabet
    uint32_t gh[288];          // the green histogram widened to its alphabet (256 literals + 24 length prefixes that are never used)
    uint32_t win[160];
    uint32_t nused[5], sym0[5], sym1[5], last[5];   // per code: symbols with a non-zero count, the first two of them, the highest
};
__device__ __forceinline__ static uint32_t rev4(uint32_t v) { return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3); }

__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_vp8l_pack(const Vp8lImg *imgs, int nimg, const uint32_t *work, const uint8_t *modes, const uint32_t *hist, uint8_t *outp, uint32_t *file_len,
                                                                uint32_t *status) {
    CSH_SHARED PackLds S;
    const int image = blockIdx.x;
    if (image >= nimg) return;
    const Vp8lImg im = imgs[image];
    uint8_t *file = outp + im.out_off;
    const uint32_t *hh = hist + uint64_t(image) * 1024u;
    const uint32_t nblk = im.bw * im.bh;
    LFOR(l) for (int i = l; i < 288; i += 64) { S.mh[i] = 0; S.gh[i] = i < 256 ? hh[i] : 0u; }
    LFOR(l) for (int i = l; i < 160; i += 64) S.win[i] = 0;
    CSP_WAVE_SYNC();
    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) LFOR(l) if (b0 + uint32_t(l) < nblk) atomicAdd(&S.mh[modes[im.mode_off + b0 + uint32_t(l)]], 1u);
    CSP_WAVE_SYNC();
    // five codes, one lane each (the arrays of code_lengths live in scratch)
    LFOR(l) if (l < 5) {
        const int n = (l == 0 || l == 3) ? 280 : 256;
        const uint32_t *f = l == 0 ? S.gh : l == 3 ? S.mh : l == 4 ? hh + 768u : hh + 256u * uint32_t(l);
        csp::code_lengths(f, n, 15, S.len[l]);
        uint32_t used = 0, s0 = 0, s1 = 0, hi = 0;
        for (int i = 0; i < n; i++) if (f[i]) { if (used == 0) s0 = uint32_t(i); else if (used == 1) s1 = uint32_t(i); used++; hi = uint32_t(i); }
        if (used <= 1) for (int i = 0; i < n; i++) S.len[l][i] = 0;   // a code with one symbol costs no bits (code_lengths always codes two)
        csp::canonical(S.len[l], n, S.code[l]);
        S.nused[l] = used; S.sym0[l] = s0; S.sym1[l] = s1; S.last[l] = hi;
    }
    CSP_WAVE_SYNC();
    csp::BitOut bo;
    bo.win = S.win; bo.out = file + 20; bo.bitpos = 0; bo.wbase = 0;
    auto put1 = [&](uint64_t v, uint32_t n) __attribute__((always_inline)) {   // one field, from the first lane
        LV<uint64_t> val; LV<uint32_t> nb;
        LFOR(l) { val[l] = l == 0 ? v : 0ull; nb[l] = l == 0 ? n : 0u; }
        bo.put(val, nb);
    };
    // a code with one or two symbols is written as such (8-bit symbol fields); any other the long way: the code-length code gives the lengths
    // 0..15 four bits each and the run-length symbols 16..18 none, the lengths are cut behind the last symbol in use
    auto put_code = [&](int t) __attribute__((always_inline)) {
        const uint32_t used = S.nused[t];
        if (used <= 2) {
            const uint64_t two = used == 2 ? 1 : 0, wide = S.sym0[t] > 1 ? 1 : 0;   // the first symbol's field is one bit wide when that is enough
            const uint32_t w0 = wide ? 8u : 1u;
            put1(1ull | (two << 1) | (wide << 2) | (uint64_t(S.sym0[t]) << 3) | (two ? uint64_t(S.sym1[t]) << (3 + w0) : 0ull), 3 + w0 + (two ? 8u : 0u));
            return;
        }
        put1(0, 1);          // not a simple code
        put1(15, 4);         // 19 code-length code lengths follow, in the format's order 17 18 0 1 2 3 4 5 16 6 .. 15
        put1((4ull << 6) | (4ull << 9) | (4ull << 12) | (4ull << 15) | (4ull << 18) | (4ull << 21) | (4ull << 27) | (4ull << 30) | (4ull << 33) | (4ull << 36) | (4ull << 39), 42);   // 14 of the 19
        put1(4ull | (4ull << 3) | (4ull << 6) | (4ull << 9) | (4ull << 12), 15);                                                                                                   // symbols 11 .. 15
        const int n = int(S.last[t]) + 1;   // >= 3 here
        put1(1ull | (4ull << 1) | (uint64_t(n - 2) << 4), 14);   // the number of lengths that follow: a 10-bit field (2 + 2 * 4), holding n - 2
        for (int i0 = 0; i0 < n; i0 += 64) {
            LV<uint64_t> val; LV<uint32_t> nb;
            LFOR(l) { const int i = i0 + l; nb[l] = i < n ? 4u : 0u; val[l] = i < n ? rev4(S.len[t][i]) : 0u; }
            bo.put(val, nb);
        }
    };
    auto put_single = [&]() __attribute__((always_inline)) { put1(1 | (0 << 1) | (0 << 2) | (0 << 3), 4); };   // simple code, one symbol, 1-bit symbol field, symbol 0
    put1(0x2F, 8);
    const bool has_alpha = im.channels == 2 || im.channels == 4;
    put1(uint64_t(im.width - 1) | (uint64_t(im.height - 1) << 14) | (uint64_t(has_alpha ? 1 : 0) << 28) | (0ull << 29), 32);   // sizes, alpha_is_used (a hint), version 0
    put1(1 | (2u << 1), 3);                      // a transform follows: subtract green
    put1(1 | (0u << 1) | (2u << 3), 6);          // a transform follows: predictor, block side 1 << (2 + 2)
    put1(0, 1);                                   // the mode image: no colour cache
    put_code(3); put_single(); put_single(); put_single(); put_single();
    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) {
        LV<uint64_t> val; LV<uint32_t> nb;
        LFOR(l) {
            const uint32_t b = b0 + uint32_t(l);
            const uint32_t m = b < nblk ? modes[im.mode_off + b] : 0u;
            nb[l] = b < nblk ? S.len[3][m] : 0u; val[l] = S.code[3][m];
        }
        bo.put(val, nb);
    }
    put1(0, 1);                                   // no further transform
    put1(0, 1);                                   // the picture: no colour cache
    put1(0, 1);                                   // no meta prefix image
    put_code(0); put_code(1); put_code(2); put_code(4); put_single();   // green, red, blue, alpha (one symbol, no bits, in an opaque picture), distance
    const uint64_t npx = uint64_t(im.width) * im.height;
    for (uint64_t i0 = 0; i0 < npx; i0 += 64) {
        LV<uint64_t> val; LV<uint32_t> nb;
        LFOR(l) {
            const uint64_t i = i0 + uint32_t(l);
            const uint32_t v = i < npx ? work[im.res_off + i] : 0u;
            const uint32_t g = (v >> 8) & 255u, r = (v >> 16) & 255u, b = v & 255u, a = v >> 24;
            const uint32_t lg = S.len[0][g], lr = S.len[1][r], lb = S.len[2][b], la = S.len[4][a];
            nb[l] = i < npx ? lg + lr + lb + la : 0u;
            val[l] = uint64_t(S.code[0][g]) | (uint64_t(S.code[1][r]) << lg) | (uint64_t(S.code[2][b]) << (lg + lr)) | (uint64_t(S.code[4][a]) << (lg + lr + lb));
        }
        bo.put(val, nb);
    }
    const uint64_t payload = (bo.bitpos + 7) >> 3;
    bo.finish();
    CSP_WAVE_SYNC();
    const uint64_t padded = payload + (payload & 1u), total = 20 + padded;
    LFOR(l) if (l == 0) {
        if (total > im.out_cap) { status[image] = 1; file_len[image] = 0; }
        else {
            if (payload & 1u) file[20 + payload] = 0;
            const uint8_t hd[20] = {'R', 'I', 'F', 'F', uint8_t(total - 8), uint8_t((total - 8) >> 8), uint8_t((total - 8) >> 16), uint8_t((total - 8) >> 24), 'W', 'E', 'B', 'P',
                                    'V', 'P', '8', 'L', uint8_t(payload), uint8_t(payload >> 8), uint8_t(payload >> 16), uint8_t(payload >> 24)};
            for (int k = 0; k < 20; k++) file[k] = hd[k];
            status[image] = 0; file_len[image] = uint32_t(total);
        }
    }
}

void launch_vp8l_encode(hipStream_t st, const Vp8lImg *imgs, int nimg, uint32_t max_blocks, uint64_t max_pixels, uint32_t *work, uint8_t *modes, uint32_t *hist, uint8_t *out, uint32_t *file_len,
                        uint32_t *status) {
    if (!nimg) return;
    CSH_LAUNCH_PHASED(k_vp8l_residuals, 4, dim3(max_blocks, unsigned(nimg)), dim3(256), st, imgs, work, modes);
    CSH_LAUNCH_PHASED(k_vp8l_hist, 3, dim3(unsigned((max_pixels + 4095) / 4096), unsigned(nimg)), dim3(256), st, imgs, work, hist);
    CSH_LAUNCH(k_vp8l_pack, dim3(unsigned(nimg)), dim3(CSP_WAVE_THREADS), st, imgs, nimg, work, modes, hist, out, file_len, status);
}

}  // namespace csw
