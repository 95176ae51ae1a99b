/*
 * webp_oracle.c -- CPU ORACLE of the lossy WebP row's import: RGB -> YUV 4:2:0 as libwebp does it.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, tools/): nothing under caesium-clt_amd/ links or calls this.
 *
 * The reference reaches this path through `caesium::convert_in_memory(.., SupportedFileTypes::WebP)` and the WebP recompression
 * (/root/reference/src/compressor.rs:289, :300, :417, :429), i.e. libwebp (libwebp-sys 0.9.5, Cargo.lock:956).  The import is PINNED bit for bit
 * against WebPPictureImportRGB of the libwebp in this container (tests/test_oracle_webp.py); the encoder behind it (W2, W3) is vp8enc_oracle.c,
 * pinned byte for byte against WebPEncode.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vp8_tables.h"
#include "webp_oracle.h"

/* ------------------------------------------------------------------------------------------------ W1: RGB -> YUV 4:2:0
 * libwebp's import (WebPPictureImportRGB, picture_csp_enc.c; PINNED bit-exact against the library itself in tests/test_oracle_webp.py):
 * BT.601 limited range in 16-bit fixed point; chroma from the 2x2 block's mean taken in (gamma 0.80) linear light -- each sample through
 * kVp8GammaToLinear, the sum of four back through kVp8LinearToGamma with linear interpolation, which leaves four times a gamma-domain
 * value for the 18-bit U / V formulas; odd widths / heights repeat the last column / row.  Planes are padded to whole macroblocks by
 * repeating the last sample of each plane (what libwebp's macroblock iterator does on import). */
static int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static int gamma_sum4(int a, int b, int c, int d) {
    const int s = kVp8GammaToLinear[a] + kVp8GammaToLinear[b] + kVp8GammaToLinear[c] + kVp8GammaToLinear[d], pos = s >> 9, x = s & 511;
    return (kVp8LinearToGamma[pos + 1] * x + kVp8LinearToGamma[pos] * (512 - x) + 64) >> 7;
}
void cso_webp_rgb_to_yuv(const uint8_t *rgb, int w, int h, uint8_t *yp, uint8_t *up, uint8_t *vp) {
    const int mbw = (w + 15) >> 4, mbh = (h + 15) >> 4, ys = mbw * 16, cs = mbw * 8, cw = (w + 1) >> 1, ch = (h + 1) >> 1;
    for (int y = 0; y < mbh * 16; y++)
        for (int x = 0; x < ys; x++) {
            const uint8_t *p = rgb + ((size_t)(y < h ? y : h - 1) * w + (x < w ? x : w - 1)) * 3;
            yp[(size_t)y * ys + x] = (uint8_t)((16839 * p[0] + 33059 * p[1] + 6420 * p[2] + (16 << 16) + (1 << 15)) >> 16);
        }
    for (int y = 0; y < mbh * 8; y++)
        for (int x = 0; x < cs; x++) {
            const int cx = x < cw ? x : cw - 1, cy = y < ch ? y : ch - 1;
            const uint8_t *q[4];
            for (int k = 0; k < 4; k++) {
                const int yy = 2 * cy + (k >> 1), xx = 2 * cx + (k & 1);
                q[k] = rgb + ((size_t)(yy < h ? yy : h - 1) * w + (xx < w ? xx : w - 1)) * 3;
            }
            const int r = gamma_sum4(q[0][0], q[1][0], q[2][0], q[3][0]), g = gamma_sum4(q[0][1], q[1][1], q[2][1], q[3][1]), b = gamma_sum4(q[0][2], q[1][2], q[2][2], q[3][2]);
            up[(size_t)y * cs + x] = (uint8_t)clip8((-9719 * r - 19081 * g + 28800 * b + (128 << 18) + (1 << 17)) >> 18);
            vp[(size_t)y * cs + x] = (uint8_t)clip8((28800 * r - 24116 * g - 4684 * b + (128 << 18) + (1 << 17)) >> 18);
        }
}

/* the row's encoder is libwebp's, restated in vp8enc_oracle.c (pinned to WebPEncode); this is the entry point the conversion oracles call */
int cso_vp8enc_encode_rgb(const uint8_t *rgb, int width, int height, float quality, uint8_t **out, size_t *out_len);
int cso_webp_encode_rgb(const uint8_t *rgb, int width, int height, int quality, uint8_t **out, size_t *out_len) {
    return cso_vp8enc_encode_rgb(rgb, width, height, (float)quality, out, out_len);
}
