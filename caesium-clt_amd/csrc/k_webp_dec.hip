// k_webp_dec.hip -- WebP inputs: one VP8 key frame (lossy) or VP8L stream (lossless, vp8l_dec.h) per wave (lane 0 walks it; the pictures of a batch are the parallel axis),
// vp8_dec.h holds the decoder.  Replaces libwebp's decoder on libcaesium's WebP input paths (reference call sites
// /root/reference/src/compressor.rs:289-305; file type sniffed as /root/reference/src/compressor.rs:589-598 does).
#include "webp_kernels.h"
#include "vp8_dec.h"
#include "vp8l_dec.h"

namespace csw {

__global__ void __launch_bounds__(64) k_vp8_decode(const uint8_t *pool, Vp8In *imgs, int n, uint8_t *work, uint8_t *rgb) {
    const int i = int(blockIdx.x);
    if (i >= n || threadIdx.x != 0) return;
    Vp8In &im = imgs[i];
    const bool room = im.rgba_off != ~0ull;
    uint8_t *rgba = room ? rgb + im.rgba_off : nullptr, *aplane = room ? rgb + im.a_off : nullptr;
    uint32_t has_alpha = 0;
    if (im.lossless) im.status = uint32_t(vp8l_decode_frame(pool + im.data_off, im.data_len, im.width, im.height, work + im.work_off, rgb + im.rgb_off, im.data_len, false, rgba, aplane, &has_alpha));
    else {
        im.status = uint32_t(vp8_decode_frame(pool + im.data_off, im.data_len, im.width, im.height, work + im.work_off, rgb + im.rgb_off));
        if (!im.status && im.alph_len) {   // the alpha plane of a lossy file (the frame's work area is free again)
            if (!room) im.status = 3;
            else {
                im.status = uint32_t(alph_decode(pool + im.alph_off, im.alph_len, im.width, im.height, work + im.work_off, aplane));
                if (!im.status) {
                    const uint64_t npx = uint64_t(im.width) * im.height;
                    const uint8_t *c = rgb + im.rgb_off;
                    uint32_t amin = 255;
                    for (uint64_t k = 0; k < npx; k++) { rgba[4 * k] = c[3 * k]; rgba[4 * k + 1] = c[3 * k + 1]; rgba[4 * k + 2] = c[3 * k + 2]; rgba[4 * k + 3] = aplane[k]; if (aplane[k] < amin) amin = aplane[k]; }
                    has_alpha = amin < 255 ? 1u : 0u;
                }
            }
        }
    }
    im.has_alpha = has_alpha;
}
void launch_vp8_decode(hipStream_t st, const uint8_t *pool, Vp8In *imgs, int n, uint8_t *work, uint8_t *rgb) {
    if (n) CSH_LAUNCH(k_vp8_decode, dim3(unsigned(n)), dim3(64), st, pool, imgs, n, work, rgb);
}

// RGB + alpha plane -> interleaved RGBA (the resized halves of a picture with transparency, joined for the PNG / lossless WebP coders): four pixels per lane
__global__ void __launch_bounds__(256) k_rgba_join(const uint8_t *rgb, const uint8_t *alpha, uint8_t *rgba, uint64_t npx) {
    const uint64_t p0 = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
    for (uint64_t p = p0; p < npx && p < p0 + 4; p++) {
        rgba[4 * p] = rgb[3 * p]; rgba[4 * p + 1] = rgb[3 * p + 1]; rgba[4 * p + 2] = rgb[3 * p + 2]; rgba[4 * p + 3] = alpha[p];
    }
}
void launch_rgba_join(hipStream_t st, const uint8_t *rgb, const uint8_t *alpha, uint8_t *rgba, uint64_t npx) {
    if (npx) CSH_LAUNCH(k_rgba_join, dim3(unsigned((npx + 1023) / 1024)), dim3(256), st, rgb, alpha, rgba, npx);
}

}  // namespace csw
