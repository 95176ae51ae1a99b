// k_webp_dec.hip -- WebP inputs: one VP8 key frame (lossy) or VP8L stream (lossless, vp8l_dec.h) per workgroup (the serial part on lane 0, the rest across the lanes; the pictures of a batch are the other parallel axis),
// vp8_dec.h holds the decoder.  Replaces libwebp's decoder on libcaesium's WebP input paths (reference call sites
// /root/reference/src/compressor.rs:289-305; file type sniffed as /root/reference/src/compressor.rs:589-598 does).
#include "webp_kernels.h"
#include "vp8_dec.h"
#include "vp8l_dec.h"

namespace csw {

// One workgroup per picture, in phases: (0) lane 0 walks the serial part -- a lossless picture altogether (vp8l_dec.h), a lossy frame's parse
// (vp8_parse_frame: modes and coefficients into per-macroblock records) --; (1 .. nsteps) the reconstruction and (nsteps + 1 .. 2 nsteps) the loop filter,
// each as a wave front over the macroblock rows: row r at column t - 2 r in step t (a macroblock needs its left, upper and upper-right neighbours); then
// the RGB conversion by row pairs across the lanes; then a lossy file's alpha plane (lane 0: a VP8L stream of its own) and its join with the colour (all
// lanes).  nsteps = the largest mbw + 2 mbh of the batch's lossy frames.  The reconstruction predicts in a scratch patch per lane: 64 of them (rows 64
// apart are in flight together only in frames wider than 2048 samples; a lane then takes its rows one after the other), in the LDS the parse has left.
#define CSW_RECON_LANES 64
union Vp8Lds { Vp8Hot hot; Vp8Scratch scratch[CSW_RECON_LANES]; };
__global__ void __launch_bounds__(256) k_vp8_decode(const uint8_t *pool, Vp8In *imgs, int n, uint8_t *work, uint8_t *rgb, int nsteps) {
    CSH_SHARED Vp8Lds lds;
    CSH_SHARED uint32_t s_translucent;
    const int i = int(blockIdx.x);
    Vp8In &im = imgs[i];
    const bool room = im.rgba_off != ~0ull;
    uint8_t *rgba = room ? rgb + im.rgba_off : nullptr, *aplane = room ? rgb + im.a_off : nullptr;
    uint8_t *wk = work + im.work_off, *out = rgb + im.rgb_off;
    const uint32_t W = im.width, H = im.height;
    CSH_PHASE_LOOP(2 * nsteps + 5) {
        if (phase == 0) {
            if (threadIdx.x == 0) {
                uint32_t has_alpha = 0;
                s_translucent = 0;
                if (im.lossless) im.status = uint32_t(vp8l_decode_frame(pool + im.data_off, im.data_len, W, H, wk, out, im.data_len, false, rgba, aplane, &has_alpha));
                else im.status = uint32_t(vp8_parse_frame(pool + im.data_off, im.data_len, W, H, wk, lds.hot, im.debug));
                im.has_alpha = has_alpha;
            }
            continue;
        }
        if (im.lossless || im.status) continue;
        if (phase <= nsteps) {
            if ((im.debug & 1u) || threadIdx.x >= CSW_RECON_LANES) continue;
            const int t = phase - 1;
            for (uint32_t r = threadIdx.x; r < im.mbh; r += CSW_RECON_LANES) {
                const int mx = t - 2 * int(r);
                if (mx >= 0 && mx < int(im.mbw)) vp8_recon_mb(wk, W, H, uint32_t(mx), r, lds.scratch[threadIdx.x]);
            }
            continue;
        }
        if (phase <= 2 * nsteps) {
            if (im.debug & 3u) continue;
            const int t = phase - nsteps - 1;
            for (uint32_t r = threadIdx.x; r < im.mbh; r += blockDim.x) {
                const int mx = t - 2 * int(r);
                if (mx >= 0 && mx < int(im.mbw)) vp8_filter_mb(wk, W, H, uint32_t(mx), r);
            }
            continue;
        }
        if (phase == 2 * nsteps + 1) {
            if (im.debug & 5u) continue;
            for (uint32_t k = threadIdx.x; k <= (H + 1) >> 1; k += blockDim.x) vp8_rgb_rows(wk, W, H, k, out);
            continue;
        }
        if (!im.alph_len) continue;
        if (phase == 2 * nsteps + 2) {   // the alpha plane of a lossy file (the frame's work area is free again)
            if (threadIdx.x == 0) im.status = !room ? 3u : uint32_t(alph_decode(pool + im.alph_off, im.alph_len, W, H, wk, aplane));
            continue;
        }
        if (phase == 2 * nsteps + 3) {
            const uint64_t npx = uint64_t(W) * H;
            bool translucent = false;
            for (uint64_t k = threadIdx.x; k < npx; k += blockDim.x) {
                rgba[4 * k] = out[3 * k]; rgba[4 * k + 1] = out[3 * k + 1]; rgba[4 * k + 2] = out[3 * k + 2]; rgba[4 * k + 3] = aplane[k];
                translucent |= aplane[k] < 255;
            }
            if (translucent) s_translucent = 1;   // (several lanes may store the same 1)
            continue;
        }
        if (threadIdx.x == 0) im.has_alpha = s_translucent;
    }
}
void launch_vp8_decode(hipStream_t st, const uint8_t *pool, Vp8In *imgs, int n, uint8_t *work, uint8_t *rgb, int nsteps) {
    if (n) CSH_LAUNCH_PHASED(k_vp8_decode, 2 * nsteps + 5, dim3(unsigned(n)), dim3(256), st, pool, imgs, n, work, rgb, nsteps);
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
