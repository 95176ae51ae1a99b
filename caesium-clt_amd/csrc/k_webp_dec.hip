// k_webp_dec.hip -- WebP inputs: one VP8 key frame (lossy) or VP8L stream (lossless, vp8l_dec.h) per workgroup (the serial part on lane 0, the rest across the lanes; the pictures of a batch are the other parallel axis),
// vp8_dec.h holds the decoder.  Replaces libwebp's decoder on libcaesium's WebP input paths (reference call sites
// /root/reference/src/compressor.rs:289-305; file type sniffed as /root/reference/src/compressor.rs:589-598 does).
#include "webp_kernels.h"
#include "vp8_dec.h"
#include "vp8l_dec.h"

namespace csw {

// The pictures of a batch are one parallel axis (a workgroup each); inside a picture:
//   k_webp_parse    lane 0 walks the serial part, its working set in LDS.  mode 0: a lossy frame's parse (vp8_parse_frame: modes and coefficients into
//                   per-macroblock records), a lossless picture's entropy layer (vp8l_entropy: the transformed ARGB frame); mode 1: the entropy layer of a
//                   lossy file's alpha plane (a VP8L stream of its own, in the frame's work area, which is free again by then)
//   k_vp8_pixels    lossy frames: (1 .. nsteps) the reconstruction and (nsteps + 1 .. 2 nsteps) the loop filter, each as a wave front over the macroblock
//                   rows: row r at column t - 2 r in step t (a macroblock needs its left, upper and upper-right neighbours); then the RGB conversion by row
//                   pairs across the lanes.  nsteps = the largest mbw + 2 mbh.  The reconstruction predicts in a scratch patch per lane: 64 of them (rows
//                   64 apart are in flight together only in frames wider than 2048 samples; a lane then takes its rows one after the other)
//   k_vp8l_pixels   VP8L frames: the inverse transforms (vp8l_transform_step: a pass each, the predictor a wave front over the rows), then mode 0 the
//                   RGB / RGBA / alpha-plane output, mode 1 the alpha plane out of the green channel and its unfilter (another wave front)
//   k_vp8_alpha_join  a lossy file's colour and alpha plane joined, across the lanes
// (separate kernels: as one, the compiler had 900 registers to spill -- and got it wrong)
#define CSW_RECON_LANES 64
union ParseLds { Vp8Hot hot; LHot lossless; };
__global__ void __launch_bounds__(64) k_webp_parse(const uint8_t *pool, Vp8In *imgs, uint8_t *work, int mode) {
    CSH_SHARED ParseLds lds;
#ifdef CSH_EMUL
    if (threadIdx.x != 0) return;
#endif
    Vp8In &im = imgs[blockIdx.x];
    uint8_t *wk = work + im.work_off;
    if (mode == 0 && !im.lossless) {   // the whole wave walks the frame in step (vp8_dec.h, CSW_U): every lane the same values and the same stores
        im.has_alpha = 0;
        im.status = uint32_t(vp8_parse_frame(pool + im.data_off, im.data_len, im.width, im.height, wk, lds.hot, im.debug));
        return;
    }
    if (threadIdx.x != 0) return;
    if (mode == 0) {
        im.has_alpha = 0;
        im.status = uint32_t(vp8l_entropy(pool + im.data_off, im.data_len, im.width, im.height, wk, im.data_len, false, &lds.lossless));
        return;
    }
    if (im.lossless || im.status || !im.alph_len) return;
    im.status = im.rgba_off == ~0ull ? 3u : uint32_t(alph_entropy(pool + im.alph_off, im.alph_len, im.width, im.height, wk, &lds.lossless));
}
__global__ void __launch_bounds__(256) k_vp8_pixels(const Vp8In *imgs, uint8_t *work, uint8_t *rgb, int nsteps) {
    CSH_SHARED Vp8Scratch scratch[CSW_RECON_LANES];
    const Vp8In &im = imgs[blockIdx.x];
    uint8_t *wk = work + im.work_off, *out = rgb + im.rgb_off;
    const uint32_t W = im.width, H = im.height;
    const bool skip = im.lossless || im.status;
    CSH_PHASE_LOOP(2 * nsteps + 1) {
        if (skip) continue;
        // the rows of a step are dealt over the four waves (row r to wave r & 3): the macroblocks of a step differ in their modes, and a wave walks every
        // path its lanes take -- sixteen rows per wave diverge a quarter as much as sixty-four
        const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
        if (phase < nsteps) {
            if ((im.debug & 1u) || ln >= CSW_RECON_LANES / 4) continue;
            const uint32_t slot = ln * 4 + wv;
            for (uint32_t r = slot; r < im.mbh; r += CSW_RECON_LANES) {
                const int mx = phase - 2 * int(r);
                if (mx >= 0 && mx < int(im.mbw)) vp8_recon_mb(wk, W, H, uint32_t(mx), r, scratch[slot]);
            }
            continue;
        }
        if (phase < 2 * nsteps) {
            if (im.debug & 3u) continue;
            for (uint32_t r = ln * 4 + wv; r < im.mbh; r += blockDim.x) {
                const int mx = phase - nsteps - 2 * int(r);
                if (mx >= 0 && mx < int(im.mbw)) vp8_filter_mb(wk, W, H, uint32_t(mx), r);
            }
            continue;
        }
        if (im.debug & 5u) continue;
        for (uint32_t k = threadIdx.x; k <= (H + 1) >> 1; k += blockDim.x) vp8_rgb_rows(wk, W, H, k, out);
    }
}
// phases: the undo slots one after the other (the predictor's takes psteps phases, the others one each: at most psteps + 3), then two of output (mode 0) or
// 1 + psteps of the alpha plane (mode 1).  psteps = the batch's largest vp8l_pred_steps.
__global__ void __launch_bounds__(256) k_vp8l_pixels(const uint8_t *pool, Vp8In *imgs, uint8_t *work, uint8_t *rgb, int psteps, int mode) {
    CSH_SHARED uint32_t s_amin;
    Vp8In &im = imgs[blockIdx.x];
    const bool mine = mode == 0 ? im.lossless != 0 : (!im.lossless && im.alph_len != 0);
    const uint32_t W = im.width, H = im.height;
    const LWork lw = lwork(work + im.work_off, W, H);
    const int tail = psteps + 3;
    CSH_PHASE_LOOP(tail + (mode == 0 ? 2 : 1 + psteps)) {
        if (!mine || im.status) continue;
        if (im.debug & 8u) continue;   // (timing probe: the entropy layer alone)
        const LFrame &f = *lw.info;
        if (phase < tail) {
            // which undo slot, which of its steps
            int start = 0, slot = -1, step = 0;
            for (int j = 0; j < int(f.ntr); j++) {
                const int len = f.tr[f.ntr - 1 - uint32_t(j)].type == 0 ? psteps : 1;
                if (phase >= start && phase < start + len) { slot = j; step = phase - start; }
                start += len;
            }
            if (slot >= 0) vp8l_transform_step(lw, H, slot, uint32_t(step), threadIdx.x, blockDim.x);
            if (phase == 0 && threadIdx.x == 0) s_amin = 255;
            continue;
        }
        if (mode == 1) {
            uint8_t *aplane = rgb + im.a_off;
            if (phase == tail) alph_plane_step(lw, pool + im.alph_off, aplane, threadIdx.x, blockDim.x);
            else alph_unfilter_step(lw, aplane, W, H, uint32_t(phase - tail - 1), threadIdx.x, blockDim.x);
            continue;
        }
        if (im.debug & 16u) continue;   // (timing probe: no output)
        const uint32_t *cur = vp8l_result(lw);
        uint8_t *out = rgb + im.rgb_off;
        if (phase == tail) {   // ARGB -> RGB for the three-channel encoders; is the picture opaque?
            uint32_t amin = 255;
            for (uint64_t i = threadIdx.x; i < lw.npx; i += blockDim.x) {
                const uint32_t v = cur[i];
                out[3 * i] = uint8_t(v >> 16); out[3 * i + 1] = uint8_t(v >> 8); out[3 * i + 2] = uint8_t(v);
                if ((v >> 24) < amin) amin = v >> 24;
            }
            if (amin < 255) atomicMin(&s_amin, amin);
            continue;
        }
        if (s_amin == 255) continue;   // a picture that is not opaque leaves RGBA and its alpha plane as well
        if (im.rgba_off == ~0ull) { if (threadIdx.x == 0) im.status = 3; continue; }
        uint8_t *rgba = rgb + im.rgba_off, *aplane = rgb + im.a_off;
        for (uint64_t i = threadIdx.x; i < lw.npx; i += blockDim.x) {
            const uint32_t v = cur[i];
            rgba[4 * i] = uint8_t(v >> 16); rgba[4 * i + 1] = uint8_t(v >> 8); rgba[4 * i + 2] = uint8_t(v); rgba[4 * i + 3] = uint8_t(v >> 24);
            aplane[i] = uint8_t(v >> 24);
        }
        if (threadIdx.x == 0) im.has_alpha = 1;
    }
}
__global__ void __launch_bounds__(256) k_vp8_alpha_join(Vp8In *imgs, uint8_t *rgb) {
    CSH_SHARED uint32_t s_translucent;
    Vp8In &im = imgs[blockIdx.x];
    const bool skip = im.lossless || im.status || !im.alph_len;
    CSH_PHASE_LOOP(3) {
        if (skip) continue;
        if (phase == 0) { if (threadIdx.x == 0) s_translucent = 0; continue; }
        if (phase == 1) {
            const uint8_t *out = rgb + im.rgb_off, *aplane = rgb + im.a_off;
            uint8_t *rgba = rgb + im.rgba_off;
            const uint64_t npx = uint64_t(im.width) * im.height;
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
void launch_vp8_decode(hipStream_t st, const uint8_t *pool, Vp8In *imgs, int n, uint8_t *work, uint8_t *rgb, int nsteps, int psteps_lossless, int psteps_alpha) {
    if (!n) return;
    CSH_LAUNCH(k_webp_parse, dim3(unsigned(n)), dim3(64), st, pool, imgs, work, 0);
    if (nsteps) CSH_LAUNCH_PHASED(k_vp8_pixels, 2 * nsteps + 1, dim3(unsigned(n)), dim3(256), st, imgs, work, rgb, nsteps);
    if (psteps_lossless) CSH_LAUNCH_PHASED(k_vp8l_pixels, psteps_lossless + 3 + 2, dim3(unsigned(n)), dim3(256), st, pool, imgs, work, rgb, psteps_lossless, 0);
    if (psteps_alpha) {
        CSH_LAUNCH(k_webp_parse, dim3(unsigned(n)), dim3(64), st, pool, imgs, work, 1);
        CSH_LAUNCH_PHASED(k_vp8l_pixels, psteps_alpha + 3 + 1 + psteps_alpha, dim3(unsigned(n)), dim3(256), st, pool, imgs, work, rgb, psteps_alpha, 1);
        CSH_LAUNCH_PHASED(k_vp8_alpha_join, 3, dim3(unsigned(n)), dim3(256), st, imgs, rgb);
    }
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
