// webp_kernels.h -- the lossy WebP row on the device: descriptors and launchers (k_webp.hip).
// Statement: oracle/webp_oracle.c (a minimal conformant VP8 key-frame encoder; see its header for what is and is not pinned).
#pragma once
#include "gpu_rt.h"

namespace csw {

enum { WEBP_MB_REC = 432 };   // int16 per macroblock in the level pool: 25 blocks x 16 levels + the info block (k_webp.hip)

struct WebpImg {
    uint32_t width, height, mbw, mbh, ncomp;   // ncomp: samples per input pixel: 3 = interleaved RGB, 1 = grey, 4 / 2 = the same with an alpha sample behind (skipped here)
    int32_t qi;                                // quantiser index 0..127
    uint64_t rgb_off;                          // input pixels in the RGB pool
    uint64_t y_off, u_off, v_off;              // source planes, padded to whole macroblocks (work pool)
    uint64_t ry_off, ru_off, rv_off;           // the encoder's reconstruction (what a decoder will see)
    uint64_t lev_off;                          // quantised levels: WEBP_MB_REC int16 per macroblock (int16 index)
    uint64_t out_off;                          // output file region
    uint32_t out_cap;
    uint32_t image;                            // index of the image in the batch's status / size arrays
};

void launch_webp_yuv(hipStream_t st, const WebpImg *imgs, int nimg, uint32_t max_luma, const uint8_t *rgb, uint8_t *work);
void launch_webp_mb(hipStream_t st, const WebpImg *imgs, int nimg, uint32_t max_mbw, uint32_t max_mbh, uint8_t *work, int16_t *levels);   // one launch per skewed diagonal of macroblocks
// stats: nimg x 1056 x 2 counters (zeroed by the caller); probs / update: nimg x 1056 bytes; scratch: a second region laid out like the
// output pool (every partition is coded into its own slice of it); part_size: nimg x 9
// himgs: the host's copy of imgs (the launcher lays the token partitions' decision streams out from the pictures' sizes).  Returns with the stream idle.
void launch_webp_code(hipStream_t st, const WebpImg *imgs, const WebpImg *himgs, int nimg, uint32_t max_mbh, const int16_t *levels, uint32_t *stats, uint8_t *probs, uint8_t *update,
                      uint8_t *scratch, uint32_t *part_size, uint8_t *out, uint32_t *img_size, uint32_t *status);

enum { VP8L_ALPHA_OF = 16 };   // Vp8lImg::channels = VP8L_ALPHA_OF + 2 / + 4: code the alpha sample of a grey + alpha / RGBA picture as a grey picture
// lossless WebP output (k_vp8l_enc.hip): one picture of 8-bit grey (channels 1), grey + alpha (2), RGB (3) or RGBA (4) pixels in device memory
struct Vp8lImg {
    const uint8_t *rgb;
    uint32_t width, height, channels, bw, bh;   // bw x bh blocks of 16 x 16 pixels
    uint64_t res_off;      // residual ARGB, one u32 per pixel (u32 index into the work pool)
    uint64_t mode_off;     // predictor mode of every block (byte index)
    uint64_t out_off;      // output file region
    uint32_t out_cap;
};
void launch_vp8l_encode(hipStream_t st, const Vp8lImg *imgs, int nimg, uint32_t max_blocks, uint64_t max_pixels, uint32_t *work, uint8_t *modes, uint32_t *hist, uint8_t *out, uint32_t *file_len,
                        uint32_t *status);

struct Vp8In;
// lossy WebP inputs (k_webp_dec.hip): every image's VP8 key frame -> RGB in the pixel pool; imgs[i].status = 0 or an error
void launch_vp8_decode(hipStream_t st, const uint8_t *pool, Vp8In *imgs, int n, uint8_t *work, uint8_t *rgb, int nsteps, int psteps_lossless, int psteps_alpha);
void launch_rgba_join(hipStream_t st, const uint8_t *rgb, const uint8_t *alpha, uint8_t *rgba, uint64_t npx);

}  // namespace csw
