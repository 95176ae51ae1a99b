// webp_kernels.h -- the lossy WebP row on the device: descriptors and launchers (k_webp.hip: import and coder back end; k_vp8enc.hip:
// libwebp's encoder).  Statement: oracle/vp8enc_oracle.c (pinned to libwebp's WebPEncode byte for byte) and oracle/webp_oracle.c (the import).
#pragma once
#include <vector>

#include "gpu_rt.h"

namespace csh { template <class T> struct DevBuf; }

namespace csw {

enum { WEBP_MB_REC = 432 };   // int16 per macroblock in the level pool: 25 blocks x 16 levels + the info block (k_webp.hip)

struct WebpImg {
    uint32_t width, height, mbw, mbh, ncomp;   // ncomp: samples per input pixel: 3 = interleaved RGB, 1 = grey, 4 / 2 = the same with an alpha sample behind (skipped here)
    int32_t quality;                           // libwebp's quality 0..100
    uint32_t cls, qtab;                        // set by launch_webp_encode: the picture's size class (its plan of steps) and its quality table
    uint64_t rgb_off;                          // input pixels in the RGB pool
    uint64_t y_off, u_off, v_off;              // source planes, padded to whole macroblocks (work pool)
    uint64_t ry_off, ru_off, rv_off;           // the encoder's reconstruction (what a decoder will see)
    uint64_t lev_off;                          // quantised levels: WEBP_MB_REC int16 per macroblock (int16 index)
    uint64_t out_off;                          // output file region
    uint32_t out_cap;
    uint32_t image;                            // index of the image in the batch's status / size arrays
};

void launch_webp_yuv(hipStream_t st, const WebpImg *imgs, int nimg, uint32_t max_luma, const uint8_t *rgb, uint8_t *work);
// the encoder behind the import: analysis, segments, the macroblock loop step by step, statistics, then the two partitions and the file.  himgs: the host's
// pictures (cls / qtab are filled in here and the array copied over d_imgs, which the caller uploaded for launch_webp_yuv);
// scratch: a second region laid out like the output pool; part_size: nimg x 2.  `mid` (optional) is recorded between the macroblock loop and the coder.
// Returns with the stream idle; != 0 on an allocation / launch failure.
int launch_webp_encode(hipStream_t st, WebpImg *himgs, int nimg, WebpImg *d_imgs, uint8_t *work, int16_t *levels, uint8_t *scratch, uint32_t *part_size, uint8_t *out,
                       uint32_t *img_size, uint32_t *status, hipEvent_t mid);
struct Vp8FrameDev;
int launch_webp_backend(hipStream_t st, const WebpImg *imgs, const WebpImg *himgs, int nimg, const int16_t *levels, const Vp8FrameDev *frames, const std::vector<uint64_t> &base,
                        const uint64_t *d_base, csh::DevBuf<uint32_t> &d_cnt, const uint16_t *d_blk, uint8_t *scratch, uint32_t *part_size, uint8_t *out, uint32_t *img_size, uint32_t *status);

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
