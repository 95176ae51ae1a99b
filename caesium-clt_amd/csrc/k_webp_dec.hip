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
    if (im.lossless) im.status = uint32_t(vp8l_decode_frame(pool + im.data_off, im.data_len, im.width, im.height, work + im.work_off, rgb + im.rgb_off, im.data_len));
    else im.status = uint32_t(vp8_decode_frame(pool + im.data_off, im.data_len, im.width, im.height, work + im.work_off, rgb + im.rgb_off));
}
void launch_vp8_decode(hipStream_t st, const uint8_t *pool, Vp8In *imgs, int n, uint8_t *work, uint8_t *rgb) {
    if (n) CSH_LAUNCH(k_vp8_decode, dim3(unsigned(n)), dim3(64), st, pool, imgs, n, work, rgb);
}

}  // namespace csw
