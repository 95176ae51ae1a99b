// k_png_resize.hip -- the resize of a PNG source (width / height set: /root/reference/src/compressor.rs:503-536; engine: image 0.25.9
// `resize_exact(.., Lanczos3)` over the decoded image).  The same two passes as k_resize.hip (vertical to an f32
// image, horizontal back to u8; weights from the host; __fmul_rn / __fadd_rn in image-rs's left-to-right order, so the result is
// bit-identical to the oracle's cso_lanczos3_resize / _resize16) over interleaved samples, 1 to 4 per pixel, 8 bits or 16 (big-endian in
// memory, as a PNG stores them; image-rs keeps 16-bit images at 16 bits).  One lane per output sample.
#include "png_kernels.h"

namespace csp {

__global__ void __launch_bounds__(256) k_png_lanczos_v(const PngResize *jobs, const csh::ResizeTap *taps, const float *weights, const uint8_t *src, float *tmp) {
    const PngResize j = jobs[blockIdx.y];
    const size_t rowlen = size_t(j.width) * j.nc;
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(j.nh) * rowlen) return;
    const size_t oy = i / rowlen, xc = i - oy * rowlen;
    const csh::ResizeTap t = taps[j.vtap_base + oy];
    const float *ws = weights + t.woff;
    float acc = 0.0f;
    if (j.bps == 2) {
        const uint8_t *s = src + j.src_off + (size_t(t.left) * rowlen + xc) * 2;
        for (int k = 0; k < t.n; k++) { const uint8_t *q = s + size_t(k) * rowlen * 2; acc = __fadd_rn(acc, __fmul_rn(float((uint32_t(q[0]) << 8) | q[1]), ws[k])); }
    } else {
        const uint8_t *s = src + j.src_off + size_t(t.left) * rowlen + xc;
        for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(float(s[size_t(k) * rowlen]), ws[k]));
    }
    tmp[j.tmp_off + i] = acc;
}

__global__ void __launch_bounds__(256) k_png_lanczos_h(const PngResize *jobs, const csh::ResizeTap *taps, const float *weights, const float *tmp, uint8_t *dst) {
    const PngResize j = jobs[blockIdx.y];
    const size_t nc = j.nc, rowlen_in = size_t(j.width) * nc, rowlen_out = size_t(j.nw) * nc;
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(j.nh) * rowlen_out) return;
    const size_t y = i / rowlen_out, r = i - y * rowlen_out, ox = r / nc, c = r - ox * nc;
    const csh::ResizeTap t = taps[j.htap_base + ox];
    const float *ws = weights + t.woff;
    const float *s = tmp + j.tmp_off + y * rowlen_in + size_t(t.left) * nc + c;
    float acc = 0.0f;
    for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(s[size_t(k) * nc], ws[k]));
    const float top = j.bps == 2 ? 65535.0f : 255.0f;
    acc = acc < 0.0f ? 0.0f : (acc > top ? top : acc);
    int q = int(acc);                                     // round half away from zero (acc >= 0)
    q += (acc - float(q) >= 0.5f) ? 1 : 0;
    if (j.bps == 2) { dst[j.dst_off + 2 * i] = uint8_t(q >> 8); dst[j.dst_off + 2 * i + 1] = uint8_t(q); }
    else dst[j.dst_off + i] = uint8_t(q);
}

void launch_png_resize(hipStream_t st, const PngResize *jobs, int njobs, const csh::ResizeTap *taps, const float *weights, const uint8_t *src, float *tmp, uint8_t *dst,
                       uint64_t max_tmp, uint64_t max_dst) {
    if (!njobs) return;
    CSH_LAUNCH(k_png_lanczos_v, dim3(unsigned((max_tmp + 255) / 256), njobs), dim3(256), st, jobs, taps, weights, src, tmp);
    CSH_LAUNCH(k_png_lanczos_h, dim3(unsigned((max_dst + 255) / 256), njobs), dim3(256), st, jobs, taps, weights, tmp, dst);
}

}  // namespace csp
