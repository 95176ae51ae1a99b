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

// Both passes in one kernel, the f32 row between them in LDS (as k_resize.hip's k_lanczos_fused, round 4): a workgroup is one output row -- phase 0 the
// vertical pass of that row over every source column, phase 1 the horizontal pass out of LDS.  The two kernels' own arithmetic in their order: their bytes.
#define CSP_RZ_CAP 16128   // floats of LDS: 4032 RGBA pixels a row; wider pictures keep the two kernels
__global__ void __launch_bounds__(256) k_png_lanczos_fused(const PngResize *jobs, const csh::ResizeTap *taps, const float *weights, const uint8_t *src, uint8_t *dst) {
    CSH_SHARED float s_row[CSP_RZ_CAP];
    const PngResize j = jobs[blockIdx.y];
    const uint32_t nc = j.nc, rowlen = j.width * nc, rowlen_out = j.nw * nc, oy = blockIdx.x;
    CSH_PHASE_LOOP(2) {
        if (oy >= j.nh || rowlen > uint32_t(CSP_RZ_CAP)) continue;
        if (phase == 0) {
            const csh::ResizeTap t = taps[j.vtap_base + oy];
            const float *ws = weights + t.woff;
            for (uint32_t xc = threadIdx.x; xc < rowlen; xc += blockDim.x) {
                float acc = 0.0f;
                if (j.bps == 2) {
                    const uint8_t *s = src + j.src_off + (size_t(t.left) * rowlen + xc) * 2;
                    for (int k = 0; k < t.n; k++) { const uint8_t *q = s + size_t(k) * rowlen * 2; acc = __fadd_rn(acc, __fmul_rn(float((uint32_t(q[0]) << 8) | q[1]), ws[k])); }
                } else {
                    const uint8_t *s = src + j.src_off + size_t(t.left) * rowlen + xc;
                    for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(float(s[size_t(k) * rowlen]), ws[k]));
                }
                s_row[xc] = acc;
            }
            continue;
        }
        const float top = j.bps == 2 ? 65535.0f : 255.0f;
        for (uint32_t r = threadIdx.x; r < rowlen_out; r += blockDim.x) {
            const uint32_t ox = r / nc, c = r - ox * nc;
            const csh::ResizeTap t = taps[j.htap_base + ox];
            const float *ws = weights + t.woff;
            const float *s = s_row + size_t(t.left) * nc + c;
            float acc = 0.0f;
            for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(s[size_t(k) * nc], ws[k]));
            acc = acc < 0.0f ? 0.0f : (acc > top ? top : acc);
            int q = int(acc);                                     // round half away from zero (acc >= 0)
            q += (acc - float(q) >= 0.5f) ? 1 : 0;
            const size_t i = size_t(oy) * rowlen_out + r;
            if (j.bps == 2) { dst[j.dst_off + 2 * i] = uint8_t(q >> 8); dst[j.dst_off + 2 * i + 1] = uint8_t(q); }
            else dst[j.dst_off + i] = uint8_t(q);
        }
    }
}
bool png_resize_is_fused(const PngResize *hjobs, int njobs) {
    if (getenv("CSH_RESIZE_TWO_PASS")) return false;
    for (int k = 0; k < njobs; k++) if (uint64_t(hjobs[k].width) * hjobs[k].nc > uint64_t(CSP_RZ_CAP)) return false;
    return true;
}
void launch_png_resize(hipStream_t st, const PngResize *jobs, const PngResize *hjobs, int njobs, const csh::ResizeTap *taps, const float *weights, const uint8_t *src, float *tmp, uint8_t *dst,
                       uint64_t max_tmp, uint64_t max_dst) {
    if (!njobs) return;
    if (png_resize_is_fused(hjobs, njobs)) {
        uint32_t max_nh = 0;
        for (int k = 0; k < njobs; k++) max_nh = hjobs[k].nh > max_nh ? hjobs[k].nh : max_nh;
        CSH_LAUNCH_PHASED(k_png_lanczos_fused, 2, dim3(max_nh, unsigned(njobs)), dim3(256), st, jobs, taps, weights, src, dst);
        return;
    }
    CSH_LAUNCH(k_png_lanczos_v, dim3(unsigned((max_tmp + 255) / 256), njobs), dim3(256), st, jobs, taps, weights, src, tmp);
    CSH_LAUNCH(k_png_lanczos_h, dim3(unsigned((max_dst + 255) / 256), njobs), dim3(256), st, jobs, taps, weights, tmp, dst);
}

}  // namespace csp
