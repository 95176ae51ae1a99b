// k_resize.hip -- the resize branch of libcaesium's JPEG path (width/height set: reference parameter mapping
// /root/reference/src/compressor.rs:503-536; engine: image 0.25.9 `resize_exact(.., Lanczos3)`, SURVEY.md 8a row R1 + J4).
//   k_planes_to_rgb   decoded component planes -> interleaved RGB (jdsample fancy upsample + jdcolor, both libjpeg integer)
//   k_lanczos_v / _h  image-rs's separable resample: vertical pass to an f32 image, horizontal pass back to u8.
//                     Weights (sinc(x)sinc(x/3), normalised, all f32) are computed on the HOST with the same libm calls the
//                     oracle uses; the kernels only multiply and add -- with __fmul_rn/__fadd_rn so nothing is contracted
//                     into an FMA -- in image-rs's left-to-right order, so the result is bit-identical to the oracle.
//   k_rgb_to_planes   RGB -> full-resolution Y/Cb/Cr planes (jccolor fixed point) for the encoder-side kernels of k_pixel.hip.
// One lane per sample (vertical pass: per output sample column-wise coalesced; horizontal pass: per output sample).
#include "kernels.h"

namespace csh {

__device__ __forceinline__ static int rv(const uint8_t *p, int pitch, int cw, int ch, int y, int x) {
    y = y < 0 ? 0 : (y > ch - 1 ? ch - 1 : y);
    x = x < 0 ? 0 : (x > cw - 1 ? cw - 1 : x);
    return p[size_t(y) * pitch + x];
}
// full-resolution chroma sample: kind 0 full plane, 1 h2v2 fancy, 2 h2v1 fancy (same formulas as k_pixel.hip / jdsample.c)
__device__ static int chroma_at(const uint8_t *p, int pitch, int cw, int ch, int kind, int r, int xx) {
    if (kind == 0) return rv(p, pitch, cw, ch, r, xx);
    int cx = xx >> 1;
    if (kind == 2) {
        if (cw <= 2) return rv(p, pitch, cw, ch, r, cx);
        int nb = (xx & 1) ? cx + 1 : cx - 1;
        return (3 * rv(p, pitch, cw, ch, r, cx) + rv(p, pitch, cw, ch, r, nb) + ((xx & 1) ? 2 : 1)) >> 2;
    }
    int cy = r >> 1;
    if (cw <= 2) return rv(p, pitch, cw, ch, cy, cx);
    int fy = (r & 1) ? cy + 1 : cy - 1, nb = (xx & 1) ? cx + 1 : cx - 1;
    int cs = 3 * rv(p, pitch, cw, ch, cy, cx) + rv(p, pitch, cw, ch, fy, cx);
    int cn = 3 * rv(p, pitch, cw, ch, cy, nb) + rv(p, pitch, cw, ch, fy, nb);
    return (3 * cs + cn + ((xx & 1) ? 7 : 8)) >> 4;
}

__global__ void __launch_bounds__(256) k_planes_to_rgb(const ImgDesc *imgs, const ResizeWork *work, const uint8_t *planes, uint8_t *rgb) {
    const ResizeWork w = work[blockIdx.y];
    if (w.in_kind < 0) return;   // a pixel source: its RGB was copied in when the batch was made
    const ImgDesc &im = imgs[w.image];
    const int W = im.width, H = im.height;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    int y = i / W, x = i - y * W;
    const CompGeom &g0 = im.in[0];
    int Y = planes[im.plane_off[0] + size_t(y) * (g0.real_bw * 8) + x];
    uint8_t *o = rgb + w.rgb_src_off + size_t(i) * im.ncomp;
    if (im.ncomp == 1) { o[0] = uint8_t(Y); return; }
    int cc[2];
    for (int c = 1; c < 3; c++) {
        const CompGeom &g = im.in[c];
        cc[c - 1] = chroma_at(planes + im.plane_off[c], g.real_bw * 8, g.comp_w, g.comp_h, w.in_kind, y, x) - 128;
    }
    // jdcolor.c ycc_rgb_convert, SCALEBITS 16
    int r = Y + ((91881 * cc[1] + 32768) >> 16);
    int b = Y + ((116130 * cc[0] + 32768) >> 16);
    int g = Y + ((-22554 * cc[0] + (-46802 * cc[1] + 32768)) >> 16);
    o[0] = uint8_t(r < 0 ? 0 : r > 255 ? 255 : r);
    o[1] = uint8_t(g < 0 ? 0 : g > 255 ? 255 : g);
    o[2] = uint8_t(b < 0 ? 0 : b > 255 ? 255 : b);
}

// vertical pass: tmp[oy][x][c] = sum_i src[left+i][x][c] * w[i]      (f32, no clamp).  A workgroup is a piece of ONE output row, so the
// row's taps and weights are scalar loads; a lane takes four consecutive samples (one dword load per tap, four accumulators in the
// order image-rs adds them)
__global__ void __launch_bounds__(256) k_lanczos_v(const ImgDesc *imgs, const ResizeWork *work, const ResizeTap *taps, const float *weights,
                                                    const uint8_t *rgb, float *tmp) {
    const ResizeWork w = work[blockIdx.z];
    const ImgDesc &im = imgs[w.image];
    const uint32_t rowlen = uint32_t(im.width) * uint32_t(im.ncomp);   // samples per source row
    const uint32_t oy = blockIdx.y, xc = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (oy >= uint32_t(w.nh) || xc >= rowlen) return;
    const ResizeTap t = taps[w.vtap_base + oy];
    const float *ws = weights + t.woff;
    const uint8_t *s = rgb + w.rgb_src_off + size_t(t.left) * rowlen + xc;
    float *o = tmp + w.tmp_off + size_t(oy) * rowlen + xc;
    if (xc + 4 <= rowlen) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int k = 0; k < t.n; k++) {
            uint32_t v;
            memcpy(&v, s + size_t(k) * rowlen, 4);   // a row need not start on a word
            const float wk = ws[k];
            a0 = __fadd_rn(a0, __fmul_rn(float(v & 255u), wk)); a1 = __fadd_rn(a1, __fmul_rn(float((v >> 8) & 255u), wk));
            a2 = __fadd_rn(a2, __fmul_rn(float((v >> 16) & 255u), wk)); a3 = __fadd_rn(a3, __fmul_rn(float(v >> 24), wk));
        }
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    } else {
        for (uint32_t j = 0; xc + j < rowlen; j++) {
            float acc = 0.0f;
            for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(float(s[size_t(k) * rowlen + j]), ws[k]));
            o[j] = acc;
        }
    }
}

// horizontal pass: dst[y][ox][c] = round(clamp(sum_i tmp[y][left+i][c] * w[i])); a workgroup is a piece of one row.  A lane is one output PIXEL: its
// tap record and weights are fetched once for the three channels and every tap is one 12-byte load (a lane per sample issued three times the loads and
// was bound by issuing them: 36 ms per 1024 pictures of 1500 x 844).  Each channel still sums its products in tap order, one rounding per operation.
__device__ __forceinline__ static uint8_t lanczos_round(float acc) {
    acc = acc < 0.0f ? 0.0f : (acc > 255.0f ? 255.0f : acc);
    int q = int(acc);                                     // round half away from zero (acc >= 0), without the
    q += (acc - float(q) >= 0.5f) ? 1 : 0;                // double rounding of int(acc + 0.5f)
    return uint8_t(q);
}
__global__ void __launch_bounds__(256) k_lanczos_h(const ImgDesc *imgs, const ResizeWork *work, const ResizeTap *taps, const float *weights,
                                                    const float *tmp, uint8_t *rgb) {
    const ResizeWork w = work[blockIdx.z];
    const ImgDesc &im = imgs[w.image];
    const uint32_t nc = uint32_t(im.ncomp), rowlen_in = uint32_t(im.width) * nc, rowlen_out = uint32_t(w.nw) * nc;
    const uint32_t y = blockIdx.y, ox = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= uint32_t(w.nh) || ox >= uint32_t(w.nw)) return;
    const ResizeTap t = taps[w.htap_base + ox];
    const float *ws = weights + t.woff;
    const float *s = tmp + w.tmp_off + size_t(y) * rowlen_in + size_t(t.left) * nc;
    uint8_t *d = rgb + w.rgb_dst_off + size_t(y) * rowlen_out + size_t(ox) * nc;
    if (nc == 3) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        for (int k = 0; k < t.n; k++) {
            const float wk = ws[k], s0 = s[3 * k], s1 = s[3 * k + 1], s2 = s[3 * k + 2];
            a0 = __fadd_rn(a0, __fmul_rn(s0, wk)); a1 = __fadd_rn(a1, __fmul_rn(s1, wk)); a2 = __fadd_rn(a2, __fmul_rn(s2, wk));
        }
        d[0] = lanczos_round(a0); d[1] = lanczos_round(a1); d[2] = lanczos_round(a2);
        return;
    }
    for (uint32_t c = 0; c < nc; c++) {
        float acc = 0.0f;
        for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(s[size_t(k) * nc + c], ws[k]));
        d[c] = lanczos_round(acc);
    }
}

// Both passes in one kernel, the f32 row between them in LDS: a workgroup is ONE output row -- phase 0 is the vertical pass of that row (every source column,
// the row's taps as scalars, four samples per lane: k_lanczos_v's arithmetic), phase 1 the horizontal pass over it (a lane per output pixel: k_lanczos_h's).
// The values and the order of every multiplication and addition are the two kernels' own, so the bytes are too; what goes is the intermediate image
// (1920 x 844 x 3 floats = 19.4 MB per picture written and read back through HBM, and 20 GB of pool per 1024 pictures).  CAP = floats of LDS: the launch
// takes the smallest that holds the batch's widest source row; rows beyond the largest keep the two-kernel form.
#define CSH_RZ_CAP_S 6144    // 2048 pixels of RGB: 24 KB, six workgroups per CU
#define CSH_RZ_CAP_L 16128   // 5376 pixels of RGB: 63 KB
template <int CAP>
__global__ void __launch_bounds__(256) k_lanczos_fused(const ImgDesc *imgs, const ResizeWork *work, const ResizeTap *taps, const float *weights, const uint8_t *rgb_in,
                                                        uint8_t *rgb_out) {
    CSH_SHARED float s_row[CAP];
    const ResizeWork w = work[blockIdx.y];
    const ImgDesc &im = imgs[w.image];
    const uint32_t nc = uint32_t(im.ncomp), rowlen = uint32_t(im.width) * nc, rowlen_out = uint32_t(w.nw) * nc;
    const uint32_t oy = blockIdx.x;
    CSH_PHASE_LOOP(2) {
        if (oy >= uint32_t(w.nh) || rowlen > uint32_t(CAP)) continue;
        if (phase == 0) {
            const ResizeTap t = taps[w.vtap_base + oy];
            const float *ws = weights + t.woff;
            for (uint32_t xc = threadIdx.x * 4u; xc < rowlen; xc += blockDim.x * 4u) {
                const uint8_t *s = rgb_in + w.rgb_src_off + size_t(t.left) * rowlen + xc;
                if (xc + 4 <= rowlen) {
                    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                    for (int k = 0; k < t.n; k++) {
                        uint32_t v;
                        memcpy(&v, s + size_t(k) * rowlen, 4);   // a row need not start on a word
                        const float wk = ws[k];
                        a0 = __fadd_rn(a0, __fmul_rn(float(v & 255u), wk)); a1 = __fadd_rn(a1, __fmul_rn(float((v >> 8) & 255u), wk));
                        a2 = __fadd_rn(a2, __fmul_rn(float((v >> 16) & 255u), wk)); a3 = __fadd_rn(a3, __fmul_rn(float(v >> 24), wk));
                    }
                    s_row[xc] = a0; s_row[xc + 1] = a1; s_row[xc + 2] = a2; s_row[xc + 3] = a3;
                } else {
                    for (uint32_t j = 0; xc + j < rowlen; j++) {
                        float acc = 0.0f;
                        for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(float(s[size_t(k) * rowlen + j]), ws[k]));
                        s_row[xc + j] = acc;
                    }
                }
            }
            continue;
        }
        uint8_t *drow = rgb_out + w.rgb_dst_off + size_t(oy) * rowlen_out;
        for (uint32_t ox = threadIdx.x; ox < uint32_t(w.nw); ox += blockDim.x) {
            const ResizeTap t = taps[w.htap_base + ox];
            const float *ws = weights + t.woff;
            const float *s = s_row + size_t(t.left) * nc;
            uint8_t *d = drow + size_t(ox) * nc;
            if (nc == 3) {
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
                for (int k = 0; k < t.n; k++) {
                    const float wk = ws[k], s0 = s[3 * k], s1 = s[3 * k + 1], s2 = s[3 * k + 2];
                    a0 = __fadd_rn(a0, __fmul_rn(s0, wk)); a1 = __fadd_rn(a1, __fmul_rn(s1, wk)); a2 = __fadd_rn(a2, __fmul_rn(s2, wk));
                }
                d[0] = lanczos_round(a0); d[1] = lanczos_round(a1); d[2] = lanczos_round(a2);
                continue;
            }
            for (uint32_t c = 0; c < nc; c++) {
                float acc = 0.0f;
                for (int k = 0; k < t.n; k++) acc = __fadd_rn(acc, __fmul_rn(s[size_t(k) * nc + c], ws[k]));
                d[c] = lanczos_round(acc);
            }
        }
    }
}
bool resize_is_fused(uint32_t max_row_in) { return max_row_in <= uint32_t(CSH_RZ_CAP_L) && !getenv("CSH_RESIZE_TWO_PASS"); }

// RGB -> full-resolution component planes (jccolor.c rgb_ycc_convert); pitch = the luma plane's padded width
__global__ void __launch_bounds__(256) k_rgb_to_planes(const ImgDesc *imgs, const ResizeWork *work, const uint8_t *rgb, uint8_t *planes) {
    const ResizeWork w = work[blockIdx.y];
    const ImgDesc &im = imgs[w.image];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.nw * w.nh) return;
    int y = i / w.nw, x = i - y * w.nw;
    const uint8_t *s = rgb + w.rgb_dst_off + size_t(i) * im.ncomp;
    const int pitch = im.src[0].real_bw * 8;
    if (im.ncomp == 1) { planes[im.splane_off[0] + size_t(y) * pitch + x] = s[0]; return; }
    int r = s[0], g = s[1], b = s[2];
    planes[im.splane_off[0] + size_t(y) * pitch + x] = uint8_t((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
    planes[im.splane_off[1] + size_t(y) * pitch + x] = uint8_t((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
    planes[im.splane_off[2] + size_t(y) * pitch + x] = uint8_t((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
}

void launch_resize(hipStream_t st, const ImgDesc *imgs, const ResizeWork *work, int nwork, const ResizeTap *taps, const float *weights,
                   uint8_t *planes, uint8_t *rgb, float *tmp, uint32_t max_src_px, uint64_t max_tmp, uint64_t max_dst, uint32_t max_row_in, uint32_t max_out_w, uint32_t max_nh, bool to_planes) {
    (void)max_tmp;
    if (!nwork) return;
    CSH_LAUNCH(k_planes_to_rgb, dim3((max_src_px + 255) / 256, nwork), dim3(256), st, imgs, work, planes, rgb);
    if (resize_is_fused(max_row_in)) {   // (source and result lie in different stretches of the RGB pool: rgb_src_off / rgb_dst_off)
        if (max_row_in <= uint32_t(CSH_RZ_CAP_S)) CSH_LAUNCH_PHASED(k_lanczos_fused<CSH_RZ_CAP_S>, 2, dim3(max_nh, nwork), dim3(256), st, imgs, work, taps, weights, rgb, rgb);
        else CSH_LAUNCH_PHASED(k_lanczos_fused<CSH_RZ_CAP_L>, 2, dim3(max_nh, nwork), dim3(256), st, imgs, work, taps, weights, rgb, rgb);
    } else {
        CSH_LAUNCH(k_lanczos_v, dim3((max_row_in + 1023) / 1024, max_nh, nwork), dim3(256), st, imgs, work, taps, weights, rgb, tmp);
        CSH_LAUNCH(k_lanczos_h, dim3((max_out_w + 255) / 256, max_nh, nwork), dim3(256), st, imgs, work, taps, weights, tmp, rgb);
    }
    if (to_planes) CSH_LAUNCH(k_rgb_to_planes, dim3(unsigned((max_dst + 255) / 256), nwork), dim3(256), st, imgs, work, rgb, planes);   // only the JPEG encoder reads planes; the WebP / PNG rows take the RGB
}

}  // namespace csh
