// vp8l_encode.cpp -- lossless WebP OUTPUT of the batch queue (webp.lossless; libcaesium's webp::compress with libwebp's lossless coder,
// /root/reference/src/compressor.rs:427-429 and 289-305): pixels that are already in device memory -> one VP8L file each
// (k_vp8l_enc.hip).  Sources are the decoders of this library (WebP inputs: webp_decode.cpp; JPEG inputs: the pixel tap of the JPEG row).
#include <cstring>
#include <vector>

#include "../../include/caesium_hip.h"
#include "devmem.hpp"
#include "webp_kernels.h"

using namespace csh;

static CCSResult make_res(uint32_t code, const char *msg) {
    CCSResult r; r.success = code == 0; r.code = code; r.error_message = nullptr;
    if (code && msg) { size_t n = strlen(msg); char *m = static_cast<char *>(malloc(n + 1)); memcpy(m, msg, n + 1); r.error_message = m; }
    return r;
}

extern "C" int csl_encode_pixels(const csp_pixels *px, size_t count, int device, CByteArray *outputs, CCSResult *results) {
    for (size_t i = 0; i < count; i++) { outputs[i].data = nullptr; outputs[i].length = 0; }
    if (csh_device_count() <= device || hipSetDevice(device) != hipSuccess) {
        for (size_t i = 0; i < count; i++) results[i] = make_res(CS_ERR_NO_DEVICE, "no HIP device available (libcaesium_hip has no CPU path)");
        return int(count);
    }
    int failed = 0;
    std::vector<csw::Vp8lImg> imgs;
    std::vector<size_t> at;
    uint64_t work = 0, modes = 0, out = 0, max_px = 0;
    uint32_t max_blocks = 0;
    for (size_t i = 0; i < count; i++) {
        const uint32_t ch = px[i].channels;
        if (!((ch >= 1 && ch <= 4) || ch == csw::VP8L_ALPHA_OF + 2 || ch == csw::VP8L_ALPHA_OF + 4) || !px[i].width || !px[i].height || px[i].width > 16384 || px[i].height > 16384) {
            results[i] = make_res(CS_ERR_UNSUPPORTED, "lossless WebP output takes 8-bit grey / RGB pictures, with or without alpha, of at most 16384 x 16384"); failed++; continue;
        }
        csw::Vp8lImg im;
        memset(&im, 0, sizeof im);
        im.rgb = px[i].device_pixels; im.width = px[i].width; im.height = px[i].height; im.channels = px[i].channels;
        im.bw = (im.width + 15) / 16; im.bh = (im.height + 15) / 16;
        const uint64_t npx = uint64_t(im.width) * im.height;
        im.res_off = work; work += (npx + 63) & ~uint64_t(63);
        im.mode_off = modes; modes += (uint64_t(im.bw) * im.bh + 63) & ~uint64_t(63);
        const uint64_t cap = 8 * npx + 2 * uint64_t(im.bw) * im.bh + 8192;   // four codes of at most 15 bits per pixel, the mode image, the code descriptions
        if (cap > 0xFFFFFFF0ull) { results[i] = make_res(CS_ERR_UNSUPPORTED, "picture too large for one lossless WebP batch item"); failed++; continue; }
        im.out_off = out; im.out_cap = uint32_t(cap); out += (cap + 255) & ~uint64_t(255);
        max_px = std::max(max_px, npx); max_blocks = std::max(max_blocks, im.bw * im.bh);
        imgs.push_back(im); at.push_back(i);
    }
    if (imgs.empty()) return failed;
    hipStream_t st = 0;
    bool have_st = false;
    DevBuf<csw::Vp8lImg> d_imgs;
    DevBuf<uint32_t> d_work, d_hist, d_len, d_status;
    DevBuf<uint8_t> d_modes, d_out;
    std::vector<uint32_t> len(imgs.size()), status(imgs.size());
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    have_st = ok;
    ok = ok && !d_imgs.upload(imgs, st) && !d_work.alloc(work + 64) && !d_hist.alloc(imgs.size() * 1024 + 8) && !d_hist.zero(st) && !d_len.alloc(imgs.size() + 1) && !d_status.alloc(imgs.size() + 1) &&
         !d_modes.alloc(modes + 64) && !d_out.alloc(out + 256);
    if (ok) {
        csw::launch_vp8l_encode(st, d_imgs.p, int(imgs.size()), max_blocks, max_px, d_work.p, d_modes.p, d_hist.p, d_out.p, d_len.p, d_status.p);
        ok = hipMemcpyAsync(len.data(), d_len.p, len.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipMemcpyAsync(status.data(), d_status.p, status.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
    }
    for (size_t k = 0; k < imgs.size(); k++) {
        const size_t i = at[k];
        if (!ok) { results[i] = make_res(CS_ERR_NO_DEVICE, "lossless WebP coding failed on the device"); failed++; continue; }
        if (status[k] || !len[k]) { results[i] = make_res(CS_ERR_POOL_OVERFLOW, "lossless WebP output larger than its region"); failed++; continue; }
        outputs[i].data = static_cast<uint8_t *>(malloc(len[k]));
        if (!outputs[i].data || csh_copy_wait(outputs[i].data, d_out.p + imgs[k].out_off, len[k], hipMemcpyDeviceToHost, st) != hipSuccess) {
            free(outputs[i].data); outputs[i].data = nullptr; results[i] = make_res(CS_ERR_NO_DEVICE, "D2H failed"); failed++; continue;
        }
        outputs[i].length = len[k];
        results[i] = make_res(0, nullptr);
    }
    if (have_st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }   // nothing queued may outlive the device blocks
    return failed;
}

// A lossy file and its alpha plane as a VP8L file (csl_encode_pixels over the plane as a grey picture) -> the extended-format file libwebp writes for a picture with
// transparency: VP8X (alpha flag, canvas size), ALPH (one header byte: lossless compression, no filter, no pre-processing; then the VP8L stream without its five
// header bytes -- signature and sizes, which ALPH implies; its green channel is the alpha), the VP8 frame.  lossy is replaced; 0 ok.
extern "C" int csl_attach_alpha(CByteArray *lossy, const CByteArray *alpha_vp8l, uint32_t width, uint32_t height) {
    if (!lossy || !lossy->data || lossy->length < 30 || memcmp(lossy->data + 12, "VP8 ", 4) || !alpha_vp8l || !alpha_vp8l->data || alpha_vp8l->length < 26 || memcmp(alpha_vp8l->data + 12, "VP8L", 4)) return -1;
    const uint8_t *a = alpha_vp8l->data;
    const size_t pl = size_t(a[16]) | (size_t(a[17]) << 8) | (size_t(a[18]) << 16) | (size_t(a[19]) << 24);
    if (pl < 5 || 20 + pl > alpha_vp8l->length) return -1;
    const size_t vp8 = lossy->length - 12, apay = pl - 5, alph = 1 + apay, total = 12 + 18 + 8 + alph + (alph & 1) + vp8;
    uint8_t *o = static_cast<uint8_t *>(malloc(total)), *w = o;
    if (!o) return -1;
    auto le32 = [](uint8_t *q, uint32_t v) { q[0] = uint8_t(v); q[1] = uint8_t(v >> 8); q[2] = uint8_t(v >> 16); q[3] = uint8_t(v >> 24); };
    memcpy(w, "RIFF", 4); le32(w + 4, uint32_t(total - 8)); memcpy(w + 8, "WEBPVP8X", 8); le32(w + 16, 10);
    w[20] = 0x10; w[21] = w[22] = w[23] = 0;
    const uint32_t cw = width - 1, ch = height - 1;
    w[24] = uint8_t(cw); w[25] = uint8_t(cw >> 8); w[26] = uint8_t(cw >> 16); w[27] = uint8_t(ch); w[28] = uint8_t(ch >> 8); w[29] = uint8_t(ch >> 16);
    w += 30;
    memcpy(w, "ALPH", 4); le32(w + 4, uint32_t(alph)); w[8] = 0x01; memcpy(w + 9, a + 25, apay); w += 8 + alph;
    if (alph & 1) *w++ = 0;
    memcpy(w, lossy->data + 12, vp8);
    free(lossy->data);
    lossy->data = o; lossy->length = total;
    return 0;
}
