// webp_decode.cpp -- WebP INPUTS of the batch queue: container parsing on the host (RIFF / VP8 / VP8X chunk walk), the key frame on the
// device (k_webp_dec.hip), the decoded RGB stays in HBM and is handed to the encoders as csp_pixels -- what libcaesium's
// webp::compress and convert_in_memory do with libwebp's decoder in front (/root/reference/src/compressor.rs:289-305).
// Built: lossy (VP8) and lossless (VP8L) still pictures, with or without transparency (an ALPH chunk next to the VP8 frame, or a VP8L picture that is
// not opaque): such a picture leaves its RGB, its RGBA and its alpha plane in HBM (cswd_batch_alpha), and the caller picks what its encoder takes.
// Animation answers CS_ERR_UNSUPPORTED per file.
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/caesium_hip.h"
#include "devmem.hpp"
#include "webp_kernels.h"
#include "vp8_dec.h"
#include "vp8l_dec.h"

using namespace csh;

struct cswd_batch {
    int device = 0;
    hipStream_t stream = 0;
    bool have_stream = false;
    struct Item { int code = 0; std::string msg; int image = -1; };
    std::vector<Item> items;
    std::vector<csw::Vp8In> imgs;
    std::vector<uint8_t> pool;
    DevBuf<uint8_t> d_pool, d_work, d_rgb;
    DevBuf<csw::Vp8In> d_imgs;
    uint64_t work_bytes = 0, rgb_bytes = 0;
    bool ran = false;
    ~cswd_batch() { if (have_stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); } }   // nothing queued may outlive the device blocks
};

static uint32_t rd32le(const uint8_t *d) { return uint32_t(d[0]) | (uint32_t(d[1]) << 8) | (uint32_t(d[2]) << 16) | (uint32_t(d[3]) << 24); }

// RIFF walk: where the VP8 key frame lies; refuses what this build does not decode
static int parse_webp(const uint8_t *d, size_t n, size_t &off, size_t &len, uint32_t &w, uint32_t &h, bool &lossless, size_t &alph_off, size_t &alph_len, std::string &msg) {
    lossless = false; alph_off = 0; alph_len = 0;
    if (n < 20 || memcmp(d, "RIFF", 4) || memcmp(d + 8, "WEBP", 4)) { msg = "not a WebP file"; return CS_ERR_UNKNOWN_TYPE; }
    size_t end = size_t(rd32le(d + 4)) + 8;
    if (end > n) end = n;   // libwebp tolerates a RIFF size past the file's end as long as the chunks are there
    bool found = false;
    for (size_t i = 12; i + 8 <= end;) {
        const size_t cl = rd32le(d + i + 4);
        if (i + 8 + cl > n) { msg = "truncated WebP chunk"; return CS_ERR_BAD_WEBP; }
        if (!memcmp(d + i, "VP8 ", 4)) { if (!found) { off = i + 8; len = cl; found = true; } }
        else if (!memcmp(d + i, "VP8L", 4)) { if (!found) { off = i + 8; len = cl; found = true; lossless = true; } }
        else if (!memcmp(d + i, "ALPH", 4)) { if (!found && !alph_len) { alph_off = i + 8; alph_len = cl; } }   // the alpha plane of the VP8 frame that follows
        else if (!memcmp(d + i, "ANIM", 4) || !memcmp(d + i, "ANMF", 4)) { msg = "animated WebP input has no device path in this build"; return CS_ERR_UNSUPPORTED; }
        i += 8 + cl + (cl & 1);
    }
    if (!found || len < (lossless ? 5u : 10u)) { msg = "no VP8 frame in the WebP file"; return CS_ERR_BAD_WEBP; }
    const uint8_t *f = d + off;
    if (lossless) alph_len = 0;
    if (lossless) {   // signature, 14 + 14 bits of size minus one, alpha hint, version
        const uint32_t bits = rd32le(f + 1);
        if (f[0] != 0x2F || (bits >> 29) != 0) { msg = "malformed VP8L header"; return CS_ERR_BAD_WEBP; }
        w = (bits & 0x3FFFu) + 1; h = ((bits >> 14) & 0x3FFFu) + 1;
        return 0;
    }
    if ((f[0] & 1) || f[3] != 0x9D || f[4] != 0x01 || f[5] != 0x2A) { msg = "malformed VP8 frame header"; return CS_ERR_BAD_WEBP; }
    w = (uint32_t(f[6]) | (uint32_t(f[7]) << 8)) & 0x3FFF; h = (uint32_t(f[8]) | (uint32_t(f[9]) << 8)) & 0x3FFF;
    if (!w || !h) { msg = "empty VP8 frame"; return CS_ERR_BAD_WEBP; }
    return 0;
}

extern "C" int cswd_batch_create(const CByteArray *inputs, size_t count, int device, cswd_batch **out) {
    *out = nullptr;
    if (csh_device_count() <= device) { csh_set_error("no HIP device %d available (libcaesium_hip has no CPU path)", device); return CS_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { csh_set_error("hipSetDevice(%d) failed", device); return CS_ERR_NO_DEVICE; }
    std::unique_ptr<cswd_batch> b(new cswd_batch);
    b->device = device;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { csh_set_error("hipStreamCreate failed"); return CS_ERR_NO_DEVICE; }
    b->have_stream = true;
    b->items.resize(count);
    for (size_t n = 0; n < count; n++) {
        cswd_batch::Item &it = b->items[n];
        size_t off = 0, len = 0;
        uint32_t w = 0, h = 0;
        bool lossless = false;
        size_t alph_off = 0, alph_len = 0;
        it.code = parse_webp(inputs[n].data, inputs[n].length, off, len, w, h, lossless, alph_off, alph_len, it.msg);
        if (it.code) continue;
        csw::Vp8In im;
        memset(&im, 0, sizeof im);
        im.data_off = b->pool.size(); im.data_len = uint32_t(len);
        b->pool.insert(b->pool.end(), inputs[n].data + off, inputs[n].data + off + len);
        b->pool.resize((b->pool.size() + 15) & ~size_t(15));
        im.width = w; im.height = h; im.mbw = (w + 15) / 16; im.mbh = (h + 15) / 16;
        im.lossless = lossless ? 1u : 0u;
        im.debug = getenv("CSH_WEBP_DEBUG") ? uint32_t(atoi(getenv("CSH_WEBP_DEBUG"))) : 0u;
        uint64_t work = lossless ? csw::vp8l_work_bytes(w, h, len) : csw::vp8_work_bytes(im.mbw, im.mbh);
        if (alph_len) {
            im.alph_off = b->pool.size(); im.alph_len = uint32_t(alph_len);
            b->pool.insert(b->pool.end(), inputs[n].data + alph_off, inputs[n].data + alph_off + alph_len);
            b->pool.resize((b->pool.size() + 15) & ~size_t(15));
            work = std::max(work, csw::vp8l_work_bytes(w, h, alph_len));   // the plane's VP8L stream is decoded in the frame's work area, after the frame
        }
        im.work_off = b->work_bytes; b->work_bytes += (work + 63) & ~uint64_t(63);
        im.rgb_off = b->rgb_bytes; b->rgb_bytes += (uint64_t(w) * h * 3 + 63) & ~uint64_t(63);
        im.rgba_off = im.a_off = ~0ull;
        if (lossless || alph_len) {   // a picture that may turn out not to be opaque: room for its RGBA and its alpha plane
            im.rgba_off = b->rgb_bytes; b->rgb_bytes += (uint64_t(w) * h * 4 + 63) & ~uint64_t(63);
            im.a_off = b->rgb_bytes; b->rgb_bytes += (uint64_t(w) * h + 63) & ~uint64_t(63);
        }
        it.image = int(b->imgs.size());
        b->imgs.push_back(im);
    }
    if (!b->imgs.empty()) {
        if (b->d_pool.upload(b->pool, b->stream) || b->d_imgs.upload(b->imgs, b->stream) || b->d_work.alloc(b->work_bytes + 64) || b->d_rgb.alloc(b->rgb_bytes + 64)) return CS_ERR_NO_DEVICE;
        if (hipStreamSynchronize(b->stream) != hipSuccess) { csh_set_error("upload failed"); return CS_ERR_NO_DEVICE; }
    }
    *out = b.release();
    return 0;
}

extern "C" int cswd_batch_run(cswd_batch *b) {
    if (b->imgs.empty()) { b->ran = true; return 0; }
    if (hipSetDevice(b->device) != hipSuccess) { csh_set_error("hipSetDevice failed"); return CS_ERR_NO_DEVICE; }
    int nsteps = 0, psteps_lossless = 0, psteps_alpha = 0;   // of the wave fronts: the lossy frames' mbw + 2 mbh; the lossless pictures' and the alpha planes' predictor steps
    for (const csw::Vp8In &im : b->imgs) {
        if (im.lossless) psteps_lossless = std::max(psteps_lossless, int(csw::vp8l_pred_steps(im.width, im.height)));
        else {
            nsteps = std::max(nsteps, int(im.mbw + 2 * im.mbh));
            if (im.alph_len) psteps_alpha = std::max(psteps_alpha, int(csw::vp8l_pred_steps(im.width, im.height)));
        }
    }
    csw::launch_vp8_decode(b->stream, b->d_pool.p, b->d_imgs.p, int(b->imgs.size()), b->d_work.p, b->d_rgb.p, nsteps, psteps_lossless, psteps_alpha);
    if (hipMemcpyAsync(b->imgs.data(), b->d_imgs.p, b->imgs.size() * sizeof(csw::Vp8In), hipMemcpyDeviceToHost, b->stream) != hipSuccess ||
        hipStreamSynchronize(b->stream) != hipSuccess || hipGetLastError() != hipSuccess) { csh_set_error("VP8 decode failed on the device"); return CS_ERR_NO_DEVICE; }
    for (cswd_batch::Item &it : b->items)
        if (it.image >= 0 && b->imgs[size_t(it.image)].status) {
            const uint32_t st = b->imgs[size_t(it.image)].status;
            it.code = st >= 2 ? CS_ERR_UNSUPPORTED : CS_ERR_BAD_WEBP;
            it.msg = st == 3 ? "WebP input with transparency: no room was reserved for its alpha" : st == 2 ? "WebP frame beyond this build (frame type / prefix-code work area)" : "malformed WebP stream";
        }
    b->ran = true;
    return 0;
}

extern "C" int cswd_batch_pixels(cswd_batch *b, size_t image, const uint8_t **device_pixels, uint32_t *width, uint32_t *height, uint32_t *channels, const char **message) {
    if (!b->ran || image >= b->items.size()) { csh_set_error("cswd_batch_pixels: batch not run / index out of range"); return -1; }
    const cswd_batch::Item &it = b->items[image];
    if (message) *message = it.msg.c_str();
    if (it.code) return it.code;
    const csw::Vp8In &im = b->imgs[size_t(it.image)];
    *device_pixels = b->d_rgb.p + im.rgb_off; *width = im.width; *height = im.height; *channels = 3;
    return 0;
}

// a picture that is not opaque: its RGBA (width * height * 4) and its alpha plane (width * height) in device memory; both null for an opaque one
extern "C" int cswd_batch_alpha(cswd_batch *b, size_t image, const uint8_t **device_rgba, const uint8_t **device_alpha) {
    *device_rgba = nullptr; *device_alpha = nullptr;
    if (!b->ran || image >= b->items.size()) { csh_set_error("cswd_batch_alpha: batch not run / index out of range"); return -1; }
    const cswd_batch::Item &it = b->items[image];
    if (it.code) return it.code;
    const csw::Vp8In &im = b->imgs[size_t(it.image)];
    if (im.has_alpha) { *device_rgba = b->d_rgb.p + im.rgba_off; *device_alpha = b->d_rgb.p + im.a_off; }
    return 0;
}

extern "C" int cswd_batch_read_pixels(cswd_batch *b, size_t image, uint8_t *dst) {
    const uint8_t *p; uint32_t w, h, c; const char *m;
    int rc = cswd_batch_pixels(b, image, &p, &w, &h, &c, &m);
    if (rc) return rc;
    if (csh_copy_wait(dst, p, size_t(w) * h * c, hipMemcpyDeviceToHost, b->stream) != hipSuccess) { csh_set_error("D2H failed"); return CS_ERR_NO_DEVICE; }
    return 0;
}

extern "C" int cswd_batch_read_rgba(cswd_batch *b, size_t image, uint8_t *dst) {   // width * height * 4 bytes of a picture that is not opaque; 1 for an opaque one
    const uint8_t *rgba, *a;
    const int rc = cswd_batch_alpha(b, image, &rgba, &a);
    if (rc) return rc;
    if (!rgba) return 1;
    const csw::Vp8In &im = b->imgs[size_t(b->items[image].image)];
    if (csh_copy_wait(dst, rgba, size_t(im.width) * im.height * 4, hipMemcpyDeviceToHost, b->stream) != hipSuccess) { csh_set_error("D2H failed"); return CS_ERR_NO_DEVICE; }
    return 0;
}

extern "C" void cswd_batch_destroy(cswd_batch *b) { delete b; }

// RGB and an alpha plane of the same size, both in device memory (the two halves of a picture with transparency after their resize) -> one RGBA picture
struct cswd_rgba { DevBuf<uint8_t> d; };
extern "C" int cswd_rgba_join(const uint8_t *device_rgb, const uint8_t *device_alpha, uint32_t width, uint32_t height, int device, cswd_rgba **out, const uint8_t **device_rgba) {
    *out = nullptr; *device_rgba = nullptr;
    if (!device_rgb || !device_alpha || !width || !height) { csh_set_error("cswd_rgba_join: null pixels / empty picture"); return -1; }
    if (hipSetDevice(device) != hipSuccess) { csh_set_error("hipSetDevice failed"); return CS_ERR_NO_DEVICE; }
    cswd_rgba *r = new cswd_rgba;
    const uint64_t npx = uint64_t(width) * height;
    if (r->d.alloc(size_t(npx) * 4)) { delete r; return CS_ERR_NO_DEVICE; }
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { delete r; csh_set_error("hipStreamCreate failed"); return CS_ERR_NO_DEVICE; }
    csw::launch_rgba_join(st, device_rgb, device_alpha, r->d.p, npx);
    const bool ok = hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
    (void)hipStreamDestroy(st);
    if (!ok) { delete r; csh_set_error("k_rgba_join failed on the device"); return CS_ERR_NO_DEVICE; }
    *out = r; *device_rgba = r->d.p;
    return 0;
}
extern "C" void cswd_rgba_destroy(cswd_rgba *r) { delete r; }
