// capi.cpp -- the libcaesium-shaped entry points on top of the device batch queue.
// Reference semantics: /root/reference/src/compressor.rs:287-306 (call shapes), :411-446 (parameters).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <thread>
#include <vector>

#include "../../include/caesium_hip.h"

extern "C" {

void cs_default_parameters(CCSParameters *p) {
    memset(p, 0, sizeof *p);
    p->jpeg_quality = 80; p->jpeg_chroma_subsampling = 0; p->jpeg_progressive = true; p->jpeg_optimize = false; p->jpeg_preserve_icc = true;
    p->png_quality = 80; p->png_optimization_level = 3; p->gif_quality = 80; p->webp_quality = 80;
    p->tiff_deflate_level = 6;
}

static CCSResult make_result(int code, const char *msg) {
    CCSResult r;
    r.success = code == 0; r.code = uint32_t(code); r.error_message = nullptr;
    if (code && msg) { size_t n = strlen(msg); char *m = (char *)malloc(n + 1); memcpy(m, msg, n + 1); r.error_message = m; }
    return r;
}

// one device batch = at most CS_GROUP files: launch grids index (image, scan) pairs in gridDim.y (<= 65535), and a group
// of 2048 1080p files already occupies ~40 GB of HBM and tens of thousands of workgroups per launch (cs_batch_extent's cap; the JPEG row
// cuts its groups further into spans of CS_SPAN files -- CSH_GROUP overrides that span, see jpeg_batch_compress)
enum { CS_GROUP = 2048 };
// declared pixel count of a JPEG / PNG (0 when the header does not say): what a file will occupy on the device is known before
// anything is decoded
// canvas of a WebP file from its first chunk: VP8X (24-bit width-1 / height-1), a bare VP8 key frame (14-bit sizes behind the start
// code) or a bare VP8L stream (14 + 14 bits behind the signature); 0 when the header does not say
static uint64_t webp_declared_pixels(const uint8_t *d, size_t n) {
    if (n < 30 || memcmp(d, "RIFF", 4) || memcmp(d + 8, "WEBP", 4)) return 0;
    const uint8_t *f = d + 20;
    if (!memcmp(d + 12, "VP8X", 4)) return (uint64_t(f[4] | (f[5] << 8) | (f[6] << 16)) + 1) * (uint64_t(f[7] | (f[8] << 8) | (f[9] << 16)) + 1);
    if (!memcmp(d + 12, "VP8 ", 4)) return (f[3] == 0x9D && f[4] == 0x01 && f[5] == 0x2A) ? uint64_t((f[6] | (f[7] << 8)) & 0x3FFF) * uint64_t((f[8] | (f[9] << 8)) & 0x3FFF) : 0;
    if (!memcmp(d + 12, "VP8L", 4) && f[0] == 0x2F) { const uint32_t b = uint32_t(f[1]) | (uint32_t(f[2]) << 8) | (uint32_t(f[3]) << 16) | (uint32_t(f[4]) << 24); return uint64_t((b & 0x3FFF) + 1) * uint64_t(((b >> 14) & 0x3FFF) + 1); }
    return 0;
}
static uint64_t declared_pixels(const uint8_t *d, size_t n) {
    if (n >= 30 && !memcmp(d, "RIFF", 4)) return webp_declared_pixels(d, n);
    if (n >= 24 && !memcmp(d, "\x89PNG\r\n\x1a\n", 8)) {
        const uint64_t w = (uint64_t(d[16]) << 24) | (d[17] << 16) | (d[18] << 8) | d[19], h = (uint64_t(d[20]) << 24) | (d[21] << 16) | (d[22] << 8) | d[23];
        return w * h;
    }
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return 0;
    for (size_t i = 2; i + 4 <= n;) {
        if (d[i] != 0xFF) { i++; continue; }
        const int m = d[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
        const size_t L = (size_t(d[i + 2]) << 8) | d[i + 3];
        if (L < 2 || i + 2 + L > n) break;
        if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC && L >= 7) return (uint64_t((d[i + 5] << 8) | d[i + 6])) * uint64_t((d[i + 7] << 8) | d[i + 8]);
        if (m == 0xDA) break;
        i += 2 + L;
    }
    return 0;
}
size_t cs_batch_extent(const CByteArray *inputs, size_t count) {
    const uint64_t byte_cap = uint64_t(2) << 30, pool_cap = uint64_t(96) << 30;
    uint64_t bytes = 0, pools = 0;
    size_t n = 0;
    while (n < count && n < size_t(CS_GROUP)) {
        const uint64_t len = inputs[n].length;
        // ~25 bytes of pools per pixel on the JPEG path (coefficients both ways, planes, per-unit arrays), 4 x the file for the streams
        const uint64_t est = declared_pixels(inputs[n].data, inputs[n].length) * 25 + 4 * len + (1u << 20);
        if (n && (bytes + len > byte_cap || pools + est > pool_cap)) break;
        bytes += len; pools += est; n++;
    }
    return n;
}
// Device batches of one boundary call: at most CS_SPAN files each (CSH_GROUP overrides), two in flight on two host threads -- one is parsed
// and uploaded while the other is in its kernels, and the second batch onwards reuses the first ones' device pools (devmem.hpp).  One batch
// of 2048 files allocated ~100 GB at once: 0.7 s when everything was cached, 4-10 s when it was not (measured); batches of
// 256 take 13 GB each and the call's time no longer depends on what the cache happens to hold (2048 files: 0.50-0.55 s warm, 0.9 s cold).
enum { CS_SPAN = 256 };
static int jpeg_span_compress(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results, size_t span);
static int jpeg_batch_compress(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results) {
    for (size_t i = 0; i < count; i++) { outputs[i].data = nullptr; outputs[i].length = 0; }
    const size_t span = getenv("CSH_GROUP") ? std::max<size_t>(1, size_t(atol(getenv("CSH_GROUP")))) : size_t(CS_SPAN);
    if (count <= span) return jpeg_span_compress(inputs, count, p, device, outputs, results, span);
    std::vector<std::pair<size_t, size_t>> spans;   // [first, count)
    for (size_t g0 = 0; g0 < count;) { const size_t n = std::min(span, count - g0); spans.emplace_back(g0, n); g0 += n; }
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto worker = [&]() {
        for (size_t k; (k = next++) < spans.size();)
            failed += jpeg_span_compress(inputs + spans[k].first, spans[k].second, p, device, outputs + spans[k].first, results ? results + spans[k].first : nullptr, span);
    };
    const size_t nworkers = std::min<size_t>(spans.size(), getenv("CSH_WORKERS") ? std::max<size_t>(1, size_t(atol(getenv("CSH_WORKERS")))) : 3);   // three batches in flight: one in its kernels, one being parsed and uploaded, one being fetched (2048 x 1080p: 181 ms with 2, 156 with 3, no gain beyond)
    std::vector<std::thread> others;
    for (size_t t = 1; t < nworkers; t++) others.emplace_back(worker);
    worker();
    for (auto &t : others) t.join();
    return failed.load();
}
static int jpeg_span_compress(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results, size_t span) {
    int failed_total = 0;
    size_t limit = span;   // halved when a whole group fails for want of memory: one oversized neighbour must not fail the others;
    for (size_t g0 = 0; g0 < count;) {   // back to the full group once a group has gone through (the shortage belonged to those files)
        size_t n = cs_batch_extent(inputs + g0, count - g0);
        if (n > limit) n = limit;
        csh_batch *b = nullptr;
        const bool trace = getenv("CSH_TRACE") != nullptr;   // host-side phase times of the boundary call, on stderr
        auto t0 = std::chrono::steady_clock::now();
        int rc = csh_batch_create(inputs + g0, n, p, device, &b);
        auto t1 = std::chrono::steady_clock::now();
        if (rc == 0) rc = csh_batch_run(b, nullptr);
        auto t2 = std::chrono::steady_clock::now();
        if (rc != 0) {
            csh_batch_destroy(b);
            if (n > 1 && (rc == CS_ERR_NO_DEVICE || rc == CS_ERR_POOL_OVERFLOW) && csh_device_count() > device) { limit = (n + 1) / 2; continue; }   // same files again, in smaller groups
            for (size_t i = 0; i < n; i++) if (results) results[g0 + i] = make_result(rc, csh_last_error());
            failed_total += int(n);
            g0 += n;
            continue;
        }
        int failed = csh_batch_fetch(b, outputs + g0, results ? results + g0 : nullptr);
        auto t3 = std::chrono::steady_clock::now();
        csh_batch_destroy(b);
        if (trace) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[csh] %zu files on device %d: create %.1f ms, run %.1f ms, fetch %.1f ms, destroy %.1f ms\n", n, device, ms(t0, t1), ms(t1, t2), ms(t2, t3),
                    ms(t3, std::chrono::steady_clock::now()));
        }
        failed_total += failed < 0 ? int(n) : failed;
        g0 += n;
        limit = span;
    }
    return failed_total;
}

static int sniff(const uint8_t *d, size_t n);
// lossless PNG (png.optimize): groups sized by what they occupy in HBM -- per file the inflated stream, the pixels, one
// filtered stream per trial slot (up to 10) and the output region, ~13 x the raw size
static int png_batch_compress(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results, bool to_webp = false) {
    int failed_total = 0;
    // png.force_zopfli (--zopfli): libcaesium hands the streams to zopfli; here the same coder runs CSP_DEEP_ITERS_ZOPFLI passes of its cost model
    // over the chunks that take the min-cost-path parse (png_pipeline.cpp, png_parse.h)
    const uint64_t budget = uint64_t(96) << 30;
    for (size_t g0 = 0; g0 < count;) {
        uint64_t bytes = 0;
        size_t n = 0;
        while (g0 + n < count && n < 4096) {
            const uint8_t *d = inputs[g0 + n].data;
            uint64_t est = 1 << 20;
            if (inputs[g0 + n].length >= 33) {
                const uint64_t w = (uint64_t(d[16]) << 24) | (d[17] << 16) | (d[18] << 8) | d[19], h = (uint64_t(d[20]) << 24) | (d[21] << 16) | (d[22] << 8) | d[23];
                est += w * h * 8 * 14;   // at most 8 bytes per pixel
            }
            if (n && bytes + est > budget) break;
            bytes += est; n++;
        }
        csp_batch *b = nullptr;
        const bool trace = getenv("CSH_TRACE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        int rc = to_webp ? csp_batch_create_webp(inputs + g0, n, p, device, &b) : csp_batch_create(inputs + g0, n, p, device, &b);
        auto t1 = std::chrono::steady_clock::now();
        if (rc == 0) rc = csp_batch_run(b, nullptr);
        auto t2 = std::chrono::steady_clock::now();
        if (rc != 0) {
            for (size_t i = 0; i < n; i++) { outputs[g0 + i].data = nullptr; outputs[g0 + i].length = 0; if (results) results[g0 + i] = make_result(rc, csh_last_error()); }
            failed_total += int(n);
        } else {
            int failed = csp_batch_fetch(b, outputs + g0, results ? results + g0 : nullptr);
            failed_total += failed < 0 ? int(n) : failed;
        }
        csp_batch_destroy(b);
        if (trace) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[csh] %zu PNG files on device %d: create %.1f ms, run %.1f ms, fetch+destroy %.1f ms\n", n, device, ms(t0, t1), ms(t1, t2), ms(t2, std::chrono::steady_clock::now()));
        }
        g0 += n;
    }
    return failed_total;
}

// WebP inputs (libcaesium: libwebp decodes, then webp::compress / convert_in_memory, compressor.rs:289-305): the device decodes the
// key frame (cswd_batch), the RGB stays in HBM and goes to the encoder of the target: the WebP encoder at webp.quality (`compress`
// on a WebP file), the JPEG or the PNG row (conversions).  A size, if given, is applied by the JPEG row's Lanczos branch on the way.
// keep_metadata on WebP -> WebP (libcaesium's webp::compress: ICC profile and EXIF of the SOURCE read with img-parts and set on the encoded
// file, which turns it into the extended format): RIFF { VP8X(flags, canvas), ICCP?, VP8, EXIF? }.  XMP is not carried (img-parts' API
// sets profile and EXIF only).  The chunk walk is the container's: host work, like the JPEG row's marker copy.
static void webp_carry_metadata(const CByteArray &src, CByteArray &out) {
    auto rd32 = [](const uint8_t *d) { return uint32_t(d[0]) | (uint32_t(d[1]) << 8) | (uint32_t(d[2]) << 16) | (uint32_t(d[3]) << 24); };
    const uint8_t *d = src.data;
    const size_t n = src.length;
    const bool extended = out.data && out.length >= 38 && !memcmp(out.data + 12, "VP8X", 4);   // made here: VP8X, ALPH, VP8 (a picture with transparency)
    const bool lossless = out.data && out.length >= 26 && !memcmp(out.data + 12, "VP8L", 4) && out.data[20] == 0x2F;   // webp.lossless: a bare VP8L stream
    if (n < 20 || !out.data || out.length < 26 || (memcmp(out.data + 12, "VP8 ", 4) && !extended && !lossless)) return;
    const uint8_t *icc = nullptr, *exif = nullptr;
    size_t icc_len = 0, exif_len = 0;
    for (size_t i = 12; i + 8 <= n;) {
        const size_t cl = rd32(d + i + 4);
        if (i + 8 + cl > n) break;
        if (!memcmp(d + i, "ICCP", 4) && !icc) { icc = d + i + 8; icc_len = cl; }
        if (!memcmp(d + i, "EXIF", 4) && !exif) { exif = d + i + 8; exif_len = cl; }
        i += 8 + cl + (cl & 1);
    }
    if (!icc && !exif) return;
    const uint8_t *f = out.data + 20;   // VP8 frame header: tag (3), start code (3), 14-bit width and height -- or the VP8X chunk's flags and canvas size
    // a VP8L stream says its size (14 + 14 bits, each minus one) and whether its alpha is in use right behind the signature byte
    const uint32_t lbits = lossless ? uint32_t(f[1]) | (uint32_t(f[2]) << 8) | (uint32_t(f[3]) << 16) | (uint32_t(f[4]) << 24) : 0u;
    const uint32_t w = lossless ? (lbits & 0x3FFFu) + 1 : extended ? (uint32_t(f[4]) | (uint32_t(f[5]) << 8) | (uint32_t(f[6]) << 16)) + 1 : (uint32_t(f[6]) | (uint32_t(f[7]) << 8)) & 0x3FFFu;
    const uint32_t h = lossless ? ((lbits >> 14) & 0x3FFFu) + 1 : extended ? (uint32_t(f[7]) | (uint32_t(f[8]) << 8) | (uint32_t(f[9]) << 16)) + 1 : (uint32_t(f[8]) | (uint32_t(f[9]) << 8)) & 0x3FFFu;
    const uint8_t flags0 = lossless ? uint8_t(((lbits >> 28) & 1u) ? 0x10 : 0) : extended ? f[0] : 0;   // VP8X alpha flag from the stream's alpha_is_used bit
    const size_t body0 = extended ? 30 : 12, body = out.length - body0;   // the chunks behind the file header (and behind VP8X), with their padding
    const size_t total = 12 + 18 + (icc ? 8 + icc_len + (icc_len & 1) : 0) + body + (exif ? 8 + exif_len + (exif_len & 1) : 0);
    uint8_t *o = static_cast<uint8_t *>(malloc(total));
    if (!o) return;
    auto wr32 = [](uint8_t *q, uint32_t v) { q[0] = uint8_t(v); q[1] = uint8_t(v >> 8); q[2] = uint8_t(v >> 16); q[3] = uint8_t(v >> 24); };
    size_t at = 0;
    memcpy(o, "RIFF", 4); wr32(o + 4, uint32_t(total - 8)); memcpy(o + 8, "WEBP", 4); at = 12;
    memcpy(o + at, "VP8X", 4); wr32(o + at + 4, 10);
    o[at + 8] = uint8_t(flags0 | (icc ? 0x20 : 0) | (exif ? 0x08 : 0)); o[at + 9] = o[at + 10] = o[at + 11] = 0;
    const uint32_t cw = w - 1, chh = h - 1;
    o[at + 12] = uint8_t(cw); o[at + 13] = uint8_t(cw >> 8); o[at + 14] = uint8_t(cw >> 16);
    o[at + 15] = uint8_t(chh); o[at + 16] = uint8_t(chh >> 8); o[at + 17] = uint8_t(chh >> 16);
    at += 18;
    auto chunk = [&](const char *id, const uint8_t *pl, size_t len) {
        memcpy(o + at, id, 4); wr32(o + at + 4, uint32_t(len)); memcpy(o + at + 8, pl, len); at += 8 + len;
        if (len & 1) o[at++] = 0;
    };
    if (icc) chunk("ICCP", icc, icc_len);
    memcpy(o + at, out.data + body0, body); at += body;
    if (exif) chunk("EXIF", exif, exif_len);
    free(out.data);
    out.data = o; out.length = total;
}

static int webp_inputs(const CByteArray *inputs, size_t count, const CCSParameters *p, uint32_t target, int device, CByteArray *outputs, CCSResult *results) {
    int failed_total = 0;
    // groups by what the decoder reserves, not by file bytes: per declared pixel 3 B of RGB, 5 B of alpha room and 8 B of VP8L work area
    // (+ 4 MiB + 16 x the file), so ~20 B per pixel of the canvas the header declares -- a tiny file may declare 16383 x 16383.  A group
    // that still fails for want of memory is halved and tried again (as the JPEG row does), so that one oversized header fails alone.
    const uint64_t pool_cap = uint64_t(64) << 30;
    size_t limit = 512;
    for (size_t g0 = 0, n = 0; g0 < count; g0 += n) {
        uint64_t bytes = 0, pools = 0;
        for (n = 0; g0 + n < count && n < limit; n++) {
            const uint64_t est = webp_declared_pixels(inputs[g0 + n].data, inputs[g0 + n].length) * 20 + 16 * uint64_t(inputs[g0 + n].length) + (uint64_t(4) << 20);
            if (n && (bytes + inputs[g0 + n].length > (uint64_t(256) << 20) || pools + est > pool_cap)) break;
            bytes += inputs[g0 + n].length; pools += est;
        }
        for (size_t k = 0; k < n; k++) { outputs[g0 + k].data = nullptr; outputs[g0 + k].length = 0; }
        cswd_batch *wb = nullptr;
        int rc = cswd_batch_create(inputs + g0, n, device, &wb);
        if (rc == 0) rc = cswd_batch_run(wb);
        if (rc != 0 && n > 1 && (rc == CS_ERR_NO_DEVICE || rc == CS_ERR_POOL_OVERFLOW) && csh_device_count() > device) {   // same files again, in smaller groups
            cswd_batch_destroy(wb);
            limit = (n + 1) / 2; n = 0;
            continue;
        }
        limit = 512;
        std::vector<csp_pixels> px;
        std::vector<const uint8_t *> aplane;   // per picture: its alpha plane in device memory, or null (opaque)
        std::vector<csp_pixels> rgb3;          // per picture: its RGB whatever its transparency ...
        std::vector<const uint8_t *> rplane;   // ... and the plane of a picture the RGBA consumers (PNG, lossless WebP) get: what a resize works on
        std::vector<size_t> at;
        for (size_t k = 0; k < n && rc == 0; k++) {
            csp_pixels s; const char *msg = "";
            int code = cswd_batch_pixels(wb, k, &s.device_pixels, &s.width, &s.height, &s.channels, &msg);
            const uint8_t *rgba = nullptr, *plane = nullptr;
            if (!code) cswd_batch_alpha(wb, k, &rgba, &plane);
            if (code) { if (results) results[g0 + k] = make_result(code, msg); failed_total++; continue; }
            // a picture with transparency: the PNG coder and the lossless WebP coder take its RGBA; the lossy WebP encoder its RGB, the plane becomes the
            // ALPH chunk afterwards; a JPEG drops the plane (as image-rs does)
            const bool wants_rgba = plane && (target == CS_TYPE_PNG || (target == CS_TYPE_WEBP && p->webp_lossless));
            rgb3.push_back(s); rplane.push_back(wants_rgba ? plane : nullptr);
            if (wants_rgba) { s.device_pixels = rgba; s.channels = 4; }
            px.push_back(s); aplane.push_back((plane && target == CS_TYPE_WEBP && !p->webp_lossless) ? plane : nullptr); at.push_back(g0 + k);
        }
        if (rc) { for (size_t k = 0; k < n; k++) if (results) results[g0 + k] = make_result(rc, csh_last_error()); failed_total += int(n); cswd_batch_destroy(wb); continue; }
        if (!px.empty()) {
            std::vector<CByteArray> out(px.size());
            std::vector<CCSResult> res(px.size());
            int failed = -1;
            csh_batch *jb = nullptr, *rb = nullptr, *rb2 = nullptr;
            csp_batch *pb = nullptr;
            std::vector<cswd_rgba *> joined;
            // the pictures at the size asked for, for the PNG / lossless WebP coders: every picture's RGB through the Lanczos branch, the alpha planes of the
            // pictures with transparency through it as grey pictures (image-rs resamples the four channels alike), and the two halves joined again
            auto resized = [&](std::vector<csp_pixels> &src) -> int {
                int r = csh_batch_create_from_pixels_rgb(rgb3.data(), rgb3.size(), p, device, &rb);
                if (r == 0) r = csh_batch_run(rb, nullptr);
                for (size_t k = 0; k < rgb3.size() && r == 0; k++) { const char *m = ""; if (csh_batch_pixels(rb, k, &src[k].device_pixels, &src[k].width, &src[k].height, &src[k].channels, &m)) r = CS_ERR_NO_DEVICE; }
                std::vector<csp_pixels> planes;
                std::vector<size_t> whose;
                for (size_t k = 0; k < rgb3.size(); k++) if (rplane[k]) { planes.push_back(csp_pixels{rplane[k], rgb3[k].width, rgb3[k].height, 1}); whose.push_back(k); }
                if (r || planes.empty()) return r;
                r = csh_batch_create_from_pixels_rgb(planes.data(), planes.size(), p, device, &rb2);
                if (r == 0) r = csh_batch_run(rb2, nullptr);
                for (size_t j = 0; j < planes.size() && r == 0; j++) {
                    const char *m = ""; csp_pixels a;
                    if (csh_batch_pixels(rb2, j, &a.device_pixels, &a.width, &a.height, &a.channels, &m)) { r = CS_ERR_NO_DEVICE; break; }
                    csp_pixels &c = src[whose[j]];
                    if (a.width != c.width || a.height != c.height || a.channels != 1 || c.channels != 3) { r = CS_ERR_NO_DEVICE; break; }
                    cswd_rgba *jn = nullptr;
                    r = cswd_rgba_join(c.device_pixels, a.device_pixels, c.width, c.height, device, &jn, &c.device_pixels);
                    if (r == 0) { joined.push_back(jn); c.channels = 4; } else r = CS_ERR_NO_DEVICE;
                }
                return r;
            };
            if (target == CS_TYPE_PNG) {
                std::vector<csp_pixels> src = px;
                if (p->width || p->height) rc = resized(src);   // resized pixels first: the JPEG row's resize branch, stopped behind its RGB
                if (rc == 0) rc = csp_batch_create_pixels(src.data(), src.size(), p, device, &pb);
                if (rc == 0) rc = csp_batch_run(pb, nullptr);
                if (rc == 0) failed = csp_batch_fetch(pb, out.data(), res.data());
            } else if (target == CS_TYPE_WEBP && p->webp_lossless) {   // webp.lossless: the VP8L coder over the (resized) pixels
                std::vector<csp_pixels> src = px;
                if (p->width || p->height) rc = resized(src);
                if (rc == 0) failed = csl_encode_pixels(src.data(), src.size(), device, out.data(), res.data());
            } else {
                rc = target == CS_TYPE_WEBP ? csh_batch_create_webp_from_pixels(px.data(), px.size(), p, device, &jb) : csh_batch_create_from_pixels(px.data(), px.size(), p, device, &jb);
                if (rc == 0) rc = csh_batch_run(jb, nullptr);
                if (rc == 0) failed = csh_batch_fetch(jb, out.data(), res.data());
                // the alpha planes of the pictures that have one: the VP8L coder over each plane as a grey picture, then VP8X + ALPH + VP8
                std::vector<csp_pixels> apx;
                std::vector<size_t> aat;
                for (size_t k = 0; k < px.size() && rc == 0 && failed >= 0; k++)
                    if (aplane[k] && out[k].data) { apx.push_back(csp_pixels{aplane[k], px[k].width, px[k].height, 1}); aat.push_back(k); }
                if (!apx.empty() && (p->width || p->height)) {   // the planes through the same Lanczos branch as the colour, as grey pictures (image-rs resamples the four channels alike)
                    rc = csh_batch_create_from_pixels_rgb(apx.data(), apx.size(), p, device, &rb);
                    if (rc == 0) rc = csh_batch_run(rb, nullptr);
                    for (size_t j = 0; j < apx.size() && rc == 0; j++) { const char *m = ""; if (csh_batch_pixels(rb, j, &apx[j].device_pixels, &apx[j].width, &apx[j].height, &apx[j].channels, &m)) rc = CS_ERR_NO_DEVICE; }
                }
                if (!apx.empty() && rc == 0) {
                    std::vector<CByteArray> aout(apx.size());
                    std::vector<CCSResult> ares(apx.size());
                    csl_encode_pixels(apx.data(), apx.size(), device, aout.data(), ares.data());
                    for (size_t j = 0; j < apx.size(); j++) {
                        const size_t k = aat[j];
                        if (!aout[j].data || csl_attach_alpha(&out[k], &aout[j], apx[j].width, apx[j].height)) {
                            cs_free_bytes(&out[k]); cs_free_result(&res[k]);
                            res[k] = make_result(ares[j].code ? ares[j].code : CS_ERR_NO_DEVICE, "alpha plane coder failed"); failed++;
                        }
                        cs_free_bytes(&aout[j]); cs_free_result(&ares[j]);
                    }
                }
            }
            if (rc || failed < 0) { for (size_t k = 0; k < px.size(); k++) if (results) results[at[k]] = make_result(rc ? rc : CS_ERR_NO_DEVICE, csh_last_error()); failed_total += int(px.size()); }
            else {
                failed_total += failed;
                for (size_t k = 0; k < px.size(); k++) {
                    if (target == CS_TYPE_WEBP && p->keep_metadata && out[k].data) webp_carry_metadata(inputs[at[k]], out[k]);
                    outputs[at[k]] = out[k]; if (results) results[at[k]] = res[k]; else cs_free_result(&res[k]);
                }
            }
            csp_batch_destroy(pb); csh_batch_destroy(jb); csh_batch_destroy(rb); csh_batch_destroy(rb2);
            for (cswd_rgba *jn : joined) cswd_rgba_destroy(jn);
        }
        cswd_batch_destroy(wb);
    }
    return failed_total;
}

// one call, mixed inputs: PNG files go to the PNG pipeline, everything else to the JPEG pipeline (which
// answers per file for what it has no device path for); results keep the order of the inputs
// GIF and TIFF files (libcaesium: gifski / the tiff crate; SURVEY.md 2 rows 9-10: outside the north-star's JPEG / PNG / WebP scope, "CPU
// passthrough") are passed through: the file comes back as it is, with Success -- as for a PNG that oxipng cannot make smaller.  A tree
// that holds such files (configs[4], /root/reference/src/compressor.rs:774 hands the engine a .tif) then completes without error lines; the
// caller's overwrite / min-savings policy sees equal sizes and acts on that.  A resize or a conversion of such a file is still refused.
static bool passthrough(const CByteArray &in, const CCSParameters *p, CByteArray *out, CCSResult *res) {
    const int t = sniff(in.data, in.length);
    if (t != CS_TYPE_GIF && t != CS_TYPE_TIFF) return false;
    out->data = nullptr; out->length = 0;
    if (p->width || p->height) { if (res) *res = make_result(CS_ERR_UNSUPPORTED, "resizing a GIF / TIFF file has no path in this build (the file itself is passed through without a size)"); return true; }
    out->data = static_cast<uint8_t *>(malloc(in.length ? in.length : 1));
    if (!out->data) { if (res) *res = make_result(CS_ERR_NO_DEVICE, "out of memory"); return true; }
    memcpy(out->data, in.data, in.length); out->length = in.length;
    if (res) *res = make_result(0, nullptr);
    return true;
}
int cs_batch_compress(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results) {
    std::vector<size_t> png, webp, other;   // PNG files: lossless under png.optimize, else the lossy (quantising) form of the same pipeline
    int passed_failed = 0;
    bool any_pass = false;
    for (size_t i = 0; i < count; i++) {
        const int t = sniff(inputs[i].data, inputs[i].length);
        if (t == CS_TYPE_GIF || t == CS_TYPE_TIFF) {
            CCSResult r = make_result(0, nullptr);
            passthrough(inputs[i], p, &outputs[i], &r);
            if (!r.success) passed_failed++;
            if (results) results[i] = r; else cs_free_result(&r);
            any_pass = true;
            continue;
        }
        (t == CS_TYPE_PNG ? png : t == CS_TYPE_WEBP ? webp : other).push_back(i);
    }
    if (png.empty() && webp.empty() && !any_pass) return jpeg_batch_compress(inputs, count, p, device, outputs, results);
    std::atomic<int> failed{0};
    auto run = [&](const std::vector<size_t> &idx, int kind) {
        if (idx.empty()) return;
        std::vector<CByteArray> in(idx.size()), out(idx.size());
        std::vector<CCSResult> res(idx.size());
        for (size_t k = 0; k < idx.size(); k++) { in[k] = inputs[idx[k]]; res[k] = make_result(0, nullptr); out[k].data = nullptr; out[k].length = 0; }
        failed += kind == 1 ? png_batch_compress(in.data(), in.size(), p, device, out.data(), res.data())
                    : kind == 2 ? webp_inputs(in.data(), in.size(), p, CS_TYPE_WEBP, device, out.data(), res.data())
                                : jpeg_batch_compress(in.data(), in.size(), p, device, out.data(), res.data());
        for (size_t k = 0; k < idx.size(); k++) { outputs[idx[k]] = out[k]; if (results) results[idx[k]] = res[k]; else cs_free_result(&res[k]); }
    };
    // the three rows side by side (each batch keeps to its own stream; their outputs are disjoint): a tree of all three kinds waits for its slowest row only
    std::vector<std::thread> rows;
    if (!png.empty() && (!webp.empty() || !other.empty())) rows.emplace_back(run, std::cref(png), 1); else run(png, 1);
    if (!webp.empty() && !other.empty()) rows.emplace_back(run, std::cref(webp), 2); else run(webp, 2);
    run(other, 0);
    for (std::thread &t : rows) t.join();
    return failed.load() + passed_failed;
}
CCSResult cs_compress_in_memory(const uint8_t *in, size_t n, const CCSParameters *p, CByteArray *out) {
    CByteArray input; input.data = const_cast<uint8_t *>(in); input.length = n;
    CCSResult r; r.success = false; r.code = 0; r.error_message = nullptr;
    out->data = nullptr; out->length = 0;
    cs_batch_compress(&input, 1, p, 0, out, &r);
    return r;
}

// libcaesium's size-targeting (compress_to_size_in_memory; SURVEY.md 2b [UPSTREAM-RECALL], corroborated by
// /root/reference/samples/j0.JPG whose quantiser scale is the q=51 this walk reaches: 80,40,60,50,55,52,51):
// bisection on quality from 80 inside (1,101); stop when the result fits and is within 2 % of the target, or when the
// midpoint stops moving (the file of that last try is returned, fitting or not); a walk that bottoms out at q=1 still
// too large returns that smallest file only if return_smallest; more than 10 tries is an error.
// Device form: the whole group is decoded and transformed ONCE (unquantised DCT retained in HBM); every round re-quantises
// and re-codes with each file's current quality, and files leave the search as they finish.
struct SizeWalk { int quality = 80, last_less = 1, last_high = 101, tries = 0; bool done = false; };

// PNG files under --max-size: the same walk over png.quality (the quantiser's palette size follows it).  Every try is a full run of the
// PNG pipeline for the files still searching, grouped by the quality they are at (the first round is one batch, later rounds a few);
// under png.optimize the quality does not apply and one try is all there is
static int png_compress_to_size(const CByteArray *inputs, size_t count, const CCSParameters *p, size_t max_output_size, bool return_smallest, int device,
                                CByteArray *outputs, CCSResult *results, bool webp) {
    const size_t tolerance = max_output_size * 2 / 100;
    int failed_total = 0;
    std::vector<SizeWalk> walk(count);
    for (;;) {   // rounds: at most ten per file (SizeWalk::tries)
        std::map<int, std::vector<size_t>> by_quality;
        for (size_t i = 0; i < count; i++) if (!walk[i].done) by_quality[walk[i].quality].push_back(i);
        if (by_quality.empty()) break;
        for (auto &g : by_quality) {
            const std::vector<size_t> &idx = g.second;
            std::vector<CByteArray> in(idx.size()), cur(idx.size());
            std::vector<CCSResult> res(idx.size());
            for (size_t k = 0; k < idx.size(); k++) in[k] = inputs[idx[k]];
            CCSParameters q = *p;
            q.png_quality = q.webp_quality = uint32_t(g.first);
            // WebP files: the same walk over webp.quality, every try a decode + encode of the files still searching (libcaesium's webp::compress per try)
            if (webp) { for (size_t k = 0; k < idx.size(); k++) res[k] = make_result(0, nullptr); webp_inputs(in.data(), in.size(), &q, CS_TYPE_WEBP, device, cur.data(), res.data()); }
            else png_batch_compress(in.data(), in.size(), &q, device, cur.data(), res.data());
            for (size_t k = 0; k < idx.size(); k++) {
                const size_t i = idx[k];
                SizeWalk &w = walk[i];
                auto finish = [&](bool keep_file, int code, const char *msg) {
                    w.done = true;
                    if (keep_file) outputs[i] = cur[k]; else cs_free_bytes(&cur[k]);
                    cs_free_result(&results[i]);
                    if (code) { results[i] = make_result(code, msg); cs_free_result(&res[k]); failed_total++; }
                    else results[i] = res[k];
                };
                if (!res[k].success) { w.done = true; cs_free_result(&results[i]); results[i] = res[k]; failed_total++; continue; }
                if (!webp && p->png_optimize) { finish(true, 0, nullptr); continue; }
                const size_t len = cur[k].length;
                if (len <= max_output_size && max_output_size - len < tolerance) { finish(true, 0, nullptr); continue; }
                if (len <= max_output_size) w.last_less = w.quality; else w.last_high = w.quality;
                int nq = (w.last_high + w.last_less) / 2;
                nq = nq < 1 ? 1 : (nq > 100 ? 100 : nq);
                if (nq == w.quality) {
                    if (nq == 1 && w.last_high == 1 && !return_smallest) finish(false, CS_ERR_TOO_BIG, "cannot compress to the requested size");
                    else finish(true, 0, nullptr);
                    continue;
                }
                if (++w.tries >= 10) { finish(false, CS_ERR_TOO_BIG, "max tries reached while compressing to size"); continue; }
                w.quality = nq;
                cs_free_bytes(&cur[k]); cs_free_result(&res[k]);
            }
        }
    }
    return failed_total;
}

static int jpeg_batch_compress_to_size(const CByteArray *inputs, size_t count, CCSParameters *p, size_t max_output_size, bool return_smallest, int device,
                                       CByteArray *outputs, CCSResult *results);
int cs_batch_compress_to_size(const CByteArray *inputs, size_t count, CCSParameters *p, size_t max_output_size, bool return_smallest, int device,
                              CByteArray *outputs, CCSResult *results) {
    std::vector<size_t> png, webp, other;
    int failed = 0;
    bool any_pass = false;
    for (size_t i = 0; i < count; i++) {
        const int t = sniff(inputs[i].data, inputs[i].length);
        if (t == CS_TYPE_GIF || t == CS_TYPE_TIFF) {   // passed through as it is (see passthrough()): there is no quality to walk
            CCSResult r = make_result(0, nullptr);
            passthrough(inputs[i], p, &outputs[i], &r);
            // libcaesium's size walk returns its smallest try when asked to, and fails otherwise: a file that cannot be made smaller
            // and is larger than the target is a failure unless return_smallest
            if (r.success && outputs[i].length > max_output_size && !return_smallest) {
                cs_free_result(&r);
                free(outputs[i].data); outputs[i].data = nullptr; outputs[i].length = 0;
                r = make_result(CS_ERR_TOO_BIG, "the file is passed through as it is (GIF / TIFF have no device path) and is larger than the target size");
            }
            if (!r.success) failed++;
            if (results) results[i] = r; else cs_free_result(&r);
            any_pass = true;
            continue;
        }
        (t == CS_TYPE_PNG ? png : t == CS_TYPE_WEBP ? webp : other).push_back(i);
    }
    if (png.empty() && webp.empty() && !any_pass) return jpeg_batch_compress_to_size(inputs, count, p, max_output_size, return_smallest, device, outputs, results);
    auto run = [&](const std::vector<size_t> &idx, int kind) {
        if (idx.empty()) return;
        std::vector<CByteArray> in(idx.size()), out(idx.size());
        std::vector<CCSResult> res(idx.size());
        for (size_t k = 0; k < idx.size(); k++) { in[k] = inputs[idx[k]]; out[k].data = nullptr; out[k].length = 0; res[k] = make_result(0, nullptr); }
        CCSParameters q = *p;
        failed += kind ? png_compress_to_size(in.data(), in.size(), &q, max_output_size, return_smallest, device, out.data(), res.data(), kind == 2)
                       : jpeg_batch_compress_to_size(in.data(), in.size(), &q, max_output_size, return_smallest, device, out.data(), res.data());
        if (!kind) *p = q;   // the JPEG walk leaves its last quality in the caller's parameters, as the reference's &mut does
        for (size_t k = 0; k < idx.size(); k++) { outputs[idx[k]] = out[k]; results[idx[k]] = res[k]; }
    };
    run(other, 0);
    run(png, 1);
    run(webp, 2);
    return failed;
}
static int jpeg_batch_compress_to_size(const CByteArray *inputs, size_t count, CCSParameters *p, size_t max_output_size, bool return_smallest, int device,
                                       CByteArray *outputs, CCSResult *results) {
    for (size_t i = 0; i < count; i++) { outputs[i].data = nullptr; outputs[i].length = 0; results[i] = make_result(0, nullptr); }
    const size_t tolerance = max_output_size * 2 / 100;
    int failed_total = 0;
    for (size_t g0 = 0, n = 0; g0 < count; g0 += n) {
        n = cs_batch_extent(inputs + g0, count - g0);
        p->jpeg_quality = p->png_quality = p->webp_quality = 80;
        csh_batch *b = nullptr;
        int rc = csh_batch_create(inputs + g0, n, p, device, &b);
        if (rc == 0 && !p->jpeg_optimize) rc = csh_batch_retain_dct(b, 1) ? CS_ERR_NO_DEVICE : 0;
        std::vector<SizeWalk> walk(n);
        std::vector<uint32_t> q(n, 0);
        std::vector<CByteArray> cur(n);
        std::vector<CCSResult> res(n);
        for (int round = 0; rc == 0; round++) {
            rc = round == 0 ? csh_batch_run(b, nullptr) : csh_batch_rerun_encode(b, nullptr);
            if (rc) break;
            if (csh_batch_fetch(b, cur.data(), res.data()) < 0) { rc = CS_ERR_NO_DEVICE; break; }
            bool any = false;
            for (size_t i = 0; i < n; i++) {
                SizeWalk &w = walk[i];
                q[i] = 0;
                if (w.done) { cs_free_bytes(&cur[i]); cs_free_result(&res[i]); continue; }
                auto finish = [&](bool keep_file, int code, const char *msg) {
                    w.done = true;
                    if (keep_file) outputs[g0 + i] = cur[i]; else cs_free_bytes(&cur[i]);
                    cs_free_result(&results[g0 + i]);
                    if (code) { results[g0 + i] = make_result(code, msg); cs_free_result(&res[i]); failed_total++; }
                    else results[g0 + i] = res[i];
                };
                if (!res[i].success) { w.done = true; results[g0 + i] = res[i]; failed_total++; continue; }
                if (p->jpeg_optimize) { finish(true, 0, nullptr); continue; }   // lossless: quality does not apply, one try
                const size_t len = cur[i].length;
                if (len <= max_output_size && max_output_size - len < tolerance) { finish(true, 0, nullptr); continue; }
                if (len <= max_output_size) w.last_less = w.quality; else w.last_high = w.quality;
                int nq = (w.last_high + w.last_less) / 2;
                nq = nq < 1 ? 1 : (nq > 100 ? 100 : nq);
                if (nq == w.quality) {
                    if (nq == 1 && w.last_high == 1 && !return_smallest) finish(false, CS_ERR_TOO_BIG, "cannot compress to the requested size");
                    else finish(true, 0, nullptr);
                    continue;
                }
                if (++w.tries >= 10) { finish(false, CS_ERR_TOO_BIG, "max tries reached while compressing to size"); continue; }
                w.quality = nq;
                q[i] = uint32_t(nq);
                any = true;
                cs_free_bytes(&cur[i]); cs_free_result(&res[i]);
            }
            if (!any) break;
            if (csh_batch_set_quality(b, q.data())) { rc = CS_ERR_NO_DEVICE; break; }
        }
        if (rc)
            for (size_t i = 0; i < n; i++)
                if (!walk[i].done) { cs_free_result(&results[g0 + i]); results[g0 + i] = make_result(rc, csh_last_error()); failed_total++; }
        // the reference leaves the quality of the last try in its &mut CSParameters; with many files that is per file, so a
        // batch reports the first file's
        if (g0 == 0 && n) p->jpeg_quality = p->png_quality = p->webp_quality = uint32_t(walk[0].quality);
        csh_batch_destroy(b);
    }
    return failed_total;
}

CCSResult cs_compress_to_size_in_memory(const uint8_t *in, size_t n, CCSParameters *p, size_t max_output_size, bool return_smallest, CByteArray *out) {
    CByteArray input; input.data = const_cast<uint8_t *>(in); input.length = n;
    CCSResult r = make_result(0, nullptr);
    out->data = nullptr; out->length = 0;
    cs_batch_compress_to_size(&input, 1, p, max_output_size, return_smallest, 0, out, &r);
    return r;
}

static int sniff(const uint8_t *d, size_t n) {
    if (n >= 3 && d[0] == 0xFF && d[1] == 0xD8 && d[2] == 0xFF) return CS_TYPE_JPEG;
    if (n >= 8 && !memcmp(d, "\x89PNG\r\n\x1a\n", 8)) return CS_TYPE_PNG;
    if (n >= 12 && !memcmp(d, "RIFF", 4) && !memcmp(d + 8, "WEBP", 4)) return CS_TYPE_WEBP;
    if (n >= 6 && (!memcmp(d, "GIF87a", 6) || !memcmp(d, "GIF89a", 6))) return CS_TYPE_GIF;
    if (n >= 4 && (!memcmp(d, "II*\0", 4) || !memcmp(d, "MM\0*", 4))) return CS_TYPE_TIFF;
    return CS_TYPE_UNKN;
}

static uint32_t rd_be32(const uint8_t *d) { return (uint32_t(d[0]) << 24) | (uint32_t(d[1]) << 16) | (uint32_t(d[2]) << 8) | d[3]; }

// JPEG -> PNG: the JPEG path's decode and resize leave the pixels in device memory, the PNG coder takes them from there (lossless under
// png.optimize, quantising otherwise -- what png::compress does to the intermediate file libcaesium makes)
static int jpeg_to_png(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results, bool lossless_webp = false) {
    int failed_total = 0;
    for (size_t g0 = 0, n = 0; g0 < count; g0 += n) {
        // the PNG coder keeps about 14 bytes per sample in flight: groups of at most 512 files and 400 MB of JPEG (some 5 GB of pixels)
        uint64_t bytes = 0;
        for (n = 0; g0 + n < count && n < 512 && (!n || bytes + inputs[g0 + n].length <= (uint64_t(400) << 20)); n++) bytes += inputs[g0 + n].length;
        for (size_t k = 0; k < n; k++) { outputs[g0 + k].data = nullptr; outputs[g0 + k].length = 0; }
        csh_batch *jb = nullptr;
        int rc = csh_batch_create_pixels(inputs + g0, n, p, device, &jb);
        if (rc == 0) rc = csh_batch_run(jb, nullptr);
        std::vector<csp_pixels> px;
        std::vector<size_t> at;
        for (size_t k = 0; k < n && rc == 0; k++) {
            csp_pixels s; const char *msg = "";
            const int code = csh_batch_pixels(jb, k, &s.device_pixels, &s.width, &s.height, &s.channels, &msg);
            if (code) { results[g0 + k] = make_result(code, msg); failed_total++; } else { px.push_back(s); at.push_back(g0 + k); }
        }
        csp_batch *pb = nullptr;
        if (rc == 0 && !px.empty() && lossless_webp) {   // the same decoded (and resized) pixels into the VP8L coder
            std::vector<CByteArray> out(px.size());
            std::vector<CCSResult> res(px.size());
            failed_total += csl_encode_pixels(px.data(), px.size(), device, out.data(), res.data());
            for (size_t k = 0; k < px.size(); k++) { outputs[at[k]] = out[k]; results[at[k]] = res[k]; }
        } else if (rc == 0 && !px.empty()) {
            rc = csp_batch_create_pixels(px.data(), px.size(), p, device, &pb);
            if (rc == 0) rc = csp_batch_run(pb, nullptr);
            if (rc == 0) {
                std::vector<CByteArray> out(px.size());
                std::vector<CCSResult> res(px.size());
                const int failed = csp_batch_fetch(pb, out.data(), res.data());
                if (failed < 0) rc = CS_ERR_NO_DEVICE;
                else {
                    failed_total += failed;
                    for (size_t k = 0; k < px.size(); k++) { outputs[at[k]] = out[k]; results[at[k]] = res[k]; }
                }
            }
            if (rc) { for (size_t k = 0; k < px.size(); k++) results[at[k]] = make_result(rc, csh_last_error()); failed_total += int(px.size()); rc = 0; }
        } else if (rc) {
            for (size_t k = 0; k < n; k++) results[g0 + k] = make_result(rc, csh_last_error());
            failed_total += int(n);
        }
        csp_batch_destroy(pb);
        csh_batch_destroy(jb);
    }
    return failed_total;
}

// convert: JPEG -> WebP (the JPEG path's decode and resize, then the VP8 encoder), opaque PNG -> WebP (the PNG path's decode, then
// the same encoder), JPEG -> PNG and PNG -> JPEG (csp_png_to_jpeg) run on the device; every other pair of formats has no device path
int cs_batch_convert(const CByteArray *inputs, size_t count, const CCSParameters *p, uint32_t format, int device, CByteArray *outputs, CCSResult *results) {
    int failed_total = 0;
    std::vector<size_t> ok, okpng, topng, tojpeg_, fromwebp, tolossless, pnglossless;
    for (size_t i = 0; i < count; i++) {
        outputs[i].data = nullptr; outputs[i].length = 0;
        const int src = sniff(inputs[i].data, inputs[i].length);
        int code = 0; const char *msg = nullptr;
        if (src == CS_TYPE_UNKN) { code = CS_ERR_UNKNOWN_TYPE; msg = "unknown file type"; }
        else if (uint32_t(src) == format) { code = CS_ERR_SAME_FORMAT; msg = "cannot convert to the same format"; }
        else if (src == CS_TYPE_JPEG && format == CS_TYPE_PNG) { topng.push_back(i); continue; }
        else if (src == CS_TYPE_PNG && format == CS_TYPE_JPEG) { tojpeg_.push_back(i); continue; }
        else if (src == CS_TYPE_WEBP && (format == CS_TYPE_JPEG || format == CS_TYPE_PNG)) { fromwebp.push_back(i); continue; }
        else if ((src != CS_TYPE_JPEG && src != CS_TYPE_PNG) || format != CS_TYPE_WEBP) { code = CS_ERR_UNSUPPORTED; msg = "this format conversion has no device path in this build (built: JPEG / PNG -> WebP, JPEG <-> PNG, WebP -> JPEG / PNG)"; }
        else if (p->webp_lossless && src == CS_TYPE_JPEG) { tolossless.push_back(i); continue; }
        else if (p->webp_lossless) { pnglossless.push_back(i); continue; }
        if (code) { if (results) results[i] = make_result(code, msg); failed_total++; } else (src == CS_TYPE_PNG ? okpng : ok).push_back(i);
    }
    std::vector<CByteArray> ok_in(ok.size());
    for (size_t k = 0; k < ok.size(); k++) ok_in[k] = inputs[ok[k]];
    for (size_t g0 = 0, n = 0; g0 < ok.size(); g0 += n) {
        n = cs_batch_extent(ok_in.data() + g0, std::min<size_t>(ok.size() - g0, 1024));
        std::vector<CByteArray> in(ok_in.begin() + g0, ok_in.begin() + g0 + n), out(n);
        std::vector<CCSResult> res(n);
        csh_batch *b = nullptr;
        int rc = csh_batch_create_webp(in.data(), n, p, device, &b);
        if (rc == 0) rc = csh_batch_run(b, nullptr);
        int failed = rc ? int(n) : csh_batch_fetch(b, out.data(), res.data());
        for (size_t k = 0; k < n; k++) {
            if (rc || failed < 0) { if (results) results[ok[g0 + k]] = make_result(rc ? rc : CS_ERR_NO_DEVICE, csh_last_error()); }
            else { outputs[ok[g0 + k]] = out[k]; if (results) results[ok[g0 + k]] = res[k]; else cs_free_result(&res[k]); }
        }
        csh_batch_destroy(b);
        failed_total += failed < 0 ? int(n) : failed;
    }
    if (!fromwebp.empty()) {
        const size_t n = fromwebp.size();
        std::vector<CByteArray> in(n), out(n);
        std::vector<CCSResult> res(n);
        for (size_t k = 0; k < n; k++) { in[k] = inputs[fromwebp[k]]; res[k] = make_result(0, nullptr); }
        failed_total += webp_inputs(in.data(), n, p, format, device, out.data(), res.data());
        for (size_t k = 0; k < n; k++) { outputs[fromwebp[k]] = out[k]; if (results) results[fromwebp[k]] = res[k]; else cs_free_result(&res[k]); }
    }
    if (!tolossless.empty()) {
        const size_t n = tolossless.size();
        std::vector<CByteArray> in(n), out(n);
        std::vector<CCSResult> res(n);
        for (size_t k = 0; k < n; k++) { in[k] = inputs[tolossless[k]]; res[k] = make_result(0, nullptr); }
        failed_total += jpeg_to_png(in.data(), n, p, device, out.data(), res.data(), true);
        for (size_t k = 0; k < n; k++) { outputs[tolossless[k]] = out[k]; if (results) results[tolossless[k]] = res[k]; else cs_free_result(&res[k]); }
    }
    if (!topng.empty()) {
        const size_t n = topng.size();
        std::vector<CByteArray> in(n), out(n);
        std::vector<CCSResult> res(n);
        for (size_t k = 0; k < n; k++) in[k] = inputs[topng[k]];
        failed_total += jpeg_to_png(in.data(), n, p, device, out.data(), res.data());
        for (size_t k = 0; k < n; k++) { outputs[topng[k]] = out[k]; if (results) results[topng[k]] = res[k]; else cs_free_result(&res[k]); }
    }
    for (int pass = 0; pass < 2; pass++) {
    const std::vector<size_t> &tojpeg = pass ? pnglossless : tojpeg_;   // PNG -> lossless WebP: the same decode, the VP8L coder behind it
    for (size_t g0 = 0, n = 0; g0 < tojpeg.size(); g0 += n) {   // groups of at most 256 files and 96 GB of decode buffers (8 bytes per pixel at most, three regions)
        uint64_t bytes = 0;
        for (n = 0; g0 + n < tojpeg.size() && n < 256; n++) {
            const CByteArray &f = inputs[tojpeg[g0 + n]];
            uint64_t est = 1 << 20;
            if (f.length >= 33) est += (uint64_t(rd_be32(f.data + 16)) * rd_be32(f.data + 20)) * 24;
            if (n && bytes + est > (uint64_t(96) << 30)) break;
            bytes += est;
        }
        std::vector<CByteArray> in(n), out(n);
        std::vector<CCSResult> res(n);
        for (size_t k = 0; k < n; k++) in[k] = inputs[tojpeg[g0 + k]];
        failed_total += pass ? csp_png_to_lossless_webp(in.data(), n, p, device, out.data(), res.data()) : csp_png_to_jpeg(in.data(), n, p, device, out.data(), res.data());
        for (size_t k = 0; k < n; k++) { outputs[tojpeg[g0 + k]] = out[k]; if (results) results[tojpeg[g0 + k]] = res[k]; else cs_free_result(&res[k]); }
    }
    }
    if (!okpng.empty()) {
        const size_t n = okpng.size();
        std::vector<CByteArray> in(n), out(n);
        std::vector<CCSResult> res(n);
        for (size_t k = 0; k < n; k++) in[k] = inputs[okpng[k]];
        failed_total += png_batch_compress(in.data(), n, p, device, out.data(), res.data(), true);
        for (size_t k = 0; k < n; k++) { outputs[okpng[k]] = out[k]; if (results) results[okpng[k]] = res[k]; else cs_free_result(&res[k]); }
    }
    return failed_total;
}
CCSResult cs_convert_in_memory(const uint8_t *in, size_t n, const CCSParameters *p, uint32_t format, CByteArray *out) {
    CByteArray input; input.data = const_cast<uint8_t *>(in); input.length = n;
    CCSResult r; r.success = false; r.code = 0; r.error_message = nullptr;
    out->data = nullptr; out->length = 0;
    cs_batch_convert(&input, 1, p, format, 0, out, &r);
    return r;
}

void cs_free_bytes(CByteArray *b) { if (b && b->data) { free(b->data); b->data = nullptr; b->length = 0; } }
void cs_free_result(CCSResult *r) { if (r && r->error_message) { free(const_cast<char *>(r->error_message)); r->error_message = nullptr; } }

}  // extern "C"
