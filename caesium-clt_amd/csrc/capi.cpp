// capi.cpp -- the libcaesium-shaped entry points on top of the device batch queue.
// Reference semantics: /root/reference/src/compressor.rs:287-306 (call shapes), :411-446 (parameters).
#include <cstdlib>
#include <cstring>
#include <string>
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
// of 2048 1080p files already occupies ~40 GB of HBM and tens of thousands of workgroups per launch
enum { CS_GROUP = 2048 };
int cs_batch_compress(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results) {
    for (size_t i = 0; i < count; i++) { outputs[i].data = nullptr; outputs[i].length = 0; }
    int failed_total = 0;
    for (size_t g0 = 0; g0 < count; g0 += CS_GROUP) {
        size_t n = count - g0 < size_t(CS_GROUP) ? count - g0 : size_t(CS_GROUP);
        csh_batch *b = nullptr;
        int rc = csh_batch_create(inputs + g0, n, p, device, &b);
        if (rc == 0) rc = csh_batch_run(b, nullptr);
        if (rc != 0) {
            for (size_t i = 0; i < n; i++) if (results) results[g0 + i] = make_result(rc, csh_last_error());
            csh_batch_destroy(b);
            failed_total += int(n);
            continue;
        }
        int failed = csh_batch_fetch(b, outputs + g0, results ? results + g0 : nullptr);
        csh_batch_destroy(b);
        failed_total += failed < 0 ? int(n) : failed;
    }
    return failed_total;
}

CCSResult cs_compress_in_memory(const uint8_t *in, size_t n, const CCSParameters *p, CByteArray *out) {
    CByteArray input; input.data = const_cast<uint8_t *>(in); input.length = n;
    CCSResult r; r.success = false; r.code = 0; r.error_message = nullptr;
    out->data = nullptr; out->length = 0;
    cs_batch_compress(&input, 1, p, 0, out, &r);
    return r;
}

// libcaesium's size-targeting: bisection on quality, start 80, bounds (1,101), tolerance 2 % of the target,
// at most 10 tries, return the smallest attempt when the target is unreachable and return_smallest is set
// (call sites compressor.rs:295,298 always pass true).
CCSResult cs_compress_to_size_in_memory(const uint8_t *in, size_t n, CCSParameters *p, size_t max_output_size, bool return_smallest, CByteArray *out) {
    out->data = nullptr; out->length = 0;
    const size_t tolerance = max_output_size * 2 / 100;
    int lo = 1, hi = 101, q = 80;
    CByteArray best = {nullptr, 0};    // largest result that fits
    CByteArray smallest = {nullptr, 0};
    CCSResult last = make_result(0, nullptr);
    for (int tries = 0; tries < 10; tries++) {
        p->jpeg_quality = p->png_quality = p->webp_quality = uint32_t(q);
        CByteArray cur = {nullptr, 0};
        cs_free_result(&last);
        last = cs_compress_in_memory(in, n, p, &cur);
        if (!last.success) { cs_free_bytes(&best); cs_free_bytes(&smallest); return last; }
        if (!smallest.data || cur.length < smallest.length) {
            cs_free_bytes(&smallest);
            smallest.data = (uint8_t *)malloc(cur.length ? cur.length : 1); memcpy(smallest.data, cur.data, cur.length); smallest.length = cur.length;
        }
        if (cur.length <= max_output_size) {
            if (!best.data || cur.length > best.length) { cs_free_bytes(&best); best = cur; cur.data = nullptr; }
            if (max_output_size - best.length < tolerance) { cs_free_bytes(&cur); break; }
            lo = q;
        } else hi = q;
        cs_free_bytes(&cur);
        int nq = (lo + hi) / 2;
        if (nq == q) break;
        q = nq;
    }
    if (best.data) { *out = best; cs_free_bytes(&smallest); return last; }
    if (return_smallest && smallest.data) { *out = smallest; return last; }
    cs_free_bytes(&smallest);
    cs_free_result(&last);
    return make_result(CS_ERR_TOO_BIG, "cannot compress to the requested size");
}

static int sniff(const uint8_t *d, size_t n) {
    if (n >= 3 && d[0] == 0xFF && d[1] == 0xD8 && d[2] == 0xFF) return CS_TYPE_JPEG;
    if (n >= 8 && !memcmp(d, "\x89PNG\r\n\x1a\n", 8)) return CS_TYPE_PNG;
    if (n >= 12 && !memcmp(d, "RIFF", 4) && !memcmp(d + 8, "WEBP", 4)) return CS_TYPE_WEBP;
    if (n >= 6 && (!memcmp(d, "GIF87a", 6) || !memcmp(d, "GIF89a", 6))) return CS_TYPE_GIF;
    if (n >= 4 && (!memcmp(d, "II*\0", 4) || !memcmp(d, "MM\0*", 4))) return CS_TYPE_TIFF;
    return CS_TYPE_UNKN;
}

CCSResult cs_convert_in_memory(const uint8_t *in, size_t n, const CCSParameters *p, uint32_t format, CByteArray *out) {
    (void)p;
    out->data = nullptr; out->length = 0;
    int src = sniff(in, n);
    if (src == CS_TYPE_UNKN) return make_result(CS_ERR_UNKNOWN_TYPE, "unknown file type");
    if (uint32_t(src) == format) return make_result(CS_ERR_SAME_FORMAT, "cannot convert to the same format");
    return make_result(CS_ERR_UNSUPPORTED, "format conversion has no device path in this build");
}

void cs_free_bytes(CByteArray *b) { if (b && b->data) { free(b->data); b->data = nullptr; b->length = 0; } }
void cs_free_result(CCSResult *r) { if (r && r->error_message) { free(const_cast<char *>(r->error_message)); r->error_message = nullptr; } }

}  // extern "C"
