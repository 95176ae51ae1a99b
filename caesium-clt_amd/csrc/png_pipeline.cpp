// png_pipeline.cpp -- the device batch queue of the lossless PNG row: what `caesium::compress_in_memory` does for a PNG
// with png.optimize set (/root/reference/src/compressor.rs:305, parameters :411-446) -- oxipng's decode, row-filter
// trials and DEFLATE -- for a whole group of files at once.  Host work is container logic only: the chunk walk, the
// carried-chunk policy, descriptors.  Statement of every stage: oracle/png_oracle.c.
#include <cstdarg>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/caesium_hip.h"
#include "../../include/png_quality_table.h"
#include "devmem.hpp"
#include "png_kernels.h"
#include "png_parse.h"
#include "webp_kernels.h"
#include "../../include/vp8_tables.h"
#include "resize_host.h"

using namespace csp;
using csh::DevBuf;
using csh::PinnedBytes;

namespace {

uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
void put_be32(uint8_t *p, uint32_t v) { p[0] = uint8_t(v >> 24); p[1] = uint8_t(v >> 16); p[2] = uint8_t(v >> 8); p[3] = uint8_t(v); }
uint32_t crc32_host(const uint8_t *p, size_t n) {
    struct Table {   // built once, thread-safe (the boundary is called from several host threads)
        uint32_t t[256];
        Table() { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t[i] = c; } }
    };
    static const Table table;
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) crc = table.t[(crc ^ p[i]) & 255] ^ (crc >> 8);
    return ~crc;
}
const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};

struct PngItem {
    int code = 0;
    std::string msg;
    int image = -1;
    size_t file_size = 0;
    uint32_t width = 0, height = 0, rowbytes = 0, bpp = 0, channels = 0, depth = 0, ctype = 0;
    bool no_reduce = false;   // a carried chunk is tied to the colour type (tRNS, bKGD, sBIT)
    bool pal_tied = false;    // a carried chunk counts on the palette as it is (bKGD, sBIT, hIST): an indexed image keeps its depth
    bool interlace = false;   // Adam7 input (the output never is)
    bool has_plte = false, has_trns = false;
    std::vector<uint8_t> plte, trns;                // PLTE / tRNS payloads (conversion to WebP and the resize read them)
    std::vector<std::pair<size_t, size_t>> idat;   // (offset, length) of every IDAT payload in the input
    size_t idat_len = 0;
    std::vector<uint8_t> prefix, suffix;            // output bytes in front of / behind the IDAT chunk
};

// oxipng StripChunks::Safe keeps these ancillary chunks [UPSTREAM-RECALL]; tRNS is image data
bool kept_when_stripping(const uint8_t *type) {
    static const char *keep[] = {"cICP", "iCCP", "sRGB", "pHYs", "tRNS"};
    for (const char *k : keep) if (!memcmp(type, k, 4)) return true;
    return false;
}

// the chunk walk (oracle: cso_png_decode, first half)
void parse_png(const uint8_t *in, size_t n, bool keep_metadata, PngItem &it) {
    auto fail = [&](int code, const char *msg) { it.code = code; it.msg = msg; };
    it.file_size = n;
    if (n < 8 + 25 || memcmp(in, kSig, 8)) return fail(CS_ERR_BAD_PNG, "not a PNG");
    size_t pos = 8;
    bool seen_ihdr = false, seen_idat = false, seen_iend = false;
    int depth = 0, ctype = 0, nplte = 0;
    while (pos + 12 <= n && !seen_iend) {
        const uint32_t len = be32(in + pos);
        const uint8_t *type = in + pos + 4, *d = in + pos + 8;
        if (len > 0x7FFFFFFFu || pos + 12 + size_t(len) > n) return fail(CS_ERR_BAD_PNG, "truncated PNG chunk");
        if (!seen_ihdr) {
            if (memcmp(type, "IHDR", 4) || len != 13) return fail(CS_ERR_BAD_PNG, "PNG does not start with IHDR");
            if (crc32_host(type, 17) != be32(d + 13)) return fail(CS_ERR_BAD_PNG, "IHDR checksum");
            it.width = be32(d); it.height = be32(d + 4); depth = d[8]; ctype = d[9];
            if (!it.width || !it.height || it.width > 0x7FFFFFFFu || it.height > 0x7FFFFFFFu || d[10] || d[11] || d[12] > 1) return fail(CS_ERR_BAD_PNG, "bad IHDR");
            static const int chans[7] = {1, 0, 3, 1, 2, 0, 4};
            bool ok = false;
            switch (ctype) {
            case 0: ok = depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16; break;
            case 3: ok = depth == 1 || depth == 2 || depth == 4 || depth == 8; break;
            case 2: case 4: case 6: ok = depth == 8 || depth == 16; break;
            }
            if (!ok) return fail(CS_ERR_BAD_PNG, "bad colour type / bit depth");
            it.interlace = d[12] != 0;
            const uint64_t bits = uint64_t(chans[ctype]) * uint64_t(depth);
            it.bpp = bits >= 8 ? uint32_t(bits / 8) : 1u;
            it.channels = uint32_t(chans[ctype]); it.depth = uint32_t(depth); it.ctype = uint32_t(ctype);
            const uint64_t rb = (uint64_t(it.width) * bits + 7) / 8;
            if (rb > 0x7FFFFFF0u) return fail(CS_ERR_UNSUPPORTED, "PNG row too long");
            it.rowbytes = uint32_t(rb);
            it.prefix.assign(kSig, kSig + 8);
            it.prefix.insert(it.prefix.end(), in + pos, in + pos + 25);
            it.prefix[8 + 8 + 12] = 0;   // interlace method of the output
            put_be32(&it.prefix[8 + 8 + 13], crc32_host(&it.prefix[8 + 4], 17));
            seen_ihdr = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            it.idat.emplace_back(pos + 8, size_t(len)); it.idat_len += len; seen_idat = true;
        } else if (!memcmp(type, "IEND", 4)) {
            seen_iend = true;
        } else {
            if (!memcmp(type, "acTL", 4)) return fail(CS_ERR_UNSUPPORTED, "animated PNG has no device path in this build");
            if (!memcmp(type, "PLTE", 4)) { if (len % 3 || len > 768) return fail(CS_ERR_BAD_PNG, "bad PLTE"); nplte = int(len / 3); it.has_plte = true; it.plte.assign(d, d + len); }
            if (!memcmp(type, "tRNS", 4)) { it.has_trns = true; it.trns.assign(d, d + len); }
            const bool critical = !(type[0] & 0x20);
            if (critical || keep_metadata || kept_when_stripping(type)) {
                if (!memcmp(type, "tRNS", 4) || !memcmp(type, "bKGD", 4) || !memcmp(type, "sBIT", 4)) it.no_reduce = true;
                if (!memcmp(type, "bKGD", 4) || !memcmp(type, "sBIT", 4) || !memcmp(type, "hIST", 4)) it.pal_tied = true;
                std::vector<uint8_t> &dst = seen_idat ? it.suffix : it.prefix;
                dst.insert(dst.end(), in + pos, in + pos + 12 + size_t(len));
            }
        }
        pos += 12 + size_t(len);
    }
    if (!seen_ihdr || !seen_idat || !seen_iend) return fail(CS_ERR_BAD_PNG, "PNG without IHDR / IDAT / IEND");
    if (ctype == 3 && !nplte) return fail(CS_ERR_BAD_PNG, "palette PNG without PLTE");
    static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
    it.suffix.insert(it.suffix.end(), iend, iend + 12);
    // the two zlib header bytes (oracle: cso_inflate_zlib)
    uint8_t z[2]; size_t got = 0;
    for (auto &r : it.idat) for (size_t k = 0; k < r.second && got < 2; k++) z[got++] = in[r.first + k];
    if (got < 2 || (z[0] & 15) != 8 || (z[0] >> 4) > 7 || ((unsigned(z[0]) << 8) | z[1]) % 31 || (z[1] & 0x20)) return fail(CS_ERR_BAD_PNG, "bad zlib header in IDAT");
    if (it.idat_len > 0xFFFFFFF0u) return fail(CS_ERR_UNSUPPORTED, "IDAT stream too long");
}

// oxipng's presets [UPSTREAM-RECALL]: filters tried per --png-opt-level (oracle: cso_png_trials)
int trial_set(int level, int *set) {
    static const int s01[] = {5}, s2[] = {0, 1, 6, 7}, s34[] = {0, 7, 8, 9}, s5[] = {0, 1, 2, 5, 6, 7, 8, 9}, s6[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
    const int *s; int n;
    if (level <= 1) { s = s01; n = 1; } else if (level == 2) { s = s2; n = 4; } else if (level <= 4) { s = s34; n = 4; } else if (level == 5) { s = s5; n = 8; } else { s = s6; n = 10; }
    memcpy(set, s, sizeof(int) * size_t(n));
    return n;
}

const char *kPngKernelNames[CSP_NKERNELS] = {"k_png_inflate", "k_png_unfilter", "k_png_reduce", "k_png_filter5", "k_png_scores", "k_png_brute", "k_png_pick",
                                             "k_png_hist", "k_png_codes", "k_png_choose", "k_png_deep", "k_png_emit", "k_png_finish", "", "", ""};
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct csp_batch {
    int device = 0;
    hipStream_t stream{};
    bool have_stream = false;
    std::vector<PngItem> items;
    std::vector<const uint8_t *> inputs;
    std::vector<PngImg> imgs;
    std::vector<uint8_t> fixed;
    std::vector<uint32_t> flags0;   // reductions each image's format allows
    std::vector<uint32_t> cand0;    // channels of an image that may become indexed (8- or 16-bit truecolour, no PLTE, nothing tied to the colour type), else 0
    std::vector<PngPass> passes;    // reconstruction jobs: one per image, seven per Adam7 image
    std::vector<PngAdam7> adam7;
    uint64_t adam7_items = 0;
    bool reduced = false;
    bool from_pixels = false;       // csp_batch_create_pixels: no file to decode
    bool decode_only = false;       // the front half of a resize: stop at the pixels (decoded_image)
    bool to_webp = false;           // csp_batch_create_webp: the decoded pixels go to the VP8 encoder
    int webp_quality = 0;
    uint32_t webp_mb_bytes = 768, wmax_luma = 0, wmax_mbh = 0, rgb_max_h = 0;
    uint64_t wwork_bytes = 0, wlevels = 0, rgb_bytes = 0;
    std::vector<csw::WebpImg> wimgs;
    std::vector<RgbJob> rgbjobs;
    std::vector<uint8_t> plte;
    std::vector<uint32_t> h_wstatus;
    std::vector<uint8_t> walpha;    // per image: 0, or the samples per pixel (2 / 4) of a picture whose last sample is alpha
    DevBuf<csw::WebpImg> d_wimgs;
    DevBuf<RgbJob> d_rgbjobs;
    DevBuf<uint8_t> d_plte, d_rgb, d_wwork, d_wscratch, d_wprobs, d_wupdate;
    DevBuf<int16_t> d_wlevels;
    DevBuf<uint32_t> d_wstats, d_wpart, d_wstatus;
    int png_quality = 80;
    bool lossy = false;             // png.optimize not set: truecolour images with more than 256 colours are quantised (oracle: quantize)
    uint32_t n_reduced = 0;
    PngPlan plan{};
    int slot_of_strategy[10];
    uint32_t total_rows = 0, total_chunks = 0, total_groups = 0, max_pieces = 0;
    uint64_t raw_total = 0, pixels = 0;
    DevBuf<PngImg> d_imgs;
    DevBuf<uint8_t> d_idat, d_work, d_streams, d_out, d_fixed, d_choice;   // d_work: inflated streams, then pixels (one buffer: a reduction swaps the two regions of an image)
    DevBuf<ReduceJob> d_jobs;
    DevBuf<PngPass> d_passes;
    DevBuf<PaletteJob> d_pjobs;
    DevBuf<unsigned long long> d_keys;
    DevBuf<uint16_t> d_slot_index;
    DevBuf<uint32_t> d_counts, d_cand, d_qbins, d_qn, d_qpal;
    DevBuf<QuantJob> d_qjobs;
    DevBuf<QBin> d_qlist;
    DevBuf<PngAdam7> d_adam7;
    DevBuf<uint32_t> d_flags;
    DevBuf<uint32_t> d_row_image, d_chunk_image, d_chunk_first, d_group_image, d_group_first, d_status, d_nmatch, d_file_len, d_adler, d_crc;
    DevBuf<uint64_t> d_scores, d_trial_bytes;
    DevBuf<int32_t> d_winner;
    DevBuf<uint8_t> d_trial_live;
    DevBuf<PngChunk> d_chunks;
    DevBuf<uint8_t> d_deep;         // the min-cost-path kernels' scratch areas (png_parse.h)
    uint32_t deep_slots = 0;
    DevBuf<uint32_t> d_deep_queue, d_deep_list;
    int deep_iters = CSP_DEEP_ITERS;   // png.force_zopfli: CSP_DEEP_ITERS_ZOPFLI
    hipEvent_t ev[CSP_NKERNELS + 1]{};
    bool have_events = false, ran = false;
    ~csp_batch() {
        if (have_events) for (auto &e : ev) (void)hipEventDestroy(e);
        if (have_stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }   // nothing queued may outlive the device blocks
    }
};

extern "C" const char *csp_kernel_name(int i) { return (i >= 0 && i < CSP_NKERNELS) ? kPngKernelNames[i] : ""; }
extern "C" void csp_batch_destroy(csp_batch *b) { delete b; }

// (re)build the per-chunk / per-group index arrays from the current geometry of the images
static int upload_chunk_index(csp_batch *b) {
    const int nimg = int(b->imgs.size());
    std::vector<uint32_t> chunk_image, chunk_first(size_t(nimg) + 1), group_image, group_first(size_t(nimg) + 1);
    for (int i = 0; i < nimg; i++) {
        chunk_first[i] = uint32_t(chunk_image.size()); group_first[i] = uint32_t(group_image.size());
        for (uint32_t k = 0; k < b->imgs[i].nchunks; k++) chunk_image.push_back(uint32_t(i));
        for (uint32_t g = 0; g < (b->imgs[i].nchunks + CSP_GROUP - 1) / CSP_GROUP; g++) group_image.push_back(uint32_t(i));
    }
    chunk_first[nimg] = uint32_t(chunk_image.size()); group_first[nimg] = uint32_t(group_image.size());
    b->total_chunks = uint32_t(chunk_image.size()); b->total_groups = uint32_t(group_image.size());
    chunk_image.push_back(0); group_image.push_back(0);
    if (b->d_chunk_image.upload(chunk_image, b->stream) || b->d_chunk_first.upload(chunk_first, b->stream) || b->d_group_image.upload(group_image, b->stream) ||
        b->d_group_first.upload(group_first, b->stream))
        return -1;
    return hipStreamSynchronize(b->stream) == hipSuccess ? 0 : -1;   // the host vectors go out of scope
}

enum { MODE_PNG = 0, MODE_WEBP = 1, MODE_DECODE = 2, MODE_DECODE_ANY = 3 };   // DECODE: the front half of a resize (no 16-bit); DECODE_ANY: of a conversion to JPEG
struct PreFail { int code; std::string msg; };
static int png_create(const CByteArray *inputs, const csp_pixels *px, size_t count, const CCSParameters *p, int device, int mode, csp_batch **out, const std::vector<PreFail> *pre = nullptr,
                      const std::vector<uint8_t> *px_bits = nullptr);   // px_bits: 8 or 16 per pixel source (default 8)
static int png_create_resized(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, int mode, csp_batch **out);
extern "C" int csp_batch_create(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csp_batch **out) {
    return (p->width || p->height) ? png_create_resized(inputs, count, p, device, MODE_PNG, out) : png_create(inputs, nullptr, count, p, device, MODE_PNG, out);
}
extern "C" int csp_batch_create_webp(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csp_batch **out) {
    return (p->width || p->height) ? png_create_resized(inputs, count, p, device, MODE_WEBP, out) : png_create(inputs, nullptr, count, p, device, MODE_WEBP, out);
}
extern "C" int csp_batch_create_pixels(const csp_pixels *sources, size_t count, const CCSParameters *p, int device, csp_batch **out) { return png_create(nullptr, sources, count, p, device, MODE_PNG, out); }

// a source that is pixels already (csp_batch_create_pixels): the item a PNG file of that image would parse to
static void pixels_item(const csp_pixels &src, uint32_t bits, PngItem &it) {
    const uint32_t bps = bits / 8;
    static const uint8_t ctype_of[5] = {0, 0, 4, 2, 6};
    if (!src.device_pixels || !src.width || !src.height || src.channels < 1 || src.channels > 4 || src.width > 0x7FFFFFFFu || src.height > 0x7FFFFFFFu ||
        uint64_t(src.width) * src.channels * bps > 0x7FFFFFF0u) { it.code = CS_ERR_UNSUPPORTED; it.msg = "bad pixel source"; return; }
    it.width = src.width; it.height = src.height; it.depth = bits; it.ctype = ctype_of[src.channels]; it.channels = src.channels; it.bpp = src.channels * bps;
    it.rowbytes = src.width * src.channels * bps;
    uint8_t ihdr[25] = {0, 0, 0, 13, 'I', 'H', 'D', 'R'};
    put_be32(ihdr + 8, src.width); put_be32(ihdr + 12, src.height); ihdr[16] = uint8_t(bits); ihdr[17] = ctype_of[src.channels];
    put_be32(ihdr + 21, crc32_host(ihdr + 4, 17));
    it.prefix.assign(kSig, kSig + 8);
    it.prefix.insert(it.prefix.end(), ihdr, ihdr + 25);
    static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
    it.suffix.assign(iend, iend + 12);
}

static int png_create(const CByteArray *inputs, const csp_pixels *px, size_t count, const CCSParameters *p, int device, int mode, csp_batch **out, const std::vector<PreFail> *pre,
                      const std::vector<uint8_t> *px_bits) {
    *out = nullptr;
    const bool to_webp = mode == MODE_WEBP, decode_only = mode == MODE_DECODE || mode == MODE_DECODE_ANY;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { csh_set_error("no HIP device: libcaesium_hip has no CPU path"); return CS_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { csh_set_error("device %d out of range (%d visible)", device, ndev); return CS_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { csh_set_error("hipSetDevice(%d) failed", device); return CS_ERR_NO_DEVICE; }
    std::unique_ptr<csp_batch> b(new csp_batch);
    b->device = device;
    b->lossy = !p->png_optimize; b->png_quality = int(p->png_quality);
    b->deep_iters = (p->png_optimize && p->png_force_zopfli) ? int(CSP_DEEP_ITERS_ZOPFLI) : int(CSP_DEEP_ITERS);
    b->from_pixels = px != nullptr;
    b->decode_only = decode_only;
    b->to_webp = to_webp; b->webp_quality = int(p->webp_quality);
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { csh_set_error("hipStreamCreate failed"); return CS_ERR_NO_DEVICE; }
    b->have_stream = true;
    for (auto &e : b->ev) if (hipEventCreate(&e) != hipSuccess) { csh_set_error("hipEventCreate failed"); return CS_ERR_NO_DEVICE; }
    b->have_events = true;
    hipStream_t st = b->stream;

    // trial plan: the five fixed streams always exist (the adaptive ones are gathered out of them)
    int set[10];
    PngPlan &plan = b->plan;
    plan.ntrials = trial_set(int(p->png_optimization_level), set);
    for (int s = 0; s < 10; s++) b->slot_of_strategy[s] = s < 5 ? s : -1;
    for (int t = 0; t < plan.ntrials; t++) {
        const int s = set[t];
        if (s >= 5 && b->slot_of_strategy[s] < 0) { b->slot_of_strategy[s] = 5 + plan.nadaptive; plan.adaptive_strategy[plan.nadaptive++] = s; if (s == 9) plan.need_brute = 1; }
        plan.trial_slot[t] = b->slot_of_strategy[s]; plan.trial_strategy[t] = s;
    }
    const int nslots = (to_webp || decode_only) ? 0 : 5 + plan.nadaptive;   // a conversion has no filtered streams

    b->items.resize(count);
    b->inputs.resize(count);
    PinnedBytes idat_pool;
    std::vector<uint8_t> &fixed = b->fixed;
    size_t work_bytes = 0, stream_bytes = 256, out_bytes = 0;   // the tokenizer reads up to 8 bytes in front of a stream
    uint64_t nchunk_recs = 0;
    for (size_t i = 0; i < count; i++) {
        PngItem &it = b->items[i];
        b->inputs[i] = px ? nullptr : inputs[i].data;
        if (pre && (*pre)[i].code) { it.code = (*pre)[i].code; it.msg = (*pre)[i].msg; }
        else if (px) pixels_item(px[i], px_bits ? (*px_bits)[i] : 8, it);
        else parse_png(inputs[i].data, inputs[i].length, p->keep_metadata, it);
        if (it.code) continue;
        if (to_webp && uint64_t((it.width + 15) / 16) * ((it.height + 15) / 16) * 256 > 0x7FFFFFFFu) { it.code = CS_ERR_UNSUPPORTED; it.msg = "PNG too large for one device batch"; continue; }
        if (decode_only && it.has_trns && it.trns.size() != (it.ctype == 3 ? it.trns.size() : it.ctype == 0 ? 2u : it.ctype == 2 ? 6u : ~size_t(0))) { it.code = CS_ERR_BAD_PNG; it.msg = "bad tRNS"; continue; }
        PngImg im{};
        im.width = it.width; im.height = it.height; im.rowbytes = it.rowbytes; im.bpp = it.bpp;
        im.raw_len = uint64_t(it.height) * (uint64_t(it.rowbytes) + 1);
        if (im.raw_len > (uint64_t(1) << 36) || uint64_t(b->total_rows) + it.height > 0x7FFFFFFFu) { it.code = CS_ERR_UNSUPPORTED; it.msg = "PNG too large for one device batch"; continue; }
        im.idat_off = idat_pool.size(); im.idat_len = uint32_t(it.idat_len);
        if (!px) {   // the IDAT payloads back to back (a stream may be cut anywhere, even inside the zlib header)
            size_t at = idat_pool.size(), end = align_up(at + it.idat_len + 8, 256);
            if (!idat_pool.reserve(end)) { csh_set_error("out of pinned host memory"); return CS_ERR_NO_DEVICE; }
            for (auto &r : it.idat) { idat_pool.pending.push_back({at, inputs[i].data + r.first, r.second, 0}); at += r.second; }
            idat_pool.pending.push_back({at, inputs[i].data, 0, end - at});
            idat_pool.n = end;
        }
        {   // two regions of the work buffer per image.  Plain image: A takes the inflated stream, B the pixels.  Adam7: A takes the
            // seven passes' streams, B their reconstructed rows, and the gather puts the image back into A
            static const uint32_t XS[7] = {0, 4, 0, 2, 0, 1, 0}, YS[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
            const uint64_t bits = uint64_t(it.channels) * it.depth, image_bytes = uint64_t(it.height) * it.rowbytes;
            uint64_t pass_stream = 0, pass_pixels = 0;
            uint32_t pw[7], ph[7], prb[7];
            for (int p = 0; p < 7; p++) {
                pw[p] = (it.width + DX[p] - 1 - XS[p]) / DX[p]; ph[p] = (it.height + DY[p] - 1 - YS[p]) / DY[p];
                prb[p] = (pw[p] && ph[p]) ? uint32_t((uint64_t(pw[p]) * bits + 7) / 8) : 0u;
                if (prb[p]) { pass_stream += uint64_t(ph[p]) * (1 + prb[p]); pass_pixels += uint64_t(ph[p]) * prb[p]; }
            }
            im.inflate_len = it.interlace ? pass_stream : im.raw_len;
            const uint64_t A = work_bytes; work_bytes += align_up(std::max(im.inflate_len, image_bytes) + CSP_RAW_SLACK, 256);
            const uint64_t B = work_bytes; work_bytes += align_up(std::max(image_bytes, it.interlace ? pass_pixels : 0) + 64, 256);
            im.inflate_off = A;
            const uint32_t image_index = uint32_t(b->imgs.size());
            if (!it.interlace) {
                im.raw_off = A; im.pix_off = B;
                b->passes.push_back(PngPass{image_index, it.rowbytes, it.height, it.bpp, A, B});
            } else {
                im.pix_off = A; im.raw_off = B;
                PngAdam7 a{};
                a.image = image_index; a.bits = uint32_t(bits);
                uint64_t so = A, po = B;
                for (int p = 0; p < 7; p++) {
                    a.base[p] = po; a.prb[p] = prb[p];
                    if (!prb[p]) continue;
                    b->passes.push_back(PngPass{image_index, prb[p], ph[p], it.bpp, so, po});
                    so += uint64_t(ph[p]) * (1 + prb[p]); po += uint64_t(ph[p]) * prb[p];
                }
                b->adam7.push_back(a);
                b->adam7_items = std::max<uint64_t>(b->adam7_items, bits >= 8 ? uint64_t(it.width) * it.height : image_bytes);
            }
        }
        im.stream_stride = align_up(im.raw_len + 64, 256);
        im.stream_off = stream_bytes; im.match_off = stream_bytes / 8;   // a match is at least three bytes long
        stream_bytes += std::max<uint64_t>(im.stream_stride * uint64_t(nslots), align_up((im.inflate_len / 3 + 128) * 8, 256));
        im.row_base = b->total_rows; b->total_rows += it.height;
        im.nchunks = uint32_t((im.raw_len + CSP_CHUNK - 1) / CSP_CHUNK);
        im.chunk_base = uint32_t(nchunk_recs); nchunk_recs += uint64_t(im.nchunks) * nslots;
        im.chunk_stride = im.nchunks;
        im.channels = it.channels; im.bps = (it.ctype != 3 && it.depth >= 8) ? it.depth / 8 : 0;
        b->cand0.push_back((!it.no_reduce && !it.has_plte && im.bps && (im.channels == 3 || im.channels == 4)) ? im.channels : 0u);
        // bits 1 / 2 / 4 / 8-32: 16 -> 8 bits, alpha away, colour -> grey, grey depth 4 / 2 / 1; bit 64: an 8-bit indexed image (k_png_used finds the palette entries it uses)
        b->flags0.push_back((it.ctype == 3 && it.depth == 8 && !it.pal_tied && !px) ? 64u
                            : (it.no_reduce || !im.bps) ? 0u : ((im.bps == 2 ? 1u : 0u) | ((im.channels == 2 || im.channels == 4) ? 2u : 0u) | (im.channels >= 3 ? 4u : 0u) | 56u));
        if (nchunk_recs > 0x7FFFFFFFu) { csh_set_error("PNG batch too large"); return CS_ERR_POOL_OVERFLOW; }
        im.prefix_len = uint32_t(it.prefix.size()); im.suffix_len = uint32_t(it.suffix.size());
        im.fix_off = fixed.size();
        fixed.insert(fixed.end(), it.prefix.begin(), it.prefix.end());
        fixed.insert(fixed.end(), it.suffix.begin(), it.suffix.end());
        im.out_cap = uint64_t(im.prefix_len) + 1100 /* a PLTE and a tRNS chunk a reduction may add */ + 12 + im.suffix_len + 6 + uint64_t(im.nchunks) * (CSP_CHUNK + CSP_CHUNK / 8 + 1024);
        if (im.out_cap > 0xFFFFFFF0u) { it.code = CS_ERR_UNSUPPORTED; it.msg = "PNG too large for one device batch"; continue; }
        im.out_off = out_bytes;
        if (!to_webp && !decode_only) out_bytes += align_up(im.out_cap + 16, 256);
        const uint32_t pieces = uint32_t((im.out_cap + 1023) / 1024);
        if (pieces > b->max_pieces) b->max_pieces = pieces;
        it.image = int(b->imgs.size());
        if (to_webp) {
            RgbJob j{};
            j.image = uint32_t(it.image); j.width = it.width; j.height = it.height; j.rowbytes = it.rowbytes; j.ctype = it.ctype; j.depth = it.depth;
            j.plte_off = uint32_t(b->plte.size()); j.npal = uint32_t(it.plte.size() / 3);
            b->plte.insert(b->plte.end(), it.plte.begin(), it.plte.end());
            j.trns_off = uint32_t(b->plte.size()); j.ntrns = uint32_t(it.trns.size());
            b->plte.insert(b->plte.end(), it.trns.begin(), it.trns.end());
            j.src_off = im.pix_off; j.dst_off = b->rgb_bytes;
            // 8-bit grey / RGB for the VP8 encoder; an alpha channel or a tRNS chunk rides along as a fourth (second) sample: the encoder skips it, the
            // VP8L coder makes the file's ALPH chunk of it when the results are fetched
            const bool transparent = it.ctype == 4 || it.ctype == 6 || it.has_trns;
            const uint32_t nc = ((it.ctype == 0 || it.ctype == 4) ? 1u : 3u) + (transparent ? 1u : 0u);
            j.out_nc = nc;
            b->walpha.resize(b->imgs.size() + 1, 0);
            b->walpha[b->imgs.size()] = transparent ? uint8_t(nc) : uint8_t(0);
            b->rgb_bytes += align_up(uint64_t(it.width) * it.height * nc + 64, 256);
            b->rgb_max_h = std::max(b->rgb_max_h, it.height);
            b->rgbjobs.push_back(j);
            csw::WebpImg wi{};
            wi.width = it.width; wi.height = it.height; wi.mbw = (it.width + 15) / 16; wi.mbh = (it.height + 15) / 16; wi.ncomp = nc;
            wi.rgb_off = j.dst_off; wi.image = uint32_t(it.image);
            const uint64_t ly = uint64_t(wi.mbw) * wi.mbh * 256, lc = uint64_t(wi.mbw) * wi.mbh * 64;
            auto take = [&](uint64_t n) { uint64_t at = b->wwork_bytes; b->wwork_bytes += (n + 63) & ~uint64_t(63); return at; };
            wi.y_off = take(ly); wi.u_off = take(lc); wi.v_off = take(lc); wi.ry_off = take(ly); wi.ru_off = take(lc); wi.rv_off = take(lc);
            wi.lev_off = b->wlevels; b->wlevels += uint64_t(wi.mbw) * wi.mbh * csw::WEBP_MB_REC;
            b->wmax_luma = std::max<uint32_t>(b->wmax_luma, uint32_t(ly));
            b->wmax_mbh = std::max(b->wmax_mbh, wi.mbh);
            b->wimgs.push_back(wi);
        }
        b->imgs.push_back(im);
        b->raw_total += im.raw_len; b->pixels += uint64_t(it.width) * it.height;
    }
    idat_pool.flush_copies();
    const int nimg = int(b->imgs.size());
    std::vector<uint32_t> row_image(b->total_rows);
    {
        uint32_t r = 0;
        for (int i = 0; i < nimg; i++) for (uint32_t y = 0; y < b->imgs[i].height; y++) row_image[r++] = uint32_t(i);
    }
    if (upload_chunk_index(b.get())) return CS_ERR_NO_DEVICE;
    if (b->d_imgs.upload(b->imgs, st) || b->d_row_image.upload(row_image, st) || b->d_fixed.upload(fixed, st) || b->d_flags.alloc(size_t(nimg) + 1) || b->d_jobs.alloc(size_t(nimg) + 1))
        return CS_ERR_NO_DEVICE;
    if (b->d_idat.alloc(idat_pool.size() + 256) || b->d_work.alloc(work_bytes + 256) || b->d_passes.upload(b->passes, st) || b->d_adam7.upload(b->adam7, st) || b->d_streams.alloc(stream_bytes + 256) ||
        b->d_out.alloc(out_bytes + 256) || b->d_choice.alloc(size_t(5) * b->total_rows + 1) || b->d_status.alloc(size_t(nimg) + 1) || b->d_file_len.alloc(size_t(nimg) + 1) ||
        b->d_adler.alloc(2 * size_t(b->total_chunks) + 2) || b->d_crc.alloc(size_t(nimg) * b->max_pieces + 1) || b->d_scores.alloc(size_t(b->total_rows) * 25 + 1) ||
        b->d_trial_bytes.alloc(size_t(nimg) * CSP_MAX_STREAMS + 1) || b->d_nmatch.alloc(size_t(nimg) + 1) || b->d_winner.alloc(size_t(nimg) + 1) || b->d_trial_live.alloc(size_t(nimg) * CSP_MAX_STREAMS + 1) || b->d_chunks.alloc(size_t(nchunk_recs) + 1))
        return CS_ERR_NO_DEVICE;
    if (to_webp && (b->d_rgbjobs.upload(b->rgbjobs, st) || b->d_plte.upload(b->plte, st) || b->d_rgb.alloc(b->rgb_bytes + 256) || b->d_wwork.alloc(b->wwork_bytes + 64) ||
                    b->d_wlevels.alloc(b->wlevels + 64) || b->d_wstats.alloc(size_t(nimg) * 2112 + 8) || b->d_wprobs.alloc(size_t(nimg) * 1056 + 8) || b->d_wupdate.alloc(size_t(nimg) * 1056 + 8) ||
                    b->d_wpart.alloc(size_t(nimg) * 9 + 9) || b->d_wstatus.alloc(size_t(nimg) + 1)))
        return CS_ERR_NO_DEVICE;
    if (idat_pool.size()) {
        if (hipMemcpyAsync(b->d_idat.p, idat_pool.p, idat_pool.size(), hipMemcpyHostToDevice, st) != hipSuccess) { csh_set_error("upload failed"); return CS_ERR_NO_DEVICE; }
    }
    for (size_t i = 0; px && i < count; i++) {
        const PngItem &it = b->items[i];
        if (it.image < 0) continue;
        if (hipMemcpyAsync(b->d_work.p + b->imgs[it.image].pix_off, px[i].device_pixels, size_t(it.height) * it.rowbytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { csh_set_error("pixel copy failed"); return CS_ERR_NO_DEVICE; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { csh_set_error("upload failed"); return CS_ERR_NO_DEVICE; }   // the pinned pool goes back to the cache
    *out = b.release();
    return 0;
}

// P2: which reductions the pixels allow (device), the new geometry and IHDR (host), the repack (device).  One host round trip
// per batch; afterwards every later stage sees the reduced image as if it had come in that way.
static int reduce_step(csp_batch *b) {
    hipStream_t st = b->stream;
    const int nimg = int(b->imgs.size());
    b->reduced = true;
    bool any = false;
    for (int i = 0; i < nimg; i++) any |= b->flags0[i] != 0 || b->cand0[i] != 0;
    if (!any || !nimg) return 0;
    if (b->d_keys.alloc(size_t(nimg) * CSP_PAL_SLOTS) || b->d_slot_index.alloc(size_t(nimg) * CSP_PAL_SLOTS) || b->d_counts.alloc(size_t(nimg) + 1) || b->d_cand.upload(b->cand0, st)) return -1;
    if (hipMemcpyAsync(b->d_flags.p, b->flags0.data(), sizeof(uint32_t) * nimg, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(b->d_keys.p, 0xFF, sizeof(unsigned long long) * size_t(nimg) * CSP_PAL_SLOTS, st) != hipSuccess || b->d_counts.zero(st)) return -1;
    launch_png_analyze(st, b->d_imgs.p, b->total_rows, b->d_row_image.p, b->d_work.p, b->d_flags.p, b->d_status.p);
    launch_png_colors(st, b->d_imgs.p, b->total_rows, b->d_row_image.p, b->d_work.p, b->d_cand.p, b->d_keys.p, b->d_counts.p, b->d_status.p);
    std::vector<uint32_t> flags(nimg), status(nimg), counts(nimg), used(size_t(nimg) * 8, 0);
    bool any_indexed = false;
    for (int i = 0; i < nimg; i++) any_indexed |= (b->flags0[i] & 64u) != 0;
    DevBuf<uint32_t> d_used;
    if (any_indexed) {
        if (d_used.alloc(size_t(nimg) * 8) || d_used.zero(st)) return -1;
        launch_png_used(st, b->d_imgs.p, b->total_rows, b->d_row_image.p, b->d_work.p, b->d_flags.p, d_used.p, b->d_status.p);
        if (hipMemcpyAsync(used.data(), d_used.p, sizeof(uint32_t) * used.size(), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    }
    if (hipMemcpyAsync(flags.data(), b->d_flags.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(counts.data(), b->d_counts.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(status.data(), b->d_status.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        csh_set_error("PNG analysis failed");
        return -1;
    }
    std::vector<size_t> item_of(nimg, 0);
    for (size_t n = 0; n < b->items.size(); n++) if (b->items[n].image >= 0) item_of[b->items[n].image] = n;
    // lossy: truecolour images that keep more than 256 colours after the lossless reductions get their colour bins counted and compacted, and a
    // workgroup per image runs the median cut over the non-empty bins (a second round trip for the <= 256 palette entries, only for such images)
    std::map<int, std::vector<uint32_t>> qpal;
    if (b->lossy) {
        std::vector<QuantJob> qjobs;
        uint64_t list_total = 0;
        uint32_t qmax_h = 0;
        for (int i = 0; i < nimg; i++) {
            if (status[i] || !b->cand0[i] || (flags[i] & 4u) || counts[i] <= 256) continue;
            const PngImg &im = b->imgs[i];
            QuantJob q{};
            q.image = uint32_t(i); q.channels = im.channels; q.bps = im.bps; q.rowbytes = im.rowbytes; q.width = im.width; q.height = im.height; q.src_off = im.pix_off;
            q.bins_off = uint64_t(qjobs.size()) * CSP_QBINS * 5; q.list_off = list_total;
            list_total += std::min<uint64_t>(uint64_t(im.width) * im.height, CSP_QBINS);
            qmax_h = std::max(qmax_h, im.height);
            qjobs.push_back(q);
        }
        // a few images at a time: 10 MB of bins each
        for (size_t g0 = 0; g0 < qjobs.size(); g0 += 64) {
            const size_t gn = std::min<size_t>(64, qjobs.size() - g0);
            std::vector<QuantJob> part(qjobs.begin() + g0, qjobs.begin() + g0 + gn);
            uint64_t lt = 0;
            for (size_t k = 0; k < gn; k++) { part[k].bins_off = uint64_t(k) * CSP_QBINS * 5; part[k].list_off = lt; lt += std::min<uint64_t>(uint64_t(part[k].width) * part[k].height, CSP_QBINS); }
            if (b->d_qbins.alloc(gn * size_t(CSP_QBINS) * 5) || b->d_qbins.zero(st) || b->d_qlist.alloc(lt + 1) || b->d_qn.alloc(gn + 1) || b->d_qn.zero(st) || b->d_qjobs.upload(part, st)) return -1;
            launch_png_qhist(st, b->d_qjobs.p, int(gn), qmax_h, b->d_work.p, b->d_qbins.p);
            launch_png_qcompact(st, b->d_qjobs.p, int(gn), b->d_qbins.p, b->d_qlist.p, b->d_qn.p);
            // the median cut of every image's list, one workgroup each (k_png_mediancut); the host only sorts the <= 256 entries and merges equal ones
            DevBuf<uint32_t> d_qord, d_cutpal, d_ncut;
            DevBuf<uint4> d_qrec;
            if (d_qrec.alloc(lt + 1) || d_qord.alloc(2 * (lt + 1)) || d_cutpal.alloc(gn * 256) || d_ncut.alloc(gn)) return -1;
            const int q = b->png_quality < 0 ? 0 : b->png_quality > 100 ? 100 : b->png_quality;
            launch_png_mediancut(st, b->d_qjobs.p, int(gn), b->d_qlist.p, b->d_qn.p, d_qrec.p, d_qord.p, lt + 1, q, kQualityBound[q], d_cutpal.p, d_ncut.p);
            std::vector<uint32_t> cutpal(gn * 256), ncut(gn);
            if (csh_copy_wait(cutpal.data(), d_cutpal.p, sizeof(uint32_t) * gn * 256, hipMemcpyDeviceToHost, st) != hipSuccess ||
                csh_copy_wait(ncut.data(), d_ncut.p, sizeof(uint32_t) * gn, hipMemcpyDeviceToHost, st) != hipSuccess || hipGetLastError() != hipSuccess) return -1;
            std::vector<std::vector<uint32_t>> cut(gn);
            for (size_t k = 0; k < gn; k++) {
                cut[k].assign(cutpal.begin() + k * 256, cutpal.begin() + k * 256 + std::min<uint32_t>(ncut[k], 256));
                std::sort(cut[k].begin(), cut[k].end());
                cut[k].erase(std::unique(cut[k].begin(), cut[k].end()), cut[k].end());
            }
            for (size_t k = 0; k < gn; k++) qpal[int(part[k].image)] = std::move(cut[k]);
        }
    }
    std::vector<ReduceJob> jobs;
    std::vector<uint8_t> remaps;   // 256 bytes per indexed job: old palette index -> new
    std::vector<PaletteJob> pjobs;
    uint64_t dither_pixels = 0, dither_steps = 0;   // k_png_dither: line-buffer pixels of the quantised images, the longest image's steps
    std::vector<uint32_t> palettes;
    std::vector<uint16_t> slot_index(size_t(nimg) * CSP_PAL_SLOTS, 0);
    uint32_t max_height = 0;
    bool changed = false;
    for (int i = 0; i < nimg; i++) {
        if (status[i]) continue;
        PngImg &im = b->imgs[i];
        PngItem &it = b->items[item_of[i]];
        const bool narrow = flags[i] & 1u, opaque = flags[i] & 2u, grey = flags[i] & 4u;
        const uint32_t nk = im.channels - (opaque ? 1u : 0u) - (grey ? 2u : 0u), nbps = narrow ? 1u : im.bps;
        // colour -> palette (oracle: to_palette): at most 256 distinct pixels, and smaller rows even with the PLTE / tRNS chunks
        std::vector<uint32_t> pal;
        std::vector<unsigned long long> tab;
        uint32_t depth = 0, ntr = 0;
        bool nearest = false;
        if (b->cand0[i] && !grey && nbps == 1 && (nk == 3 || nk == 4) && counts[i] <= 256) {
            tab.resize(CSP_PAL_SLOTS);
            if (csh_copy_wait(tab.data(), b->d_keys.p + size_t(i) * CSP_PAL_SLOTS, sizeof(unsigned long long) * CSP_PAL_SLOTS, hipMemcpyDeviceToHost, b->stream) != hipSuccess) return -1;
            for (auto k : tab) if (k != ~0ull) pal.push_back(uint32_t(k));
            std::sort(pal.begin(), pal.end());
            const uint32_t n = uint32_t(pal.size());
            for (uint32_t k = 0; k < n; k++) if ((pal[k] >> 24) != 255) ntr = k + 1;
            depth = n <= 2 ? 1 : n <= 4 ? 2 : n <= 16 ? 4 : 8;
            const uint64_t nrb = (uint64_t(im.width) * depth + 7) / 8, extra = 12 + 3 * uint64_t(n) + (ntr ? 12 + ntr : 0);
            if (uint64_t(im.height) * (1 + nrb) + extra >= uint64_t(im.height) * (1 + uint64_t(im.width) * nk)) depth = 0;
        }
        // lossy: what is still truecolour is quantised (oracle: quantize) -- the palette was made by the median cut above this loop
        if (!depth && qpal.count(i)) {
            pal = qpal[i];
            const uint32_t n = uint32_t(pal.size());
            ntr = 0;
            for (uint32_t k = 0; k < n; k++) if ((pal[k] >> 24) != 255) ntr = k + 1;
            depth = n <= 2 ? 1 : n <= 4 ? 2 : n <= 16 ? 4 : 8;
            nearest = true;
        }
        // 8-bit grey -> 4 / 2 / 1 bit (oracle: grey_depth): the result is a single 8-bit channel whose every level fits
        uint32_t gdepth = 0;
        if (!depth && nk == 1 && nbps == 1) gdepth = (flags[i] & 32u) ? 1u : (flags[i] & 16u) ? 2u : (flags[i] & 8u) ? 4u : 0u;
        // an 8-bit indexed image that does not use its whole palette: the unused entries go, the rest is renumbered and packed at the depth it needs (oracle: index_depth)
        uint32_t idepth = 0, nused = 0;
        uint8_t imap[256], iorder[256];
        if (flags[i] & 64u) {
            const uint32_t npl = uint32_t(it.plte.size() / 3);
            bool ok = true;
            // the entries that stay (oracle: index_depth): one per distinct colour among the used ones, those that are not opaque in front of the opaque ones
            int first_of[256];
            uint32_t col[256];
            auto is_used = [&](uint32_t v) { return ((used[size_t(i) * 8 + (v >> 5)] >> (v & 31u)) & 1u) != 0; };
            for (uint32_t v = 0; v < 256; v++) {
                first_of[v] = -1; imap[v] = 0;
                if (!is_used(v)) continue;
                if (v >= npl) { ok = false; continue; }
                col[v] = (uint32_t(v < it.trns.size() ? it.trns[v] : 255) << 24) | (uint32_t(it.plte[3 * v]) << 16) | (uint32_t(it.plte[3 * v + 1]) << 8) | it.plte[3 * v + 2];
                first_of[v] = int(v);
                for (uint32_t k = 0; k < v; k++) if (first_of[k] == int(k) && col[k] == col[v]) { first_of[v] = int(k); break; }
            }
            for (int pass = 0; pass < 2 && ok; pass++)
                for (uint32_t v = 0; v < 256; v++)
                    if (first_of[v] == int(v) && ((col[v] >> 24) != 255) == (pass == 0)) { imap[v] = uint8_t(nused); iorder[nused++] = uint8_t(v); }
            for (uint32_t v = 0; v < 256 && ok; v++) if (first_of[v] >= 0 && first_of[v] != int(v)) imap[v] = imap[first_of[v]];
            idepth = nused <= 2 ? 1u : nused <= 4 ? 2u : nused <= 16 ? 4u : 8u;
            if (!ok || !nused || (idepth == 8 && nused == npl)) idepth = 0;
        }
        if (!depth && !(flags[i] & 7u) && !gdepth && !idepth) continue;
        changed = true;
        uint8_t *ihdr = &it.prefix[8];   // the new IHDR: depth, colour type, checksum
        if (depth) {
            PaletteJob j{};
            j.image = uint32_t(i); j.old_rowbytes = im.rowbytes; j.old_channels = im.channels; j.old_bps = im.bps; j.depth = depth; j.table = uint32_t(i);
            j.src_off = im.pix_off; j.dst_off = im.raw_off;
            j.nearest = nearest ? 2u : 0u; j.npal = uint32_t(pal.size()); j.pal_off = uint32_t(palettes.size());   // (2: with error diffusion, k_png_dither)
            if (nearest) {
                j.line_off = uint32_t(dither_pixels); dither_pixels += 2 * uint64_t(im.width);
                const uint64_t steps = uint64_t((im.height + CSP_DITHER_ROWS - 1) / CSP_DITHER_ROWS) * (uint64_t(im.width) + 2 * CSP_DITHER_ROWS);
                dither_steps = std::max(dither_steps, steps);
            }
            if (nearest) palettes.insert(palettes.end(), pal.begin(), pal.end());
            for (uint32_t sl = 0; sl < CSP_PAL_SLOTS && !nearest; sl++)
                if (tab[sl] != ~0ull) slot_index[size_t(i) * CSP_PAL_SLOTS + sl] = uint16_t(std::lower_bound(pal.begin(), pal.end(), uint32_t(tab[sl])) - pal.begin());
            im.channels = 1; im.bps = 0; im.bpp = 1; im.rowbytes = uint32_t((uint64_t(im.width) * depth + 7) / 8);
            ihdr[8 + 8] = uint8_t(depth); ihdr[8 + 9] = 3;
            const uint32_t n = uint32_t(pal.size());
            std::vector<uint8_t> ch(12 + 3 * n);
            put_be32(ch.data(), 3 * n); memcpy(&ch[4], "PLTE", 4);
            for (uint32_t k = 0; k < n; k++) { ch[8 + 3 * k] = uint8_t(pal[k] >> 16); ch[9 + 3 * k] = uint8_t(pal[k] >> 8); ch[10 + 3 * k] = uint8_t(pal[k]); }
            put_be32(&ch[8 + 3 * n], crc32_host(&ch[4], 4 + 3 * n));
            it.prefix.insert(it.prefix.end(), ch.begin(), ch.end());
            if (ntr) {
                std::vector<uint8_t> tr(12 + ntr);
                put_be32(tr.data(), ntr); memcpy(&tr[4], "tRNS", 4);
                for (uint32_t k = 0; k < ntr; k++) tr[8 + k] = uint8_t(pal[k] >> 24);
                put_be32(&tr[8 + ntr], crc32_host(&tr[4], 4 + ntr));
                it.prefix.insert(it.prefix.end(), tr.begin(), tr.end());
            }
            pjobs.push_back(j);
        } else if (idepth) {
            ReduceJob j{};
            j.image = uint32_t(i); j.mask = 0; j.old_rowbytes = im.rowbytes; j.old_channels = 1; j.old_bps = 1;
            j.src_off = im.pix_off; j.dst_off = im.raw_off;
            j.gdepth = idepth | 256u; j.remap = uint32_t(remaps.size() / 256);
            remaps.insert(remaps.end(), imap, imap + 256);
            im.bps = 0; im.bpp = 1; im.rowbytes = uint32_t((uint64_t(im.width) * idepth + 7) / 8);
            ihdr[8 + 8] = uint8_t(idepth);
            // PLTE and tRNS of the entries that are left (a tRNS that ends up all opaque goes): the carried chunks behind IHDR, written again
            std::vector<uint8_t> npl, ntr;
            uint32_t nt = 0;
            for (uint32_t k = 0; k < nused; k++) {
                const uint32_t v = iorder[k];
                npl.insert(npl.end(), it.plte.begin() + 3 * v, it.plte.begin() + 3 * v + 3);
                ntr.push_back(v < it.trns.size() ? it.trns[v] : uint8_t(255));
                if (ntr.back() != 255) nt = k + 1;
            }
            ntr.resize(nt);
            std::vector<uint8_t> np(it.prefix.begin(), it.prefix.begin() + 33);
            for (size_t pos = 33; pos + 12 <= it.prefix.size();) {
                const uint32_t len = be32(&it.prefix[pos]);
                const uint8_t *type = &it.prefix[pos + 4];
                const uint8_t *data = &it.prefix[pos + 8];
                uint32_t nlen = len;
                bool drop = false;
                if (!memcmp(type, "PLTE", 4)) { nlen = uint32_t(npl.size()); data = npl.data(); }
                else if (!memcmp(type, "tRNS", 4)) { nlen = nt; data = ntr.data(); drop = nt == 0; }
                if (!drop) {
                    const size_t at = np.size();
                    np.resize(at + 12 + nlen);
                    put_be32(&np[at], nlen); memcpy(&np[at + 4], type, 4); if (nlen) memcpy(&np[at + 8], data, nlen);
                    put_be32(&np[at + 8 + nlen], crc32_host(&np[at + 4], 4 + nlen));
                }
                pos += 12 + size_t(len);
            }
            it.prefix.swap(np);
            it.plte = npl; it.trns = ntr;
            ihdr = &it.prefix[8];
            jobs.push_back(j);
        } else {
            ReduceJob j{};
            j.image = uint32_t(i); j.mask = flags[i] & 7u; j.old_rowbytes = im.rowbytes; j.old_channels = im.channels; j.old_bps = im.bps;
            j.src_off = im.pix_off; j.dst_off = im.raw_off;   // the second region of the image takes the new pixels
            j.gdepth = gdepth;
            im.channels = nk; im.bps = nbps; im.bpp = nk * nbps; im.rowbytes = im.width * nk * nbps;
            if (gdepth) { im.bps = 0; im.bpp = 1; im.rowbytes = uint32_t((uint64_t(im.width) * gdepth + 7) / 8); }
            ihdr[8 + 8] = uint8_t(gdepth ? gdepth : nbps * 8); ihdr[8 + 9] = uint8_t(nk == 1 ? 0 : nk == 2 ? 4 : nk == 3 ? 2 : 6);
            jobs.push_back(j);
        }
        ihdr = &it.prefix[8];
        put_be32(ihdr + 8 + 13, crc32_host(ihdr + 4, 17));
        im.raw_len = uint64_t(im.height) * (uint64_t(im.rowbytes) + 1);
        im.nchunks = uint32_t((im.raw_len + CSP_CHUNK - 1) / CSP_CHUNK);
        std::swap(im.pix_off, im.raw_off);
        if (im.height > max_height) max_height = im.height;
    }
    if (!changed) return 0;
    b->n_reduced = uint32_t(jobs.size() + pjobs.size());
    b->raw_total = 0;
    b->fixed.clear();   // prefixes changed (IHDR, perhaps PLTE / tRNS): lay the carried bytes out again
    for (int i = 0; i < nimg; i++) {
        PngImg &im = b->imgs[i];
        const PngItem &it = b->items[item_of[i]];
        b->raw_total += im.raw_len;
        im.fix_off = b->fixed.size(); im.prefix_len = uint32_t(it.prefix.size()); im.suffix_len = uint32_t(it.suffix.size());
        b->fixed.insert(b->fixed.end(), it.prefix.begin(), it.prefix.end());
        b->fixed.insert(b->fixed.end(), it.suffix.begin(), it.suffix.end());
    }
    jobs.push_back(ReduceJob{}); pjobs.push_back(PaletteJob{});   // never empty uploads
    palettes.push_back(0);
    if (b->d_fixed.upload(b->fixed, st) || b->d_pjobs.upload(pjobs, st) || b->d_slot_index.upload(slot_index, st) || b->d_qpal.upload(palettes, st) ||
        hipMemcpyAsync(b->d_imgs.p, b->imgs.data(), sizeof(PngImg) * nimg, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(b->d_jobs.p, jobs.data(), sizeof(ReduceJob) * jobs.size(), hipMemcpyHostToDevice, st) != hipSuccess) { csh_set_error("PNG reduction upload failed"); return -1; }
    remaps.resize(remaps.size() + 256, 0);   // never an empty upload
    DevBuf<uint8_t> d_remaps;
    if (d_remaps.upload(remaps, st)) return -1;
    launch_png_repack(st, b->d_imgs.p, b->d_jobs.p, int(jobs.size()) - 1, max_height, b->d_work.p, b->d_work.p, d_remaps.p);
    launch_png_indexed(st, b->d_imgs.p, b->d_pjobs.p, int(pjobs.size()) - 1, max_height, b->d_keys.p, b->d_slot_index.p, b->d_qpal.p, b->d_work.p, b->d_work.p);
    DevBuf<int16_t> d_lines;   // the error rows the bands of k_png_dither hand down (freed behind the synchronisation below)
    if (dither_steps) {
        if (dither_steps > 0x3FFFFFFFull || d_lines.alloc(size_t(dither_pixels) * 4 + 4)) { csh_set_error("PNG dither buffers failed"); return -1; }
        launch_png_dither(st, b->d_imgs.p, b->d_pjobs.p, int(pjobs.size()) - 1, int(dither_steps), b->d_qpal.p, b->d_work.p, b->d_work.p, d_lines.p);
    }
    return upload_chunk_index(b);   // synchronises: the job vectors may go out of scope
}

// conversion to WebP: pixels -> 8-bit RGB -> the VP8 encoder of webp_kernels.h (statement: oracle/webp_oracle.c).  The output pool is
// sized per macroblock and grows when a file overflows it, as in the JPEG -> WebP path (pipeline.cpp: run_webp)
static int run_to_webp(csp_batch *b) {
    hipStream_t st = b->stream;
    const int nimg = int(b->wimgs.size());
    if (!nimg) return 0;
    launch_png_rgb(st, b->d_rgbjobs.p, nimg, b->rgb_max_h, b->d_plte.p, b->d_work.p, b->d_rgb.p, b->d_status.p);
    const int q = b->webp_quality, quality = q < 0 ? 0 : q > 100 ? 100 : q;
    b->h_wstatus.assign(size_t(nimg), 0);
    for (int attempt = 0; attempt < 4; attempt++) {
        uint64_t out_bytes = 0;
        for (auto &wi : b->wimgs) {
            const uint64_t cap = 4096 + uint64_t(wi.mbw) * wi.mbh * (b->webp_mb_bytes + 2);
            wi.out_cap = uint32_t(std::min<uint64_t>(cap, 0xFFFFFF00u)); wi.out_off = out_bytes; wi.quality = quality;
            b->imgs[wi.image].out_off = out_bytes;
            out_bytes += (wi.out_cap + 63) & ~uint64_t(63);
        }
        if (b->d_out.alloc(out_bytes + 64) || b->d_wscratch.alloc(out_bytes + 64) || b->d_wimgs.upload(b->wimgs, st) || b->d_wstats.zero(st) || b->d_wstatus.zero(st) || b->d_file_len.zero(st)) return CS_ERR_NO_DEVICE;
        csw::launch_webp_yuv(st, b->d_wimgs.p, nimg, b->wmax_luma, b->d_rgb.p, b->d_wwork.p);
        if (csw::launch_webp_encode(st, b->wimgs.data(), nimg, b->d_wimgs.p, b->d_wwork.p, b->d_wlevels.p, b->d_wscratch.p, b->d_wpart.p, b->d_out.p, b->d_file_len.p, b->d_wstatus.p, nullptr)) return CS_ERR_NO_DEVICE;
        if (hipMemcpyAsync(b->h_wstatus.data(), b->d_wstatus.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
            hipGetLastError() != hipSuccess) { csh_set_error("WebP kernels failed"); return CS_ERR_NO_DEVICE; }
        bool pool = false;
        for (uint32_t s : b->h_wstatus) if (s == CS_ERR_POOL_OVERFLOW) pool = true;
        if (!pool) break;
        if (attempt == 3) { csh_set_error("device pools overflowed after 3 retries"); return CS_ERR_POOL_OVERFLOW; }
        b->webp_mb_bytes *= 4;
    }
    return 0;
}

extern "C" int csp_batch_run(csp_batch *b, csp_timing *t) {
    if (!b) return CS_ERR_NO_DEVICE;
    if (hipSetDevice(b->device) != hipSuccess) { csh_set_error("hipSetDevice failed"); return CS_ERR_NO_DEVICE; }
    hipStream_t st = b->stream;
    const int nimg = int(b->imgs.size());
    if (!b->reduced && b->d_status.zero(st)) return CS_ERR_NO_DEVICE;
    FilterCtx f{};
    f.imgs = b->d_imgs.p; f.nimg = nimg; f.total_rows = b->total_rows; f.row_image = b->d_row_image.p; f.pix = b->d_work.p; f.streams = b->d_streams.p;
    f.scores = b->d_scores.p; f.choice = b->d_choice.p; f.plan = b->plan; f.status = b->d_status.p;
    for (const PngImg &im : b->imgs) f.max_rowbytes = std::max(f.max_rowbytes, im.rowbytes);
    DeflateCtx d{};
    d.imgs = b->d_imgs.p; d.nimg = nimg; d.total_chunks = b->total_chunks; d.chunk_image = b->d_chunk_image.p; d.chunk_first = b->d_chunk_first.p;
    d.total_groups = b->total_groups; d.group_image = b->d_group_image.p; d.group_first = b->d_group_first.p;
    d.streams = b->d_streams.p; d.chunks = b->d_chunks.p; d.plan = b->plan; d.trial_bytes = b->d_trial_bytes.p; d.winner = b->d_winner.p; d.trial_live = b->d_trial_live.p;
    d.adler_parts = b->d_adler.p; d.out = b->d_out.p; d.fixed = b->d_fixed.p; d.file_len = b->d_file_len.p; d.crc_parts = b->d_crc.p; d.status = b->d_status.p;
    bool need_scores = false;
    for (int a = 0; a < b->plan.nadaptive; a++) if (b->plan.adaptive_strategy[a] != 9) need_scores = true;
    int k = 0;
    auto mark = [&]() { (void)hipEventRecord(b->ev[k++], st); };
    mark(); if (!b->reduced && !b->from_pixels) launch_png_inflate(st, b->d_imgs.p, nimg, b->d_idat.p, b->d_work.p, reinterpret_cast<uint64_t *>(b->d_streams.p), b->d_nmatch.p, b->d_status.p);
    mark(); if (!b->reduced && !b->from_pixels) {
        { uint32_t mh = 0; for (const PngPass &pp : b->passes) mh = std::max(mh, pp.height); launch_png_unfilter(st, b->d_passes.p, int(b->passes.size()), mh, b->d_work.p, b->d_status.p); }
        launch_png_deinterlace(st, b->d_imgs.p, b->d_adam7.p, int(b->adam7.size()), b->adam7_items, b->d_work.p, b->d_status.p);
    }
    if (b->decode_only) {
        mark();
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { csh_set_error("PNG kernels failed"); return CS_ERR_NO_DEVICE; }
        b->ran = true;
        if (t) memset(t, 0, sizeof *t);
        return 0;
    }
    if (b->to_webp) {
        mark();
        const int rc = run_to_webp(b);
        if (rc) return rc;
        mark();
        b->ran = true;
        if (t) {
            memset(t, 0, sizeof *t);
            (void)hipEventElapsedTime(&t->total_ms, b->ev[0], b->ev[k - 1]);
            for (int i = 0; i + 1 < k; i++) (void)hipEventElapsedTime(&t->kernel_ms[i], b->ev[i], b->ev[i + 1]);
            t->pixels = b->pixels; t->raw_bytes = b->raw_total; t->n_images = uint32_t(nimg);
        }
        return 0;
    }
    mark(); if (!b->reduced && reduce_step(b)) return CS_ERR_NO_DEVICE;
    d.total_chunks = b->total_chunks; d.total_groups = b->total_groups; d.fixed = b->d_fixed.p;   // the reduction step may have re-laid these out
    d.chunk_image = b->d_chunk_image.p; d.chunk_first = b->d_chunk_first.p; d.group_image = b->d_group_image.p; d.group_first = b->d_group_first.p;
    mark(); launch_png_filter5(st, f);
    mark(); if (need_scores) launch_png_scores(st, f);
    mark(); if (b->plan.need_brute) launch_png_brute(st, f);
    mark(); launch_png_pick(st, f);
    {   // one scratch area per workgroup the device holds at the parse kernels' LDS footprint (three per CU), no more than there are items
        const uint64_t items = uint64_t(b->total_chunks) * uint32_t(b->plan.ntrials);
        b->deep_slots = uint32_t(std::min<uint64_t>(items, 768));
        if (b->deep_slots && (b->d_deep.alloc(size_t(b->deep_slots) * CSP_DEEP_SCRATCH) || b->d_deep_queue.alloc(4) || b->d_deep_list.alloc(size_t(items) + 1))) return CS_ERR_NO_DEVICE;
        d.deep_scratch = b->d_deep.p; d.deep_queue = b->d_deep_queue.p; d.deep_list = b->d_deep_list.p; d.deep_slots = b->deep_slots; d.deep_iters = b->deep_iters;
    }
    mark(); launch_png_hist(st, d);
    mark(); launch_png_codes(st, d);
    mark(); launch_png_choose(st, d);
    mark(); launch_png_deep(st, d);
    mark(); launch_png_emit(st, d);
    mark(); launch_png_finish(st, d, b->max_pieces);
    mark();
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { csh_set_error("PNG kernels failed"); return CS_ERR_NO_DEVICE; }
    b->ran = true;
    if (t) {
        memset(t, 0, sizeof *t);
        (void)hipEventElapsedTime(&t->total_ms, b->ev[0], b->ev[k - 1]);
        for (int i = 0; i + 1 < k; i++) (void)hipEventElapsedTime(&t->kernel_ms[i], b->ev[i], b->ev[i + 1]);
        std::vector<uint32_t> status(size_t(nimg) + 1), flen(size_t(nimg) + 1);
        if (nimg) {
            (void)csh_copy_wait(status.data(), b->d_status.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, b->stream);
            (void)csh_copy_wait(flen.data(), b->d_file_len.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, b->stream);
        }
        for (auto &it : b->items) {
            if (it.image < 0) { t->n_failed++; continue; }
            if (status[it.image]) { t->n_failed++; continue; }
            t->in_bytes += it.idat_len; t->out_bytes += flen[it.image];
        }
        t->pixels = b->pixels; t->raw_bytes = b->raw_total; t->n_images = uint32_t(nimg); t->n_trials = uint32_t(b->plan.ntrials);
    }
    return 0;
}

// width / height on PNG sources (libcaesium png::compress with a size: decode, image-rs resize_exact Lanczos3, encode): a decode-only
// batch, the two Lanczos passes over its pixels, then the coder -- or, on the way to WebP, the VP8 encoder -- over the resized pixels
// (device to device, as for JPEG -> PNG).
// The pixels are expanded first as the png crate does for image-rs (palette looked up, sub-byte grey scaled, tRNS as an alpha channel:
// k_png_rgb); 16-bit sources keep their 16 bits (a tRNS chunk becomes a 16-bit alpha sample).
static int png_create_resized(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, int mode, csp_batch **out) {
    *out = nullptr;
    csp_batch *raw = nullptr;
    int rc = png_create(inputs, nullptr, count, p, device, MODE_DECODE, &raw);
    std::unique_ptr<csp_batch> a(raw);
    if (rc == 0) rc = csp_batch_run(a.get(), nullptr);
    if (rc) return rc;
    hipStream_t st = a->stream;
    const int nimg = int(a->imgs.size());
    std::vector<uint32_t> status(size_t(nimg) + 1, 0);
    if (nimg && csh_copy_wait(status.data(), a->d_status.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, a->stream) != hipSuccess) { csh_set_error("download failed"); return CS_ERR_NO_DEVICE; }
    std::vector<PreFail> pre(count);
    std::vector<csp_pixels> px(count);
    std::vector<PngResize> jobs;
    std::vector<RgbJob> ejobs;
    std::vector<uint8_t> tables(1, 0);   // PLTE and tRNS payloads
    std::vector<size_t> job_item;
    std::vector<csh::ResizeTap> taps;
    std::vector<float> weights;
    uint64_t tmp_floats = 0, dst_bytes = 0, max_tmp = 0, max_dst = 0, src_bytes = 0;
    uint32_t max_h = 0;
    struct RawCopy { uint64_t dst, src, bytes; };
    std::vector<RawCopy> copies;
    std::vector<uint8_t> bits(count, 8);
    for (size_t i = 0; i < count; i++) {
        const PngItem &it = a->items[i];
        px[i] = csp_pixels{nullptr, 0, 0, 0};
        if (it.code) { pre[i] = PreFail{it.code, it.msg}; continue; }
        if (status[it.image]) { pre[i] = PreFail{int(status[it.image]), "malformed PNG data"}; continue; }
        // what the png crate's EXPAND transformation hands image-rs: 8-bit samples, palette looked up, tRNS as an alpha channel
        // (16-bit images stay as they are: image-rs resamples L16 / La16 / Rgb16 / Rgba16 at 16 bits)
        const bool wide = it.depth == 16;
        if (wide && it.has_trns && mode != MODE_PNG) { pre[i] = PreFail{CS_ERR_UNSUPPORTED, "resizing a 16-bit PNG with a tRNS chunk on the way to another format has no device path in this build"}; continue; }
        const uint32_t colour = (it.ctype == 2 || it.ctype == 6 || it.ctype == 3) ? 3u : 1u, nc = colour + ((it.ctype == 4 || it.ctype == 6 || it.has_trns) ? 1u : 0u), bps = wide ? 2u : 1u;
        int nw = 0, nh = 0;
        csh_compute_dimensions(int(it.width), int(it.height), int(p->width), int(p->height), nw, nh);
        const uint64_t tmpn = uint64_t(nh) * it.width * nc, dstn = uint64_t(nw) * nh * nc;
        if (uint64_t(nw) * nc * bps > 0x7FFFFFF0u || uint64_t(it.width) * nc * bps > 0x7FFFFFF0u || tmpn > 0xFFFFFF00u || dstn > 0xFFFFFF00u / bps) { pre[i] = PreFail{CS_ERR_UNSUPPORTED, "resized PNG too large for one device batch"}; continue; }   // one lane per sample: a launch holds 2^32 of them
        RgbJob e{};
        e.image = uint32_t(it.image); e.width = it.width; e.height = it.height; e.rowbytes = it.rowbytes; e.ctype = it.ctype; e.depth = it.depth; e.out_nc = nc;
        e.plte_off = uint32_t(tables.size()); e.npal = uint32_t(it.plte.size() / 3);
        tables.insert(tables.end(), it.plte.begin(), it.plte.end());
        e.trns_off = uint32_t(tables.size()); e.ntrns = uint32_t(it.trns.size());
        tables.insert(tables.end(), it.trns.begin(), it.trns.end());
        e.src_off = a->imgs[it.image].pix_off; e.dst_off = src_bytes;
        src_bytes += (uint64_t(it.width) * it.height * nc * bps + 255) & ~uint64_t(255);
        e.wide = wide && it.has_trns && (it.ctype == 0 || it.ctype == 2) ? 1u : 0u;   // the colour key becomes a 16-bit alpha sample
        if (wide && !e.wide) copies.push_back(RawCopy{e.dst_off, e.src_off, uint64_t(it.height) * it.rowbytes});   // nothing to expand
        else { max_h = std::max(max_h, it.height); ejobs.push_back(e); }
        bits[i] = uint8_t(8 * bps);
        PngResize j{};
        j.width = it.width; j.height = it.height; j.nc = nc; j.nw = uint32_t(nw); j.nh = uint32_t(nh); j.bps = bps;
        j.src_off = e.dst_off; j.tmp_off = tmp_floats; j.dst_off = dst_bytes;
        const bool same = uint32_t(nw) == it.width && uint32_t(nh) == it.height;   // image-rs copies instead of resampling
        j.vtap_base = uint32_t(taps.size()); csh_lanczos_axis(int(it.height), nh, same, taps, weights);
        j.htap_base = uint32_t(taps.size()); csh_lanczos_axis(int(it.width), nw, same, taps, weights);
        tmp_floats += (tmpn + 63) & ~uint64_t(63); dst_bytes += (dstn * bps + 255) & ~uint64_t(255);
        max_tmp = std::max(max_tmp, tmpn); max_dst = std::max(max_dst, dstn);
        px[i].width = uint32_t(nw); px[i].height = uint32_t(nh); px[i].channels = nc;
        jobs.push_back(j); job_item.push_back(i);
    }
    DevBuf<PngResize> d_jobs;
    DevBuf<csh::ResizeTap> d_taps;
    DevBuf<float> d_weights, d_tmp;
    DevBuf<uint8_t> d_dst, d_src, d_tables;
    DevBuf<RgbJob> d_ejobs;
    if (!jobs.empty()) {
        if (d_jobs.upload(jobs, st) || d_taps.upload(taps, st) || d_weights.upload(weights, st) || d_tmp.alloc((png_resize_is_fused(jobs.data(), int(jobs.size())) ? 0 : tmp_floats) + 64) || d_dst.alloc(dst_bytes + 256) ||
            d_ejobs.upload(ejobs, st) || d_tables.upload(tables, st) || d_src.alloc(src_bytes + 256)) return CS_ERR_NO_DEVICE;
        launch_png_rgb(st, d_ejobs.p, int(ejobs.size()), max_h, d_tables.p, a->d_work.p, d_src.p, a->d_status.p);
        for (const RawCopy &c : copies)
            if (hipMemcpyAsync(d_src.p + c.dst, a->d_work.p + c.src, c.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { csh_set_error("pixel copy failed"); return CS_ERR_NO_DEVICE; }
        launch_png_resize(st, d_jobs.p, jobs.data(), int(jobs.size()), d_taps.p, d_weights.p, d_src.p, d_tmp.p, d_dst.p, max_tmp, max_dst);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { csh_set_error("PNG resize kernels failed"); return CS_ERR_NO_DEVICE; }
        for (size_t k = 0; k < jobs.size(); k++) px[job_item[k]].device_pixels = d_dst.p + jobs[k].dst_off;
    }
    CCSParameters q = *p;
    q.width = 0; q.height = 0;
    return png_create(nullptr, px.data(), count, &q, device, mode, out, &pre, &bits);   // copies the pixels before d_dst goes out of scope
}

static CCSResult png_result(int code, const char *msg) {
    CCSResult r;
    r.success = code == 0; r.code = uint32_t(code); r.error_message = nullptr;
    if (code && msg) { size_t n = strlen(msg); char *m = (char *)malloc(n + 1); memcpy(m, msg, n + 1); r.error_message = m; }
    return r;
}

// PNG -> JPEG (convert_in_memory to JPEG, /root/reference/src/compressor.rs:289-299): a decode-only batch, the pixels as 8-bit grey or
// RGB (k_png_rgb: palette looked up, 16-bit narrowed, sub-byte grey scaled; an alpha channel or tRNS is dropped, as image-rs's JPEG
// encoder does [UPSTREAM-RECALL]), then the JPEG batch object from those pixels (csh_batch_create_from_pixels: its resize honours
// width / height, its encoder p's JPEG parameters).  Device to device; results in input order.
static void put_le32(uint8_t *p, uint32_t v) { p[0] = uint8_t(v); p[1] = uint8_t(v >> 8); p[2] = uint8_t(v >> 16); p[3] = uint8_t(v >> 24); }
// PNG -> JPEG and PNG -> lossless WebP share everything up to the pixels: decode (any PNG format), 8-bit grey / RGB in device memory.  The lossless
// WebP target keeps an alpha channel / tRNS chunk as the picture's alpha (grey + alpha / RGBA pixels) and sends the pixels -- resized first when a size is given,
// through the JPEG row's resize branch stopped behind its RGB -- to the VP8L coder (csl_encode_pixels).
static int png_to_pixels_then(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results, bool lossless_webp) {
    for (size_t i = 0; i < count; i++) { outputs[i].data = nullptr; outputs[i].length = 0; }
    auto fail_all = [&](int rc) { for (size_t i = 0; i < count; i++) if (results) results[i] = png_result(rc, csh_last_error()); return int(count); };
    CCSParameters q = *p;
    q.width = 0; q.height = 0;
    csp_batch *raw = nullptr;
    int rc = png_create(inputs, nullptr, count, &q, device, MODE_DECODE_ANY, &raw);
    std::unique_ptr<csp_batch> a(raw);
    if (rc == 0) rc = csp_batch_run(a.get(), nullptr);
    if (rc) return fail_all(rc);
    hipStream_t st = a->stream;
    const int nimg = int(a->imgs.size());
    std::vector<uint32_t> status(size_t(nimg) + 1, 0);
    if (nimg && csh_copy_wait(status.data(), a->d_status.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, a->stream) != hipSuccess) { csh_set_error("download failed"); return fail_all(CS_ERR_NO_DEVICE); }
    std::vector<RgbJob> ejobs;
    std::vector<uint8_t> tables(1, 0);
    std::vector<size_t> at;
    std::vector<csp_pixels> px;
    uint64_t src_bytes = 0;
    uint32_t max_h = 0;
    int failed = 0;
    for (size_t i = 0; i < count; i++) {
        const PngItem &it = a->items[i];
        int code = it.code;
        const char *msg = it.msg.c_str();
        if (!code && status[it.image]) { code = int(status[it.image]); msg = "malformed PNG data"; }
        if (!code && !lossless_webp && (it.width > 65535 || it.height > 65535)) { code = CS_ERR_UNSUPPORTED; msg = "image too large for a JPEG"; }
        if (!code && lossless_webp && (it.width > 16384 || it.height > 16384)) { code = CS_ERR_UNSUPPORTED; msg = "image too large for a WebP"; }
        const bool transparent = it.ctype == 4 || it.ctype == 6 || it.has_trns;
        if (code) { if (results) results[i] = png_result(code, msg); failed++; continue; }
        RgbJob e{};
        e.image = uint32_t(it.image); e.width = it.width; e.height = it.height; e.rowbytes = it.rowbytes; e.ctype = it.ctype; e.depth = it.depth;
        e.out_nc = ((it.ctype == 2 || it.ctype == 6 || it.ctype == 3) ? 3u : 1u) + ((lossless_webp && transparent) ? 1u : 0u);   // a JPEG drops the alpha; a lossless WebP keeps it
        e.plte_off = uint32_t(tables.size()); e.npal = uint32_t(it.plte.size() / 3);
        tables.insert(tables.end(), it.plte.begin(), it.plte.end());
        e.trns_off = uint32_t(tables.size()); e.ntrns = uint32_t(it.trns.size());
        tables.insert(tables.end(), it.trns.begin(), it.trns.end());
        e.src_off = a->imgs[it.image].pix_off; e.dst_off = src_bytes;
        src_bytes += (uint64_t(it.width) * it.height * e.out_nc + 255) & ~uint64_t(255);
        max_h = std::max(max_h, it.height);
        px.push_back(csp_pixels{nullptr, it.width, it.height, e.out_nc});   // the pointer is known once the buffer is
        ejobs.push_back(e); at.push_back(i);
    }
    if (px.empty()) return failed;
    auto fail_rest = [&](int code) { for (size_t k : at) if (results) results[k] = png_result(code, csh_last_error()); return failed + int(at.size()); };   // the files that were still good
    DevBuf<RgbJob> d_ejobs;
    DevBuf<uint8_t> d_tables, d_src;
    if (d_ejobs.upload(ejobs, st) || d_tables.upload(tables, st) || d_src.alloc(src_bytes + 256)) return fail_rest(CS_ERR_NO_DEVICE);
    launch_png_rgb(st, d_ejobs.p, int(ejobs.size()), max_h, d_tables.p, a->d_work.p, d_src.p, a->d_status.p);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { csh_set_error("PNG kernels failed"); return fail_rest(CS_ERR_NO_DEVICE); }
    for (size_t k = 0; k < px.size(); k++) px[k].device_pixels = d_src.p + ejobs[k].dst_off;
    if (lossless_webp) {
        std::vector<csp_pixels> src = px;
        csh_batch *rb = nullptr;
        DevBuf<PngResize> d_jobs;
        DevBuf<csh::ResizeTap> d_taps;
        DevBuf<float> d_weights, d_tmp;
        DevBuf<uint8_t> d_dst;
        if (p->width || p->height) {
            // opaque pictures (grey, RGB): the JPEG row's resize branch, stopped behind its pixels; pictures with transparency (grey + alpha, RGBA): the PNG
            // row's own two Lanczos passes over the interleaved samples (k_png_resize.hip, the same arithmetic; image-rs resamples the channels alike)
            std::vector<csp_pixels> opaque;
            std::vector<size_t> opaque_at, job_at;
            std::vector<PngResize> jobs;
            std::vector<csh::ResizeTap> taps;
            std::vector<float> weights;
            uint64_t tmp_floats = 0, dst_bytes = 0, max_tmp = 0, max_dst = 0;
            for (size_t k = 0; k < px.size(); k++) {
                if (px[k].channels == 1 || px[k].channels == 3) { opaque.push_back(px[k]); opaque_at.push_back(k); continue; }
                int nw = 0, nh = 0;
                csh_compute_dimensions(int(px[k].width), int(px[k].height), int(p->width), int(p->height), nw, nh);
                const uint32_t nc = px[k].channels;
                const uint64_t tmpn = uint64_t(nh) * px[k].width * nc, dstn = uint64_t(nw) * nh * nc;
                if (uint64_t(nw) * nc > 0x7FFFFFF0u || tmpn > 0xFFFFFF00u || dstn > 0xFFFFFF00u) { csh_set_error("resized PNG too large for one device batch"); rc = CS_ERR_UNSUPPORTED; break; }
                PngResize j{};
                j.width = px[k].width; j.height = px[k].height; j.nc = nc; j.nw = uint32_t(nw); j.nh = uint32_t(nh); j.bps = 1;
                j.src_off = ejobs[k].dst_off; j.tmp_off = tmp_floats; j.dst_off = dst_bytes;
                const bool same = uint32_t(nw) == px[k].width && uint32_t(nh) == px[k].height;
                j.vtap_base = uint32_t(taps.size()); csh_lanczos_axis(int(px[k].height), nh, same, taps, weights);
                j.htap_base = uint32_t(taps.size()); csh_lanczos_axis(int(px[k].width), nw, same, taps, weights);
                tmp_floats += (tmpn + 63) & ~uint64_t(63); dst_bytes += (dstn + 255) & ~uint64_t(255);
                max_tmp = std::max(max_tmp, tmpn); max_dst = std::max(max_dst, dstn);
                src[k].width = uint32_t(nw); src[k].height = uint32_t(nh);
                jobs.push_back(j); job_at.push_back(k);
            }
            if (rc == 0 && !opaque.empty()) {
                rc = csh_batch_create_from_pixels_rgb(opaque.data(), opaque.size(), p, device, &rb);
                if (rc == 0) rc = csh_batch_run(rb, nullptr);
                for (size_t j = 0; j < opaque.size() && rc == 0; j++) { const char *m = ""; csp_pixels &d = src[opaque_at[j]]; if (csh_batch_pixels(rb, j, &d.device_pixels, &d.width, &d.height, &d.channels, &m)) rc = CS_ERR_NO_DEVICE; }
            }
            if (rc == 0 && !jobs.empty()) {
                if (d_jobs.upload(jobs, st) || d_taps.upload(taps, st) || d_weights.upload(weights, st) || d_tmp.alloc((png_resize_is_fused(jobs.data(), int(jobs.size())) ? 0 : tmp_floats) + 64) || d_dst.alloc(dst_bytes + 256)) rc = CS_ERR_NO_DEVICE;
                else {
                    launch_png_resize(st, d_jobs.p, jobs.data(), int(jobs.size()), d_taps.p, d_weights.p, d_src.p, d_tmp.p, d_dst.p, max_tmp, max_dst);
                    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { csh_set_error("PNG resize kernels failed"); rc = CS_ERR_NO_DEVICE; }
                    for (size_t j = 0; j < jobs.size(); j++) src[job_at[j]].device_pixels = d_dst.p + jobs[j].dst_off;
                }
            }
        }
        std::vector<CByteArray> out(px.size());
        std::vector<CCSResult> res(px.size());
        const int lf = rc ? -1 : csl_encode_pixels(src.data(), src.size(), device, out.data(), res.data());
        for (size_t k = 0; k < px.size(); k++) {
            if (lf < 0) { if (results) results[at[k]] = png_result(rc ? rc : CS_ERR_NO_DEVICE, csh_last_error()); continue; }
            outputs[at[k]] = out[k];
            if (results) results[at[k]] = res[k]; else cs_free_result(&res[k]);
        }
        csh_batch_destroy(rb);
        return failed + (lf < 0 ? int(px.size()) : lf);
    }
    csh_batch *jb = nullptr;
    rc = csh_batch_create_from_pixels(px.data(), px.size(), p, device, &jb);
    if (rc == 0) rc = csh_batch_run(jb, nullptr);
    std::vector<CByteArray> out(px.size());
    std::vector<CCSResult> res(px.size());
    int jf = rc ? -1 : csh_batch_fetch(jb, out.data(), res.data());
    for (size_t k = 0; k < px.size(); k++) {
        if (jf < 0) { if (results) results[at[k]] = png_result(rc ? rc : CS_ERR_NO_DEVICE, csh_last_error()); continue; }
        outputs[at[k]] = out[k];
        if (results) results[at[k]] = res[k]; else cs_free_result(&res[k]);
    }
    csh_batch_destroy(jb);
    return failed + (jf < 0 ? int(px.size()) : jf);
}
extern "C" int csp_png_to_jpeg(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results) {
    return png_to_pixels_then(inputs, count, p, device, outputs, results, false);
}
extern "C" int csp_png_to_lossless_webp(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results) {
    return png_to_pixels_then(inputs, count, p, device, outputs, results, true);
}


extern "C" int csp_batch_fetch(csp_batch *b, CByteArray *outputs, CCSResult *results) {
    if (!b || !b->ran || b->decode_only) { csh_set_error("csp_batch_fetch before csp_batch_run"); return -1; }
    if (hipSetDevice(b->device) != hipSuccess) return -1;
    const int nimg = int(b->imgs.size());
    std::vector<uint32_t> status(size_t(nimg) + 1), flen(size_t(nimg) + 1);
    if (nimg) {
        if (csh_copy_wait(status.data(), b->d_status.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, b->stream) != hipSuccess ||
            csh_copy_wait(flen.data(), b->d_file_len.p, sizeof(uint32_t) * nimg, hipMemcpyDeviceToHost, b->stream) != hipSuccess) { csh_set_error("download failed"); return -1; }
    }
    int failed = 0;
    // pictures with transparency going to WebP: their alpha plane through the VP8L coder now (the pixels are still in d_rgb), one call for all of them
    std::vector<CByteArray> alpha_out;
    std::vector<CCSResult> alpha_res;
    std::vector<int> alpha_at(b->items.size(), -1);
    std::vector<csp_pixels> alpha_px;
    if (b->to_webp) {
        for (size_t i = 0; i < b->items.size(); i++) {
            const PngItem &it = b->items[i];
            if (it.code || it.image < 0 || status[it.image] || b->h_wstatus[it.image] || size_t(it.image) >= b->walpha.size() || !b->walpha[it.image]) continue;
            const csw::WebpImg *wi = nullptr;
            for (const csw::WebpImg &w : b->wimgs) if (int(w.image) == it.image) { wi = &w; break; }
            if (!wi) continue;
            alpha_at[i] = int(alpha_px.size());
            alpha_px.push_back(csp_pixels{b->d_rgb.p + wi->rgb_off, wi->width, wi->height, uint32_t(csw::VP8L_ALPHA_OF) + b->walpha[it.image]});
        }
        if (!alpha_px.empty()) {
            alpha_out.resize(alpha_px.size()); alpha_res.resize(alpha_px.size());
            csl_encode_pixels(alpha_px.data(), alpha_px.size(), b->device, alpha_out.data(), alpha_res.data());
        }
    }
    for (size_t i = 0; i < b->items.size(); i++) {
        PngItem &it = b->items[i];
        outputs[i].data = nullptr; outputs[i].length = 0;
        int code = it.code;
        const char *msg = it.msg.c_str();
        if (!code && alpha_at[i] >= 0 && !alpha_out[size_t(alpha_at[i])].data) { code = int(alpha_res[size_t(alpha_at[i])].code ? alpha_res[size_t(alpha_at[i])].code : CS_ERR_NO_DEVICE); msg = "alpha plane coder failed"; }
        if (!code && !status[it.image] && b->to_webp && b->h_wstatus[it.image]) { code = int(b->h_wstatus[it.image]); msg = "WebP encoder failed"; }
        else if (!code && status[it.image]) { code = int(status[it.image]); msg = code == int(CSP_ERR_POOL) ? "internal device pool too small" : "malformed PNG data"; }
        if (code) { failed++; if (results) results[i] = png_result(code, msg); continue; }
        const size_t n = flen[it.image];
        if (!b->lossy && !b->to_webp && !b->from_pixels && n >= it.file_size) {   // oxipng: "file already optimized" -- the input comes back unchanged
            outputs[i].data = (uint8_t *)malloc(it.file_size ? it.file_size : 1);
            memcpy(outputs[i].data, b->inputs[i], it.file_size);
            outputs[i].length = it.file_size;
        } else {
            outputs[i].data = (uint8_t *)malloc(n ? n : 1);
            if (csh_copy_wait(outputs[i].data, b->d_out.p + b->imgs[it.image].out_off, n, hipMemcpyDeviceToHost, b->stream) != hipSuccess) { csh_set_error("download failed"); return -1; }
            outputs[i].length = n;
        }
        if (alpha_at[i] >= 0) {
            const csp_pixels &ap = alpha_px[size_t(alpha_at[i])];
            if (csl_attach_alpha(&outputs[i], &alpha_out[size_t(alpha_at[i])], ap.width, ap.height)) {
                cs_free_bytes(&outputs[i]); failed++;
                if (results) results[i] = png_result(CS_ERR_NO_DEVICE, "could not assemble the WebP file with its alpha plane");
                continue;
            }
        }
        if (results) results[i] = png_result(0, nullptr);
    }
    for (size_t k = 0; k < alpha_out.size(); k++) { cs_free_bytes(&alpha_out[k]); cs_free_result(&alpha_res[k]); }
    return failed;
}

// ---- stage taps
static const PngImg *tap_image(csp_batch *b, size_t image) {
    if (!b || !b->ran || image >= b->items.size() || b->items[image].image < 0) { csh_set_error("no such decoded PNG in the batch"); return nullptr; }
    return &b->imgs[b->items[image].image];
}
extern "C" int csp_batch_geometry(csp_batch *b, size_t image, uint32_t *width, uint32_t *height, uint32_t *rowbytes) {
    const PngImg *im = tap_image(b, image);
    if (!im) return -1;
    *width = im->width; *height = im->height; *rowbytes = im->rowbytes;
    return 0;
}
extern "C" int csp_batch_read_rows(csp_batch *b, size_t image, uint8_t *dst) {
    const PngImg *im = tap_image(b, image);
    if (!im) return -1;
    return csh_copy_wait(dst, b->d_work.p + im->pix_off, size_t(im->height) * im->rowbytes, hipMemcpyDeviceToHost, b->stream) == hipSuccess ? 0 : -1;
}
extern "C" int csp_batch_read_stream(csp_batch *b, size_t image, int strategy, uint8_t *dst) {
    const PngImg *im = tap_image(b, image);
    if (!im) return -1;
    if (strategy < 0 || strategy > 9 || b->slot_of_strategy[strategy] < 0) { csh_set_error("strategy %d is not part of this level's plan", strategy); return -1; }
    return csh_copy_wait(dst, b->d_streams.p + im->stream_off + uint64_t(b->slot_of_strategy[strategy]) * im->stream_stride, im->raw_len, hipMemcpyDeviceToHost, b->stream) == hipSuccess ? 0 : -1;
}
extern "C" int csp_batch_read_scores(csp_batch *b, size_t image, uint64_t *dst, int *have) {
    const PngImg *im = tap_image(b, image);
    if (!im) return -1;
    *have = 0;
    for (int a = 0; a < b->plan.nadaptive; a++) *have |= b->plan.adaptive_strategy[a] == 9 ? 16 : 15;
    return csh_copy_wait(dst, b->d_scores.p + size_t(im->row_base) * 25, sizeof(uint64_t) * 25 * im->height, hipMemcpyDeviceToHost, b->stream) == hipSuccess ? 0 : -1;
}
extern "C" int csp_batch_trials(csp_batch *b, size_t image, int *strategies, uint64_t *zlib_bytes, int *ntrials, int *winner) {
    const PngImg *im = tap_image(b, image);
    if (!im) return -1;
    const int idx = b->items[image].image;
    *ntrials = b->plan.ntrials;
    for (int t = 0; t < b->plan.ntrials; t++) strategies[t] = b->plan.trial_strategy[t];
    int32_t w = 0;
    if (csh_copy_wait(zlib_bytes, b->d_trial_bytes.p + size_t(idx) * CSP_MAX_STREAMS, sizeof(uint64_t) * b->plan.ntrials, hipMemcpyDeviceToHost, b->stream) != hipSuccess ||
        csh_copy_wait(&w, b->d_winner.p + idx, sizeof w, hipMemcpyDeviceToHost, b->stream) != hipSuccess) return -1;
    *winner = w;
    return 0;
}
extern "C" int csp_batch_chunk_bits(csp_batch *b, size_t image, int trial, uint64_t *dst, size_t cap, size_t *nchunks) {
    const PngImg *im = tap_image(b, image);
    if (!im || trial < 0 || trial >= b->plan.ntrials) return -1;
    *nchunks = im->nchunks;
    std::vector<PngChunk> recs(im->nchunks);
    if (csh_copy_wait(recs.data(), b->d_chunks.p + size_t(im->chunk_base) + size_t(b->plan.trial_slot[trial]) * im->nchunks, sizeof(PngChunk) * im->nchunks, hipMemcpyDeviceToHost, b->stream) != hipSuccess) return -1;
    for (size_t i = 0; i < im->nchunks && i < cap; i++) dst[i] = recs[i].bits;
    return 0;
}
