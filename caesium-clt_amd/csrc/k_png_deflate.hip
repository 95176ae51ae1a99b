// k_png_deflate.hip -- row P4 of SURVEY.md 8a: DEFLATE of the filtered streams (statement: oracle/png_oracle.c
// deflate_chunk() / cso_deflate_zlib() / cso_png_optimize()).
// The reference spends >90 % of this path in libdeflate's serial optimal parser, once per filter trial.  Here a stream
// is cut into 32 KiB chunks that are coded independently -- one dynamic-Huffman block each, byte-aligned by an empty
// stored block (the sync marker pigz uses) -- so the unit of parallelism is (image, trial, chunk):
//   hist   one wave per (trial, chunk): tokenizer (png_lz.h), symbol counts into LDS
//   codes  one lane per (trial, chunk): code lengths, canonical codes, header run-lengths, exact block size
//   choose one lane per image: stream size of every trial, the winner, byte offset of each of its chunks
//   emit   one wave per chunk of the winner: tokenizer again, bits through an LDS window straight to their final place
//   finish Adler-32, CRC-32, IDAT framing, carried chunks
// Only sizes decide the winner, so the losing trials are never packed.
#include "png_emit.h"
#include "png_parse.h"

namespace csp {

// ------------------------------------------------------------------------------------------------ hist
struct HistLds { LzLds lz; uint32_t hist[CSP_NSYM]; };
struct HistSink {
    uint32_t *hist;
    LV<uint32_t> extra;
    __device__ __forceinline__ void tile(uint64_t, uint32_t, uint64_t taken, const LV<uint32_t> &mlen, const LV<uint32_t> &mdist, const LV<uint32_t> &lit) {
        LFOR(l) if ((taken >> l) & 1) {
            if (mlen[l]) {
                const uint32_t lc = len_code_of(mlen[l]), dc = dist_code_of(mdist[l]);
                atomicAdd(&hist[257 + lc], 1u); atomicAdd(&hist[CSP_NLIT + dc], 1u);
                extra[l] += len_extra_of(lc) + dist_extra_of(dc);
            } else
                atomicAdd(&hist[lit[l]], 1u);
        }
    }
};
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_png_hist(DeflateCtx c) {
    CSH_SHARED HistLds S;
    const uint32_t bg = blockIdx.x, trial = blockIdx.y;
    const uint32_t image = c.group_image[bg];
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint32_t c0 = (bg - c.group_first[image]) * CSP_GROUP, c1 = c0 + CSP_GROUP < im.nchunks ? c0 + CSP_GROUP : im.nchunks;
    const int slot = c.plan.trial_slot[trial];
    const uint8_t *data = c.streams + im.stream_off + uint64_t(slot) * im.stream_stride;
    for (uint32_t ci = c0; ci < c1; ci++) {
        const uint64_t start = uint64_t(ci) * CSP_CHUNK, end = start + CSP_CHUNK < im.raw_len ? start + CSP_CHUNK : im.raw_len;
        LFOR(l) for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) S.hist[i] = 0;
        HistSink sink; sink.hist = S.hist;
        LFOR(l) sink.extra[l] = 0;
        CSP_WAVE_SYNC();
        lz_chunk(data, im.raw_len, start, end, S.lz, sink);
        CSP_WAVE_SYNC();
        PngChunk &rec = chunk_rec(c, im, slot, ci);
        LFOR(l) for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) rec.freq[i] = i == 256 ? 1u : S.hist[i];
        LV<uint64_t> e;
        LFOR(l) e[l] = sink.extra[l];
        const uint64_t extra = lsum(e);
        // a chunk with enough matches in it gets the min-cost-path parse (k_png_deep_hist replaces these counts; oracle: deflate_chunk)
        LV<uint64_t> nm, nt;
        LFOR(l) { nm[l] = 0; nt[l] = 0; for (uint32_t i = uint32_t(l); i < CSP_NLIT; i += 64) { nt[l] += S.hist[i]; if (i > 256) nm[l] += S.hist[i]; } }
        const uint64_t matches = lsum(nm), tokens = lsum(nt);
        LFOR(l) if (l == 0) { rec.extra_bits = uint32_t(extra); rec.deep = (c.deep_iters > 0 && matches * CSP_DEEP_DIV >= tokens) ? 1u : 0u; }
        CSP_WAVE_SYNC();
    }
}

// ------------------------------------------------------------------------------------------------ codes
__global__ void __launch_bounds__(64) k_png_codes(DeflateCtx c, int only_deep) {   // only_deep: the records k_png_deep_hist has rewritten, from the list it left
    uint32_t bc = blockIdx.x * blockDim.x + threadIdx.x, trial = blockIdx.y;
    if (only_deep) {   // (a lane per LISTED record: lanes that skipped unlisted records wave by wave paid for the listed ones' code construction all the same)
        const uint32_t i = bc + trial * gridDim.x * blockDim.x;
        if (i >= c.deep_queue[2]) return;
        const uint32_t item = c.deep_list[i];
        trial = item / c.total_chunks; bc = item % c.total_chunks;
    }
    if (bc >= c.total_chunks) return;
    const uint32_t image = c.chunk_image[bc];
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint32_t ci = bc - c.chunk_first[image];
    PngChunk &rec = chunk_rec(c, im, c.plan.trial_slot[trial], ci);
    if (only_deep && !rec.deep) return;
    code_lengths(rec.freq, CSP_NLIT, 15, rec.len);
    code_lengths(rec.freq + CSP_NLIT, CSP_NDIST, 15, rec.len + CSP_NLIT);
    canonical(rec.len, CSP_NLIT, rec.code);
    canonical(rec.len + CSP_NLIT, CSP_NDIST, rec.code + CSP_NLIT);
    int nl = CSP_NLIT, nd = CSP_NDIST;
    while (nl > 257 && !rec.len[nl - 1]) nl--;
    while (nd > 1 && !rec.len[CSP_NLIT + nd - 1]) nd--;
    // run-length symbols over the concatenated lengths (runs may cross from one alphabet into the other)
    int m = 0;
    const int total = nl + nd;
    auto at = [&](int i) -> int { return i < nl ? rec.len[i] : rec.len[CSP_NLIT + (i - nl)]; };
    for (int i = 0; i < total;) {
        const int v = at(i);
        int r = 1;
        while (i + r < total && at(i + r) == v) r++;
        i += r;
        if (v == 0) {
            while (r >= 11) { const int t = r > 138 ? 138 : r; rec.hdr_sym[m] = 18; rec.hdr_extra[m++] = uint8_t(t - 11); r -= t; }
            if (r >= 3) { rec.hdr_sym[m] = 17; rec.hdr_extra[m++] = uint8_t(r - 3); r = 0; }
            while (r--) { rec.hdr_sym[m] = 0; rec.hdr_extra[m++] = 0; }
        } else {
            rec.hdr_sym[m] = uint8_t(v); rec.hdr_extra[m++] = 0; r--;
            while (r >= 3) { const int t = r > 6 ? 6 : r; rec.hdr_sym[m] = 16; rec.hdr_extra[m++] = uint8_t(t - 3); r -= t; }
            while (r--) { rec.hdr_sym[m] = uint8_t(v); rec.hdr_extra[m++] = 0; }
        }
    }
    uint32_t cf[CSP_NCL];
    for (int i = 0; i < int(CSP_NCL); i++) cf[i] = 0;
    for (int i = 0; i < m; i++) cf[rec.hdr_sym[i]]++;
    code_lengths(cf, CSP_NCL, 7, rec.cl_len);
    canonical(rec.cl_len, CSP_NCL, rec.cl_code);
    int ncl = CSP_NCL;
    while (ncl > 4 && !rec.cl_len[kClOrder[ncl - 1]]) ncl--;
    rec.hlit = uint16_t(nl); rec.hdist = uint16_t(nd); rec.hclen = uint16_t(ncl); rec.nhdr = uint16_t(m);
    uint64_t bits = 3 + 5 + 5 + 4 + 3 * uint64_t(ncl);
    for (int i = 0; i < m; i++) { const int s = rec.hdr_sym[i]; bits += rec.cl_len[s] + (s == 16 ? 2 : s == 17 ? 3 : s == 18 ? 7 : 0); }
    for (int i = 0; i < int(CSP_NSYM); i++) bits += uint64_t(rec.freq[i]) * rec.len[i];
    bits += rec.extra_bits;
    rec.bits = bits;
    const bool last = ci + 1 == im.nchunks;
    rec.bytes = last ? uint32_t((bits + 7) >> 3) : uint32_t((bits + 3 + 7) >> 3) + 4u;
}

// ------------------------------------------------------------------------------------------------ choose
__global__ void __launch_bounds__(64) k_png_choose(DeflateCtx c) {
    const int image = blockIdx.x * blockDim.x + threadIdx.x;
    if (image >= c.nimg || c.status[image]) return;
    const PngImg &im = c.imgs[image];
    int best = 0;
    uint64_t best_bytes = ~0ull;
    for (int t = 0; t < c.plan.ntrials; t++) {
        uint64_t bytes = 2 + 4;
        for (uint32_t ci = 0; ci < im.nchunks; ci++) bytes += chunk_rec(c, im, c.plan.trial_slot[t], ci).bytes;
        c.trial_bytes[uint64_t(image) * CSP_MAX_STREAMS + t] = bytes;
        if (bytes < best_bytes) { best_bytes = bytes; best = t; }
    }
    // which trials take the min-cost-path parse (read by k_png_deep_hist behind the first of the two calls; oracle: png_recode)
    for (int t = 0; t < c.plan.ntrials; t++)
        c.trial_live[uint64_t(image) * CSP_MAX_STREAMS + t] = c.trial_bytes[uint64_t(image) * CSP_MAX_STREAMS + t] * CSP_DEEP_LIVE_DEN <= best_bytes * CSP_DEEP_LIVE_NUM ? 1 : 0;
    c.winner[image] = best;
    const uint64_t file_len = uint64_t(im.prefix_len) + 12 + best_bytes + im.suffix_len;
    if (file_len > im.out_cap) { c.status[image] = CSP_ERR_POOL; return; }
    c.file_len[image] = uint32_t(file_len);
}

// ------------------------------------------------------------------------------------------------ emit
// bits are OR-ed into a window of LDS words; complete words leave for HBM after every tile (png_emit.h)
struct EmitLds { LzLds lz; uint32_t code[CSP_NSYM]; uint32_t win[160]; };
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_png_emit(DeflateCtx c) {
    CSH_SHARED EmitLds S;
    const uint32_t bg = blockIdx.x;
    const uint32_t image = c.group_image[bg];
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint32_t c0 = (bg - c.group_first[image]) * CSP_GROUP, c1 = c0 + CSP_GROUP < im.nchunks ? c0 + CSP_GROUP : im.nchunks;
    const int slot = c.plan.trial_slot[c.winner[image]];
    const uint8_t *data = c.streams + im.stream_off + uint64_t(slot) * im.stream_stride;
    // where the group's first chunk goes: after the zlib header and the chunks in front of it
    uint64_t at = uint64_t(im.prefix_len) + 8 + 2;
    {
        LV<uint64_t> part;
        LFOR(l) { uint64_t s = 0; for (uint32_t k = uint32_t(l); k < c0; k += 64) s += chunk_rec(c, im, slot, k).bytes; part[l] = s; }
        at += lsum(part);
    }
    bool ok = true;
    for (uint32_t ci = c0; ci < c1; ci++) {
        const PngChunk &rec = chunk_rec(c, im, slot, ci);
        const bool last = ci + 1 == im.nchunks;
        if (!rec.deep) {   // (the others: k_png_deep_emit)
            BitOut bo;
            emit_block_begin(rec, last, S.code, S.win, c.out + im.out_off + at, bo);
            EmitSink sink; sink.code = S.code; sink.bo = &bo;
            const uint64_t start = uint64_t(ci) * CSP_CHUNK, end = start + CSP_CHUNK < im.raw_len ? start + CSP_CHUNK : im.raw_len;
            lz_chunk(data, im.raw_len, start, end, S.lz, sink);
            if (!emit_block_end(rec, last, S.code, bo)) ok = false;   // the size pass and the emit pass disagree: never ship it
        }
        at += rec.bytes;
        CSP_WAVE_SYNC();
    }
    if (!ok) LFOR(l) if (l == 0) c.status[image] = CSP_ERR_POOL;
}

// ------------------------------------------------------------------------------------------------ finish
// Adler-32 parts of the winner's chunks: (sum of bytes, sum of (n - i) * byte) per chunk, modulo 65521
__global__ void __launch_bounds__(256) k_png_adler(DeflateCtx c) {
    CSH_SHARED unsigned long long acc[2];
    const uint32_t bc = blockIdx.x;
    const uint32_t image = c.chunk_image[bc];
    const PngImg &im = c.imgs[image];
    const uint32_t ci = bc - c.chunk_first[image];
    const bool dead = c.status[image] != 0;
    CSH_PHASE_LOOP(3) {
        if (phase == 0) { if (threadIdx.x == 0) { acc[0] = 0; acc[1] = 0; } continue; }
        if (dead) continue;
        if (phase == 1) {
            const int slot = c.plan.trial_slot[c.winner[image]];
            const uint8_t *data = c.streams + im.stream_off + uint64_t(slot) * im.stream_stride;
            const uint64_t start = uint64_t(ci) * CSP_CHUNK, end = start + CSP_CHUNK < im.raw_len ? start + CSP_CHUNK : im.raw_len;
            const uint32_t n = uint32_t(end - start);
            unsigned long long s1 = 0, s2 = 0;
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { const uint32_t b = data[start + i]; s1 += b; s2 += uint64_t(n - i) * b; }
            if (s1) { atomicAdd(&acc[0], s1); atomicAdd(&acc[1], s2); }
            continue;
        }
        if (threadIdx.x == 0) { c.adler_parts[2 * uint64_t(bc)] = uint32_t(acc[0] % 65521u); c.adler_parts[2 * uint64_t(bc) + 1] = uint32_t(acc[1] % 65521u); }
    }
}
// CRC-32 of the IDAT chunk in KiB pieces (type field + zlib stream), one lane per piece; the pieces are folded by k_png_finish
enum { CRC_PIECE = 1024 };
__device__ __forceinline__ static uint32_t crc_byte_table(uint32_t n) { uint32_t v = n; for (int k = 0; k < 8; k++) v = (v & 1u) ? 0xEDB88320u ^ (v >> 1) : v >> 1; return v; }
__device__ static uint32_t gf2_mul(uint32_t a, uint32_t b) {   // a * b modulo the CRC polynomial (zlib multmodp)
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
__device__ static uint32_t gf2_x_pow_bytes(uint64_t nbytes) {   // x^(8 * nbytes)
    uint32_t sq = 1u << 30;   // x^1
    for (int k = 0; k < 3; k++) sq = gf2_mul(sq, sq);   // x^8
    uint32_t p = 1u << 31;    // x^0
    while (nbytes) { if (nbytes & 1) p = gf2_mul(sq, p); sq = gf2_mul(sq, sq); nbytes >>= 1; }
    return p;
}
__global__ void __launch_bounds__(256) k_png_crc_pieces(DeflateCtx c, uint32_t max_pieces) {
    CSH_SHARED uint32_t table[256];
    const int image = blockIdx.y;
    const PngImg &im = c.imgs[image];
    const bool dead = c.status[image] != 0;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) { table[threadIdx.x] = crc_byte_table(threadIdx.x); continue; }
        if (dead) continue;
        const uint64_t n = uint64_t(c.file_len[image]) - im.prefix_len - im.suffix_len - 8;   // "IDAT" + stream
        const uint32_t piece = blockIdx.x * blockDim.x + threadIdx.x;
        if (uint64_t(piece) * CRC_PIECE >= n) continue;
        const uint8_t *p = c.out + im.out_off + im.prefix_len + 4 + uint64_t(piece) * CRC_PIECE;
        const uint32_t m = n - uint64_t(piece) * CRC_PIECE < CRC_PIECE ? uint32_t(n - uint64_t(piece) * CRC_PIECE) : uint32_t(CRC_PIECE);
        uint32_t crc = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < m; i++) crc = table[(crc ^ p[i]) & 255u] ^ (crc >> 8);
        c.crc_parts[uint64_t(image) * max_pieces + piece] = ~crc;
    }
}
// per image: zlib header and Adler-32, IDAT length / type / CRC, the carried chunks in front and behind
__global__ void __launch_bounds__(64) k_png_frame(DeflateCtx c) {
    const int image = blockIdx.x;
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    uint8_t *o = c.out + im.out_off;
    const uint32_t flen = c.file_len[image];
    const uint64_t zlen = uint64_t(flen) - im.prefix_len - im.suffix_len - 12;
    for (uint32_t i = threadIdx.x; i < im.prefix_len; i += blockDim.x) o[i] = c.fixed[im.fix_off + i];
    for (uint32_t i = threadIdx.x; i < im.suffix_len; i += blockDim.x) o[flen - im.suffix_len + i] = c.fixed[im.fix_off + im.prefix_len + i];
    if (threadIdx.x == 0) {
        uint8_t *d = o + im.prefix_len;
        d[0] = uint8_t(zlen >> 24); d[1] = uint8_t(zlen >> 16); d[2] = uint8_t(zlen >> 8); d[3] = uint8_t(zlen);
        d[4] = 'I'; d[5] = 'D'; d[6] = 'A'; d[7] = 'T';
        d[8] = 0x78; d[9] = 0xDA;
        uint64_t a = 1, b = 0;
        const uint32_t first = c.chunk_first[image];
        for (uint32_t ci = 0; ci < im.nchunks; ci++) {
            const uint64_t start = uint64_t(ci) * CSP_CHUNK, n = (start + CSP_CHUNK < im.raw_len ? start + CSP_CHUNK : im.raw_len) - start;
            b = (b + n % 65521u * a + c.adler_parts[2 * uint64_t(first + ci) + 1]) % 65521u;
            a = (a + c.adler_parts[2 * uint64_t(first + ci)]) % 65521u;
        }
        uint8_t *t = d + 8 + zlen - 4;
        t[0] = uint8_t(b >> 8); t[1] = uint8_t(b); t[2] = uint8_t(a >> 8); t[3] = uint8_t(a);
    }
}
// fold the piece CRCs: crc(A || B) = x^(8 |B|) * crc(A) + crc(B).  First 64 pieces (64 KiB) per lane, in place; then the
// groups of an image by one lane.
__global__ void __launch_bounds__(64) k_png_crc_fold1(DeflateCtx c, uint32_t max_pieces) {
    const int image = blockIdx.y;
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint64_t n = uint64_t(c.file_len[image]) - im.prefix_len - im.suffix_len - 8;
    const uint32_t pieces = uint32_t((n + CRC_PIECE - 1) / CRC_PIECE), g = blockIdx.x * blockDim.x + threadIdx.x;
    if (uint64_t(g) * 64 >= pieces) return;
    const uint32_t p0 = g * 64, p1 = p0 + 64 < pieces ? p0 + 64 : pieces;
    const uint32_t op = gf2_x_pow_bytes(CRC_PIECE);
    uint32_t *parts = c.crc_parts + uint64_t(image) * max_pieces;
    uint32_t crc = parts[p0];
    for (uint32_t p = p0 + 1; p < p1; p++) crc = gf2_mul(p + 1 == pieces ? gf2_x_pow_bytes(n - uint64_t(pieces - 1) * CRC_PIECE) : op, crc) ^ parts[p];
    parts[p0] = crc;
}
__global__ void __launch_bounds__(64) k_png_crc_fold2(DeflateCtx c, uint32_t max_pieces) {
    const int image = blockIdx.x * blockDim.x + threadIdx.x;
    if (image >= c.nimg || c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint64_t n = uint64_t(c.file_len[image]) - im.prefix_len - im.suffix_len - 8;
    const uint32_t pieces = uint32_t((n + CRC_PIECE - 1) / CRC_PIECE), groups = (pieces + 63) / 64;
    const uint32_t op = gf2_x_pow_bytes(uint64_t(CRC_PIECE) * 64);
    const uint32_t *parts = c.crc_parts + uint64_t(image) * max_pieces;
    uint32_t crc = 0;
    for (uint32_t g = 0; g < groups; g++) crc = gf2_mul(g + 1 == groups ? gf2_x_pow_bytes(n - uint64_t(g) * 64 * CRC_PIECE) : op, crc) ^ parts[uint64_t(g) * 64];
    uint8_t *t = c.out + im.out_off + im.prefix_len + 4 + n;
    t[0] = uint8_t(crc >> 24); t[1] = uint8_t(crc >> 16); t[2] = uint8_t(crc >> 8); t[3] = uint8_t(crc);
}

void launch_png_hist(hipStream_t st, const DeflateCtx &c) { if (c.total_groups) CSH_LAUNCH(k_png_hist, dim3(c.total_groups, c.plan.ntrials), dim3(CSP_WAVE_THREADS), st, c); }
void launch_png_codes(hipStream_t st, const DeflateCtx &c, int only_deep) { if (c.total_chunks) CSH_LAUNCH(k_png_codes, dim3((c.total_chunks + 63) / 64, c.plan.ntrials), dim3(64), st, c, only_deep); }
void launch_png_choose(hipStream_t st, const DeflateCtx &c) { if (c.nimg) CSH_LAUNCH(k_png_choose, dim3((c.nimg + 63) / 64), dim3(64), st, c); }
void launch_png_emit(hipStream_t st, const DeflateCtx &c) { if (c.total_groups) { CSH_LAUNCH(k_png_emit, dim3(c.total_groups), dim3(CSP_WAVE_THREADS), st, c); launch_png_deep_emit(st, c); } }
void launch_png_finish(hipStream_t st, const DeflateCtx &c, uint32_t max_pieces) {
    if (!c.nimg) return;
    CSH_LAUNCH_PHASED(k_png_adler, 3, dim3(c.total_chunks), dim3(256), st, c);
    CSH_LAUNCH(k_png_frame, dim3(c.nimg), dim3(64), st, c);
    CSH_LAUNCH_PHASED(k_png_crc_pieces, 2, dim3((max_pieces + 255) / 256, c.nimg), dim3(256), st, c, max_pieces);
    CSH_LAUNCH(k_png_crc_fold1, dim3((max_pieces / 64 + 64) / 64, c.nimg), dim3(64), st, c, max_pieces);
    CSH_LAUNCH(k_png_crc_fold2, dim3((c.nimg + 63) / 64), dim3(64), st, c, max_pieces);
}

}  // namespace csp
