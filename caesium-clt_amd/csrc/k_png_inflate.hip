// k_png_inflate.hip -- row P1 of SURVEY.md 8a on the device: the IDAT zlib stream back to filtered rows (RFC 1951), then
// the PNG reconstruction filters back to pixels.  Statement: oracle/png_oracle.c inflate_raw() / cso_png_decode().
//
// A deflate stream is one serial chain (every symbol's position depends on all symbols before it), so the parallelism is
// across the files of the batch: ONE WAVE PER STREAM.  Control flow is wave-uniform; what the 64 lanes add is
//   * the stream window: 64 words per load, handed to the uniform bit reader by v_readlane (png_wave.h LeReader);
//   * table construction: every lane decodes its own root-table indices with the canonical (count/first) walk, so the
//     1024-entry table is filled without a scatter;
//   * match copies: up to 258 bytes move 64 per step, out of a 32 KiB ring in LDS that holds the most recent output (only a
//     match further back than the ring -- the last 258 bytes of the window -- reads the flushed copy in HBM);
//   * the flush itself: whole KiB leave the ring as 64 x 16-byte stores.
// Adam7 inputs: the stream holds seven reduced images; each is reconstructed as a job of its own, then k_png_deinterlace
// gathers the pixels into place (the output is never interlaced).
// Codes longer than the root width take the bit-serial canonical walk (rare symbols by construction).
#include "png_kernels.h"
#include "png_wave.h"

namespace csp {

enum { LROOT = 10, DROOT = 8, RING = 32768, RING_NEAR = RING - 258 };   // the ring covers (almost) the whole deflate window: 40 KB of LDS per stream, four streams per CU

struct InflateLds {
    uint32_t lcount[16], dcount[16], ccount[16], offs[16];
    uint16_t lsorted[288], dsorted[32], csorted[20];
    uint16_t lroot[1 << LROOT], droot[1 << DROOT];
    uint8_t lens[320];
    alignas(16) uint8_t ring[RING];
};

// canonical walk over the low bits of `bits` (LSB first), at most maxlen of them: (sym << 4) | len, or 0
__device__ __forceinline__ static uint32_t canon_walk(uint32_t bits, int maxlen, const uint32_t *count, const uint16_t *sorted) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= maxlen; l++) {
        code |= int((bits >> (l - 1)) & 1u);
        const int cnt = int(count[l]);
        if (code - cnt < first) return (uint32_t(sorted[index + (code - first)]) << 4) | uint32_t(l);
        index += cnt; first += cnt; first <<= 1; code <<= 1;
    }
    return 0;
}
// counts, canonical symbol order and the root table of one code; returns "left" (0 complete, >0 incomplete, <0 over-subscribed)
__device__ static int build_code(const uint8_t *lens, int n, uint32_t *count, uint32_t *offs, uint16_t *sorted, uint16_t *root, int rootbits) {
    LFOR(l) if (l < 16) count[l] = 0;
    CSP_WAVE_SYNC();
    LFOR(l) for (int i = l; i < n; i += 64) atomicAdd(&count[lens[i]], 1u);
    CSP_WAVE_SYNC();
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= int(count[l]); if (left < 0) return left; }
    if (int(count[0]) == n) left = 0;
    LFOR(l) if (l == 0) {   // symbols in order of (length, symbol)
        offs[1] = 0;
        for (int k = 1; k < 15; k++) offs[k + 1] = offs[k] + count[k];
        for (int i = 0; i < n; i++) { const int k = lens[i]; if (k) sorted[offs[k]++] = uint16_t(i); }
    }
    CSP_WAVE_SYNC();
    // root entries: (symbol << 4) | length, bit 15 set for everything that is not a literal; 0 = longer than the root
    if (root) LFOR(l) for (int e = l; e < (1 << rootbits); e += 64) { const uint32_t v = canon_walk(uint32_t(e), rootbits, count, sorted); root[e] = uint16_t(v | ((v >> 4) >= 256u ? 0x8000u : 0u)); }
    CSP_WAVE_SYNC();
    return left;
}

// ---- the stream as the wave sees it: 64 consecutive words, one per lane, sliding by 32; the uniform side reads any word
// of it with v_readlane, the vector side takes "the 32 bits that start at bit bp + lane" for every lane at once
struct PosReader {
    const uint8_t *base;
    uint32_t len, wbase;     // wbase: multiple of 32 words; the window is words [wbase, wbase + 64)
    LV<uint32_t> win, nxt;   // nxt: lane l holds word wbase + 64 + (l & 31)
    uint64_t bp;             // current bit
    uint32_t w0, w1, w2, w3; // the four words from bp's word on (96+ bits in front of bp)
    __device__ __forceinline__ uint32_t loadw(uint32_t w) const {
        const uint64_t b = uint64_t(w) * 4u;
        if (b + 4 <= len) return *reinterpret_cast<const uint32_t *>(base + b);
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) if (b + i < len) v |= uint32_t(base[b + i]) << (8 * i);
        return v;
    }
    __device__ __forceinline__ uint32_t lane_word(uint32_t i) const {
#ifdef CSH_EMUL
        return win.v[i - wbase];
#else
        return uint32_t(__builtin_amdgcn_readlane(int(win.v), int(i - wbase)));
#endif
    }
    __device__ __forceinline__ void refresh() {   // after bp moved: slide the window, pick up the four words
        while ((bp >> 5) >= uint64_t(wbase) + 32u) {
            LV<uint32_t> up;
#ifdef CSH_EMUL
            for (int l = 0; l < 64; l++) up.v[l] = win.v[(l + 32) & 63];
#else
            up.v = uint32_t(__shfl(int(win.v), int((threadIdx.x + 32u) & 63u), 64));
#endif
            wbase += 32;
            LFOR(l) { win[l] = l < 32 ? up[l] : nxt[l]; nxt[l] = loadw(wbase + 64 + (uint32_t(l) & 31u)); }
        }
        const uint32_t q = uint32_t(bp >> 5);
        w0 = lane_word(q); w1 = lane_word(q + 1); w2 = lane_word(q + 2); w3 = lane_word(q + 3);
    }
    __device__ __forceinline__ void begin(const uint8_t *p, uint32_t n, uint64_t bit) {
        base = p; len = n; bp = bit; wbase = uint32_t(bit >> 5) & ~31u;
        LFOR(l) { win[l] = loadw(wbase + uint32_t(l)); nxt[l] = loadw(wbase + 64 + (uint32_t(l) & 31u)); }
        refresh();
    }
    // 32 bits starting `off` bits after bp (off + (bp & 31) <= 96)
    __device__ __forceinline__ uint32_t at(uint32_t off) const {
        const uint32_t s = uint32_t(bp & 31u) + off;
        const uint64_t lo = uint64_t(w0) | (uint64_t(w1) << 32), mid = uint64_t(w1) | (uint64_t(w2) << 32), hi = uint64_t(w2) | (uint64_t(w3) << 32);
        return s < 32 ? uint32_t(lo >> s) : s < 64 ? uint32_t(mid >> (s - 32)) : uint32_t(hi >> (s - 64));
    }
    __device__ __forceinline__ uint32_t peek(uint32_t off, int n) const { return n ? at(off) & (0xFFFFFFFFu >> (32 - n)) : 0u; }   // n <= 32
    __device__ __forceinline__ uint32_t get(int n) { const uint32_t v = peek(0, n); bp += uint32_t(n); refresh(); return v; }
    __device__ __forceinline__ bool overrun() const { return bp > uint64_t(len) * 8u; }
};

__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_png_inflate(const PngImg *imgs, int nimg, const uint8_t *idat, uint8_t *raw, uint32_t *status) {
    CSH_SHARED InflateLds S;
    const int image = blockIdx.x;
    if (image >= nimg || status[image]) return;
    const PngImg im = imgs[image];
    uint8_t *out = raw + im.inflate_off;
    const uint64_t cap = im.inflate_len;
    PosReader rd;
    rd.begin(idat + im.idat_off, im.idat_len, 16);   // the host checked the two zlib header bytes
    uint64_t pos = 0, flushed = 0;
    uint32_t err = 0;
    auto flush = [&]() {   // whole KiB that are complete leave the ring
        CSP_WAVE_SYNC();
        while (flushed + 1024 <= pos) {
            LFOR(l) *reinterpret_cast<uint4 *>(out + flushed + uint32_t(l) * 16u) = *reinterpret_cast<const uint4 *>(S.ring + ((uint32_t(flushed) + uint32_t(l) * 16u) & (RING - 1)));
            flushed += 1024;
        }
        CSP_MEM_FENCE();
    };
    bool last = false;
    while (!last && !err && pos < cap) {
        last = rd.get(1) != 0;
        const uint32_t type = rd.get(2);
        if (type == 0) {
            rd.get(int((8u - uint32_t(rd.bp & 7u)) & 7u));
            const uint32_t len = rd.get(16), nlen = rd.get(16);
            if ((len ^ 0xFFFFu) != nlen) { err = CSP_ERR_BAD_PNG; break; }
            const uint32_t at = uint32_t(rd.bp >> 3);
            if (uint64_t(at) + len > rd.len) { err = CSP_ERR_BAD_PNG; break; }
            for (uint32_t r0 = 0; r0 < len && pos < cap; r0 += 64) {
                const uint32_t m = len - r0 < 64u ? len - r0 : 64u;
                LFOR(l) if (uint32_t(l) < m) S.ring[(uint32_t(pos) + uint32_t(l)) & (RING - 1)] = rd.base[at + r0 + uint32_t(l)];
                CSP_WAVE_SYNC();
                pos += m;
                if ((pos >> 10) != (flushed >> 10)) flush();
            }
            rd.bp += uint64_t(len) * 8u;
            rd.refresh();
            continue;
        }
        if (type == 3) { err = CSP_ERR_BAD_PNG; break; }
        int nlen, ndist;
        if (type == 1) {
            nlen = 288; ndist = 30;
            LFOR(l) for (int i = l; i < 320; i += 64) S.lens[i] = uint8_t(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
            CSP_WAVE_SYNC();
        } else {
            nlen = int(rd.get(5)) + 257; ndist = int(rd.get(5)) + 1;
            const int ncode = int(rd.get(4)) + 4;
            if (nlen > 286 || ndist > 30) { err = CSP_ERR_BAD_PNG; break; }
            LFOR(l) if (l < 19) S.lens[l] = 0;
            CSP_WAVE_SYNC();
            for (int i = 0; i < ncode; i++) {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const uint32_t v = rd.get(3);
                LFOR(l) if (l == 0) S.lens[order[i]] = uint8_t(v);
            }
            CSP_WAVE_SYNC();
            if (build_code(S.lens, 19, S.ccount, S.offs, S.csorted, nullptr, 0) != 0) { err = CSP_ERR_BAD_PNG; break; }   // zlib: must be complete
            // the code lengths of the two alphabets, as one run-length coded sequence
            int idx = 0;
            uint32_t prev = 0;
            CSP_WAVE_SYNC();
            while (idx < nlen + ndist && !err) {
                const uint32_t e = canon_walk(rd.peek(0, 7), 7, S.ccount, S.csorted);
                if (!e) { err = CSP_ERR_BAD_PNG; break; }
                rd.get(int(e & 15u));
                const uint32_t sym = e >> 4;
                int rep = 1;
                uint32_t val = sym;
                if (sym == 16) { if (idx == 0) { err = CSP_ERR_BAD_PNG; break; } val = prev; rep = 3 + int(rd.get(2)); }
                else if (sym == 17) { val = 0; rep = 3 + int(rd.get(3)); }
                else if (sym == 18) { val = 0; rep = 11 + int(rd.get(7)); }
                if (idx + rep > nlen + ndist) { err = CSP_ERR_BAD_PNG; break; }
                // lens[] holds the litlen lengths at [0, nlen) and the distance lengths at [288, 288 + ndist)
                for (int k0 = 0; k0 < rep; k0 += 64) LFOR(l) if (k0 + l < rep) { const int k = idx + k0 + l; S.lens[k < nlen ? k : 288 + (k - nlen)] = uint8_t(val); }
                idx += rep; prev = val;
            }
            if (err || rd.overrun()) { err = CSP_ERR_BAD_PNG; break; }
            CSP_WAVE_SYNC();
            if (S.lens[256] == 0) { err = CSP_ERR_BAD_PNG; break; }
        }
        {
            int r = build_code(S.lens, nlen, S.lcount, S.offs, S.lsorted, S.lroot, LROOT);
            if (type == 2 && (r < 0 || (r > 0 && nlen - int(S.lcount[0]) != 1))) { err = CSP_ERR_BAD_PNG; break; }
            r = build_code(S.lens + 288, ndist, S.dcount, S.offs, S.dsorted, S.droot, DROOT);
            if (type == 2 && (r < 0 || (r > 0 && ndist - int(S.dcount[0]) != 1))) { err = CSP_ERR_BAD_PNG; break; }   // the fixed distance code is incomplete by definition
        }
        // The symbols.  One pass = one window: EVERY lane looks up the two root tables for "a code starting at bit bp + lane"
        // (two LDS reads for the whole wave), then the uniform side walks from code to code through those answers with
        // v_readlane -- the LDS round trip is paid once per ~44 bits instead of once per symbol.  Literals collect in a
        // vector register (one lane each) and reach the ring with one store when a match or the end of the window comes.
        bool block_done = false;
        while (!block_done && !err && pos < cap) {
            if (rd.overrun()) { err = CSP_ERR_BAD_PNG; break; }
            LV<uint32_t> E, D, pend;
            LFOR(l) { const uint32_t b = rd.at(uint32_t(l)); E[l] = S.lroot[b & ((1u << LROOT) - 1u)]; D[l] = S.droot[b & ((1u << DROOT) - 1u)]; pend[l] = 0; }
            uint32_t off = 0;
            uint64_t lits = 0;   // bit o set: a literal whose code starts at window offset o waits in lane o of `pend`
            auto lane_of = [&](const LV<uint32_t> &v, uint32_t i) -> uint32_t {
#ifdef CSH_EMUL
                return v.v[i];
#else
                return uint32_t(__builtin_amdgcn_readlane(int(v.v), int(i)));
#endif
            };
            auto spill = [&]() {   // pending literals -> ring, in offset order
                if (!lits) return;
                LFOR(l) if ((lits >> l) & 1) S.ring[(uint32_t(pos) + uint32_t(__popcll(lits & lanes_below(l)))) & (RING - 1)] = uint8_t(pend[l]);
                pos += uint32_t(__popcll(lits)); lits = 0;
                if ((pos >> 10) != (flushed >> 10)) flush();
            };
            // close to the end of the image a window is one symbol long, so that nothing is decoded past the last byte
            uint32_t wlimit = cap - pos < 512 ? 1u : 44u;
            while (off < wlimit) {
                uint32_t e = lane_of(E, off);
                // a run of literals out of the root table: the common case, a loop of its own with nothing else in it
                while ((e - 1u) < 0x7FFFu) {
#ifdef CSH_EMUL
                    pend.v[off] = e >> 4;
#else
                    pend.v = (threadIdx.x & 63u) == off ? e >> 4 : pend.v;
#endif
                    lits |= 1ull << off; off += e & 15u;
                    if (off >= wlimit) break;
                    e = lane_of(E, off);
                }
                if (off >= wlimit) break;
                if (!e) { e = uni(canon_walk(rd.peek(off, 15), 15, S.lcount, S.lsorted)); if (!e) { err = CSP_ERR_BAD_PNG; break; } }
                const uint32_t sym = (e >> 4) & 0x1FFu;
                if (sym < 256) {
#ifdef CSH_EMUL
                    pend.v[off] = sym;
#else
                    pend.v = (threadIdx.x & 63u) == off ? sym : pend.v;
#endif
                    lits |= 1ull << off; off += e & 15u;
                    continue;
                }
                off += e & 15u;
                spill();
                if (sym == 256) { block_done = true; break; }
                const uint32_t li = sym - 257;
                if (li >= 29) { err = CSP_ERR_BAD_PNG; break; }
                uint32_t len;
                if (li < 8) len = 3 + li; else if (li == 28) len = 258; else { const int eb = int(li >> 2) - 1; len = ((4u | (li & 3u)) << eb) + 3u + rd.peek(off, eb); off += uint32_t(eb); }
                uint32_t d = lane_of(D, off);   // off <= 43 + 15 + 5
                if (!d) { d = uni(canon_walk(rd.peek(off, 15), 15, S.dcount, S.dsorted)); if (!d) { err = CSP_ERR_BAD_PNG; break; } }
                off += d & 15u;
                const uint32_t ds = d >> 4;
                if (ds >= 30) { err = CSP_ERR_BAD_PNG; break; }
                uint32_t dist;
                if (ds < 4) dist = ds + 1; else { const int eb = int(ds >> 1) - 1; dist = ((2u | (ds & 1u)) << eb) + 1u + rd.peek(off, eb); off += uint32_t(eb); }
                if (uint64_t(dist) > pos || rd.bp + off > uint64_t(rd.len) * 8u) { err = CSP_ERR_BAD_PNG; break; }
                // byte pos+i is byte pos-dist+(i mod dist): every source lies in front of pos, so all lanes copy at once
                CSP_WAVE_SYNC();
                for (uint32_t r0 = 0; r0 < len; r0 += 64) {
                    LFOR(l) {
                        const uint32_t i = r0 + uint32_t(l);
                        if (i < len) {
                            const uint32_t back = dist - (dist >= len ? i : i % dist);   // source = pos - back
                            const uint8_t b = dist <= RING_NEAR ? S.ring[(uint32_t(pos) - back) & (RING - 1)] : coherent_load(out + (pos - back));
                            S.ring[(uint32_t(pos) + i) & (RING - 1)] = b;
                        }
                    }
                }
                CSP_WAVE_SYNC();
                pos += len;
                if ((pos >> 10) != (flushed >> 10)) flush();
                if (pos >= cap) break;
                if (cap - pos < 512) wlimit = 1;
            }
            if (err) break;
            spill();
            rd.bp += off;
            rd.refresh();
        }
    }
    if (!err && rd.overrun()) err = CSP_ERR_BAD_PNG;
    if (!err && pos < cap) err = CSP_ERR_BAD_PNG;   // libpng: "not enough image data"
    if (err) { LFOR(l) if (l == 0) status[image] = err; return; }
    // the tail: bytes [flushed, cap) are still only in the ring
    CSP_WAVE_SYNC();
    LFOR(l) for (uint64_t i = flushed + uint32_t(l); i < cap; i += 64) out[i] = S.ring[uint32_t(i) & (RING - 1)];
}

// ---- reconstruction filters: pixel (i, y) needs (i-1, y), (i, y-1), (i-1, y-1) -> an anti-diagonal front.  One wave per
// image; its lanes are 64 consecutive rows, lane l one pixel behind lane l-1, so the pixel above arrives by a lane
// shift from the row's upper neighbour (only lane 0 reads the band above from memory).
__device__ __forceinline__ static int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_png_unfilter(const PngPass *jobs, int njobs, uint8_t *work, uint32_t *status) {
    if (int(blockIdx.x) >= njobs) return;
    const PngPass job = jobs[blockIdx.x];
    const int image = int(job.image);
    if (status[image]) return;
    const uint8_t *src = work + job.src_off;
    uint8_t *dst = work + job.dst_off;
    const uint32_t W = job.rowbytes, bpp = job.bpp, npx = W / bpp, H = job.height;
    bool bad = false;
    for (uint32_t y0 = 0; y0 < H; y0 += 64) {
        LV<uint32_t> ft;
        LV<uint64_t> a, c, mine;   // left, upper-left, this row's latest pixel (bytes packed little-endian)
        LFOR(l) { const uint32_t y = y0 + uint32_t(l); ft[l] = y < H ? src[uint64_t(y) * (W + 1)] : 0u; a[l] = 0; c[l] = 0; mine[l] = 0; }
        if (lballot([&](int l) { return ft[l] > 4u; })) { bad = true; break; }
        if (y0) CSP_MEM_FENCE();   // the band above was written by this wave
        for (uint32_t t = 0; t < npx + 63; t++) {
            // what the row above produced one step ago is the pixel above this lane's current pixel
            LV<uint64_t> up;
#ifdef CSH_EMUL
            for (int l = 63; l >= 1; l--) up.v[l] = mine.v[l - 1];
            up.v[0] = 0;
#else
            {
                const uint32_t lo = uint32_t(__shfl_up(int(uint32_t(mine.v)), 1, 64)), hi = uint32_t(__shfl_up(int(uint32_t(mine.v >> 32)), 1, 64));
                up.v = (threadIdx.x & 63u) ? (uint64_t(hi) << 32) | lo : 0ull;
            }
#endif
            LFOR(l) {
                const uint32_t y = y0 + uint32_t(l), i = t - uint32_t(l);
                if (y < H && t >= uint32_t(l) && i < npx) {
                    uint64_t b = up[l];
                    if (l == 0 && y0) { b = 0; for (uint32_t k = 0; k < bpp; k++) b |= uint64_t(coherent_load(dst + uint64_t(y - 1) * W + uint64_t(i) * bpp + k)) << (8 * k); }
                    const uint8_t *f = src + uint64_t(y) * (W + 1) + 1 + uint64_t(i) * bpp;
                    uint64_t o = 0;
                    for (uint32_t k = 0; k < bpp; k++) {
                        const int av = int((a[l] >> (8 * k)) & 255u), bv = int((b >> (8 * k)) & 255u), cv = int((c[l] >> (8 * k)) & 255u);
                        int v = f[k];
                        switch (ft[l]) {
                        case 1: v += av; break;
                        case 2: v += bv; break;
                        case 3: v += (av + bv) >> 1; break;
                        case 4: v += paeth(av, bv, cv); break;
                        }
                        o |= uint64_t(v & 255) << (8 * k);
                        dst[uint64_t(y) * W + uint64_t(i) * bpp + k] = uint8_t(v);
                    }
                    a[l] = o; c[l] = b; mine[l] = o;
                }
            }
        }
    }
    if (bad) LFOR(l) if (l == 0) status[image] = CSP_ERR_BAD_PNG;
}

void launch_png_inflate(hipStream_t st, const PngImg *imgs, int nimg, const uint8_t *idat, uint8_t *raw, uint32_t *status) {
    if (nimg) CSH_LAUNCH(k_png_inflate, dim3(nimg), dim3(CSP_WAVE_THREADS), st, imgs, nimg, idat, raw, status);
}
void launch_png_unfilter(hipStream_t st, const PngPass *jobs, int njobs, uint8_t *work, uint32_t *status) {
    if (njobs) CSH_LAUNCH(k_png_unfilter, dim3(njobs), dim3(CSP_WAVE_THREADS), st, jobs, njobs, work, status);
}

// ---- Adam7: every pixel of the image gathers itself out of the pass it belongs to (no scatter, so sub-byte samples need no
// atomics): one lane per pixel for whole-byte pixels, one lane per byte of the row otherwise
__device__ __forceinline__ static int adam7_pass(uint32_t ox, uint32_t oy) {
    return (oy & 1u) ? 6 : (ox & 1u) ? 5 : (oy & 2u) ? 4 : (ox & 2u) ? 3 : (oy & 4u) ? 2 : (ox & 4u) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_png_deinterlace(const PngImg *imgs, const PngAdam7 *jobs, uint8_t *work, const uint32_t *status) {
    const PngAdam7 &a = jobs[blockIdx.y];
    if (status[a.image]) return;
    const PngImg &im = imgs[a.image];
    const uint32_t XS[7] = {0, 4, 0, 2, 0, 1, 0}, YS[7] = {0, 0, 4, 0, 2, 0, 1}, DXs[7] = {3, 3, 2, 2, 1, 1, 0}, DYs[7] = {3, 3, 3, 2, 2, 1, 1};   // steps as shifts
    const uint32_t bits = a.bits;
    uint8_t *dst = work + im.pix_off;
    if (bits >= 8) {
        const uint32_t bytes = bits >> 3;
        const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
        if (i >= uint64_t(im.width) * im.height) return;
        const uint32_t oy = uint32_t(i / im.width), ox = uint32_t(i - uint64_t(oy) * im.width);
        const int p = adam7_pass(ox, oy);
        const uint32_t x = (ox - XS[p]) >> DXs[p], y = (oy - YS[p]) >> DYs[p];
        const uint8_t *s = work + a.base[p] + uint64_t(y) * a.prb[p] + uint64_t(x) * bytes;
        uint8_t *d = dst + uint64_t(oy) * im.rowbytes + uint64_t(ox) * bytes;
        for (uint32_t k = 0; k < bytes; k++) d[k] = s[k];
    } else {
        const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
        if (i >= uint64_t(im.rowbytes) * im.height) return;
        const uint32_t oy = uint32_t(i / im.rowbytes), bx = uint32_t(i - uint64_t(oy) * im.rowbytes), per = 8u / bits;
        uint32_t v = 0;
        for (uint32_t k = 0; k < per; k++) {
            const uint32_t ox = bx * per + k;
            if (ox >= im.width) break;
            const int p = adam7_pass(ox, oy);
            const uint32_t x = (ox - XS[p]) >> DXs[p], y = (oy - YS[p]) >> DYs[p];
            const uint64_t sb = uint64_t(x) * bits;
            const uint32_t sample = (uint32_t(work[a.base[p] + uint64_t(y) * a.prb[p] + (sb >> 3)]) >> (8u - bits - uint32_t(sb & 7u))) & ((1u << bits) - 1u);
            v |= sample << (8u - bits - k * bits);
        }
        dst[i] = uint8_t(v);
    }
}
void launch_png_deinterlace(hipStream_t st, const PngImg *imgs, const PngAdam7 *jobs, int njobs, uint64_t max_items, uint8_t *work, const uint32_t *status) {
    if (njobs && max_items) CSH_LAUNCH(k_png_deinterlace, dim3(unsigned((max_items + 255) / 256), njobs), dim3(256), st, imgs, jobs, work, status);
}

}  // namespace csp
