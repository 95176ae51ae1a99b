// k_png_inflate.hip -- row P1 of SURVEY.md 8a on the device: the IDAT zlib stream back to filtered rows (RFC 1951), then
// the PNG reconstruction filters back to pixels.  Statement: oracle/png_oracle.c inflate_raw() / cso_png_decode().
//
// A deflate stream is one serial chain of prefix codes, and the files of a batch are the obvious parallel axis (one workgroup per
// stream).  Two things make a stream slow when it is walked token by token: waiting for every match copy before the next symbol, and
// the walk itself being one dependent chain.  Both are taken apart:
//   k_png_huff   walks the codes and nothing else, speculatively: where a prefix-coded stream is entered matters only for a few tokens,
//                so every lane of the workgroup's four waves walks its own 352-bit stretch of the block from a guessed entry, then from
//                where its left neighbour's walk really left off, until no entry moves; prefix sums of what the walks produce place
//                every literal (stored directly: its position is known) and every match (an 8-byte record: position, length,
//                distance).  The first wave also parses the block headers and builds the tables (every lane decodes its own root
//                entries with the canonical walk: no scatter); the other waves act on its commands between two barriers.
//   k_png_lz77   resolves the matches, 16 KiB of output at a time, in LDS: the bytes of a match whose source lies in front of the piece
//                are copied from the 48 KiB of history the ring holds; the others get a pointer to their source and the pointers are
//                doubled (p <- p[p]) until every byte points at a byte that is final -- a run of 258 bytes at distance 1 takes 9 rounds,
//                not 258 steps -- then gathered.  No step waits for a single copy.
// Adam7 inputs: the stream holds seven reduced images; each is reconstructed as a job of its own, then k_png_deinterlace
// gathers the pixels into place (the output is never interlaced).
// Codes longer than the root tables (11 / 10 bits) resume the canonical walk behind the root width (rare symbols by construction).
#include "png_kernels.h"
#include "png_wave.h"

namespace csp {

#ifdef CSH_EMUL
enum { HUFF_WAVES = 1 };   // the emulation plays one wave: the same code, the hand-overs between waves degenerate
#define HUFF_BARRIER() ((void)0)
#else
#ifndef CSP_HUFF_WAVES
#define CSP_HUFF_WAVES 4
#endif
enum { HUFF_WAVES = CSP_HUFF_WAVES };
#define HUFF_BARRIER() __syncthreads()
#endif
#ifndef CSP_HUFF_SUB
#define CSP_HUFF_SUB 352   // bits of the block a lane walks per round.  Swept on the MI355X in round 6 (64 4K files, k_png_huff + k_png_lz77): 224: 458 ms, 288: 416, 320: 401, 352: 361, 384: ~360, 416: 377, 480: 409, 544: 441
#endif
#ifndef CSP_HUFF_PRE
#define CSP_HUFF_PRE 192
#endif
#ifndef CSP_LROOT
#define CSP_LROOT 11      // root table of the literal / length code: 13 bits 361 ms, 12: 330, 11: 315, 10: 312, 9: 318 (at 352 bits a lane) -- a smaller table is built faster for every block, longer codes resume the canonical walk
#endif
#ifndef CSP_DROOT
#define CSP_DROOT 10
#endif
enum { LROOT = CSP_LROOT, DROOT = CSP_DROOT, LZ_RING = 65536, LZ_PIECE = 16384, HUFF_SUB = CSP_HUFF_SUB, HUFF_PRE = CSP_HUFF_PRE, HUFF_LANES = 64 * HUFF_WAVES, HUFF_STAGE_WORDS = HUFF_SUB * HUFF_LANES / 32 + 8 };   // k_png_lz77: 64 KiB ring = the piece being resolved + 48 KiB behind it (a match reaches back 32 KiB)

struct InflateLds {
    uint32_t lcount[16], dcount[16], ccount[16], offs[16];
    uint16_t lsorted[288], dsorted[32], csorted[20];
    uint16_t lroot[1 << LROOT], droot[1 << DROOT];
    uint8_t lens[320];
    uint32_t stage[HUFF_STAGE_WORDS];   // k_png_huff: the round's stretch of the stream
    uint32_t lresume[2], dresume[2];    // canon_resume's starting point for codes longer than the root tables
    // k_png_huff, between its waves (the first wave decides, all waves act: see the kernel)
    uint32_t cmd, any, nlanes, bad;
    uint32_t rbase;                      // the round's first bit, relative to the staged copy
    uint64_t stage_bit0, pos, limit;     // the staged copy's first bit in the stream; bytes produced in front of the round; the stream's bits
    uint32_t mtotal;
    uint32_t leave[HUFF_LANES], stopk[HUFF_LANES];
    uint32_t wfirst[HUFF_WAVES], wout[HUFF_WAVES], wmat[HUFF_WAVES], woff[HUFF_WAVES], wmoff[HUFF_WAVES], wwrote[HUFF_WAVES], wlast[HUFF_WAVES], wpfin[HUFF_WAVES];
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
// the same walk for a code that is known to be longer than `root` bits: it starts where the walk stands behind length `root` (first /
// index there depend on the counts alone: `resume`, made by build_code), with the first `root` bits as the code so far
__device__ __forceinline__ static uint32_t canon_resume(uint32_t bits, int root, const uint32_t *count, const uint16_t *sorted, const uint32_t *resume) {
#ifdef CSH_EMUL
    uint32_t rev = 0;
    for (int k = 0; k < root; k++) rev |= ((bits >> k) & 1u) << (root - 1 - k);
#else
    const uint32_t rev = __brev(bits & ((1u << root) - 1u)) >> (32 - root);
#endif
    int code = int(rev << 1), first = int(resume[0]), index = int(resume[1]);
    for (int l = root + 1; l <= 15; l++) {
        code |= int((bits >> (l - 1)) & 1u);
        const int cnt = int(count[l]);
        if (code - cnt < first) return (uint32_t(sorted[index + (code - first)]) << 4) | uint32_t(l);
        index += cnt; first += cnt; first <<= 1; code <<= 1;
    }
    return 0;
}
// counts, canonical symbol order and the root table of one code; returns "left" (0 complete, >0 incomplete, <0 over-subscribed)
__device__ static int build_code(const uint8_t *lens, int n, uint32_t *count, uint32_t *offs, uint16_t *sorted, uint16_t *root, int rootbits, uint32_t *resume = nullptr) {
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
        if (resume) {   // where canon_walk stands behind length `rootbits`
            int first = 0, index = 0;
            for (int k = 1; k <= rootbits; k++) { const int cnt = int(count[k]); index += cnt; first += cnt; first <<= 1; }
            resume[0] = uint32_t(first); resume[1] = uint32_t(index);
        }
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
    uint32_t w0, w1, w2, w3, w4; // the five words from bp's word on (128+ bits in front of bp)
    uint32_t u0, u1, u2, u3;     // the 128 bits that start AT bp (the same words, shifted once, uniformly)
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
        w0 = lane_word(q); w1 = lane_word(q + 1); w2 = lane_word(q + 2); w3 = lane_word(q + 3); w4 = lane_word(q + 4);
        const uint32_t sh = uint32_t(bp & 31u);
        u0 = uint32_t((uint64_t(w0) | (uint64_t(w1) << 32)) >> sh); u1 = uint32_t((uint64_t(w1) | (uint64_t(w2) << 32)) >> sh);
        u2 = uint32_t((uint64_t(w2) | (uint64_t(w3) << 32)) >> sh); u3 = uint32_t((uint64_t(w3) | (uint64_t(w4) << 32)) >> sh);
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
    // 64 bits starting `lane` bits after bp (lane <= 63): what a whole token needs (15 + 5 + 15 + 13 bits at most).  Which words a lane
    // takes depends on the lane alone (u0.. start at bp), so this is two-way selects on a lane constant and two shifts
    __device__ __forceinline__ uint64_t at64(uint32_t lane) const {
        const bool hi = lane >= 32u;
        const uint32_t r = lane & 31u;
        const uint32_t a = hi ? u1 : u0, b = hi ? u2 : u1, c = hi ? u3 : u2;
        const uint64_t lo = (uint64_t(a) | (uint64_t(b) << 32)) >> r;
        return r ? lo | (uint64_t(c) << (64u - r)) : lo;
    }
    __device__ __forceinline__ uint32_t peek(uint32_t off, int n) const { return n ? at(off) & (0xFFFFFFFFu >> (32 - n)) : 0u; }   // n <= 32
    __device__ __forceinline__ uint32_t get(int n) { const uint32_t v = peek(0, n); bp += uint32_t(n); refresh(); return v; }
    __device__ __forceinline__ bool overrun() const { return bp > uint64_t(len) * 8u; }
};

__global__ void __launch_bounds__(CSP_WAVE_THREADS * HUFF_WAVES) k_png_huff(const PngImg *imgs, int nimg, const uint8_t *idat, uint8_t *raw, uint64_t *matches, uint32_t *nmatch, uint32_t *status) {
    CSH_SHARED InflateLds S;
    const int image = blockIdx.x;
    if (image >= nimg) return;
#ifdef CSH_EMUL
    const uint32_t wv = 0;
#else
    const uint32_t wv = threadIdx.x >> 6;
#endif
    if (wv == 0) LFOR(l) if (l == 0) nmatch[image] = 0;
    if (status[image]) return;
    const PngImg im = imgs[image];
    uint8_t *out = raw + im.inflate_off;
    uint64_t *mlist = matches + im.match_off;
    const uint64_t cap = im.inflate_len;
    const uint8_t *sbase = idat + im.idat_off;
    const uint32_t slen = im.idat_len;
    // ---- what every wave does when told to (HUFF_LANES lanes walk HUFF_LANES stretches; lane g = 64 wv + l)
    LV<uint32_t> entry, leave, nout, nmat, stopk, redo, before, mbefore;   // a lane's state over the passes of a round
    LFOR(l) { entry[l] = 0; leave[l] = 0; nout[l] = 0; nmat[l] = 0; stopk[l] = 0; redo[l] = 0; before[l] = 0; mbefore[l] = 0; }
    enum { CMD_EXIT = 0, CMD_STAGE, CMD_WALK, CMD_LINK, CMD_STOP, CMD_SUM, CMD_WRITE };
    auto lane_of = [&](const LV<uint32_t> &v, uint32_t i) __attribute__((always_inline)) -> uint32_t {
#ifdef CSH_EMUL
        return v.v[i];
#else
        return uint32_t(__builtin_amdgcn_readlane(int(v.v), int(i)));
#endif
    };
    // at least 33 bits of the stream from relative bit r on
    auto bits_at = [&](uint32_t r) __attribute__((always_inline)) -> uint64_t {
        const uint32_t w = r >> 5;
        return (uint64_t(S.stage[w]) | (uint64_t(S.stage[w + 1]) << 32)) >> (r & 31u);
    };
    // one token at relative bit r: kind 0 literal, 1 match, 2 end of block, 3 not a token (an error if it is on the true walk).
    // tl: its bits; val: the literal, or the match length (VALUES: length | distance << 16)
    auto token = [&](uint32_t r, bool values, uint32_t &kind, uint32_t &tl, uint32_t &val) __attribute__((always_inline)) {
        const uint64_t b = bits_at(r);
        uint32_t e = S.lroot[uint32_t(b) & ((1u << LROOT) - 1u)];
        if (!e) e = canon_resume(uint32_t(b) & 0x7FFFu, LROOT, S.lcount, S.lsorted, S.lresume);
        kind = 3; tl = 1; val = 0;
        if (!e) return;
        const uint32_t sym = (e >> 4) & 0x1FFu;
        tl = e & 15u;
        if (sym < 256) { kind = 0; val = sym; return; }
        if (sym == 256) { kind = 2; return; }
        const uint32_t li = sym - 257;
        if (li >= 29) return;
        const uint32_t eb = (li < 8 || li == 28) ? 0u : (li >> 2) - 1u;
        const uint32_t len = (li < 8 ? 3u + li : li == 28 ? 258u : ((4u | (li & 3u)) << eb) + 3u) + (uint32_t(b >> tl) & ((1u << eb) - 1u));
        tl += eb;   // <= 20
        uint32_t d = S.droot[uint32_t(b >> tl) & ((1u << DROOT) - 1u)];
        if (!d) d = canon_resume(uint32_t(bits_at(r + tl)) & 0x7FFFu, DROOT, S.dcount, S.dsorted, S.dresume);
        const uint32_t ds = (d >> 4) & 0x7FFu;
        if (!d || ds >= 30) return;
        const uint32_t deb = ds < 4 ? 0u : (ds >> 1) - 1u;
        tl += d & 15u;
        val = len;
        if (values) val |= ((ds < 4 ? ds + 1u : ((2u | (ds & 1u)) << deb) + 1u) + (uint32_t(bits_at(r + tl)) & ((1u << deb) - 1u))) << 16;
        tl += deb;
        kind = 1;
    };
    auto act = [&](uint32_t cmd) __attribute__((always_inline)) {
        const uint32_t rbase = S.rbase;
        if (cmd == CMD_STAGE) {   // the round's stretch of the stream -> LDS (zero behind the stream's end); every lane's first guess
            const uint64_t b0 = S.stage_bit0 >> 3;
            LFOR(l) for (uint32_t i = 64u * wv + uint32_t(l); i < uint32_t(HUFF_STAGE_WORDS); i += uint32_t(HUFF_LANES)) {
                const uint64_t at = b0 + uint64_t(i) * 4u;
                uint32_t w = 0;
                if (at + 4 <= slen) w = *reinterpret_cast<const uint32_t *>(sbase + at);
                else for (int k = 0; k < 4; k++) if (at + uint32_t(k) < slen) w |= uint32_t(sbase[at + uint32_t(k)]) << (8 * k);
                S.stage[i] = w;
            }
            LFOR(l) { entry[l] = rbase + uint32_t(HUFF_SUB) * (64u * wv + uint32_t(l)); redo[l] = 2; leave[l] = 0; nout[l] = 0; nmat[l] = 0; stopk[l] = 0; }   // redo 2: first walk
        } else if (cmd == CMD_WALK) {   // the lanes whose entry moved walk their stretch; everyone publishes where its walk left off
            LFOR(l) {
                if (redo[l]) {
                    const uint32_t end = rbase + uint32_t(HUFF_SUB) * (64u * wv + uint32_t(l) + 1u);
                    if (redo[l] == 2 && (wv || l)) {
                        // a better first guess than the stretch's first bit: a walk that comes in from HUFF_PRE bits in front of it has
                        // usually fallen into step by the time it arrives, so most lanes start their first pass at a true token
                        const uint32_t s0 = entry[l];
                        uint32_t q = s0 - uint32_t(HUFF_PRE);
                        while (q < s0) {
                            uint32_t kind, tl, val;
                            token(q, false, kind, tl, val);
                            if (kind >= 2) { q = s0; break; }   // no token, or an end of block, on a walk that may be nobody's: the plain guess
                            q += tl;
                        }
                        entry[l] = q;
                    }
                    uint32_t p = entry[l], no = 0, nm = 0, sk = 0;
                    if (p != ~0u) {
                        while (p < end) {
                            uint32_t kind, tl, val;
                            token(p, false, kind, tl, val);
                            if (kind == 3) { sk = 3; break; }
                            p += tl;
                            if (kind == 2) { sk = 2; break; }
                            if (kind == 1) { no += val; nm++; } else no++;
                        }
                    } else sk = 3;   // no entry: the walk in front of it ended the block (or was no walk)
                    leave[l] = p; nout[l] = no; nmat[l] = nm; stopk[l] = sk;
                }
                S.leave[64u * wv + uint32_t(l)] = stopk[l] ? ~0u : leave[l];
            }
        } else if (cmd == CMD_LINK) {   // everyone's new entry: where the left neighbour left off (nowhere, if it stopped)
            LFOR(l) {
                const uint32_t g = 64u * wv + uint32_t(l);
                const uint32_t from = g ? S.leave[g - 1] : rbase;
                redo[l] = from != entry[l] ? 1u : 0u;
                entry[l] = from;
            }
            if (lballot([&](int l) { return redo[l] != 0u; })) LFOR(l) if (l == 0) S.any = 1;
        } else if (cmd == CMD_STOP) {   // the first lane of this wave whose walk stopped (64: none)
            const uint64_t st = lballot([&](int l) { return stopk[l] != 0u; });
            LFOR(l) { S.stopk[64u * wv + uint32_t(l)] = stopk[l]; S.leave[64u * wv + uint32_t(l)] = leave[l]; if (l == 0) S.wfirst[wv] = st ? uint32_t(__builtin_ctzll(st)) : 64u; }   // leave: as it is, the stopped walks' too
        } else if (cmd == CMD_SUM) {   // bytes and matches of the lanes on the true walk: scans inside the wave, totals for the first wave to add up
            const uint32_t nl = S.nlanes;
            LV<uint32_t> mo, mm;
            LFOR(l) { const bool on = 64u * wv + uint32_t(l) < nl; mo[l] = on ? nout[l] : 0u; mm[l] = on ? nmat[l] : 0u; }
            uint32_t to = 0, tm = 0;
            before = lscan(mo, to);
            mbefore = lscan(mm, tm);
            LFOR(l) if (l == 0) { S.wout[wv] = to; S.wmat[wv] = tm; }
        } else if (cmd == CMD_WRITE) {   // literals to their bytes, matches to their records.  Nothing is decoded past the image's last byte
            const uint32_t nl = S.nlanes, mt = S.mtotal + S.wmoff[wv];
            const uint64_t p0 = S.pos + S.woff[wv], limit = S.limit, sb0 = S.stage_bit0;
            LV<uint32_t> badl, wrote, pfin, started;
            LFOR(l) {
                badl[l] = 0; wrote[l] = 0; pfin[l] = 0; started[l] = 0;
                const uint32_t g = 64u * wv + uint32_t(l);
                if (g < nl && entry[l] != ~0u && p0 + before[l] < cap) {
                    started[l] = 1;
                    const uint32_t end = rbase + uint32_t(HUFF_SUB) * (g + 1u);
                    uint32_t p = entry[l];
                    uint64_t at = p0 + before[l];
                    uint32_t mi = mt + mbefore[l];
                    while (p < end && at < cap) {
                        uint32_t kind, tl, val;
                        token(p, true, kind, tl, val);
                        if (kind == 3) { badl[l] = 1; break; }
                        if (kind == 2) break;
                        if (kind == 1) {
                            const uint32_t len = val & 0xFFFFu, dist = val >> 16;
                            if (uint64_t(dist) > at || sb0 + p + tl > limit) { badl[l] = 1; break; }
                            mlist[mi++] = (at & 0xFFFFFFFFull) | (uint64_t(len) << 32) | (uint64_t(dist) << 48);
                            at += len;
                        } else out[at++] = uint8_t(val);
                        p += tl;
                    }
                    wrote[l] = mi - (mt + mbefore[l]); pfin[l] = p;
                }
            }
            if (lballot([&](int l) { return badl[l] != 0u; })) LFOR(l) if (l == 0) S.bad = 1;
            uint32_t nw = 0;
            (void)lscan(wrote, nw);
            const uint64_t sm = lballot([&](int l) { return started[l] != 0u; });
            const uint32_t lastl = sm ? uint32_t(63 - __builtin_clzll(sm)) : 64u;
            const uint32_t pf = sm ? lane_of(pfin, lastl) : 0u;
            LFOR(l) if (l == 0) { S.wwrote[wv] = nw; S.wlast[wv] = lastl; S.wpfin[wv] = pf; }
        }
    };
    if (wv != 0) {   // the other waves: told what to do between two barriers, until told to leave
        for (;;) {
            HUFF_BARRIER();
            const uint32_t cmd = uni(S.cmd);
            if (cmd == CMD_EXIT) return;
            act(cmd);
            HUFF_BARRIER();
        }
    }
    auto run = [&](uint32_t cmd) __attribute__((always_inline)) {   // first wave: publish the command, act on it with everyone, meet again
        LFOR(l) if (l == 0) S.cmd = cmd;
        HUFF_BARRIER();
        act(cmd);
        HUFF_BARRIER();
    };
    PosReader rd;
    rd.begin(idat + im.idat_off, im.idat_len, 16);   // the host checked the two zlib header bytes
    uint64_t pos = 0;
    uint32_t err = 0;
    uint32_t mtotal = 0;   // matches so far: record k of the stream is mlist[k]
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
                LFOR(l) if (uint32_t(l) < m && pos + uint32_t(l) < cap) out[pos + uint32_t(l)] = rd.base[at + r0 + uint32_t(l)];
                pos += m;
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
            int r = build_code(S.lens, nlen, S.lcount, S.offs, S.lsorted, S.lroot, LROOT, S.lresume);
            if (type == 2 && (r < 0 || (r > 0 && nlen - int(S.lcount[0]) != 1))) { err = CSP_ERR_BAD_PNG; break; }
            r = build_code(S.lens + 288, ndist, S.dcount, S.offs, S.dsorted, S.droot, DROOT, S.dresume);
            if (type == 2 && (r < 0 || (r > 0 && ndist - int(S.dcount[0]) != 1))) { err = CSP_ERR_BAD_PNG; break; }   // the fixed distance code is incomplete by definition
        }
        // The symbols, HUFF_SUB bits per lane and round.  Where a prefix-coded stream is entered matters only for a few tokens: a walk that
        // starts at a wrong bit falls into step with the true one after a handful of codes.  So every lane of every wave walks its own
        // stretch of the block -- lane g the tokens that start in [base + g SUB, base + (g + 1) SUB) -- first from the stretch's first bit
        // (a guess; lane 0's is the truth), then from where its left neighbour's walk actually left off, again and again until no lane's
        // entry moves (a handful of passes; each only for the lanes whose entry moved).  Then the walks are the block's token sequence cut
        // in HUFF_LANES: prefix sums of what each produces give every lane its place in the output and in the match list, and a last
        // pass writes.  This wave decides and does the serial parts (headers, tables, sums over the waves); act() is what all waves do.
        bool block_done = false;
        while (!block_done && !err && pos < cap) {
            if (rd.overrun()) { err = CSP_ERR_BAD_PNG; break; }
            const uint64_t base = rd.bp, stage_bit0 = (base >> 5) << 5;
            LFOR(l) if (l == 0) { S.rbase = uint32_t(base - stage_bit0); S.stage_bit0 = stage_bit0; S.pos = pos; S.mtotal = mtotal; S.limit = uint64_t(rd.len) * 8u; S.bad = 0; }
            run(CMD_STAGE);
            for (int pass = 0; pass < HUFF_LANES + 2; pass++) {
                run(CMD_WALK);
                LFOR(l) if (l == 0) S.any = 0;
                run(CMD_LINK);
                if (!uni(S.any)) break;
            }
            run(CMD_STOP);
            // the lanes of the true walk: up to and including the first that stopped
            uint32_t nlanes = uint32_t(HUFF_LANES), how = 0;
            for (uint32_t w = 0; w < uint32_t(HUFF_WAVES); w++) { const uint32_t f = uni(S.wfirst[w]); if (f < 64u) { nlanes = 64u * w + f + 1u; how = uni(S.stopk[64u * w + f]); break; } }
            LFOR(l) if (l == 0) S.nlanes = nlanes;
            run(CMD_SUM);
            uint32_t tot_out = 0, tot_mat = 0;
            for (uint32_t w = 0; w < uint32_t(HUFF_WAVES); w++) { LFOR(l) if (l == 0) { S.woff[w] = tot_out; S.wmoff[w] = tot_mat; } tot_out += uni(S.wout[w]); tot_mat += uni(S.wmat[w]); }
            run(CMD_WRITE);
            if (uni(S.bad)) { err = CSP_ERR_BAD_PNG; break; }
            uint32_t nw = 0, lastw = ~0u;
            for (uint32_t w = 0; w < uint32_t(HUFF_WAVES); w++) { nw += uni(S.wwrote[w]); if (uni(S.wlast[w]) < 64u) lastw = w; }
            mtotal += nw;   // == tot_mat unless the image's last byte came first
            (void)tot_mat;
            pos += tot_out;
            if (pos >= cap) {   // complete: the stream position is where the last walk that wrote anything stopped
                if (lastw != ~0u) { rd.bp = stage_bit0 + uni(S.wpfin[lastw]); rd.refresh(); }
                break;
            }
            if (how == 3) { err = CSP_ERR_BAD_PNG; break; }   // the true walk met something that is no token
            rd.bp = stage_bit0 + uni(S.leave[nlanes - 1]);
            rd.refresh();
            if (how == 2) block_done = true;
        }
    }
    LFOR(l) if (l == 0) S.cmd = CMD_EXIT;
    HUFF_BARRIER();   // the other waves leave
    if (!err && rd.overrun()) err = CSP_ERR_BAD_PNG;
    if (!err && pos < cap) err = CSP_ERR_BAD_PNG;   // libpng: "not enough image data"
    if (err) { LFOR(l) if (l == 0) status[image] = err; return; }
    LFOR(l) if (l == 0) nmatch[image] = mtotal;
}

// ---- the matches of one stream, resolved piece by piece (header of this file).  One workgroup of HUFF_WAVES waves per stream: every
// step of a piece is a loop over its bytes or its matches shared out over all lanes, with a barrier behind it; what decides the control
// flow (are there more matches for this piece? did a pointer move?) is read by all waves from LDS behind a barrier, so they agree.
struct Lz77Lds {
    alignas(16) uint8_t ring[LZ_RING];      // byte at absolute position p: ring[p & 65535]
    alignas(16) uint16_t ptr[LZ_PIECE];     // for the bytes of the piece: ring index of a byte this one equals (itself: final)
    uint32_t wfin[HUFF_WAVES];              // matches a wave finished in this step (64: all of its 64)
    uint32_t moved[3];                      // "a pointer moved" of the doubling rounds, three in rotation
};
__global__ void __launch_bounds__(CSP_WAVE_THREADS * HUFF_WAVES) k_png_lz77(const PngImg *imgs, int nimg, uint8_t *raw, const uint64_t *matches, const uint32_t *nmatch, const uint32_t *status) {
    CSH_SHARED Lz77Lds S;
    const int image = blockIdx.x;
    if (image >= nimg || status[image]) return;
#ifdef CSH_EMUL
    const uint32_t wv = 0;
#else
    const uint32_t wv = threadIdx.x >> 6;
#endif
    const uint32_t NL = uint32_t(HUFF_LANES);
    const PngImg im = imgs[image];
    uint8_t *out = raw + im.inflate_off;
    const uint64_t *mlist = matches + im.match_off;
    const uint64_t cap = im.inflate_len;
    const uint32_t nm = nmatch[image];
    uint32_t mi = 0;   // first match that may still have bytes at or behind the piece's start (the same in every wave)
    if (wv == 0) LFOR(l) if (l < 3) S.moved[l] = 0;
    HUFF_BARRIER();
    for (uint64_t c0 = 0; c0 < cap; c0 += LZ_PIECE) {
        const uint64_t c1 = c0 + LZ_PIECE < cap ? c0 + LZ_PIECE : cap;
        const uint32_t n = uint32_t(c1 - c0);
        // the piece as k_png_huff left it (literals in place, match bytes undefined), every byte its own source
        for (uint32_t i0 = 0; i0 < n; i0 += 16u * NL) {
            LFOR(l) {
                const uint32_t i = i0 + (64u * wv + uint32_t(l)) * 16u;
                if (i < n) {
                    uint4 v;
                    if (i + 16 <= n) v = *reinterpret_cast<const uint4 *>(out + c0 + i);
                    else { uint8_t t[16]; for (int k = 0; k < 16; k++) t[k] = i + uint32_t(k) < n ? out[c0 + i + uint32_t(k)] : uint8_t(0); v = *reinterpret_cast<const uint4 *>(t); }
                    *reinterpret_cast<uint4 *>(S.ring + ((uint32_t(c0) + i) & (LZ_RING - 1))) = v;
                }
            }
        }
        for (uint32_t i0 = 0; i0 < LZ_PIECE; i0 += 4u * NL) {
            LFOR(l) {
                const uint32_t i = i0 + (64u * wv + uint32_t(l)) * 4u, r = (uint32_t(c0) + i) & (LZ_RING - 1);
                uint2 v; v.x = r | ((r + 1) << 16); v.y = (r + 2) | ((r + 3) << 16);
                *reinterpret_cast<uint2 *>(&S.ptr[i]) = v;
            }
        }
        if (wv == 0) LFOR(l) if (l < 3) S.moved[l] = 0;
        CSP_WAVE_SYNC();
        HUFF_BARRIER();
        // the matches that reach into the piece, one per lane and step.  A byte whose source lies in front of the piece takes its value
        // now (that part of the ring is final); one whose source is in the piece points at it.
        for (bool more = mi < nm; more;) {
            LV<uint32_t> done;
            LFOR(l) {
                done[l] = 0;
                const uint32_t k = mi + 64u * wv + uint32_t(l);
                if (k < nm) {
                    const uint64_t rec = mlist[k];
                    const uint64_t mpos = rec & 0xFFFFFFFFull;   // streams are shorter than 4 GiB (plan: at most 2^28 pixels of at most 8 bytes)
                    const uint32_t len = uint32_t(rec >> 32) & 0xFFFFu, dist = uint32_t(rec >> 48);
                    if (mpos < c1) {
                        const uint64_t e = mpos + len < c1 ? mpos + len : c1;
                        for (uint64_t p = mpos > c0 ? mpos : c0; p < e; p++) {
                            const uint64_t src = p - dist;
                            if (src < c0) S.ring[uint32_t(p) & (LZ_RING - 1)] = S.ring[uint32_t(src) & (LZ_RING - 1)];
                            else S.ptr[uint32_t(p - c0)] = uint16_t(uint32_t(src) & (LZ_RING - 1));
                        }
                        done[l] = mpos + len <= c1 ? 1u : 0u;
                    }
                }
            }
            // records are in stream order: the finished ones are a prefix of the step's; stop at the first that goes on into the next piece
            const uint64_t fin = lballot([&](int l) { return done[l] != 0u; });
            LFOR(l) if (l == 0) S.wfin[wv] = fin == ~0ull ? 64u : uint32_t(__builtin_ctzll(~fin));
            CSP_WAVE_SYNC();
            HUFF_BARRIER();
            uint32_t total = 0;
            bool all = true;
            for (uint32_t w = 0; w < uint32_t(HUFF_WAVES) && all; w++) { const uint32_t f = uni(S.wfin[w]); total += f; all = f == 64u; }
            mi += total;
            more = all && mi < nm;
            HUFF_BARRIER();   // wfin is read by all before the next step writes it
        }
        // pointer doubling, four bytes per lane and step: p <- ptr[p].  A final byte points at itself, so the step needs no test; the
        // piece starts on a multiple of its size, so a ring index's low 14 bits are the byte's index in the piece.  Entries behind the
        // stream's last byte point at themselves.
        for (int round = 0; round < 15; round++) {
            bool changed = false;
            for (uint32_t i0 = 0; i0 < n; i0 += 4u * NL) {
                LV<uint32_t> moved;
                LFOR(l) {
                    moved[l] = 0;
                    const uint32_t i = i0 + (64u * wv + uint32_t(l)) * 4u;
                    if (i < n) {
                        const uint2 v = *reinterpret_cast<const uint2 *>(&S.ptr[i]);
                        uint2 w;
                        w.x = uint32_t(S.ptr[v.x & (LZ_PIECE - 1)]) | (uint32_t(S.ptr[(v.x >> 16) & (LZ_PIECE - 1)]) << 16);
                        w.y = uint32_t(S.ptr[v.y & (LZ_PIECE - 1)]) | (uint32_t(S.ptr[(v.y >> 16) & (LZ_PIECE - 1)]) << 16);
                        if (w.x != v.x || w.y != v.y) { *reinterpret_cast<uint2 *>(&S.ptr[i]) = w; moved[l] = 1; }
                    }
                }
                if (lballot([&](int l) { return moved[l] != 0u; })) changed = true;
            }
            LFOR(l) if (l == 0) { if (changed) S.moved[round % 3] = 1; if (wv == 0) S.moved[(round + 1) % 3] = 0; }
            CSP_WAVE_SYNC();
            HUFF_BARRIER();
            if (!uni(S.moved[round % 3])) break;
        }
        for (uint32_t i0 = 0; i0 < n; i0 += 4u * NL) {
            LFOR(l) {
                const uint32_t i = i0 + (64u * wv + uint32_t(l)) * 4u;
                if (i < n) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(&S.ptr[i]);
                    const uint32_t bytes = uint32_t(S.ring[v.x & 0xFFFFu]) | (uint32_t(S.ring[v.x >> 16]) << 8) | (uint32_t(S.ring[v.y & 0xFFFFu]) << 16) | (uint32_t(S.ring[v.y >> 16]) << 24);
                    *reinterpret_cast<uint32_t *>(S.ring + ((uint32_t(c0) + i) & (LZ_RING - 1))) = bytes;
                }
            }
        }
        CSP_WAVE_SYNC();
        HUFF_BARRIER();
        for (uint32_t i0 = 0; i0 < n; i0 += 16u * NL) {
            LFOR(l) {
                const uint32_t i = i0 + (64u * wv + uint32_t(l)) * 16u;
                if (i < n) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(S.ring + ((uint32_t(c0) + i) & (LZ_RING - 1)));
                    if (i + 16 <= n) *reinterpret_cast<uint4 *>(out + c0 + i) = v;
                    else { const uint8_t *t = reinterpret_cast<const uint8_t *>(&v); for (uint32_t k = 0; i + k < n; k++) out[c0 + i + k] = t[k]; }
                }
            }
        }
        HUFF_BARRIER();   // the next piece rewrites ptr and a quarter of the ring
    }
}

// ---- reconstruction filters: pixel (i, y) needs (i-1, y), (i, y-1), (i-1, y-1) -> an anti-diagonal front.  One wave per
// image; its lanes are 64 consecutive rows, lane l one pixel behind lane l-1, so the pixel above arrives by a lane
// shift from the row's upper neighbour (only lane 0 reads the band above from memory).
__device__ __forceinline__ static uint64_t load64u_(const uint8_t *p) {
#ifdef CSH_EMUL
    uint64_t v; memcpy(&v, p, 8); return v;
#else
    return *reinterpret_cast<const uint64_t *>(p);   // unaligned global loads are legal on gfx9
#endif
}
__device__ __forceinline__ static int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
// A row filtered with None or Sub does not look at the row above it, so the rows of an image fall into independent runs that start at
// such rows (and at row 0).  Wave k of a job takes the runs that START in rows [64 k, 64 k + 64): from the first such row there up to the
// first one at or behind row 64 (k + 1).  An image written with adaptive filters has hundreds of runs; one filtered with Paeth throughout
// has one, and wave 0 walks it alone as before.
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_png_unfilter(const PngPass *jobs, int njobs, uint8_t *work, uint32_t *status) {
    if (int(blockIdx.y) >= njobs) return;
    const PngPass job = jobs[blockIdx.y];
    const int image = int(job.image);
    if (status[image]) return;
    const uint8_t *src = work + job.src_off;
    uint8_t *dst = work + job.dst_off;
    const uint32_t W = job.rowbytes, bpp = job.bpp, npx = W / bpp, H = job.height;
    const uint32_t band0 = blockIdx.x * 64u;
    if (band0 >= H) return;
    // first row of a run in this wave's band; first row of a run at or behind the band's end
    auto run_start_in = [&](uint32_t r0) __attribute__((always_inline)) -> uint32_t {   // first y in [r0, r0 + 64) that starts a run, or ~0
        const uint64_t m = lballot([&](int l) { const uint32_t y = r0 + uint32_t(l); return y < H && (y == 0 || src[uint64_t(y) * (W + 1)] <= 1u); });
        return m ? r0 + uint32_t(__builtin_ctzll(m)) : ~0u;
    };
    const uint32_t ys = run_start_in(band0);
    if (ys == ~0u) return;
    uint32_t ye = H;
    for (uint32_t r0 = band0 + 64u; r0 < H; r0 += 64) { const uint32_t f = run_start_in(r0); if (f != ~0u) { ye = f; break; } }
    bool bad = false;
    for (uint32_t y0 = ys; y0 < ye; y0 += 64) {
        LV<uint32_t> ft;
        LV<uint64_t> a, c, mine, ahead;   // left, upper-left, this row's latest pixel (bytes packed little-endian); the next pixel's filtered bytes
        LFOR(l) { const uint32_t y = y0 + uint32_t(l); ft[l] = y < ye ? src[uint64_t(y) * (W + 1)] : 0u; a[l] = 0; c[l] = 0; mine[l] = 0; ahead[l] = 0; }
        if (lballot([&](int l) { return ft[l] > 4u; })) { bad = true; break; }
        if (y0 > ys) CSP_MEM_FENCE();   // the band above was written by this wave
        LV<uint64_t> upper;   // the last row of the band above, 64 pixels at a time (lane l: pixel t0 + l): lane 0's upper neighbour comes out of it by a lane
        LFOR(l) upper[l] = 0;   // read -- one coalesced, cache-bypassing fetch per 64 steps instead of one per step, which was what a step waited for
        for (uint32_t t = 0; t < npx + 63; t++) {
            if (y0 > ys && (t & 63u) == 0) LFOR(l) {
                const uint32_t i = t + uint32_t(l);
                uint64_t b = 0;
                if (i < npx) for (uint32_t k = 0; k < bpp; k++) b |= uint64_t(coherent_load(dst + uint64_t(y0 - 1) * W + uint64_t(i) * bpp + k)) << (8 * k);
                upper[l] = b;
            }
            uint64_t up0 = 0;
            if (y0 > ys) {
#ifdef CSH_EMUL
                up0 = upper.v[t & 63u];
#else
                up0 = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(upper.v)), int(t & 63u)))) | (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(upper.v >> 32)), int(t & 63u)))) << 32);
#endif
            }
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
                if (y < ye && t >= uint32_t(l) && i < npx) {
                    uint64_t b = up[l];
                    if (l == 0 && y0 > ys) b = up0;   // (lane 0 is at pixel t)
                    // the pixel's filtered bytes: one unaligned 8-byte load (bpp <= 8; what it takes past the row lies inside the buffer's slack), asked for a step
                    // ahead -- the lanes of a wave read 64 different rows, a line each, and a step used to wait for three such byte loads
                    const uint8_t *f = src + uint64_t(y) * (W + 1) + 1 + uint64_t(i) * bpp;
                    const uint64_t f8 = i == 0 ? load64u_(f) : ahead[l];
                    if (i + 1 < npx) ahead[l] = load64u_(f + bpp);
                    uint64_t o = 0;
                    for (uint32_t k = 0; k < bpp; k++) {
                        const int av = int((a[l] >> (8 * k)) & 255u), bv = int((b >> (8 * k)) & 255u), cv = int((c[l] >> (8 * k)) & 255u);
                        int v = int((f8 >> (8 * k)) & 255u);
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

void launch_png_inflate(hipStream_t st, const PngImg *imgs, int nimg, const uint8_t *idat, uint8_t *raw, uint64_t *matches, uint32_t *nmatch, uint32_t *status) {
    if (!nimg) return;
    CSH_LAUNCH(k_png_huff, dim3(nimg), dim3(CSP_WAVE_THREADS * HUFF_WAVES), st, imgs, nimg, idat, raw, matches, nmatch, status);
    CSH_LAUNCH(k_png_lz77, dim3(nimg), dim3(CSP_WAVE_THREADS * HUFF_WAVES), st, imgs, nimg, raw, matches, nmatch, status);
}
void launch_png_unfilter(hipStream_t st, const PngPass *jobs, int njobs, uint32_t max_height, uint8_t *work, uint32_t *status) {
    if (njobs && max_height) CSH_LAUNCH(k_png_unfilter, dim3((max_height + 63) / 64, unsigned(njobs)), dim3(CSP_WAVE_THREADS), st, jobs, njobs, work, status);
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
