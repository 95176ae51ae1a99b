// vp8l_dec.h -- WebP lossless (VP8L) decoder, the lossless WebP inputs of libcaesium's webp::compress / convert paths
// (/root/reference/src/compressor.rs:289-305, 589-598 name WebP among the inputs; libcaesium decodes them with libwebp).
// Statement followed: the WebP Lossless Bitstream Specification (prefix codes over an LSB-first bit reader, the colour cache, LZ77
// with the 120-entry neighbourhood distance map, meta prefix image, and the four transforms: predictor (14 modes), cross-colour,
// subtract-green, colour indexing), with libwebp's behaviour where the text leaves room (a code with a single used symbol costs no
// bits; every pixel enters the colour cache, copied and looked-up ones included; incomplete codes are an error).
// One image is decoded by ONE lane, start to end: the entropy layer is a serial chain and the pictures of a batch are the parallel
// axis.  Pixel-exact against libwebp (through Pillow) in tests/test_webp_decode*.py.  Host + device code: the emulation build compiles
// it as plain C++.
#pragma once
#include <cstdint>

// everything of the decoders is inlined into its kernel: only then does the compiler see which pointers are LDS (ds_read instead of flat_load)
#ifdef CSH_EMUL
#define CSW_INLINE
#define CSW_NOINLINE
#ifndef CSW_NOUNROLL
#define CSW_NOUNROLL
#endif
#else
#define CSW_INLINE __attribute__((always_inline))
#ifndef CSW_NOUNROLL
#define CSW_NOUNROLL _Pragma("clang loop unroll(disable)")
#endif
#define CSW_NOINLINE __attribute__((noinline))   // set-up paths: a copy per call site would push the hot loop out of the instruction cache
#endif
namespace csw {

// work area of one image (bytes): two ARGB frames (the decoded -- possibly pixel-packed -- picture and the one colour indexing expands
// into), three sub-images (predictor modes, cross-colour elements, meta prefix groups: at most a quarter of the picture's side each,
// as their block side is >= 4), the palette, the colour cache, and the prefix-code arena
__host__ __device__ CSW_INLINE static inline uint64_t vp8l_arena_bytes(uint64_t file_bytes) { return 4ull * 1024 * 1024 + 16ull * file_bytes; }
__host__ __device__ CSW_INLINE static inline uint64_t vp8l_sub_pixels(uint32_t w, uint32_t h) { return uint64_t((w + 3) / 4) * ((h + 3) / 4) + 16; }
__host__ __device__ CSW_INLINE static inline uint64_t vp8l_work_bytes(uint32_t w, uint32_t h, uint64_t file_bytes) {
    return 2ull * 4 * w * h + 3ull * 4 * vp8l_sub_pixels(w, h) + 4ull * 256 + 4ull * 2048 + vp8l_arena_bytes(file_bytes) + 256;
}

// LSB-first bit reader.  The stream is read through a window of L_WIN bytes kept close (LDS in the kernel): one lane walks the stream, and fetching
// it from the memory system eight bytes at a time was a round trip every other pixel; the window is refilled 256 words at a stretch.
// fills a bit reader's window (L_WIN + 64 bytes at `win`) from stream position `at` on; returns the stream position of win[0].  Not inlined -- a copy per
// refill site would push the pixel loop out of the instruction cache -- and called with values, so that the reader itself stays in registers.
enum { L_WIN = 1024 };
__host__ __device__ CSW_NOINLINE static uint32_t lwin_load(const uint8_t *data, uint32_t len, uint8_t *win, uint32_t at) {
    const uint32_t mis = uint32_t((reinterpret_cast<uintptr_t>(data) + at) & 15u);   // the window starts at a 16-byte boundary of memory at or in front of `at`
    const uint32_t wstart = at >= mis ? at - mis : 0u;
    const uint8_t *src = data + wstart;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && wstart + L_WIN + 64 <= len) {   // the copy runs in 64-byte steps up to L_WIN + 64: all of it must lie inside the stream;   // 64 bytes per step, the four loads in flight together (the files' streams start at
        for (uint32_t i = 0; i < L_WIN + 16; i += 64) {                                    // multiples of 16 in the pool); a load per word would be a memory round trip per word
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src + i);
            const uint4 a = s4[0], b = s4[1], c = s4[2], d = s4[3];
            uint4 *d4 = reinterpret_cast<uint4 *>(win + i);
            d4[0] = a; d4[1] = b; d4[2] = c; d4[3] = d;
        }
        return wstart;
    }
    for (uint32_t i = 0; i < L_WIN + 16; i++) win[i] = wstart + i < len ? src[i] : uint8_t(0);   // the ragged end of a stream, or its very start
    return wstart;
}
struct LBits {
    const uint8_t *data;
    uint8_t *win;          // L_WIN + 64 bytes, 16-byte aligned
    uint32_t len, pos;     // bytes of the stream; next byte to take
    uint32_t wstart;       // stream position of win[0] (a multiple of 4)
    uint64_t val;
    int nbits;
    bool eos;
    __host__ __device__ CSW_INLINE void load_window(uint32_t at) { wstart = lwin_load(data, len, win, at); }
    __host__ __device__ CSW_INLINE void init(const uint8_t *d, size_t n, uint8_t *window) { data = d; win = window; len = uint32_t(n); pos = 0; val = 0; nbits = 0; eos = false; load_window(0); }
    __host__ __device__ CSW_INLINE void fill() {
        if (pos >= len) return;
        if (pos - wstart > L_WIN) load_window(pos);
        const uint8_t *q = win + (pos - wstart);
        uint64_t w = 0;
        for (int i = 0; i < 8; i++) w |= uint64_t(q[i]) << (8 * i);
        int k = (64 - nbits) >> 3;
        if (uint32_t(k) > len - pos) k = int(len - pos);
        if (k >= 8) { val = w; nbits = 64; pos += 8; return; }   // (nbits == 0: a shift by 64 is not a shift)
        if (k <= 0) return;
        val |= (w & ((1ull << (8 * k)) - 1ull)) << nbits; nbits += 8 * k; pos += uint32_t(k);
    }
    __host__ __device__ CSW_INLINE uint32_t peek(int n) { if (nbits < n) fill(); return uint32_t(val) & ((1u << n) - 1u); }   // n <= 16
    __host__ __device__ CSW_INLINE void drop(int n) { if (nbits < n) { eos = true; val = 0; nbits = 0; return; } val >>= n; nbits -= n; }
    __host__ __device__ CSW_INLINE uint32_t read(int n) { if (!n) return 0; const uint32_t v = peek(n); drop(n); return v; }
};

// one prefix code: canonical, decoded length by length; codes of up to 8 bits also through a 256-entry table (when the arena had room)
struct LCode {
    uint16_t first_code[16], first_idx[16], count[16];
    uint32_t syms;       // sorted symbols: u16 index into the arena
    uint32_t lut;        // 256 x u16 (symbol | length << 12; 0: longer than 8 bits), or 0xFFFFFFFF
    uint16_t single;     // the only symbol, when nsym == 1 (costs no bits)
    uint16_t nsym;
};
struct LGroup { LCode c[5]; };   // green + length prefixes + cache indices, red, blue, alpha, distance

struct LArena { uint16_t *base; uint64_t cap, used; };   // in u16 units

__host__ __device__ CSW_INLINE static inline int lsym(LBits &br, const LCode &c, const uint16_t *arena) {
    if (c.nsym <= 1) return c.single;
    if (c.lut != 0xFFFFFFFFu) {
        const uint16_t e = arena[c.lut + br.peek(8)];
        if (e) { br.drop(e >> 12); return e & 0xFFF; }
    }
    uint32_t code = 0;
    for (int len = 1; len <= 15; len++) {
        code = (code << 1) | br.read(1);
        const uint32_t d = code - c.first_code[len];
        if (d < c.count[len]) return arena[c.syms + c.first_idx[len] + d];
    }
    br.eos = true;   // cannot happen with a complete code
    return 0;
}

// lengths[0..n) -> code.  false: not a complete prefix code (or no room for its symbols)
__host__ __device__ CSW_NOINLINE static bool lbuild(const uint8_t *lengths, int n, LCode &c, LArena &ar, bool want_lut) {
    for (int l = 0; l < 16; l++) c.count[l] = 0;
    int used = 0, last = 0;
    for (int s = 0; s < n; s++) if (lengths[s]) { c.count[lengths[s]]++; used++; last = s; }
    c.nsym = uint16_t(used > 0xFFFF ? 0xFFFF : used); c.single = uint16_t(last); c.lut = 0xFFFFFFFFu; c.syms = 0;
    if (used == 0) return false;
    if (used == 1) { c.nsym = 1; return true; }
    uint32_t code = 0, idx = 0, left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        if (c.count[l] > left) return false;
        left -= c.count[l];
        c.first_code[l] = uint16_t(code); c.first_idx[l] = uint16_t(idx);
        code = (code + c.count[l]) << 1; idx += c.count[l];
    }
    if (left) return false;
    if (ar.used + uint64_t(used) > ar.cap) return false;
    c.syms = uint32_t(ar.used); ar.used += uint64_t(used);
    uint16_t next[16];
    for (int l = 1; l <= 15; l++) next[l] = c.first_idx[l];
    for (int s = 0; s < n; s++) if (lengths[s]) ar.base[c.syms + next[lengths[s]]++] = uint16_t(s);
    if (want_lut && ar.used + 257 <= ar.cap) {
        ar.used += ar.used & 1;   // at an even position: the tables are copied to the hot set as words
        c.lut = uint32_t(ar.used); ar.used += 256;
        uint16_t *t = ar.base + c.lut;
        for (int i = 0; i < 256; i++) t[i] = 0;
        for (int l = 1; l <= 8; l++)
            for (uint32_t k = 0; k < c.count[l]; k++) {
                const uint32_t cd = c.first_code[l] + k;
                uint32_t rev = 0;
                for (int b = 0; b < l; b++) rev |= ((cd >> b) & 1u) << (l - 1 - b);
                const uint16_t e = uint16_t(ar.base[c.syms + c.first_idx[l] + k] | (l << 12));
                for (uint32_t i = rev; i < 256; i += 1u << l) t[i] = e;
            }
    }
    return true;
}

// reads one prefix code of `alphabet` symbols (lengths: scratch of >= alphabet bytes)
__host__ __device__ CSW_NOINLINE static bool lread_code(LBits &br, int alphabet, LCode &c, LArena &ar, uint8_t *lengths, bool want_lut) {
    for (int i = 0; i < alphabet; i++) lengths[i] = 0;
    if (br.read(1)) {   // simple code: one or two symbols
        const int nsym = int(br.read(1)) + 1;
        const int s0 = int(br.read(br.read(1) ? 8 : 1));
        if (s0 >= alphabet) return false;
        lengths[s0] = 1;
        if (nsym == 2) { const int s1 = int(br.read(8)); if (s1 >= alphabet) return false; lengths[s1] = 1; }
    } else {
        const uint8_t order[19] = {17, 18, 0, 1, 2, 3, 4, 5, 16, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
        uint8_t cl[19];
        for (int i = 0; i < 19; i++) cl[i] = 0;
        const int ncl = 4 + int(br.read(4));
        for (int i = 0; i < ncl; i++) cl[order[i]] = uint8_t(br.read(3));
        // the code of the code lengths: 19 symbols, lengths up to 7 -- decoded length by length from a small local description
        LCode cc;
        uint16_t cc_syms[19];
        LArena la; la.base = cc_syms; la.cap = 19; la.used = 0;
        if (!lbuild(cl, 19, cc, la, false)) return false;
        int max_symbol = alphabet;
        if (br.read(1)) {
            const int nb = 2 + 2 * int(br.read(3));
            max_symbol = 2 + int(br.read(nb));
            if (max_symbol > alphabet) return false;
        }
        int s = 0, prev = 8;
        while (s < alphabet) {
            if (max_symbol-- == 0) break;
            const int v = lsym(br, cc, cc_syms);
            if (v < 16) { lengths[s++] = uint8_t(v); if (v) prev = v; }
            else {
                const int extra = v == 16 ? 2 : (v == 17 ? 3 : 7), base = v == 18 ? 11 : 3;
                const int rep = int(br.read(extra)) + base;
                if (s + rep > alphabet) return false;
                const uint8_t fill = v == 16 ? uint8_t(prev) : uint8_t(0);
                for (int i = 0; i < rep; i++) lengths[s++] = fill;
            }
        }
    }
    if (br.eos) return false;
    return lbuild(lengths, alphabet, c, ar, want_lut);
}

// LZ77 prefix value (length or distance code): the prefix symbol, then its extra bits
__host__ __device__ CSW_INLINE static inline uint32_t lprefix_value(LBits &br, int sym) {
    if (sym < 4) return uint32_t(sym) + 1;
    const int extra = (sym - 2) >> 1;
    const uint32_t off = uint32_t(2 + (sym & 1)) << extra;
    return off + br.read(extra) + 1;
}
// distance code 1..120 -> pixel distance through the neighbourhood map: the 120 positions (dx, dy), dy in 0..7, dx in -7..8 (dy = 0: dx > 0),
// ordered by dx^2 + dy^2, then |dx|, then dx > 0 first (the specification's table, generated instead of spelled out); kept packed as
// (dy << 4) | (8 - dx)
__host__ __device__ CSW_INLINE static inline void lplane_table(uint8_t t[120]) {
    int n = 0;
    for (int d2 = 1; d2 <= 113 && n < 120; d2++)
        for (int ax = 0; ax <= 8; ax++)
            for (int sgn = 0; sgn < 2; sgn++) {
                const int dx = sgn ? -ax : ax;
                if (sgn && ax == 0) continue;
                const int r = d2 - ax * ax;
                if (r < 0) continue;
                int dy = 0;
                while (dy * dy < r) dy++;
                if (dy * dy != r || dy > 7) continue;
                if (dx < -7 || dx > 8) continue;
                if (dy == 0 && dx <= 0) continue;
                t[n++] = uint8_t((dy << 4) | (8 - dx));
            }
}

__host__ __device__ CSW_INLINE static inline uint32_t ladd(uint32_t a, uint32_t b) { return (((a & 0xFF00FF00u) + (b & 0xFF00FF00u)) & 0xFF00FF00u) | (((a & 0x00FF00FFu) + (b & 0x00FF00FFu)) & 0x00FF00FFu); }
__host__ __device__ CSW_INLINE static inline uint32_t lavg(uint32_t a, uint32_t b) { return (((a ^ b) & 0xFEFEFEFEu) >> 1) + (a & b); }
__host__ __device__ CSW_INLINE static inline int labs(int v) { return v < 0 ? -v : v; }
__host__ __device__ CSW_INLINE static inline uint32_t lclip(int v) { return v < 0 ? 0u : v > 255 ? 255u : uint32_t(v); }
__host__ __device__ CSW_INLINE static inline uint32_t lselect(uint32_t L, uint32_t T, uint32_t TL) {
    int pl = 0, pt = 0;   // distance of the gradient estimate L + T - TL to L and to T
    for (int s = 0; s < 32; s += 8) { const int l = int((L >> s) & 255u), t = int((T >> s) & 255u), tl = int((TL >> s) & 255u); pl += labs(t - tl); pt += labs(l - tl); }
    return pl < pt ? L : T;
}
__host__ __device__ CSW_INLINE static inline uint32_t lclamp_full(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r = 0;
    for (int s = 0; s < 32; s += 8) r |= lclip(int((a >> s) & 255u) + int((b >> s) & 255u) - int((c >> s) & 255u)) << s;
    return r;
}
__host__ __device__ CSW_INLINE static inline uint32_t lclamp_half(uint32_t a, uint32_t b) {
    uint32_t r = 0;
    for (int s = 0; s < 32; s += 8) { const int x = int((a >> s) & 255u), y = int((b >> s) & 255u); r |= lclip(x + (x - y) / 2) << s; }
    return r;
}
__host__ __device__ CSW_INLINE static inline uint32_t lpredict_vals(int mode, uint32_t L, uint32_t T, uint32_t TR, uint32_t TL) {
    switch (mode) {
    case 0: return 0xFF000000u;
    case 1: return L;
    case 2: return T;
    case 3: return TR;
    case 4: return TL;
    case 5: return lavg(lavg(L, TR), T);
    case 6: return lavg(L, TL);
    case 7: return lavg(L, T);
    case 8: return lavg(TL, T);
    case 9: return lavg(T, TR);
    case 10: return lavg(lavg(L, TL), lavg(T, TR));
    case 11: return lselect(L, T, TL);
    case 12: return lclamp_full(L, T, TL);
    case 13: return lclamp_half(lavg(L, T), TL);
    default: return 0xFF000000u;   // modes 14, 15: libwebp treats them as "black"
    }
}
__host__ __device__ CSW_INLINE static inline uint32_t lpredict(int mode, const uint32_t *px, int w) {   // px: the pixel being decoded, in a frame of width w
    return lpredict_vals(mode, px[-1], px[-w], px[-w + 1], px[-w - 1]);
}

struct LTransform { int type, bits; uint32_t xsize; uint32_t *data; uint32_t ncolors; };

// What the pixel loop touches per symbol, kept close (LDS in the kernel; one lane walks the stream, and a look-up in the arena is a round trip to the memory
// system): the colour cache, the distance map, and the prefix-code groups in use -- a direct-mapped cache of L_SLOTS groups (their descriptions and their
// 256-entry tables; the symbols of codes longer than 8 bits stay in the arena).
enum { L_SLOTS = 8 };
struct LHot {
    uint32_t cache[2048];
    uint16_t lut[L_SLOTS][5][256];
    LCode code[L_SLOTS][5];
    uint32_t tag[L_SLOTS];      // group index + 1 held by the slot (0: none)
    uint8_t plane[120];
    alignas(16) uint8_t win[L_WIN + 64];   // the bit reader's window of the stream
};
struct LDec {
    LBits br;
    LArena ar;
    LHot *hot;
    uint32_t *cache;      // 2048 entries (hot->cache)
    uint8_t *lengths;     // scratch: 2328 bytes
    uint8_t *plane;       // hot->plane
};
// a symbol of code k of the group in slot `sl`.  `fl` = what the loop keeps of that code in a register (a symbol is a chain of dependent look-ups, and each
// one in LDS costs a round trip): bit 31 the code has a single symbol (bits 0-15, costs no bits), bit 30 it has a 256-entry table
__host__ __device__ CSW_INLINE static inline int lsym_hot(LBits &br, const LHot &h, int sl, int k, uint32_t fl, const uint16_t *arena) {
    if (fl & 0x80000000u) return int(fl & 0xFFFFu);
    if (fl & 0x40000000u) {
        const uint16_t e = h.lut[sl][k][br.peek(8)];
        if (e) { br.drop(e >> 12); return e & 0xFFF; }
    }
    const LCode &c = h.code[sl][k];
    uint32_t code = 0;
    for (int len = 1; len <= 15; len++) {
        code = (code << 1) | br.read(1);
        const uint32_t d = code - c.first_code[len];
        if (d < c.count[len]) return arena[c.syms + c.first_idx[len] + d];
    }
    br.eos = true;   // cannot happen with a complete code
    return 0;
}
// An entropy-coded ARGB image of xs x ys pixels, in two halves.  lsetup_pixels (not inlined: set-up code, one copy) reads the colour-cache size, the meta prefix
// image (level0: the picture itself may have one; meta_buf receives it) and the prefix codes of every group, into the arena.  lrun_pixels is the pixel loop;
// it exists twice: inlined into the kernel for the picture (there the compiler knows that `hot` is LDS and that the bit reader -- a local copy nothing else
// sees -- stays in registers), and once more inside ldecode_sub for the small images (meta prefix image, transform data, palette).
struct LPix { int cache_bits, prec; uint32_t mw, ngroups; LGroup *groups; };
__host__ __device__ static int ldecode_sub(LDec &d, uint32_t xs, uint32_t ys, uint32_t *out);
template <bool level0>
__host__ __device__ CSW_NOINLINE static int lsetup_pixels(LDec &d, uint32_t xs, uint32_t ys, uint32_t *meta_buf, uint64_t meta_cap, LPix &px) {
    LBits &br = d.br;
    int cache_bits = 0;
    if (br.read(1)) { cache_bits = int(br.read(4)); if (cache_bits < 1 || cache_bits > 11) return 1; }
    int prec = 0; uint32_t mw = 0; uint32_t ngroups = 1;
    if (level0 && br.read(1)) {
        const uint64_t arena_mark = d.ar.used;
        prec = int(br.read(3)) + 2;
        mw = (xs + (1u << prec) - 1) >> prec;
        const uint32_t mh = (ys + (1u << prec) - 1) >> prec;
        if (uint64_t(mw) * mh > meta_cap) return 1;
        const int rc = ldecode_sub(d, mw, mh, meta_buf);
        d.ar.used = arena_mark;   // the meta image's codes are done with
        if (rc) return rc;
        uint32_t mx = 0;
        for (uint64_t i = 0; i < uint64_t(mw) * mh; i++) { meta_buf[i] = (meta_buf[i] >> 8) & 0xFFFFu; if (meta_buf[i] > mx) mx = meta_buf[i]; }
        ngroups = mx + 1;
    }
    // the groups' descriptions live in the arena too (u16 units)
    const uint64_t gwords = (sizeof(LGroup) + 1) / 2;
    d.ar.used = (d.ar.used + 3) & ~uint64_t(3);   // 8-byte alignment
    if (d.ar.used + gwords * ngroups > d.ar.cap) return 2;
    LGroup *groups = reinterpret_cast<LGroup *>(d.ar.base + d.ar.used);
    d.ar.used += gwords * ngroups;
    const int cache_size = cache_bits ? 1 << cache_bits : 0;
    // tables of 256 entries for every code while a quarter of the arena is free; beyond that the codes are decoded length by length
    for (uint32_t g = 0; g < ngroups; g++) {
        const int alpha[5] = {256 + 24 + cache_size, 256, 256, 256, 40};
        for (int k = 0; k < 5; k++) {
            const bool want_lut = d.ar.used + 4096 < d.ar.cap - d.ar.cap / 4;
            if (!lread_code(br, alpha[k], groups[g].c[k], d.ar, d.lengths, want_lut)) return br.eos ? 1 : (d.ar.used + 4096 >= d.ar.cap ? 2 : 1);
        }
    }
    px.cache_bits = cache_bits; px.prec = prec; px.mw = mw; px.ngroups = ngroups; px.groups = groups;
    return 0;
}
__host__ __device__ CSW_INLINE static inline int lrun_pixels(LDec &d, LHot &hot, const LPix &px, uint32_t xs, uint32_t ys, uint32_t *out, const uint32_t *meta_buf) {
    LBits br = d.br;        // the reader of the loop: a copy nothing else sees; handed back at the end
    br.win = hot.win;
    const int cache_bits = px.cache_bits, prec = px.prec;
    const uint32_t mw = px.mw;
    const LGroup *groups = px.groups;
    const int cache_size = cache_bits ? 1 << cache_bits : 0;
    uint32_t *const cache = hot.cache;
    const uint8_t *const plane = hot.plane;
    if (cache_bits) { CSW_NOUNROLL for (int i = 0; i < cache_size; i++) cache[i] = 0; }
    const uint64_t total = uint64_t(xs) * ys;
    uint64_t pos = 0;
    uint32_t x = 0, y = 0;
    const uint32_t pmask = prec ? (1u << prec) - 1 : 0xFFFFFFFFu;
    const uint16_t *A = d.ar.base;
    for (int i = 0; i < L_SLOTS; i++) hot.tag[i] = 0;
    int sl = 0;
    uint32_t fl0 = 0, fl1 = 0, fl2 = 0, fl3 = 0, fl4 = 0;   // the five codes of the group in use (lsym_hot)
    auto flags_of = [&](int k) -> uint32_t { const LCode &c = hot.code[sl][k]; return c.nsym <= 1 ? (0x80000000u | c.single) : (c.lut != 0xFFFFFFFFu ? 0x40000000u : 0u); };
    auto take_flags = [&]() { fl0 = flags_of(0); fl1 = flags_of(1); fl2 = flags_of(2); fl3 = flags_of(3); fl4 = flags_of(4); };
    auto load_group = [&](uint32_t g) {   // group g into its slot, unless it is there
        const int was = sl;
        sl = int(g & uint32_t(L_SLOTS - 1));
        if (hot.tag[sl] == g + 1) { if (sl != was) take_flags(); return; }
        hot.tag[sl] = g + 1;
        CSW_NOUNROLL
        for (int k = 0; k < 5; k++) {
            const LCode c = groups[g].c[k];
            hot.code[sl][k] = c;
            if (c.nsym > 1 && c.lut != 0xFFFFFFFFu) {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(A + c.lut);   // (tables start at even arena positions: see lbuild)
                uint32_t *dst = reinterpret_cast<uint32_t *>(hot.lut[sl][k]);
                CSW_NOUNROLL
                for (int i = 0; i < 128; i += 4) { const uint32_t a = src[i], b = src[i + 1], c2 = src[i + 2], d2 = src[i + 3]; dst[i] = a; dst[i + 1] = b; dst[i + 2] = c2; dst[i + 3] = d2; }
            }
        }
        take_flags();
    };
    auto pick = [&]() { if (prec) load_group(meta_buf[uint64_t(y >> prec) * mw + (x >> prec)]); };
    auto put_cache = [&](uint32_t v) { if (cache_bits) cache[(0x1E35A7BDu * v) >> (32 - cache_bits)] = v; };
    load_group(0);
    pick();
    while (pos < total) {
        if ((x & pmask) == 0) pick();
        const int code = lsym_hot(br, hot, sl, 0, fl0, A);
        if (code < 256) {
            const uint32_t r = uint32_t(lsym_hot(br, hot, sl, 1, fl1, A)), b = uint32_t(lsym_hot(br, hot, sl, 2, fl2, A)), a = uint32_t(lsym_hot(br, hot, sl, 3, fl3, A));
            const uint32_t v = (a << 24) | (r << 16) | (uint32_t(code) << 8) | b;
            out[pos++] = v; put_cache(v);
            if (++x == xs) { x = 0; y++; }
        } else if (code < 256 + 24) {
            const uint32_t len = lprefix_value(br, code - 256);
            const int ds = lsym_hot(br, hot, sl, 4, fl4, A);
            const uint32_t dcode = lprefix_value(br, ds);
            uint64_t dist;
            if (dcode > 120) dist = dcode - 120;
            else {
                const uint8_t e = plane[dcode - 1];
                const int64_t dd = int64_t(e >> 4) * xs + (8 - int(e & 15));
                dist = dd < 1 ? 1 : uint64_t(dd);
            }
            if (dist > pos || pos + len > total) { d.br = br; return 1; }
            for (uint32_t i = 0; i < len; i++) { const uint32_t v = out[pos - dist]; out[pos++] = v; put_cache(v); }
            x += len;
            while (x >= xs) { x -= xs; y++; }
            if (pos < total && prec) pick();
        } else {
            const int idx = code - (256 + 24);
            if (idx >= cache_size) { d.br = br; return 1; }
            const uint32_t v = cache[idx];
            out[pos++] = v; put_cache(v);
            if (++x == xs) { x = 0; y++; }
        }
        if (br.eos) { d.br = br; return 1; }
    }
    d.br = br;
    return 0;
}
// a small image (meta prefix image, transform data, palette) from set-up to pixels; not inlined
__host__ __device__ CSW_NOINLINE static int ldecode_sub(LDec &d, uint32_t xs, uint32_t ys, uint32_t *out) {
    LPix px;
    const int rc = lsetup_pixels<false>(d, xs, ys, nullptr, 0, px);
    return rc ? rc : lrun_pixels(d, *d.hot, px, xs, ys, out, nullptr);
}



// what the entropy layer leaves for the pixel stages (vp8l_transform_step, vp8l_output_*): the transforms in the order they were read, in the work area
struct LTrRec { uint32_t type, bits, xsize, ncolors; };
struct LFrame { uint32_t ntr, xs, alph_filter, alph_raw; LTrRec tr[4]; };
struct LWork { uint32_t *frame0, *frame1, *subimg[3], *palette; LFrame *info; uint64_t npx, sub; };
__host__ __device__ CSW_INLINE static inline LWork lwork(uint8_t *work, uint32_t W, uint32_t H) {
    LWork w;
    w.npx = uint64_t(W) * H; w.sub = vp8l_sub_pixels(W, H);
    w.frame0 = reinterpret_cast<uint32_t *>(work); w.frame1 = w.frame0 + w.npx;
    w.subimg[0] = w.frame1 + w.npx; w.subimg[1] = w.subimg[0] + w.sub; w.subimg[2] = w.subimg[1] + w.sub;
    w.palette = w.subimg[2] + w.sub;
    w.info = reinterpret_cast<LFrame *>(w.palette + 256);   // (8 KiB here were the colour cache's before it moved to the hot set)
    return w;
}
// The entropy layer of one VP8L stream -- ONE lane: transforms' sub-images, palette, then the picture into frame0, all still transformed; leaves LFrame.
// 0 ok, 1 malformed, 2 beyond this build (work area exhausted).  headerless: the stream of an ALPH chunk -- no signature, no sizes (W x H are the picture's).
__host__ __device__ CSW_INLINE static inline int vp8l_entropy(const uint8_t *data, size_t n, uint32_t W, uint32_t H, uint8_t *work, uint64_t file_bytes, bool headerless, LHot *hot) {
    LDec d;
    d.hot = hot;
    d.br.init(data, n, hot->win);
    LBits &br = d.br;
    if (!headerless) {
        if (br.read(8) != 0x2F) return 1;
        const uint32_t w = br.read(14) + 1, h = br.read(14) + 1;
        br.read(1);   // alpha_is_used: a hint; the pixels decide
        if (br.read(3) != 0 || w != W || h != H) return 1;
    }
    const LWork lw = lwork(work, W, H);
    const uint64_t sub = lw.sub;
    uint32_t *frame0 = lw.frame0, *const *subimg = lw.subimg, *palette = lw.palette;
    d.cache = hot->cache;
    d.ar.base = reinterpret_cast<uint16_t *>(palette + 256 + 2048);   // (the 8 KiB in between were the colour cache's before it moved to the hot set)
    d.ar.cap = vp8l_arena_bytes(file_bytes) / 2 - 2328 / 2 - 8; d.ar.used = 0;
    d.lengths = reinterpret_cast<uint8_t *>(d.ar.base + d.ar.cap);
    d.plane = hot->plane;
    lplane_table(d.plane);
    // ---- transforms, in the order they are undone LAST to FIRST
    LTransform tr[4];
    int ntr = 0, seen = 0;
    uint32_t xs = W;
    while (br.read(1)) {
        const int type = int(br.read(2));
        if (seen & (1 << type)) return 1;
        seen |= 1 << type;
        LTransform &t = tr[ntr++];
        t.type = type; t.bits = 0; t.xsize = xs; t.data = nullptr; t.ncolors = 0;
        if (type == 0 || type == 1) {
            t.bits = int(br.read(3)) + 2;
            const uint32_t bw = (xs + (1u << t.bits) - 1) >> t.bits, bh = (H + (1u << t.bits) - 1) >> t.bits;
            if (uint64_t(bw) * bh > sub) return 1;
            t.data = subimg[type];
            const uint64_t keep = d.ar.used;
            const int rc = ldecode_sub(d, bw, bh, t.data);
            d.ar.used = keep;   // the sub-image's codes are done with
            if (rc) return rc;
        } else if (type == 3) {
            t.ncolors = br.read(8) + 1;
            t.bits = t.ncolors > 16 ? 0 : t.ncolors > 4 ? 1 : t.ncolors > 2 ? 2 : 3;
            t.data = palette;
            const uint64_t keep = d.ar.used;
            const int rc = ldecode_sub(d, t.ncolors, 1, palette);
            d.ar.used = keep;
            if (rc) return rc;
            for (uint32_t i = 1; i < t.ncolors; i++) palette[i] = ladd(palette[i], palette[i - 1]);
            for (uint32_t i = t.ncolors; i < 256; i++) palette[i] = 0;
            xs = (xs + (1u << t.bits) - 1) >> t.bits;
        }
        if (br.eos) return 1;
    }
    // ---- the picture itself
    {
        LPix px;
        int rc = lsetup_pixels<true>(d, xs, H, subimg[2], sub, px);
        if (!rc) rc = lrun_pixels(d, *hot, px, xs, H, frame0, subimg[2]);   // `hot`: the kernel's LDS, as the compiler can see here
        if (rc) return rc;
    }
    LFrame &f = *lw.info;
    f.ntr = uint32_t(ntr); f.xs = xs; f.alph_filter = 0; f.alph_raw = 0;
    for (int k = 0; k < ntr; k++) { f.tr[k].type = uint32_t(tr[k].type); f.tr[k].bits = uint32_t(tr[k].bits); f.tr[k].xsize = tr[k].xsize; f.tr[k].ncolors = tr[k].ncolors; }
    return 0;
}

// ---- the pixel stages: every lane of the picture's workgroup (k_vp8l_pixels).  The transforms are undone last to first; undo slot j is transform ntr - 1 - j.
// Subtract-green, cross-colour and colour indexing are a pass over the pixels each (one step); the predictor is a wave front over the rows: a pixel needs its
// left, upper, upper-left and upper-RIGHT neighbours, so row r takes the L_CHUNK pixels from L_CHUNK (t - r) - r in step t -- one pixel behind the row above
// for every row, vp8l_pred_steps(W, H) steps in all.
enum { L_CHUNK = 8 };
__host__ __device__ CSW_INLINE static inline uint32_t vp8l_pred_steps(uint32_t W, uint32_t H) { return H + (W + H + L_CHUNK - 1) / L_CHUNK + 2; }
// the frame the undo slot j reads (colour indexing, undone at most once, writes the other one)
__host__ __device__ CSW_INLINE static inline bool vp8l_swapped_before(const LFrame &f, int j) {
    for (int i = 0; i < j && i < int(f.ntr); i++) if (f.tr[f.ntr - 1 - uint32_t(i)].type == 3) return true;
    return false;
}
__host__ __device__ CSW_INLINE static inline void vp8l_transform_step(const LWork &lw, uint32_t H, int j, uint32_t step, uint32_t tid, uint32_t nlanes) {
    const LFrame &f = *lw.info;
    const LTrRec t = f.tr[f.ntr - 1 - uint32_t(j)];
    const bool sw = vp8l_swapped_before(f, j);
    uint32_t *cur = sw ? lw.frame1 : lw.frame0, *other = sw ? lw.frame0 : lw.frame1;
    const uint32_t tw = t.xsize;
    if (t.type == 2) {
        if (step) return;
        const uint64_t cnt = uint64_t(tw) * H;
        for (uint64_t i = tid; i < cnt; i += nlanes) { const uint32_t v = cur[i], g = (v >> 8) & 255u; cur[i] = (v & 0xFF00FF00u) | ((((v & 0x00FF00FFu) + ((g << 16) | g))) & 0x00FF00FFu); }
    } else if (t.type == 1) {
        if (step) return;
        const uint32_t bw = (tw + (1u << t.bits) - 1) >> t.bits;
        const uint32_t *data = lw.subimg[1];
        const uint64_t cnt = uint64_t(tw) * H;
        for (uint64_t i = tid; i < cnt; i += nlanes) {
            const uint32_t y = uint32_t(i / tw), x = uint32_t(i - uint64_t(y) * tw);
            const uint32_t m = data[uint64_t(y >> t.bits) * bw + (x >> t.bits)];
            const int8_t g2r = int8_t(m & 255u), g2b = int8_t((m >> 8) & 255u), r2b = int8_t((m >> 16) & 255u);
            const uint32_t p = cur[i];
            const int8_t green = int8_t((p >> 8) & 255u);
            int nr = int((p >> 16) & 255u), nb = int(p & 255u);
            nr = (nr + ((int(g2r) * int(green)) >> 5)) & 255;
            nb = (nb + ((int(g2b) * int(green)) >> 5) + ((int(r2b) * int(int8_t(nr))) >> 5)) & 255;
            cur[i] = (p & 0xFF00FF00u) | (uint32_t(nr) << 16) | uint32_t(nb);
        }
    } else if (t.type == 0) {
        const uint32_t bw = (tw + (1u << t.bits) - 1) >> t.bits;
        const uint32_t *data = lw.subimg[0];
        for (uint32_t r = tid; r < H; r += nlanes) {
            const int64_t x0 = int64_t(L_CHUNK) * (int64_t(step) - int64_t(r)) - int64_t(r);
            if (x0 + L_CHUNK <= 0 || x0 >= int64_t(tw)) continue;
            const uint32_t xa = x0 < 0 ? 0u : uint32_t(x0), xb = x0 + L_CHUNK > int64_t(tw) ? tw : uint32_t(x0 + L_CHUNK);
            uint32_t *row = cur + uint64_t(r) * tw;
            for (uint32_t x = xa; x < xb; x++) {
                uint32_t pred;
                if (r == 0) pred = x == 0 ? 0xFF000000u : row[x - 1];
                else if (x == 0) pred = row[-int64_t(tw)];
                else pred = lpredict(int((data[uint64_t(r >> t.bits) * bw + (x >> t.bits)] >> 8) & 15u), row + x, int(tw));
                row[x] = ladd(row[x], pred);
            }
        }
    } else {
        // colour indexing: the frame holds 1 << bits indices per pixel (bits > 0), the low ones first
        if (step) return;
        const uint32_t pw = (tw + (1u << t.bits) - 1) >> t.bits;
        const int per = 1 << t.bits, nb = 8 >> t.bits;
        const uint64_t cnt = uint64_t(tw) * H;
        for (uint64_t i = tid; i < cnt; i += nlanes) {
            const uint32_t y = uint32_t(i / tw), x = uint32_t(i - uint64_t(y) * tw);
            const uint32_t packed = (cur[uint64_t(y) * pw + (x >> t.bits)] >> 8) & 255u;
            const uint32_t idx = t.bits ? (packed >> (nb * int(x & uint32_t(per - 1)))) & ((1u << nb) - 1u) : packed;
            other[i] = idx < t.ncolors ? lw.palette[idx] : 0u;
        }
    }
}
// the finished ARGB frame
__host__ __device__ CSW_INLINE static inline const uint32_t *vp8l_result(const LWork &lw) { return vp8l_swapped_before(*lw.info, int(lw.info->ntr)) ? lw.frame1 : lw.frame0; }

// The ALPH chunk of a lossy file (WebP container specification): a header byte -- compression in bits 0-1 (0 raw, 1 a headerless VP8L stream), filter in bits
// 2-3 (none, horizontal, vertical, gradient), pre-processing in bits 4-5 (nothing to undo) -- then the plane; the filters predict a sample from its left /
// upper neighbours as libwebp's unfilters do (first row: from the left, first sample 0; first sample of a later row: from the one above).
// alph_entropy: ONE lane; 0 ok, else as vp8l_entropy.  The plane itself is made by the pixel stages: the transforms, then alph_plane_step (green channel or the
// raw bytes into aplane), then alph_unfilter_step as a wave front like the predictor's (a sample needs its left, upper and upper-left neighbours).
__host__ __device__ CSW_INLINE static inline int alph_entropy(const uint8_t *d, size_t n, uint32_t W, uint32_t H, uint8_t *work, LHot *hot) {
    if (n < 1) return 1;
    const int method = d[0] & 3, filter = (d[0] >> 2) & 3, pre = (d[0] >> 4) & 3, rsrv = (d[0] >> 6) & 3;
    const uint64_t npx = uint64_t(W) * H;
    if (method > 1 || pre > 1 || rsrv > 1) return 1;
    LFrame &f = *lwork(work, W, H).info;
    if (method == 0) {
        if (uint64_t(n - 1) < npx) return 1;
        f.ntr = 0; f.xs = W; f.alph_raw = 1;
    } else {
        const int rc = vp8l_entropy(d + 1, n - 1, W, H, work, n - 1, true, hot);
        if (rc) return rc;
    }
    f.alph_filter = uint32_t(filter);
    return 0;
}
__host__ __device__ CSW_INLINE static inline void alph_plane_step(const LWork &lw, const uint8_t *chunk, uint8_t *aplane, uint32_t tid, uint32_t nlanes) {
    if (lw.info->alph_raw) { for (uint64_t i = tid; i < lw.npx; i += nlanes) aplane[i] = chunk[1 + i]; return; }
    const uint32_t *cur = vp8l_result(lw);
    for (uint64_t i = tid; i < lw.npx; i += nlanes) aplane[i] = uint8_t(cur[i] >> 8);   // the plane travels in the green channel
}
__host__ __device__ CSW_INLINE static inline void alph_unfilter_step(const LWork &lw, uint8_t *aplane, uint32_t W, uint32_t H, uint32_t step, uint32_t tid, uint32_t nlanes) {
    const uint32_t filter = lw.info->alph_filter;
    if (!filter) return;
    for (uint32_t r = tid; r < H; r += nlanes) {
        const int64_t x0 = int64_t(L_CHUNK) * (int64_t(step) - int64_t(r)) - int64_t(r);
        if (x0 + L_CHUNK <= 0 || x0 >= int64_t(W)) continue;
        const uint32_t xa = x0 < 0 ? 0u : uint32_t(x0), xb = x0 + L_CHUNK > int64_t(W) ? W : uint32_t(x0 + L_CHUNK);
        uint8_t *row = aplane + uint64_t(r) * W;
        const uint8_t *prev = r ? row - W : nullptr;
        for (uint32_t x = xa; x < xb; x++) {
            int pred;
            if (!prev || filter == 1) pred = x ? row[x - 1] : (prev ? prev[0] : 0);   // horizontal (and every filter's first row)
            else if (filter == 2) pred = prev[x];
            else {   // gradient; the first sample of a row from the one above
                const int top = prev[x], left = x ? row[x - 1] : prev[0], top_left = x ? prev[x - 1] : prev[0];
                const int g = left + top - top_left;
                pred = g < 0 ? 0 : g > 255 ? 255 : g;
            }
            row[x] = uint8_t((int(row[x]) + pred) & 255);
        }
    }
}

}  // namespace csw
