// vp8_dec.h -- VP8 key-frame decoder (RFC 6386), the lossy WebP inputs of libcaesium's webp::compress / convert paths
// (/root/reference/src/compressor.rs:289-305, 589-598 name WebP among the inputs; libcaesium decodes them with libwebp).  The frame's
// boolean-coded data is a serial chain: ONE lane parses it and reconstructs the macroblocks as it goes (vp8_parse_frame, its working set in LDS);
// the loop filter then runs as a wave front over the macroblock rows and libwebp's "fancy" chroma upsampling + fixed-point YCbCr -> RGB by
// row pairs, both across the lanes of the picture's workgroup (k_webp_dec.hip).  Pixel-exact against libwebp (through Pillow) in
// tests/test_webp_decode*.py.
// Everything here is host + device code: the emulation build compiles it as plain C++.
#pragma once
#include <cstdint>
#include "../../include/vp8_tables.h"

// everything of the decoders is inlined into its kernel: only then does the compiler see which pointers are LDS (ds_read instead of flat_load)
#ifdef CSH_EMUL
#define CSW_INLINE
#define CSW_NOUNROLL
#define CSW_U(x) (x)
#else
// a value every lane of the wave holds alike, moved to a scalar register: what is computed from it then runs on the scalar unit.  The parse is ONE chain of
// dependent operations; as vector instructions of a wave with one live lane each costs ~8 cycles, as scalar instructions a fraction of that -- so the
// whole wave walks the chain in step (every lane the same values, the same stores) and the loaded values are declared uniform here.
#define CSW_U(x) (static_cast<decltype(x)>(__builtin_amdgcn_readfirstlane(static_cast<int>(x))))
#define CSW_INLINE __attribute__((always_inline))
#define CSW_NOUNROLL _Pragma("clang loop unroll(disable)")   // (the coefficient reader is inlined: one copy per call site, not per block)
#endif
namespace csw {

struct Vp8In {              // one input file, host-parsed container
    uint64_t data_off;      // VP8 chunk payload in the input pool
    uint32_t data_len;
    uint32_t width, height, mbw, mbh;
    uint64_t work_off;      // per-image work area in the work pool (layout below)
    uint64_t rgb_off;       // width * height * 3 bytes in the pixel pool
    uint32_t status;        // device: 0 ok, else an error code
    uint32_t lossless;      // 1: the payload is a VP8L stream (vp8l_dec.h), work area sized by vp8l_work_bytes
    uint32_t has_alpha;     // device: 1 when the picture is not opaque -- then rgba_off / a_off hold it as RGBA and its alpha plane as well
    uint64_t alph_off;      // lossy files: the ALPH chunk's payload in the input pool (alph_len 0: none)
    uint32_t alph_len, debug;   // debug: CSH_WEBP_DEBUG (timing probes: 1 no reconstruction, 2 no loop filter, 4 no RGB)
    uint64_t rgba_off, a_off;   // width * height * 4 and width * height bytes in the pixel pool (~0: not reserved)
};
// work area: Y plane (mbw*16 x mbh*16), U, V (mbw*8 x mbh*8), per-macroblock filter info (4 bytes), per-column contexts, the frame info, and what the
// parse hands the reconstruction: per macroblock a record of its modes (24 bytes) and its 24 dequantised 4 x 4 blocks (768 bytes, only those in nzmask written)
__host__ __device__ CSW_INLINE static inline uint64_t vp8_work_bytes(uint32_t mbw, uint32_t mbh) {
    const uint64_t ly = uint64_t(mbw) * 16 * mbh * 16, lc = uint64_t(mbw) * 8 * mbh * 8;
    return ly + 2 * lc + uint64_t(mbw) * mbh * 4 + uint64_t(mbw) * 16 + 256 + uint64_t(mbw) * mbh * (24 + 768);
}

// RFC 6386 section 7's boolean decoder with a wide window: `value` holds the decoder's 8-bit value and nbits bits of look-ahead below it, so
// bytes come in four at a time and the renormalisation is one count-leading-zeros shift instead of a loop (the decisions depend on the top
// eight bits only: value16 >= split << 8 is top8 >= split).  eof() answers what the byte-wise form's flag did: a byte past the end has been
// shifted INTO the 16-bit window (the reference form loads its next byte after every eighth shift).
struct BoolDec {
    const uint8_t *p, *end;
    uint64_t value;
    uint32_t range, loaded, len;
    int nbits;
    // (the bytes come straight from the memory system, four per 32 bits of look-ahead: a window of the stream in LDS was measured and bought nothing)
    __host__ __device__ CSW_INLINE void refill() {
        uint32_t w;
        if (end - p >= 4) { w = (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); p += 4; }
        else { w = 0; for (int i = 0; i < 4; i++) w = (w << 8) | (p < end ? uint32_t(*p++) : 0u); }
        w = CSW_U(w);
        value = (value << 32) | w; nbits += 32; loaded += 4;
    }
    __host__ __device__ CSW_INLINE void init(const uint8_t *d, size_t n) {
        p = d; end = d + n; len = uint32_t(n);
        value = 0; nbits = -8; loaded = 0; range = 255;
        refill();
    }
    __host__ __device__ CSW_INLINE int get(int prob) {
        prob = CSW_U(prob);
        if (nbits < 8) refill();
        const uint32_t split = 1u + (((range - 1u) * uint32_t(prob)) >> 8);
        const uint32_t top = uint32_t(value >> nbits);
        int r;
        if (top >= split) { r = 1; range -= split; value -= uint64_t(split) << nbits; } else { r = 0; range = split; }
        const int shift = __builtin_clz(range) - 24;
        range <<= shift; nbits -= shift;
        return r;
    }
    __host__ __device__ CSW_INLINE bool eof() const {
        const int64_t groups = (int64_t(8) * loaded - 8 - nbits) >> 3;   // completed groups of eight shifts = bytes the byte-wise form has loaded behind its first two
        return groups >= 1 && groups + 1 >= int64_t(len);
    }
    __host__ __device__ CSW_INLINE uint32_t lit(int n) { uint32_t v = 0; while (n-- > 0) v = (v << 1) | uint32_t(get(128)); return v; }
    __host__ __device__ CSW_INLINE int slit(int n) { const int v = int(lit(n)); return get(128) ? -v : v; }
};

__host__ __device__ CSW_INLINE static inline int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
__host__ __device__ CSW_INLINE static inline int clipq(int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; }

// inverse DCT of one 4x4 block, added to the prediction in dst (libwebp TransformOne: the two multipliers of RFC 6386 14.3)
__host__ __device__ CSW_INLINE static inline void vp8_idct_add(const int16_t *in16, uint8_t *dst, int stride) {
    // the block comes out of the work area (16-byte aligned): two loads up front instead of thirty-two the compiler has to order against the stores below
    const uint4 q0 = reinterpret_cast<const uint4 *>(in16)[0], q1 = reinterpret_cast<const uint4 *>(in16)[1];
    const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    int in[16];
    for (int i = 0; i < 8; i++) { in[2 * i] = int16_t(w[i] & 0xFFFFu); in[2 * i + 1] = int16_t(w[i] >> 16); }
    int tmp[16];
    for (int i = 0; i < 4; i++) {
        const int a = in[i] + in[8 + i], b = in[i] - in[8 + i];
        const int c = ((in[4 + i] * 35468) >> 16) - (((in[12 + i] * 20091) >> 16) + in[12 + i]);
        const int d = (((in[4 + i] * 20091) >> 16) + in[4 + i]) + ((in[12 + i] * 35468) >> 16);
        tmp[4 * i] = a + d; tmp[4 * i + 1] = b + c; tmp[4 * i + 2] = b - c; tmp[4 * i + 3] = a - d;
    }
    for (int i = 0; i < 4; i++) {
        const int dc = tmp[i] + 4;
        const int a = dc + tmp[8 + i], b = dc - tmp[8 + i];
        const int c = ((tmp[4 + i] * 35468) >> 16) - (((tmp[12 + i] * 20091) >> 16) + tmp[12 + i]);
        const int d = (((tmp[4 + i] * 20091) >> 16) + tmp[4 + i]) + ((tmp[12 + i] * 35468) >> 16);
        uint8_t *o = dst + i * stride;
        o[0] = uint8_t(clip8(o[0] + ((a + d) >> 3))); o[1] = uint8_t(clip8(o[1] + ((b + c) >> 3)));
        o[2] = uint8_t(clip8(o[2] + ((b - c) >> 3))); o[3] = uint8_t(clip8(o[3] + ((a - d) >> 3)));
    }
}
// inverse Walsh-Hadamard of the 16 luma DCs (libwebp TransformWHT); out[k * 16] = DC of block k
__host__ __device__ CSW_INLINE static inline void vp8_iwht(const int16_t *in, int16_t *out) {
    int tmp[16];
    for (int i = 0; i < 4; i++) {
        const int a0 = in[i] + in[12 + i], a1 = in[4 + i] + in[8 + i], a2 = in[4 + i] - in[8 + i], a3 = in[i] - in[12 + i];
        tmp[i] = a0 + a1; tmp[8 + i] = a0 - a1; tmp[4 + i] = a3 + a2; tmp[12 + i] = a3 - a2;
    }
    for (int i = 0; i < 4; i++) {
        const int dc = tmp[4 * i] + 3;
        const int a0 = dc + tmp[4 * i + 3], a1 = tmp[4 * i + 1] + tmp[4 * i + 2], a2 = tmp[4 * i + 1] - tmp[4 * i + 2], a3 = dc - tmp[4 * i + 3];
        out[(4 * i) * 16] = int16_t((a0 + a1) >> 3); out[(4 * i + 1) * 16] = int16_t((a3 + a2) >> 3);
        out[(4 * i + 2) * 16] = int16_t((a0 - a1) >> 3); out[(4 * i + 3) * 16] = int16_t((a3 - a2) >> 3);
    }
}

// ---- intra prediction.  `d` points at the block's top-left sample in a plane with `s` bytes per row whose row above and column to the
// left hold the neighbours (frame edges: 127 above, 129 to the left, as libwebp initialises them)
__host__ __device__ CSW_INLINE static inline void pred_fill(uint8_t *d, int s, int n, int v) { for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) d[y * s + x] = uint8_t(v); }
__host__ __device__ CSW_INLINE static inline void pred_tm(uint8_t *d, int s, int n) {
    const int tl = d[-s - 1];
    for (int y = 0; y < n; y++) { const int l = d[y * s - 1]; for (int x = 0; x < n; x++) d[y * s + x] = uint8_t(clip8(l + d[-s + x] - tl)); }
}
__host__ __device__ CSW_INLINE static inline void pred_v(uint8_t *d, int s, int n) { for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) d[y * s + x] = d[-s + x]; }
__host__ __device__ CSW_INLINE static inline void pred_h(uint8_t *d, int s, int n) { for (int y = 0; y < n; y++) { const uint8_t l = d[y * s - 1]; for (int x = 0; x < n; x++) d[y * s + x] = l; } }
// DC of an n x n block (n = 16 or 8): has_top / has_left say which neighbours exist
__host__ __device__ CSW_INLINE static inline void pred_dc(uint8_t *d, int s, int n, bool has_top, bool has_left) {
    int sum = 0, cnt = 0;
    if (has_top) { for (int x = 0; x < n; x++) sum += d[-s + x]; cnt += n; }
    if (has_left) { for (int y = 0; y < n; y++) sum += d[y * s - 1]; cnt += n; }
    const int v = cnt ? (sum + (cnt >> 1)) / cnt : 128;
    pred_fill(d, s, n, v);
}
#define AVG3(a, b, c) uint8_t(((a) + 2 * (b) + (c) + 2) >> 2)
#define AVG2(a, b) uint8_t(((a) + (b) + 1) >> 1)
// the ten 4x4 modes (RFC 6386 12.3); tr = the four samples above and to the right
__host__ __device__ CSW_INLINE static inline void pred4(uint8_t *d, int s, int mode, const uint8_t *tr) {
    const int A = d[-s], B = d[-s + 1], C = d[-s + 2], D = d[-s + 3], E = tr[0], F = tr[1], G = tr[2], H = tr[3];
    const int I = d[-1], J = d[s - 1], K = d[2 * s - 1], L = d[3 * s - 1], X = d[-s - 1];
#define P(x, y) d[(y) * s + (x)]
    switch (mode) {
    case 0: { const int v = (A + B + C + D + I + J + K + L + 4) >> 3; pred_fill(d, s, 4, v); break; }          // B_DC_PRED
    case 1: pred_tm(d, s, 4); break;                                                                                // B_TM_PRED
    case 2: { const uint8_t v[4] = {AVG3(X, A, B), AVG3(A, B, C), AVG3(B, C, D), AVG3(C, D, E)};                   // B_VE_PRED
              for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) P(x, y) = v[x]; break; }
    case 3: { const uint8_t v[4] = {AVG3(X, I, J), AVG3(I, J, K), AVG3(J, K, L), AVG3(K, L, L)};                   // B_HE_PRED
              for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) P(x, y) = v[y]; break; }
    case 4:                                                                                                         // B_RD_PRED
        P(0, 3) = AVG3(J, K, L); P(1, 3) = P(0, 2) = AVG3(I, J, K); P(2, 3) = P(1, 2) = P(0, 1) = AVG3(X, I, J);
        P(3, 3) = P(2, 2) = P(1, 1) = P(0, 0) = AVG3(A, X, I); P(3, 2) = P(2, 1) = P(1, 0) = AVG3(B, A, X);
        P(3, 1) = P(2, 0) = AVG3(C, B, A); P(3, 0) = AVG3(D, C, B); break;
    case 5:                                                                                                         // B_VR_PRED
        P(0, 0) = P(1, 2) = AVG2(X, A); P(1, 0) = P(2, 2) = AVG2(A, B); P(2, 0) = P(3, 2) = AVG2(B, C); P(3, 0) = AVG2(C, D);
        P(0, 3) = AVG3(K, J, I); P(0, 2) = AVG3(J, I, X); P(0, 1) = P(1, 3) = AVG3(I, X, A); P(1, 1) = P(2, 3) = AVG3(X, A, B);
        P(2, 1) = P(3, 3) = AVG3(A, B, C); P(3, 1) = AVG3(B, C, D); break;
    case 6:                                                                                                         // B_LD_PRED
        P(0, 0) = AVG3(A, B, C); P(1, 0) = P(0, 1) = AVG3(B, C, D); P(2, 0) = P(1, 1) = P(0, 2) = AVG3(C, D, E);
        P(3, 0) = P(2, 1) = P(1, 2) = P(0, 3) = AVG3(D, E, F); P(3, 1) = P(2, 2) = P(1, 3) = AVG3(E, F, G);
        P(3, 2) = P(2, 3) = AVG3(F, G, H); P(3, 3) = AVG3(G, H, H); break;
    case 7:                                                                                                         // B_VL_PRED
        P(0, 0) = AVG2(A, B); P(1, 0) = P(0, 2) = AVG2(B, C); P(2, 0) = P(1, 2) = AVG2(C, D); P(3, 0) = P(2, 2) = AVG2(D, E);
        P(0, 1) = AVG3(A, B, C); P(1, 1) = P(0, 3) = AVG3(B, C, D); P(2, 1) = P(1, 3) = AVG3(C, D, E); P(3, 1) = P(2, 3) = AVG3(D, E, F);
        P(3, 2) = AVG3(E, F, G); P(3, 3) = AVG3(F, G, H); break;
    case 8:                                                                                                         // B_HD_PRED
        P(0, 0) = P(2, 1) = AVG2(I, X); P(0, 1) = P(2, 2) = AVG2(J, I); P(0, 2) = P(2, 3) = AVG2(K, J); P(0, 3) = AVG2(L, K);
        P(3, 0) = AVG3(A, B, C); P(2, 0) = AVG3(X, A, B); P(1, 0) = P(3, 1) = AVG3(I, X, A); P(1, 1) = P(3, 2) = AVG3(J, I, X);
        P(1, 2) = P(3, 3) = AVG3(K, J, I); P(1, 3) = AVG3(L, K, J); break;
    default:                                                                                                        // B_HU_PRED
        P(0, 0) = AVG2(I, J); P(2, 0) = P(0, 1) = AVG2(J, K); P(2, 1) = P(0, 2) = AVG2(K, L);
        P(1, 0) = AVG3(I, J, K); P(3, 0) = P(1, 1) = AVG3(J, K, L); P(3, 1) = P(1, 2) = AVG3(K, L, L);
        P(3, 2) = P(2, 2) = P(0, 3) = P(1, 3) = P(2, 3) = P(3, 3) = uint8_t(L); break;
    }
#undef P
}

// ---- loop filter (RFC 6386 section 15, libwebp's arithmetic)
__host__ __device__ CSW_INLINE static inline int sclip1(int v) { return v < -128 ? -128 : v > 127 ? 127 : v; }
__host__ __device__ CSW_INLINE static inline int sclip2(int v) { return v < -16 ? -16 : v > 15 ? 15 : v; }
__host__ __device__ CSW_INLINE static inline int iabs(int v) { return v < 0 ? -v : v; }
__host__ __device__ CSW_INLINE static inline void lf2(uint8_t *p, int st) {
    const int p1 = p[-2 * st], p0 = p[-st], q0 = p[0], q1 = p[st];
    const int a = 3 * (q0 - p0) + sclip1(p1 - q1);
    const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3);
    p[-st] = uint8_t(clip8(p0 + a2)); p[0] = uint8_t(clip8(q0 - a1));
}
__host__ __device__ CSW_INLINE static inline void lf4(uint8_t *p, int st) {
    const int p1 = p[-2 * st], p0 = p[-st], q0 = p[0], q1 = p[st];
    const int a = 3 * (q0 - p0);
    const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3), a3 = (a1 + 1) >> 1;
    p[-2 * st] = uint8_t(clip8(p1 + a3)); p[-st] = uint8_t(clip8(p0 + a2)); p[0] = uint8_t(clip8(q0 - a1)); p[st] = uint8_t(clip8(q1 - a3));
}
__host__ __device__ CSW_INLINE static inline void lf6(uint8_t *p, int st) {
    const int p2 = p[-3 * st], p1 = p[-2 * st], p0 = p[-st], q0 = p[0], q1 = p[st], q2 = p[2 * st];
    const int a = sclip1(3 * (q0 - p0) + sclip1(p1 - q1));
    const int a1 = (27 * a + 63) >> 7, a2 = (18 * a + 63) >> 7, a3 = (9 * a + 63) >> 7;
    p[-3 * st] = uint8_t(clip8(p2 + a3)); p[-2 * st] = uint8_t(clip8(p1 + a2)); p[-st] = uint8_t(clip8(p0 + a1));
    p[0] = uint8_t(clip8(q0 - a1)); p[st] = uint8_t(clip8(q1 - a2)); p[2 * st] = uint8_t(clip8(q2 - a3));
}
__host__ __device__ CSW_INLINE static inline bool lf_hev(const uint8_t *p, int st, int t) { return iabs(p[-2 * st] - p[-st]) > t || iabs(p[st] - p[0]) > t; }
__host__ __device__ CSW_INLINE static inline bool lf_needs(const uint8_t *p, int st, int t) { return 4 * iabs(p[-st] - p[0]) + iabs(p[-2 * st] - p[st]) <= t; }
__host__ __device__ CSW_INLINE static inline bool lf_needs2(const uint8_t *p, int st, int t, int it) {
    if (4 * iabs(p[-st] - p[0]) + iabs(p[-2 * st] - p[st]) > t) return false;
    return iabs(p[-4 * st] - p[-3 * st]) <= it && iabs(p[-3 * st] - p[-2 * st]) <= it && iabs(p[-2 * st] - p[-st]) <= it &&
           iabs(p[3 * st] - p[2 * st]) <= it && iabs(p[2 * st] - p[st]) <= it && iabs(p[st] - p[0]) <= it;
}
// an edge of `size` samples: hs = step across the edge, vs = step along it
__host__ __device__ CSW_INLINE static inline void lf_simple(uint8_t *p, int hs, int vs, int size, int thresh) {
    const int t2 = 2 * thresh + 1;
    for (int i = 0; i < size; i++, p += vs) if (lf_needs(p, hs, t2)) lf2(p, hs);
}
__host__ __device__ CSW_INLINE static inline void lf_edge(uint8_t *p, int hs, int vs, int size, int thresh, int ithresh, int hev, bool mb_edge) {
    const int t2 = 2 * thresh + 1;
    for (int i = 0; i < size; i++, p += vs)
        if (lf_needs2(p, hs, t2, ithresh)) { if (lf_hev(p, hs, hev)) lf2(p, hs); else if (mb_edge) lf6(p, hs); else lf4(p, hs); }
}

// libwebp yuv.h: 14-bit fixed point with the rounding folded into the constants
__host__ __device__ CSW_INLINE static inline int yuv_clip(int v) { return (v & ~16383) == 0 ? (v >> 6) : (v < 0 ? 0 : 255); }
__host__ __device__ CSW_INLINE static inline void yuv_rgb(int y, int u, int v, uint8_t *o) {
    const int yy = (y * 19077) >> 8;
    o[0] = uint8_t(yuv_clip(yy + ((v * 26149) >> 8) - 14234));
    o[1] = uint8_t(yuv_clip(yy - ((u * 6419) >> 8) - ((v * 13320) >> 8) + 8708));
    o[2] = uint8_t(yuv_clip(yy + ((u * 33050) >> 8) - 17685));
}

struct Vp8Seg { int y1[2], y2[2], uv[2]; };
struct Vp8FInfo { uint8_t limit, ilevel, inner, hev; };
struct Vp8Frame { uint32_t filtering, simple; };   // what the stages behind the parse need of the frame header (kept behind the contexts in the work area)
struct Vp8MbRec { uint32_t nzmask; uint8_t i4, ymode, uvmode, pad; uint8_t bmodes[16]; };   // a macroblock as the parse leaves it (nzmask: blocks with coefficients, 0..15 y, 16..19 u, 20..23 v)

// the work area of one picture (vp8_work_bytes)
struct Vp8Planes {
    uint8_t *Y, *U, *V;
    Vp8FInfo *finfo;
    uint8_t *ctx;        // per macroblock column: [0..3] sub-block modes above, [4..12] non-zero flags above (4 y, 2 u, 2 v, 1 y2)
    Vp8Frame *frame;
    Vp8MbRec *rec;       // [mbw * mbh]
    int16_t *mbcoef;     // [mbw * mbh][24][16]
    int ys, cs;
    uint32_t mbw, mbh;
};
__host__ __device__ CSW_INLINE static inline Vp8Planes vp8_planes(uint8_t *work, uint32_t W, uint32_t H) {
    Vp8Planes p;
    p.mbw = (W + 15) >> 4; p.mbh = (H + 15) >> 4;
    p.ys = int(p.mbw * 16); p.cs = int(p.mbw * 8);
    p.Y = work; p.U = p.Y + size_t(p.ys) * p.mbh * 16; p.V = p.U + size_t(p.cs) * p.mbh * 8;
    p.finfo = reinterpret_cast<Vp8FInfo *>(p.V + size_t(p.cs) * p.mbh * 8);
    p.ctx = reinterpret_cast<uint8_t *>(p.finfo + size_t(p.mbw) * p.mbh);
    p.frame = reinterpret_cast<Vp8Frame *>(p.ctx + size_t(p.mbw) * 16);
    p.rec = reinterpret_cast<Vp8MbRec *>(p.ctx + size_t(p.mbw) * 16 + 64);
    // 16-byte blocks: the offset rounded up (the work area itself starts at a multiple of 64 bytes; the + 256 of vp8_work_bytes pays for the gaps)
    const size_t rec_end = size_t(reinterpret_cast<uint8_t *>(p.rec + size_t(p.mbw) * p.mbh) - work);
    p.mbcoef = reinterpret_cast<int16_t *>(work + ((rec_end + 15) & ~size_t(15)));
    return p;
}

// The parse's working set.  One lane walks a frame (a boolean-coded partition is one chain), and everything it touches per symbol -- the
// probabilities, the small constant tables, the macroblock's coefficients, the contexts -- would sit in private memory (scratch: a
// round trip to the memory system per look-up) if it were local arrays; the kernel keeps this struct in LDS.
struct Vp8Hot {
    alignas(16) int16_t coef[25 * 16];      // zero between macroblocks
    uint8_t probs[4 * 8 * 3 * 11];          // [4 types][8 bands][3 contexts][11 nodes]
    uint8_t bmode[10 * 10 * 9];
    uint8_t bands[17], zigzag[16], cat[4][12];
    uint8_t bmodes[16], ctx[16];
    uint8_t left_modes[4], lnz[9];          // to the left: sub-block modes; non-zero flags (4 y, 2 u, 2 v, y2)
    int16_t dc[16];                         // the Y2 block of an i16 macroblock
};

// the parse: frame header, then macroblock by macroblock the modes and the coefficients (dequantised, the Y2 block folded into the luma DCs); leaves the
// macroblock records, their coefficient blocks, the filter strengths and the frame info in the work area.  Returns 0 or an error code (CS_ERR_* numbers are the caller's: 1 = malformed,
// 2 = unsupported feature)
__host__ __device__ CSW_INLINE static inline int vp8_parse_frame(const uint8_t *data, size_t n, uint32_t W, uint32_t H, uint8_t *work, Vp8Hot &hot, uint32_t debug = 0) {
    (void)debug;
    if (n < 10) return 1;
    const uint32_t tag = data[0] | (data[1] << 8) | (data[2] << 16);
    if (tag & 1) return 2;                                   // not a key frame
    const uint32_t part0_len = tag >> 5;
    if (data[3] != 0x9D || data[4] != 0x01 || data[5] != 0x2A) return 1;
    const uint32_t fw = (data[6] | (data[7] << 8)) & 0x3FFF, fh = (data[8] | (data[9] << 8)) & 0x3FFF;
    if (fw != W || fh != H || !W || !H) return 1;
    if (10 + size_t(part0_len) > n) return 1;
    const Vp8Planes pl = vp8_planes(work, W, H);
    const uint32_t mbw = pl.mbw, mbh = pl.mbh;
    Vp8FInfo *finfo = pl.finfo;
    uint8_t *ctx = pl.ctx;
    BoolDec br; br.init(data + 10, part0_len);
    br.get(128); br.get(128);                                // colour space, clamping type: no effect on decoding
    // segments
    bool use_seg = br.get(128) != 0, update_map = false, seg_abs = true;
    int seg_q[4] = {0, 0, 0, 0}, seg_lf[4] = {0, 0, 0, 0}, seg_prob[3] = {255, 255, 255};
    if (use_seg) {
        update_map = br.get(128) != 0;
        if (br.get(128)) {
            seg_abs = br.get(128) != 0;
            for (int i = 0; i < 4; i++) seg_q[i] = br.get(128) ? br.slit(7) : 0;
            for (int i = 0; i < 4; i++) seg_lf[i] = br.get(128) ? br.slit(6) : 0;
        }
        if (update_map) for (int i = 0; i < 3; i++) seg_prob[i] = br.get(128) ? int(br.lit(8)) : 255;
    }
    // filter
    const bool simple = br.get(128) != 0;
    const int level = int(br.lit(6)), sharp = int(br.lit(3));
    const bool use_delta = br.get(128) != 0;
    int ref_delta[4] = {0, 0, 0, 0}, mode_delta[4] = {0, 0, 0, 0};
    if (use_delta && br.get(128)) {
        for (int i = 0; i < 4; i++) if (br.get(128)) ref_delta[i] = br.slit(6);
        for (int i = 0; i < 4; i++) if (br.get(128)) mode_delta[i] = br.slit(6);
    }
    // token partitions
    const int nparts = 1 << br.lit(2);
    const uint8_t *psz = data + 10 + part0_len;
    if (size_t(psz - data) + 3 * size_t(nparts - 1) > n) return 1;
    const uint8_t *pstart = psz + 3 * (nparts - 1);
    BoolDec tok[8];
    {
        const uint8_t *q = pstart;
        for (int i = 0; i < nparts; i++) {
            size_t len = i + 1 < nparts ? size_t(psz[3 * i] | (psz[3 * i + 1] << 8) | (psz[3 * i + 2] << 16)) : size_t(data + n - q);
            if (q > data + n) return 1;
            if (len > size_t(data + n - q)) len = size_t(data + n - q);
            tok[i].init(q, len);
            q += len;
        }
    }
    // quantisers
    const int base_q = int(br.lit(7));
    int dq[5];
    for (int i = 0; i < 5; i++) dq[i] = br.get(128) ? br.slit(4) : 0;   // y1 dc, y2 dc, y2 ac, uv dc, uv ac
    Vp8Seg seg[4];
    for (int i = 0; i < 4; i++) {
        int q = base_q;
        if (use_seg) q = seg_abs ? seg_q[i] : q + seg_q[i];
        seg[i].y1[0] = kVp8DcQ[clipq(q + dq[0], 127)]; seg[i].y1[1] = kVp8AcQ[clipq(q, 127)];
        seg[i].y2[0] = kVp8DcQ[clipq(q + dq[1], 127)] * 2; seg[i].y2[1] = (kVp8AcQ[clipq(q + dq[2], 127)] * 101581) >> 16;
        if (seg[i].y2[1] < 8) seg[i].y2[1] = 8;
        seg[i].uv[0] = kVp8DcQ[clipq(q + dq[3], 117)]; seg[i].uv[1] = kVp8AcQ[clipq(q + dq[4], 127)];
    }
    br.get(128);                                             // refresh_entropy_probs: a single frame
    // coefficient probabilities, and the constant tables of the hot loop next to them
    CSW_NOUNROLL
    for (int i = 0; i < 4 * 8 * 3 * 11; i++) hot.probs[i] = br.get(kVp8CoefUpdateProbs[i]) ? uint8_t(br.lit(8)) : kVp8CoefProbs[i];
    CSW_NOUNROLL
    for (int i = 0; i < 10 * 10 * 9; i++) hot.bmode[i] = kVp8BModeProbs[i];
    for (int i = 0; i < 17; i++) hot.bands[i] = kVp8Bands[i];
    for (int i = 0; i < 16; i++) hot.zigzag[i] = kVp8Zigzag[i];
    for (int i = 0; i < 12; i++) { hot.cat[0][i] = i < 4 ? kVp8Cat3[i] : 0; hot.cat[1][i] = i < 5 ? kVp8Cat4[i] : 0; hot.cat[2][i] = i < 6 ? kVp8Cat5[i] : 0; hot.cat[3][i] = kVp8Cat6[i]; }
    CSW_NOUNROLL
    for (int k = 0; k < 25 * 16; k++) hot.coef[k] = 0;
    const bool use_skip = br.get(128) != 0;
    const int skip_p = use_skip ? int(br.lit(8)) : 0;
    // filter strengths per segment and block type
    Vp8FInfo fstr[4][2];
    for (int s = 0; s < 4; s++)
        for (int i4 = 0; i4 < 2; i4++) {
            int base = level;
            if (use_seg) base = seg_abs ? seg_lf[s] : base + seg_lf[s];
            int lv = base;
            if (use_delta) { lv += ref_delta[0]; if (i4) lv += mode_delta[0]; }
            lv = clipq(lv, 63);
            Vp8FInfo f = {0, 0, uint8_t(i4), 0};
            if (lv > 0) {
                int il = lv;
                if (sharp > 0) { il >>= (sharp > 4) ? 2 : 1; if (il > 9 - sharp) il = 9 - sharp; }
                if (il < 1) il = 1;
                f.ilevel = uint8_t(il); f.limit = uint8_t(2 * lv + il); f.hev = uint8_t(lv >= 40 ? 2 : lv >= 15 ? 1 : 0);
            }
            fstr[s][i4] = f;
        }
    const bool filtering = level != 0;   // libwebp: a frame-level 0 switches the filter off whatever the segments say
    pl.frame->filtering = filtering ? 1u : 0u; pl.frame->simple = simple ? 1u : 0u;

    for (uint32_t i = 0; i < mbw * 16; i++) ctx[i] = 0;
    int16_t *const coef = hot.coef;
    // ---- macroblocks
    for (uint32_t my = 0; my < mbh; my++) {
        BoolDec tb = tok[my & uint32_t(nparts - 1)];   // the row's token reader in registers; handed back at the end of the row
        uint8_t *const left_modes = hot.left_modes, *const lnz = hot.lnz;
        for (int k = 0; k < 4; k++) left_modes[k] = 0;
        for (int k = 0; k < 9; k++) lnz[k] = 0;
        for (uint32_t mx = 0; mx < mbw; mx++) {
            uint8_t *top_modes = hot.ctx, *tnz = top_modes + 4;
            for (int k = 0; k < 16; k++) hot.ctx[k] = ctx[size_t(mx) * 16 + k];
            // modes (first partition)
            int segment = 0;
            if (update_map) segment = !br.get(seg_prob[0]) ? br.get(seg_prob[1]) : br.get(seg_prob[2]) + 2;
            const bool skip_flag = use_skip ? br.get(skip_p) != 0 : false;
            const bool i4 = !br.get(145);
            uint8_t *bmodes = hot.bmodes;
            int ymode = 0;
            if (!i4) {
                ymode = br.get(156) ? (br.get(128) ? 1 : 3) : (br.get(163) ? 2 : 0);   // B_ order: 0 DC, 1 TM, 2 V, 3 H
                for (int k = 0; k < 4; k++) { top_modes[k] = uint8_t(ymode); left_modes[k] = uint8_t(ymode); }
            } else {
                CSW_NOUNROLL
                for (int y = 0; y < 4; y++) {
                    int lm = left_modes[y];
                    CSW_NOUNROLL
                    for (int x = 0; x < 4; x++) {
                        const uint8_t *pr = hot.bmode + (size_t(top_modes[x]) * 10 + size_t(lm)) * 9;
                        int m;
                        if (!br.get(pr[0])) m = 0;
                        else if (!br.get(pr[1])) m = 1;
                        else if (!br.get(pr[2])) m = 2;
                        else if (!br.get(pr[3])) { if (!br.get(pr[4])) m = 3; else m = !br.get(pr[5]) ? 4 : 5; }
                        else if (!br.get(pr[6])) m = 6;
                        else if (!br.get(pr[7])) m = 7;
                        else m = !br.get(pr[8]) ? 8 : 9;
                        bmodes[4 * y + x] = uint8_t(m); top_modes[x] = uint8_t(m); lm = m;
                    }
                    left_modes[y] = uint8_t(lm);
                }
            }
            const int uvmode = !br.get(142) ? 0 : !br.get(114) ? 2 : br.get(183) ? 1 : 3;
            // residuals (token partition of this macroblock row)
            uint32_t nzmask = 0;   // blocks with coefficients: bit b (0..15 y, 16..19 u, 20..23 v)
            const Vp8Seg &sq = seg[segment];
            auto get_coeffs = [&](int type, int ctx0, const int *q2, int first, int16_t *out) -> int {
                const int q_dc = q2[0], q_ac = q2[1];
                const uint8_t *bp = hot.probs + size_t(type) * 8 * 33;
                int nn = first;
                const uint8_t *p = bp + size_t(hot.bands[nn]) * 33 + size_t(ctx0) * 11;
                for (; nn < 16; nn++) {
                    if (!tb.get(p[0])) return nn;
                    while (!tb.get(p[1])) { p = bp + size_t(hot.bands[++nn]) * 33; if (nn == 16) return 16; }
                    const uint8_t *pn = bp + size_t(hot.bands[nn + 1]) * 33;
                    int v;
                    if (!tb.get(p[2])) { v = 1; p = pn + 11; }
                    else {
                        if (!tb.get(p[3])) { v = !tb.get(p[4]) ? 2 : 3 + tb.get(p[5]); }
                        else if (!tb.get(p[6])) {
                            if (!tb.get(p[7])) v = 5 + tb.get(159);
                            else { v = 7 + 2 * tb.get(165); v += tb.get(145); }
                        } else {
                            const int b1 = tb.get(p[8]), b0 = tb.get(p[9 + b1]), cat = 2 * b1 + b0;
                            const uint8_t *tab = hot.cat[cat];
                            v = 0;
                            for (; *tab; ++tab) v += v + tb.get(*tab);
                            v += 3 + (8 << cat);
                        }
                        p = pn + 22;
                    }
                    out[hot.zigzag[nn]] = int16_t((tb.get(128) ? -v : v) * (nn > 0 ? q_ac : q_dc));
                }
                return 16;
            };
            if (!skip_flag) {
                int first = 0, ytype = 3;
                if (!i4) {
                    int16_t *const dc = hot.dc;
                    for (int k = 0; k < 16; k++) dc[k] = 0;
                    const int nz = get_coeffs(1, tnz[8] + lnz[8], sq.y2, 0, dc);
                    tnz[8] = lnz[8] = uint8_t(nz > 0);
                    vp8_iwht(dc, coef);
                    for (int k = 0; k < 16; k++) if (coef[16 * k]) nzmask |= 1u << k;
                    first = 1; ytype = 0;
                }
                CSW_NOUNROLL
                for (int y = 0; y < 4; y++)
                    CSW_NOUNROLL
                    for (int x = 0; x < 4; x++) {
                        const int nz = get_coeffs(ytype, tnz[x] + lnz[y], sq.y1, first, coef + (4 * y + x) * 16);
                        tnz[x] = lnz[y] = uint8_t(nz > first);
                        if (nz > first) nzmask |= 1u << (4 * y + x);
                    }
                CSW_NOUNROLL
                for (int ch = 0; ch < 2; ch++)
                    CSW_NOUNROLL
                    for (int y = 0; y < 2; y++)
                        CSW_NOUNROLL
                        for (int x = 0; x < 2; x++) {
                            const int nz = get_coeffs(2, tnz[4 + 2 * ch + x] + lnz[4 + 2 * ch + y], sq.uv, 0, coef + (16 + 4 * ch + 2 * y + x) * 16);
                            tnz[4 + 2 * ch + x] = lnz[4 + 2 * ch + y] = uint8_t(nz > 0);
                            if (nz > 0) nzmask |= 1u << (16 + 4 * ch + 2 * y + x);
                        }
            } else {
                for (int k = 0; k < 8; k++) { tnz[k] = 0; lnz[k] = 0; }
                if (!i4) { tnz[8] = 0; lnz[8] = 0; }
            }
            for (int k = 0; k < 16; k++) ctx[size_t(mx) * 16 + k] = hot.ctx[k];
            // libwebp: f_inner |= !skip with skip = !(non_zero_y | non_zero_uv), the bits AFTER the inverse WHT -- a Y2 block whose transform comes out all zero does not count
            if (filtering) { Vp8FInfo f = fstr[segment][i4 ? 1 : 0]; f.inner |= uint8_t((skip_flag || nzmask == 0) ? 0 : 1); finfo[size_t(my) * mbw + mx] = f; }
            // what the reconstruction needs of this macroblock: its record, and the blocks that hold something (two 16-byte stores each)
            {
                Vp8MbRec r;
                r.nzmask = nzmask; r.i4 = i4 ? 1 : 0; r.ymode = uint8_t(ymode); r.uvmode = uint8_t(uvmode); r.pad = 0;
                for (int k = 0; k < 16; k++) r.bmodes[k] = bmodes[k];
                const size_t mb = size_t(my) * mbw + mx;
                pl.rec[mb] = r;
                uint4 *dst = reinterpret_cast<uint4 *>(pl.mbcoef + mb * 384);
                const uint4 *src = reinterpret_cast<const uint4 *>(coef);
                CSW_NOUNROLL
                for (int k = 0; k < 24; k++) if (nzmask & (1u << k)) { dst[2 * k] = src[2 * k]; dst[2 * k + 1] = src[2 * k + 1]; }
            }
            // the coefficient store goes back to all zeros: only blocks in nzmask hold anything
            CSW_NOUNROLL
            for (int k = 0; k < 24; k++) if (nzmask & (1u << k)) for (int q = 0; q < 16; q++) coef[16 * k + q] = 0;
        }
        tok[my & uint32_t(nparts - 1)] = tb;
        if (br.eof()) return 1;
    }
    return 0;
}

// the reconstruction of one macroblock from its record: intra prediction from the UNFILTERED neighbours (the loop filter runs over the finished frame
// afterwards) + the inverse transforms of the blocks in nzmask.  Needs (mx - 1, my), (mx, my - 1), (mx - 1, my - 1) and (mx + 1, my - 1) done: the same wave
// front as the loop filter.  `sc` is scratch for one macroblock (LDS in the kernel: the planes have no margin, so the block is predicted in a 17 x 32 patch
// whose row above and column to the left get the neighbours -- frame edges: 127 above, 129 to the left -- and sits word-aligned at row 1, column 4).
struct Vp8Scratch { alignas(4) uint8_t s[(16 + 1) * 32]; };
__host__ __device__ CSW_INLINE static inline void vp8_recon_mb(uint8_t *work, uint32_t W, uint32_t H, uint32_t mx, uint32_t my, Vp8Scratch &sc) {
    const Vp8Planes pl = vp8_planes(work, W, H);
    const uint32_t mbw = pl.mbw;
    const int ys = pl.ys, cs = pl.cs;
    const size_t mb = size_t(my) * mbw + mx;
    const Vp8MbRec r = pl.rec[mb];
    const int16_t *coef = pl.mbcoef + mb * 384;
    const uint32_t nzmask = r.nzmask;
    const int S = 32;
    uint8_t *const py = sc.s + S + 4;
    uint8_t *yd = pl.Y + size_t(my) * 16 * ys + size_t(mx) * 16;
    for (int x = -1; x < 20; x++) {
        int v = 127;
        if (my > 0) {
            if (x < 0) v = mx > 0 ? yd[-ys - 1] : 129;
            else if (x < 16) v = yd[-ys + x];
            else v = mx + 1 < mbw ? yd[-ys + x] : yd[-ys + 15];
        }
        py[-S + x] = uint8_t(v);
    }
    for (int y = 0; y < 16; y++) py[y * S - 1] = mx > 0 ? yd[y * ys - 1] : uint8_t(129);
    if (!r.i4) {
        switch (r.ymode) {
        case 0: pred_dc(py, S, 16, my > 0, mx > 0); break;
        case 1: pred_tm(py, S, 16); break;
        case 2: pred_v(py, S, 16); break;
        default: pred_h(py, S, 16); break;
        }
        for (int k = 0; k < 16; k++) if (nzmask & (1u << k)) vp8_idct_add(coef + 16 * k, py + (k >> 2) * 4 * S + (k & 3) * 4, S);
    } else {
        for (int k = 0; k < 16; k++) {
            uint8_t *d = py + (k >> 2) * 4 * S + (k & 3) * 4;
            uint8_t tr[4];
            if ((k & 3) == 3) { for (int q = 0; q < 4; q++) tr[q] = py[-S + 16 + q]; }    // right column: the four samples above and right of the MACROBLOCK, whatever the row
            else for (int q = 0; q < 4; q++) tr[q] = d[-S + 4 + q];
            pred4(d, S, r.bmodes[k], tr);
            if (nzmask & (1u << k)) vp8_idct_add(coef + 16 * k, d, S);
        }
    }
    // whole words out (the block sits word-aligned in the scratch, the planes are 16 / 8 samples per macroblock wide)
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x += 4) *reinterpret_cast<uint32_t *>(yd + y * ys + x) = *reinterpret_cast<const uint32_t *>(py + y * S + x);
    for (int ch = 0; ch < 2; ch++) {   // the chroma planes, one after the other in the same scratch
        uint8_t *cd = (ch ? pl.V : pl.U) + size_t(my) * 8 * cs + size_t(mx) * 8;
        uint8_t *pc = py;
        for (int x = -1; x < 8; x++) {
            int a = 127;
            if (my > 0) a = x < 0 ? (mx > 0 ? cd[-cs - 1] : 129) : cd[-cs + x];
            pc[-S + x] = uint8_t(a);
        }
        for (int y = 0; y < 8; y++) pc[y * S - 1] = mx > 0 ? cd[y * cs - 1] : uint8_t(129);
        switch (r.uvmode) {
        case 0: pred_dc(pc, S, 8, my > 0, mx > 0); break;
        case 1: pred_tm(pc, S, 8); break;
        case 2: pred_v(pc, S, 8); break;
        default: pred_h(pc, S, 8); break;
        }
        for (int k = 0; k < 4; k++) if (nzmask & (1u << (16 + 4 * ch + k))) vp8_idct_add(coef + (16 + 4 * ch + k) * 16, pc + (k >> 1) * 4 * S + (k & 1) * 4, S);
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x += 4) *reinterpret_cast<uint32_t *>(cd + y * cs + x) = *reinterpret_cast<const uint32_t *>(pc + y * S + x);
    }
}

// the loop filter of one macroblock (RFC 6386 section 15: macroblocks in raster order).  What a macroblock's filter reads and changes reaches three
// samples into the macroblocks to its left and above, so (mx, my) only needs (mx - 1, my), (mx, my - 1) and (mx + 1, my - 1) done: the kernel runs the
// frame as a wave front, macroblock row r at column t - 2 r in step t.
__host__ __device__ CSW_INLINE static inline void vp8_filter_mb(uint8_t *work, uint32_t W, uint32_t H, uint32_t mx, uint32_t my) {
    const Vp8Planes pl = vp8_planes(work, W, H);
    if (!pl.frame->filtering) return;
    const bool simple = pl.frame->simple != 0;
    const int ys = pl.ys, cs = pl.cs;
    const Vp8FInfo f = pl.finfo[size_t(my) * pl.mbw + mx];
    if (!f.limit) return;
    uint8_t *yd = pl.Y + size_t(my) * 16 * ys + size_t(mx) * 16, *ud = pl.U + size_t(my) * 8 * cs + size_t(mx) * 8, *vd = pl.V + size_t(my) * 8 * cs + size_t(mx) * 8;
    if (simple) {
        if (mx > 0) lf_simple(yd, 1, ys, 16, f.limit + 4);
        if (f.inner) for (int k = 4; k < 16; k += 4) lf_simple(yd + k, 1, ys, 16, f.limit);
        if (my > 0) lf_simple(yd, ys, 1, 16, f.limit + 4);
        if (f.inner) for (int k = 4; k < 16; k += 4) lf_simple(yd + k * ys, ys, 1, 16, f.limit);
    } else {
        if (mx > 0) { lf_edge(yd, 1, ys, 16, f.limit + 4, f.ilevel, f.hev, true); lf_edge(ud, 1, cs, 8, f.limit + 4, f.ilevel, f.hev, true); lf_edge(vd, 1, cs, 8, f.limit + 4, f.ilevel, f.hev, true); }
        if (f.inner) {
            for (int k = 4; k < 16; k += 4) lf_edge(yd + k, 1, ys, 16, f.limit, f.ilevel, f.hev, false);
            lf_edge(ud + 4, 1, cs, 8, f.limit, f.ilevel, f.hev, false); lf_edge(vd + 4, 1, cs, 8, f.limit, f.ilevel, f.hev, false);
        }
        if (my > 0) { lf_edge(yd, ys, 1, 16, f.limit + 4, f.ilevel, f.hev, true); lf_edge(ud, cs, 1, 8, f.limit + 4, f.ilevel, f.hev, true); lf_edge(vd, cs, 1, 8, f.limit + 4, f.ilevel, f.hev, true); }
        if (f.inner) {
            for (int k = 4; k < 16; k += 4) lf_edge(yd + k * ys, ys, 1, 16, f.limit, f.ilevel, f.hev, false);
            lf_edge(ud + 4 * cs, cs, 1, 8, f.limit, f.ilevel, f.hev, false); lf_edge(vd + 4 * cs, cs, 1, 8, f.limit, f.ilevel, f.hev, false);
        }
    }
}

// libwebp's fancy upsampler (chroma at 9:3:3:1 of the four nearest samples, computed on u | v << 16 pairs) + YCbCr -> RGB, one pair of luma rows
// between two chroma rows per call: k = 0 the first row (its chroma row on both sides), k = 1 .. chh - 1 rows 2k-1 and 2k between chroma rows k-1 and k,
// k = chh the last row of an even height on its own.  The calls are independent of one another.
__host__ __device__ CSW_INLINE static inline void vp8_rgb_rows(uint8_t *work, uint32_t W, uint32_t H, uint32_t k, uint8_t *rgb) {
    const Vp8Planes pl = vp8_planes(work, W, H);
    const uint8_t *Y = pl.Y, *U = pl.U, *V = pl.V;
    const int ys = pl.ys, cs = pl.cs;
    const int chh = int((H + 1) >> 1);
    int ytop, ybot, ctop, ccur;
    if (k == 0) { ytop = 0; ybot = -1; ctop = 0; ccur = 0; }
    else if (int(k) < chh) { ytop = 2 * int(k) - 1; ybot = 2 * int(k); ctop = int(k) - 1; ccur = int(k); }
    else if (int(k) == chh && !(H & 1)) { ytop = int(H) - 1; ybot = -1; ctop = chh - 1; ccur = chh - 1; }
    else return;
    auto uvp = [&](int cx, int cy) -> uint32_t { return uint32_t(U[size_t(cy) * cs + cx]) | (uint32_t(V[size_t(cy) * cs + cx]) << 16); };
    auto emit = [&](int yrow, int x, uint32_t uv) { yuv_rgb(Y[size_t(yrow) * ys + x], int(uv & 0xFF), int(uv >> 16), rgb + (size_t(yrow) * W + x) * 3); };
    const int last_pair = (int(W) - 1) >> 1;
    uint32_t tl = uvp(0, ctop), l = uvp(0, ccur);
    if (ytop >= 0) emit(ytop, 0, (3 * tl + l + 0x00020002u) >> 2);
    if (ybot >= 0) emit(ybot, 0, (3 * l + tl + 0x00020002u) >> 2);
    for (int x = 1; x <= last_pair; x++) {
        const uint32_t t = uvp(x, ctop), c = uvp(x, ccur);
        const uint32_t avg = tl + t + l + c + 0x00080008u;
        const uint32_t d12 = (avg + 2 * (t + l)) >> 3, d03 = (avg + 2 * (tl + c)) >> 3;
        if (ytop >= 0) { emit(ytop, 2 * x - 1, (d12 + tl) >> 1); emit(ytop, 2 * x, (d03 + t) >> 1); }
        if (ybot >= 0) { emit(ybot, 2 * x - 1, (d03 + l) >> 1); emit(ybot, 2 * x, (d12 + c) >> 1); }
        tl = t; l = c;
    }
    if (!(W & 1)) {
        if (ytop >= 0) emit(ytop, int(W) - 1, (3 * tl + l + 0x00020002u) >> 2);
        if (ybot >= 0) emit(ybot, int(W) - 1, (3 * l + tl + 0x00020002u) >> 2);
    }
}

}  // namespace csw
