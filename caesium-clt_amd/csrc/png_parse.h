// png_parse.h -- the min-cost-path parse of a deflate chunk, one wave per chunk (statement: oracle/png_oracle.c deep_parse()).
// What libdeflate's levels 10-12 (oxipng -o3 / -o4) and zopfli do in their own ways -- candidate matches per position, symbol costs from the
// previous parse's statistics, the cheapest path, again -- laid out for a 64-lane wave:
//   M  candidates, tile by tile as png_lz.h does it (here a tile = 256 consecutive positions, four per lane): the six fixed distances, the four entries of the
//      4-byte-hash table, the eight of a second table keyed by 8 bytes (the long matches far back).  Kept per position, in ONE 64-bit word in the wave's
//      scratch area: c0 = the nearest candidate with >= 3 bytes, c1 = the longest, and the byte itself.
//   C  costs (1/16 bit) of every literal, length and distance code from the counts of the parse before.
//   D  the path: the chunk is 64 SEGMENTS of 512 bytes, one per lane; every lane walks its segment backwards with the costs of the next sixteen
//      positions in registers (lengths above sixteen are tried only at a candidate's full length and look their cost up in the scratch area).  No lane
//      waits for another; the scratch arrays are laid out [position in segment][lane], so a step's loads and stores are one line per array.
//   F  every lane follows its segment's choices forwards and counts the symbols (LDS atomics).
// C-D-F repeat `iters` times; the caller's sink then sees the tokens tile by tile, as from lz_chunk.
#pragma once
#include "png_lz.h"

namespace csp {

enum : uint32_t {
    CSP_DEEP_DIV = 512,        // a chunk qualifies when matches * DIV >= tokens in its greedy parse
    CSP_HASH4_BITS = 11,       // the parse's own 4-byte table: 2048 buckets x 4 (the greedy tokenizer's has 512)
    CSP_HASH8_BITS = 11, CSP_WAYS8 = 8,
    CSP_DEEP_TILE = 256,       // positions whose candidates come out of one state of the tables (four per lane)
    CSP_DEEP_SEG = 512, CSP_DEEP_CAP = 16, CSP_DEEP_START = 64,
    CSP_DEEP_ITERS = 5, CSP_DEEP_ITERS_ZOPFLI = 15,
    CSP_DEEP_LIVE_NUM = 9, CSP_DEEP_LIVE_DEN = 8,   // a trial takes the parse when its greedy stream is within NUM / DEN of the picture's smallest
    // the wave's scratch area in HBM: candidates (8 B), choices (2 B), costs (4 B, one row more), visit marks (1 B) per position
    CSP_DEEP_CAND_OFF = 0, CSP_DEEP_CHOICE_OFF = 262144, CSP_DEEP_COST_OFF = 327680, CSP_DEEP_TAKEN_OFF = 327680 + 131584, CSP_DEEP_SCRATCH = 524288,
};
static_assert(uint32_t(CSP_HASH8_BITS) == uint32_t(CSP_HASH4_BITS), "one last-lane map serves both tables");
static_assert(CSP_DEEP_TAKEN_OFF + 32768 <= CSP_DEEP_SCRATCH, "scratch layout");

struct DeepLds {
    uint64_t bucket[1u << CSP_HASH4_BITS];        // 4-byte table: four 16-bit positions, most recent in the low bits (png_lz.h)
    uint64_t bucket8[2][1u << CSP_HASH8_BITS];   // 8-byte table: ways 0..3 in [0], 4..7 in [1]
    uint8_t lastlane[1u << CSP_HASH4_BITS];
    uint32_t hist[CSP_NSYM];                     // (its first 256 words serve deep_last_lanes while the candidates are made: the counts start after that)
    uint32_t item;                               // the workgroup's current work item (next_item)
    uint16_t lit_cost[256], len_cost[260], dist_cost[32];
};

__device__ __forceinline__ static uint32_t lz_hash4(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - CSP_HASH4_BITS); }
__device__ __forceinline__ static uint32_t lz_hash8(uint64_t v) { return ((uint32_t(v) * 0x9E3779B1u) ^ (uint32_t(v >> 32) * 0x85EBCA6Bu)) >> (32 - CSP_HASH8_BITS); }
__device__ __forceinline__ static uint32_t cost16_of(uint32_t c, uint32_t total) {   // 16 log2(total / c), 1 .. 240 (oracle: cost16_of)
    const uint32_t q = (total << 8) / c;
    const uint32_t e = 31u - uint32_t(__clz(q));
    const uint32_t v = 16u * (e - 8u) + (((q << 4) >> e) & 15u);
    return v < 1u ? 1u : v > 240u ? 240u : v;
}
// The candidate search runs on a WORKGROUP of four waves that share the tables: wave u takes positions t0 + 64 u + lane of a 256-position tile, and the
// steps that read and write the tables are separated by workgroup barriers.  The emulation plays the four waves one after the other inside UFOR.
#ifdef CSH_EMUL
#define UFOR(u) for (int u = 0; u < 4; u++)
#define UIX(u) (u)
#define CSP_WG_SYNC() ((void)0)
#define CSP_WAVE0 true
enum { CSP_DEEP_THREADS = 1, CSP_UN = 4 };
#else
#define UFOR(u) for (int u [[maybe_unused]] = int(threadIdx.x >> 6), once_u_ = 1; once_u_; once_u_ = 0)
#define UIX(u) 0
#define CSP_WG_SYNC() __syncthreads()
#define CSP_WAVE0 (threadIdx.x < 64u)
enum { CSP_DEEP_THREADS = 256, CSP_UN = 1 };
#endif
// per-lane small arrays (registers on the device: every index is a constant after unrolling)
#ifdef CSH_EMUL
template <class T, int N> struct LVArr { T v[64][N]; __device__ T *operator[](int j) { return v[j]; } };
#else
template <class T, int N> struct LVArr { T v[N]; __device__ T *operator[](int) { return v; } };
#endif

// A TILE of the candidate search is 256 consecutive positions, four per lane (position t0 + 64 u + lane): the four are independent work -- their loads are in
// flight together, which is what a wave that has its SIMD to itself (the tables leave room for three waves per CU) needs instead of neighbours.
// Which position of the tile is the last with each hash: a byte per hash names SOME position that has it (whichever write the LDS kept), and the positions that
// share it take the maximum of their numbers in that position's slot -- two LDS round trips whatever the tile holds (a retry loop until the highest write has
// stuck took one round per position in a run of equal bytes)
typedef LVArr<uint32_t, CSP_UN> LV4;
__device__ __forceinline__ static void deep_last_lanes(DeepLds &S, LV4 &hash, LV4 &hashable, LV4 &is_last) {
    LFOR(l) {
        UFOR(u) { S.hist[u * 64 + l] = 0; if (hashable[l][UIX(u)]) S.lastlane[hash[l][UIX(u)]] = uint8_t(u * 64 + l); }
    }
    CSP_WG_SYNC();
    LV4 rep;
    LFOR(l) {
        UFOR(u) { rep[l][UIX(u)] = hashable[l][UIX(u)] ? uint32_t(S.lastlane[hash[l][UIX(u)]]) : 0u; if (hashable[l][UIX(u)]) atomicMax(&S.hist[rep[l][UIX(u)]], uint32_t(u * 64 + l)); }
    }
    CSP_WG_SYNC();
    LFOR(l) {
        UFOR(u) is_last[l][UIX(u)] = hashable[l][UIX(u)] && S.hist[rep[l][UIX(u)]] == uint32_t(u * 64 + l) ? 1u : 0u;
    }
    CSP_WG_SYNC();
}
__device__ __forceinline__ static void deep_insert(DeepLds &S, LV4 &h4, LV4 &ok4, LV4 &h8, LV4 &ok8, LV4 &rel) {
    LV4 last;
    deep_last_lanes(S, h4, ok4, last);
    LFOR(l) {
        UFOR(u) if (last[l][UIX(u)] && rel[l][UIX(u)] != 0xFFFFu) S.bucket[h4[l][UIX(u)]] = (S.bucket[h4[l][UIX(u)]] << 16) | rel[l][UIX(u)];
    }
    CSP_WG_SYNC();
    deep_last_lanes(S, h8, ok8, last);
    LFOR(l) {
        UFOR(u) if (last[l][UIX(u)] && rel[l][UIX(u)] != 0xFFFFu) {
            const uint64_t b0 = S.bucket8[0][h8[l][UIX(u)]], b1 = S.bucket8[1][h8[l][UIX(u)]];
            S.bucket8[0][h8[l][UIX(u)]] = (b0 << 16) | rel[l][UIX(u)];
            S.bucket8[1][h8[l][UIX(u)]] = (b1 << 16) | (b0 >> 48);
        }
    }
    CSP_WG_SYNC();
}

// what a block with these counts takes, in 1/16 bit, by the entropy of its two alphabets (+ the extra bits): the parse replaces the greedy one only if this
// says it is smaller (oracle: deep_estimate).  freq: CSP_NSYM counts, end-of-block included
template <class F>
__device__ __forceinline__ static uint64_t deep_estimate(F freq) {
    LV<uint64_t> a, b;
    LFOR(l) { a[l] = 0; b[l] = 0; for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) { if (i < CSP_NLIT) a[l] += freq(i); else b[l] += freq(i); } }
    const uint32_t tl = uint32_t(lsum(a)), td = uint32_t(lsum(b));
    LV<uint64_t> e;
    LFOR(l) {
        e[l] = 0;
        for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) {
            const uint32_t f = freq(i);
            if (!f) continue;
            e[l] += uint64_t(f) * (cost16_of(f, i < CSP_NLIT ? tl : td) + 16u * (i > 256 && i < CSP_NLIT ? len_extra_of(i - 257) : i >= CSP_NLIT ? dist_extra_of(i - CSP_NLIT) : 0u));
        }
    }
    return lsum(e);
}

struct NoSink { __device__ __forceinline__ void tile(uint64_t, uint32_t, uint64_t, const LV<uint32_t> &, const LV<uint32_t> &, const LV<uint32_t> &) {} };

// data[start, end): a chunk of a stream of `total` bytes.  S.hist holds the last pass's counts on exit.  want_tokens: the sink sees the final parse.
template <class Sink>
__device__ static void deep_chunk(const uint8_t *data, uint64_t total, uint64_t start, uint64_t end, DeepLds &S, uint8_t *scratch, int iters, bool want_tokens, Sink &sink) {
    unsigned long long *cand = reinterpret_cast<unsigned long long *>(scratch + CSP_DEEP_CAND_OFF);   // [i][lane]: len0 | d0 << 9 | len1 << 25 | d1 << 34 | byte << 50
    uint16_t *choice = reinterpret_cast<uint16_t *>(scratch + CSP_DEEP_CHOICE_OFF);
    uint32_t *costs = reinterpret_cast<uint32_t *>(scratch + CSP_DEEP_COST_OFF);
    uint8_t *tk = scratch + CSP_DEEP_TAKEN_OFF;
    const uint32_t n = uint32_t(end - start);
    // ---------------------------------------------------------------------------------------------------------------- M
    LFOR(l) for (uint32_t i = uint32_t(l); i < (1u << CSP_HASH4_BITS); i += 64) { S.bucket[i] = ~0ull; S.bucket8[0][i] = ~0ull; S.bucket8[1][i] = ~0ull; }
    CSP_WG_SYNC();
    {
        const uint64_t seed0 = start > 32768 ? start - 32768 : 0;
        for (uint64_t t0 = seed0; t0 < start; t0 += CSP_DEEP_TILE) {
            LV4 h4, ok4, h8, ok8, rel;
            LFOR(l) {
                UFOR(u) {
                    const uint64_t p = t0 + uint32_t(u * 64 + l);
                    const uint64_t v = load64u(data + p);
                    ok4[l][UIX(u)] = p + 4 <= total ? 1u : 0u; ok8[l][UIX(u)] = p + 8 <= total ? 1u : 0u;
                    h4[l][UIX(u)] = ok4[l][UIX(u)] ? lz_hash4(uint32_t(v)) : 0u; h8[l][UIX(u)] = ok8[l][UIX(u)] ? lz_hash8(v) : 0u;
                    rel[l][UIX(u)] = uint32_t(p + 32768 - start);
                }
            }
            deep_insert(S, h4, ok4, h8, ok8, rel);
        }
    }
    for (uint64_t t0 = start; t0 < end; t0 += CSP_DEEP_TILE) {
        LV4 h4, ok4, h8, ok8, rel;
        LFOR(l) {
            enum { NW = CSP_WAYS + CSP_WAYS8 };
            uint64_t hi[CSP_UN], lo[CSP_UN], xw[CSP_UN][NW];
            uint32_t dw[CSP_UN][NW], maxlen[CSP_UN];
            int nw[CSP_UN];
            bool in[CSP_UN];
            // stage 1: the positions' own bytes
            UFOR(u) {
                const uint64_t p = t0 + uint32_t(u * 64 + l);
                in[UIX(u)] = p < end;
                h4[l][UIX(u)] = 0; ok4[l][UIX(u)] = 0; h8[l][UIX(u)] = 0; ok8[l][UIX(u)] = 0; rel[l][UIX(u)] = uint32_t(p + 32768 - start);
                hi[UIX(u)] = in[UIX(u)] ? load64u(data + p) : 0ull;
                lo[UIX(u)] = in[UIX(u)] ? load64u(data + p - 8) : 0ull;   // (p < 8: whatever the pool holds there; guarded by d <= p)
                maxlen[UIX(u)] = in[UIX(u)] ? (end - p < 258 ? uint32_t(end - p) : 258u) : 0u;
            }
            // stage 2: both tables' entries, and their first eight bytes asked for -- all four positions' at once
            UFOR(u) {
                const uint64_t p = t0 + uint32_t(u * 64 + l);
                nw[UIX(u)] = 0;
                if (in[UIX(u)] && p + 4 <= total) {
                    ok4[l][UIX(u)] = 1; h4[l][UIX(u)] = lz_hash4(uint32_t(hi[UIX(u)]));
                    const uint64_t b = S.bucket[h4[l][UIX(u)]];
                    for (int w = 0; w < int(CSP_WAYS); w++) {
                        const uint32_t r = uint32_t(b >> (16 * w)) & 0xFFFFu;
                        if (r == 0xFFFFu) break;
                        const uint32_t d = rel[l][UIX(u)] - r;
                        if (d > 32768u) break;
                        dw[UIX(u)][nw[UIX(u)]++] = d;
                    }
                }
                if (in[UIX(u)] && p + 8 <= total) {
                    ok8[l][UIX(u)] = 1; h8[l][UIX(u)] = lz_hash8(hi[UIX(u)]);
                    const uint64_t b0 = S.bucket8[0][h8[l][UIX(u)]], b1 = S.bucket8[1][h8[l][UIX(u)]];
                    for (int w = 0; w < int(CSP_WAYS8); w++) {
                        const uint32_t r = uint32_t((w < 4 ? b0 : b1) >> (16 * (w & 3))) & 0xFFFFu;
                        if (r == 0xFFFFu) break;
                        const uint32_t d = rel[l][UIX(u)] - r;
                        if (d > 32768u) break;
                        dw[UIX(u)][nw[UIX(u)]++] = d;
                    }
                }
            }
            UFOR(u) {
                const uint64_t p = t0 + uint32_t(u * 64 + l);
                CSH_UNROLL
                for (int w = 0; w < NW; w++) xw[UIX(u)][w] = w < nw[UIX(u)] ? hi[UIX(u)] ^ load64u(data + p - dw[UIX(u)][w]) : 0ull;
            }
            // stage 3: judge them.  c0: the nearest with >= 3 bytes; c1: the longest (ties: the nearer) -- whatever the order of the offers
            UFOR(u) if (in[UIX(u)]) {
                const uint64_t p = t0 + uint32_t(u * 64 + l);
                const uint32_t ml = maxlen[UIX(u)];
                uint32_t len0 = 0, d0 = 0, len1 = 0, d1 = 0;
                auto offer = [&](uint32_t L, uint32_t D) {
                    if (L >= 3 && (!d0 || D < d0)) { len0 = L; d0 = D; }
                    if (L > len1 || (L == len1 && L && D < d1)) { len1 = L; d1 = D; }
                };
                {   // the fixed distances against the 8 bytes in front of p
                    const uint32_t cap8 = ml < 8 ? ml : 8u;
                    uint32_t l8best = 0, dbest = 0, f0l = 0, f0d = 0;
                    CSH_UNROLL
                    for (int k = 0; k < 6; k++) {
                        const uint32_t d = k < 4 ? uint32_t(k + 1) : (k == 4 ? 6u : 8u);
                        if (uint64_t(d) > p) continue;
                        const uint64_t shifted = d == 8 ? lo[UIX(u)] : ((hi[UIX(u)] << (8 * d)) | (lo[UIX(u)] >> (64 - 8 * d)));
                        const uint64_t x = hi[UIX(u)] ^ shifted;
                        uint32_t l8 = x ? ctz64(x) >> 3 : 8u;
                        if (l8 > cap8) l8 = cap8;
                        if (l8 > l8best) { l8best = l8; dbest = d; }
                        if (l8 >= 3 && !f0d) { f0l = l8; f0d = d; }
                    }
                    if (l8best) {
                        uint32_t ln = l8best;
                        if (l8best == 8 && ml > 8) ln = lz_lcp(data, p, dbest, ml, 8);
                        if (f0d && f0d != dbest) offer(f0l, f0d);
                        offer(ln, dbest);
                    }
                }
                CSH_UNROLL
                for (int w = 0; w < NW; w++) if (w < nw[UIX(u)]) {
                    uint32_t ln;
                    const uint32_t d = dw[UIX(u)][w];
                    if (xw[UIX(u)][w]) { ln = ctz64(xw[UIX(u)][w]) >> 3; if (ln > ml) ln = ml; }
                    else if (ml <= 8) ln = ml;
                    else {
                        // all eight agree.  How far it goes matters only if it can become c0 (nearer than c0) or c1 (longer than c1, or as long and nearer):
                        // a candidate that differs inside the `need` bytes it would have to match changes nothing, whatever else is offered later (c0 only
                        // gets nearer, c1 only better) -- in a run of equal bytes every candidate but the first leaves here
                        if (d0 && d > d0) {
                            const uint32_t need = d < d1 ? len1 : len1 + 1u;
                            if (need > ml) continue;
                            if (need > 8 && load64u(data + p + need - 8) != load64u(data + p - d + need - 8)) continue;
                        }
                        ln = lz_lcp(data, p, d, ml, 8);
                    }
                    offer(ln, d);
                }
                if (len1 <= len0) { len1 = 0; d1 = 0; }
                const uint32_t off = uint32_t(p - start);
                cand[uint64_t(off & (CSP_DEEP_SEG - 1)) * 64 + (off / CSP_DEEP_SEG)] =
                    uint64_t(len0) | (uint64_t(d0) << 9) | (uint64_t(len1) << 25) | (uint64_t(d1) << 34) | ((hi[UIX(u)] & 255ull) << 50);
            }
        }
        CSP_WG_SYNC();
        deep_insert(S, h4, ok4, h8, ok8, rel);
    }
    CSP_MEM_FENCE();
    CSP_WG_SYNC();
    if (!CSP_WAVE0) return;   // (the caller's barrier behind deep_chunk is where the other waves wait)
    // ---------------------------------------------------------------------------------------------------------------- C, D, F  (one wave)
    LV<uint32_t> ns;   // lane = segment: how many positions it holds
    LFOR(l) { const uint32_t s0 = uint32_t(l) * CSP_DEEP_SEG; ns[l] = s0 >= n ? 0u : (n - s0 < CSP_DEEP_SEG ? n - s0 : uint32_t(CSP_DEEP_SEG)); }
    // the first pass's counts: every byte of the chunk a literal
    LFOR(l) for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) S.hist[i] = 0;
    CSP_WAVE_SYNC();
    for (uint64_t p0 = start; p0 < end; p0 += 512) LFOR(l) {   // eight bytes per lane and step
        const uint64_t p = p0 + uint32_t(l) * 8u;
        if (p + 8 <= end) { const uint64_t v = load64u(data + p); CSH_UNROLL for (int k = 0; k < 8; k++) atomicAdd(&S.hist[uint32_t(v >> (8 * k)) & 255u], 1u); }
        else for (uint64_t q = p; q < end; q++) atomicAdd(&S.hist[data[q]], 1u);
    }
    CSP_WAVE_SYNC();
    for (int it = 0; it < iters; it++) {
        const bool final_pass = it + 1 == iters;
        {
            LV<uint64_t> a, b;
            LFOR(l) { a[l] = 0; b[l] = 0; for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) { if (i < CSP_NLIT) a[l] += S.hist[i]; else b[l] += S.hist[i]; } }
            const uint32_t tl = uint32_t(lsum(a)), td = uint32_t(lsum(b));
            LFOR(l) {
                for (uint32_t i = uint32_t(l); i < 256; i += 64) S.lit_cost[i] = uint16_t(S.hist[i] ? cost16_of(S.hist[i], tl) : cost16_of(1, 2 * tl));
                for (uint32_t ln = 3 + uint32_t(l); ln <= 258; ln += 64) {
                    const uint32_t c = len_code_of(ln), f = S.hist[257 + c];
                    S.len_cost[ln] = uint16_t((it == 0 ? uint32_t(CSP_DEEP_START) : f ? cost16_of(f, tl) : cost16_of(1, 2 * tl)) + 16u * len_extra_of(c));
                }
                if (l < 30) {
                    const uint32_t f = S.hist[CSP_NLIT + l];
                    S.dist_cost[l] = uint16_t((it == 0 ? uint32_t(CSP_DEEP_START) : td == 0 ? 80u : f ? cost16_of(f, td) : cost16_of(1, 2 * td)) + 16u * dist_extra_of(uint32_t(l)));
                }
            }
            CSP_WAVE_SYNC();
            LFOR(l) for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) S.hist[i] = i == 256 ? 1u : 0u;
            if (final_pass && want_tokens) LFOR(l) for (uint32_t i = uint32_t(l) * 8u; i < 32768u; i += 512u) *reinterpret_cast<unsigned long long *>(tk + i) = 0ull;
            CSP_WAVE_SYNC();
        }
        // D: backwards through the segments; W[l][k] = cost of position i + 1 + k
        LVArr<uint32_t, CSP_DEEP_CAP> W;
        LFOR(l) { uint32_t *w = W[l]; CSH_UNROLL for (int k = 0; k < int(CSP_DEEP_CAP); k++) w[k] = 0; costs[uint64_t(ns[l]) * 64 + uint32_t(l)] = 0; }   // (the end of a segment costs nothing)
        uint32_t lenc[CSP_DEEP_CAP + 1];
        CSH_UNROLL
        for (int k = 3; k <= int(CSP_DEEP_CAP); k++) lenc[k] = S.len_cost[k];
        LV<uint64_t> cnext;   // the candidate word of the step to come: asked for a step ahead (the addresses do not depend on the path)
        LFOR(l) cnext[l] = coherent_load(&cand[uint64_t(CSP_DEEP_SEG - 1) * 64 + uint32_t(l)]);
        for (uint32_t i = CSP_DEEP_SEG; i-- > 0;) {
            LFOR(l) {
                const uint64_t c = cnext[l];
                if (i) cnext[l] = coherent_load(&cand[uint64_t(i - 1) * 64 + uint32_t(l)]);
                if (i >= ns[l]) continue;
                uint32_t *w = W[l];
                const uint32_t avail = ns[l] - i;
                uint32_t l0 = uint32_t(c) & 511u, l1 = uint32_t(c >> 25) & 511u;
                const uint32_t d0 = uint32_t(c >> 9) & 0xFFFFu, d1 = uint32_t(c >> 34) & 0xFFFFu;
                if (l0 > avail) l0 = avail;
                if (l1 > avail) l1 = avail;
                uint32_t best = ((w[0] + S.lit_cost[uint32_t(c >> 50) & 255u]) << 9) | 1u;
                if (l0 >= 3) {
                    const uint32_t dc0 = S.dist_cost[dist_code_of(d0)];
                    const uint32_t dc1 = l1 > l0 ? uint32_t(S.dist_cost[dist_code_of(d1)]) : 0u;
                    const uint32_t top = l1 > l0 ? l1 : l0;
                    CSH_UNROLL
                    for (int k = 3; k <= int(CSP_DEEP_CAP); k++) {
                        const uint32_t key = ((w[k - 1] + lenc[k] + (uint32_t(k) <= l0 ? dc0 : dc1)) << 9) | uint32_t(k);
                        if (uint32_t(k) <= top && key < best) best = key;
                    }
                    if (l0 > CSP_DEEP_CAP) { const uint32_t key = ((costs[uint64_t(i + l0) * 64 + uint32_t(l)] + S.len_cost[l0] + dc0) << 9) | l0; if (key < best) best = key; }
                    if (l1 > l0 && l1 > CSP_DEEP_CAP) { const uint32_t key = ((costs[uint64_t(i + l1) * 64 + uint32_t(l)] + S.len_cost[l1] + dc1) << 9) | l1; if (key < best) best = key; }
                }
                const uint32_t cst = best >> 9;
                choice[uint64_t(i) * 64 + uint32_t(l)] = uint16_t(best & 511u);
                costs[uint64_t(i) * 64 + uint32_t(l)] = cst;
                CSH_UNROLL
                for (int k = int(CSP_DEEP_CAP) - 1; k > 0; k--) w[k] = w[k - 1];
                w[0] = cst;
            }
        }
        // F: forwards over all positions; a lane acts at the positions its path visits (nxt) -- no load waits for the path -- and counts their symbols
        // (final pass: and marks them)
        LV<uint32_t> nxt;
        LFOR(l) nxt[l] = 0;
        for (uint32_t i0 = 0; i0 < CSP_DEEP_SEG; i0 += 8) {
            LVArr<uint64_t, 8> cw;
            LVArr<uint32_t, 8> chw;
            LFOR(l) {
                uint64_t *cc = cw[l]; uint32_t *hh = chw[l];
                CSH_UNROLL
                for (int k = 0; k < 8; k++) { cc[k] = coherent_load(&cand[uint64_t(i0 + k) * 64 + uint32_t(l)]); hh[k] = choice[uint64_t(i0 + k) * 64 + uint32_t(l)]; }
            }
            LFOR(l) {
                const uint64_t *cc = cw[l]; const uint32_t *hh = chw[l];
                CSH_UNROLL
                for (int k = 0; k < 8; k++) {
                    const uint32_t i = i0 + uint32_t(k);
                    if (i != nxt[l] || i >= ns[l]) continue;
                    const uint64_t c = cc[k];
                    const uint32_t ch = hh[k];
                    if (final_pass && want_tokens) tk[uint32_t(l) * CSP_DEEP_SEG + i] = 1;
                    if (ch == 1) { atomicAdd(&S.hist[uint32_t(c >> 50) & 255u], 1u); nxt[l] = i + 1; }
                    else {
                        const uint32_t avail = ns[l] - i;
                        uint32_t l0 = uint32_t(c) & 511u;
                        if (l0 > avail) l0 = avail;
                        const uint32_t d = ch <= l0 ? uint32_t(c >> 9) & 0xFFFFu : uint32_t(c >> 34) & 0xFFFFu;
                        atomicAdd(&S.hist[257 + len_code_of(ch)], 1u); atomicAdd(&S.hist[CSP_NLIT + dist_code_of(d)], 1u);
                        nxt[l] = i + ch;
                    }
                }
            }
        }
        CSP_WAVE_SYNC();
    }
    if (!want_tokens) return;
    // ---------------------------------------------------------------------------------------------------------------- the tokens, tile by tile
    CSP_MEM_FENCE();
    for (uint64_t t0 = start; t0 < end; t0 += 64) {
        const uint32_t count = end - t0 < 64 ? uint32_t(end - t0) : 64u;
        LV<uint32_t> mlen, mdist, lit, visited;
        LFOR(l) {
            mlen[l] = 0; mdist[l] = 0; lit[l] = 0; visited[l] = 0;
            if (uint32_t(l) < count) {
                const uint32_t off = uint32_t(t0 - start) + uint32_t(l), seg = off / CSP_DEEP_SEG, i = off & (CSP_DEEP_SEG - 1);
                const uint64_t c = coherent_load(&cand[uint64_t(i) * 64 + seg]);
                lit[l] = uint32_t(c >> 50) & 255u;
                visited[l] = coherent_load(&tk[off]);
                if (visited[l]) {
                    const uint32_t ch = coherent_load(&choice[uint64_t(i) * 64 + seg]);
                    if (ch != 1) {
                        const uint32_t nseg = n - seg * CSP_DEEP_SEG < CSP_DEEP_SEG ? n - seg * CSP_DEEP_SEG : uint32_t(CSP_DEEP_SEG), avail = nseg - i;
                        uint32_t l0 = uint32_t(c) & 511u;
                        if (l0 > avail) l0 = avail;
                        mlen[l] = ch; mdist[l] = ch <= l0 ? uint32_t(c >> 9) & 0xFFFFu : uint32_t(c >> 34) & 0xFFFFu;
                    }
                }
            }
        }
        const uint64_t taken = lballot([&](int l) { return visited[l] != 0; });
        sink.tile(t0, count, taken, mlen, mdist, lit);
    }
}

}  // namespace csp
