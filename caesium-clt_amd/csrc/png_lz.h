// png_lz.h -- the LZ77 tokenizer of the deflate coder, one wave per chunk (statement: oracle/png_oracle.c tokenize()).
// Per step the wave takes a TILE of 64 consecutive positions, one per lane:
//   * candidates: up to four earlier positions from a 2048-bucket x 4-entry hash table in LDS (entries are positions
//     in front of the tile, so every lane can look up before anyone inserts), plus the fixed distances 1,2,3,4,6,8 checked
//     against a 16-byte window held in registers;
//   * after the lookups the tile inserts, per hash, its LAST position (found by a max-by-retry write to a byte map);
//   * parse: lazy flags by a lane shift, then a uniform walk that only stops at matches (ballot + count-trailing-zeros).
// The caller's sink sees, per tile, which lanes are visited and their (length, distance) or literal.
#pragma once
#include "png_types.h"
#include "png_wave.h"

namespace csp {

struct LzLds {
    uint64_t bucket[1u << CSP_HASH_BITS];   // four 16-bit positions, most recent in the low bits; position = offset from (chunk start - 32768)
    uint8_t lastlane[1u << CSP_HASH_BITS];
    uint32_t cnt[256];      // byte counts of the chunk; lz_insert borrows the first 64 words (the counts are made and used between the seeding and the tiles)      // byte counts of the chunk
    uint16_t cost16[256];   // what a literal costs in this chunk, in 1/16 bit (oracle/png_oracle.c literal_costs)
};
enum { CSP_MATCH_BASE16 = 320, CSP_MATCH_RULE_MAXLEN = 8 };

__device__ __forceinline__ static uint64_t load64u(const uint8_t *p) {
#ifdef CSH_EMUL
    uint64_t v; memcpy(&v, p, 8); return v;
#else
    return *reinterpret_cast<const uint64_t *>(p);   // unaligned global loads are legal on gfx9
#endif
}
__device__ __forceinline__ static uint32_t load32u(const uint8_t *p) {
#ifdef CSH_EMUL
    uint32_t v; memcpy(&v, p, 4); return v;
#else
    return *reinterpret_cast<const uint32_t *>(p);
#endif
}
// bytes `shift` .. `shift` + 3 of the eight bytes lo, hi (v_alignbyte_b32)
__device__ __forceinline__ static uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t shift) {
#ifdef CSH_EMUL
    return uint32_t(((uint64_t(hi) << 32) | lo) >> (8u * shift));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, shift);
#endif
}
__device__ __forceinline__ static uint32_t lz_hash(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - CSP_HASH_BITS); }
__device__ __forceinline__ static uint32_t ctz64(uint64_t x) { return uint32_t(__ffsll((unsigned long long)x) - 1); }
// common prefix of data[p..] and data[p-d..], at most maxlen, given that the first `from` bytes are known to agree.  32 bytes per round trip (eight loads
// in flight; a wave that follows a 258-byte match waits eight times, not thirty-two); the bytes past maxlen it may fetch lie inside the pool's slack
__device__ __forceinline__ static uint32_t lz_lcp(const uint8_t *data, uint64_t p, uint32_t d, uint32_t maxlen, uint32_t from) {
    uint32_t k = from;
    while (k < maxlen) {
        const uint8_t *a = data + p + k, *b = a - d;
        const uint64_t x0 = load64u(a) ^ load64u(b), x1 = load64u(a + 8) ^ load64u(b + 8), x2 = load64u(a + 16) ^ load64u(b + 16), x3 = load64u(a + 24) ^ load64u(b + 24);
        if (x0) { k += ctz64(x0) >> 3; break; }
        if (x1) { k += 8 + (ctz64(x1) >> 3); break; }
        if (x2) { k += 16 + (ctz64(x2) >> 3); break; }
        if (x3) { k += 24 + (ctz64(x3) >> 3); break; }
        k += 32;
    }
    return k < maxlen ? k : maxlen;
}
__device__ __forceinline__ static uint32_t len_code_of(uint32_t len) {   // 0..28
    if (len < 11) return len - 3;
    if (len == 258) return 28;
    const uint32_t x = len - 3, eb = (31u - uint32_t(__clz(x))) - 2u;
    return 4u * eb + 4u + ((x >> eb) & 3u);
}
__device__ __forceinline__ static uint32_t len_extra_of(uint32_t code) { return code < 8 || code == 28 ? 0u : (code >> 2) - 1u; }
__device__ __forceinline__ static uint32_t len_base_of(uint32_t code) { return code < 8 ? 3u + code : code == 28 ? 258u : ((4u | (code & 3u)) << ((code >> 2) - 1u)) + 3u; }
__device__ __forceinline__ static uint32_t dist_code_of(uint32_t dist) {   // 0..29
    if (dist < 5) return dist - 1;
    const uint32_t x = dist - 1, eb = (31u - uint32_t(__clz(x))) - 1u;
    return 2u * eb + 2u + ((x >> eb) & 1u);
}
__device__ __forceinline__ static uint32_t dist_extra_of(uint32_t code) { return code < 4 ? 0u : (code >> 1) - 1u; }
__device__ __forceinline__ static uint32_t dist_base_of(uint32_t code) { return code < 4 ? code + 1u : ((2u | (code & 1u)) << ((code >> 1) - 1u)) + 1u; }

// insert the tile [t0, t1) into the table: per hash its last position.  Which lane that is: a byte per hash names SOME lane that has it (whichever write the
// LDS kept), and the lanes that share it take the maximum of their numbers in that lane's slot -- two LDS round trips whatever the tile holds
__device__ __forceinline__ static void lz_insert(LzLds &L, const LV<uint32_t> &hash, const LV<uint32_t> &hashable, const LV<uint32_t> &rel) {
    LFOR(l) { L.cnt[l] = 0; if (hashable[l]) L.lastlane[hash[l]] = uint8_t(l); }
    CSP_WAVE_SYNC();
    LV<uint32_t> rep;
    LFOR(l) { rep[l] = hashable[l] ? uint32_t(L.lastlane[hash[l]]) : 0u; if (hashable[l]) atomicMax(&L.cnt[rep[l]], uint32_t(l)); }
    CSP_WAVE_SYNC();
    LFOR(l) if (hashable[l] && L.cnt[rep[l]] == uint32_t(l) && rel[l] != 0xFFFFu) L.bucket[hash[l]] = (L.bucket[hash[l]] << 16) | rel[l];
    CSP_WAVE_SYNC();
}

// Tokenize data[start, end), a chunk of a stream of `total` bytes whose tiles are aligned to multiples of 64 of the
// stream.  sink.tile(t0, count, taken, len, dist, lit) is called once per tile: lane l describes position t0 + l; bit l
// of `taken` says the parse visits it; len 0 = literal `lit`.
template <class Sink>
__device__ static void lz_chunk(const uint8_t *data, uint64_t total, uint64_t start, uint64_t end, LzLds &L, Sink &sink) {
    {
        // a table of its own: empty, then seeded with the 32 KiB in front of the chunk
        LFOR(l) for (uint32_t i = uint32_t(l); i < (1u << CSP_HASH_BITS); i += 64) L.bucket[i] = ~0ull;
        CSP_WAVE_SYNC();
        const uint64_t seed0 = start > 32768 ? start - 32768 : 0;
        for (uint64_t t0 = seed0; t0 < start; t0 += 64) {
            LV<uint32_t> hash, hashable, rel;
            LFOR(l) {
                const uint64_t p = t0 + uint32_t(l);
                hashable[l] = p + 4 <= total ? 1u : 0u;
                hash[l] = hashable[l] ? lz_hash(uint32_t(load64u(data + p))) : 0u;
                rel[l] = uint32_t(p + 32768 - start);
            }
            lz_insert(L, hash, hashable, rel);
        }
    }
    {
        // literal costs of this chunk: log2(total / count) in integer arithmetic (exponent + four mantissa bits), 1 .. 15 bits.  A match
        // of up to eight bytes is taken only if the literals it replaces cost more than it does (a code pair + the distance's extra bits)
        LFOR(l) for (uint32_t i = uint32_t(l); i < 256; i += 64) L.cnt[i] = 0;
        CSP_WAVE_SYNC();
        for (uint64_t p0 = start; p0 < end; p0 += 512) LFOR(l) {   // eight bytes per lane and step
            const uint64_t p = p0 + uint32_t(l) * 8u;
            if (p + 8 <= end) { const uint64_t v = load64u(data + p); CSH_UNROLL for (int k = 0; k < 8; k++) atomicAdd(&L.cnt[uint32_t(v >> (8 * k)) & 255u], 1u); }
            else for (uint64_t q = p; q < end; q++) atomicAdd(&L.cnt[data[q]], 1u);
        }
        CSP_WAVE_SYNC();
        const uint32_t total = uint32_t(end - start);
        LFOR(l) for (uint32_t i = uint32_t(l); i < 256; i += 64) {
            uint32_t c = 240;
            if (L.cnt[i]) {
                const uint32_t q = (total << 8) / L.cnt[i];   // >= 256; total <= 32768
                const uint32_t e = 31u - uint32_t(__clz(q));
                c = 16u * (e - 8u) + (((q << 4) >> e) & 15u);
                c = c < 16u ? 16u : c > 240u ? 240u : c;
            }
            L.cost16[i] = uint16_t(c);
        }
        CSP_WAVE_SYNC();
    }
    uint64_t carry = start;
    for (uint64_t t0 = start; t0 < end; t0 += 64) {
        const uint32_t count = end - t0 < 64 ? uint32_t(end - t0) : 64u;
        LV<uint32_t> hash, hashable, rel, mlen, mdist, lit;
        LFOR(l) {
            const uint64_t p = t0 + uint32_t(l);
            mlen[l] = 0; mdist[l] = 0; lit[l] = 0; hashable[l] = 0; hash[l] = 0; rel[l] = uint32_t(p + 32768 - start);
            if (uint32_t(l) < count) {
                const uint32_t maxlen = end - p < 258 ? uint32_t(end - p) : 258u;
                const uint64_t hi = load64u(data + p);
                lit[l] = uint32_t(hi & 255u);
                // First, cheaply: does ANY candidate -- the six fixed distances, the bucket's entries -- agree with p in its first three bytes?  Nothing shorter is a
                // match, and on photographic data hardly a position has one (the bucket's entries are mostly other strings with the same hash): 32-bit compares on
                // the bytes at hand decide it, and the judging below -- 64-bit compares, lengths, the longest-wins rules -- runs for the lanes that need it only
                uint32_t bl = 0, bd = 0;
                const uint64_t lo = load64u(data + p - 8);   // bytes p-8..p-1; in front of the data (p < 8) whatever the pool holds there: guarded by d <= p
                const uint32_t h0 = uint32_t(hi), l0 = uint32_t(lo), l1 = uint32_t(lo >> 32);
                bool any3 = false;
                CSH_UNROLL
                for (int k = 0; k < 6; k++) {
                    const uint32_t d = k < 4 ? uint32_t(k + 1) : (k == 4 ? 6u : 8u);
                    if (uint64_t(d) > p) continue;
                    const uint32_t s4 = d == 8 ? l0 : d == 4 ? l1 : d == 6 ? align_bytes(l1, l0, 2) : align_bytes(h0, l1, 4u - d);   // bytes p-d .. p-d+3
                    any3 |= ((h0 ^ s4) & 0xFFFFFFu) == 0u;
                }
                uint32_t dw[CSP_WAYS];
                int nw = 0;
                if (p + 4 <= total) {
                    hashable[l] = 1;
                    hash[l] = lz_hash(h0);
                    const uint64_t b = L.bucket[hash[l]];
                    for (int w = 0; w < int(CSP_WAYS); w++) {
                        const uint32_t r = uint32_t(b >> (16 * w)) & 0xFFFFu;
                        if (r == 0xFFFFu) break;
                        const uint32_t d = rel[l] - r;
                        if (d > 32768u) break;
                        dw[nw++] = d;
                    }
                    uint32_t cw[CSP_WAYS];
                    CSH_UNROLL
                    for (int w = 0; w < int(CSP_WAYS); w++) cw[w] = w < nw ? load32u(data + p - dw[w]) : ~h0;   // (fetched together: one round trip)
                    CSH_UNROLL
                    for (int w = 0; w < int(CSP_WAYS); w++) any3 |= ((h0 ^ cw[w]) & 0xFFFFFFu) == 0u;
                }
                if (any3 && maxlen >= 3) {
                    // the six fixed distances against the 8 bytes in front of p
                    {
                        const uint32_t cap8 = maxlen < 8 ? maxlen : 8u;
                        uint32_t l8best = 0, dbest = 0;
                        CSH_UNROLL
                        for (int k = 0; k < 6; k++) {
                            const uint32_t d = k < 4 ? uint32_t(k + 1) : (k == 4 ? 6u : 8u);
                            if (uint64_t(d) > p) continue;
                            const uint64_t shifted = d == 8 ? lo : ((hi << (8 * d)) | (lo >> (64 - 8 * d)));   // bytes p-d .. p-d+7
                            const uint64_t x = hi ^ shifted;
                            uint32_t l8 = x ? ctz64(x) >> 3 : 8u;
                            if (l8 > cap8) l8 = cap8;
                            if (l8 > l8best) { l8best = l8; dbest = d; }
                        }
                        if (l8best) { bl = l8best; bd = dbest; if (l8best == 8 && maxlen > 8) bl = lz_lcp(data, p, dbest, maxlen, 8); }
                    }
                    // the bucket's candidates: their first eight bytes fetched together; the order of evaluation, and with it every tie, stays the serial statement's
                    uint64_t xw[CSP_WAYS];
                    CSH_UNROLL
                    for (int w = 0; w < int(CSP_WAYS); w++) xw[w] = w < nw ? hi ^ load64u(data + p - dw[w]) : 0ull;
                    CSH_UNROLL
                    for (int w = 0; w < int(CSP_WAYS); w++) if (w < nw) {
                        uint32_t ln;
                        if (xw[w]) { ln = ctz64(xw[w]) >> 3; if (ln > maxlen) ln = maxlen; }
                        else ln = maxlen > 8 ? lz_lcp(data, p, dw[w], maxlen, 8) : maxlen;
                        if (ln > bl) { bl = ln; bd = dw[w]; }
                    }
                }
                if (bl < 3 || (bl == 3 && bd > 8)) { bl = 0; bd = 0; }
                if (bl && bl <= uint32_t(CSP_MATCH_RULE_MAXLEN)) {   // the bytes are the low ones of `hi`; lengths 3..8 have no extra bits
                    uint32_t litc = 0;
                    for (uint32_t k = 0; k < bl; k++) litc += L.cost16[uint32_t(hi >> (8 * k)) & 255u];
                    if (litc < uint32_t(CSP_MATCH_BASE16) + 16u * dist_extra_of(dist_code_of(bd))) { bl = 0; bd = 0; }
                }
                mlen[l] = bl; mdist[l] = bd;
            }
        }
        CSP_WAVE_SYNC();
        lz_insert(L, hash, hashable, rel);
        // lazy evaluation inside the tile, then the walk
        LV<uint32_t> nextlen;
#ifdef CSH_EMUL
        for (int l = 0; l < 64; l++) nextlen.v[l] = l < 63 ? mlen.v[l + 1] : 0u;
#else
        nextlen.v = uint32_t(__shfl_down(int(mlen.v), 1, 64));
#endif
        LFOR(l) if (mlen[l] && uint32_t(l) + 1 < count && nextlen[l] > mlen[l]) mlen[l] = 0;
        const uint64_t matches = lballot([&](int l) { return mlen[l] != 0; });
        uint64_t taken = 0;
        uint32_t cur = carry > t0 ? uint32_t(carry - t0) : 0u;
        while (cur < count) {
            const uint64_t ahead = matches & ~lanes_below(int(cur));
            if (!ahead) { taken |= (count == 64 ? ~0ull : lanes_below(int(count))) & ~lanes_below(int(cur)); cur = count; break; }
            const uint32_t m = ctz64(ahead);
            taken |= (m == 63 ? ~0ull : lanes_below(int(m) + 1)) & ~lanes_below(int(cur));
#ifdef CSH_EMUL
            cur = m + mlen.v[m];
#else
            cur = m + uint32_t(__builtin_amdgcn_readlane(int(mlen.v), int(m)));
#endif
        }
        carry = t0 + cur;
        sink.tile(t0, count, taken, mlen, mdist, lit);
    }
}

}  // namespace csp
