// k_webp.hip -- the lossy WebP row on the device: the import and the coder back end; W2 (libwebp's encoder) is k_vp8enc.hip.
// Statement: oracle/webp_oracle.c (W1, pinned to WebPPictureImportRGB) and oracle/vp8enc_oracle.c (the bitstream, pinned to WebPEncode).
//   k_webp_yuv        W1: RGB -> YUV 4:2:0 planes padded to whole macroblocks, one lane per sample; libwebp's import (chroma averaged in gamma-0.80 linear light)
//   W3: the boolean entropy coder is a serial chain per partition; WHICH decisions it takes (the token tree over the levels, the contexts out of the
//   non-zero masks, the frame's probabilities) is known for every block at once.  So:
//   chunk_stats       (k_vp8enc.hip, inside k_vp8_loop: the walk that keeps the frame's statistics) also counts every block's decisions;
//   an exclusive scan over the macroblocks in raster order gives every macroblock its place in the stream;
//   k_webp_decisions  the same walk, lanes = the blocks of a macroblock, writes (bit, probability) pairs -- two bytes a decision;
//   k_webp_hdr        partition 0 the same way (segment / filter / quantiser fields, probability updates, every macroblock's segment and modes);
//   k_webp_bool       ONE LANE per partition runs the coder over its stretch of pairs: no tree, no table, no levels -- 64 chains to a wave on the vector unit;
//   k_webp_assemble   RIFF / VP8 headers and the two partitions into the file.
// libwebp's token buffer makes ONE token partition; parallelism is across the files of the batch, as in the reference's par_iter.
#include "vp8enc_dev.h"
#include "devmem.hpp"
#include "kernels.h"

namespace csw {

__device__ __forceinline__ static int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

// libwebp's import (oracle: cso_webp_rgb_to_yuv, pinned against WebPPictureImportRGB): chroma from the 2x2 block's mean in gamma-0.80 linear light
__global__ void __launch_bounds__(256) k_webp_yuv(const WebpImg *imgs, const uint8_t *rgb, uint8_t *work) {
    const WebpImg &im = imgs[blockIdx.y];
    const int w = int(im.width), h = int(im.height), ys = int(im.mbw) * 16, cs = int(im.mbw) * 8, nc = int(im.ncomp), cw = (w + 1) >> 1, ch = (h + 1) >> 1;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint8_t *src = rgb + im.rgb_off;
    auto px = [&](int yy, int xx, int &r, int &g, int &b) {
        const uint8_t *p = src + (size_t(yy < h ? yy : h - 1) * w + (xx < w ? xx : w - 1)) * nc;
        r = p[0]; g = nc >= 3 ? p[1] : p[0]; b = nc >= 3 ? p[2] : p[0];   // nc 2 / 4: the last sample is alpha, which the ALPH chunk carries (png_pipeline.cpp)
    };
    if (i < uint32_t(ys) * im.mbh * 16) {
        const int y = int(i / uint32_t(ys)), x = int(i - uint32_t(y) * ys);
        int r, g, b;
        px(y, x, r, g, b);
        work[im.y_off + i] = uint8_t((16839 * r + 33059 * g + 6420 * b + (16 << 16) + (1 << 15)) >> 16);
    }
    if (i < uint32_t(cs) * im.mbh * 8) {
        const int y = int(i / uint32_t(cs)), x = int(i - uint32_t(y) * cs);
        const int cx = x < cw ? x : cw - 1, cy = y < ch ? y : ch - 1;   // beyond the picture: the plane's last sample again
        // the two tables (578 bytes) stay in the first-level cache; twelve gathers per chroma sample next to twelve bytes from HBM
        int r = 0, g = 0, b = 0;
        for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) { int r1, g1, b1; px(2 * cy + dy, 2 * cx + dx, r1, g1, b1); r += kVp8GammaToLinear[r1]; g += kVp8GammaToLinear[g1]; b += kVp8GammaToLinear[b1]; }
        auto back = [&](int s) { const int pos = s >> 9, f = s & 511; return (int(kVp8LinearToGamma[pos + 1]) * f + int(kVp8LinearToGamma[pos]) * (512 - f) + 64) >> 7; };
        r = back(r); g = back(g); b = back(b);
        work[im.u_off + i] = uint8_t(clip8((-9719 * r - 19081 * g + 28800 * b + (128 << 18) + (1 << 17)) >> 18));
        work[im.v_off + i] = uint8_t(clip8((28800 * r - 24116 * g - 4684 * b + (128 << 18) + (1 << 17)) >> 18));
    }
}

// ---- W3.  A picture's room in the scratch region: partition 0 first, the token partition behind it
__device__ __forceinline__ static uint32_t webp_hdr_cap(const WebpImg &im) { return 2048u + ((im.out_cap - 4096u) >> 4); }   // a sixteenth of the file's room (48 bytes per macroblock to begin with): the frame header and the modes, sixteen of them in an i4x4 macroblock; grows with out_cap when a run is repeated
__device__ __forceinline__ static uint32_t webp_part_cap(const WebpImg &im) { return im.out_cap - 128u - webp_hdr_cap(im); }

// the token walk writing its decisions down: (bit, probability) pairs, two bytes each, in the order the coder takes them -- the probability is resolved here
// (the frame's table is final by now), so the coder behind it needs neither the levels nor the tables (k_webp_bool)
struct WriteSink {
    const uint8_t *probs;   // LDS
    uint16_t *out;
    __device__ __forceinline__ void ad(int bit, int idx) { *out++ = uint16_t((bit ? 1u : 0u) | (uint32_t(probs[idx]) << 1)); }
    __device__ __forceinline__ void ad10(int bit, int idx) { ad(bit, idx); }
    __device__ __forceinline__ void fx(int bit, int prob) { *out++ = uint16_t((bit ? 1u : 0u) | (uint32_t(prob) << 1)); }
};
// one wave per macroblock row; two macroblocks to a step: lanes 0..24 the blocks of one, lanes 32..56 those of the next
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_webp_decisions(const WebpImg *imgs, const int16_t *levels, const Vp8FrameDev *frames, const uint64_t *mb_base, const uint64_t *mb_off,
                                                                      const uint16_t *blk_cnt, uint16_t *stream, const uint32_t *status) {
    CSH_SHARED uint8_t s_probs[VP8_NSLOT];
    const WebpImg im = imgs[blockIdx.y];
    const int mbw = int(im.mbw), my = int(blockIdx.x);
    if (my >= int(im.mbh) || status[im.image]) return;
    const uint8_t *probs_g = frames[blockIdx.y].coeffs;
    LFOR(l) for (int i = l; i < VP8_NSLOT; i += 64) s_probs[i] = probs_g[i];
    CSP_WAVE_SYNC();
    const int16_t *row = levels + im.lev_off + size_t(my) * mbw * WEBP_MB_REC;
    for (int mx0 = 0; mx0 < mbw; mx0 += 2) {
        LV<uint32_t> nd;
        LV<uint64_t> at;
        LFOR(l) {
            const int mx = mx0 + (l >> 5), k = l & 31;
            at[l] = mx < mbw ? mb_base[blockIdx.y] + uint64_t(my) * mbw + mx : 0ull;
            nd[l] = mx < mbw ? uint32_t(blk_cnt[at[l] * 32 + uint32_t(k)]) : 0u;
        }
        uint32_t total;
        const LV<uint32_t> ex = lscan(nd, total);
        const uint32_t lower = csh::lget(ex, 32);
        LFOR(l) {
            const int mx = mx0 + (l >> 5), k = l & 31;
            if (mx < mbw && k < 25) {
                const int16_t *L = row + size_t(mx) * WEBP_MB_REC;
                const bool i4 = L[MB_INFO + 2] == 4;
                if (!(i4 && k == 0)) {
                    const uint32_t cur = nz_mask(L), top = my ? nz_mask(L - size_t(mbw) * WEBP_MB_REC) : 0u, left = mx ? nz_mask(L - WEBP_MB_REC) : 0u;
                    int type, first, ctx;
                    block_info(k, cur, top, left, i4, type, first, ctx);
                    WriteSink sink{s_probs, stream + mb_off[at[l]] + (ex[l] - ((l >> 5) ? lower : 0u))};
                    put_coeffs(sink, type, ctx, L + k * 16, first);
                }
            }
        }
    }
}
// ---- partition 0 the same way: its chain is the frame header's fields, the frame's probability updates, then every macroblock's segment and modes in raster
// order.  One lane per ITEM (item 0: the fields and the updates; item 1 + i: macroblock i) counts its decisions or writes them: the modes' contexts are the
// neighbours' modes, which k_vp8_loop left with the levels.  (oracle: part D of cso_vp8enc_encode_yuv)
struct HdrSink {
    uint16_t *out;   // nullptr: count only
    uint32_t n;
    __device__ __forceinline__ int put(int bit, int prob) { if (out) *out++ = uint16_t((bit ? 1u : 0u) | (uint32_t(prob) << 1)); n++; return bit; }
    __device__ __forceinline__ void bits(uint32_t v, int nb) { while (nb--) put(int((v >> nb) & 1u), 128); }
    __device__ __forceinline__ void sbits(int v, int nb) { if (!put(v != 0, 128)) return; if (v < 0) bits((uint32_t(-v) << 1) | 1u, nb + 1); else bits(uint32_t(v) << 1, nb + 1); }
};
__device__ static void hdr_item(HdrSink &e, const WebpImg &im, const int16_t *lev, const Vp8FrameDev *F, uint32_t item) {
    const int mbw = int(im.mbw);
    if (item == 0) {
        e.bits(0, 1); e.bits(0, 1);                         // colour space, clamping
        if (e.put(F->nseg > 1, 128)) {
            e.bits(uint32_t(F->update_map), 1);
            e.bits(1, 1); e.bits(1, 1);                     // segment data follows, as absolute values
            for (int s = 0; s < 4; s++) e.sbits(F->seg[s].quant, 7);
            for (int s = 0; s < 4; s++) e.sbits(F->seg[s].fstrength, 6);
            if (F->update_map) for (int s = 0; s < 3; s++) if (e.put(F->seg_probs[s] != 255, 128)) e.bits(uint32_t(F->seg_probs[s]), 8);
        }
        e.bits(0, 1); e.bits(uint32_t(F->filter_level), 6); e.bits(0, 3);   // normal loop filter, level, sharpness
        e.bits(0, 1);                                       // no filter deltas
        e.bits(0, 2);                                       // one token partition
        e.bits(uint32_t(F->base_quant), 7);
        e.sbits(0, 4); e.sbits(0, 4); e.sbits(0, 4); e.sbits(F->dq_uv_dc, 4); e.sbits(F->dq_uv_ac, 4);
        e.bits(0, 1);                                       // refresh_entropy_probs
        for (int i = 0; i < VP8_NSLOT; i++) if (e.put(F->coeffs[i] != kVp8CoefProbs[i], kVp8CoefUpdateProbs[i])) e.bits(F->coeffs[i], 8);   // the frame's coefficient probabilities
        e.bits(0, 1);                                       // no skip flags
        return;
    }
    const int i = int(item) - 1, my = i / mbw, mx = i - my * mbw;
    const int16_t *I = lev + size_t(i) * WEBP_MB_REC + MB_INFO, *It = I - size_t(mbw) * WEBP_MB_REC, *Il = I - WEBP_MB_REC;
    const int ym = I[2], cm = I[3];
    if (F->update_map) { const int sg = I[21]; if (e.put(sg >= 2, F->seg_probs[0])) e.put(sg & 1, F->seg_probs[2]); else e.put(sg & 1, F->seg_probs[1]); }
    if (ym == 4) {
        e.put(0, 145);                                                                    // i4x4: sixteen sub-block modes, each after the modes above and to the left
        for (int k = 0; k < 16; k++) {
            const int bx = k & 3, by = k >> 2, m = I[4 + k];
            const int tmode = by ? int(I[4 + k - 4]) : (my ? int(It[4 + 12 + bx]) : 0), lmode = bx ? int(I[4 + k - 1]) : (mx ? int(Il[4 + by * 4 + 3]) : 0);
            const uint8_t *pr = kVp8BModeProbs + (tmode * 10 + lmode) * 9;
            // the key-frame sub-block mode tree (RFC 6386 11.2) in libwebp's mode numbering
            if (e.put(m != 0, pr[0]) && e.put(m != 1, pr[1]) && e.put(m != 2, pr[2])) {
                if (!e.put(m >= 6, pr[3])) { if (e.put(m != 3, pr[4])) e.put(m != 4, pr[5]); }
                else if (e.put(m != 6, pr[6]) && e.put(m != 7, pr[7])) e.put(m != 8, pr[8]);
            }
        }
    } else {
        e.put(1, 145);                                                                    // i16x16: (TM | H) : (V | DC)
        if (e.put(ym == 1 || ym == 3, 156)) e.put(ym == 1, 128); else e.put(ym == 2, 163);
    }
    if (e.put(cm != 0, 142) && e.put(cm != 2, 114)) e.put(cm != 3, 183);
}
// hdr_base[image]: the picture's first header item among all counted things (behind the macroblocks); WRITE: cnt is the scan's output
template <bool WRITE>
__global__ void __launch_bounds__(256) k_webp_hdr(const WebpImg *imgs, const int16_t *levels, const Vp8FrameDev *frames, const uint64_t *hdr_base, uint32_t *cnt,
                                                  const uint64_t *off, uint16_t *stream, const uint32_t *status) {
    const WebpImg &im = imgs[blockIdx.y];
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item > im.mbw * im.mbh) return;
    if (WRITE && status[im.image]) return;
    const uint64_t at = hdr_base[blockIdx.y] + item;
    HdrSink e{WRITE ? stream + off[at] : nullptr, 0u};
    hdr_item(e, im, levels + im.lev_off, frames + blockIdx.y, item);
    if (!WRITE) cnt[at] = e.n;
}

// ---- the boolean coder, a partition in many pieces.  One chain = one partition of one picture: d1 - d0 decisions.  What makes the coder serial is its range (one of
// 128 values after every renormalisation); the low end of the interval only ADDS up.  So:
//   k_bool_scan   per piece of BOOL_SEG decisions and for each of the 128 ranges it could start from: the range it ends with and the bits it shifts out (lanes = start ranges);
//   k_bool_chain  per chain, piece after piece: the range and the bit position every piece really starts with (a walk through the maps);
//   k_bool_code   ONE LANE per piece runs the coder (oracle: boolenc) from its true range, aligned to its true bit position, with a low end of 0: the bytes it
//                 completes are its own places in the partition; what it still holds at its end, and a carry out of its first byte, go to
//   k_bool_merge  per chain, piece after piece: the held bits of a piece are added into the first bytes of the next (the sum of the pieces' numbers IS the coder's
//                 number), carries ripple back through the bytes already there; then the coder's closing bits.
struct WebpChain { uint32_t image, part; uint64_t first, nmb; };
struct LVA4 {   // four values per lane (emulation: per lane)
#ifdef CSH_EMUL
    int v[64][4];
    __device__ int (&operator[](int l))[4] { return v[l]; }
#else
    int v[4];
    __device__ __forceinline__ int (&operator[](int))[4] { return v; }
#endif
};   // part 0: the token partition (first / nmb: macroblocks); part 0xFFFFFFFF: partition 0 (first / nmb: its items)
enum { BOOL_SEG = 2048 };   // decisions per piece (the launcher passes it on: tests shorten it to make pieces that complete no byte at all)
struct BoolPiece { uint32_t range, carry_front, pend; uint32_t pad; uint64_t bits0; };   // what k_bool_chain / k_bool_code leave per piece
__device__ __forceinline__ static uint32_t bytes_after(uint64_t S) { return S ? uint32_t((S - 1) >> 3) : 0u; }   // bytes the coder has completed after S shifted bits
__device__ __forceinline__ static int nb_after(uint64_t S) { return -8 + int(S - 8ull * bytes_after(S)); }        // and its bit count then (-7 .. 0; -8 before the first bit)
// chain of piece `seg` (seg_start: pieces before each chain, nchains + 1 entries)
__device__ __forceinline__ static uint32_t chain_of(const uint32_t *seg_start, uint32_t nchains, uint32_t seg) {
    uint32_t lo = 0, hi = nchains;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_start[mid] <= seg) lo = mid; else hi = mid; }
    return lo;
}
// one decision for one range: the range after it and the bits it shifts out
__device__ __forceinline__ static void range_step(int &r, int &n, uint32_t v) {
    const int sp = (r * int(v >> 1)) >> 8, q = (v & 1u) ? r - sp - 1 : sp, sh = __clz(uint32_t(q + 1)) - 24;
    r = ((q + 1) << sh) - 1; n += sh;
}
// A wave takes FOUR pieces.  All 128 start ranges are followed through the first BOOL_WARM decisions of a piece only (two to a lane): by then they have run together into
// a handful of distinct ranges (16 or fewer for 99 pieces in 100; measured on the configs[3] pictures), and a row of sixteen lanes carries those through the rest of the
// piece, the four pieces side by side.  A piece with more than sixteen survivors is followed with all 128 to its end, as before.
enum { BOOL_WARM = 128, BOOL_SLOTS = 16 };
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_bool_scan(const WebpChain *chains, uint32_t nchains, const uint32_t *seg_start, const uint64_t *mb_off, const uint16_t *stream, uint32_t *maps, uint32_t seglen) {
    CSH_SHARED uint32_t s_owner[4][256];        // per piece: the first start range (0..127) that arrived at each range
    CSH_SHARED uint32_t s_slot[4][256];         // per piece: range -> slot
    CSH_SHARED uint32_t s_res[4][BOOL_SLOTS];   // per piece and slot: range | bits << 8 after the rest of the piece
    CSH_SHARED uint32_t s_rng[4][BOOL_SLOTS];   // per piece and slot: the range it stands for after the warm-up
    const uint32_t nseg = seg_start[nchains], seg0 = blockIdx.x * 4u;
    if (seg0 >= nseg) return;
    LVA4 r0, r1, n0, n1;
    uint64_t dbeg[4], dend[4];
    uint32_t nslots[4];
    CSH_UNROLL
    for (int g = 0; g < 4; g++) {
        const uint32_t seg = seg0 + uint32_t(g);
        dbeg[g] = 0; dend[g] = 0; nslots[g] = 0;
        LFOR(l) { r0[l][g] = 127 + l; r1[l][g] = 191 + l; n0[l][g] = 0; n1[l][g] = 0; }
        if (seg >= nseg) continue;
        const uint32_t c = chain_of(seg_start, nchains, seg);
        const WebpChain ch = chains[c];
        const uint64_t d0 = mb_off[ch.first] + uint64_t(seg - seg_start[c]) * seglen, de = mb_off[ch.first + ch.nmb], d1 = d0 + seglen < de ? d0 + seglen : de;
        const uint64_t dw = d0 + BOOL_WARM < d1 ? d0 + BOOL_WARM : d1;
        for (uint64_t d = d0; d < dw; d++) {
            const uint32_t v = stream[d];
            LFOR(l) { range_step(r0[l][g], n0[l][g], v); range_step(r1[l][g], n1[l][g], v); }
        }
        dbeg[g] = dw; dend[g] = d1;
        // the distinct ranges among the 128: the lowest start range that reached each one owns it
        LFOR(l) for (int i = l; i < 256; i += 64) s_owner[g][i] = 0xFFFFFFFFu;
        CSP_WAVE_SYNC();
        LFOR(l) { atomicMin(&s_owner[g][r0[l][g]], uint32_t(l)); atomicMin(&s_owner[g][r1[l][g]], uint32_t(64 + l)); }
        CSP_WAVE_SYNC();
        const uint64_t own0 = lballot([&](int l) { return s_owner[g][r0[l][g]] == uint32_t(l); }), own1 = lballot([&](int l) { return s_owner[g][r1[l][g]] == uint32_t(64 + l); });
        nslots[g] = csh::popc64(own0) + csh::popc64(own1);
        LFOR(l) {
            if ((own0 >> l) & 1u) { const uint32_t k = csh::popc64(own0 & lanes_below(l)); s_slot[g][r0[l][g]] = k; if (k < BOOL_SLOTS) s_rng[g][k] = uint32_t(r0[l][g]); }
            if ((own1 >> l) & 1u) { const uint32_t k = csh::popc64(own0) + csh::popc64(own1 & lanes_below(l)); s_slot[g][r1[l][g]] = k; if (k < BOOL_SLOTS) s_rng[g][k] = uint32_t(r1[l][g]); }
        }
        CSP_WAVE_SYNC();
    }
    // the rest of the four pieces side by side: lane (g, k) carries the range of slot k of piece g
    {
        LV<int> r, n;
        LV<uint64_t> d, de;
        LFOR(l) {
            const int g = l >> 4, k = l & 15;
            r[l] = 127; n[l] = 0; d[l] = 0; de[l] = 0;
            if (uint32_t(k) < nslots[g] && nslots[g] <= BOOL_SLOTS) {
                r[l] = int(s_rng[g][k]);
                d[l] = dbeg[g]; de[l] = dend[g];
            }
        }
        for (;;) {
            if (lballot([&](int l) { return d[l] < de[l]; }) == 0) break;
            LFOR(l) if (d[l] < de[l]) { range_step(r[l], n[l], stream[d[l]]); d[l]++; }
        }
        LFOR(l) { const int g = l >> 4, k = l & 15; s_res[g][k] = uint32_t(r[l]) | (uint32_t(n[l]) << 8); }
        CSP_WAVE_SYNC();
    }
    CSH_UNROLL
    for (int g = 0; g < 4; g++) {
        const uint32_t seg = seg0 + uint32_t(g);
        if (seg >= nseg) continue;
        if (nslots[g] > BOOL_SLOTS) {   // too many survivors: all 128 to the end
            for (uint64_t d = dbeg[g]; d < dend[g]; d++) {
                const uint32_t v = stream[d];
                LFOR(l) { range_step(r0[l][g], n0[l][g], v); range_step(r1[l][g], n1[l][g], v); }
            }
            LFOR(l) { maps[size_t(seg) * 128 + l] = uint32_t(r0[l][g]) | (uint32_t(n0[l][g]) << 8); maps[size_t(seg) * 128 + 64 + l] = uint32_t(r1[l][g]) | (uint32_t(n1[l][g]) << 8); }
        } else
            LFOR(l) {
                const uint32_t a = s_res[g][s_slot[g][r0[l][g]]], b2 = s_res[g][s_slot[g][r1[l][g]]];
                maps[size_t(seg) * 128 + l] = (a & 255u) | (((a >> 8) + uint32_t(n0[l][g])) << 8);
                maps[size_t(seg) * 128 + 64 + l] = (b2 & 255u) | (((b2 >> 8) + uint32_t(n1[l][g])) << 8);
            }
    }
}
__global__ void __launch_bounds__(64) k_bool_chain(uint32_t nchains, const uint32_t *seg_start, const uint32_t *maps, BoolPiece *pieces, uint64_t *chain_bits) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    uint32_t r = 254;
    uint64_t bits = 0;
    for (uint32_t seg = seg_start[c]; seg < seg_start[c + 1]; seg++) {
        pieces[seg].range = r; pieces[seg].bits0 = bits;
        const uint32_t m = maps[size_t(seg) * 128 + (r - 127)];
        r = m & 255u; bits += m >> 8;
    }
    chain_bits[2 * c] = bits; chain_bits[2 * c + 1] = r;
}
struct BoolEncLane {   // the boolean coder (oracle: boolenc) with per-lane state.  Sixty-four of these run side by side in a wave, every lane at its own place, and whatever one lane
                       // branches into, the whole wave executes.  So: a decision is straight-line arithmetic (the renormalisation shifts by 0 when none is due); the bits a
                       // decision completes stay in a 64-bit accumulator and leave on a FIXED schedule (flush() after every fourth decision: every lane emits its zero to
                       // four complete bytes then -- the same digits the byte-at-a-time coder produces, a carry that it would have patched into a written byte is simply
                       // added before the byte is written); and the output never READS (a carry that does reach a written byte goes into the last one, which the lane
                       // still holds in a register; one that would reach in front of the piece's first byte is left to k_bool_merge)
    uint8_t *buf;
    uint32_t pos, first, cap;
    int32_t range;
    uint64_t value;
    int run, nb_bits;
    uint32_t last, carry_front;
    bool overflow;
    __device__ __forceinline__ void init(uint8_t *b, uint32_t c, int r, int nb, uint32_t at) { buf = b; pos = at; first = at; cap = c; run = 0; nb_bits = nb; overflow = false; range = r; value = 0; last = 0; carry_front = 0; }
    __device__ __forceinline__ void emit(uint32_t bits) {   // eight bits and a carry
        if ((bits & 0xffu) != 0xffu) {
            if (pos + uint32_t(run) + 1 > cap) { overflow = true; run = 0; nb_bits = -8; value = 0; return; }
            if (bits & 0x100u) { if (pos > first) { last = (last + 1u) & 0xffu; buf[pos - 1] = uint8_t(last); } else carry_front = 1; }   // (the byte in front of a run of 0xff is never 0xff itself)
            const uint8_t v = (bits & 0x100u) ? 0x00 : 0xff;
            for (int k = 0; k < run; k++) buf[pos + uint32_t(k)] = v;
            last = bits & 0xffu;
            buf[pos + uint32_t(run)] = uint8_t(last);
            pos += uint32_t(run) + 1; run = 0;
        } else
            run++;
    }
    __device__ __forceinline__ void flush() {
        while (nb_bits > 0 && !overflow) {
            const int s = 8 + nb_bits;
            const uint32_t bits = uint32_t(value >> s);
            value -= uint64_t(bits) << s;
            nb_bits -= 8;
            emit(bits);
        }
    }
    __device__ __forceinline__ void put(int bit, int prob) {   // at most four of these between two flush(): 8 + 8 + 4 x 7 + 1 bits fit the accumulator many times over
        const int32_t split = (range * prob) >> 8, m = -int32_t(bit != 0);
        value += uint64_t(uint32_t((split + 1) & m));
        range = split + ((range - 2 * split - 1) & m);               // bit ? range - split - 1 : split
        const int shift = __clz(uint32_t(range + 1)) - 24;          // 0 for range >= 127: no renormalisation due
        range = ((range + 1) << shift) - 1;
        value <<= shift;
        nb_bits += shift;
    }
    __device__ __forceinline__ void settle() {   // the 0xff bytes held back for a carry that did not come inside this piece: written as they are
        if (pos + uint32_t(run) > cap) { overflow = true; run = 0; return; }
        for (int k = 0; k < run; k++) buf[pos + uint32_t(k)] = 0xff;
        pos += uint32_t(run); run = 0;
    }
};
__global__ void __launch_bounds__(64) k_bool_code(const WebpImg *imgs, const WebpChain *chains, uint32_t nchains, const uint32_t *seg_start, const uint64_t *mb_off, const uint16_t *stream,
                                                  uint8_t *scratch, BoolPiece *pieces, uint32_t *part_size, const uint32_t *status, uint32_t seglen) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= seg_start[nchains]) return;
    const uint32_t c = chain_of(seg_start, nchains, seg);
    const WebpChain ch = chains[c];
    const WebpImg &im = imgs[ch.image];
    if (status[im.image]) return;
    const bool header = ch.part == 0xFFFFFFFFu;
    uint8_t *base = header ? scratch + im.out_off : scratch + im.out_off + webp_hdr_cap(im);
    const uint32_t cap = header ? webp_hdr_cap(im) : webp_part_cap(im);
    const uint64_t B0 = pieces[seg].bits0;
    BoolEncLane e;
    e.init(base, cap, int(pieces[seg].range), B0 ? nb_after(B0) : -8, bytes_after(B0));
    const uint64_t d0 = mb_off[ch.first] + uint64_t(seg - seg_start[c]) * seglen, dend = mb_off[ch.first + ch.nmb], d1 = d0 + seglen < dend ? d0 + seglen : dend;
    // eight decisions to a 16-byte load, the next load in flight while these are coded (a lane's loads are its own: nothing hides their latency but this)
    uint64_t d = d0;
    for (; d < d1 && (d & 7u) && !e.overflow; d++) { const uint32_t v = stream[d]; e.put(int(v & 1u), int(v >> 1)); e.flush(); }
    if (d + 8 <= d1 && !e.overflow) {
        uint4 nxt = *reinterpret_cast<const uint4 *>(stream + d);
        for (; d + 8 <= d1 && !e.overflow; d += 8) {
            const uint4 cur = nxt;
            if (d + 16 <= d1) nxt = *reinterpret_cast<const uint4 *>(stream + d + 8);
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
            CSH_UNROLL
            for (int k = 0; k < 4; k++) {
                e.put(int(w[k] & 1u), int((w[k] >> 1) & 0x7FFFu));
                e.put(int((w[k] >> 16) & 1u), int(w[k] >> 17));
                if (k & 1) e.flush();
            }
        }
    }
    for (; d < d1 && !e.overflow; d++) { const uint32_t v = stream[d]; e.put(int(v & 1u), int(v >> 1)); e.flush(); }
    e.flush();
    e.settle();
    pieces[seg].carry_front = e.carry_front;
    pieces[seg].pend = uint32_t(e.value);
    if (e.overflow) part_size[size_t(im.image) * 2 + (header ? 0u : 1u)] = 0xFFFFFFFFu;
}
// one lane per chain.  Piece p held `pend` at its end: had piece p + 1 started with it instead of 0, its number would be larger by pend << (bits it shifts): that lands in the
// first two bytes piece p + 1 completed (it shifts at least 16 + nb bits when it completes two), or -- a piece that completed fewer -- partly in what it holds itself
__global__ void __launch_bounds__(64) k_bool_merge(const WebpImg *imgs, const WebpChain *chains, uint32_t nchains, const uint32_t *seg_start, const BoolPiece *pieces, const uint64_t *chain_bits,
                                                   uint8_t *scratch, uint32_t *part_size, const uint32_t *status) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    const WebpChain ch = chains[c];
    const WebpImg &im = imgs[ch.image];
    if (status[im.image]) return;
    const bool header = ch.part == 0xFFFFFFFFu;
    uint32_t *size_out = part_size + size_t(im.image) * 2 + (header ? 0u : 1u);
    if (*size_out == 0xFFFFFFFFu) return;   // a piece ran out of room
    uint8_t *buf = header ? scratch + im.out_off : scratch + im.out_off + webp_hdr_cap(im);
    const uint32_t cap = header ? webp_hdr_cap(im) : webp_part_cap(im);
    auto ripple = [&](uint32_t at_plus_1, uint32_t cy) {   // add cy to the byte in front of at_plus_1 and carry on backwards
        for (uint32_t q = at_plus_1; cy && q > 0; q--) { const uint32_t t = uint32_t(buf[q - 1]) + cy; buf[q - 1] = uint8_t(t); cy = t >> 8; }
    };
    uint64_t v = 0;
    const uint32_t s0 = seg_start[c], s1 = seg_start[c + 1];
    for (uint32_t seg = s0; seg < s1; seg++) {
        const uint64_t B0 = pieces[seg].bits0, B1 = seg + 1 < s1 ? pieces[seg + 1].bits0 : chain_bits[2 * c];
        const uint32_t E0 = bytes_after(B0), k = bytes_after(B1) - E0;
        const int nb0 = B0 ? nb_after(B0) : -8, w_end = 16 + (B1 ? nb_after(B1) : -8);
        if (pieces[seg].carry_front) ripple(E0, 1);
        const uint64_t pend = pieces[seg].pend;
        if (!v) { v = pend; continue; }
        if (k >= 2) {
            const uint64_t add = v << (-nb0);   // at most 17 bits: two bytes and a carry
            uint32_t t = uint32_t(buf[E0 + 1]) + uint32_t(add & 0xffu);
            buf[E0 + 1] = uint8_t(t);
            t = uint32_t(buf[E0]) + uint32_t((add >> 8) & 0xffu) + (t >> 8);
            buf[E0] = uint8_t(t);
            ripple(E0, (t >> 8) + uint32_t(add >> 16));
            v = pend;
        } else {
            const uint64_t total = (v << (B1 - B0)) + pend;
            if (k == 1) {
                const uint64_t em = total >> w_end;
                const uint32_t t = uint32_t(buf[E0]) + uint32_t(em & 0xffu);
                buf[E0] = uint8_t(t);
                ripple(E0, (t >> 8) + uint32_t(em >> 8));
                v = total & ((1ull << w_end) - 1ull);
            } else
                v = total;
        }
    }
    // the coder's closing bits from its real state (oracle: be_finish): 9 - nb_bits zero bits at even odds, then the last byte
    const uint64_t S = chain_bits[2 * c];
    uint32_t pos = bytes_after(S);
    int32_t range = int32_t(chain_bits[2 * c + 1]), nb_bits = S ? nb_after(S) : -8;
    uint64_t value = v;
    bool overflow = false;
    auto out_byte = [&](uint32_t bits) {
        if (pos + 1 > cap) { overflow = true; return; }
        buf[pos] = uint8_t(bits);
        ripple(pos, bits >> 8);
        pos++;
    };
    for (int n = 9 - nb_bits; n > 0; n--) {
        const int32_t split = range >> 1;
        range = split;
        if (range < 127) { const int shift = __clz(uint32_t(range + 1)) - 24; range = ((range + 1) << shift) - 1; value <<= shift; nb_bits += shift; }
        if (nb_bits > 0) { const int s = 8 + nb_bits; const uint32_t bits = uint32_t(value >> s); value -= uint64_t(bits) << s; nb_bits -= 8; out_byte(bits); }
    }
    { const uint32_t bits = uint32_t(value >> 8); out_byte(bits); }
    *size_out = overflow ? 0xFFFFFFFFu : pos;
}
// where every chain's decisions begin and end (for the host: it lays the pieces out)
__global__ void k_chain_bounds(const WebpChain *chains, uint32_t nchains, const uint64_t *mb_off, uint64_t *bounds) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchains) { bounds[2 * c] = mb_off[chains[c].first]; bounds[2 * c + 1] = mb_off[chains[c].first + chains[c].nmb]; }
}
// the file: RIFF / WEBP / "VP8 " headers (the chunk size counts the padding byte, as libwebp writes it), frame tag, start code, dimensions, the two partitions
__global__ void __launch_bounds__(256) k_webp_assemble(const WebpImg *imgs, const uint8_t *scratch, const uint32_t *part_size, uint8_t *out, uint32_t *img_size, uint32_t *status) {
    const WebpImg im = imgs[blockIdx.x];
    if (status[im.image]) return;
    const uint32_t p0 = part_size[size_t(im.image) * 2], p1 = part_size[size_t(im.image) * 2 + 1];
    const bool bad = p0 == 0xFFFFFFFFu || p1 == 0xFFFFFFFFu;
    const uint32_t raw = 10u + p0 + p1, vp8 = raw + (raw & 1u), total = 20u + vp8;
    if (bad || total > im.out_cap || p0 >= (1u << 19)) { if (threadIdx.x == 0) status[im.image] = 20200; return; }
    uint8_t *o = out + im.out_off;
    const uint8_t *base = scratch + im.out_off;
    if (threadIdx.x == 0) {
        const uint8_t hdr[20] = {'R', 'I', 'F', 'F', uint8_t(total - 8), uint8_t((total - 8) >> 8), uint8_t((total - 8) >> 16), uint8_t((total - 8) >> 24), 'W', 'E', 'B', 'P',
                                 'V', 'P', '8', ' ', uint8_t(vp8), uint8_t(vp8 >> 8), uint8_t(vp8 >> 16), uint8_t(vp8 >> 24)};
        for (int k = 0; k < 20; k++) o[k] = hdr[k];
        const uint32_t tag = (p0 << 5) | (1u << 4);   // key frame, profile 0 (normal loop filter), shown
        o[20] = uint8_t(tag); o[21] = uint8_t(tag >> 8); o[22] = uint8_t(tag >> 16);
        o[23] = 0x9D; o[24] = 0x01; o[25] = 0x2A;
        o[26] = uint8_t(im.width); o[27] = uint8_t(im.width >> 8); o[28] = uint8_t(im.height); o[29] = uint8_t(im.height >> 8);
        if (raw & 1u) o[20 + raw] = 0;
        img_size[im.image] = total;
    }
    for (uint32_t i = threadIdx.x; i < p0; i += blockDim.x) o[30 + i] = base[i];
    const uint8_t *src = base + webp_hdr_cap(im);
    for (uint32_t i = threadIdx.x; i < p1; i += blockDim.x) o[30 + p0 + i] = src[i];
}

void launch_webp_yuv(hipStream_t st, const WebpImg *imgs, int nimg, uint32_t max_luma, const uint8_t *rgb, uint8_t *work) {
    if (nimg && max_luma) CSH_LAUNCH(k_webp_yuv, dim3((max_luma + 255) / 256, nimg), dim3(256), st, imgs, rgb, work);
}
// base[i]: picture i's first macroblock among the batch's (host copy of d_base); d_cnt holds the macroblocks' decision counts (k_vp8_loop: chunk_stats) and has room for
// the header items behind them
int launch_webp_backend(hipStream_t st, const WebpImg *imgs, const WebpImg *himgs, int nimg, const int16_t *levels, const Vp8FrameDev *frames, const std::vector<uint64_t> &base,
                        const uint64_t *d_base, csh::DevBuf<uint32_t> &d_cnt, const uint16_t *d_blk, uint8_t *scratch, uint32_t *part_size, uint8_t *out, uint32_t *img_size, uint32_t *status) {
    std::vector<uint64_t> hbase(size_t(nimg) + 1);
    std::vector<WebpChain> chains;
    const uint64_t nmb = base[size_t(nimg)];
    uint32_t max_items = 0, max_mbh = 0;
    uint64_t nall = nmb;
    for (int i = 0; i < nimg; i++) {
        const uint64_t n = uint64_t(himgs[i].mbw) * himgs[i].mbh;
        chains.push_back(WebpChain{uint32_t(i), 0u, base[size_t(i)], n});
        hbase[size_t(i)] = nall;
        chains.push_back(WebpChain{uint32_t(i), 0xFFFFFFFFu, nall, n + 1});
        nall += n + 1;
        max_items = std::max(max_items, uint32_t(n) + 1u);
        max_mbh = std::max(max_mbh, himgs[i].mbh);
    }
    hbase[size_t(nimg)] = nall;
    csh::DevBuf<uint64_t> d_hbase, d_off;
    csh::DevBuf<uint16_t> d_stream;
    csh::DevBuf<WebpChain> d_chains;
    csh::DevBuf<uint8_t> d_tmp;
    const size_t tmp_bytes = csh::exclusive_scan_tmp_bytes(nall);
    if (d_cnt.n < nall + 1) { csh_set_error("webp: decision counters too small"); return -1; }
    if (d_hbase.upload(hbase, st) || d_chains.upload(chains, st) || d_off.alloc(nall + 2) || d_tmp.alloc(tmp_bytes + 64)) return -1;
    const dim3 items((max_items + 255) / 256, unsigned(nimg));
    CSH_LAUNCH(k_webp_hdr<false>, items, dim3(256), st, imgs, levels, frames, d_hbase.p, d_cnt.p, d_off.p, d_stream.p, status);
    csh::launch_exclusive_scan(st, d_cnt.p, d_off.p, nall, d_tmp.p, tmp_bytes + 64);
    const uint32_t nchains = uint32_t(chains.size());
    csh::DevBuf<uint64_t> d_bounds, d_cbits;
    if (d_bounds.alloc(2 * size_t(nchains) + 2) || d_cbits.alloc(2 * size_t(nchains) + 2)) return -1;
    CSH_LAUNCH(k_chain_bounds, dim3((nchains + 255) / 256), dim3(256), st, d_chains.p, nchains, d_off.p, d_bounds.p);
    std::vector<uint64_t> bounds(2 * size_t(nchains));
    if (csh_copy_wait(bounds.data(), d_bounds.p, bounds.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    const char *sl = getenv("CSH_TEST_BOOL_SEG");   // test hook: shorter pieces
    const uint32_t seglen = sl && atoi(sl) > 0 ? uint32_t(atoi(sl)) : uint32_t(BOOL_SEG);
    uint64_t total = 0;
    std::vector<uint32_t> seg_start(size_t(nchains) + 1, 0);
    for (uint32_t c = 0; c < nchains; c++) {
        total = std::max(total, bounds[2 * c + 1]);
        seg_start[c + 1] = seg_start[c] + uint32_t((bounds[2 * c + 1] - bounds[2 * c] + seglen - 1) / seglen);
    }
    const uint32_t nseg = seg_start[nchains];
    csh::DevBuf<uint32_t> d_seg_start, d_maps;
    csh::DevBuf<BoolPiece> d_pieces;
    if (d_stream.alloc(size_t(total) + 64) || d_seg_start.upload(seg_start, st) || d_maps.alloc(size_t(nseg) * 128 + 128) || d_pieces.alloc(size_t(nseg) + 1)) return -1;
    CSH_CHECK(hipMemsetAsync(part_size, 0, size_t(nimg) * 2 * sizeof(uint32_t), st));
    CSH_LAUNCH(k_webp_decisions, dim3(max_mbh, nimg), dim3(CSP_WAVE_THREADS), st, imgs, levels, frames, d_base, d_off.p, d_blk, d_stream.p, status);
    CSH_LAUNCH(k_webp_hdr<true>, items, dim3(256), st, imgs, levels, frames, d_hbase.p, d_cnt.p, d_off.p, d_stream.p, status);
    if (!nseg) { csh_set_error("webp: empty decision streams"); return -1; }
    {
        CSH_LAUNCH(k_bool_scan, dim3((nseg + 3) / 4), dim3(CSP_WAVE_THREADS), st, d_chains.p, nchains, d_seg_start.p, d_off.p, d_stream.p, d_maps.p, seglen);
        CSH_LAUNCH(k_bool_chain, dim3((nchains + 63) / 64), dim3(64), st, nchains, d_seg_start.p, d_maps.p, d_pieces.p, d_cbits.p);
        CSH_LAUNCH(k_bool_code, dim3((nseg + 63) / 64), dim3(64), st, imgs, d_chains.p, nchains, d_seg_start.p, d_off.p, d_stream.p, scratch, d_pieces.p, part_size, status, seglen);
    }
    CSH_LAUNCH(k_bool_merge, dim3((nchains + 63) / 64), dim3(64), st, imgs, d_chains.p, nchains, d_seg_start.p, d_pieces.p, d_cbits.p, scratch, part_size, status);
    CSH_LAUNCH(k_webp_assemble, dim3(nimg), dim3(256), st, imgs, scratch, part_size, out, img_size, status);
    CSH_CHECK(hipStreamSynchronize(st));
    return 0;
}

}  // namespace csw
