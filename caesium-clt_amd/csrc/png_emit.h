// png_emit.h -- one deflate block out of a chunk's tokens: what k_png_emit (greedy chunks, k_png_deflate.hip) and k_png_deep_emit (the chunks of the
// min-cost-path parse, k_png_parse.hip) share.  Statement: oracle/png_oracle.c deflate_chunk().
#pragma once
#include "png_kernels.h"
#include "png_lz.h"
#include "png_codes.h"

namespace csp {

__device__ static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ static PngChunk &chunk_rec(const DeflateCtx &c, const PngImg &im, int slot, uint32_t ci) {
    return c.chunks[uint64_t(im.chunk_base) + uint64_t(slot) * im.chunk_stride + ci];
}

// the tokens of a tile as bits: code | extra bits of the length, code | extra bits of the distance, or a literal's code
struct EmitSink {
    const uint32_t *code;   // code | length << 16, litlen then distance
    BitOut *bo;
    __device__ __forceinline__ void tile(uint64_t, uint32_t, uint64_t taken, const LV<uint32_t> &mlen, const LV<uint32_t> &mdist, const LV<uint32_t> &lit) {
        LV<uint64_t> val; LV<uint32_t> nb;
        LFOR(l) {
            val[l] = 0; nb[l] = 0;
            if ((taken >> l) & 1) {
                if (mlen[l]) {
                    const uint32_t lc = len_code_of(mlen[l]), dc = dist_code_of(mdist[l]);
                    const uint32_t cl = code[257 + lc], cd = code[CSP_NLIT + dc];
                    uint64_t v = cl & 0xFFFFu; uint32_t n = cl >> 16;
                    v |= uint64_t(mlen[l] - len_base_of(lc)) << n; n += len_extra_of(lc);
                    v |= uint64_t(cd & 0xFFFFu) << n; n += cd >> 16;
                    v |= uint64_t(mdist[l] - dist_base_of(dc)) << n; n += dist_extra_of(dc);
                    val[l] = v; nb[l] = n;
                } else { const uint32_t cl = code[lit[l]]; val[l] = cl & 0xFFFFu; nb[l] = cl >> 16; }
            }
        }
        bo->put(val, nb);
    }
};

// the chunk's codes into LDS (code | length << 16), the bit window cleared, the block header written
__device__ __forceinline__ static void emit_block_begin(const PngChunk &rec, bool last, uint32_t *code, uint32_t *win, uint8_t *out, BitOut &bo) {
    LFOR(l) {
        for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) code[i] = uint32_t(rec.code[i]) | (uint32_t(rec.len[i]) << 16);
        for (uint32_t i = uint32_t(l); i < 160; i += 64) win[i] = 0;
    }
    CSP_WAVE_SYNC();
    bo.win = win; bo.out = out; bo.bitpos = 0; bo.wbase = 0;
    LV<uint64_t> val; LV<uint32_t> nb;
    // block header: BFINAL, BTYPE=2, HLIT, HDIST, HCLEN (lane 0), the code-length code's lengths (lanes 1..)
    LFOR(l) {
        val[l] = 0; nb[l] = 0;
        if (l == 0) { val[l] = (last ? 1u : 0u) | (2u << 1) | (uint64_t(rec.hlit - 257) << 3) | (uint64_t(rec.hdist - 1) << 8) | (uint64_t(rec.hclen - 4) << 13); nb[l] = 17; }
        else if (l <= int(rec.hclen)) { val[l] = rec.cl_len[kClOrder[l - 1]]; nb[l] = 3; }
    }
    bo.put(val, nb);
    for (uint32_t h0 = 0; h0 < rec.nhdr; h0 += 64) {
        LFOR(l) {
            val[l] = 0; nb[l] = 0;
            const uint32_t h = h0 + uint32_t(l);
            if (h < rec.nhdr) {
                const uint32_t s = rec.hdr_sym[h], n = rec.cl_len[s];
                val[l] = uint64_t(rec.cl_code[s]) | (uint64_t(rec.hdr_extra[h]) << n);
                nb[l] = n + (s == 16 ? 2u : s == 17 ? 3u : s == 18 ? 7u : 0u);
            }
        }
        bo.put(val, nb);
    }
}
// end of block; then the sync marker (empty stored block) that byte-aligns every chunk but the last.  false: the size pass and this pass disagree
__device__ __forceinline__ static bool emit_block_end(const PngChunk &rec, bool last, const uint32_t *code, BitOut &bo) {
    LV<uint64_t> val; LV<uint32_t> nb;
    LFOR(l) {
        val[l] = 0; nb[l] = 0;
        if (l == 0) { val[l] = code[256] & 0xFFFFu; nb[l] = code[256] >> 16; }
        if (l == 1 && !last) nb[l] = 3;
    }
    bo.put(val, nb);
    if (!last) {
        const uint32_t pad = uint32_t((8 - (bo.bitpos & 7)) & 7);
        LFOR(l) { val[l] = 0; nb[l] = 0; if (l == 0) nb[l] = pad; if (l == 1) nb[l] = 16; if (l == 2) { val[l] = 0xFFFF; nb[l] = 16; } }
        bo.put(val, nb);
    }
    CSP_WAVE_SYNC();
    bo.finish();
    return bo.bitpos == (last ? rec.bits : ((rec.bits + 3 + 7) & ~7ull) + 32);
}

}  // namespace csp
