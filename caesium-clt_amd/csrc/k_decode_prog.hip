// k_decode_prog.hip -- phase 0 for PROGRESSIVE inputs: one wave per chain of scans.  Since round 4 this is the path behind the switches (CSH_PROG_PAR=0 / 1)
// and for files beyond the other paths' limits: DC first / AC first scans go through k_decode_par.hip, AC refinement chains through k_decode_refine.hip.
//
// A progressive scan cannot be cut into self-synchronising pieces the way a sequential scan can (k_decode_par.hip): in
// a refinement scan the number of correction bits between two Huffman symbols depends on which coefficients of the
// CURRENT block are already non-zero, so a decoder that does not know which block it is in cannot even find the next
// symbol.  What is independent is (a) the images of a batch and (b), inside an image, the chains of scans that touch
// disjoint coefficients: all DC scans, and the AC scans of each component (T.81 G.1.1.1: an AC scan has one component).
// One wave runs one chain, scans in file order:
//   * control flow is wave-uniform (one bit stream), the 64 lanes are the 64 zig-zag positions of the current block:
//     "which coefficients are non-zero" is one ballot, "skip r zero coefficients" a rank-select on that mask, and all
//     correction bits between two symbols are applied in one step (lane j takes bit rank_j of the group);
//   * the stream is the unstuffed copy made by the pre-pass of k_decode_par.hip, read 64 words (one per lane) at a time
//     and handed to the uniform bit reader by v_readlane; Huffman tables are the two-level LDS tables of ParHuffSet;
//   * a block's coefficients move as one 2-byte access per lane (eight 16-byte segments in the octet tile layout).
// The statement of T.81 G.2 / libjpeg jdphuff.c this must reproduce bit for bit is decode_block() in k_decode.hip, which
// stays the fallback for inputs this kernel does not take (restart markers, marker bytes inside a scan, oversized table
// sets).  Reference call site: /root/reference/src/compressor.rs:305.
//
// Emulation build: one thread plays the wave and the 64-lane values are arrays (VFOR loops); the decode logic is shared.
#include "kernels.h"

namespace csh {

#ifdef CSH_EMUL
#define VFOR(j) for (int j = 0; j < 64; j++)
#define VAT(x, j) ((x).v[j])
struct Vec64 { int v[64]; };
__device__ __forceinline__ static uint32_t uniform32(uint32_t x) { return x; }
#else
#define VFOR(j) for (int j = int(threadIdx.x & 63u), once_ = 1; once_; once_ = 0)
#define VAT(x, j) ((x).v)
struct Vec64 { int v; };
__device__ __forceinline__ static uint32_t uniform32(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }
#endif

template <class F>
__device__ __forceinline__ static uint64_t vballot(F pred) {
#ifdef CSH_EMUL
    uint64_t m = 0;
    for (int j = 0; j < 64; j++) m |= uint64_t(pred(j) ? 1 : 0) << j;
    return m;
#else
    return __ballot(pred(int(threadIdx.x & 63u)));
#endif
}
__device__ __forceinline__ static uint64_t below(int j) { return (1ull << j) - 1ull; }                                  // positions < j
__device__ __forceinline__ static uint64_t span(int lo, int hi) { return lo > hi ? 0ull : (~0ull >> (63 - hi)) & (~0ull << lo); }  // lo..hi inclusive, hi <= 63

// ---- wave-uniform bit reader over the unstuffed stream
struct WaveReader {
    const uint8_t *base;
    uint32_t len;       // bytes; zero bits beyond (as libjpeg feeds zeros after the data ends)
    uint32_t wbase;     // first word of the window
#ifdef CSH_EMUL
    uint32_t win[64], nxt[64];   // the emulation plays all 64 lanes of the window
#else
    uint32_t win, nxt;  // this lane's word of the current and of the next window (big-endian)
#endif
    uint64_t acc;
    int nb;
    uint32_t wi;        // next word to append
    uint64_t consumed;  // bits taken so far; past len*8 the rest of the MCU runs on zero bits and every later MCU is skipped
                        // (libjpeg jdphuff.c `insufficient_data`)
    __device__ __forceinline__ bool insufficient() const { return consumed > uint64_t(len) * 8u; }
    __device__ __forceinline__ uint32_t loadw(uint32_t w) const {
        uint32_t b = w * 4;
        if (b + 4 <= len) {
            uint32_t v = *reinterpret_cast<const uint32_t *>(base + b);
            return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
        }
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) v = (v << 8) | (b + i < len ? base[b + i] : 0u);
        return v;
    }
    __device__ __forceinline__ uint32_t word(uint32_t i) {
#ifdef CSH_EMUL
        if (i >= wbase + 64) { wbase += 64; for (int l = 0; l < 64; l++) { win[l] = nxt[l]; nxt[l] = loadw(wbase + 64 + l); } }
        if (i < wbase || i - wbase >= 64) { fprintf(stderr, "WaveReader: non-sequential access\n"); abort(); }
        return win[i - wbase];
#else
        if (i >= wbase + 64) { win = nxt; wbase += 64; nxt = loadw(wbase + 64 + (threadIdx.x & 63u)); }
        return uint32_t(__builtin_amdgcn_readlane(int(win), int(i - wbase)));
#endif
    }
    __device__ __forceinline__ void begin(const uint8_t *p, uint32_t n) {
        base = p; len = n; wbase = 0;
#ifndef CSH_EMUL
        win = loadw(threadIdx.x & 63u); nxt = loadw(64 + (threadIdx.x & 63u));
#else
        for (int l = 0; l < 64; l++) { win[l] = loadw(l); nxt[l] = loadw(64 + l); }
#endif
        acc = (uint64_t(word(0)) << 32) | word(1);
        nb = 64; wi = 2; consumed = 0;
    }
    __device__ __forceinline__ uint32_t peek16() const { return uint32_t(acc >> 48); }
    __device__ __forceinline__ void skip(int n) {   // n <= 32
        acc <<= n; nb -= n; consumed += uint32_t(n);
        if (nb < 32) { acc |= uint64_t(word(wi)) << (32 - nb); wi++; nb += 32; }
    }
    __device__ __forceinline__ uint32_t get(int n) {   // n <= 32
        if (n == 0) return 0;
        uint32_t v = uint32_t(acc >> (64 - n));
        skip(n);
        return v;
    }
    __device__ __forceinline__ uint64_t get64(int n) {   // n <= 64
        if (n <= 32) return get(n);
        uint64_t hi = get(n - 32);
        return (hi << 32) | get(32);
    }
};

__device__ __forceinline__ static int prog_huff(WaveReader &rd, const ParHuffSet &hs, int tbl) {
    uint32_t top16 = rd.peek16();
    uint32_t e = hs.root[tbl][top16 >> 7];
    if (e & 0x8000u) e = hs.sub[(e & 0xFFFu) + ((top16 & 127u) >> (7u - ((e >> 12) & 7u)))];
    e = uniform32(e);
    rd.skip(e ? int(e >> 8) : 16);   // no such code: 16 bits, symbol 0 (k_decode.hip huff_decode)
    return int(e & 255u);
}
__device__ __forceinline__ static int prog_extend(int r, int n) { return r < (1 << (n - 1)) ? r - (1 << n) + 1 : r; }
// one Huffman symbol AND the raw bits that follow it (at most 15 + 16 <= 31 bits together) out of the same 32-bit window:
// one trip through the bit reader per symbol instead of two or three.  extra(sym) = number of raw bits that belong to it.
template <class F>
__device__ __forceinline__ static int prog_huff_with_bits(WaveReader &rd, const ParHuffSet &hs, int tbl, F extra, uint32_t &bits) {
    const uint32_t w = uint32_t(rd.acc >> 32), top16 = w >> 16;
    uint32_t e = hs.root[tbl][top16 >> 7];
    if (e & 0x8000u) e = hs.sub[(e & 0xFFFu) + ((top16 & 127u) >> (7u - ((e >> 12) & 7u)))];
    e = uniform32(e);
    const int len = e ? int(e >> 8) : 16, sym = int(e & 255u);
    const int n = extra(sym);
    bits = n ? (w << len) >> (32 - n) : 0u;
    rd.skip(len + n);
    return sym;
}

// ---- block <-> lanes
__device__ __forceinline__ static void load_block(Vec64 &c, const int16_t *blk) { VFOR(j) VAT(c, j) = blk[coef_off(j)]; }
__device__ __forceinline__ static void store_lanes(const Vec64 &c, int16_t *blk, uint64_t which) {
    VFOR(j) if ((which >> j) & 1) blk[coef_off(j)] = int16_t(VAT(c, j));
}
// every position of `Hc` consumes one correction bit, in position order (bits: n = popcount(Hc) bits, first bit highest)
__device__ __forceinline__ static void apply_corrections(Vec64 &c, uint64_t Hc, uint64_t bits, int n, int p1, uint64_t &dirty) {
    uint64_t changed = vballot([&](int j) {
        if (!((Hc >> j) & 1)) return false;
        int rank = __popcll(Hc & below(j));
        return ((bits >> (n - 1 - rank)) & 1) != 0 && (VAT(c, j) & p1) == 0;
    });
    VFOR(j) if ((changed >> j) & 1) VAT(c, j) += VAT(c, j) >= 0 ? p1 : -p1;
    dirty |= changed;
}
// position of the r-th (0-based) set bit of Z, or 64 if Z has no more than r bits
__device__ __forceinline__ static int select_bit(uint64_t Z, int r) {
    uint64_t hit = vballot([&](int j) { return ((Z >> j) & 1) && __popcll(Z & below(j)) == r; });
    return hit ? __ffsll((unsigned long long)hit) - 1 : 64;
}

// ---- one scan
struct ScanCtx {
    const ImgDesc *im;
    const DecScan *sc;
    const ParHuffSet *hs;
    int16_t *coef;
};

__device__ static void scan_dc_first(const ScanCtx &x, WaveReader &rd) {
    const DecScan &sc = *x.sc;
    const ImgDesc &im = *x.im;
    int pred[CSH_MAX_COMPS] = {0, 0, 0};
    auto one = [&](int ci, int by, int bx) {
        const CompGeom &g = im.in[sc.comp[ci]];
        uint32_t raw;
        int t = prog_huff_with_bits(rd, *x.hs, sc.td[ci] & 3, [](int s) { return s & 15; }, raw) & 15;   // DC categories are <= 15 (host-checked)
        int diff = t ? prog_extend(int(raw), t) : 0;
        pred[ci] += diff;
        int16_t *blk = x.coef + coef_index(g.tile_base, by * g.bw + bx, 0);
        const int v = pred[ci] * (1 << sc.Al);
        VFOR(j) if (j == 0) blk[0] = int16_t(v);
    };
    if (sc.ncomp == 1) {
        const CompGeom &g = im.in[sc.comp[0]];
        for (int by = 0; by < g.real_bh; by++) for (int bx = 0; bx < g.real_bw; bx++) { if (rd.insufficient()) return; one(0, by, bx); }
    } else {
        for (int my = 0; my < im.mcus_y; my++)
            for (int mx = 0; mx < im.mcus_x; mx++) {
                if (rd.insufficient()) return;
                for (int ci = 0; ci < sc.ncomp; ci++) {
                    const CompGeom &g = im.in[sc.comp[ci]];
                    for (int y = 0; y < g.v; y++) for (int xx = 0; xx < g.h; xx++) one(ci, my * g.v + y, mx * g.h + xx);
                }
            }
    }
}

// DC refinement: one raw bit per block, in scan order -- 64 blocks per step, lane j takes bit j of the group
__device__ static void scan_dc_refine(const ScanCtx &x, WaveReader &rd) {
    const DecScan &sc = *x.sc;
    const ImgDesc &im = *x.im;
    int nb_mcu = 0;
    for (int ci = 0; ci < sc.ncomp; ci++) nb_mcu += sc.ncomp > 1 ? im.in[sc.comp[ci]].h * im.in[sc.comp[ci]].v : 1;
    const uint32_t total = sc.ncomp > 1 ? uint32_t(im.mcus_x) * uint32_t(im.mcus_y) * uint32_t(nb_mcu)
                                        : uint32_t(im.in[sc.comp[0]].real_bw) * uint32_t(im.in[sc.comp[0]].real_bh);
    for (uint32_t base = 0; base < total; base += 64) {
        const int cnt = total - base < 64 ? int(total - base) : 64;
        const uint64_t bits = rd.get64(cnt);
        VFOR(j) {
            if (j >= cnt || !((bits >> (cnt - 1 - j)) & 1)) continue;
            uint32_t o = base + uint32_t(j);
            int16_t *blk;
            if (sc.ncomp == 1) {
                const CompGeom &g = im.in[sc.comp[0]];
                int by = int(o) / g.real_bw, bx = int(o) - by * g.real_bw;
                blk = x.coef + coef_index(g.tile_base, by * g.bw + bx, 0);
            } else {
                uint32_t mcu = o / uint32_t(nb_mcu);
                int m = int(o - mcu * uint32_t(nb_mcu)), my = int(mcu) / im.mcus_x, mx = int(mcu) - my * im.mcus_x, ci = 0;
                while (m >= im.in[sc.comp[ci]].h * im.in[sc.comp[ci]].v) { m -= im.in[sc.comp[ci]].h * im.in[sc.comp[ci]].v; ci++; }
                const CompGeom &g = im.in[sc.comp[ci]];
                int y = m / g.h, xx = m - y * g.h;
                blk = x.coef + coef_index(g.tile_base, (my * g.v + y) * g.bw + mx * g.h + xx, 0);
            }
            blk[0] = int16_t(blk[0] | (1 << sc.Al));
        }
    }
}

__device__ static void scan_ac_first(const ScanCtx &x, WaveReader &rd) {
    const DecScan &sc = *x.sc;
    const CompGeom &g = x.im->in[sc.comp[0]];
    const int tbl = 4 + (sc.ta[0] & 3);
    uint32_t eobrun = 0;
    const uint32_t total = uint32_t(g.real_bw) * uint32_t(g.real_bh);
    uint32_t o = 0;
    int by = 0, bx = 0;
    while (o < total) {
        if (rd.insufficient()) return;
        if (eobrun) {   // blocks inside an EOB run carry no bits in a first pass: jump over them
            uint32_t skip = eobrun < total - o ? eobrun : total - o;
            eobrun -= skip; o += skip;
            uint32_t col = uint32_t(bx) + skip;
            by += int(col / uint32_t(g.real_bw)); bx = int(col % uint32_t(g.real_bw));
            continue;
        }
        Vec64 c;
        VFOR(j) { (void)j; VAT(c, j) = 0; }
        uint64_t placed = 0;
        for (int k = sc.Ss; k <= sc.Se; k++) {
            uint32_t raw;
            // raw bits of a symbol: the coefficient's n value bits -- unless the run overflows the block (no bits are read then,
            // k_decode.hip) -- or the r low bits of an EOB run length
            const int kk = k;
            int rs = prog_huff_with_bits(rd, *x.hs, tbl, [kk](int s) { int r = s >> 4, n = s & 15; return n ? (kk + r > 63 ? 0 : n) : (r == 15 ? 0 : r); }, raw);
            int r = rs >> 4, n = rs & 15;
            if (n) {
                k += r;
                if (k > 63) break;
                const int v = prog_extend(int(raw), n) * (1 << sc.Al);
                VFOR(j) if (j == k) VAT(c, j) = v;
                placed |= 1ull << k;
            } else {
                if (r == 15) k += 15;
                else { eobrun = (1u << r) + raw - 1u; break; }
            }
        }
        if (placed) store_lanes(c, x.coef + coef_index(g.tile_base, by * g.bw + bx, 0), placed);
        o++;
        if (++bx == g.real_bw) { bx = 0; by++; }
    }
}

__device__ static void scan_ac_refine(const ScanCtx &x, WaveReader &rd) {
    const DecScan &sc = *x.sc;
    const CompGeom &g = x.im->in[sc.comp[0]];
    const int tbl = 4 + (sc.ta[0] & 3);
    const int p1 = 1 << sc.Al, m1 = -p1, Se = sc.Se;
    uint32_t eobrun = 0;
    Vec64 c, ahead;
    if (g.real_bw > 0 && g.real_bh > 0) load_block(ahead, x.coef + coef_index(g.tile_base, 0, 0));
    for (int by = 0; by < g.real_bh; by++)
        for (int bx = 0; bx < g.real_bw; bx++) {
            if (rd.insufficient()) return;
            int16_t *blk = x.coef + coef_index(g.tile_base, by * g.bw + bx, 0);
            c = ahead;
            {   // fetch the next block of the scan while this one is decoded
                int nbx = bx + 1, nby = by;
                if (nbx == g.real_bw) { nbx = 0; nby++; }
                if (nby < g.real_bh) load_block(ahead, x.coef + coef_index(g.tile_base, nby * g.bw + nbx, 0));
            }
            const uint64_t H = vballot([&](int j) { return VAT(c, j) != 0; });   // history: what earlier scans made non-zero
            uint64_t dirty = 0;
            int k = sc.Ss;
            if (eobrun == 0) {
                while (k <= Se) {
                    uint32_t raw;
                    int rs = prog_huff_with_bits(rd, *x.hs, tbl, [](int s) { int r = s >> 4, n = s & 15; return n ? 1 : (r == 15 ? 0 : r); }, raw);   // sign bit, or EOB run bits
                    int r = rs >> 4, n = rs & 15, val = 0;
                    if (n) val = raw ? p1 : m1;
                    else if (r != 15) { eobrun = (1u << r) + raw; break; }
                    // pass over coefficients until r zero-history positions are skipped and the next one is reached; every
                    // non-zero-history position on the way takes a correction bit
                    const uint64_t Z = ~H & span(k, Se);
                    int kz = select_bit(Z, r);
                    if (kz > Se) kz = Se + 1;
                    const uint64_t Hc = H & span(k, kz - 1);
                    const int nc = __popcll(Hc);
                    if (nc) apply_corrections(c, Hc, rd.get64(nc), nc, p1, dirty);
                    k = kz;
                    if (val && k <= 63) { VFOR(j) if (j == k) VAT(c, j) = val; dirty |= 1ull << k; }
                    k++;
                }
            }
            if (eobrun > 0) {
                const uint64_t Hc = H & span(k, Se);
                const int nc = __popcll(Hc);
                if (nc) apply_corrections(c, Hc, rd.get64(nc), nc, p1, dirty);
                eobrun--;
            }
            if (dirty) store_lanes(c, blk, dirty);
        }
}

// ---- kernel: one wave (a 64-thread workgroup) per chain
__global__ void __launch_bounds__(64) k_decode_prog(const uint8_t *clean, const ParScan *pss, const ParHuffSet *huffs, const DecScan *scans, const ProgChain *chains,
                                                    const int *chain_scans, int nchains, const ImgDesc *imgs, int16_t *coef, const uint32_t *need_seq) {
    CSH_SHARED ParHuffSet lhs;
    const int ch = blockIdx.x;
    if (ch >= nchains) return;
    const ProgChain pc = chains[ch];
    if (pc.refine || need_seq[pc.image] != 4) return;   // refinement chains: k_decode_refine.hip
    int cur_set = -1;
    for (int s = 0; s < pc.count; s++) {
        const DecScan &sc = scans[chain_scans[pc.first + s]];
        const ParScan &ps = pss[sc.par_index];
        if (sc.huff_set != cur_set) {   // the scan's tables into LDS (one wave: no other wave shares this copy)
#ifndef CSH_EMUL
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            const uint4 *src = reinterpret_cast<const uint4 *>(&huffs[sc.huff_set]);
            uint4 *dst = reinterpret_cast<uint4 *>(&lhs);
            for (uint32_t i = threadIdx.x; i < sizeof(ParHuffSet) / 16; i += 64) dst[i] = src[i];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#else
            lhs = huffs[sc.huff_set];
#endif
            cur_set = sc.huff_set;
        }
        WaveReader rd;
        rd.begin(clean + ps.bits_off, ps.clean_len);
        ScanCtx x; x.im = &imgs[pc.image]; x.sc = &sc; x.hs = &lhs; x.coef = coef;
        if (sc.Ss == 0) { if (sc.Ah == 0) scan_dc_first(x, rd); else scan_dc_refine(x, rd); }
        else if (sc.Ah == 0) scan_ac_first(x, rd);
        else scan_ac_refine(x, rd);
    }
}

void launch_decode_prog(hipStream_t st, const uint8_t *clean, const ParScan *pss, const ParHuffSet *huffs, const DecScan *scans, const ProgChain *chains,
                        const int *chain_scans, int nchains, const ImgDesc *imgs, int16_t *coef, uint32_t *need_seq) {
    if (!nchains) return;
#ifdef CSH_EMUL
    CSH_LAUNCH(k_decode_prog, dim3(nchains), dim3(1), st, clean, pss, huffs, scans, chains, chain_scans, nchains, imgs, coef, need_seq);
#else
    CSH_LAUNCH(k_decode_prog, dim3(nchains), dim3(64), st, clean, pss, huffs, scans, chains, chain_scans, nchains, imgs, coef, need_seq);
#endif
}

}  // namespace csh
