// k_decode_refine.hip -- AC REFINEMENT scans of progressive inputs: a serial parse that only finds where every block starts, then a parallel apply.
//
// A refinement scan is the one kind of scan a decoder cannot enter in the middle (k_decode_prog.hip's header: the number of correction bits between
// two Huffman symbols depends on which coefficients of the CURRENT block are non-zero already).  What a decoder that walks the scan from its start
// needs of a block, though, is very little: the block's 64-bit HISTORY mask (which coefficients earlier scans made non-zero).  With it the walk is
// arithmetic on masks -- "skip r zero-history positions" = clear r low bits of ~H, "correction bits on the way" = a population count -- and never touches
// a coefficient.  So the chain of a component's refinement scans is three kernels:
//   k_refine_hist   one thread per block: the history mask in front of the chain's first scan (from the coefficients the first scans left);
//   k_refine_parse  one WAVE per scan, wave-uniform control flow on the scalar unit.  The lanes decode the symbol that would start at each of 64 consecutive
//                   bit positions (what a symbol is does not depend on the decoder's state); the walk then costs one v_readlane, the mask arithmetic and an
//                   addition per symbol.  It writes every block's bit position (and whether the block starts inside an EOB
//                   run) and rewrites the masks for the chain's next scan (a coefficient a scan makes non-zero is history for the next one) -- whose wave
//                   follows one window of 64 blocks behind: the scans of a chain are parsed side by side and nothing waits for anything to be applied.
//                   Blocks inside an EOB run take 64 per step (one population count per lane, a prefix sum).  This is the serial part: ~10^2 scalar
//                   instructions per symbol where k_decode_prog spent ~10^3;
//   k_refine_apply  one thread per block, the chain's scans in file order: decode the block's symbols from its own bit position and change its
//                   coefficients (T.81 G.1.2.3 / libjpeg jdphuff.c decode_mcu_AC_refine; the statement this must reproduce bit for bit is
//                   scan_ac_refine() in k_decode_prog.hip, which stays for the chains this path does not take).
// Bit positions are 30-bit (the host keeps scans of >= 2^27 bytes on the old chains).  A stream that ends early: libjpeg runs the block it ends in on
// zero bits and skips every later block; here a block whose position lies beyond the data is marked skipped.
// Reference call site: /root/reference/src/compressor.rs:305.
#include "kernels.h"
#include "wave.h"

namespace csh {

#define CSH_RF_SKIP 0xFFFFFFFFu
#define CSH_RF_INRUN 0x80000000u

__device__ __forceinline__ static uint64_t rf_span(int lo, int hi) { return lo > hi ? 0ull : (~0ull >> (63 - hi)) & (~0ull << lo); }   // lo..hi inclusive, hi <= 63
// big-endian word w of a stream of `len` bytes, zero beyond its end.  No branch (a divergent branch inside the parse loop would have the compiler
// structurise the wave-uniform control flow around it): the load is always in bounds -- a scan's bytes start 64-byte aligned in a pool with 64 bytes of
// slack, a word that lies beyond the stream is loaded from offset 0 and masked away
__device__ __forceinline__ static uint32_t rf_loadw(const uint8_t *base, uint32_t len, uint32_t w) {
    const uint32_t b = w * 4u;
    const uint32_t keep = b < len ? len - b : 0u;   // bytes of the word that belong to the stream
    const uint32_t v = *reinterpret_cast<const uint32_t *>(base + (keep ? b : 0u));
    const uint32_t be = (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
    const uint32_t mask = keep >= 4u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (8u * keep));   // keep = 0: no bit
    return be & mask;
}
__device__ __forceinline__ static uint32_t rf_popc64(uint64_t m) { return uint32_t(__popcll((unsigned long long)m)); }
// position of the r-th (0-based) set bit of Z; 64 if there is none
__device__ __forceinline__ static int rf_select(uint64_t Z, int r) {
    for (int i = 0; i < r && Z; i++) Z &= Z - 1;
    return Z ? __ffsll((unsigned long long)Z) - 1 : 64;
}

// ---------------------------------------------------------------------------------------------------------------- history masks
__device__ __forceinline__ static uint64_t rf_block_hist(const int16_t *blk) {
    uint64_t H = 0;
    for (int oct = 0; oct < 8; oct++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(blk + oct * CSH_OCT_STRIDE);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 4; i++) {
            if (w[i] & 0xFFFFu) H |= 1ull << (oct * 8 + 2 * i);
            if (w[i] >> 16) H |= 1ull << (oct * 8 + 2 * i + 1);
        }
    }
    return H;
}
__global__ void __launch_bounds__(256) k_refine_hist(const ProgChain *chains, int nchains, const ImgDesc *imgs, const int16_t *coef, const uint32_t *need_seq,
                                                     uint64_t *hist) {
    const int ch = blockIdx.y;
    if (ch >= nchains) return;
    const ProgChain pc = chains[ch];
    if (!pc.refine || need_seq[pc.image] != 4) return;
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    if (o >= pc.nblocks) return;
    const CompGeom &g = imgs[pc.image].in[pc.comp];
    const int by = int(o / uint32_t(g.real_bw)), bx = int(o - uint32_t(by) * uint32_t(g.real_bw));
    const int16_t *blk = coef + coef_index(g.tile_base, by * g.bw + bx, 0);
    const uint64_t H = rf_block_hist(blk);
    hist[size_t(pc.hist_off) + o] = H;
}

// ---------------------------------------------------------------------------------------------------------------- the parse
// the stream as a window of 64 words in a VGPR (lane l = word wbase + l) and the next one behind it, 48 words on: the words around any bit position are
// a few v_readlane away
struct RfWindow {
    const uint8_t *base;
    uint32_t len, wbase;
    LV<uint32_t> win, nxt;
    __device__ __forceinline__ void fill(LV<uint32_t> &x, uint32_t w0) { LFOR(l) x[l] = rf_loadw(base, len, w0 + uint32_t(l)); }
    __device__ __forceinline__ void begin(const uint8_t *p, uint32_t n) { base = p; len = n; wbase = 0; fill(win, 0); fill(nxt, 48); }
    // index of word wi in `win` (words wi .. wi + 3 are in it); bit positions only grow
    __device__ __forceinline__ uint32_t seek(uint32_t wi) {
        while (wi - wbase >= 48u) {
            if (wi - wbase < 96u) { LFOR(l) win[l] = nxt[l]; wbase += 48u; }
            else { wbase = wi; fill(win, wbase); }
            fill(nxt, wbase + 48u);
        }
        return wi - wbase;
    }
};

// The walk's straight line: symbols that are a new coefficient behind r zero-history positions, one after the other, until the block's band is finished
// (returns 0), the next symbol is not of that kind or lies outside the 64 decoded positions (returns 1: nothing of it consumed), or the band has fewer
// zero-history positions than the run wants (returns 2: the symbol is in `e`, its bits not yet counted, Z is empty).  State as in k_refine_parse: Z / Hr the
// band's zero- / non-zero-history positions still ahead, Nb the coefficients this scan makes non-zero.  A lone wave pays ~10 cycles per scalar instruction and
// more per taken branch, and the compiler structurises the wave-uniform loop around the kernel's per-lane code (flag registers, four taken branches per
// symbol): so the device form is written out -- 23 instructions and one taken branch per symbol; the emulation runs the statement it stands for.
__device__ __forceinline__ static uint32_t rf_fast_symbols(uint32_t &pos, uint64_t &Z, uint64_t &Hr, uint64_t &Nb, uint32_t &e, uint32_t ebase, const LV<uint32_t> &E) {
#ifdef CSH_EMUL
    for (;;) {
        const uint32_t idx = pos - ebase;
        if (idx >= 64u) return 1;
        e = E.v[idx];
        if (e & 0xC00u) return 1;
        for (uint32_t r = (e >> 5) & 15u; r; r--) Z &= Z - 1;
        if (!Z) return 2;
        const uint64_t low = Z & (0ull - Z), below = low - 1;   // the position reached, the positions in front of it
        pos += (e & 31u) + rf_popc64(Hr & below);
        Nb |= low;
        Hr &= ~below;
        Z &= ~low;
        if (!(Z | Hr)) return 0;
    }
#else
    uint32_t why, idx, r, k, c32;
    uint64_t below, low;
    asm volatile(
        ".Lrf_loop_%=:\n\t"
        "s_sub_u32 %[idx], %[pos], %[ebase]\n\t"
        "s_cmp_ge_u32 %[idx], 64\n\t"
        "s_cbranch_scc1 .Lrf_other_%=\n\t"
        "v_readlane_b32 %[e], %[E], %[idx]\n\t"
        "s_and_b32 %[r], %[e], 0xc00\n\t"            // EOBn, ZRL: not here
        "s_cbranch_scc1 .Lrf_other_%=\n\t"
        "s_bfe_u32 %[r], %[e], 0x40005\n\t"          // the run
        "s_cbranch_scc1 .Lrf_run_%=\n"
        ".Lrf_land_%=:\n\t"
        "s_ff1_i32_b64 %[k], %[Z]\n\t"               // the position reached (-1: none left)
        "s_cmp_lt_i32 %[k], 0\n\t"
        "s_cbranch_scc1 .Lrf_short_%=\n\t"
        "s_and_b32 %[c32], %[e], 31\n\t"
        "s_add_u32 %[pos], %[pos], %[c32]\n\t"
        "s_bfm_b64 %[below], %[k], 0\n\t"            // the positions in front of it
        "s_lshl_b64 %[low], 1, %[k]\n\t"
        "s_and_b64 %[below], %[Hr], %[below]\n\t"    // ... with history: one correction bit each
        "s_bcnt1_i32_b64 %[c32], %[below]\n\t"
        "s_add_u32 %[pos], %[pos], %[c32]\n\t"
        "s_or_b64 %[Nb], %[Nb], %[low]\n\t"
        "s_andn2_b64 %[Hr], %[Hr], %[below]\n\t"
        "s_andn2_b64 %[Z], %[Z], %[low]\n\t"
        "s_or_b64 %[low], %[Z], %[Hr]\n\t"           // anything of the band left?
        "s_cbranch_scc1 .Lrf_loop_%=\n\t"
        "s_mov_b32 %[why], 0\n\t"
        "s_branch .Lrf_end_%=\n"
        ".Lrf_run_%=:\n\t"                            // r zero-history positions are passed
        "s_ff1_i32_b64 %[k], %[Z]\n\t"
        "s_bitset0_b64 %[Z], %[k]\n\t"               // (Z empty: k = -1, bit 63 of nothing)
        "s_add_u32 %[r], %[r], -1\n\t"
        "s_cmp_lg_u32 %[r], 0\n\t"
        "s_cbranch_scc1 .Lrf_run_%=\n\t"
        "s_branch .Lrf_land_%=\n"
        ".Lrf_other_%=:\n\t"
        "s_mov_b32 %[why], 1\n\t"
        "s_branch .Lrf_end_%=\n"
        ".Lrf_short_%=:\n\t"
        "s_mov_b32 %[why], 2\n"
        ".Lrf_end_%=:\n\t"
        : [pos] "+s"(pos), [Z] "+s"(Z), [Hr] "+s"(Hr), [Nb] "+s"(Nb), [e] "=&s"(e), [why] "=&s"(why), [idx] "=&s"(idx), [r] "=&s"(r), [k] "=&s"(k),
          [c32] "=&s"(c32), [below] "=&s"(below), [low] "=&s"(low)
        : [ebase] "s"(ebase), [E] "v"(E.v)
        : "scc");
    return why;
#endif
}

// One wave per (chain, scan) UNIT.  The scans of a chain are parsed side by side, one window of 64 blocks behind each other: a unit publishes how many blocks'
// masks it has rewritten (prog[unit], after a release fence) and the unit of the chain's next scan waits for that count before it loads a window.  Units are
// numbered scan-major (a chain's scan s before any chain's scan s + 1) and a wave takes its unit from a ticket counter, so the unit a wave waits for belongs
// to a wave that started earlier: nothing waits for a wave that is not running yet, whatever order the workgroups are dispatched in.
__global__ void __launch_bounds__(64) k_refine_parse(const uint8_t *clean, const ParScan *pss, const ParHuffSet *huffs, const DecScan *scans, const ProgChain *chains,
                                                     const int *chain_scans, const RefineUnit *units, int nunits, uint32_t *need_seq, uint64_t *hist, uint32_t *posv,
                                                     uint32_t *prog) {
    LV<uint32_t> tv;
    LFOR(l) tv[l] = l == 0 ? atomicAdd(&prog[nunits], 1u) : 0u;
    const uint32_t unit = lget(tv, 0);
    if (unit >= uint32_t(nunits)) return;
    const RefineUnit u = units[unit];
    const ProgChain pc = chains[u.chain];
    uint64_t *H = hist + pc.hist_off;
    auto wait_for = [&](uint32_t upto) {   // the previous scan of the chain has rewritten the masks of blocks < upto
        if (u.prev < 0) return;
        while (coherent_load(&prog[u.prev]) < upto) {
#ifdef CSH_EMUL
            fprintf(stderr, "k_refine_parse: unit %u waits for unit %d, which has not run\n", unit, u.prev); abort();
#else
            __builtin_amdgcn_s_sleep(8);
#endif
        }
        CSP_ACQUIRE_FENCE();
    };
    auto publish = [&](uint32_t upto) {
        CSP_MEM_FENCE();
        LFOR(l) if (l == 0) coherent_store(&prog[unit], upto);
    };
    // need_seq changes INSIDE this kernel (a damaged run that leaves its band hands the image to the sequential decoder, below): a unit of the chain's
    // next scan may have read 4 before that store landed and be waiting for this one -- so a unit that steps aside says "all blocks done" first
    // (the image is decoded again by k_decode_seq, whatever the waiting unit then parses from the masks is thrown away)
    if (coherent_load(&need_seq[pc.image]) != 4u) { publish(pc.nblocks); return; }
    {
        const int s = u.s;
        const DecScan &sc = scans[chain_scans[pc.first + s]];
        const ParScan &ps = pss[sc.par_index];
        const ParHuffSet &hs = huffs[sc.huff_set];
        const int tbl = 4 + (sc.ta[0] & 3), Ss = sc.Ss, Se = sc.Se;
        const uint64_t band = rf_span(Ss, Se), beyond = Se + 1 <= 63 ? 1ull << (Se + 1) : 0ull;
        uint32_t *rec_out = posv + size_t(pc.pos_off) + size_t(s) * pc.nblocks;
        // the scan's AC table in LDS (this wave's own copy): root[top 9 bits], sub[] behind it (types.h ParHuffSet)
        CSH_SHARED uint16_t l_root[512];
        CSH_SHARED uint16_t l_sub[CSH_PAR_SUB];
        LFOR(l) {
            for (int i = l; i < 512; i += 64) l_root[i] = hs.root[tbl][i];
            for (int i = l; i < CSH_PAR_SUB; i += 64) l_sub[i] = hs.sub[i];
        }
        CSP_WAVE_SYNC();
        RfWindow rd;
        rd.begin(clean + ps.bits_off, ps.clean_len);
        const uint32_t len8 = ps.clean_len * 8u;
        uint32_t pos = 0, eobrun = 0;
        bool dead = false;
        // What a Huffman symbol at a given bit position IS does not depend on the decoder's state (only how many correction bits follow it does): the 64
        // lanes decode the symbol that WOULD start at each of 64 consecutive bit positions, one entry each --
        //   [4:0] bits of the symbol and its own raw bits (sign bit / EOB run length)   [8:5] run r   [10] EOBn   [11] ZRL   [31:16] EOB run
        // -- and the serial walk below is one v_readlane per symbol instead of a bit-window extraction and a table look-up.
        LV<uint32_t> E;
        uint32_t ebase = 0x80000000u;   // bit position of lane 0's entry; none yet (positions are below 2^30: pos - ebase >= 64)
        auto symbols_at = [&](uint32_t base) {
            const uint32_t sh0 = base & 31u;
            const uint32_t i = rd.seek(base >> 5);
            const uint32_t w0 = lget(rd.win, i), w1 = lget(rd.win, i + 1), w2 = lget(rd.win, i + 2), w3 = lget(rd.win, i + 3);
            LFOR(l) {
                const uint32_t sft = sh0 + uint32_t(l);   // 0..94
                const uint32_t a = sft < 32u ? w0 : sft < 64u ? w1 : w2, b = sft < 32u ? w1 : sft < 64u ? w2 : w3;
                const uint32_t v = uint32_t((((uint64_t(a) << 32) | b) << (sft & 31u)) >> 32), top16 = v >> 16;
                const uint32_t e1 = l_root[top16 >> 7];
                const bool two = (e1 & 0x8000u) != 0;   // a 9-bit prefix shared by longer codes: the second level (read by every lane: no branch)
                const uint32_t e2 = l_sub[two ? (e1 & 0xFFFu) + ((top16 & 127u) >> (7u - ((e1 >> 12) & 7u))) : 0u];
                const uint32_t e = two ? e2 : e1;
                const uint32_t len = e ? e >> 8 : 16u, sym = e & 255u;   // no such code: 16 bits, symbol 0 (k_decode.hip huff_decode)
                const uint32_t r = sym >> 4, n = sym & 15u;
                const bool eob = !n && r != 15u, zrl = !n && r == 15u;
                const uint32_t raw = ((v << len) >> 1) >> (31u - r);   // the r bits behind the symbol (r = 0: none)
                // (masks, not conditions: no lane may branch in here)
                E[l] = (len + (n ? 1u : eob ? r : 0u)) | (r << 5) | ((1024u | (((1u << r) + raw) << 16)) & (0u - uint32_t(eob))) | (2048u & (0u - uint32_t(zrl)));
            }
            ebase = base;
        };
        for (uint32_t O0 = 0; O0 < pc.nblocks; O0 += 64) {
            const uint32_t cnt = pc.nblocks - O0 < 64u ? pc.nblocks - O0 : 64u;
            LV<uint32_t> hlo, hhi, nlo, nhi, rec;
            wait_for(O0 + cnt);
            LFOR(l) {
                const uint64_t h = uint32_t(l) < cnt ? coherent_load(&H[O0 + uint32_t(l)]) : 0ull;
                hlo[l] = nlo[l] = uint32_t(h); hhi[l] = nhi[l] = uint32_t(h >> 32);
                rec[l] = CSH_RF_SKIP;
            }
            uint32_t i = 0;
            while (i < cnt && !dead) {
                if (pos > len8) { dead = true; break; }
                if (eobrun) {   // blocks inside an EOB run carry nothing but the correction bits of their history: up to 64 of them in one step
                    const uint32_t n = eobrun < cnt - i ? eobrun : cnt - i;
                    LV<uint32_t> nc;
                    LFOR(l) nc[l] = (uint32_t(l) >= i && uint32_t(l) < i + n) ? rf_popc64(((uint64_t(hhi[l]) << 32) | hlo[l]) & band) : 0u;
                    uint32_t total;
                    const LV<uint32_t> ex = lscan(nc, total);
                    LFOR(l) if (uint32_t(l) >= i && uint32_t(l) < i + n) rec[l] = pos + ex[l] > len8 ? CSH_RF_SKIP : ((pos + ex[l]) | CSH_RF_INRUN);
                    pos += total; eobrun -= n; i += n;
                    continue;
                }
                const uint64_t Hb = (uint64_t(lget(hhi, i)) << 32) | lget(hlo, i);
                uint64_t Nb = 0;
                lset(rec, i, pos);
                // the band's positions that are still ahead: Z those with zero history, Hr those with non-zero history (every position is in one of them)
                uint64_t Z = ~Hb & band, Hr = Hb & band;
                for (;;) {
                    uint32_t e;
                    const uint32_t why = rf_fast_symbols(pos, Z, Hr, Nb, e, ebase, E);
                    if (why == 0u) break;
                    if (why == 2u) {   // fewer zero-history positions than the run wants: the band ends here (libjpeg writes the coefficient behind it)
                        pos += (e & 31u) + rf_popc64(Hr);
                        Nb |= beyond;
                        Hr = 0;
                        // ... which is a coefficient of ANOTHER scan's band, and this file's first scans all ran before its refinement scans: whether the
                        // value stays is a matter of file order -- the sequential kernel's (damaged data only: an encoder's run ends inside the band)
                        if (beyond) need_seq[pc.image] = 2u;
                        break;
                    }
                    // the next symbol the general way: outside the decoded positions, EOBn or ZRL
                    if (pos - ebase >= 64u) { symbols_at(pos); continue; }
                    e = lget(E, pos - ebase);
                    pos += e & 31u;
                    if (e & 1024u) { eobrun = e >> 16; break; }   // EOBn: the rest of this block below
                    for (uint32_t r = 15; r && Z; r--) Z &= Z - 1;   // ZRL: fifteen zero-history positions are passed, the sixteenth is reached (no coefficient)
                    if (!Z) { pos += rf_popc64(Hr); Hr = 0; break; }   // the band ends first
                    const uint64_t low = Z & (0ull - Z), below = low - 1;
                    pos += rf_popc64(Hr & below);
                    Hr &= ~below;
                    Z &= ~low;
                    if (!(Z | Hr)) break;
                }
                if (eobrun > 0) { pos += rf_popc64(Hr); eobrun--; }
                if (Nb) { lset(nlo, i, uint32_t(Hb | Nb)); lset(nhi, i, uint32_t((Hb | Nb) >> 32)); }
                i++;
            }
            LFOR(l) if (uint32_t(l) < cnt) {
                rec_out[O0 + uint32_t(l)] = rec[l];
                H[O0 + uint32_t(l)] = (uint64_t(nhi[l]) << 32) | nlo[l];
            }
            if (dead) {   // the data ended: every later block of this scan is skipped; their masks stay what the previous scan makes them
                for (uint32_t o = O0 + 64u; o < pc.nblocks; o += 64u) LFOR(l) if (o + uint32_t(l) < pc.nblocks) rec_out[o + uint32_t(l)] = CSH_RF_SKIP;
                wait_for(pc.nblocks);
                publish(pc.nblocks);
                break;
            }
            publish(O0 + cnt);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- the apply
struct RfLaneBits {   // one lane's bit reader over the unstuffed stream
    const uint8_t *base;
    uint32_t len, wi;
    uint64_t acc;
    int nb;
    __device__ __forceinline__ void begin(const uint8_t *p, uint32_t n, uint32_t pos) {
        base = p; len = n; wi = pos >> 5;
        acc = (uint64_t(rf_loadw(base, len, wi)) << 32) | rf_loadw(base, len, wi + 1);
        wi += 2;
        const int sh = int(pos & 31u);
        acc <<= sh; nb = 64 - sh;
        if (nb < 32) refill();
    }
    __device__ __forceinline__ void refill() { acc |= uint64_t(rf_loadw(base, len, wi)) << (32 - nb); wi++; nb += 32; }
    __device__ __forceinline__ uint32_t top32() const { return uint32_t(acc >> 32); }
    __device__ __forceinline__ void skip(int n) { acc <<= n; nb -= n; if (nb < 32) refill(); }   // n <= 32
    __device__ __forceinline__ uint32_t get1() { const uint32_t v = uint32_t(acc >> 63); skip(1); return v; }
};

__global__ void __launch_bounds__(256) k_refine_apply(const uint8_t *clean, const ParScan *pss, const ParHuffSet *huffs, const DecScan *scans, const ProgChain *chains,
                                                      const int *chain_scans, int nchains, const ImgDesc *imgs, int16_t *coef, const uint32_t *need_seq,
                                                      const uint32_t *posv) {
    const int ch = blockIdx.y;
    if (ch >= nchains) return;
    const ProgChain pc = chains[ch];
    if (!pc.refine || need_seq[pc.image] != 4) return;
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    if (o >= pc.nblocks) return;
    const CompGeom &g = imgs[pc.image].in[pc.comp];
    const int by = int(o / uint32_t(g.real_bw)), bx = int(o - uint32_t(by) * uint32_t(g.real_bw));
    int16_t *blk = coef + coef_index(g.tile_base, by * g.bw + bx, 0);
    uint64_t H = rf_block_hist(blk);   // kept up to date below: memory is touched only where a coefficient changes
    for (int s = 0; s < pc.count; s++) {
        const uint32_t rec = posv[size_t(pc.pos_off) + size_t(s) * pc.nblocks + o];
        if (rec == CSH_RF_SKIP) continue;
        const DecScan &sc = scans[chain_scans[pc.first + s]];
        const ParScan &ps = pss[sc.par_index];
        const ParHuffSet &hs = huffs[sc.huff_set];
        const int tbl = 4 + (sc.ta[0] & 3), Se = sc.Se, p1 = 1 << sc.Al;
        RfLaneBits rd;
        rd.begin(clean + ps.bits_off, ps.clean_len, rec & ~CSH_RF_INRUN);
        // every position of Hc takes one correction bit, in position order
        auto corrections = [&](uint64_t Hc) {
            while (Hc) {
                const int j = __ffsll((unsigned long long)Hc) - 1;
                Hc &= Hc - 1;
                if (!rd.get1()) continue;
                const int c = blk[coef_off(j)];
                if ((c & p1) == 0) blk[coef_off(j)] = int16_t(c + (c >= 0 ? p1 : -p1));
            }
        };
        int k = sc.Ss;
        bool inrun = (rec & CSH_RF_INRUN) != 0;
        while (!inrun && k <= Se) {
            const uint32_t w = rd.top32(), top16 = w >> 16;
            uint32_t e = hs.root[tbl][top16 >> 7];
            if (e & 0x8000u) e = hs.sub[(e & 0xFFFu) + ((top16 & 127u) >> (7u - ((e >> 12) & 7u)))];
            const int len = e ? int(e >> 8) : 16, sym = int(e & 255u);
            const int r = sym >> 4, n = sym & 15;
            rd.skip(len);
            int val = 0;
            if (n) val = rd.get1() ? p1 : -p1;
            else if (r != 15) { if (r) rd.skip(r); inrun = true; break; }
            int kz = rf_select(~H & rf_span(k, Se), r);
            if (kz > Se) kz = Se + 1;
            corrections(H & rf_span(k, kz - 1));
            if (val && kz <= 63) { blk[coef_off(kz)] = int16_t(val); H |= 1ull << kz; }   // positions up to kz are not looked at again in this scan
            k = kz + 1;
        }
        if (inrun) corrections(H & rf_span(k, Se));
    }
}

// ---------------------------------------------------------------------------------------------------------------- launch
void launch_refine_chains(hipStream_t st, const uint8_t *clean, const ParScan *pss, const ParHuffSet *huffs, const DecScan *scans, const ProgChain *chains,
                          const int *chain_scans, int nchains, const RefineUnit *units, int nunits, uint32_t max_blocks, const ImgDesc *imgs, int16_t *coef,
                          uint32_t *need_seq, uint64_t *hist, uint32_t *posv, uint32_t *prog) {
    if (!nunits || !max_blocks) return;
    const dim3 per_block((max_blocks + 255u) / 256u, unsigned(nchains));
    (void)hipMemsetAsync(prog, 0, (size_t(nunits) + 1) * sizeof(uint32_t), st);
    CSH_LAUNCH(k_refine_hist, per_block, dim3(256), st, chains, nchains, imgs, coef, need_seq, hist);
    CSH_LAUNCH(k_refine_parse, dim3(unsigned(nunits)), dim3(CSP_WAVE_THREADS), st, clean, pss, huffs, scans, chains, chain_scans, units, nunits, need_seq, hist, posv, prog);
    CSH_LAUNCH(k_refine_apply, per_block, dim3(256), st, clean, pss, huffs, scans, chains, chain_scans, nchains, imgs, coef, need_seq, posv);
}

}  // namespace csh
