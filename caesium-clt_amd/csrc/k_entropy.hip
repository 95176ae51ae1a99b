// k_entropy.hip -- phases 2-5: progressive Huffman entropy ENCODE with optimal tables, on the device.
//
// Replaces mozjpeg's jcphuff.c (progressive scans, EOBRUN, correction bits) + jchuff.c
// jpeg_gen_optimal_table for libcaesium's JPEG path (reference call site
// /root/reference/src/compressor.rs:305).
//
// Formulation (DESIGN.md "Entropy encode"): the coefficient planes are read ONCE.  k_tokens gives a workgroup 256 consecutive
// blocks of one component; a lane holds its block's 64 coefficients in registers, derives the per-block bit planes
// (bit k: |c_k| >= 2^l, bit l of |c_k|, sign) and codes EVERY AC scan of the component from them:
//   first pass  (Ah=0)     : coded positions NZ = sig[Al] & band;  zero run = gap between set bits
//   refinement  (Ah=Al+1)  : history H = sig[Al+1] & band, newly significant N = sig[Al] & ~sig[Al+1] & band;
//                            zero run = gap minus popcount(H in the gap); correction bits = plane `bit Al` at the positions of H
//   block ends with EOB    : bit Se of NZ (resp. N) is clear
// What a block emits in a scan becomes a short run of TOKENS (one u32 each: a Huffman symbol with its raw bits, a first-pass
// coefficient with its zero run, raw correction bits, or the place where the block's EOBRUN symbol goes); the symbol histograms
// are taken in the same pass (LDS).  EOB runs span blocks: k_ac_runs resolves them from two per-scan bit vectors (has-symbol,
// ends-with-EOB) -- the first block of every (sub-)run owns the EOBRUN symbol -- and adds those symbols to the histograms.
// After the optimal tables exist, k_chunk_sizes sums code lengths over each chunk's tokens, one exclusive scan places every
// chunk, and k_pack turns a chunk's tokens into bits in LDS and moves them to their final place with one shifted copy.
// Passes: tokens(+flags+stats) -> runs -> optimal tables -> chunk sizes -> exclusive scan -> pack.
#include "kernels.h"

namespace csh {

// ------------------------------------------------------------------------------------------------ tokens
//  kind (bits 0-1)
//   SYM  Huffman symbol + raw bits behind it:  [9:2] symbol  [11:10] table of the scan's group  [15:12] n raw bits (0..15)  [30:16] the bits
//   RAW  raw bits only:                        [15:12] n (1..15)  [30:16] the bits
//   ACF  first-pass AC coefficient:            [7:2] zero run in front of it (0..62; every 16 cost one ZRL)  [11:8] size  [27:12] its bits
//   EOB  the block ends with an EOB here: the packer emits the unit's EOBRUN symbol (eobrun[unit], if it owns one)
enum : uint32_t { TK_SYM = 0u, TK_RAW = 1u, TK_ACF = 2u, TK_EOB = 3u };
#define CSH_TK_STAGE 4096   // tokens a workgroup stages in LDS (16 KB); larger chunks go to / come from HBM directly
#define CSH_PK_WORDS 2048   // bit buffer of the packer in LDS (64 Kbit); larger chunks are packed into HBM directly

__device__ __forceinline__ static uint64_t band_mask(int Ss, int Se) { return (~0ull >> (63 - Se)) & (~0ull << Ss); }
__device__ __forceinline__ static int msb64(uint64_t v) { return 63 - __clzll(v); }
__device__ __forceinline__ static int bitlen32(unsigned v) { return 32 - __clz(v); }
__device__ __forceinline__ static int lane_id() { return int(threadIdx.x & 63); }

// unit -> padded block index for a non-interleaved scan of component geometry g
__device__ __forceinline__ static int unit_block(const CompGeom &g, uint32_t u) {
    int by = int(u) / g.real_bw, bx = int(u) - by * g.real_bw;
    return by * g.bw + bx;
}
__device__ __forceinline__ static bool get_bit(const uint64_t *w, uint32_t i) { return (w[i >> 6] >> (i & 63)) & 1; }

// inclusive scan over the 64 lanes of a wave of the values in[0..63] (LDS, written in an earlier phase), for lane `lane`
__device__ __forceinline__ static uint32_t wave_incl_scan(const uint32_t *in, int lane) {
#ifdef CSH_EMUL
    uint32_t s = 0;
    for (int i = 0; i <= lane; i++) s += in[i];
    return s;
#else
    uint32_t v = in[lane];
    CSH_UNROLL
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = uint32_t(__shfl_up(int(v), o, 64)); if (lane >= o) v += t; }
    return v;
#endif
}

// where a lane's tokens go: the workgroup's LDS stage, or (chunks with more tokens than the stage holds) the pool itself
struct TokOut {
    uint32_t *stage, *pool;
    bool staged;
    uint32_t pos;   // next token, relative to the chunk's first
    __device__ __forceinline__ void put(uint32_t t) { if (staged) stage[pos] = t; else pool[pos] = t; pos++; }
};

// the sink of the walkers below.  EMIT = false only counts tokens (the same merge rules, so the count is what EMIT = true writes);
// EMIT = true writes them and counts the symbols into the workgroup's histogram.
template <bool EMIT>
struct TokSink {
    uint32_t n;         // tokens so far
    uint32_t pend;      // a SYM / RAW token that may still take raw bits
    bool has_pend;
    TokOut out;
    uint32_t *hist;     // [h0 + table][257] in LDS
    int h0;
    static constexpr bool kValues = true;
    __device__ __forceinline__ void begin() { n = 0; pend = 0; has_pend = false; }
    __device__ __forceinline__ void flush() { if (has_pend) { if (EMIT) out.put(pend); n++; has_pend = false; } }
    __device__ __forceinline__ void sym(int t, int s) {
        flush();
        pend = TK_SYM | (uint32_t(s) << 2) | (uint32_t(t) << 10); has_pend = true;
        if (EMIT) atomicAdd(&hist[(h0 + t) * 257 + s], 1u);
    }
    __device__ __forceinline__ void syms(int t, int s, int cnt) { for (int i = 0; i < cnt; i++) sym(t, s); }
    __device__ __forceinline__ void raw(unsigned v, int nb) {   // the nb low bits of v, most significant first
        while (nb > 0) {
            int have = has_pend ? int((pend >> 12) & 15u) : 0;
            if (has_pend && have == 15) { flush(); have = 0; }
            if (!has_pend) { pend = TK_RAW; has_pend = true; }
            const int take = nb < 15 - have ? nb : 15 - have;
            const uint32_t piece = (v >> (nb - take)) & ((1u << take) - 1u);
            const uint32_t val = (((pend >> 16) << take) | piece) & 0x7FFFu;
            pend = (pend & 0xFFFu) | (uint32_t(have + take) << 12) | (val << 16);
            nb -= take;
        }
    }
    __device__ __forceinline__ void eob() { flush(); if (EMIT) out.put(TK_EOB); n++; }
    __device__ __forceinline__ void finish() { flush(); }
};

// refinement scan of one block (jcphuff.c encode_mcu_AC_refine order: correction bits ride behind the next symbol); no coefficient
// is read: the correction bit and the sign come from the bit planes
template <class Sink>
__device__ static void walk_ac_refine(Sink &sink, uint64_t H, uint64_t N, uint64_t C, uint64_t S, int Ss, bool ends_eob) {
    int eobpos = N ? msb64(N) : -1;
    uint64_t all = H | N;
    int prev = Ss - 1, r = 0;
    uint64_t br = 0; int brn = 0;  // pending correction bits, oldest first in the high end
    while (all) {
        int k = __ffsll((unsigned long long)all) - 1;
        all &= all - 1;
        r += k - prev - 1;
        prev = k;
        if (k <= eobpos)
            while (r > 15) {
                sink.sym(0, 0xF0); r -= 16;
                if (brn > 32) { sink.raw(unsigned(br >> 32), brn - 32); }
                if (brn) sink.raw(unsigned(br), brn > 32 ? 32 : brn);
                br = 0; brn = 0;
            }
        if ((H >> k) & 1) { br = (br << 1) | ((C >> k) & 1); brn++; }
        else {
            sink.sym(0, (r << 4) | 1);
            sink.raw(unsigned((~S >> k) & 1), 1);
            if (brn > 32) { sink.raw(unsigned(br >> 32), brn - 32); }
            if (brn) sink.raw(unsigned(br), brn > 32 ? 32 : brn);
            br = 0; brn = 0; r = 0;
        }
    }
    if (ends_eob) sink.eob();
    if (brn > 32) { sink.raw(unsigned(br >> 32), brn - 32); }
    if (brn) sink.raw(unsigned(br), brn > 32 ? 32 : brn);
}

// DC scans: unit = MCU (interleaved) or block (single component)
template <class Sink>
__device__ static void walk_dc(Sink &sink, const EncCtx &c, const ImgDesc &im, const EncScan &sc, uint32_t u) {
    for (int ci = 0; ci < sc.ncomp; ci++) {
        const CompGeom &g = im.out[sc.comp[ci]];
        int nb_x = sc.ncomp > 1 ? g.h : 1, nb_y = sc.ncomp > 1 ? g.v : 1;
        int mx = 0, my = 0;
        if (sc.ncomp > 1) { my = int(u) / im.omcus_x; mx = int(u) - my * im.omcus_x; }
        int pred = 0;
        bool have_pred = false;
        for (int y = 0; y < nb_y; y++)
            for (int x = 0; x < nb_x; x++) {
                int b = sc.ncomp > 1 ? (my * g.v + y) * g.bw + mx * g.h + x : unit_block(g, u);
                int dc = c.coef[coef_index(g.tile_base, b, 0)];
                if (sc.Ah) { sink.raw(unsigned(dc >> sc.Al) & 1u, 1); continue; }
                if (!have_pred) {
                    // predictor = previous block of this component in scan order
                    if (u == 0) pred = 0;
                    else if (sc.ncomp > 1) {
                        int pu = int(u) - 1, pmy = pu / im.omcus_x, pmx = pu - pmy * im.omcus_x;
                        int pb = (pmy * g.v + g.v - 1) * g.bw + pmx * g.h + g.h - 1;
                        pred = c.coef[coef_index(g.tile_base, pb, 0)] >> sc.Al;
                    } else pred = c.coef[coef_index(g.tile_base, unit_block(g, u - 1), 0)] >> sc.Al;
                    have_pred = true;
                }
                int t2 = dc >> sc.Al;
                int t = t2 - pred;
                pred = t2;
                unsigned a = unsigned(t < 0 ? -t : t);
                int nb = bitlen32(a);
                sink.sym(sc.dc_tbl[ci], nb);
                sink.raw(unsigned(t < 0 ? t - 1 : t), nb);
            }
    }
}

// |c| >= 1 plane of one block, straight from its coefficients (eight 16-byte loads)
__device__ static uint64_t block_nz_mask(const int16_t *blk) {
    uint64_t m = 0;
    CSH_UNROLL
    for (int j = 0; j < 8; j++) {
        const uint4 q = *reinterpret_cast<const uint4 *>(blk + CSH_OCT_STRIDE * j);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        CSH_UNROLL
        for (int i = 0; i < 8; i++) { const uint32_t h = (i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xFFFFu); m |= uint64_t(h != 0) << (8 * j + i); }
    }
    return m;
}
// sequential-mode scan (jchuff.c encode_one_block behaviour): unit = MCU (interleaved) or block; per block DC difference,
// then the AC coefficients straight off the |c|>=1 mask (zero run = gap between set bits), EOB unless position 63 is coded
template <class Sink>
__device__ static void walk_seq(Sink &sink, const EncCtx &c, const ImgDesc &im, const EncScan &sc, uint32_t u) {
    for (int ci = 0; ci < sc.ncomp; ci++) {
        const CompGeom &g = im.out[sc.comp[ci]];
        int nb_x = sc.ncomp > 1 ? g.h : 1, nb_y = sc.ncomp > 1 ? g.v : 1;
        int mx = 0, my = 0;
        if (sc.ncomp > 1) { my = int(u) / im.omcus_x; mx = int(u) - my * im.omcus_x; }
        int pred = 0;
        bool have_pred = false;
        for (int y = 0; y < nb_y; y++)
            for (int x = 0; x < nb_x; x++) {
                int b = sc.ncomp > 1 ? (my * g.v + y) * g.bw + mx * g.h + x : unit_block(g, u);
                const int16_t *blk = c.coef + coef_index(g.tile_base, b, 0);
                if (!have_pred) {
                    if (u == 0) pred = 0;
                    else if (sc.ncomp > 1) {
                        int pu = int(u) - 1, pmy = pu / im.omcus_x, pmx = pu - pmy * im.omcus_x;
                        pred = c.coef[coef_index(g.tile_base, (pmy * g.v + g.v - 1) * g.bw + pmx * g.h + g.h - 1, 0)];
                    } else pred = c.coef[coef_index(g.tile_base, unit_block(g, u - 1), 0)];
                    have_pred = true;
                }
                int dc = blk[0];
                int t = dc - pred;
                pred = dc;
                unsigned a = unsigned(t < 0 ? -t : t);
                int nb = bitlen32(a);
                sink.sym(sc.dc_tbl[ci], nb);
                sink.raw(unsigned(t < 0 ? t - 1 : t), nb);
                uint64_t NZ = block_nz_mask(blk) & ~1ull;
                int prev = 0;
                while (NZ) {
                    int k = __ffsll((unsigned long long)NZ) - 1;
                    NZ &= NZ - 1;
                    int r = k - prev - 1;
                    prev = k;
                    sink.syms(sc.ac_tbl[ci], 0xF0, r >> 4);
                    int v = blk[coef_off(k)];
                    unsigned av = unsigned(v < 0 ? -v : v);
                    int nv = bitlen32(av);
                    sink.sym(sc.ac_tbl[ci], ((r & 15) << 4) | nv);
                    sink.raw(v < 0 ? ~av : av, nv);
                }
                if (prev < 63) sink.sym(sc.ac_tbl[ci], 0x00);
            }
    }
}

// |x| of both 16-bit halves of a word
__device__ __forceinline__ static uint32_t pk_abs16(uint32_t w) {
#ifdef CSH_EMUL
    const int lo = int(w << 16) >> 16, hi = int(w) >> 16;
    return uint32_t(lo < 0 ? -lo : lo) | (uint32_t(hi < 0 ? -hi : hi) << 16);
#else
    typedef short short2v __attribute__((ext_vector_type(2)));
    const short2v v = __builtin_bit_cast(short2v, w);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, -v));   // v_pk_sub_i16 + v_pk_max_i16
#endif
}
// 32 x 32 bit matrix transpose in registers (recursive block swap): afterwards bit r of word c is what bit c of word r was
__device__ __forceinline__ static void transpose32(uint32_t (&A)[32]) {
    uint32_t msk = 0x0000FFFFu;
    CSH_UNROLL
    for (int j = 16; j != 0; j >>= 1) {
        CSH_UNROLL
        for (int k = 0; k < 32; k++)
            if (!(k & j)) {
                const uint32_t t = ((A[k] >> j) ^ A[k + j]) & msk;
                A[k + j] ^= t; A[k] ^= t << j;
            }
        msk ^= msk << (j >> 1);
        CSH_SCHED_FENCE();
    }
}

// ---- the bit planes of one block, in registers
struct Planes {
    uint64_t sig[5];   // bit k: |c_k| >= 1, 2, 4, 8, 16
    uint64_t bit[4];   // bit k: bit 0..3 of |c_k|
    uint64_t sgn;      // bit k: c_k < 0
};
__device__ __forceinline__ static uint64_t pick_sig(const uint64_t *s, int l) { return l == 0 ? s[0] : l == 1 ? s[1] : l == 2 ? s[2] : l == 3 ? s[3] : s[4]; }
__device__ __forceinline__ static uint64_t pick_bit(const uint64_t *s, int l) { return l == 0 ? s[0] : l == 1 ? s[1] : l == 2 ? s[2] : s[3]; }

// is scan `sc` one of the AC scans a kind-0 chunk of component `comp` carries?
__device__ __forceinline__ static bool ac_scan_of(const EncScan &sc, int comp) { return sc.Ss > 0 && !sc.sequential && sc.comp[0] == comp; }

// ---- pass A: tokens, flags, statistics
// state of a lane across the phases: the planes.  The coefficients themselves are loaded again where the first-pass scans are coded
// (from the L2: the workgroup read the same lines a few microseconds earlier) -- 32 registers per lane held across the barriers cost
// more in occupancy than the second load does.
__global__ void __launch_bounds__(256, 4) k_tokens(EncCtx c) {
    CSH_SHARED uint32_t hist[CSH_TK_MAXSLOT * 257];
    CSH_SHARED uint32_t cnt[CSH_TK_MAXSLOT][256];   // tokens per (slot, lane); after the scan: exclusive offsets inside the slot
    CSH_SHARED uint32_t stage[CSH_TK_STAGE];
    CSH_SHARED uint32_t s_tot[CSH_TK_MAXSLOT], s_base[CSH_TK_MAXSLOT + 1], s_tbase[CSH_TK_MAXSLOT], s_nh, s_flags;
    CSH_SHARED unsigned long long s_gbase;
    CSH_PERSIST(uint64_t, pl, 10);     // bit k of: |c_k| >= 1, 2, 4, 8, 16; bit 0..3 of |c_k|; c_k < 0
    const EChunk ch = c.echunks[blockIdx.x];
    const int tid = int(threadIdx.x), lane = lane_id(), wv = tid >> 6;
    const ImgDesc &im = c.imgs[ch.kind == 0 ? ch.a : c.work[ch.a].image];
    const uint32_t u = ch.j * 256u + uint32_t(tid);

    CSH_PHASE_LOOP(5) {
        if (phase == 0) {
            // ---------------------------------------------------------------- load, planes, counts, flags
            for (int i = tid; i < CSH_TK_MAXSLOT * 257; i += 256) hist[i] = 0;
            if (ch.kind == 1) {
                const ScanWork &w = c.work[ch.a];
                const EncScan &sc = c.script[w.scan];
                TokSink<false> s; s.begin();
                if (u < w.nunits) { if (sc.sequential) walk_seq(s, c, im, sc, u); else walk_dc(s, c, im, sc, u); s.finish(); }
                cnt[0][tid] = s.n;
                continue;
            }
            const CompGeom &g = im.out[ch.comp];
            const bool valid = u < uint32_t(g.real_bw * g.real_bh);
            CSH_UNROLL
            for (int i = 0; i < 10; i++) pl[i] = 0;
            if (valid) {
                const int b = unit_block(g, u);
                const int16_t *p = c.coef + (size_t(g.tile_base) + size_t(b >> 6)) * CSH_TILE_I16 + size_t((b & 63) * CSH_BLK_STRIDE);
                uint32_t sm[32];   // sign-and-magnitude halves: bits 0-14 |c|, bit 15 the sign
                CSH_UNROLL
                for (int j = 0; j < 8; j++) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(p + CSH_OCT_STRIDE * j);
                    const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
                    CSH_UNROLL
                    for (int i = 0; i < 4; i++) sm[4 * j + i] = pk_abs16(w4[i]) | (w4[i] & 0x80008000u);
                }
                CSH_SCHED_FENCE();   // stage after stage: interleaving them for instruction-level parallelism costs a hundred registers
                // all sixteen planes at once: a 32 x 32 bit transpose.  Row 2t pairs coefficient 2t with 2t + 32, row 2t + 1 pairs 2t + 1 with
                // 2t + 33, so that after the transpose word p holds plane p of k = 0..31 and word p + 16 plane p of k = 32..63
                uint32_t m[32];
                CSH_UNROLL
                for (int t = 0; t < 16; t++) {
                    m[2 * t] = (sm[t] & 0xFFFFu) | (sm[t + 16] << 16);
                    m[2 * t + 1] = (sm[t] >> 16) | (sm[t + 16] & 0xFFFF0000u);
                }
                CSH_SCHED_FENCE();
                transpose32(m);
                CSH_SCHED_FENCE();
                uint64_t acc = 0;
                CSH_UNROLL
                for (int pb = 14; pb >= 0; pb--) {
                    const uint64_t plane = uint64_t(m[pb]) | (uint64_t(m[pb + 16]) << 32);
                    acc |= plane;
                    if (pb <= 4) pl[pb] = acc;          // |c| >= 2^pb
                    if (pb <= 3) pl[5 + pb] = plane;    // bit pb of |c|
                }
                pl[9] = uint64_t(m[15]) | (uint64_t(m[31]) << 32);
                CSH_SCHED_FENCE();
            }
            const uint64_t sgn = pl[9];
            int slot = 0;
            for (int sl = 0; sl < im.nscans_out && slot < CSH_TK_MAXSLOT; sl++) {
                const ScanWork &w = c.work[im.first_work + sl];
                const EncScan &sc = c.script[w.scan];
                if (!ac_scan_of(sc, ch.comp)) continue;
                const uint64_t band = band_mask(sc.Ss, sc.Se);
                const uint64_t lo = pick_sig(pl, sc.Al);
                bool has_sym = false, ends_eob = false;
                uint32_t n = 0; int tail = 0;
                if (valid) {
                    if (sc.Ah == 0) {
                        const uint64_t NZ = lo & band;
                        has_sym = NZ != 0; ends_eob = !((NZ >> sc.Se) & 1);
                        n = uint32_t(__popcll(NZ)) + (ends_eob ? 1u : 0u);
                    } else {
                        const uint64_t hi = pick_sig(pl, sc.Al + 1), H = hi & band, N = lo & ~hi & band;
                        has_sym = N != 0; ends_eob = !((N >> sc.Se) & 1);
                        tail = N ? __popcll(H & ~((2ull << msb64(N)) - 1)) : __popcll(H);
                        TokSink<false> s; s.begin();
                        walk_ac_refine(s, H, N, pick_bit(pl + 5, sc.Al), sgn, sc.Ss, ends_eob);
                        s.finish();
                        n = s.n;
                    }
                    c.tail[w.unit_base + u] = uint8_t(tail);
                }
                cnt[slot][tid] = n;
#ifdef CSH_EMUL
                if (has_sym) atomicOr(reinterpret_cast<unsigned long long *>(c.sym_bits + w.word_base + (u >> 6)), 1ull << (u & 63));
                if (ends_eob) atomicOr(reinterpret_cast<unsigned long long *>(c.eob_bits + w.word_base + (u >> 6)), 1ull << (u & 63));
#else
                // lane = block, so a wave's 64 flags ARE one word of the scan's bit vectors: one ballot, one 8-byte store
                const uint64_t ms = __ballot(has_sym), me = __ballot(ends_eob);
                if (lane == 0 && (u >> 6) < ((w.nunits + 63) >> 6)) { c.sym_bits[w.word_base + (u >> 6)] = ms; c.eob_bits[w.word_base + (u >> 6)] = me; }
#endif
                slot++;
            }
            continue;
        }
        if (phase == 1) {
            // ---------------------------------------------------------------- exclusive scan of the counts, slot by slot (wave w: slots w, w + 4)
            for (int slot = wv; slot < CSH_TK_MAXSLOT; slot += 4) {
#ifdef CSH_EMUL
                if (lane == 0) { uint32_t acc = 0; for (int i = 0; i < 256; i++) { uint32_t v = cnt[slot][i]; cnt[slot][i] = acc; acc += v; } s_tot[slot] = acc; }
#else
                uint32_t carry = 0;
                CSH_UNROLL
                for (int q = 0; q < 4; q++) {
                    const uint32_t v = cnt[slot][64 * q + lane];
                    const uint32_t incl = wave_incl_scan(&cnt[slot][64 * q], lane);
                    cnt[slot][64 * q + lane] = carry + incl - v;
                    carry += uint32_t(__shfl(int(incl), 63, 64));
                }
                if (lane == 0) s_tot[slot] = carry;
#endif
            }
            continue;
        }
        if (phase == 2) {
            // ---------------------------------------------------------------- one lane: room in the pool for the whole chunk, slot by slot
            if (tid != 0) continue;
            uint32_t total = 0;
            int nslot = 0;
            if (ch.kind == 1) { nslot = 1; s_base[0] = 0; total = s_tot[0]; }
            else
                for (int sl = 0; sl < im.nscans_out && nslot < CSH_TK_MAXSLOT; sl++)
                    if (ac_scan_of(c.script[c.work[im.first_work + sl].scan], ch.comp)) { s_base[nslot] = total; total += s_tot[nslot]; nslot++; }
            s_base[nslot] = total;
            const unsigned long long gb = atomicAdd(c.tok_cursor, (unsigned long long)total);
            const bool ok = gb + total <= c.tok_cap;
            if (!ok) c.overflow[1] = 1;
            s_gbase = gb;
            s_flags = (ok ? 1u : 0u) | (total <= CSH_TK_STAGE ? 2u : 0u);
            if (ch.kind == 1) {
                const ScanWork &w = c.work[ch.a];
                const EncScan &sc = c.script[w.scan];
                c.tok_off[w.first_chunk + ch.j] = gb; c.chunk_ntok[w.first_chunk + ch.j] = ok ? total : 0u;
                s_nh = uint32_t(sc.ntables);
                for (int t = 0; t < sc.ntables; t++) s_tbase[t] = w.table_base + uint32_t(t);
            } else {
                int slot = 0;
                for (int sl = 0; sl < im.nscans_out && slot < CSH_TK_MAXSLOT; sl++) {
                    const ScanWork &w = c.work[im.first_work + sl];
                    if (!ac_scan_of(c.script[w.scan], ch.comp)) continue;
                    c.tok_off[w.first_chunk + ch.j] = gb + s_base[slot]; c.chunk_ntok[w.first_chunk + ch.j] = ok ? s_tot[slot] : 0u;
                    s_tbase[slot] = w.table_base;
                    slot++;
                }
                s_nh = uint32_t(slot);
            }
            continue;
        }
        if (phase == 3) {
            // ---------------------------------------------------------------- tokens + histograms
            if (!(s_flags & 1u)) continue;
            TokOut out; out.stage = stage; out.pool = c.tokens + s_gbase; out.staged = (s_flags & 2u) != 0;
            if (ch.kind == 1) {
                const ScanWork &w = c.work[ch.a];
                const EncScan &sc = c.script[w.scan];
                if (u >= w.nunits) continue;
                TokSink<true> s; s.begin(); s.out = out; s.out.pos = cnt[0][tid]; s.hist = hist; s.h0 = 0;
                if (sc.sequential) walk_seq(s, c, im, sc, u); else walk_dc(s, c, im, sc, u);
                s.finish();
                c.unit_ntok[w.unit_base + u] = uint16_t(s.n);
                continue;
            }
            const CompGeom &g = im.out[ch.comp];
            if (u >= uint32_t(g.real_bw * g.real_bh)) continue;
            const uint64_t sgn = pl[9];
            uint32_t aw[32];   // |c_k|: k = 2 i in the low half of word i, k = 2 i + 1 in the high half
            {
                const int b = unit_block(g, u);
                const int16_t *p = c.coef + (size_t(g.tile_base) + size_t(b >> 6)) * CSH_TILE_I16 + size_t((b & 63) * CSH_BLK_STRIDE);
                CSH_UNROLL
                for (int j = 0; j < 8; j++) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(p + CSH_OCT_STRIDE * j);
                    aw[4 * j] = pk_abs16(q.x); aw[4 * j + 1] = pk_abs16(q.y); aw[4 * j + 2] = pk_abs16(q.z); aw[4 * j + 3] = pk_abs16(q.w);
                }
                CSH_SCHED_FENCE();
            }
            int slot = 0;
            for (int sl = 0; sl < im.nscans_out && slot < CSH_TK_MAXSLOT; sl++) {
                const ScanWork &w = c.work[im.first_work + sl];
                const EncScan &sc = c.script[w.scan];
                if (!ac_scan_of(sc, ch.comp)) continue;
                const uint64_t band = band_mask(sc.Ss, sc.Se);
                const uint64_t lo = pick_sig(pl, sc.Al);
                out.pos = s_base[slot] + cnt[slot][tid];
                const uint32_t pos0 = out.pos;
                if (sc.Ah == 0) {
                    // first pass: one token per coded coefficient; the 63 positions are a static sweep, so that |c_k| is a register
                    CSH_UNROLL
                    for (int i = 0; i < 32; i++) CSH_PIN(aw[i]);   // what the steps derive from loop-invariant registers (the unpacked halves, the sign
                    uint32_t sg_lo = uint32_t(sgn), sg_hi = uint32_t(sgn >> 32);   // bits) is not to be hoisted out of the loop over the scans: 64 + 63 live registers
                    CSH_PIN(sg_lo); CSH_PIN(sg_hi);
                    const uint64_t NZ = lo & band;
                    uint32_t *h = hist + slot * 257;
                    int prev = sc.Ss - 1;
                    CSH_UNROLL
                    for (int k = 1; k < 64; k++) {
                        CSH_SCHED_FENCE();
                        if ((NZ >> k) & 1) {
                            const uint32_t a = ((k & 1) ? (aw[k >> 1] >> 16) : (aw[k >> 1] & 0xFFFFu)) >> sc.Al;
                            const int nb = bitlen32(a);
                            const int r = k - prev - 1;
                            prev = k;
                            const uint32_t val = ((((k < 32 ? sg_lo >> (k & 31) : sg_hi >> (k & 31)) & 1u) ? ~a : a)) & ((1u << nb) - 1u);
                            out.put(TK_ACF | (uint32_t(r) << 2) | (uint32_t(nb) << 8) | (val << 12));
                            if (r >> 4) atomicAdd(&h[0xF0], uint32_t(r >> 4));
                            atomicAdd(&h[((r & 15) << 4) | nb], 1u);
                        }
                    }
                    if (!((NZ >> sc.Se) & 1)) out.put(TK_EOB);
                } else {
                    const uint64_t hi = pick_sig(pl, sc.Al + 1), H = hi & band, N = lo & ~hi & band;
                    TokSink<true> s; s.begin(); s.out = out; s.hist = hist; s.h0 = slot;
                    walk_ac_refine(s, H, N, pick_bit(pl + 5, sc.Al), sgn, sc.Ss, !((N >> sc.Se) & 1));
                    s.finish();
                    out.pos = s.out.pos;
                }
                c.unit_ntok[w.unit_base + u] = uint16_t(out.pos - pos0);
                slot++;
            }
            continue;
        }
        // -------------------------------------------------------------------- stage -> pool (coalesced), histograms -> tables
        if (!(s_flags & 1u)) continue;
        if (s_flags & 2u) {
            uint32_t *dst = c.tokens + s_gbase;
            const uint32_t total = s_base[ch.kind == 1 ? 1 : int(s_nh)];
            for (uint32_t i = uint32_t(tid); i < total; i += 256) dst[i] = stage[i];
        }
        for (uint32_t i = uint32_t(tid); i < s_nh * 257u; i += 256) {
            const uint32_t v = hist[i];
            if (v) atomicAdd(&c.tables[s_tbase[i / 257u]].freq[i % 257u], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ---- pass B: EOB run structure -> EOBRUN value owned by the first block of each (sub-)run, and those symbols' statistics
// A run [u .. t] is cut into sub-runs as jcphuff.c does: after 0x7FFF blocks, and (refinement) as soon as more than
// MAX_CORR_BITS - DCTSIZE2 + 1 = 937 correction bits are pending.  Serial form (short runs, and the emulation build):
// `h` = histogram of the EOBn symbols (index = n), LDS or the scan's table
template <class Count>
__device__ static void eob_run_serial(const EncCtx &c, const ScanWork &w, const EncScan &sc, uint32_t u, uint32_t t, Count count) {
    uint16_t *er = c.eobrun + w.unit_base;
    if (sc.Ah == 0) {
        uint32_t L = t - u + 1, pos = u;
        while (L > 0) { uint32_t l = L < 0x7FFF ? L : 0x7FFF; er[pos] = uint16_t(l); count(l); pos += l; L -= l; }
        return;
    }
    const uint8_t *tl = c.tail + w.unit_base;
    uint32_t cnt = 0, be = 0, s0 = u;
    auto step = [&](uint32_t j, uint32_t tail_bits) {
        cnt++; be += tail_bits;
        if (cnt == 0x7FFF || be > 937) { er[s0] = uint16_t(cnt); count(cnt); cnt = 0; be = 0; s0 = j + 1; }
    };
    uint32_t j = u;
    while (j <= t && (reinterpret_cast<uintptr_t>(tl + j) & 7)) { step(j, tl[j]); j++; }
    for (; j + 7 <= t; j += 8) {
        const uint64_t v = *reinterpret_cast<const uint64_t *>(tl + j);
        CSH_UNROLL
        for (int i = 0; i < 8; i++) step(j + i, uint32_t(v >> (8 * i)) & 255u);
    }
    for (; j <= t; j++) step(j, tl[j]);
    if (cnt) { er[s0] = uint16_t(cnt); count(cnt); }
}
// last block of the run that starts at u: the block before the next one that carries a symbol.  Looks at most `max_words`
// words of the has-symbol vector ahead; returns false if the end lies further on.
__device__ static bool eob_run_end(const uint64_t *sym, uint32_t nunits, uint32_t u, uint32_t max_words, uint32_t &t) {
    t = nunits - 1;
    uint32_t i = u + 1;
    const uint32_t nwords = (nunits + 63) >> 6;
    for (uint32_t n = 0; i < nunits; n++) {
        if (n == max_words) return false;
        uint32_t wi = i >> 6;
        uint64_t bits = sym[wi] & (~0ull << (i & 63));
        if (bits) { uint32_t p = (wi << 6) + uint32_t(__ffsll((unsigned long long)bits) - 1); if (p < nunits) t = p - 1; return true; }
        i = (wi + 1) << 6;
        if (wi + 1 >= nwords) break;
    }
    return true;
}
#define CSH_LONG_RUN_WORDS 8   // a run whose end is not within 8 words (512 blocks) goes to k_ac_runs_long: one WAVE per run
__global__ void __launch_bounds__(256) k_ac_runs(EncCtx c) {
    CSH_SHARED uint32_t eh[16];
    const uint32_t wi = c.slot_work[blockIdx.x];
    const ScanWork w = c.work[wi];
    const EncScan &sc = c.script[w.scan];
    CSH_PHASE_LOOP(3) {
        if (sc.Ss == 0 || sc.sequential) continue;
        if (phase == 0) { if (threadIdx.x < 16) eh[threadIdx.x] = 0; continue; }
        if (phase == 2) { if (threadIdx.x < 15 && eh[threadIdx.x]) atomicAdd(&c.tables[w.table_base].freq[threadIdx.x << 4], eh[threadIdx.x]); continue; }
        uint32_t u = (blockIdx.x - w.first_chunk) * blockDim.x + threadIdx.x;
        if (u >= w.nunits) continue;
        const uint64_t *sym = c.sym_bits + w.word_base, *eob = c.eob_bits + w.word_base;
        if (!get_bit(eob, u)) continue;
        bool start = get_bit(sym, u) || u == 0 || !get_bit(eob, u - 1);
        if (!start) continue;
        uint32_t t;
        if (eob_run_end(sym, w.nunits, u, CSH_LONG_RUN_WORDS, t)) eob_run_serial(c, w, sc, u, t, [&](uint32_t run) { atomicAdd(&eh[bitlen32(run) - 1], 1u); });
        else { uint32_t e = atomicAdd(c.long_cnt, 1u); c.long_runs[2 * e] = wi; c.long_runs[2 * e + 1] = u; }
    }
}
// long runs (flat regions, low-quality sources: a run can span a whole scan of 32 k blocks, and a single lane walking it held
// the kernel for a millisecond): the 64 lanes look for the end 4096 blocks at a time and cut the run 64 blocks at a time
// (wave prefix sum of the pending correction bits; the first lane over a limit ends the sub-run).
__global__ void __launch_bounds__(64) k_ac_runs_long(EncCtx c) {
    const uint32_t n = *c.long_cnt;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const ScanWork w = c.work[c.long_runs[2 * e]];
        const EncScan &sc = c.script[w.scan];
        const uint32_t u = c.long_runs[2 * e + 1];
        const uint64_t *sym = c.sym_bits + w.word_base;
        uint32_t *freq = c.tables[w.table_base].freq;
        auto count = [&](uint32_t run) { atomicAdd(&freq[(bitlen32(run) - 1) << 4], 1u); };
#ifdef CSH_EMUL
        uint32_t t;
        eob_run_end(sym, w.nunits, u, 0xFFFFFFFFu, t);
        eob_run_serial(c, w, sc, u, t, count);
#else
        const uint32_t lane = threadIdx.x, nwords = (w.nunits + 63) >> 6;
        uint32_t t = w.nunits - 1;
        for (uint32_t w0 = (u + 1) >> 6; w0 < nwords; w0 += 64) {   // 64 words = 4096 blocks per step
            uint64_t bits = w0 + lane < nwords ? sym[w0 + lane] : 0ull;
            if (w0 + lane == ((u + 1) >> 6)) bits &= ~0ull << ((u + 1) & 63);
            const uint64_t hit = __ballot(bits != 0);
            if (hit) {
                const int l0 = __ffsll((unsigned long long)hit) - 1;
                const uint32_t lo = uint32_t(__shfl(int(uint32_t(bits)), l0, 64)), hi = uint32_t(__shfl(int(uint32_t(bits >> 32)), l0, 64));
                const uint64_t b = (uint64_t(hi) << 32) | lo;
                const uint32_t p = ((w0 + uint32_t(l0)) << 6) + uint32_t(__ffsll((unsigned long long)b) - 1);
                if (p < w.nunits) t = p - 1;
                break;
            }
        }
        uint16_t *er = c.eobrun + w.unit_base;
        if (sc.Ah == 0) {
            if (lane == 0) { uint32_t L = t - u + 1, pos = u; while (L > 0) { uint32_t l = L < 0x7FFF ? L : 0x7FFF; er[pos] = uint16_t(l); count(l); pos += l; L -= l; } }
            continue;
        }
        const uint8_t *tl = c.tail + w.unit_base;
        uint32_t cnt = 0, be = 0, s0 = u, pos = u;
        while (pos <= t) {
            const uint32_t here = pos + lane <= t ? uint32_t(tl[pos + lane]) : 0u;
            uint32_t incl = here;
            CSH_UNROLL
            for (int o = 1; o < 64; o <<= 1) { uint32_t v = uint32_t(__shfl_up(int(incl), o, 64)); if (int(lane) >= o) incl += v; }
            const bool over = pos + lane <= t && (be + incl > 937 || cnt + lane + 1 == 0x7FFF);
            const uint64_t om = __ballot(over);
            if (om) {
                const uint32_t l0 = uint32_t(__ffsll((unsigned long long)om) - 1);
                if (lane == 0) { er[s0] = uint16_t(cnt + l0 + 1); count(cnt + l0 + 1); }
                s0 = pos + l0 + 1; pos = s0; cnt = 0; be = 0;
            } else {
                const uint32_t len = t - pos + 1 < 64 ? t - pos + 1 : 64;
                be += uint32_t(__shfl(int(incl), 63, 64)); cnt += len; pos += len;
            }
        }
        if (cnt && lane == 0) { er[s0] = uint16_t(cnt); count(cnt); }
#endif
    }
}

// ---- pass D: optimal Huffman tables (libjpeg jpeg_gen_optimal_table behaviour).
// The merge loop runs over the COMPACTED list of used symbols (ascending symbol order, pseudo-symbol 256 last), which
// preserves libjpeg's tie-breaking ("least frequency, ties to the larger symbol") while doing nnz^2 instead of 257*nnz work.
#ifdef CSH_EMUL
// emulation build: the plain serial form, one lane per table (the statement of the algorithm the wave kernel must match)
__global__ void k_gen_tables(DevEncTable *tables, int ntables) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntables) return;
    DevEncTable &T = tables[t];
    uint32_t freq[257];
    int16_t symof[257], codesize[257], others[257];
    uint8_t bits[33];
    int n = 0;
    for (int i = 0; i < 256; i++) { uint32_t f = T.freq[i]; if (f) { freq[n] = f; symof[n] = int16_t(i); n++; } }
    freq[n] = 1; symof[n] = 256; n++;   // reserved code point: guarantees no all-ones code
    for (int i = 0; i < n; i++) { codesize[i] = 0; others[i] = -1; }
    for (int i = 0; i < 33; i++) bits[i] = 0;
    for (;;) {
        int c1 = -1, c2 = -1;
        uint32_t v = 0xFFFFFFFFu;
        for (int i = 0; i < n; i++) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 0xFFFFFFFFu;
        for (int i = 0; i < n; i++) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2]; freq[c2] = 0;
        codesize[c1]++; while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = int16_t(c2);
        codesize[c2]++; while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i < n; i++) if (codesize[i]) bits[codesize[i] > 32 ? 32 : codesize[i]]++;
    for (int i = 32; i > 16; i--)
        while (bits[i] > 0) {
            int j = i - 2; while (bits[j] == 0) j--;
            bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
        }
    int i = 16; while (i > 0 && bits[i] == 0) i--;
    if (i > 0) bits[i]--;
    for (int l = 0; l <= 16; l++) T.bits[l] = l ? bits[l] : 0;
    int p = 0;
    for (int l = 1; l <= 32; l++) for (int s = 0; s < n - 1; s++) if (codesize[s] == l) T.vals[p++] = uint8_t(symof[s]);
    T.nsym = p;
    for (int s = 0; s < 256; s++) { T.size[s] = 0; T.code[s] = 0; }
    int code = 0; p = 0;
    for (int l = 1; l <= 16; l++) { for (int k2 = 0; k2 < T.bits[l]; k2++, p++) { T.code[T.vals[p]] = uint16_t(code++); T.size[T.vals[p]] = uint8_t(l); } code <<= 1; }
}
#else
// product build: one WAVE per table.  Entry e of the compacted list lives in lane e & 63, slot e >> 6 (registers).  A merge
// is two wave-wide arg-min reductions (key = freq << 32 | ~index: least frequency, ties to the larger index) and one
// data-parallel update: instead of walking libjpeg's `others` chain, every entry carries the id of the tree it belongs to
// and all entries of the two merged trees bump their code size at once -- the same code sizes, without the serial chain.
#define CSH_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ static uint64_t wave_min_u64(uint64_t v) {
    CSH_UNROLL
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(v)), o, 64)), hi = uint32_t(__shfl_xor(int(uint32_t(v >> 32)), o, 64));
        uint64_t x = (uint64_t(hi) << 32) | lo;
        v = x < v ? x : v;
    }
    return v;
}
__global__ void __launch_bounds__(256) k_gen_tables(DevEncTable *tables, int ntables) {
    __shared__ uint32_t s_freq[4][260];
    __shared__ uint16_t s_sym[4][260], s_grp[4][260], s_cs[4][260], s_code[4][256];
    __shared__ uint8_t s_size[4][256], s_vals[4][256];
    __shared__ uint32_t s_bits[4][34];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wv;
    if (t >= ntables) return;   // whole wave: only wave-level synchronisation is used below
    DevEncTable &T = tables[t];
    uint32_t *freq0 = s_freq[wv]; uint16_t *symof = s_sym[wv], *grp = s_grp[wv], *csz = s_cs[wv], *ocode = s_code[wv];
    uint8_t *osize = s_size[wv], *ovals = s_vals[wv];
    uint32_t *bits = s_bits[wv];
    const uint64_t lt = (1ull << lane) - 1ull;
    // compact the used symbols, ascending
    int n = 0;
    CSH_UNROLL
    for (int j = 0; j < 4; j++) {
        uint32_t f = T.freq[lane + 64 * j];
        uint64_t m = __ballot(f != 0);
        if (f) { int pos = n + __popcll(m & lt); freq0[pos] = f; symof[pos] = uint16_t(lane + 64 * j); }
        n += __popcll(m);
    }
    if (lane == 0) { freq0[n] = 1; symof[n] = 256; }   // reserved code point: guarantees no all-ones code
    n++;
    if (lane < 34) bits[lane] = 0;
    for (int i = lane; i < 256; i += 64) { ocode[i] = 0; osize[i] = 0; ovals[i] = 0; }
    CSH_WAVE_SYNC();
    uint32_t f[5]; int cs[5], g[5];
    CSH_UNROLL
    for (int j = 0; j < 5; j++) { int e = lane + 64 * j; f[j] = e < n ? freq0[e] : 0u; cs[j] = 0; g[j] = e; if (e < n) grp[e] = uint16_t(e); }
    CSH_WAVE_SYNC();
    for (;;) {
        uint64_t k1 = ~0ull;
        CSH_UNROLL
        for (int j = 0; j < 5; j++) { uint64_t k = (uint64_t(f[j]) << 32) | uint32_t(~uint32_t(lane + 64 * j)); if (f[j] && k < k1) k1 = k; }
        k1 = wave_min_u64(k1);
        const int c1 = int(~uint32_t(k1));
        uint64_t k2 = ~0ull;
        CSH_UNROLL
        for (int j = 0; j < 5; j++) { uint64_t k = (uint64_t(f[j]) << 32) | uint32_t(~uint32_t(lane + 64 * j)); if (f[j] && lane + 64 * j != c1 && k < k2) k2 = k; }
        k2 = wave_min_u64(k2);
        if (k2 == ~0ull) break;
        const int c2 = int(~uint32_t(k2));
        const uint32_t f2 = uint32_t(k2 >> 32);
        const int g1 = grp[c1], g2 = grp[c2];
        CSH_WAVE_SYNC();   // everyone has read the tree ids before they are rewritten
        CSH_UNROLL
        for (int j = 0; j < 5; j++) {
            const int e = lane + 64 * j;
            if (e == c1) f[j] += f2;
            if (e == c2) f[j] = 0;
            if (e < n && (g[j] == g1 || g[j] == g2)) { cs[j]++; if (g[j] != g1) { g[j] = g1; grp[e] = uint16_t(g1); } }
        }
        CSH_WAVE_SYNC();
    }
    // code-length counts (of every entry, the reserved one included), then libjpeg's length limiting
    CSH_UNROLL
    for (int j = 0; j < 5; j++) {
        const int e = lane + 64 * j;
        if (e < n) {
            csz[e] = uint16_t(cs[j]);
            if (cs[j]) atomicAdd(&bits[cs[j] > 32 ? 32 : cs[j]], 1u);
            if (e < n - 1 && cs[j] >= 1 && cs[j] <= 32) atomicAdd(&bits[33], 1u);   // listed symbols
        }
    }
    CSH_WAVE_SYNC();
    // order of the symbols: by code size, then by symbol (the reserved entry n-1 is not listed); sizes above 32 are not coded
    {
        int before[5] = {0, 0, 0, 0, 0};
        for (int q = 0; q < n - 1; q++) {
            const int cq = csz[q];
            CSH_UNROLL
            for (int j = 0; j < 5; j++) before[j] += (cq >= 1 && cq <= 32 && (cq < cs[j] || (cq == cs[j] && q < lane + 64 * j))) ? 1 : 0;
        }
        CSH_UNROLL
        for (int j = 0; j < 5; j++) { const int e = lane + 64 * j; if (e < n - 1 && cs[j] >= 1 && cs[j] <= 32) ovals[before[j]] = uint8_t(symof[e]); }
    }
    if (lane == 0) {
        for (int i = 32; i > 16; i--)
            while (bits[i] > 0) {
                int j = i - 2; while (bits[j] == 0) j--;
                bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
            }
        int i = 16; while (i > 0 && bits[i] == 0) i--;
        if (i > 0) bits[i]--;
    }
    CSH_WAVE_SYNC();
    const int nsym = int(bits[33]);
    // canonical codes for the listed symbols: position p has the length l with start[l] <= p < start[l] + bits[l]
    for (int p = lane; p < nsym; p += 64) {
        int code = 0, st = 0, len = 0, mine = 0;
        for (int l = 1; l <= 16; l++) {
            const int bl = int(bits[l]);
            if (!len && p < st + bl) { len = l; mine = code + (p - st); }
            code = (code + bl) << 1; st += bl;
        }
        if (len) { ocode[ovals[p]] = uint16_t(mine); osize[ovals[p]] = uint8_t(len); }
    }
    CSH_WAVE_SYNC();
    if (lane <= 16) T.bits[lane] = lane ? uint8_t(bits[lane]) : 0;
    if (lane == 0) T.nsym = nsym;
    for (int i = lane; i < 256; i += 64) { T.vals[i] = ovals[i]; T.code[i] = ocode[i]; T.size[i] = osize[i]; }
}
#endif
void launch_gen_tables(hipStream_t st, DevEncTable *tables, int ntables) {
#ifdef CSH_EMUL
    if (ntables) CSH_LAUNCH(k_gen_tables, dim3((ntables + 63) / 64), dim3(64), st, tables, ntables);
#else
    if (ntables) CSH_LAUNCH(k_gen_tables, dim3((ntables + 3) / 4), dim3(256), st, tables, ntables);
#endif
}

// ------------------------------------------------------------------------------------------------ tokens -> bits
// the scan's tables in LDS: ltab[t * 256 + s] = size << 16 | code
__device__ __forceinline__ static void stage_enc_tables(uint32_t *ltab, const DevEncTable *tab, int ntables) {
    for (int i = threadIdx.x; i < ntables * 256; i += blockDim.x) { const DevEncTable &T = tab[i >> 8]; ltab[i] = (uint32_t(T.size[i & 255]) << 16) | T.code[i & 255]; }
}
// one token through a bit sink (put(value, n bits)); `run` = the unit's EOBRUN (0: it owns none)
template <class Sink>
__device__ __forceinline__ static void token_bits(Sink &s, uint32_t t, const uint32_t *ltab, unsigned run) {
    const uint32_t kind = t & 3u;
    if (kind == TK_SYM) {
        const uint32_t e = ltab[((t >> 10) & 3u) * 256u + ((t >> 2) & 255u)];
        s.put(e & 0xFFFFu, int(e >> 16));
        s.put(t >> 16, int((t >> 12) & 15u));
    } else if (kind == TK_RAW) s.put(t >> 16, int((t >> 12) & 15u));
    else if (kind == TK_ACF) {
        const uint32_t r = (t >> 2) & 63u, nb = (t >> 8) & 15u;
        if (r >> 4) { const uint32_t z = ltab[0xF0]; for (uint32_t i = 0; i < (r >> 4); i++) s.put(z & 0xFFFFu, int(z >> 16)); }
        const uint32_t e = ltab[((r & 15u) << 4) | nb];
        s.put(e & 0xFFFFu, int(e >> 16));
        s.put((t >> 12) & 0xFFFFu, int(nb));
    } else if (run) {
        const int nb = bitlen32(run) - 1;
        const uint32_t e = ltab[nb << 4];
        s.put(e & 0xFFFFu, int(e >> 16));
        s.put(run & ((1u << nb) - 1u), nb);
    }
}
struct BitCount {
    uint32_t bits;
    __device__ __forceinline__ void put(unsigned, int n) { bits += uint32_t(n); }
};

// ---- pass E: size in bits of every chunk.  Flat over the chunk's tokens (coalesced), plus the EOBRUN symbols its units own.
__global__ void __launch_bounds__(256) k_chunk_sizes(EncCtx c) {
    CSH_SHARED uint32_t ltab[4 * 256];
    CSH_SHARED uint32_t s_sum;
    const uint32_t cs = blockIdx.x;
    const ScanWork w = c.work[c.slot_work[cs]];
    const EncScan &sc = c.script[w.scan];
    CSH_PHASE_LOOP(3) {
        if (phase == 0) { stage_enc_tables(ltab, c.tables + w.table_base, sc.ntables); if (threadIdx.x == 0) s_sum = 0; continue; }
        if (phase == 2) { if (threadIdx.x == 0) c.chunk_bits[cs] = s_sum; continue; }
        BitCount b; b.bits = 0;
        const uint32_t n = c.chunk_ntok[cs];
        const uint32_t *tk = c.tokens + c.tok_off[cs];
        for (uint32_t i = threadIdx.x; i < n; i += 256) token_bits(b, tk[i], ltab, 0u);
        const uint32_t u = (cs - w.first_chunk) * 256u + threadIdx.x;
        if (sc.Ss > 0 && !sc.sequential && u < w.nunits) {
            const unsigned run = c.eobrun[w.unit_base + u];
            if (run) { const int nb = bitlen32(run) - 1; b.bits += (ltab[nb << 4] >> 16) + uint32_t(nb); }
        }
        if (b.bits) atomicAdd(&s_sum, b.bits);
    }
}

// ---- pass G: pack.  Bits are gathered in a 64-bit accumulator and leave as whole big-endian-logical 32-bit words.  Only the first
// and the last word of a unit's bit string can be shared with a neighbouring unit, so only those two need an atomic OR; the words in
// between are exclusively this lane's and are stored plainly (the target is zero-initialised).  The target is the workgroup's LDS bit
// buffer, or -- chunks too large for it -- the raw pool itself.
struct PackSink {
    uint32_t *words;
    uint64_t pos;      // bit position (relative to words) of the next bit to emit
    uint64_t acc;      // pending bits, right-aligned
    int nacc;          // number of pending bits (< 32 after every put)
    bool first;        // the next word written is the unit's first (possibly shared) word
    __device__ __forceinline__ void begin(uint32_t *w, uint64_t p) { words = w; pos = p; nacc = int(p & 31); acc = 0; first = true; }
    __device__ __forceinline__ void put(unsigned v, int n) {
        if (n == 0) return;
        v &= (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
        acc = (acc << n) | v;
        nacc += n;
        pos += n;
        if (nacc >= 32) {
            uint32_t w = uint32_t(acc >> (nacc - 32));
            uint64_t wi = (pos - uint64_t(nacc)) >> 5;   // word that holds the oldest pending bit
            if (first) { if (w) atomicOr(words + wi, w); first = false; }
            else words[wi] = w;
            nacc -= 32;
            acc &= (nacc ? ((1ull << nacc) - 1ull) : 0ull);
        }
    }
    __device__ __forceinline__ void finish() {
        if (nacc == 0) return;
        uint32_t w = uint32_t(acc << (32 - nacc));
        uint64_t wi = (pos - uint64_t(nacc)) >> 5;
        if (w) atomicOr(words + wi, w);
        nacc = 0;
    }
};

__global__ void __launch_bounds__(256) k_pack(EncCtx c) {
    CSH_SHARED uint32_t ltab[4 * 256];
    CSH_SHARED uint32_t tstage[CSH_TK_STAGE];
    CSH_SHARED uint32_t bitbuf[CSH_PK_WORDS + 2];
    CSH_SHARED uint32_t a_cnt[256], a_first[256], a_size[256], a_off[256], w_tok[4], w_bit[4];
    const uint32_t cs = blockIdx.x;
    const ScanWork w = c.work[c.slot_work[cs]];
    const EncScan &sc = c.script[w.scan];
    const int tid = int(threadIdx.x), lane = lane_id(), wv = tid >> 6;
    const uint32_t nch = (w.nunits + 255u) / 256u, j = cs - w.first_chunk;
    const uint32_t u = j * 256u + uint32_t(tid);
    const bool valid = u < w.nunits;
    const uint32_t ntok_chunk = c.chunk_ntok[cs];
    const bool staged = ntok_chunk <= CSH_TK_STAGE;
    const uint32_t *tk = staged ? tstage : c.tokens + c.tok_off[cs];
    // the chunk's place: bits [raw_bit0, raw_bit0 + nbits) of the raw pool; the scan's last chunk also carries the 1-bits that fill the last byte
    const uint64_t scan0 = c.chunk_off[w.first_chunk];
    const uint64_t raw_bit0 = w.raw_off * 8 + (c.chunk_off[cs] - scan0);
    uint32_t nbits = c.chunk_bits[cs];
    int pad = 0;
    if (j == nch - 1) { const uint64_t total = c.chunk_off[w.first_chunk + nch] - scan0; pad = int((8 - (total & 7)) & 7); nbits += uint32_t(pad); }
    const bool in_lds = nbits <= CSH_PK_WORDS * 32u;
    const bool is_ac = sc.Ss > 0 && !sc.sequential;
    CSH_PHASE_LOOP(6) {
        if (w.no_room) { if (phase == 0 && tid == 0) c.status[w.image] = 20200; continue; }   // decided per scan by k_scan_place
        if (phase == 0) {
            stage_enc_tables(ltab, c.tables + w.table_base, sc.ntables);
            a_cnt[tid] = valid ? uint32_t(c.unit_ntok[w.unit_base + u]) : 0u;
            if (staged) { const uint32_t *src = c.tokens + c.tok_off[cs]; for (uint32_t i = uint32_t(tid); i < ntok_chunk; i += 256) tstage[i] = src[i]; }
            if (in_lds) for (uint32_t i = uint32_t(tid); i < (nbits + 31u) / 32u + 2u; i += 256) bitbuf[i] = 0;
            continue;
        }
        if (phase == 1) {   // first token of every unit: scan of the counts inside each wave ...
            const uint32_t incl = wave_incl_scan(a_cnt + 64 * wv, lane);
            a_first[tid] = incl - a_cnt[tid];
            if (lane == 63) w_tok[wv] = incl;
            continue;
        }
        if (phase == 2) {   // ... and across the waves; then the size of every unit
            uint32_t first = a_first[tid];
            for (int q = 0; q < wv; q++) first += w_tok[q];
            const bool sane = w_tok[0] + w_tok[1] + w_tok[2] + w_tok[3] == ntok_chunk;   // a pool that overflowed leaves stale counts behind: nothing is read then
            BitCount b; b.bits = 0;
            if (valid && sane) {
                const unsigned run = is_ac ? unsigned(c.eobrun[w.unit_base + u]) : 0u;
                const uint32_t n = a_cnt[tid];
                for (uint32_t i = 0; i < n; i++) token_bits(b, tk[first + i], ltab, run);
                if (u == w.nunits - 1) b.bits += uint32_t(pad);
            }
            a_first[tid] = sane ? first : 0xFFFFFFFFu;
            a_size[tid] = b.bits;
            continue;
        }
        if (phase == 3) {   // bit offsets: the same two steps
            const uint32_t incl = wave_incl_scan(a_size + 64 * wv, lane);
            a_off[tid] = incl - a_size[tid];
            if (lane == 63) w_bit[wv] = incl;
            continue;
        }
        if (phase == 4) {   // pack
            uint32_t off = a_off[tid];
            for (int q = 0; q < wv; q++) off += w_bit[q];
            const uint32_t first = a_first[tid];
            if (!valid || first == 0xFFFFFFFFu) continue;
            const uint32_t n = a_cnt[tid];
            const unsigned run = is_ac ? unsigned(c.eobrun[w.unit_base + u]) : 0u;
            PackSink s;
            if (in_lds) s.begin(bitbuf, uint64_t(off)); else s.begin(c.raw, raw_bit0 + off);
            for (uint32_t i = 0; i < n; i++) token_bits(s, tk[first + i], ltab, run);
            if (u == w.nunits - 1 && pad) s.put((1u << pad) - 1u, pad);
            s.finish();
            continue;
        }
        // phase 5: the LDS bit buffer moves to its place, shifted by the chunk's bit offset inside its first word
        if (!in_lds || nbits == 0) continue;
        const uint32_t sh = uint32_t(raw_bit0 & 31u);
        const uint64_t w0 = raw_bit0 >> 5;
        const uint32_t nout = (sh + nbits + 31u) >> 5;
        for (uint32_t i = uint32_t(tid); i < nout; i += 256) {
            const uint32_t lo = bitbuf[i], hi = i ? bitbuf[i - 1] : 0u;   // the buffer is zero behind the chunk's last word
            const uint32_t v = sh ? ((hi << (32u - sh)) | (lo >> sh)) : lo;
            if (i == 0 || i == nout - 1) { if (v) atomicOr(c.raw + w0 + i, v); }
            else c.raw[w0 + i] = v;
        }
    }
}

void launch_tokens(hipStream_t st, const EncCtx &c) { if (c.nechunks) CSH_LAUNCH_PHASED(k_tokens, 5, dim3(c.nechunks), dim3(256), st, c); }
void launch_ac_runs(hipStream_t st, const EncCtx &c) {
    if (!c.nslots) return;
    CSH_LAUNCH_PHASED(k_ac_runs, 3, dim3(c.nslots), dim3(256), st, c);
#ifdef CSH_EMUL
    CSH_LAUNCH(k_ac_runs_long, dim3(64), dim3(1), st, c);
#else
    CSH_LAUNCH(k_ac_runs_long, dim3(4096), dim3(64), st, c);
#endif
}
void launch_chunk_sizes(hipStream_t st, const EncCtx &c) { if (c.nslots) CSH_LAUNCH_PHASED(k_chunk_sizes, 3, dim3(c.nslots), dim3(256), st, c); }
void launch_pack(hipStream_t st, const EncCtx &c) { if (c.nslots) CSH_LAUNCH_PHASED(k_pack, 6, dim3(c.nslots), dim3(256), st, c); }

}  // namespace csh
