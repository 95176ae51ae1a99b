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
#include "wave.h"

namespace csh {

// ------------------------------------------------------------------------------------------------ tokens
//  kind (bits 0-2)
//   SYM  Huffman symbol + raw bits behind it:  [10:3] symbol  [12:11] table of the scan's group  [16:13] n raw bits (0..15)  [31:17] the bits
//   RAW  raw bits only:                        [16:13] n (0..15; 0 = a token that emits nothing)  [31:17] the bits
//   ACF  first-pass AC coefficient:            [8:3] zero run in front of it (0..62; every 16 cost one ZRL)  [12:9] size  [28:13] its bits
//   REF  refinement-scan event:                [10:3] unit inside the chunk  [14:11] zero run  [15] sign bit  [21:16] n correction bits behind it
//                                              [27:22] where they start in the unit's correction word  [28] 1 = the event is a ZRL (no sign bit)
//   EOB  the block ends with an EOB here:      [10:3] unit inside the chunk  [21:16] / [27:22] its trailing correction bits, as in REF;
//                                              the packer emits the unit's EOBRUN symbol in front of them (eobrun[unit], if it owns one)
enum : uint32_t { TK_SYM = 0u, TK_RAW = 1u, TK_ACF = 2u, TK_REF = 3u, TK_EOB = 4u };
#define CSH_PK_WORDS 1024   // the packer's window of the bit stream, per wave, in LDS words (a step of 256 tokens adds at most 768)

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

#ifndef CSH_EMUL
// inclusive scan over the 64 lanes of a wave with the cross-lane data path of the VALU (DPP: row shifts inside the rows of 16, then
// the two row broadcasts) -- six dependent VALU instructions, where a __shfl_up ladder is six LDS round trips.  Op: + or |
// (identity 0: lanes a step does not reach take `old` = 0).
template <class Op>
__device__ __forceinline__ static uint32_t wave_scan_dpp(uint32_t v, Op op) {
    v = op(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false)));   // row_shr:1
    v = op(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false)));   // row_shr:2
    v = op(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false)));   // row_shr:4
    v = op(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false)));   // row_shr:8
    v = op(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false)));   // row_bcast:15 into rows 1 and 3
    v = op(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false)));   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ static uint32_t wave_incl_sum(uint32_t v) { return wave_scan_dpp(v, [](uint32_t a, uint32_t b) { return a + b; }); }
// value of the last ACTIVE lane: the scan's total when the active lanes are lanes 0 .. m (a DPP step leaves a lane alone whose
// source lane is switched off, so a scan is valid exactly over such a prefix)
__device__ __forceinline__ static uint32_t wave_last(uint32_t v) { return uint32_t(__builtin_amdgcn_readlane(int(v), 63 - __clzll((unsigned long long)__ballot(1)))); }
#endif

// OR over the 64 lanes of a wave, in a scalar register pair (the emulation, where a lane cannot see the others, answers "all ones":
// callers use it only to skip work no lane has)
__device__ __forceinline__ static uint64_t wave_or64(uint64_t v) {
#ifdef CSH_EMUL
    (void)v;
    return ~0ull;
#else
    auto bor = [](uint32_t a, uint32_t b) { return a | b; };
    const uint32_t lo = wave_last(wave_scan_dpp(uint32_t(v), bor)), hi = wave_last(wave_scan_dpp(uint32_t(v >> 32), bor));
    return (uint64_t(hi) << 32) | lo;
#endif
}
// inclusive scan over the 64 lanes of a wave of the values in[0..63] (LDS, written in an earlier phase), for lane `lane`
__device__ __forceinline__ static uint32_t wave_incl_scan(const uint32_t *in, int lane) {
#ifdef CSH_EMUL
    uint32_t s = 0;
    for (int i = 0; i <= lane; i++) s += in[i] & 0xFFFFu;   // (k_tokens keeps another value in the high halves, written by the lanes that ran before this one)
    return s;
#else
    return wave_incl_sum(in[lane]);
#endif
}

// where a lane's tokens go: straight into its stretch of the pool
struct TokOut {
    uint32_t *pool;
    uint32_t pos;   // next token, relative to pool
    bool dry;
    __device__ __forceinline__ void put(uint32_t t) { if (!dry) pool[pos] = t; pos++; }
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
    uint32_t rawbits;   // raw bits emitted (EMIT)
    static constexpr bool kValues = true;
    __device__ __forceinline__ void begin() { n = 0; pend = 0; has_pend = false; }
    __device__ __forceinline__ void flush() { if (has_pend) { if (EMIT) out.put(pend); n++; has_pend = false; } }
    __device__ __forceinline__ void sym(int t, int s) {
        flush();
        pend = TK_SYM | (uint32_t(s) << 3) | (uint32_t(t) << 11); has_pend = true;
        if (EMIT) atomicAdd(&hist[(h0 + t) * 257 + s], 1u);
    }
    __device__ __forceinline__ void syms(int t, int s, int cnt) { for (int i = 0; i < cnt; i++) sym(t, s); }
    __device__ __forceinline__ void raw(unsigned v, int nb) {   // the nb low bits of v, most significant first
        if (EMIT) rawbits += uint32_t(nb);
        while (nb > 0) {
            int have = has_pend ? int((pend >> 13) & 15u) : 0;
            if (has_pend && have == 15) { flush(); have = 0; }
            if (!has_pend) { pend = TK_RAW; has_pend = true; }
            const int take = nb < 15 - have ? nb : 15 - have;
            const uint32_t piece = (v >> (nb - take)) & ((1u << take) - 1u);
            const uint32_t val = (((pend >> 17) << take) | piece) & 0x7FFFu;
            pend = (pend & 0x1FFFu) | (uint32_t(have + take) << 13) | (val << 17);
            nb -= take;
        }
    }
    __device__ __forceinline__ void finish() { flush(); }
};

// refinement scan of one block (jcphuff.c encode_mcu_AC_refine order: correction bits ride behind the next symbol).  No coefficient is
// read: H = positions with history, N = newly significant ones, C = their correction bits, S = signs.  One token per event (a newly
// significant coefficient, a ZRL, the EOB); the correction bits are not in the tokens: they sit, in stream order and left-aligned, in
// one 64-bit word per unit (at most 63 of them), and every token says which stretch of that word follows it.
// Tokens written: at most `room` (the closed-form count of the caller: popcount(N) + zeros/16 + EOB); the rest is filled with empty ones.
__device__ __forceinline__ static uint32_t refine_room(uint64_t H, uint64_t N, int Ss, bool ends_eob) {
    if (!N) return ends_eob ? 1u : 0u;
    const int eobpos = msb64(N);
    const int zeros = (eobpos - Ss + 1) - __popcll((H | N) & band_mask(Ss, eobpos));
    return uint32_t(__popcll(N)) + uint32_t(zeros >> 4) + (ends_eob ? 1u : 0u);
}
__device__ __forceinline__ static uint64_t correction_word(uint64_t H, uint64_t C) {
    uint64_t corr = 0;
    int i = 63;
    while (H) { const int k = __ffsll((unsigned long long)H) - 1; H &= H - 1; corr |= ((C >> k) & 1ull) << i; i--; }
    return corr;
}
__device__ static void emit_ac_refine(TokOut &out, uint32_t *hist, uint32_t unit, uint64_t H, uint64_t N, uint64_t S, int Ss, bool ends_eob, uint32_t room) {
    const uint32_t pos0 = out.pos;
    uint32_t cursor = 0;
    int prev = Ss - 1;
    uint64_t n = N;
    while (n) {
        const int k = __ffsll((unsigned long long)n) - 1;
        n &= n - 1;
        const uint64_t gap = (prev + 1 <= k - 1) ? band_mask(prev + 1, k - 1) : 0ull;
        uint32_t hc = uint32_t(__popcll(H & gap));
        int z = (k - prev - 1) - int(hc);
        if (z > 15) {
            // rare: ZRLs inside the gap.  Each is emitted at the first non-zero position after 16 more zeros and takes the correction bits
            // seen up to there
            int r = 0, p = prev;
            uint32_t pending = 0;
            uint64_t hm = (H & gap) | (1ull << k);
            while (hm) {
                const int q = __ffsll((unsigned long long)hm) - 1;
                hm &= hm - 1;
                r += q - p - 1; p = q;
                while (r > 15) {
                    out.put(TK_REF | (unit << 3) | (pending << 16) | (cursor << 22) | (1u << 28));
                    atomicAdd(&hist[0xF0], 1u);
                    cursor += pending; pending = 0; r -= 16;
                }
                if (q != k) pending++;
            }
            z = r; hc = pending;
        }
        out.put(TK_REF | (unit << 3) | (uint32_t(z) << 11) | (uint32_t((~S >> k) & 1ull) << 15) | (hc << 16) | (cursor << 22));
        atomicAdd(&hist[(z << 4) | 1], 1u);
        cursor += hc;
        prev = k;
    }
    if (ends_eob) out.put(TK_EOB | (unit << 3) | ((uint32_t(__popcll(H)) - cursor) << 16) | (cursor << 22));
    while (out.pos - pos0 < room) out.put(TK_RAW);
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
                // exchange the high-column half of row k with the low-column half of row k + j: two bit-field inserts
                const uint32_t a = A[k], b = A[k + j];
                A[k] = (a & msk) | ((b << j) & ~msk);
                A[k + j] = ((a >> j) & msk) | (b & ~msk);
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
// A workgroup is four independent waves of 64 units that share only the histogram: between the phases below a wave synchronises
// with itself alone (no s_barrier: a wave that waits for its memory does not hold the other three), except after the histogram is
// cleared and before it is flushed.  Each wave gets its own stretch of the token pool per slot ("segment": slot x 4 + wave).
// State of a lane across the phases: the planes.  The coefficients themselves are loaded again where the first-pass scans are coded
// (from the L2: the wave read the same lines a few microseconds earlier) -- 32 registers per lane held across the phases cost more
// in occupancy than the second load does.
__global__ void __launch_bounds__(256, 3) k_tokens(EncCtx c) {
    CSH_SHARED uint32_t hist[CSH_TK_MAXSLOT * 257];
    CSH_SHARED uint32_t cnt[CSH_TK_MAXSLOT][256];   // per (slot, lane): its tokens (low half: at most a few hundred) | the exclusive scan of them inside the lane's wave << 16
                                                     // (one array instead of two: 12 KB of LDS less; the kernel stays at 3 waves per SIMD for its 167 VGPRs -- at 128 it spills 39)
    CSH_SHARED uint32_t s_wtot[4][CSH_TK_MAXSLOT];  // per wave: tokens of the slot
    CSH_SHARED unsigned long long s_wbase[4][CSH_TK_MAXSLOT];   // per wave: first token of its segment in the pool (~0: no room)
    CSH_SHARED uint32_t s_raw[CSH_TK_MAXSLOT];      // raw (non-Huffman) bits of the slot, EOBRUN bits excluded
    CSH_SHARED TokPlan s_plan;                      // kind 0: the component's geometry and AC scans
    CSH_SHARED ScanWork s_w;                        // kind 1: the work item, its scan and its image (copied once: the walkers read them
    CSH_SHARED EncScan s_sc;                        //         field by field, and every read from HBM is a dependent scalar load)
    CSH_SHARED ImgDesc s_im;
    CSH_PERSIST(uint64_t, pl, 10);     // bit k of: |c_k| >= 1, 2, 4, 8, 16; bit 0..3 of |c_k|; c_k < 0
    const EChunk ch = c.echunks[blockIdx.x];
    const int tid = int(threadIdx.x), lane = lane_id(), wv = tid >> 6;
    const uint32_t u = ch.j * 256u + uint32_t(tid);
    const TokPlan &P = s_plan;

    // a conditional stage of the scan search codes only the images whose search asks for it: the others' workgroups have nothing to do
    const bool skipped = c.work_active && !c.work_active[ch.kind == 1 ? ch.a : c.plans[ch.plan].work0];
    CSH_PHASE_LOOP_MIXED(6, 0x0Eu) {   // after phases 1, 2, 3 only the wave synchronises
        if (skipped) continue;
        if ((c.debug & 1024u) && ch.kind == 1) continue;
        if ((c.debug & 2048u) && ch.kind == 0) continue;
        if (phase == 0) {
            for (int i = tid; i < CSH_TK_MAXSLOT * 257; i += 256) hist[i] = 0;
            if (tid < CSH_TK_MAXSLOT) s_raw[tid] = 0;
            if (ch.kind == 0 && tid < int(sizeof(TokPlan) / 4)) reinterpret_cast<uint32_t *>(&s_plan)[tid] = reinterpret_cast<const uint32_t *>(c.plans + ch.plan)[tid];
            if (ch.kind == 1) {
                const ScanWork &gw = c.work[ch.a];
                if (tid < int(sizeof(ScanWork) / 4)) reinterpret_cast<uint32_t *>(&s_w)[tid] = reinterpret_cast<const uint32_t *>(&gw)[tid];
                if (tid < int(sizeof(EncScan) / 4)) reinterpret_cast<uint32_t *>(&s_sc)[tid] = reinterpret_cast<const uint32_t *>(c.script + gw.scan)[tid];
                if (tid < int(sizeof(ImgDesc) / 4)) reinterpret_cast<uint32_t *>(&s_im)[tid] = reinterpret_cast<const uint32_t *>(c.imgs + gw.image)[tid];
            }
            continue;
        }
        if (phase == 1) {
            // ---------------------------------------------------------------- load, planes, counts, flags
            if (ch.kind == 1) {
                const ScanWork &w = s_w;
                const EncScan &sc = s_sc;
                const ImgDesc &im = s_im;
                uint32_t n = 0;
                if (u < w.nunits) {
                    if (sc.sequential) { TokSink<false> s; s.begin(); walk_seq(s, c, im, sc, u); s.finish(); n = s.n; }
                    else {   // DC scans: one SYM token per block (the difference's bits ride in it), or one bit per block, fifteen to a token
                        uint32_t nb = 0;
                        for (int ci = 0; ci < sc.ncomp; ci++) nb += sc.ncomp > 1 ? uint32_t(im.out[sc.comp[ci]].h * im.out[sc.comp[ci]].v) : 1u;
                        n = sc.Ah ? (nb + 14u) / 15u : nb;
                    }
                }
                cnt[0][tid] = n;
                continue;
            }
            const bool valid = u < P.nunits;
            CSH_UNROLL
            for (int i = 0; i < 10; i++) pl[i] = 0;
            if (valid) {
                const int by = int(u) / P.real_bw, b = by * P.bw + (int(u) - by * P.real_bw);
                const int16_t *p = c.coef + (size_t(P.tile_base) + size_t(b >> 6)) * CSH_TILE_I16 + size_t((b & 63) * CSH_BLK_STRIDE);
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
                if (!(c.debug & 512u)) transpose32(m);
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
            for (int slot = 0; slot < int(P.nslot); slot++) {
                const AcSlot &a = P.s[slot];
                const uint64_t band = band_mask(a.Ss, a.Se);
                const uint64_t lo = pick_sig(pl, a.Al);
                bool has_sym = false, ends_eob = false;
                uint32_t n = 0; int tail = 0;
                if (valid) {   // (a plan holds refinement scans only: the first-pass scans are coded from the compacted lists, k_aclist.hip)
                    const uint64_t hi = pick_sig(pl, a.Al + 1), H = hi & band, N = lo & ~hi & band;
                    has_sym = N != 0; ends_eob = !((N >> a.Se) & 1);
                    tail = N ? __popcll(H & ~((2ull << msb64(N)) - 1)) : __popcll(H);
                    n = refine_room(H, N, a.Ss, ends_eob);
                    if (!(c.debug & 256u)) c.corr[a.corr_base + u] = correction_word(H, pick_bit(pl + 5, a.Al));
                    c.tail[a.unit_base + u] = uint8_t(tail);
                }
                cnt[slot][tid] = n;
#ifdef CSH_EMUL
                if (has_sym) atomicOr(reinterpret_cast<unsigned long long *>(c.sym_bits + a.word_base + (u >> 6)), 1ull << (u & 63));
                if (ends_eob) atomicOr(reinterpret_cast<unsigned long long *>(c.eob_bits + a.word_base + (u >> 6)), 1ull << (u & 63));
#else
                // lane = block, so a wave's 64 flags ARE one word of the scan's bit vectors: one ballot, one 8-byte store
                const uint64_t ms = __ballot(has_sym), me = __ballot(ends_eob);
                if (lane == 0 && (u >> 6) < ((a.nunits_work + 63) >> 6)) { c.sym_bits[a.word_base + (u >> 6)] = ms; c.eob_bits[a.word_base + (u >> 6)] = me; }
#endif
            }
            continue;
        }
        if (phase == 2) {
            // ---------------------------------------------------------------- exclusive scan of the counts inside the wave, slot by slot
            const int nslot = ch.kind == 1 ? 1 : int(P.nslot);   // the slots above hold nothing (and are not read below)
            for (int slot = 0; slot < nslot; slot++) {
                const uint32_t incl = wave_incl_scan(&cnt[slot][64 * wv], lane);   // counts only: the high halves are still zero
                cnt[slot][tid] |= (incl - cnt[slot][tid]) << 16;
                if (lane == 63) s_wtot[wv][slot] = incl;
            }
            continue;
        }
        if (phase == 3) {
            // ---------------------------------------------------------------- one lane per wave: room in the pool for the wave's segments
            if (lane != 0) continue;
            uint32_t total = 0;
            const int nslot = ch.kind == 1 ? 1 : int(P.nslot);
            if (c.stats_only) {   // the trellis stage's statistics scans: nothing is written to the pool (phase 4 runs dry)
                for (int slot = 0; slot < nslot; slot++) s_wbase[wv][slot] = 0ull;
                continue;
            }
            for (int slot = 0; slot < nslot; slot++) total += s_wtot[wv][slot];
            const TokRegion rg = c.regions[ch.region];
            uint32_t rel;
            if (c.debug & 16384u) {   // timing experiment: a static share of the region instead of the cursor (tokens may collide)
                const uint32_t nch = ((ch.kind == 1 ? s_w.nunits : P.nunits) + 255u) >> 8;
                rel = (ch.j * 4u + uint32_t(wv)) * (rg.cap / (nch * 4u));
            } else rel = atomicAdd(&c.tok_cursor[ch.region], total);
            const bool ok = uint64_t(rel) + total <= rg.cap;
            const unsigned long long gb = rg.base + rel;
            if (!ok) c.overflow[1] = 1;
            unsigned long long at = gb;
            if (ch.kind == 1) {
                const ScanWork &w = s_w;
                const uint32_t seg = (w.first_chunk + ch.j) * 4u + uint32_t(wv);
                c.tok_off[seg] = gb; c.chunk_ntok[seg] = ok ? total : 0u;
                s_wbase[wv][0] = ok ? gb : ~0ull;
            } else {
                for (int slot = 0; slot < int(P.nslot); slot++) {
                    const uint32_t seg = (P.s[slot].first_chunk + ch.j) * 4u + uint32_t(wv);
                    c.tok_off[seg] = at; c.chunk_ntok[seg] = ok ? s_wtot[wv][slot] : 0u;
                    s_wbase[wv][slot] = ok ? at : ~0ull;
                    at += s_wtot[wv][slot];
                }
            }
            continue;
        }
        if (phase == 4) {
            // ---------------------------------------------------------------- tokens + histograms
            if (s_wbase[wv][0] == ~0ull) continue;
            if (c.debug & 1u) continue;
            TokOut out; out.pool = c.tokens; out.dry = (c.debug & 4u) != 0 || c.stats_only != 0;
            if (ch.kind == 1) {
                const ScanWork &w = s_w;
                const EncScan &sc = s_sc;
                if (u >= w.nunits) continue;
                TokSink<true> s; s.begin(); s.out = out; s.out.pool = c.tokens + s_wbase[wv][0] + (cnt[0][tid] >> 16); s.out.pos = 0; s.hist = hist; s.h0 = 0; s.rawbits = 0;
                if (sc.sequential) walk_seq(s, c, s_im, sc, u); else walk_dc(s, c, s_im, sc, u);
                s.finish();
                while (s.out.pos < (cnt[0][tid] & 0xFFFFu)) s.out.put(TK_RAW);
                if (s.rawbits) atomicAdd(&s_raw[0], s.rawbits);
                continue;
            }
            if (u >= P.nunits) continue;
            const uint64_t sgn = pl[9];
            for (int slot = 0; slot < int(P.nslot); slot++) {
                const AcSlot &a = P.s[slot];
                const uint64_t band = band_mask(a.Ss, a.Se);
                const uint64_t lo = pick_sig(pl, a.Al);
                out.pool = c.tokens + s_wbase[wv][slot] + (cnt[slot][tid] >> 16);
                out.pos = 0;
                const uint64_t hi = pick_sig(pl, a.Al + 1), H = hi & band, N = lo & ~hi & band;
                if (c.debug & 8u) continue;
                emit_ac_refine(out, hist + slot * 257, uint32_t(tid), H, N, sgn, a.Ss, !((N >> a.Se) & 1), cnt[slot][tid] & 0xFFFFu);
                const uint32_t rawbits = uint32_t(__popcll(N) + __popcll(H));   // a sign bit per new coefficient, a correction bit per old one
                if (rawbits) atomicAdd(&s_raw[slot], rawbits);
            }
            continue;
        }
        // -------------------------------------------------------------------- histograms -> tables, and per slot for k_chunk_sizes
        if (c.debug & 32u) continue;
        if (ch.kind == 1) {
            const ScanWork &w = s_w;
            const EncScan &sc = s_sc;
            const SlotRec &r = c.slots[w.first_chunk + ch.j];
            for (uint32_t i = uint32_t(tid); i < uint32_t(sc.ntables) * 256u; i += 256) {
                const uint32_t v = hist[(i >> 8) * 257u + (i & 255u)];
                c.slot_hist[size_t(r.hist_row) * 256u + i] = uint16_t(v);
                if (v) atomicAdd(&c.tables[w.table_base + (i >> 8)].freq[i & 255u], v);
            }
            if (tid == 0) c.slot_raw[w.first_chunk + ch.j] = s_raw[0];
        } else {
            for (int slot = 0; slot < int(P.nslot); slot++) {
                const AcSlot &a = P.s[slot];
                const uint32_t cs = a.first_chunk + ch.j;
                const uint32_t v = hist[slot * 257 + tid];
                if (!(c.debug & 64u)) c.slot_hist[size_t(c.slots[cs].hist_row) * 256u + uint32_t(tid)] = uint16_t(v);
                if (v && !(c.debug & 128u)) atomicAdd(&c.tables[a.table_base].freq[tid], v);
                if (tid == 0) c.slot_raw[cs] = s_raw[slot];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ---- pass B: EOB run structure -> EOBRUN value owned by the first block of each (sub-)run, and those symbols' statistics
// A run [u .. t] is cut into sub-runs as jcphuff.c does: after 0x7FFF blocks, and (refinement) as soon as more than
// MAX_CORR_BITS - DCTSIZE2 + 1 = 937 correction bits are pending.  Serial form (short runs, and the emulation build):
// count(run, owner): called for every EOBRUN symbol, with the block that emits it
template <class Count>
__device__ static void eob_run_serial(const EncCtx &c, const ScanWork &w, const EncScan &sc, uint32_t u, uint32_t t, Count count) {
    uint16_t *er = c.eobrun + w.unit_base;
    if (sc.Ah == 0) {
        uint32_t L = t - u + 1, pos = u;
        while (L > 0) { uint32_t l = L < 0x7FFF ? L : 0x7FFF; er[pos] = uint16_t(l); count(l, pos); pos += l; L -= l; }
        return;
    }
    const uint8_t *tl = c.tail + w.unit_base;
    uint32_t cnt = 0, be = 0, s0 = u;
    auto step = [&](uint32_t j, uint32_t tail_bits) {
        cnt++; be += tail_bits;
        if (cnt == 0x7FFF || be > 937) { er[s0] = uint16_t(cnt); count(cnt, s0); cnt = 0; be = 0; s0 = j + 1; }
    };
    uint32_t j = u;
    while (j <= t && (reinterpret_cast<uintptr_t>(tl + j) & 7)) { step(j, tl[j]); j++; }
    for (; j + 7 <= t; j += 8) {
        const uint64_t v = *reinterpret_cast<const uint64_t *>(tl + j);
        CSH_UNROLL
        for (int i = 0; i < 8; i++) step(j + i, uint32_t(v >> (8 * i)) & 255u);
    }
    for (; j <= t; j++) step(j, tl[j]);
    if (cnt) { er[s0] = uint16_t(cnt); count(cnt, s0); }
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
// One WAVE per slot (256 blocks: four per lane), four slots per workgroup -- the slots' chains of dependent loads (record, bit-vector
// words, counters) run side by side instead of one after the other.  The common case needs no loop and no further load: the block
// starts a run (it ends with an EOB and has symbols, or its predecessor does not end with one), the next block with a symbol lies in
// this or the next word of the has-symbol vector, and the run is at most 14 blocks long -- then no correction-bit limit can cut it
// (14 x 63 <= 937) and its EOBRUN is its length.  Everything else takes the general path (eob_run_end / eob_run_serial).
__global__ void __launch_bounds__(256) k_ac_runs(EncCtx c) {
    const int lane = lane_id();
    const uint32_t cs = c.slot0 + blockIdx.x * 4u + (threadIdx.x >> 6);
    if (cs >= c.slot0 + c.nslots) return;
    const SlotRec r = c.slots[cs];
    if (!(r.flags & 1u)) return;
    if (c.work_active && !c.work_active[r.work]) return;
    const uint32_t nunits = r.nunits_work;
    uint32_t *freq = c.tables[r.table_base].freq;
    const uint64_t *sym = c.sym_bits + r.word_base, *eob = c.eob_bits + r.word_base;
    uint32_t mine = 0;   // lane nb: EOBn symbols of ordinary runs counted so far
    (void)mine; (void)lane;
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const uint32_t u = r.j * 256u + uint32_t(i) * 64u + uint32_t(lane);
        int my_nb = -1;
        if (u < nunits) {
            const uint32_t w0 = u >> 6, nwords = (nunits + 63) >> 6;
            const int bit = int(u & 63);
            const uint64_t s0 = sym[w0], e0 = eob[w0];
            const uint64_t s1 = w0 + 1 < nwords ? sym[w0 + 1] : ~0ull;   // behind the scan's last block: a block "with a symbol" ends the run there
            const bool prev_eob = u == 0 ? false : (bit ? ((e0 >> (bit - 1)) & 1) != 0 : ((eob[w0 - 1] >> 63) & 1) != 0);
            const bool start = ((e0 >> bit) & 1) && (((s0 >> bit) & 1) || !prev_eob);
            if (start) {
                // next block with a symbol, looking at this word and the next one
                const uint64_t a0 = bit == 63 ? 0ull : (s0 & (~0ull << (bit + 1)));
                uint32_t next = 0xFFFFFFFFu;
                if (a0) next = (w0 << 6) + uint32_t(__ffsll((unsigned long long)a0) - 1);
                else if (s1) next = ((w0 + 1) << 6) + uint32_t(__ffsll((unsigned long long)s1) - 1);
                if (next != 0xFFFFFFFFu && next > nunits) next = nunits;
                const uint32_t len = next == 0xFFFFFFFFu ? 0xFFFFFFFFu : next - u;
                if (len <= 14u || (len < 0x7FFFu && r.Ah == 0)) {
                    c.eobrun[r.unit_base + u] = uint16_t(len);
                    my_nb = bitlen32(len) - 1;
                } else {
                    ScanWork w; w.unit_base = r.unit_base; w.word_base = r.word_base; w.nunits = nunits; w.first_chunk = r.first_chunk; w.table_base = r.table_base;
                    EncScan sc; sc.Ss = r.Ss; sc.Se = r.Se; sc.Ah = r.Ah; sc.Al = r.Al;
                    uint32_t t;
                    if (eob_run_end(sym, nunits, u, CSH_LONG_RUN_WORDS, t))
                        eob_run_serial(c, w, sc, u, t, [&](uint32_t run, uint32_t owner) {   // a run cut into sub-runs, or a long one: counted one by one
                            const int nb = bitlen32(run) - 1;
                            atomicAdd(&freq[nb << 4], 1u); atomicAdd(&c.slot_eobh[size_t(r.first_chunk + (owner >> 8)) * 16u + uint32_t(nb)], 1u);   // the symbol belongs to the chunk of the block that emits it
                        });
                    else { uint32_t e = atomicAdd(c.long_cnt, 1u); c.long_runs[2 * e] = r.work; c.long_runs[2 * e + 1] = u; }
                }
            }
        }
        // the EOBn symbol of an ordinary run (one per lane at most) is counted wave-wide: a ballot per class instead of 64 atomics on a few counters
#ifdef CSH_EMUL
        if (my_nb >= 0) { atomicAdd(&freq[my_nb << 4], 1u); atomicAdd(&c.slot_eobh[cs * 16u + uint32_t(my_nb)], 1u); }
#else
        CSH_UNROLL
        for (int nb = 0; nb < 4; nb++) {   // runs of up to 14 blocks: EOB0 .. EOB3; longer first-pass runs below
            const uint64_t m = __ballot(my_nb == nb);
            if (lane == nb) mine += uint32_t(__popcll(m));
        }
        if (__ballot(my_nb >= 4)) {
            for (int nb = 4; nb < 15; nb++) {
                const uint64_t m = __ballot(my_nb == nb);
                if (lane == nb) mine += uint32_t(__popcll(m));
            }
        }
#endif
    }
#ifndef CSH_EMUL
    if (mine && lane < 15) { atomicAdd(&freq[lane << 4], mine); atomicAdd(&c.slot_eobh[cs * 16u + uint32_t(lane)], mine); }
#endif
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
        auto count = [&](uint32_t run, uint32_t owner) {   // the symbol belongs to the chunk of the block that emits it
            const int nb = bitlen32(run) - 1;
            atomicAdd(&freq[nb << 4], 1u); atomicAdd(&c.slot_eobh[size_t(w.first_chunk + (owner >> 8)) * 16u + uint32_t(nb)], 1u);
        };
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
            if (lane == 0) { uint32_t L = t - u + 1, pos = u; while (L > 0) { uint32_t l = L < 0x7FFF ? L : 0x7FFF; er[pos] = uint16_t(l); count(l, pos); pos += l; L -= l; } }
            continue;
        }
        const uint8_t *tl = c.tail + w.unit_base;
        uint32_t cnt = 0, be = 0, s0 = u, pos = u;
        while (pos <= t) {
            const uint32_t here = pos + lane <= t ? uint32_t(tl[pos + lane]) : 0u;
            const uint32_t incl = wave_incl_sum(here);
            const bool over = pos + lane <= t && (be + incl > 937 || cnt + lane + 1 == 0x7FFF);
            const uint64_t om = __ballot(over);
            if (om) {
                const uint32_t l0 = uint32_t(__ffsll((unsigned long long)om) - 1);
                if (lane == 0) { er[s0] = uint16_t(cnt + l0 + 1); count(cnt + l0 + 1, s0); }
                s0 = pos + l0 + 1; pos = s0; cnt = 0; be = 0;
            } else {
                const uint32_t len = t - pos + 1 < 64 ? t - pos + 1 : 64;
                be += wave_last(incl); cnt += len; pos += len;
            }
        }
        if (cnt && lane == 0) { er[s0] = uint16_t(cnt); count(cnt, s0); }
#endif
    }
}

// ---- pass D: optimal Huffman tables (libjpeg jpeg_gen_optimal_table behaviour).
// The merge loop runs over the COMPACTED list of used symbols (ascending symbol order, pseudo-symbol 256 last), which
// preserves libjpeg's tie-breaking ("least frequency, ties to the larger symbol") while doing nnz^2 instead of 257*nnz work.
// One WAVE per table, written once for both builds (wave.h: LV<T> per-lane values, LFOR "for every lane"; the emulation plays the wave's lanes in a loop).
// Entry e of the compacted list lives in lane e & 63, slot e >> 6 (registers).  A merge is two wave-wide arg-min reductions (key = freq << 32 | ~index:
// least frequency, ties to the larger index) and one data-parallel update: instead of walking libjpeg's `others` chain, every entry carries the id of the
// tree it belongs to and all entries of the two merged trees bump their code size at once -- the same code sizes, without the serial chain.
__global__ void __launch_bounds__(4 * CSP_WAVE_THREADS) k_gen_tables(DevEncTable *tables, int ntables) {
    CSH_SHARED uint32_t s_freq[4][260];
    CSH_SHARED uint16_t s_sym[4][260], s_grp[4][260], s_cs[4][260], s_code[4][256];
    CSH_SHARED uint8_t s_size[4][256], s_vals[4][256];
    CSH_SHARED uint32_t s_bits[4][34];
    const int wv = int(threadIdx.x) / CSP_WAVE_THREADS;
    const int t = blockIdx.x * 4 + wv;
    if (t >= ntables) return;   // whole wave: only wave-level synchronisation is used below
    DevEncTable &T = tables[t];
    uint32_t *freq0 = s_freq[wv]; uint16_t *symof = s_sym[wv], *grp = s_grp[wv], *csz = s_cs[wv], *ocode = s_code[wv];
    uint8_t *osize = s_size[wv], *ovals = s_vals[wv];
    uint32_t *bits = s_bits[wv];
    // compact the used symbols, ascending
    int n = 0;
    for (int j = 0; j < 4; j++) {
        LV<uint32_t> fj;
        LFOR(l) fj[l] = T.freq[l + 64 * j];
        const uint64_t m = lballot([&](int l) { return fj[l] != 0; });
        LFOR(l) if (fj[l]) { const int pos = n + int(popc64(m & lanes_below(l))); freq0[pos] = fj[l]; symof[pos] = uint16_t(l + 64 * j); }
        n += int(popc64(m));
    }
    LFOR(l) if (l == 0) { freq0[n] = 1; symof[n] = 256; }   // reserved code point: guarantees no all-ones code
    n++;
    LFOR(l) {
        if (l < 34) bits[l] = 0;
        for (int i = l; i < 256; i += 64) { ocode[i] = 0; osize[i] = 0; ovals[i] = 0; }
    }
    CSP_WAVE_SYNC();
    LV<uint32_t> f[5], cs[5], g[5];
    for (int j = 0; j < 5; j++) LFOR(l) { const int e = l + 64 * j; f[j][l] = e < n ? freq0[e] : 0u; cs[j][l] = 0; g[j][l] = uint32_t(e); if (e < n) grp[e] = uint16_t(e); }
    CSP_WAVE_SYNC();
    for (;;) {
        LV<uint64_t> key;
        LFOR(l) {
            uint64_t k1 = ~0ull;
            for (int j = 0; j < 5; j++) { const uint64_t k = (uint64_t(f[j][l]) << 32) | uint32_t(~uint32_t(l + 64 * j)); if (f[j][l] && k < k1) k1 = k; }
            key[l] = k1;
        }
        const uint64_t k1 = lmin64(key);
        const int c1 = int(~uint32_t(k1));
        LFOR(l) {
            uint64_t k2 = ~0ull;
            for (int j = 0; j < 5; j++) { const uint64_t k = (uint64_t(f[j][l]) << 32) | uint32_t(~uint32_t(l + 64 * j)); if (f[j][l] && l + 64 * j != c1 && k < k2) k2 = k; }
            key[l] = k2;
        }
        const uint64_t k2 = lmin64(key);
        if (k2 == ~0ull) break;
        const int c2 = int(~uint32_t(k2));
        const uint32_t f2 = uint32_t(k2 >> 32);
        const uint32_t g1 = grp[c1], g2 = grp[c2];
        CSP_WAVE_SYNC();   // everyone has read the tree ids before they are rewritten
        for (int j = 0; j < 5; j++) LFOR(l) {
            const int e = l + 64 * j;
            if (e == c1) f[j][l] += f2;
            if (e == c2) f[j][l] = 0;
            if (e < n && (g[j][l] == g1 || g[j][l] == g2)) { cs[j][l]++; if (g[j][l] != g1) { g[j][l] = g1; grp[e] = uint16_t(g1); } }
        }
        CSP_WAVE_SYNC();
    }
    // code-length counts (of every entry, the reserved one included), then libjpeg's length limiting
    for (int j = 0; j < 5; j++) LFOR(l) {
        const int e = l + 64 * j;
        if (e < n) {
            csz[e] = uint16_t(cs[j][l]);
            if (cs[j][l]) atomicAdd(&bits[cs[j][l] > 32 ? 32 : cs[j][l]], 1u);
            if (e < n - 1 && cs[j][l] >= 1 && cs[j][l] <= 32) atomicAdd(&bits[33], 1u);   // listed symbols
        }
    }
    CSP_WAVE_SYNC();
    // order of the symbols: by code size, then by symbol (the reserved entry n-1 is not listed); sizes above 32 are not coded
    {
        LV<uint32_t> before[5];
        for (int j = 0; j < 5; j++) LFOR(l) before[j][l] = 0;
        for (int q = 0; q < n - 1; q++) {
            const uint32_t cq = csz[q];
            for (int j = 0; j < 5; j++) LFOR(l) before[j][l] += (cq >= 1 && cq <= 32 && (cq < cs[j][l] || (cq == cs[j][l] && q < l + 64 * j))) ? 1u : 0u;
        }
        for (int j = 0; j < 5; j++) LFOR(l) { const int e = l + 64 * j; if (e < n - 1 && cs[j][l] >= 1 && cs[j][l] <= 32) ovals[before[j][l]] = uint8_t(symof[e]); }
    }
    LFOR(l) if (l == 0) {
        for (int i = 32; i > 16; i--)
            while (bits[i] > 0) {
                int j = i - 2; while (bits[j] == 0) j--;
                bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
            }
        int i = 16; while (i > 0 && bits[i] == 0) i--;
        if (i > 0) bits[i]--;
    }
    CSP_WAVE_SYNC();
    const int nsym = int(bits[33]);
    // canonical codes for the listed symbols: position p has the length l with start[l] <= p < start[l] + bits[l]
    LFOR(lane) for (int p = lane; p < nsym; p += 64) {
        int code = 0, st = 0, len = 0, mine = 0;
        for (int l = 1; l <= 16; l++) {
            const int bl = int(bits[l]);
            if (!len && p < st + bl) { len = l; mine = code + (p - st); }
            code = (code + bl) << 1; st += bl;
        }
        if (len) { ocode[ovals[p]] = uint16_t(mine); osize[ovals[p]] = uint8_t(len); }
    }
    CSP_WAVE_SYNC();
    LFOR(lane) {
        if (lane <= 16) T.bits[lane] = lane ? uint8_t(bits[lane]) : 0;
        if (lane == 0) T.nsym = nsym;
        for (int i = lane; i < 256; i += 64) { T.vals[i] = ovals[i]; T.code[i] = ocode[i]; T.size[i] = osize[i]; T.lut[i] = (uint32_t(osize[i]) << 16) | ocode[i]; }
    }
}
void launch_gen_tables(hipStream_t st, DevEncTable *tables, int ntables) {
    if (ntables) CSH_LAUNCH(k_gen_tables, dim3((ntables + 3) / 4), dim3(4 * CSP_WAVE_THREADS), st, tables, ntables);
}

// ------------------------------------------------------------------------------------------------ tokens -> bits
// What one token puts into the stream: at most three pieces of at most 32 bits, in order (piece i: the n[i] low bits of v[i]).
struct Pieces { uint32_t v[3], n[3]; };
struct TokenCtx {                 // what the packer kernels know about the slot their tokens belong to
    const uint32_t *lut;          // the scan's tables: lut[t * lut_stride + s] = size << 16 | code   (DevEncTable::lut in HBM, or its copy in LDS)
    uint32_t lut_stride;
    const uint16_t *eobrun;       // of the chunk's units
    const uint64_t *corr;         // of the chunk's units (refinement scans)
};
#define CSH_LUT_STRIDE uint32_t(sizeof(DevEncTable) / 4)
__device__ __forceinline__ static void corr_pieces(Pieces &p, uint64_t corr, uint32_t t) {
    const uint32_t cnt = (t >> 16) & 63u, cur = (t >> 22) & 63u;
    if (!cnt) return;
    const uint64_t bits = (corr << cur) >> (64u - cnt);   // cnt >= 1, cur + cnt <= 63
    if (cnt > 32) { p.v[1] = uint32_t(bits >> 32); p.n[1] = cnt - 32; p.v[2] = uint32_t(bits); p.n[2] = 32; }
    else { p.v[2] = uint32_t(bits); p.n[2] = cnt; }
}
// MODE names what the scan's tokens can be, so that a wave does not walk through the code of kinds it cannot meet:
// 0 anything, 1 first-pass AC scan (ACF, EOB), 2 refinement scan (REF, EOB, empty RAW), 3 DC or sequential-mode scan (SYM, RAW)
template <int MODE = 0>
__device__ __forceinline__ static Pieces token_pieces(uint32_t t, const TokenCtx &x) {
    Pieces p;
    p.v[0] = p.v[1] = p.v[2] = 0; p.n[0] = p.n[1] = p.n[2] = 0;
    uint32_t kind = t & 7u;
    if (MODE == 1) { if (kind == TK_RAW) return p; kind = kind == TK_ACF ? uint32_t(TK_ACF) : uint32_t(TK_EOB); }   // RAW: the empty token behind a segment's end
    if (MODE == 2) { if (kind == TK_RAW) return p; kind = kind == TK_REF ? uint32_t(TK_REF) : uint32_t(TK_EOB); }
    if (MODE == 3) kind = kind == TK_SYM ? uint32_t(TK_SYM) : uint32_t(TK_RAW);
    if (kind == TK_SYM) {
        const uint32_t e = x.lut[((t >> 11) & 3u) * x.lut_stride + ((t >> 3) & 255u)], nr = (t >> 13) & 15u;
        p.v[0] = ((e & 0xFFFFu) << nr) | (t >> 17); p.n[0] = (e >> 16) + nr;            // <= 16 + 15 bits
    } else if (kind == TK_RAW) { p.v[0] = t >> 17; p.n[0] = (t >> 13) & 15u; }
    else if (kind == TK_ACF) {
        const uint32_t r = (t >> 3) & 63u, nb = (t >> 9) & 15u, zr = r >> 4;
        if (zr) {
            const uint32_t z = x.lut[0xF0], zc = z & 0xFFFFu, zl = z >> 16;
            p.v[0] = zc; p.n[0] = zl;
            if (zr > 1) { p.v[0] = (zc << zl) | zc; p.n[0] = 2 * zl; }
            if (zr > 2) { p.v[1] = zc; p.n[1] = zl; }
        }
        const uint32_t e = x.lut[((r & 15u) << 4) | nb];
        p.v[2] = ((e & 0xFFFFu) << nb) | ((t >> 13) & 0xFFFFu); p.n[2] = (e >> 16) + nb;   // <= 16 + 15 bits
    } else if (kind == TK_REF) {
        const uint32_t unit = (t >> 3) & 255u;
        if (t & (1u << 28)) { const uint32_t e = x.lut[0xF0]; p.v[0] = e & 0xFFFFu; p.n[0] = e >> 16; }
        else { const uint32_t e = x.lut[(((t >> 11) & 15u) << 4) | 1u]; p.v[0] = ((e & 0xFFFFu) << 1) | ((t >> 15) & 1u); p.n[0] = (e >> 16) + 1; }
        if ((t >> 16) & 63u) corr_pieces(p, x.corr[unit], t);
    } else {   // TK_EOB
        const uint32_t unit = (t >> 3) & 255u;
        const uint32_t run = x.eobrun[unit];
        if (run) {
            const int nb = bitlen32(run) - 1;
            const uint32_t e = x.lut[nb << 4];
            p.v[0] = ((e & 0xFFFFu) << nb) | (run & ((1u << nb) - 1u)); p.n[0] = (e >> 16) + uint32_t(nb);   // <= 16 + 14 bits
        }
        if ((t >> 16) & 63u) corr_pieces(p, x.corr[unit], t);
    }
    return p;
}
__device__ __forceinline__ static TokenCtx token_ctx(const EncCtx &c, const SlotRec &r) {
    TokenCtx x;
    x.lut = c.tables[r.table_base].lut; x.lut_stride = CSH_LUT_STRIDE;
    x.eobrun = c.eobrun + r.unit0;
    x.corr = c.corr + r.corr0;   // (read by refinement scans only)
    return x;
}
// four consecutive tokens of the chunk, starting at i (one 16-byte load where all four exist: the pool is 16-byte aligned per chunk only
// by chance, so the load is of single words)
__device__ __forceinline__ static void load4(const uint32_t *tk, uint32_t i, uint32_t n, uint32_t (&t)[4]) {
    CSH_UNROLL
    for (int q = 0; q < 4; q++) t[q] = i + uint32_t(q) < n ? tk[i + uint32_t(q)] : uint32_t(TK_RAW);
}

// ---- pass E: size in bits of every chunk: ONE WAVE per (scan, chunk) slot.  No token is read: the chunk's symbol counts (k_tokens kept
// them per slot), the code lengths, the raw bits k_tokens counted, and the EOBRUN symbols k_ac_runs counted per slot.
__global__ void __launch_bounds__(256) k_chunk_sizes(EncCtx c) {
    const uint32_t cs = c.slot0 + blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = lane_id();
    if (cs >= c.slot0 + c.nslots) return;
    const SlotRec r = c.slots[cs];
    if (c.work_active && !c.work_active[r.work]) { if (lane == 0) c.chunk_bits[cs] = 0; return; }   // a skipped scan has no bits
#ifdef CSH_EMUL
    if (lane) return;
    uint32_t bits = c.slot_raw[cs];
    for (uint32_t t = 0; t < r.ntables; t++)
        for (uint32_t sy = 0; sy < 256; sy++) bits += uint32_t(c.slot_hist[size_t(r.hist_row + t) * 256u + sy]) * c.tables[r.table_base + t].size[sy];
    if (r.flags & 1u) for (uint32_t nb = 0; nb < 15; nb++) bits += c.slot_eobh[cs * 16u + nb] * (uint32_t(c.tables[r.table_base].size[nb << 4]) + nb);
    c.chunk_bits[cs] = bits;
#else
    uint32_t bits = lane == 0 ? c.slot_raw[cs] : 0u;
    for (uint32_t t = 0; t < r.ntables; t++) {
        const uint2 h = reinterpret_cast<const uint2 *>(c.slot_hist + size_t(r.hist_row + t) * 256u)[lane];          // symbols 4 lane .. 4 lane + 3
        const uint32_t z = reinterpret_cast<const uint32_t *>(c.tables[r.table_base + t].size)[lane];
        bits += (h.x & 0xFFFFu) * (z & 255u) + (h.x >> 16) * ((z >> 8) & 255u) + (h.y & 0xFFFFu) * ((z >> 16) & 255u) + (h.y >> 16) * (z >> 24);
    }
    if ((r.flags & 1u) && lane < 15) bits += c.slot_eobh[cs * 16u + uint32_t(lane)] * (uint32_t(c.tables[r.table_base].size[lane << 4]) + uint32_t(lane));
    CSH_UNROLL
    for (int o = 32; o >= 1; o >>= 1) bits += uint32_t(__shfl_xor(int(bits), o, 64));
    if (lane == 0) c.chunk_bits[cs] = bits;
#endif
}

// ---- pass G: pack: ONE WAVE per slot.  256 tokens at a time, four per lane: their pieces, a wave scan of the lengths, every lane ORs
// its pieces into the wave's LDS window of the bit stream (the window is word-aligned with the raw pool, so flushing it is a plain
// copy: only the chunk's first and last word can be shared with a neighbouring chunk and need an atomic OR; the pool is
// zero-initialised).  The scan's tables, the units' EOBRUNs and correction words are staged in LDS first: everything a token needs
// is then one LDS read away.
__device__ __forceinline__ static void or_bits(uint32_t *words, uint64_t pos, uint32_t v, uint32_t n) {   // n in 1..32, at bit `pos` of a big-endian-logical word array
    v &= n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    const uint64_t t = uint64_t(v) << (64u - n - uint32_t(pos & 31u));
    const uint32_t hi = uint32_t(t >> 32), lo = uint32_t(t);
    if (hi) atomicOr(words + (pos >> 5), hi);
    if (lo) atomicOr(words + (pos >> 5) + 1, lo);
}
#define CSH_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// the common case of token_pieces: everything a token emits fits ONE piece of at most 32 bits (no ZRL in front of a first-pass coefficient,
// symbol + sign + correction bits of a refinement event within 32 bits).  Returns false when it does not: the caller then takes the
// three-piece form for the whole step.
template <int MODE>
__device__ __forceinline__ static bool token_main(uint32_t t, const TokenCtx &x, uint32_t &v, uint32_t &n) {
    v = 0; n = 0;
    const uint32_t kind = t & 7u;
    if (kind == TK_RAW) { v = t >> 17; n = (t >> 13) & 15u; return true; }
    if (MODE == 3) {   // SYM
        const uint32_t e = x.lut[((t >> 11) & 3u) * x.lut_stride + ((t >> 3) & 255u)], nr = (t >> 13) & 15u;
        v = ((e & 0xFFFFu) << nr) | (t >> 17); n = (e >> 16) + nr;
        return true;
    }
    if (MODE == 1 && kind == TK_ACF) {
        const uint32_t r = (t >> 3) & 63u, nb = (t >> 9) & 15u;
        if (r >> 4) return false;
        const uint32_t e = x.lut[(r << 4) | nb];
        v = ((e & 0xFFFFu) << nb) | ((t >> 13) & 0xFFFFu); n = (e >> 16) + nb;
        return true;
    }
    const uint32_t unit = (t >> 3) & 255u, cnt = (t >> 16) & 63u, cur = (t >> 22) & 63u;
    if (MODE == 2 && kind == TK_REF) {
        if (t & (1u << 28)) { const uint32_t e = x.lut[0xF0]; v = e & 0xFFFFu; n = e >> 16; }
        else { const uint32_t e = x.lut[(((t >> 11) & 15u) << 4) | 1u]; v = ((e & 0xFFFFu) << 1) | ((t >> 15) & 1u); n = (e >> 16) + 1; }
    } else {   // EOB
        const uint32_t run = x.eobrun[unit];
        if (run) {
            const int nb = bitlen32(run) - 1;
            const uint32_t e = x.lut[nb << 4];
            v = ((e & 0xFFFFu) << nb) | (run & ((1u << nb) - 1u)); n = (e >> 16) + uint32_t(nb);
        }
    }
    if (MODE == 2 && cnt) {
        if (n + cnt > 32) return false;
        const uint32_t bits = uint32_t((x.corr[unit] << cur) >> (64u - cnt));
        v = (n ? (v << cnt) : 0u) | bits; n += cnt;
    }
    return true;
}

#ifndef CSH_EMUL
struct PackState { uint32_t *buf, *out; uint64_t pos; uint32_t ww; bool first_flush; };
template <int MODE>
__device__ __forceinline__ static void pack_segments(const EncCtx &c, const TokenCtx &x, PackState &S, uint32_t cs, uint32_t pad, int lane) {
    uint32_t *buf = S.buf, *out = S.out;
    uint64_t pos = S.pos;
    uint32_t ww = S.ww;
    bool first_flush = S.first_flush;
    const uint32_t seg_n = lane < 4 ? c.chunk_ntok[cs * 4u + uint32_t(lane)] : 0u;              // the slot's four segments (one per wave of k_tokens)
    const unsigned long long seg_o = lane < 4 ? c.tok_off[cs * 4u + uint32_t(lane)] : 0ull;
    // 256 tokens, four per lane: pieces, wave scan of the lengths, OR into the window, slide the window when it fills
    auto step = [&](const uint32_t (&t)[4], uint32_t i, uint32_t n, bool last_seg) {
        // the one-piece form first: most steps have nothing else
        {
            uint32_t v4[4], n4[4];
            bool slow = false;
            CSH_UNROLL
            for (int q = 0; q < 4; q++) {
                slow |= !token_main<MODE>(t[q], x, v4[q], n4[q]);
                if (pad && last_seg && i + uint32_t(q) == n) { v4[q] = (1u << pad) - 1u; n4[q] = pad; }
            }
            if (!__ballot(slow) && !(c.debug & 4096u)) {
                const uint32_t len1 = n4[0] + n4[1] + n4[2] + n4[3];
                const uint32_t incl1 = wave_incl_sum(len1);
                uint64_t at1 = pos + incl1 - len1 - uint64_t(ww) * 32u;
                CSH_UNROLL
                for (int q = 0; q < 4; q++) if (n4[q]) { or_bits(buf, at1, v4[q], n4[q]); at1 += n4[q]; }
                pos += wave_last(incl1);
                goto slide;
            }
        }
        {
        Pieces p[4];
        uint32_t len = 0;
        CSH_UNROLL
        for (int q = 0; q < 4; q++) {
            p[q] = token_pieces<MODE>(t[q], x);
            if (pad && last_seg && i + uint32_t(q) == n) { p[q].v[0] = (1u << pad) - 1u; p[q].n[0] = pad; }   // the byte fill of the scan's last chunk rides as one more token
            len += p[q].n[0] + p[q].n[1] + p[q].n[2];
        }
        const uint32_t incl = wave_incl_sum(len);
        uint64_t at = pos + incl - len - uint64_t(ww) * 32u;    // bit position inside the window
        CSH_UNROLL
        for (int q = 0; q < 4; q++) {
            CSH_UNROLL
            for (int k = 0; k < 3; k++) if (p[q].n[k]) { or_bits(buf, at, p[q].v[k], p[q].n[k]); at += p[q].n[k]; }
        }
        pos += wave_last(incl);
        }
    slide:
        // slide the window when another 256 tokens' worth of bits (256 x 96) might not fit any more
        const uint32_t done = uint32_t(pos >> 5) - ww;   // complete words in the window
        if (done > CSH_PK_WORDS - 770) {
            CSH_WAVE_FENCE();
            for (uint32_t q = uint32_t(lane); q < done; q += 64) {
                const uint32_t v = buf[q];
                if (first_flush && q == 0) { if (v) atomicOr(out + ww, v); } else out[ww + q] = v;
            }
            const uint32_t partial = buf[done];
            CSH_WAVE_FENCE();
            for (int q = lane; q < CSH_PK_WORDS; q += 64) buf[q] = (q == 0) ? partial : 0u;
            CSH_WAVE_FENCE();
            ww += done; first_flush = false;
        }
    };
    // the four segments are walked as ONE list (most are far shorter than a step: a step per segment would run with three quarters
    // of the lanes idle): token g of the list lies in segment (g >= b1) + (g >= b2) + (g >= b3)
    uint32_t n[4]; const uint32_t *tk[4];
    CSH_UNROLL
    for (int seg = 0; seg < 4; seg++) {
        n[seg] = uint32_t(__builtin_amdgcn_readlane(int(seg_n), seg));
        tk[seg] = c.tokens + ((unsigned long long)uint32_t(__builtin_amdgcn_readlane(int(uint32_t(seg_o)), seg)) | ((unsigned long long)uint32_t(__builtin_amdgcn_readlane(int(uint32_t(seg_o >> 32)), seg)) << 32));
    }
    const uint32_t b1 = n[0], b2 = b1 + n[1], b3 = b2 + n[2], total = b3 + n[3];
    const uint32_t n_ext = total + (pad ? 1u : 0u);
    for (uint32_t g0 = 0; g0 < n_ext; g0 += 256) {
        uint32_t t[4];
        CSH_UNROLL
        for (int q = 0; q < 4; q++) {
            const uint32_t g = g0 + 4u * uint32_t(lane) + uint32_t(q);
            const uint32_t *src = g >= b3 ? tk[3] + (g - b3) : g >= b2 ? tk[2] + (g - b2) : g >= b1 ? tk[1] + (g - b1) : tk[0] + g;
            t[q] = g < total ? *src : uint32_t(TK_RAW);
        }
        step(t, g0 + 4u * uint32_t(lane), total, true);
    }
    S.pos = pos; S.ww = ww; S.first_flush = first_flush;
}
#endif
__global__ void __launch_bounds__(256) k_pack(EncCtx c) {
    const int wv = int(threadIdx.x >> 6), lane = lane_id();
    if (blockIdx.x * 4u + uint32_t(wv) >= c.ntok_slots) return;
    const uint32_t cs = c.tok_slots[blockIdx.x * 4u + uint32_t(wv)];   // the run's slots that are packed from tokens (the others: k_aclist.hip)
    const SlotRec r = c.slots[cs];
    const ScanWork &w = c.work[r.work];
    if (c.work_active && !c.work_active[r.work]) return;
    if (w.no_room) { if (lane == 0) c.status[w.image] = 20200; return; }   // decided per scan by k_scan_place
    TokenCtx x = token_ctx(c, r);
    // the chunk's place: bits [raw_bit0, raw_bit0 + nbits) of the raw pool; the scan's last chunk also carries the 1-bits that fill the last byte
    const uint64_t scan0 = c.chunk_off[r.first_chunk];
    const uint64_t raw_bit0 = w.raw_off * 8 + (c.chunk_off[cs] - scan0);
    uint32_t pad = 0;
    if (r.j == r.nch - 1) { const uint64_t total = c.chunk_off[r.first_chunk + r.nch] - scan0; pad = uint32_t((8 - (total & 7)) & 7); }
#ifdef CSH_EMUL
    // the same pieces, one after the other, straight into the pool
    if (lane) return;
    uint64_t pos = raw_bit0;
    for (uint32_t seg = 0; seg < 4; seg++) {
        const uint32_t n = c.chunk_ntok[cs * 4u + seg];
        const uint32_t *tk = c.tokens + c.tok_off[cs * 4u + seg];
        for (uint32_t i = 0; i < n; i++) {
            const Pieces p = token_pieces(tk[i], x);
            for (int q = 0; q < 3; q++) if (p.n[q]) { or_bits(c.raw, pos, p.v[q], p.n[q]); pos += p.n[q]; }
        }
    }
    if (pad) or_bits(c.raw, pos, (1u << pad) - 1u, pad);
#else
    __shared__ uint32_t win[4][CSH_PK_WORDS];
    __shared__ uint32_t s_lut[4][2][256];
    __shared__ uint16_t s_eob[4][256];
    __shared__ uint64_t s_corr[4][256];
    uint32_t *buf = win[wv];
    // stage (uniform branches: one scan per wave).  A scan with more than two tables (sequential mode) keeps its look-ups in HBM; the two
    // cases are separate instantiations below so that every look-up has ONE address space (a pointer that may be either makes them flat loads)
    const bool in_lds = r.ntables <= 2;
    if (in_lds) {
        for (int i = lane; i < int(r.ntables) * 256; i += 64) s_lut[wv][i >> 8][i & 255] = x.lut[(i >> 8) * CSH_LUT_STRIDE + (i & 255)];
        if (r.flags & 1u) for (int i = lane; i < 256; i += 64) s_eob[wv][i] = uint32_t(i) < r.nun ? x.eobrun[i] : uint16_t(0);
        if (r.flags & 2u) for (int i = lane; i < 256; i += 64) s_corr[wv][i] = uint32_t(i) < r.nun ? x.corr[i] : 0ull;
    }
    TokenCtx xl;   // the same, out of LDS
    xl.lut = &s_lut[wv][0][0]; xl.lut_stride = 256; xl.eobrun = &s_eob[wv][0]; xl.corr = &s_corr[wv][0];
    for (int i = lane; i < CSH_PK_WORDS; i += 64) buf[i] = 0;
    uint32_t *out = c.raw + (raw_bit0 >> 5);        // word 0 of the frame below
    uint64_t pos = raw_bit0 & 31u;                   // next bit, in the frame whose word 0 is the chunk's first word in the pool
    uint32_t ww = 0;                                 // first word of the window
    bool first_flush = true;
    CSH_WAVE_FENCE();
    PackState S; S.buf = buf; S.out = out; S.pos = pos; S.ww = ww; S.first_flush = first_flush;
    if (!in_lds) pack_segments<3>(c, x, S, cs, pad, lane);
    else if (!(r.flags & 1u)) pack_segments<3>(c, xl, S, cs, pad, lane);
    else if (r.flags & 2u) pack_segments<2>(c, xl, S, cs, pad, lane);
    else pack_segments<1>(c, xl, S, cs, pad, lane);
    pos = S.pos; ww = S.ww; first_flush = S.first_flush;
    CSH_WAVE_FENCE();
    const uint32_t last = uint32_t((pos + 31) >> 5) - ww;   // words in the window that carry bits
    for (uint32_t q = uint32_t(lane); q < last; q += 64) {
        const uint32_t v = buf[q];
        if ((first_flush && q == 0) || q == last - 1) { if (v) atomicOr(out + ww + q, v); } else out[ww + q] = v;
    }
#endif
}

// ---- before the pack: the words two chunks may share -- every chunk's first and last -- are the only ones that are ORed into, so they
// are the only ones that must start at zero (the pool itself is not cleared: under the scan search it is ten files' worth per file)
__global__ void __launch_bounds__(256) k_zero_edges(EncCtx c) {
    const uint32_t cs = c.slot0 + blockIdx.x * 256u + threadIdx.x;
    if (cs >= c.slot0 + c.nslots) return;
    const SlotRec r = c.slots[cs];
    const ScanWork &w = c.work[r.work];
    if (w.no_room || (c.work_active && !c.work_active[r.work])) return;
    const uint64_t scan0 = c.chunk_off[r.first_chunk];
    const uint64_t bit0 = w.raw_off * 8 + (c.chunk_off[cs] - scan0);
    const uint64_t nbits = c.chunk_bits[cs];
    c.raw[bit0 >> 5] = 0;
    // the chunk's last word (ORed by the packer even when the chunk ends on a word boundary) and the first one behind it; the scan's
    // last chunk also carries up to 7 bits of byte fill
    const uint64_t lo = nbits ? (bit0 + nbits - 1) >> 5 : bit0 >> 5, hi = (bit0 + nbits + (r.j == r.nch - 1 ? 8 : 0)) >> 5;
    for (uint64_t wi = lo; wi <= hi; wi++) c.raw[wi] = 0;
}
void launch_zero_edges(hipStream_t st, const EncCtx &c) { if (c.nslots) CSH_LAUNCH(k_zero_edges, dim3((c.nslots + 255) / 256), dim3(256), st, c); }
void launch_tokens(hipStream_t st, const EncCtx &c) { if (c.nechunks) CSH_LAUNCH_PHASED(k_tokens, 6, dim3(c.nechunks), dim3(256), st, c); }
void launch_ac_runs(hipStream_t st, const EncCtx &c) {
    if (!c.nslots) return;
    CSH_LAUNCH(k_ac_runs, dim3((c.nslots + 3) / 4), dim3(256), st, c);
#ifdef CSH_EMUL
    CSH_LAUNCH(k_ac_runs_long, dim3(64), dim3(1), st, c);
#else
    CSH_LAUNCH(k_ac_runs_long, dim3(4096), dim3(64), st, c);
#endif
}
void launch_chunk_sizes(hipStream_t st, const EncCtx &c) { if (c.nslots) CSH_LAUNCH(k_chunk_sizes, dim3((c.nslots + 3) / 4), dim3(256), st, c); }
void launch_pack(hipStream_t st, const EncCtx &c) { if (c.ntok_slots) CSH_LAUNCH(k_pack, dim3((c.ntok_slots + 3) / 4), dim3(256), st, c); }

}  // namespace csh
