// k_trellis.hip -- mozjpeg's trellis quantiser on the device (CSH_PROFILE=mozjpeg).
//
// Replaces mozjpeg's jcdctmgr.c quantize_trellis as jccoefct.c compress_trellis_pass drives it for libcaesium's `-q N` path
// (reference call sites /root/reference/src/compressor.rs:415,427 -> libcaesium 0.20.3 -> mozjpeg-sys 2.2.1,
// /root/reference/Cargo.lock:1035-1044; the source is not under /root/reference -- the statement these kernels are checked against,
// bit for bit, is oracle/jpeg_oracle.c quantize_trellis_row, [UPSTREAM-RECALL], parity with the real crate UNPINNED).
//
// Per component: a statistics scan over the scalar-quantised coefficients (k_tokens in its stats-only form + k_ac_runs + k_gen_tables)
// has left the optimal AC table's code lengths; then
//   k_trellis_ac   lane = block.  The block's unquantised DCT (retained by the pixel kernels) is swept once, in zig-zag order, in
//                  registers: lambda from the block's AC energy, the running sum Z of "distortion if dropped", and one LIST ENTRY per
//                  coefficient whose scalar level is not zero (the only positions the dynamic programme can keep).  The programme then
//                  runs over list entries, not positions: step t gives entry t its cheapest (predecessor entry, candidate level) --
//                  the t-th entry has exactly t + 1 possible predecessors whatever the block, so the inner loop bound is the same for
//                  all 64 lanes of a wave and only the candidate count (1 + log2 level) differs.  Lists live in LDS, entry-major
//                  ([entry][lane]: every access of a step is one conflict-free row); entries past CSH_TR_CAP spill to a per-workgroup
//                  stretch of HBM (dense blocks: high qualities).  Then the cheapest last coefficient, the path back, the block.
//   k_trellis_dc   lane = one iMCU row of a component: the Viterbi path over up to 9 DC levels per block along each row of blocks
//                  (81 transitions per block, all in registers), back-pointers in a side array, the path written back.
// All cost arithmetic is float in mozjpeg's order of operations (no contraction: the Makefile passes -ffp-contract=off); lambda is
// double arithmetic rounded once, as in the C source.  No MFMA: the programme is a data-dependent minimisation, not a contraction.
#include <utility>

#include "kernels.h"

namespace csh {

#define CSH_TR_CAP 16                      // list entries per block that live in LDS (192 bytes per block)
#define CSH_TR_WGU uint32_t(CSH_TR_WG)     // blocks (= lanes) per workgroup of the AC kernel: kernels.h
#define CSH_TR_SPILL (63 - CSH_TR_CAP)     // the rest, in HBM
#define CSH_TR_MAXWG 2048                  // workgroups of the AC kernel (each loops over its share of the chunks)
size_t trellis_spill_words() { return size_t(CSH_TR_MAXWG) * CSH_TR_SPILL * 3u * CSH_TR_WGU; }

#ifdef CSH_EMUL
#define CSH_ANY(p) (p)                     // a lane cannot see the others there: its own loop bounds
#define CSH_WAVE_OR(v) (v)
#define CSH_SPILL_LD(ptr) (*(ptr))
#define CSH_SPILL_ST(ptr, v) (*(ptr) = (v))
#else
#define CSH_ANY(p) (__ballot(p) != 0ull)   // wave-uniform loop conditions
__device__ __forceinline__ static unsigned tr_wave_or(unsigned v) {   // a value with the bit length of the OR over the wave's active lanes (all its users take the bit length): the largest bit length, found in four ballots (ten, one per bit, before round 5)
    const unsigned n = 32u - unsigned(__clz(v));   // 0..10: levels have ten bits
    unsigned r = 0;
    if (__ballot(n >= 8u)) r = 8u;
    if (__ballot(n >= r + 4u)) r += 4u;
    if (__ballot(n >= r + 2u)) r += 2u;
    if (__ballot(n >= r + 1u)) r += 1u;
    return r ? 1u << (r - 1u) : 0u;
}
#define CSH_WAVE_OR(v) tr_wave_or(v)
// the spilled entries are read and written as streaming accesses: besides the hint, that keeps them from being merged with the LDS
// accesses of the other branch into one generic-address (flat) access -- which is what the compiler made of `e < CAP ? lds : hbm`,
// three flat loads with a full wait per predecessor
#define CSH_SPILL_LD(ptr) __builtin_nontemporal_load(ptr)
#define CSH_SPILL_ST(ptr, v) __builtin_nontemporal_store((v), (ptr))
#endif

using Oct8 = std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>;
// natural index -> zig-zag index (the energy sum runs in natural order, as the C source's loop over the JBLOCK does)
static constexpr uint8_t kN2Z[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                     41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                     46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
template <int I>
__device__ __forceinline__ static int tr_half(const uint4 &v) {
    uint32_t w = (I / 2 == 0) ? v.x : (I / 2 == 1) ? v.y : (I / 2 == 2) ? v.z : v.w;
    return (I & 1) ? (int(w) >> 16) : (int(w << 16) >> 16);
}
template <int J, int... I>
__device__ __forceinline__ static void tr_unpack(int r[64], const uint4 &v, std::integer_sequence<int, I...>) { ((r[8 * J + I] = tr_half<I>(v)), ...); }
template <int... J>
__device__ __forceinline__ static void tr_load(const int16_t *__restrict__ blk, int r[64], std::integer_sequence<int, J...>) {   // zig-zag order
    const uint4 v[8] = {*reinterpret_cast<const uint4 *>(blk + CSH_RAW_OCT * J)...};
    (tr_unpack<J>(r, v[J], Oct8()), ...);
}
// exact (x + d / 2) / d for 0 <= x <= 2^15, d = 8 q: the pixel kernels' quantiser -- two full-rate instructions (types.h DevQuant::mul)
__device__ __forceinline__ static int tr_level(int x, int d, uint32_t mul, uint32_t sh) {
    const uint32_t a = uint32_t(x + (d >> 1)) << sh;
    return int(uint32_t((uint64_t(a & 0xFFFFFFu) * uint64_t(mul & 0xFFFFFFu)) >> 32));   // v_mul_hi_u32_u24
}
__device__ __forceinline__ static int tr_bitlen(unsigned v) { return 32 - __clz(v); }
__device__ __forceinline__ static float tr_bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
__device__ __forceinline__ static uint32_t tr_f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

#define TRELLIS_LAMBDA_C1 0x1.ae89f995ad3adp+14   // pow(2.0, lambda_log_scale1 = 14.75)
#define TRELLIS_LAMBDA_C2 0x1.6a09e667f3bcdp+16   // pow(2.0, lambda_log_scale2 = 16.5)
#define TRELLIS_MAX_LEVEL 1023                    // (1 << MAX_COEF_BITS) - 1

struct TrLds {
    float (*A)[CSH_TR_WG];      // [entry][lane]: cost of the cheapest path that ends with this entry
    float (*Z)[CSH_TR_WG];      // before the entry's step: Z just in front of its position; after it: Z at its position
    uint32_t (*P)[CSH_TR_WG];   // before: |DCT| (15) | position << 15 (6) | scalar level << 21 (10) | sign << 31;  after: predecessor entry + 1 (6) | position << 15 | chosen level << 21 | sign << 31
    const float *lenf;    // what an AC symbol costs, as the float the C source converts the rate to: code length + size bits (the symbol's low nibble);
                          // 1e38 for a symbol the statistics pass never saw (its code length is 0: mozjpeg skips such candidates -- at that cost none can win)
    const float *runf;    // [4] what 0..3 ZRLs in front of a symbol cost; 1e38 where ZRL has no code
    int lenEOB;           // plain code length of the end-of-block symbol
    const int32_t *q8;    // 8 q, zig-zag order
    const uint32_t *qmul, *qsh;   // the exact-division pair of 8 q
    const float *lt;      // 1 / q^2
    const float *rcp;     // the one-fma quantiser's reciprocal of 8 q (DevQuant::rcp)
};

// tables of the chunk's component -> LDS (all 256 lanes)
// (the sweep reads the quantiser's three values per position from here: as loads from HBM they were 189 dependent round trips per wave)
__device__ __forceinline__ static void trellis_stage(const TrellisCtx &c, uint32_t chi, float *s_lenf, float *s_runf, int *s_eob, int32_t *s_q8, uint32_t *s_qmul, uint32_t *s_qsh, float *s_lt, float *s_rcp) {
    const TrellisWork &w = c.work[c.chunks[chi].work];
    const ImgDesc &im = c.imgs[w.image];
    const DevQuant &Q = c.quant[im.qt_out[w.comp]];
    const int tid = int(threadIdx.x);
    const uint8_t *size = c.tables[w.table_ac].size;
    // s_lenf[11 r + s - 1] = what symbol r << 4 | s (s = 1..10) costs.  Rows of 11: the candidate loop reads row (zero run & 15) at every
    // predecessor, and with rows of 16 the runs 0, 2, 4 .. met in one bank (42 % of the kernel's LDS cycles were bank conflicts)
    for (int i = tid; i < 16 * 11; i += CSH_TR_WG) {
        const int r = i / 11, sz = i % 11 + 1, l = sz <= 10 ? size[(r << 4) | sz] : 0;
        s_lenf[i] = l ? float(l + sz) : 1e38f;
    }
    if (tid < 4) s_runf[tid] = tid == 0 ? 0.0f : (size[0xF0] ? float(tid * int(size[0xF0])) : 1e38f);
    if (tid == 0) *s_eob = size[0x00];
    if (tid < 64) { s_q8[tid] = Q.div[tid]; s_qmul[tid] = Q.mul[tid]; s_qsh[tid] = Q.sh[tid]; s_lt[tid] = Q.lt[tid]; s_rcp[tid] = Q.rcp[tid]; }
}

__device__ __forceinline__ static void trellis_block(const TrellisCtx &c, uint32_t chi, uint32_t wg_slot, const TrLds &L) {
    const TrellisChunk ch = c.chunks[chi];
    const TrellisWork &w = c.work[ch.work];
    const ImgDesc &im = c.imgs[w.image];
    const CompGeom g = im.out[w.comp];
    const int tid = int(threadIdx.x);
    const uint32_t slot = ch.j * CSH_TR_WGU + uint32_t(tid);
    if (slot >= w.nunits) return;
    const uint32_t u = c.perm ? c.perm[w.unit_base + slot] : slot;   // the blocks of a chunk: neighbours in the order of list length (k_trellis_sort), or in raster order
    const int by = int(u) / g.real_bw, b = by * g.bw + (int(u) - by * g.real_bw);
    uint32_t *sp = c.spill + size_t(wg_slot) * (CSH_TR_SPILL * 3u * CSH_TR_WGU) + uint32_t(tid);   // entry e >= CAP: sp[((e - CAP) * 3 + {0 A, 1 Z, 2 P}) * WG]

    int r[64];
    tr_load(c.raw + raw_index(g.tile_base - c.raw_tile0, b), r, Oct8());
    CSH_SCHED_FENCE();
    // lambda: the block's mean squared AC value (float accumulation in natural order), two roundings from double as in the C source
    float norm = 0.0f;
    CSH_UNROLL
    for (int n = 1; n < 64; n++) { const float vf = float(r[kN2Z[n]]); norm = norm + vf * vf; }   // (the float product of the converted value IS the converted integer square: both are the exact square rounded once)
    norm = float(double(norm) / 63.0);
    const float lambda = float(TRELLIS_LAMBDA_C1 / (TRELLIS_LAMBDA_C2 + double(norm)));
    c.dcrec[w.unit_base + u] = (uint64_t(tr_f_bits(lambda)) << 32) | (uint32_t(r[0]) & 0xFFFFu);   // what the DC kernel needs of this block, in one load
    CSH_SCHED_FENCE();
    if (c.debug & 1u) return;

    // ---- sweep: Z, the list of positions whose scalar level is not zero
    uint32_t ne = 0;
    float Zrun = 0.0f;
    CSH_UNROLL
    for (int k = 1; k < 64; k++) {
        const int v = r[k], x = v < 0 ? -v : v;
        const float xf = float(x);
        int qv = int(tr_f_bits(__builtin_fmaf(xf, L.rcp[k], 12582912.0f)) & 0xFFFFu);   // (x + d / 2) / d exactly: the pixel kernels' one-fma quantiser (k_pixel.hip quant_one, DevQuant::rcp)
        qv = qv > TRELLIS_MAX_LEVEL ? TRELLIS_MAX_LEVEL : qv;
        if (qv) {
            const uint32_t P = uint32_t(x) | (uint32_t(k) << 15) | (uint32_t(qv) << 21) | (v < 0 ? 0x80000000u : 0u);
            if (ne < CSH_TR_CAP) { L.Z[ne][tid] = Zrun; L.P[ne][tid] = P; }
            else { CSH_SPILL_ST(sp + ((ne - CSH_TR_CAP) * 3u + 1u) * CSH_TR_WGU, tr_f_bits(Zrun)); CSH_SPILL_ST(sp + ((ne - CSH_TR_CAP) * 3u + 2u) * CSH_TR_WGU, P); }
            ne++;
        }
        Zrun = ((xf * xf) * lambda) * L.lt[k] + Zrun;
    }
    const float Z63 = Zrun;
    CSH_SCHED_FENCE();

    // ---- the programme over list entries.  Every candidate's cost is the C source's float expression, operation for operation:
    // (float(rate) + dist) + ((Z in front - Z at the predecessor) + the predecessor's path cost), rate = symbol bits + size + ZRL bits --
    // float(rate) as the sum of two small exact floats (symbol cost, ZRL cost).  What mozjpeg skips (a symbol or a ZRL without a code, a
    // candidate the level does not offer) is priced at 1e38 instead: such a sum is never below the running minimum, which starts at 1e38.
    const float lenEOBf = float(L.lenEOB);
    if (c.debug & 2u) ne = 0;
    for (uint32_t t = 0; CSH_ANY(t < ne); t++) {
        const bool on = t < ne;
        float Zp; uint32_t P;
        if (t < CSH_TR_CAP) { Zp = L.Z[t][tid]; P = L.P[t][tid]; }
        else { Zp = tr_bits_f(CSH_SPILL_LD(sp + ((t - CSH_TR_CAP) * 3u + 1u) * CSH_TR_WGU)); P = CSH_SPILL_LD(sp + ((t - CSH_TR_CAP) * 3u + 2u) * CSH_TR_WGU); }
        if (!on) { Zp = 0.0f; P = 0u; }
        const int x = int(P & 0x7FFFu), kpos = int((P >> 15) & 63u), qval = int((P >> 21) & 1023u);
        const int q8 = L.q8[kpos];
        const float ltk = L.lt[kpos];
        const int ncand = on ? tr_bitlen(unsigned(qval)) : 0;
        // most candidates any lane of the wave has at this step -- the (uniform) bound of the candidate loops: the bit length of the OR of the levels
        const int ncmax = tr_bitlen(CSH_WAVE_OR(on ? unsigned(qval) : 0u));
        float dist[10];
        CSH_UNROLL
        for (int kc = 0; kc < 10; kc++) {
            dist[kc] = 1e38f;
            if (kc >= ncmax) continue;   // (uniform)
            const int cand = kc < ncand - 1 ? (2 << kc) - 1 : qval;
            const int delta = __mul24(cand, q8) - x;   // |delta| <= 2^15: the full-rate 24-bit multiplier is exact (v_mul_lo_u32 is quarter rate)
            dist[kc] = kc < ncand ? (float(__mul24(delta, delta)) * lambda) * ltk : 1e38f;
        }
        float bestc = 1e38f;
        uint32_t bestsel = 0;   // (predecessor entry + 1) << 4 | candidate
        // one predecessor: entry jj (position posj, path cost Aj, Z at its position Zj), or jj = -1: the start of the block
        auto from = [&](int jj, int posj, float Aj, float Zj) {
            const int zr = kpos - 1 - posj;
            const float runf = L.runf[(zr >> 4) & 3];
            const float *lf = L.lenf + 11 * (zr & 15);   // symbols (zr & 15) << 4 | 1 .. : in range whatever zr is (a lane that is not `on` reads valid, unused floats)
            const float tj = (Zp - Zj) + Aj;
            const uint32_t sel0 = uint32_t(jj + 1) << 4;
            CSH_UNROLL
            for (int kc = 0; kc < 10; kc++) {
                if (kc >= ncmax) break;
                const float cost = ((lf[kc] + runf) + dist[kc]) + tj;
                const bool better = cost < bestc;
                bestc = better ? cost : bestc;
                bestsel = better ? (sel0 | uint32_t(kc)) : bestsel;
            }
        };
        from(-1, 0, 0.0f, 0.0f);
        const int t_lds = int(t) < CSH_TR_CAP ? int(t) : CSH_TR_CAP;
        // The predecessors in LDS.  One predecessor per round trip -- read its entry, then the table row its zero run selects, then compare -- is
        // a chain of two LDS latencies per predecessor.  So, while no lane of the wave has more than four candidates (levels below 16), four
        // predecessors at a time: their twelve values, then the sixteen table entries, then the compares -- in the order of the plain loop,
        // predecessor by predecessor, candidate by candidate (ties go to the first).  A row at or above t is read and priced at 1e38; a candidate
        // the level does not offer has dist 1e38: neither can win.
        if (ncmax <= 4) {
            // (one instantiation per candidate count of the step, 1..4: with levels 1 and 2..3 -- most steps -- half of the four-candidate form's adds and compares
            // priced candidates no lane has)
            auto batch = [&](auto KCV) {
                constexpr int KC = decltype(KCV)::value;
                for (int j0 = 0; j0 < t_lds; j0 += 4) {
                    uint32_t Pj[4]; float Aj[4], Zj[4];
                    CSH_UNROLL
                    for (int i = 0; i < 4; i++) { const int r = j0 + i < CSH_TR_CAP ? j0 + i : CSH_TR_CAP - 1; Pj[i] = L.P[r][tid]; Aj[i] = L.A[r][tid]; Zj[i] = L.Z[r][tid]; }
                    float cost[4][KC];
                    CSH_UNROLL
                    for (int i = 0; i < 4; i++) {
                        const int zr = kpos - 1 - int((Pj[i] >> 15) & 63u);
                        const float runf = L.runf[(zr >> 4) & 3];
                        const float *lf = L.lenf + 11 * (zr & 15);
                        const float tj = j0 + i < t_lds ? (Zp - Zj[i]) + Aj[i] : 1e38f;
                        CSH_UNROLL
                        for (int kc = 0; kc < KC; kc++) cost[i][kc] = ((lf[kc] + runf) + dist[kc]) + tj;
                    }
                    CSH_UNROLL
                    for (int i = 0; i < 4; i++) {
                        CSH_UNROLL
                        for (int kc = 0; kc < KC; kc++) {
                            const bool better = cost[i][kc] < bestc;
                            bestc = better ? cost[i][kc] : bestc;
                            bestsel = better ? ((uint32_t(j0 + i + 1) << 4) | uint32_t(kc)) : bestsel;
                        }
                    }
                }
            };
            if (ncmax <= 1) batch(std::integral_constant<int, 1>());
            else if (ncmax == 2) batch(std::integral_constant<int, 2>());
            else if (ncmax == 3) batch(std::integral_constant<int, 3>());
            else batch(std::integral_constant<int, 4>());
        } else
            for (int jj = 0; jj < t_lds; jj++) from(jj, int((L.P[jj][tid] >> 15) & 63u), L.A[jj][tid], L.Z[jj][tid]);
        for (int jj = CSH_TR_CAP; jj < int(t); jj++) {
            const uint32_t *q = sp + (uint32_t(jj - CSH_TR_CAP) * 3u) * CSH_TR_WGU;
            from(jj, int((CSH_SPILL_LD(q + 2 * CSH_TR_WG) >> 15) & 63u), tr_bits_f(CSH_SPILL_LD(q)), tr_bits_f(CSH_SPILL_LD(q + CSH_TR_WG)));
        }
        if (on) {
            const int bk = int(bestsel & 15u);
            const uint32_t level = uint32_t(bk < ncand - 1 ? (2 << bk) - 1 : qval);
            const float Zi = (float(__mul24(x, x)) * lambda) * ltk + Zp;
            // (bit 14: the chosen level IS the scalar level -- what the coefficient tile already holds: the write-back leaves such a coefficient alone;
            // not at the largest level, which the sweep may have cut the scalar level down to)
            const uint32_t P2 = (bestsel >> 4) | ((level == uint32_t(qval) && qval < TRELLIS_MAX_LEVEL) ? 0x4000u : 0u) | (uint32_t(kpos) << 15) | (level << 21) | (P & 0x80000000u);
            if (t < CSH_TR_CAP) { L.A[t][tid] = bestc; L.Z[t][tid] = Zi; L.P[t][tid] = P2; }
            else { uint32_t *q = sp + ((t - CSH_TR_CAP) * 3u) * CSH_TR_WGU; CSH_SPILL_ST(q, tr_f_bits(bestc)); CSH_SPILL_ST(q + CSH_TR_WG, tr_f_bits(Zi)); CSH_SPILL_ST(q + 2 * CSH_TR_WG, P2); }
        }
    }

    // ---- the cheapest last coefficient
    float best = Z63 + lenEOBf;
    int last = -1;
    for (uint32_t e = 0; CSH_ANY(e < ne); e++) {
        float Ae, Ze; uint32_t Pe;
        if (e < CSH_TR_CAP) { Ae = L.A[e][tid]; Ze = L.Z[e][tid]; Pe = L.P[e][tid]; }
        else { const uint32_t *q = sp + ((e - CSH_TR_CAP) * 3u) * CSH_TR_WGU; Ae = tr_bits_f(CSH_SPILL_LD(q)); Ze = tr_bits_f(CSH_SPILL_LD(q + CSH_TR_WG)); Pe = CSH_SPILL_LD(q + 2 * CSH_TR_WG); }
        float cost = (Ae + Z63) - Ze;
        if (int((Pe >> 15) & 63u) < 63) cost = cost + lenEOBf;
        if (e < ne && cost < best) { best = cost; last = int(e); }
    }

    if (c.debug & 4u) return;
    // ---- the block.  It holds the scalar quantiser's output (the pixel kernels wrote it: zeros wherever the list has no entry, the scalar DC,
    // which k_trellis_dc replaces): only the list's positions change -- the level chosen on the path back from the last coefficient, zero off
    // the path.  (Writing the block anew -- zeros, DC, levels -- was 128 bytes a block and, with the blocks in order of list length, most of
    // what that order cost: scattered 16-byte stores.)
    uint64_t kept = 0;   // entries on the path
    for (int e = last; CSH_ANY(e >= 0);) {
        if (e >= 0) {
            uint32_t Pe;
            if (e < CSH_TR_CAP) Pe = L.P[e][tid]; else Pe = CSH_SPILL_LD(sp + (uint32_t(e - CSH_TR_CAP) * 3u + 2u) * CSH_TR_WGU);
            kept |= 1ull << e;
            e = int(Pe & 63u) - 1;
        }
    }
    int16_t *dst = c.coef + coef_index(g.tile_base, b, 0);
    // the block's entries in the statistics scan's level-0 list (the same coefficients in the same order: both are "scalar level not zero", in
    // zig-zag order): they take the chosen levels, so that the coding stages' lists are filtered from this list (k_nzfilter) and the
    // coefficient tiles are not swept a second time
    uint32_t *lst = nullptr;
    if (c.nz_pool && w.nzset != 0xFFFFFFFFu) {
        const NzSet &S = c.nzsets[w.nzset];
        const NzList &L0 = c.nzlists[S.list[0]];
        const uint32_t rec = L0.chunk0 + (u >> 8);
        if (c.nz_chunk_cnt[rec]) lst = c.nz_pool + L0.base + c.nz_chunk_off[rec] + c.blk_off[w.unit_base + u];   // (a chunk that found no room has no entries: the run is repeated with larger pools)
    }
    // the list was laid out from the statistics scan's count of this block's non-zero scalar levels; the programme's own count must be that number (the same
    // quantiser on the same coefficients).  Should the two ever part -- a change to either quantiser -- the entries are not written past the block's room
    // (ADVICE r05); the emulation build stops on it
    uint32_t room = ne;
    if (lst && c.blk_cnt) {
        room = c.blk_cnt[w.unit_base + u];
#ifdef CSH_EMUL
        if (room != ne) { fprintf(stderr, "k_trellis_ac: block %u of unit %u has %u list entries but %u levels\n", u, w.unit_base, room, ne); abort(); }
#endif
    }
    // (the list entries four at a time: one 16-byte store where the block has four more -- a 4-byte store per entry and lane was 0.9 ms per 1024 files)
    for (uint32_t e0 = 0; CSH_ANY(e0 < ne); e0 += 4) {
        uint32_t ent[4] = {0u, 0u, 0u, 0u};
        CSH_UNROLL
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t e = e0 + i;
            if (e < ne) {
                uint32_t Pe;
                if (e < CSH_TR_CAP) Pe = L.P[e][tid]; else Pe = CSH_SPILL_LD(sp + ((e - CSH_TR_CAP) * 3u + 2u) * CSH_TR_WGU);
                const int pos = int((Pe >> 15) & 63u), level = ((kept >> e) & 1ull) ? int((Pe >> 21) & 1023u) : 0;
                if (!(((kept >> e) & 1ull) && (Pe & 0x4000u))) dst[coef_off(pos)] = int16_t((Pe >> 31) ? -level : level);   // kept at the scalar level: the tile holds it
                ent[i] = uint32_t(pos) | ((Pe >> 31) ? 128u : 0u) | (uint32_t(level) << 8) | ((u & 255u) << 23);
            }
        }
        if (lst && e0 < ne && e0 < room) {
            if (e0 + 4 <= ne && e0 + 4 <= room) {
#ifdef CSH_EMUL
                memcpy(lst + e0, ent, 16);
#else
                typedef uint32_t tr_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
                tr_u32x4_a4 v; v.x = ent[0]; v.y = ent[1]; v.z = ent[2]; v.w = ent[3];
                *reinterpret_cast<tr_u32x4_a4 *>(lst + e0) = v;
#endif
            } else {
                CSH_UNROLL
                for (uint32_t i = 0; i < 4; i++) if (e0 + i < ne && e0 + i < room) lst[e0 + i] = ent[i];
            }
        }
    }
}

// The blocks of a component in order of list length (counting sort over 64 keys, one workgroup per component; which of two equally long
// blocks comes first is left to the atomics -- the order only decides which lanes run side by side, never a block's result).
__global__ void __launch_bounds__(256) k_trellis_sort(TrellisCtx c) {
    CSH_SHARED uint32_t s_cnt[64];
    const TrellisWork w = c.work[blockIdx.x];
    const uint8_t *cnt = c.blk_cnt + w.unit_base;
    uint32_t *perm = c.perm + w.unit_base;
    CSH_PHASE_LOOP(4) {
        if (phase == 0) { if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0; continue; }
        if (phase == 1) { for (uint32_t u = threadIdx.x; u < w.nunits; u += 256) atomicAdd(&s_cnt[cnt[u] & 63u], 1u); continue; }
        if (phase == 2) {   // exclusive prefix, longest lists first: the chunks at the front are the long-running ones
            if (threadIdx.x == 0) { uint32_t at = 0; for (int k = 63; k >= 0; k--) { const uint32_t n = s_cnt[k]; s_cnt[k] = at; at += n; } }
            continue;
        }
        for (uint32_t u = threadIdx.x; u < w.nunits; u += 256) perm[atomicAdd(&s_cnt[cnt[u] & 63u], 1u)] = u;
    }
}
void launch_trellis_sort(hipStream_t st, const TrellisCtx &c) { if (c.nwork && c.perm && c.blk_cnt) CSH_LAUNCH_PHASED(k_trellis_sort, 4, dim3(unsigned(c.nwork)), dim3(256), st, c); }

__global__ void __launch_bounds__(CSH_TR_WG) k_trellis_ac(TrellisCtx c) {
    CSH_SHARED float s_A[CSH_TR_CAP][CSH_TR_WG];
    CSH_SHARED float s_Z[CSH_TR_CAP][CSH_TR_WG];
    CSH_SHARED uint32_t s_P[CSH_TR_CAP][CSH_TR_WG];
    CSH_SHARED float s_lenf[16 * 11 + 16];   // + 16: the candidate loop's reads past a row's tenth entry stay inside
    CSH_SHARED float s_runf[4];
    CSH_SHARED int s_eob;
    CSH_SHARED int32_t s_q8[64];
    CSH_SHARED uint32_t s_qmul[64];
    CSH_SHARED uint32_t s_qsh[64];
    CSH_SHARED float s_lt[64];
    CSH_SHARED float s_rcp[64];
    TrLds L; L.A = s_A; L.Z = s_Z; L.P = s_P; L.lenf = s_lenf; L.runf = s_runf; L.q8 = s_q8; L.qmul = s_qmul; L.qsh = s_qsh; L.lt = s_lt; L.rcp = s_rcp;
#ifdef CSH_EMUL
    CSH_PHASE_LOOP(2) {
        if (phase == 0) { trellis_stage(c, blockIdx.x, s_lenf, s_runf, &s_eob, s_q8, s_qmul, s_qsh, s_lt, s_rcp); continue; }
        L.lenEOB = s_eob;
        trellis_block(c, blockIdx.x, 0u, L);
    }
#else
    for (uint32_t chi = blockIdx.x; chi < c.nchunks; chi += gridDim.x) {
        trellis_stage(c, chi, s_lenf, s_runf, &s_eob, s_q8, s_qmul, s_qsh, s_lt, s_rcp);
        __syncthreads();
        L.lenEOB = s_eob;
        trellis_block(c, chi, blockIdx.x, L);
        __syncthreads();
    }
#endif
}

// ------------------------------------------------------------------------------------------------ DC
// T.81 Tables K.3 / K.4 code lengths (what jpeg_set_defaults installs): in progressive mode no DC statistics exist when the trellis
// passes run, so the DC path is priced with these
__device__ static const uint8_t kStdDcLen[2][12] = {{2, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9}, {2, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}};

__global__ void __launch_bounds__(64) k_trellis_dc(TrellisCtx c) {
    // one lane per iMCU row of any component of any image (TrellisCtx::rows: longest rows first, so that a wave's 64 walks are of a length:
    // a grid of (rows of a component) x (components) left most waves a quarter full)
    const uint32_t ri = blockIdx.x * blockDim.x + threadIdx.x;
    if (ri >= c.nrows) return;
    const uint32_t rr = c.rows[ri];
    const TrellisWork w = c.work[rr >> 16];
    const ImgDesc &im = c.imgs[w.image];
    const CompGeom g = im.out[w.comp];
    const DevQuant &Q = c.quant[im.qt_out[w.comp]];
    const int row = int(rr & 0xFFFFu);
    int ncand = (2 + 60 / int(Q.q[0])) | 1;
    ncand = ncand > 9 ? 9 : ncand;
    // what a DC difference of category 0..11 costs: its code length + that many raw bits, as the float the C source converts it to.  Every
    // lane writes the (same) twelve values and reads them back itself: no barrier (the twelve as five-bit fields of a register pair, shifted
    // out where they are needed: measured, no faster)
    CSH_SHARED float s_costs[64][13];   // (a column per lane: the lanes of a wave belong to different components now)
    float *s_cost = s_costs[threadIdx.x & 63u];
    for (int i = 0; i < 12; i++) s_cost[i] = float(i + int(w.table_dc < 0 ? kStdDcLen[w.comp ? 1 : 0][i] : c.tables[w.table_dc].size[i]));
    const int q = Q.div[0];
    const uint32_t qmul = Q.mul[0], qsh = Q.sh[0];
    const float lt0 = Q.lt[0];
    const int half = ncand / 2;
    int last_dc = 0;   // compress_trellis_pass: 0 at the start of every iMCU row, then the last DC of the row before
    const int by1 = (row + 1) * g.v < g.real_bh ? (row + 1) * g.v : g.real_bh;
    for (int by = row * g.v; by < by1; by++) {
        float acc[9];
        int cprev[9];
        CSH_UNROLL
        for (int k = 0; k < 9; k++) { acc[k] = 0.0f; cprev[k] = 0; }
        int qprev = 0, sprev = 0;   // the block before: its rounded level and sign
        bool cprev_clamped = false;
        // a lane walks its row block by block and every step needs that block's record: the next one is fetched a step ahead (the walk
        // was bound by the latency of these loads: 6 waves per SIMD, two dependent loads per block)
        uint64_t rec_next = c.dcrec[w.unit_base + uint32_t(by * g.real_bw)];
        for (int bi = 0; bi < g.real_bw; bi++) {
            const uint32_t u = uint32_t(by * g.real_bw + bi);
            const uint64_t rec = rec_next;
            if (bi + 1 < g.real_bw) rec_next = c.dcrec[w.unit_base + u + 1u];
            const int raw0 = int(int16_t(uint16_t(rec & 0xFFFFu)));
            const float lambda_dc = tr_bits_f(uint32_t(rec >> 32)) * lt0;
            const int x = raw0 < 0 ? -raw0 : raw0;
            const int qval = tr_level(x, q, qmul, qsh);
            const int sgn = raw0 < 0 ? 1 : 0;
            const bool clamped = qval + half > TRELLIS_MAX_LEVEL;   // a candidate is cut off at the largest level: the closed form below does not hold
            uint64_t bt = 0;
            float nacc[9];
            int ccur[9];
            float dist[9];
            // levels past the ncand-th cost infinity: they never win a comparison (strict <, the first level of a block always stands), so the
            // loops below run over all nine without a test per level (the tests were 200 branches on the lanes' masks)
            CSH_UNROLL
            for (int k = 0; k < 9; k++) {
                int cand = qval - half + k;
                cand = cand > TRELLIS_MAX_LEVEL ? TRELLIS_MAX_LEVEL : cand;
                cand = cand < -TRELLIS_MAX_LEVEL ? -TRELLIS_MAX_LEVEL : cand;
                const int delta = cand * q - x;
                dist[k] = k < ncand ? float(delta * delta) * lambda_dc : __builtin_inff();
                ccur[k] = sgn ? -cand : cand;
            }
            if (bi == 0) {
                CSH_UNROLL
                for (int k = 0; k < 9; k++) { const int d = ccur[k] - last_dc, bits = tr_bitlen(unsigned(d < 0 ? -d : d)); nacc[k] = s_cost[bits] + dist[k]; }
            } else if (clamped || cprev_clamped) {
                // the general statement: every pair of levels by itself
                CSH_UNROLL
                for (int k = 0; k < 9; k++) {
                    float bc = 0.0f;
                    uint32_t bl = 0;
                    CSH_UNROLL
                    for (int l = 0; l < 9; l++) {
                        const int d = ccur[k] - cprev[l], bits = tr_bitlen(unsigned(d < 0 ? -d : d));
                        const float cost = (s_cost[bits] + dist[k]) + acc[l];
                        if (l == 0 || cost < bc) { bc = cost; bl = uint32_t(l); }
                    }
                    nacc[k] = bc;
                    bt |= uint64_t(bl) << (4 * k);
                }
            } else {
                // the levels of a block are consecutive integers, so the difference of level k to the previous block's level l depends on
                // k - l (equal signs: |qval - qprev + k - l|) or on k + l (opposite signs: |qval + qprev - 2 half + k + l|) alone: 17 + 17
                // category look-ups instead of 81
                const bool same = sgn == sprev;
                const int base = same ? qval - qprev : qval + qprev - 2 * half;
                float cd[17];
                CSH_UNROLL
                for (int i = 0; i < 17; i++) { const int d = base + (i - 8); const int dd = same ? d : d + 8; cd[i] = s_cost[tr_bitlen(unsigned(dd < 0 ? -dd : dd))]; }   // same: index k - l + 8; opposite: index k + l
                CSH_UNROLL
                for (int k = 0; k < 9; k++) {
                    float bc = 0.0f;
                    uint32_t bl = 0;
                    CSH_UNROLL
                    for (int l = 0; l < 9; l++) {
                        const float cbits = same ? cd[k - l + 8] : cd[k + l];
                        const float cost = (cbits + dist[k]) + acc[l];
                        if (l == 0 || cost < bc) { bc = cost; bl = uint32_t(l); }
                    }
                    nacc[k] = bc;
                    bt |= uint64_t(bl) << (4 * k);
                }
            }
            CSH_UNROLL
            for (int k = 0; k < 9; k++) { acc[k] = nacc[k]; cprev[k] = ccur[k]; }
            qprev = qval; sprev = sgn; cprev_clamped = clamped;
            c.dcbt[w.unit_base + u] = bt | (uint64_t(uint32_t(qval)) << 36) | (uint64_t(raw0 < 0 ? 1u : 0u) << 47);
        }
        float bv = acc[0];
        uint32_t j = 0;
        CSH_UNROLL
        for (int i = 1; i < 9; i++) if (i < ncand && acc[i] < bv) { bv = acc[i]; j = uint32_t(i); }
        uint64_t bt_next = c.dcbt[w.unit_base + uint32_t(by * g.real_bw + g.real_bw - 1)];
        for (int bi = g.real_bw - 1; bi >= 0; bi--) {
            const uint64_t rec = bt_next;
            if (bi > 0) bt_next = c.dcbt[w.unit_base + uint32_t(by * g.real_bw + bi - 1)];
            int cand = int((rec >> 36) & 2047u) - half + int(j);
            cand = cand > TRELLIS_MAX_LEVEL ? TRELLIS_MAX_LEVEL : cand;
            cand = cand < -TRELLIS_MAX_LEVEL ? -TRELLIS_MAX_LEVEL : cand;
            cand = ((rec >> 47) & 1u) ? -cand : cand;
            c.coef[coef_index(g.tile_base, by * g.bw + bi, 0)] = int16_t(cand);
            if (bi == g.real_bw - 1) last_dc = cand;
            j = uint32_t((rec >> (4 * j)) & 15u);
        }
    }
}

void launch_trellis_ac(hipStream_t st, const TrellisCtx &c) {
    if (!c.nchunks) return;
#ifdef CSH_EMUL
    CSH_LAUNCH_PHASED(k_trellis_ac, 2, dim3(c.nchunks), dim3(CSH_TR_WG), st, c);
#else
    CSH_LAUNCH(k_trellis_ac, dim3(c.nchunks < CSH_TR_MAXWG ? c.nchunks : CSH_TR_MAXWG), dim3(CSH_TR_WG), st, c);
#endif
}
void launch_trellis_dc(hipStream_t st, const TrellisCtx &c) {
    if (c.nrows) CSH_LAUNCH(k_trellis_dc, dim3((c.nrows + 63) / 64), dim3(64), st, c);
}

}  // namespace csh
