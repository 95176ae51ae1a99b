// k_entropy.hip -- phases 2-5: progressive Huffman entropy ENCODE with optimal tables, on the device.
//
// Replaces mozjpeg's jcphuff.c (progressive scans, EOBRUN, correction bits) + jchuff.c
// jpeg_gen_optimal_table for libcaesium's JPEG path (reference call site
// /root/reference/src/compressor.rs:305).
//
// Formulation (DESIGN.md "Entropy encode"): one lane per block ("unit"), no bit-serial loop over the
// 63 AC positions.  Per block three 64-bit significance masks (|c|>=1,2,4; bit = zig-zag index) turn the
// coder's state machine into bit algebra:
//   first pass  (Ah=0)     : coded positions NZ = M[Al] & band;  zero run = gap between set bits
//   refinement  (Ah=Al+1)  : history H = M[Al+1] & band, newly significant N = M[Al] & ~M[Al+1] & band;
//                            zero run = gap minus popcount(H in the gap); correction bits = bits of H
//   block ends with EOB    : bit Se of NZ (resp. N) is clear
// EOB runs span blocks; they are resolved from two per-scan bit vectors (has-symbol, ends-with-EOB):
// the first block of every (sub-)run owns the EOBRUN symbol, so every block's output is one contiguous
// bit string: [its symbols][EOBRUN symbol if it starts a (sub-)run][its trailing correction bits].
// Passes: flags -> runs -> stats -> optimal tables -> sizes -> exclusive scan -> pack.
#include "kernels.h"

namespace csh {

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_masks(const int16_t *__restrict__ coef, uint64_t *__restrict__ masks, uint32_t first_tile, uint32_t ntiles) {
    uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    tile += first_tile;
    const int16_t *p = coef + size_t(tile) * CSH_TILE_I16 + lane * CSH_BLK_STRIDE;
    uint64_t m0 = 0, m1 = 0, m2 = 0, b0 = 0, b1 = 0, sg = 0;
    CSH_UNROLL
    for (int j = 0; j < 8; j++) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p + CSH_OCT_STRIDE * j);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        CSH_UNROLL
        for (int i = 0; i < 8; i++) {
            int v = (i & 1) ? (int(w[i >> 1]) >> 16) : (int(w[i >> 1] << 16) >> 16);
            unsigned a = unsigned(v < 0 ? -v : v);
            const int k = 8 * j + i;
            m0 |= uint64_t(a >= 1) << k;
            m1 |= uint64_t(a >= 2) << k;
            m2 |= uint64_t(a >= 4) << k;
            b0 |= uint64_t(a & 1u) << k;
            b1 |= uint64_t((a >> 1) & 1u) << k;
            sg |= uint64_t(v < 0) << k;
        }
    }
    uint64_t *o = masks + size_t(tile) * CSH_MASK_TILE + lane;
    o[0] = m0; o[64] = m1; o[128] = m2; o[192] = b0; o[256] = b1; o[320] = sg;
}
void launch_masks(hipStream_t st, const int16_t *coef, uint64_t *masks, uint32_t first_tile, uint32_t ntiles) {
    if (ntiles) CSH_LAUNCH(k_masks, dim3((ntiles + 3) / 4), dim3(256), st, coef, masks, first_tile, ntiles);
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ static uint64_t band_mask(int Ss, int Se) { return (~0ull >> (63 - Se)) & (~0ull << Ss); }
__device__ __forceinline__ static int msb64(uint64_t v) { return 63 - __clzll(v); }
__device__ __forceinline__ static int bitlen32(unsigned v) { return 32 - __clz(v); }

// unit -> padded block index for a non-interleaved scan of component geometry g
__device__ __forceinline__ static int unit_block(const CompGeom &g, uint32_t u) {
    int by = int(u) / g.real_bw, bx = int(u) - by * g.real_bw;
    return by * g.bw + bx;
}
__device__ __forceinline__ static uint64_t load_mask(const uint64_t *masks, const CompGeom &g, int b, int level) {
    return masks[(size_t(g.tile_base) + size_t(b >> 6)) * CSH_MASK_TILE + size_t(level) * 64 + size_t(b & 63)];
}
__device__ __forceinline__ static bool get_bit(const uint64_t *w, uint32_t i) { return (w[i >> 6] >> (i & 63)) & 1; }

struct AcMasks { uint64_t NZ, H, N, C, S; };  // first pass uses NZ; refinement uses H and N (+ C: correction-bit plane, S: signs when packing)
template <bool VALUES = false>
__device__ __forceinline__ static AcMasks ac_masks(const uint64_t *masks, const CompGeom &g, int b, const EncScan &sc) {
    uint64_t band = band_mask(sc.Ss, sc.Se);
    AcMasks m;
    uint64_t lo = load_mask(masks, g, b, sc.Al);
    m.C = 0; m.S = 0;
    if (sc.Ah == 0) { m.NZ = lo & band; m.H = 0; m.N = 0; }
    else {
        uint64_t hi = load_mask(masks, g, b, sc.Al + 1); m.H = hi & band; m.N = lo & ~hi & band; m.NZ = 0;
        if (VALUES) { m.C = load_mask(masks, g, b, 3 + sc.Al); m.S = load_mask(masks, g, b, 5); }   // Al <= 1 whenever Ah != 0 (three significance planes)
    }
    return m;
}

// ---- pass A: per-block flags of every AC scan.  Lane = block, so a wave's 64 has-symbol / ends-with-EOB flags ARE one
// word of the scan's bit vectors: one ballot, one 8-byte store, no atomics.
__global__ void __launch_bounds__(256) k_ac_flags(EncCtx c) {
    const ScanWork w = c.work[c.chunk_work[blockIdx.x]];
    const EncScan &sc = c.script[w.scan];
    if (sc.Ss == 0) return;
    uint32_t u = (blockIdx.x - w.first_chunk) * blockDim.x + threadIdx.x;
    bool has_sym = false, ends_eob = false;
    if (u < w.nunits) {
        const CompGeom &g = c.imgs[w.image].out[sc.comp[0]];
        AcMasks m = ac_masks(c.masks, g, unit_block(g, u), sc);
        uint64_t S = sc.Ah == 0 ? m.NZ : m.N;
        has_sym = S != 0;
        ends_eob = !((S >> sc.Se) & 1);
        int tail = 0;
        if (sc.Ah) tail = S ? __popcll(m.H & ~((2ull << msb64(S)) - 1)) : __popcll(m.H);
        c.tail[w.unit_base + u] = uint8_t(tail);
    }
#ifdef CSH_EMUL
    if (has_sym) atomicOr(reinterpret_cast<unsigned long long *>(c.sym_bits + w.word_base + (u >> 6)), 1ull << (u & 63));
    if (ends_eob) atomicOr(reinterpret_cast<unsigned long long *>(c.eob_bits + w.word_base + (u >> 6)), 1ull << (u & 63));
#else
    uint64_t ms = __ballot(has_sym), me = __ballot(ends_eob);
    if ((threadIdx.x & 63) == 0 && (u >> 6) < ((w.nunits + 63) >> 6)) { c.sym_bits[w.word_base + (u >> 6)] = ms; c.eob_bits[w.word_base + (u >> 6)] = me; }
#endif
}

// ---- pass B: EOB run structure -> EOBRUN value owned by the first block of each (sub-)run
// A run [u .. t] is cut into sub-runs as jcphuff.c does: after 0x7FFF blocks, and (refinement) as soon as more than
// MAX_CORR_BITS - DCTSIZE2 + 1 = 937 correction bits are pending.  Serial form (short runs, and the emulation build):
__device__ static void eob_run_serial(const EncCtx &c, const ScanWork &w, const EncScan &sc, uint32_t u, uint32_t t) {
    uint16_t *er = c.eobrun + w.unit_base;
    if (sc.Ah == 0) {
        uint32_t L = t - u + 1, pos = u;
        while (L > 0) { uint32_t l = L < 0x7FFF ? L : 0x7FFF; er[pos] = uint16_t(l); pos += l; L -= l; }
        return;
    }
    const uint8_t *tl = c.tail + w.unit_base;
    uint32_t cnt = 0, be = 0, s0 = u;
    auto step = [&](uint32_t j, uint32_t tail_bits) {
        cnt++; be += tail_bits;
        if (cnt == 0x7FFF || be > 937) { er[s0] = uint16_t(cnt); cnt = 0; be = 0; s0 = j + 1; }
    };
    uint32_t j = u;
    while (j <= t && (reinterpret_cast<uintptr_t>(tl + j) & 7)) { step(j, tl[j]); j++; }
    for (; j + 7 <= t; j += 8) {
        const uint64_t v = *reinterpret_cast<const uint64_t *>(tl + j);
        CSH_UNROLL
        for (int i = 0; i < 8; i++) step(j + i, uint32_t(v >> (8 * i)) & 255u);
    }
    for (; j <= t; j++) step(j, tl[j]);
    if (cnt) er[s0] = uint16_t(cnt);
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
    const uint32_t wi = c.chunk_work[blockIdx.x];
    const ScanWork w = c.work[wi];
    const EncScan &sc = c.script[w.scan];
    if (sc.Ss == 0) return;
    uint32_t u = (blockIdx.x - w.first_chunk) * blockDim.x + threadIdx.x;
    if (u >= w.nunits) return;
    const uint64_t *sym = c.sym_bits + w.word_base, *eob = c.eob_bits + w.word_base;
    if (!get_bit(eob, u)) return;
    bool start = get_bit(sym, u) || u == 0 || !get_bit(eob, u - 1);
    if (!start) return;
    uint32_t t;
    if (eob_run_end(sym, w.nunits, u, CSH_LONG_RUN_WORDS, t)) eob_run_serial(c, w, sc, u, t);
    else { uint32_t e = atomicAdd(c.long_cnt, 1u); c.long_runs[2 * e] = wi; c.long_runs[2 * e + 1] = u; }
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
#ifdef CSH_EMUL
        uint32_t t;
        eob_run_end(sym, w.nunits, u, 0xFFFFFFFFu, t);
        eob_run_serial(c, w, sc, u, t);
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
            if (lane == 0) { uint32_t L = t - u + 1, pos = u; while (L > 0) { uint32_t l = L < 0x7FFF ? L : 0x7FFF; er[pos] = uint16_t(l); pos += l; L -= l; } }
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
                if (lane == 0) er[s0] = uint16_t(cnt + l0 + 1);
                s0 = pos + l0 + 1; pos = s0; cnt = 0; be = 0;
            } else {
                const uint32_t len = t - pos + 1 < 64 ? t - pos + 1 : 64;
                be += uint32_t(__shfl(int(incl), 63, 64)); cnt += len; pos += len;
            }
        }
        if (cnt && lane == 0) er[s0] = uint16_t(cnt);
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// the walker: visits exactly what the block emits, in stream order, and hands it to a sink
//   sink.sym(t, s)    Huffman symbol s of table t (index inside the scan's table group)
//   sink.raw(v, n)    n raw bits (n <= 16)
struct StatsSink {
    DevEncTable *tab;
    static constexpr bool kValues = false;
    __device__ __forceinline__ void sym(int t, int s) { atomicAdd(&tab[t].freq[s], 1u); }
    __device__ __forceinline__ void syms(int t, int s, int n) { if (n) atomicAdd(&tab[t].freq[s], unsigned(n)); }
    __device__ __forceinline__ void raw(unsigned, int) {}
    __device__ __forceinline__ void rawcount(int) {}
};
// k_pack reads the scan's tables from LDS: ltab[t * 256 + s] = size << 16 | code (staged once per workgroup; for k_sizes,
// which only needs one byte per symbol, the staging barrier costs more than it saves -- measured)
__device__ __forceinline__ static void stage_enc_tables(uint32_t *ltab, const DevEncTable *tab, int ntables) {
    for (int i = threadIdx.x; i < ntables * 256; i += blockDim.x) { const DevEncTable &T = tab[i >> 8]; ltab[i] = (uint32_t(T.size[i & 255]) << 16) | T.code[i & 255]; }
}
struct SizeSink {
    const DevEncTable *tab;
    uint32_t bits;
    static constexpr bool kValues = false;
    __device__ __forceinline__ void sym(int t, int s) { bits += tab[t].size[s]; }
    __device__ __forceinline__ void syms(int t, int s, int n) { bits += unsigned(n) * tab[t].size[s]; }
    __device__ __forceinline__ void raw(unsigned, int n) { bits += n; }
    __device__ __forceinline__ void rawcount(int n) { bits += n; }
};
struct PackSink {
    // Bits are gathered in a 64-bit accumulator and leave as whole big-endian-logical 32-bit words.  Only the first and the
    // last word of a unit's bit string can be shared with a neighbouring unit, so only those two need an atomic OR; the
    // words in between are exclusively this lane's and are stored plainly (the pool is zero-initialised).
    const uint32_t *tab;
    uint32_t *raw_words;
    uint64_t pos;      // absolute bit position in the raw pool of the next bit to emit
    uint64_t acc;      // pending bits, right-aligned
    int nacc;          // number of pending bits (< 32 after every put)
    bool first;        // the next word written is the unit's first (possibly shared) word
    uint32_t fw; uint64_t fwi;   // that first word, held back: both atomics of a unit are issued together at the end, when the
                                 // wave has reconverged -- neighbouring lanes' atomics then travel in the same instruction
    static constexpr bool kValues = true;
    __device__ __forceinline__ void begin(uint64_t p) { pos = p; nacc = int(p & 31); acc = 0; first = true; fw = 0; fwi = 0; }
    __device__ __forceinline__ void put(unsigned v, int n) {
        if (n == 0) return;
        v &= (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
        acc = (acc << n) | v;
        nacc += n;
        pos += n;
        if (nacc >= 32) {
            uint32_t w = uint32_t(acc >> (nacc - 32));
            uint64_t wi = (pos - uint64_t(nacc)) >> 5;   // word that holds the oldest pending bit
            if (first) { fw = w; fwi = wi; first = false; }
            else raw_words[wi] = w;
            nacc -= 32;
            acc &= (nacc ? ((1ull << nacc) - 1ull) : 0ull);
        }
    }
    __device__ __forceinline__ void finish() {
        if (fw) atomicOr(raw_words + fwi, fw);
        if (nacc == 0) return;
        uint32_t w = uint32_t(acc << (32 - nacc));
        uint64_t wi = (pos - uint64_t(nacc)) >> 5;
        if (w) atomicOr(raw_words + wi, w);
        nacc = 0;
    }
    __device__ __forceinline__ void sym(int t, int s) { uint32_t e = tab[t * 256 + s]; put(e & 0xFFFFu, int(e >> 16)); }
    __device__ __forceinline__ void syms(int t, int s, int n) { for (int i = 0; i < n; i++) sym(t, s); }
    __device__ __forceinline__ void raw(unsigned v, int n) { put(v, n); }
    __device__ __forceinline__ void rawcount(int) {}
};

template <class Sink>
__device__ __forceinline__ static void emit_eobrun(Sink &sink, unsigned run) {
    if (!run) return;
    int nb = bitlen32(run) - 1;
    sink.sym(0, nb << 4);
    if (nb) sink.raw(run, nb);
}

template <class Sink>
__device__ static void walk_ac_first(Sink &sink, const int16_t *blk, uint64_t NZ, const EncScan &sc, unsigned run) {
    int prev = sc.Ss - 1;
    while (NZ) {
        int k = __ffsll((unsigned long long)NZ) - 1;
        NZ &= NZ - 1;
        int r = k - prev - 1;
        prev = k;
        sink.syms(0, 0xF0, r >> 4);
        int v = blk[coef_off(k)];
        unsigned a = unsigned(v < 0 ? -v : v) >> sc.Al;
        int nb = bitlen32(a);
        sink.sym(0, ((r & 15) << 4) | nb);
        sink.raw(v < 0 ? ~a : a, nb);
    }
    emit_eobrun(sink, run);
}

template <class Sink>
__device__ static void walk_ac_refine(Sink &sink, uint64_t H, uint64_t N, uint64_t C, uint64_t S, const EncScan &sc, unsigned run) {
    if (!Sink::kValues) {
        // symbols and bit counts only: zero run = gap - popcount(history in the gap)
        int prev = sc.Ss - 1;
        uint64_t n = N;
        while (n) {
            int k = __ffsll((unsigned long long)n) - 1;
            n &= n - 1;
            uint64_t between = (prev + 1 <= k - 1) ? band_mask(prev + 1, k - 1) : 0ull;
            int z = (k - prev - 1) - __popcll(H & between);
            prev = k;
            sink.syms(0, 0xF0, z >> 4);
            sink.sym(0, ((z & 15) << 4) | 1);
            sink.rawcount(1);
        }
        sink.rawcount(__popcll(H));
        emit_eobrun(sink, run);
        return;
    }
    // exact stream order (jcphuff.c encode_mcu_AC_refine): correction bits ride behind the next symbol
    int eobpos = N ? msb64(N) : -1;
    uint64_t all = H | N;
    int prev = sc.Ss - 1, r = 0;
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
        // no coefficient is read: the correction bit and the sign come from the bit planes
        if ((H >> k) & 1) { br = (br << 1) | ((C >> k) & 1); brn++; }
        else {
            sink.sym(0, (r << 4) | 1);
            sink.raw(unsigned((~S >> k) & 1), 1);
            if (brn > 32) { sink.raw(unsigned(br >> 32), brn - 32); }
            if (brn) sink.raw(unsigned(br), brn > 32 ? 32 : brn);
            br = 0; brn = 0; r = 0;
        }
    }
    emit_eobrun(sink, run);
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
                uint64_t NZ = load_mask(c.masks, g, b, 0) & ~1ull;
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

template <class Sink>
__device__ __forceinline__ static void walk_unit(Sink &sink, const EncCtx &c, const ScanWork &w, const EncScan &sc, uint32_t u) {
    const ImgDesc &im = c.imgs[w.image];
    if (sc.sequential) { walk_seq(sink, c, im, sc, u); return; }
    if (sc.Ss == 0) { walk_dc(sink, c, im, sc, u); return; }
    const CompGeom &g = im.out[sc.comp[0]];
    int b = unit_block(g, u);
    AcMasks m = ac_masks<Sink::kValues>(c.masks, g, b, sc);
    const int16_t *blk = c.coef + coef_index(g.tile_base, b, 0);
    unsigned run = c.eobrun[w.unit_base + u];
    if (sc.Ah == 0) walk_ac_first(sink, blk, m.NZ, sc, run);
    else walk_ac_refine(sink, m.H, m.N, m.C, m.S, sc, run);
}

// ---- pass C: symbol statistics.  A workgroup walks its 256 consecutive units of one scan into an LDS histogram (ds_add),
// then flushes the non-zero bins with one global atomic each.  (One unit per lane: with eight units per lane the kernel was
// 40 % slower -- a lane's walks ran one after the other, each with its own round trips to memory.)
struct LdsStatsSink {
    uint32_t *hist;  // [4][257] in LDS
    static constexpr bool kValues = false;
    __device__ __forceinline__ void sym(int t, int s) { atomicAdd(&hist[t * 257 + s], 1u); }
    __device__ __forceinline__ void syms(int t, int s, int n) { if (n) atomicAdd(&hist[t * 257 + s], unsigned(n)); }
    __device__ __forceinline__ void raw(unsigned, int) {}
    __device__ __forceinline__ void rawcount(int) {}
};
__global__ void __launch_bounds__(256) k_stats(EncCtx c) {
    CSH_SHARED uint32_t hist[4 * 257];
    const ScanWork w = c.work[c.chunk_work[blockIdx.x]];
    const EncScan &sc = c.script[w.scan];
    CSH_PHASE_LOOP(3) {
        if (sc.ntables == 0) continue;
        if (phase == 0) { for (int i = threadIdx.x; i < 4 * 257; i += blockDim.x) hist[i] = 0; continue; }
        if (phase == 1) {
            LdsStatsSink s; s.hist = hist;
            uint32_t u = (blockIdx.x - w.first_chunk) * blockDim.x + threadIdx.x;
            if (u < w.nunits) walk_unit(s, c, w, sc, u);
            continue;
        }
        for (int i = threadIdx.x; i < sc.ntables * 257; i += blockDim.x) {
            uint32_t v = hist[i];
            if (v) atomicAdd(&c.tables[w.table_base + i / 257].freq[i % 257], v);
        }
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

// ---- pass E: size in bits of every unit's output
__global__ void __launch_bounds__(256) k_sizes(EncCtx c) {
    const ScanWork w = c.work[c.chunk_work[blockIdx.x]];
    const EncScan &sc = c.script[w.scan];
    uint32_t u = (blockIdx.x - w.first_chunk) * blockDim.x + threadIdx.x;
    if (u >= w.nunits) return;
    SizeSink s; s.tab = c.tables + w.table_base; s.bits = 0;
    walk_unit(s, c, w, sc, u);
    c.unit_bits[w.unit_base + u] = s.bits;
}

// ---- pass G: pack.  raw_off (bytes, multiple of 64) per scan comes from k_scan_layout.
__global__ void __launch_bounds__(256) k_pack(EncCtx c) {
    const ScanWork w = c.work[c.chunk_work[blockIdx.x]];
    const EncScan &sc = c.script[w.scan];
    CSH_SHARED uint32_t ltab[4 * 256];
    uint32_t u = (blockIdx.x - w.first_chunk) * blockDim.x + threadIdx.x;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) { stage_enc_tables(ltab, c.tables + w.table_base, sc.ntables); continue; }
        if (u >= w.nunits) continue;
        if (w.no_room) { c.status[w.image] = 20200; continue; }   // decided per scan by k_scan_place: no data-dependent branch in front of the loads
        uint64_t base = c.unit_off[w.unit_base];
        uint64_t total = c.unit_off[w.unit_base + w.nunits] - base;
        uint64_t raw_bit0 = w.raw_off * 8;
        PackSink s; s.tab = ltab; s.raw_words = c.raw;
        s.begin(raw_bit0 + (c.unit_off[w.unit_base + u] - base));
        walk_unit(s, c, w, sc, u);
        if (u == w.nunits - 1) {  // flush_bits: pad the last byte with 1-bits
            int pad = int((8 - (total & 7)) & 7);
            if (pad) s.put((1u << pad) - 1u, pad);
        }
        s.finish();
    }
}

static dim3 unit_grid(const EncCtx &c) { return dim3(c.nchunks); }   // flat: one workgroup per 256-unit chunk that exists
void launch_ac_flags(hipStream_t st, const EncCtx &c) { if (c.nchunks) CSH_LAUNCH(k_ac_flags, unit_grid(c), dim3(256), st, c); }
void launch_ac_runs(hipStream_t st, const EncCtx &c) {
    if (!c.nchunks) return;
    CSH_LAUNCH(k_ac_runs, unit_grid(c), dim3(256), st, c);
#ifdef CSH_EMUL
    CSH_LAUNCH(k_ac_runs_long, dim3(64), dim3(1), st, c);
#else
    CSH_LAUNCH(k_ac_runs_long, dim3(4096), dim3(64), st, c);
#endif
}
void launch_stats(hipStream_t st, const EncCtx &c) {
    if (c.nchunks) CSH_LAUNCH_PHASED(k_stats, 3, unit_grid(c), dim3(256), st, c);
}
void launch_sizes(hipStream_t st, const EncCtx &c) { if (c.nchunks) CSH_LAUNCH(k_sizes, unit_grid(c), dim3(256), st, c); }
void launch_pack(hipStream_t st, const EncCtx &c) { if (c.nchunks) CSH_LAUNCH_PHASED(k_pack, 2, unit_grid(c), dim3(256), st, c); }

}  // namespace csh
