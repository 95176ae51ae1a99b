// k_decode_par.hip -- phase 0, fast path: intra-image parallel Huffman decode of sequential-mode scans.
//
// A Huffman bit stream has no random access, but it is self-synchronising: a decoder started at a wrong
// bit position falls into step with the true decoder after a few symbols.  The scan is cut into
// sub-sequences of CSH_SUBSEQ_BYTES; decoder state at a cut = (bit position, block-in-MCU index m,
// zig-zag position k).  With F_t = "decode sub-sequence t from a given state up to the next cut":
//   pass A  (speculate): g[t+1] = F_t(guess)                     -- every lane, 1x decode
//   pass B  (relax): s[t+1] = F_t(s[t]) for every lane once, then only for the lanes whose start state
//            changed (compact work lists), until the list is empty = a fixed point (s[0] is exact, so the
//            fixed point is the true state chain); the lane also counts the blocks n[t] it completes.
//            Bit position and zig-zag index re-synchronise within a few symbols; the block-in-MCU index m
//            only re-synchronises at luma/chroma table changes, so a 1080p scan needs ~8 shrinking rounds.
//   scan    block ordinal of every cut = exclusive scan of n[t]
//   pass W  (write): decode from the true state, store AC coefficients into the k-major tiles and DC
//            DIFFERENCES in scan order; an exclusive scan + scatter turns them into DC values.
// Scans that do not converge in R launches, or end short, are flagged and re-done by the sequential kernel
// (k_decode.hip), as are progressive and restart-interval scans.  Replaces mozjpeg's jdhuff.c for
// libcaesium's JPEG path (reference call site /root/reference/src/compressor.rs:305);
// the formulation follows the published self-synchronisation decoders (Weissenberger & Schmidt 2018/2021).
//
// Pre-pass: 0xFF00 byte stuffing is removed (count -> scan -> compact) so that a position is a plain bit
// index and the hot loop has no per-byte branches.
#include "kernels.h"

namespace csh {

// ---- unstuffing ---------------------------------------------------------------------------------
// stream_of_chunk: binary search of the 64-byte chunk in the (16-byte aligned, sorted) scan table
__device__ static int scan_of_byte(const ParScan *ps, int nps, uint32_t byte) {
    int lo = 0, hi = nps;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (ps[mid].bits_off <= byte) lo = mid; else hi = mid; }
    return lo;
}

// a stuffed zero = 0x00 that follows 0xFF inside the scan (T.81 F.1.2.3).  A workgroup owns 256 consecutive 64-byte chunks:
// phase 0 pulls its 16 KiB (+ the dword in front) into LDS with coalesced loads, phase 1 lets each lane scan its own chunk
// word-wise (zero-byte / 0xFF-byte bit tricks), so nothing is read from global memory byte by byte.
#define CSH_US_STRIDE 17  // 16 data dwords per chunk + 1 pad: lanes start in distinct banks
__device__ __forceinline__ static uint32_t zero_bytes(uint32_t w) { return ~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w | 0x7F7F7F7Fu); }  // 0x80 in every 0x00 byte
__device__ __forceinline__ static void unstuff_stage(const uint8_t *raw, uint32_t c0, uint32_t nchunks, uint32_t *lw, uint32_t tid) {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(raw);
    for (uint32_t i = 0; i < 16; i++) {
        uint32_t d = i * 256 + tid, chunk = c0 + d / 16;
        lw[(d / 16) * CSH_US_STRIDE + d % 16] = chunk < nchunks ? g[size_t(c0) * 16 + d] : 0u;
    }
    if (tid == 0) lw[256 * CSH_US_STRIDE] = c0 ? g[size_t(c0) * 16 - 1] : 0u;  // dword in front of the first chunk
}
// per-lane: 16 "delete" nibble masks (bit 7 of byte i of del[j] set <=> byte 4j+i of the chunk is a stuffed zero)
__device__ __forceinline__ static uint32_t unstuff_masks(const uint32_t *lw, uint32_t tid, uint32_t b0, uint32_t lo, uint32_t end, uint32_t del[16]) {
    uint32_t prev = tid ? lw[(tid - 1) * CSH_US_STRIDE + 15] : lw[256 * CSH_US_STRIDE];
    uint32_t carry = (b0 > lo) ? (zero_bytes(~prev) >> 24) : 0u;  // 0x80 if the byte in front is 0xFF and belongs to this scan
    uint32_t n = 0;
    for (int j = 0; j < 16; j++) {
        uint32_t w = lw[tid * CSH_US_STRIDE + j];
        uint32_t f = zero_bytes(~w), z = zero_bytes(w);
        uint32_t d = z & ((f << 8) | carry);
        carry = f >> 24;
        // bytes outside [lo, end) never count
        uint32_t pos = b0 + 4 * j;
        if (pos + 4 > end) { uint32_t keep = pos >= end ? 0u : (0xFFFFFFFFu >> (8 * (pos + 4 - end))); d &= keep; }
        del[j] = d;
        n += __popc(d);
    }
    return n;
}

__global__ void __launch_bounds__(256) k_unstuff_count(const uint8_t *raw, const ParScan *ps, int nps, uint32_t nchunks, uint32_t *cnt) {
    CSH_SHARED uint32_t lw[256 * CSH_US_STRIDE + 1];
    const uint32_t tid = threadIdx.x, c0 = blockIdx.x * 256, c = c0 + tid;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) { unstuff_stage(raw, c0, nchunks, lw, tid); continue; }
        if (c >= nchunks) continue;
        uint32_t n = 0, b0 = c * 64;
        if (nps) {
            const ParScan &s = ps[scan_of_byte(ps, nps, b0)];
            uint32_t end = s.bits_off + s.bits_len;
            if (b0 < end) { uint32_t del[16]; n = unstuff_masks(lw, tid, b0, s.bits_off, end, del); }
        }
        cnt[c] = n;
    }
}

// byte sink that turns a lane's contiguous output run into aligned stores: it takes up to four bytes at a time through a 64-bit window
// (nearly every dword of a scan holds no stuffed zero and goes in whole), and the dwords that leave the window go out sixteen bytes at a time
// once the position is 16-byte aligned -- the lanes of a wave write 64 different lines with every store, and the copy's time was the count of
// its stores (a dword each: 2.15 ms per 2048 files whatever the arithmetic in front; 1.78 now, with five waves per SIMD asked for -- the rest is
// the scan search's dependent loads.  The range put together in LDS and stored by the workgroup in whole units, tried: the lanes then hold their
// words across a barrier, 188 VGPRs, 3.6 ms).  Single bytes only at the ragged ends.
struct ByteRun {
    uint8_t *p;      // where the next store goes: the dword the window's low bytes belong to, less the dwords waiting in q
    uint64_t acc;
    uint32_t n, lead;   // bytes in the window (the first dword's `lead` bytes in front of the run included: they are not this lane's to write)
    uint32_t q0, q1, q2, q3, qn;   // dwords waiting for a 16-byte store
    __device__ __forceinline__ void begin(uint8_t *dst) { lead = uint32_t(reinterpret_cast<uintptr_t>(dst) & 3); p = dst - lead; acc = 0; n = lead; q0 = q1 = q2 = q3 = 0; qn = 0; }
    __device__ __forceinline__ void dword(uint32_t v) {
        if (qn == 0 && (reinterpret_cast<uintptr_t>(p) & 15)) { *reinterpret_cast<uint32_t *>(p) = v; p += 4; return; }
        q0 = qn == 0 ? v : q0; q1 = qn == 1 ? v : q1; q2 = qn == 2 ? v : q2; q3 = qn == 3 ? v : q3;
        if (++qn == 4) {
#ifdef CSH_EMUL
            const uint32_t four[4] = {q0, q1, q2, q3};
            memcpy(p, four, 16);
#else
            typedef uint32_t us_u32x4 __attribute__((ext_vector_type(4)));
            us_u32x4 v4; v4.x = q0; v4.y = q1; v4.z = q2; v4.w = q3;
            *reinterpret_cast<us_u32x4 *>(p) = v4;
#endif
            p += 16; qn = 0;
        }
    }
    __device__ __forceinline__ void push(uint32_t bytes, uint32_t cnt) {   // the low cnt (0..4) bytes of `bytes`; the ones above them are zero
        acc |= uint64_t(bytes) << (8 * n);
        n += cnt;
        if (n >= 4) {
            if (lead) { for (uint32_t i = lead; i < 4; i++) p[i] = uint8_t(acc >> (8 * i)); lead = 0; p += 4; }
            else dword(uint32_t(acc));
            acc >>= 32; n -= 4;
        }
    }
    __device__ __forceinline__ void finish() {
        const uint32_t q[4] = {q0, q1, q2, q3};
        for (uint32_t i = 0; i < qn; i++) { *reinterpret_cast<uint32_t *>(p) = q[i]; p += 4; }
        for (uint32_t i = lead; i < n; i++) p[i] = uint8_t(acc >> (8 * i));
        n = 0; lead = 0; qn = 0;
    }
};

__global__ void __launch_bounds__(256, 5) k_unstuff_copy(const uint8_t *raw, uint8_t *clean, ParScan *ps, int nps, uint32_t nchunks, const uint64_t *off) {
    CSH_SHARED uint32_t lw[256 * CSH_US_STRIDE + 1];
    const uint32_t tid = threadIdx.x, c0 = blockIdx.x * 256, c = c0 + tid;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) { unstuff_stage(raw, c0, nchunks, lw, tid); continue; }
        if (c >= nchunks || !nps) continue;
        uint32_t b0 = c * 64;
        int si = scan_of_byte(ps, nps, b0);
        const uint32_t lo = ps[si].bits_off, len = ps[si].bits_len, end = lo + len;
        if (b0 >= end) continue;
        uint32_t removed = uint32_t(off[c] - off[lo >> 6]);
        uint32_t del[16];
        unstuff_masks(lw, tid, b0, lo, end, del);
        ByteRun out; out.begin(clean + (b0 - removed));
        for (int j = 0; j < 16; j++) {
            const uint32_t w = lw[tid * CSH_US_STRIDE + j], d = del[j], pos = b0 + 4 * uint32_t(j);
            if (pos >= end) break;
            const uint32_t valid = end - pos < 4u ? end - pos : 4u;
            if (d == 0 && valid == 4) { out.push(w, 4); continue; }
            uint32_t kept = 0, cnt = 0;
            for (uint32_t i = 0; i < valid; i++)
                if (!((d >> (8 * i + 7)) & 1)) { kept |= ((w >> (8 * i)) & 255u) << (8 * cnt); cnt++; }
            out.push(kept, cnt);
        }
        out.finish();
        if (b0 + 64 >= end) {  // last chunk of the scan: publish the unstuffed length
            uint32_t last_chunk = (end + 63) >> 6;
            ps[si].clean_len = len - uint32_t(off[last_chunk] - off[lo >> 6]);
        }
    }
}

// ---- decoder core -------------------------------------------------------------------------------
struct PState { uint32_t pos; int m, k; };
__device__ __forceinline__ static uint64_t pack_state(const PState &s) { return (uint64_t(s.pos) << 16) | (uint64_t(s.m & 255) << 8) | uint64_t(s.k & 255); }
__device__ __forceinline__ static PState unpack_state(uint64_t v) { PState s; s.pos = uint32_t(v >> 16); s.m = int((v >> 8) & 255); s.k = int(v & 255); return s; }

// bit window over the unstuffed stream; reads beyond `len` bytes return zero bits (as libjpeg feeds zeros)
struct PReader {
    const uint8_t *base;
    uint32_t len;
    __device__ __forceinline__ uint32_t word(uint32_t wi) const {  // big-endian 32-bit word wi of the stream
        uint32_t b = wi * 4;
        if (b + 4 <= len) {
            uint32_t v = *reinterpret_cast<const uint32_t *>(base + b);
            return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
        }
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) v = (v << 8) | (b + i < len ? base[b + i] : 0u);
        return v;
    }
};
// A lane reads words 0..34 relative to its sub-sequence (32 of its own + 3 of look-ahead: it enters at most 31 bits past
// its cut, a symbol is at most 31 bits, and the reader fetches one word ahead).  Two LDS layouts:
//   RowReader  (list rounds, scattered sub-sequences): a private row per lane, odd stride so lanes hit distinct banks;
//   SwzReader  (dense passes, 256 consecutive sub-sequences): the workgroup's 32 KiB of stream stored ONCE -- the look-ahead
//              of lane t is simply the row of lane t+1 -- with the word index rotated by the row number so that lanes
//              reading the same relative word hit 32 distinct banks.  No padding, no duplicated look-ahead: 32.1 KiB
//              instead of 37 KiB, which (with the compact table sets) is the difference between 3 and 4 workgroups per CU.
#define CSH_LROW_WORDS 35
#define CSH_LROW_STRIDE 35
struct RowReader {
    const uint32_t *row;
    __device__ __forceinline__ uint32_t word(uint32_t r) const {
#ifdef CSH_EMUL
        if (r >= CSH_LROW_WORDS) { fprintf(stderr, "decode_span: stream window exceeded (%u)\n", r); abort(); }
#endif
        return row[r];
    }
};
#define CSH_SWZ_WORDS (257 * 32)   // 256 rows + the look-ahead row of the last lane
struct SwzReader {
    const uint32_t *lds;
    uint32_t tl;   // lane's row
    __device__ __forceinline__ static uint32_t index(uint32_t row, uint32_t r) { return (row << 5) + ((r + row) & 31u); }
    __device__ __forceinline__ uint32_t word(uint32_t r) const {
#ifdef CSH_EMUL
        if (r >= CSH_LROW_WORDS) { fprintf(stderr, "decode_span: stream window exceeded (%u)\n", r); abort(); }
#endif
        return lds[index(tl + (r >> 5), r & 31u)];
    }
};

__device__ __forceinline__ static int extend_p(int r, int n) { return r < (1 << (n - 1)) ? r - (1 << n) + 1 : r; }

// what the loop needs from the scan descriptor, in registers
struct ParCtx {
    uint64_t sel;          // ParScan::sel
    int nb_mcu;
    uint32_t total_blocks;
    int mcus_x;            // MCUs per row (write pass)
    uint32_t first_mcu;    // where in the scan this segment starts (restart intervals)
    uint32_t real_bits;    // unstuffed length of the segment in bits
    int Ss, Se, Al;        // progressive first scans: the band (AC) and the point transform
};
__device__ __forceinline__ static ParCtx make_ctx(const ParScan &ps, const ImgDesc *im) {
    ParCtx c; c.sel = ps.sel; c.nb_mcu = ps.nb_mcu; c.total_blocks = ps.total_blocks;
    c.mcus_x = im ? (ps.ncomp > 1 ? im->mcus_x : im->in[ps.comp_of[0]].real_bw) : 1;
    c.first_mcu = ps.first_mcu;
    c.real_bits = ps.clean_len * 8u;
    c.Ss = ps.Ss; c.Se = ps.Se; c.Al = ps.Al;
    return c;
}
// per-m placement table of the write pass (LDS): a non-interleaved scan walks its component block by block (h = v = 1)
__device__ __forceinline__ static void make_block_info(const ParScan &ps, const ImgDesc &im, int m, ParBlockInfo &o) {
    const CompGeom &g = im.in[ps.comp_of[m]];
    o.tile_base = g.tile_base;
    if (ps.ncomp > 1) { o.row_step = g.v * g.bw; o.col_step = g.h; o.first = ps.by_of[m] * g.bw + ps.bx_of[m]; }
    else { o.row_step = g.bw; o.col_step = 1; o.first = 0; }
    o.dc_first = ps.dc_base[m] + ps.dc_idx[m]; o.dc_per_mcu = ps.dc_per_mcu[m];
}

// decode from state `st` until st.pos >= stop_bit; returns #blocks completed.  WRITE: store coefficients.
// The loop is INSTRUCTION-ISSUE bound (64 unrelated serial bit streams per wave: every divergent path is executed by the
// whole wave), so it is written to be short and straight:
//   * DC and AC symbols share one code path -- the table is selected by (k == 0), not branched on;
//   * two-level tables (ParHuffSet): a code of >= 10 bits costs one more LDS read, not a search; a coefficient's value bits
//     come from the same 32-bit window as its code;
//   * the stream is read from the lane's LDS window only (36 words cover every reachable position: a lane enters at most
//     31 bits past its cut and a symbol is at most 31 bits), the next word is fetched one refill ahead;
//   * table selectors for every block-in-MCU index sit in a register (ParCtx::sel); the write pass takes its per-block
//     placement from a small LDS table (ParBlockInfo).
// KIND (ParScan::kind, uniform per call): CSH_PS_SEQ a sequential-mode scan -- whole blocks, DC then AC; CSH_PS_DC_FIRST a progressive DC first
// scan -- one DC symbol per block, the difference stored unshifted (k_dc_scatter shifts the prefix sums by Al); CSH_PS_AC_FIRST a progressive AC
// first scan of one component -- per block the symbols of the band Ss..Se (T.81 G.1.2.2, libjpeg jdphuff.c decode_mcu_AC_first: a coefficient
// at k + r with its value bits, stored << Al; ZRL; EOBn = the run of 2^n + n extra bits blocks, this one included, that have nothing more in the
// band).  Its state at a cut is (bit position, zig-zag position): there is no block-in-MCU label to creep, so these scans settle in the first
// list rounds; an EOB run simply adds to the count of blocks the sub-sequence completes.  A first scan only writes coefficients nobody else
// writes (T.81 G.1.1.1.1: bands of later scans are disjoint or refine), but it shares their octets with other scans: 2-byte stores.
template <bool WRITE, int KIND, class R>
__device__ __forceinline__ static uint32_t decode_span(const R &rd, uint32_t w0, const uint8_t *hb, uint32_t sub_off, const ParCtx &cx, PState &st, uint32_t stop_bit,
                                                        uint32_t ordinal, const ParBlockInfo *bi, int16_t *coef, int32_t *dcdiff, uint32_t *hand_over = nullptr) {
    uint32_t nblk = 0;
    int16_t *blk = nullptr;
    int32_t *dcp = nullptr;
    const uint16_t *sub = reinterpret_cast<const uint16_t *>(hb + sub_off);
    uint32_t mcu = 0; int mx = 0, my = 0;
    bool in_range = false;
    // (write pass: a lane finishing the MCU in which the data ran out may walk past its window; everything there is zero bits)
    auto word = [&](uint32_t wi) -> uint32_t { const uint32_t r = wi - w0; return (WRITE && r >= CSH_LROW_WORDS) ? 0u : rd.word(r); };
    // libjpeg's insufficient-data rule (jdhuff.c): the MCU in which the data runs out is finished on zero bits (the LDS window
    // holds zeros past the end), every later MCU of the segment stays zero.  cut_mcu = that MCU, once a block has ended past
    // the last real bit; blocks of later MCUs are decoded (to keep walking) but not stored.  The lane in which the data runs out
    // finishes that MCU even past its own cut (the loop condition below); a lane that STARTS past the data stores nothing.
    uint32_t cut_mcu = 0xFFFFFFFFu;
    const bool dead = WRITE && st.pos > cx.real_bits;
    auto locate = [&](int m) {
        // 24-bit multiplies: the host sends scans of >= 2^24 blocks to the sequential decoder
        in_range = __umul24(mcu, uint32_t(cx.nb_mcu)) + uint32_t(m) < cx.total_blocks && mcu <= cut_mcu && !dead;
        const ParBlockInfo g = bi[m];
        blk = coef + coef_index(g.tile_base, __mul24(my, g.row_step) + __mul24(mx, g.col_step) + g.first, 0);
        dcp = dcdiff + (__umul24(mcu, g.dc_per_mcu) + g.dc_first);
    };
    uint32_t pos = st.pos, wi = (pos >> 5) + 2;
    int k = st.k, m = st.m;
    // WRITE: the coefficients of one octet (8 zig-zag positions = 16 contiguous bytes of the tile) are gathered in two
    // registers and leave as ONE 16-byte store when the block moves on to another octet or ends -- a 2-byte store per
    // coefficient cost the memory system a 32-byte sector write each (2.4x the planes in HBM write traffic).  An octet that
    // this lane may share with a neighbouring lane (the block it entered half-way: the octet of its entry position; the block
    // it leaves unfinished: the octet still pending at the exit) is written coefficient by coefficient instead.
    const int k_first = KIND == CSH_PS_AC_FIRST ? cx.Ss : 0;   // where a block starts
    int cur_oct = -1, shared_oct = k > 0 ? (k >> 3) : -1;
    uint64_t olo = 0, ohi = 0;
    auto flush = [&](bool piecewise) {
        int16_t *dst = blk + cur_oct * CSH_OCT_STRIDE;
        if (!piecewise) *reinterpret_cast<uint4 *>(dst) = uint4{uint32_t(olo), uint32_t(olo >> 32), uint32_t(ohi), uint32_t(ohi >> 32)};
        else {
            CSH_UNROLL
            for (int i = 0; i < 8; i++) { const uint32_t c = uint32_t(((i & 4) ? ohi : olo) >> (16 * (i & 3))) & 0xFFFFu; if (c) dst[i] = int16_t(c); }
        }
        cur_oct = -1; olo = 0; ohi = 0;
    };
    if (WRITE) {
        mcu = ordinal / uint32_t(cx.nb_mcu);
        my = int(mcu + cx.first_mcu) / cx.mcus_x; mx = int(mcu + cx.first_mcu) - my * cx.mcus_x;   // mcu itself stays relative to the segment
        locate(m);
    }
    uint32_t dco, aco;   // byte offsets of the current DC / AC root tables
    auto tables = [&](int mm) { uint32_t x = uint32_t(cx.sel >> __umul24(uint32_t(mm), 6u)); dco = (x & 7u) << 10; aco = (x & 56u) << 7; };
    tables(m);
    // 64-bit bit buffer (MSB first), at least 32 valid bits after every refill
    uint64_t acc = ((uint64_t(word(wi - 2)) << 32) | word(wi - 1)) << (pos & 31);
    int nb = 64 - int(pos & 31);
    uint32_t nxt = word(wi);
    auto step = [&]() {
        const uint32_t w = uint32_t(acc >> 32), top16 = w >> 16;
        const bool isdc = KIND == CSH_PS_DC_FIRST ? true : (KIND == CSH_PS_AC_FIRST ? false : k == 0);
        uint32_t e = *reinterpret_cast<const uint16_t *>(hb + (isdc ? dco : aco) + ((top16 >> 7) << 1));
        if (e & 0x8000u) e = sub[(e & 0xFFFu) + ((top16 & 127u) >> (7u - ((e >> 12) & 7u)))];
        const int len = e ? int(e >> 8) : 16;      // no such code: consume 16 bits, symbol 0 (as the sequential path)
        const int sym = int(e & 255u);
        // one symbol: DC -> (run 0, size sym); AC -> (run sym>>4, size sym&15)
        const int n = sym & 15, r = isdc ? 0 : (sym >> 4);
        const int kn = k + r;
        const bool val = n != 0 && kn <= 63;                           // kn > 63: corrupt run, block ends, no value bits consumed
        // AC first scan: n == 0 and r < 15 is EOBr -- r more bits follow, the run's low bits
        const bool eobrun = KIND == CSH_PS_AC_FIRST && n == 0 && r != 15;
        const int nx = eobrun ? r : n;                                  // raw bits behind the symbol
        int v = int((w << len) >> ((32 - nx) & 31));
        uint32_t run = 1;                                               // blocks this symbol completes, if it completes one
        if (eobrun) { run = (1u << r) + (r ? uint32_t(v) : 0u); v = 0; }
        else { v = v < int(1u << ((n - 1) & 31)) ? v - (1 << n) + 1 : v; v = val ? v : 0; }   // EXTEND (T.81 F.2.2.1); n == 0 is masked
        if (WRITE && in_range) {
            if (KIND == CSH_PS_AC_FIRST) {
                if (val) blk[coef_off(kn)] = int16_t(v * (1 << cx.Al));
                // a run that leaves the scan's band (damaged data: libjpeg stores the coefficient all the same) lands in the band of ANOTHER scan, and the
                // scans of a file are written side by side here: which of the two values stays is a matter of file order -- the sequential kernel's
                if (val && kn > cx.Se) *hand_over = 2u;
            }
            else if (isdc) *dcp = v;
            else if (val) {
                const int oct = kn >> 3;
                if (oct != cur_oct) { if (cur_oct >= 0) flush(cur_oct == shared_oct); cur_oct = oct; }
                const uint64_t piece = uint64_t(uint32_t(v) & 0xFFFFu) << (16 * (kn & 3));
                if (kn & 4) ohi |= piece; else olo |= piece;
            }
        }
        const int used = len + (eobrun ? r : (val ? n : 0));
        const bool eob = KIND == CSH_PS_AC_FIRST ? eobrun : (!isdc && n == 0 && r != 15);
        const int kc = kn > 63 ? 63 : kn;
        if (KIND == CSH_PS_DC_FIRST) k = 64;                             // a block of a DC scan is its one symbol
        else if (KIND == CSH_PS_AC_FIRST) k = (eob || kn > 63 || kc + 1 > cx.Se) ? 64 : kc + 1;   // EOBr, a corrupt run, or the band's end (ZRL: n == 0, kn = k + 15)
        else k = eob ? 64 : kc + 1;                                      // ZRL: k + 16; coefficient / DC: kn + 1
        pos += uint32_t(used);
        acc <<= used;
        nb -= used;
        if (nb < 32) { acc |= uint64_t(nxt) << (32 - nb); nb += 32; wi++; nxt = word(wi); }
        if (k >= 64) {
            if (WRITE && KIND == CSH_PS_SEQ) { if (cur_oct >= 0) flush(cur_oct == shared_oct); shared_oct = -1; }
            k = k_first;
            nblk += run;
            if (WRITE) cut_mcu = (pos > cx.real_bits && mcu < cut_mcu) ? mcu : cut_mcu;
            const bool wrap = m + 1 == cx.nb_mcu;
            m = wrap ? 0 : m + 1;
            tables(m);
            if (WRITE) {
                if (KIND == CSH_PS_AC_FIRST) {   // (nb_mcu = 1: a unit is a block) on by the run: the blocks inside an EOB run carry no bits
                    mcu += run;
                    if (run == 1) { if (++mx == cx.mcus_x) { mx = 0; my++; } }
                    else { my = int(mcu + cx.first_mcu) / cx.mcus_x; mx = int(mcu + cx.first_mcu) - my * cx.mcus_x; }
                } else if (wrap) { mcu++; if (++mx == cx.mcus_x) { mx = 0; my++; } }
                locate(m);
            }
        }
    };
    while (pos < stop_bit) step();
    // the lane in which the data ran out finishes that MCU (if the frame needs it) past its own cut: nobody else will, the
    // next lanes are beyond the data.  (A separate loop: this condition inside the hot loop doubled the kernel's time.)
    if (WRITE && !dead)
        while (pos > cx.real_bits && (k != k_first || m != 0) && __umul24(mcu, uint32_t(cx.nb_mcu)) < cx.total_blocks) step();
    if (WRITE && KIND == CSH_PS_SEQ && cur_oct >= 0) flush(true);   // unfinished block: the next lane may add to this octet
    st.pos = pos; st.k = k; st.m = m;
    // WRITE: first block ordinal (relative to the segment) that stays zero because the data ran out, or 0xFFFFFFFF
    return WRITE ? ((cut_mcu == 0xFFFFFFFFu || dead) ? 0xFFFFFFFFu : (cut_mcu + 1u) * uint32_t(cx.nb_mcu)) : nblk;
}

// the span decoder of a segment's kind (uniform wherever a workgroup serves one scan)
#define CSH_SPAN(WR, KINDV, ...)                                                                        \
    ((KINDV) == CSH_PS_AC_FIRST ? decode_span<WR, CSH_PS_AC_FIRST>(__VA_ARGS__)                          \
                                : (KINDV) == CSH_PS_DC_FIRST ? decode_span<WR, CSH_PS_DC_FIRST>(__VA_ARGS__) : decode_span<WR, CSH_PS_SEQ>(__VA_ARGS__))
__device__ __forceinline__ static bool par_decoded(int kind) { return kind == CSH_PS_SEQ || kind == CSH_PS_DC_FIRST || kind == CSH_PS_AC_FIRST; }

// pass B: relaxation  s[t+1] = F_t(s[t]), in place.  Only the lane of sub-sequence t-1 ever writes s[t]; whenever it
// changes s[t] it appends t to the next work list, so t is re-evaluated in a LATER launch with the newest s[t].  An empty
// list therefore means s[t+1] == F_t(s[t]) for every t, i.e. the true state chain (s[0] is exact).  Reading a value that
// a neighbour updates during the same launch is harmless: it only decides whether this evaluation is already final.
// list rounds: the listed sub-sequences are scattered, so each wave stages its 64 lanes' 144-byte stream windows
// cooperatively -- for lane r's window, lanes 0..35 fetch its 36 consecutive words in ONE coalesced access (a lane
// reading its own window would cost 36 accesses x 64 cache lines per wave) -- then every lane decodes out of LDS.
// 512 lanes per workgroup: 70 KiB of windows + the table set + 4 KiB of lane descriptors = two workgroups = 16 waves per CU
// (with 256 lanes the table set is paid twice as often and only 12 waves fit)
#define CSH_LIST_LANES 512
// A lane whose result changes s[t+1] does not just list t+1 for the next launch: it walks on into t+1 itself (up to
// CSH_LIST_WALK sub-sequences), because with long blocks (high-quality sources: 50+ bytes per block) a wrong state survives
// ~90 % of the sub-sequences it crosses and one cut per launch would need a hundred launches.  claim[t] = number of the
// launch in which somebody took sub-sequence t: whoever exchanges it first evaluates t in this launch -- a listed lane that
// finds it taken stands down (the walker that took it carries the newest s[t]), a walker that finds it taken lists it for
// the next launch as before (the owner may have read s[t] before it changed).  So each s[t+1] still has one writer per launch
// and every change of s[t] is followed by a fresh evaluation of t: the fixed-point argument above is untouched.
// Walking only starts once the list is short or old (count < walk_below, see the launcher): in the first rounds nearly every
// wave would have a walker and wait for it (measured: the list rounds of the bench workload took twice as long).
#define CSH_LIST_WALK 8
template <class SET>
__global__ void __launch_bounds__(CSH_LIST_LANES) k_dec_relax_list(const uint8_t *clean, const ParScan *pss, const SET *huffs, uint64_t *state, const uint64_t *state_rd,
                                                         uint32_t *nblk, const uint64_t *list_in, const uint32_t *cnt_in, uint64_t *list_out, uint32_t *cnt_out,
                                                         uint32_t *claim, uint32_t epoch, uint32_t walk_below) {
    CSH_SHARED uint32_t lbits[CSH_LIST_LANES * CSH_LROW_STRIDE];
    CSH_SHARED SET lhs;
    CSH_SHARED uint32_t d_scan[CSH_LIST_LANES], d_t[CSH_LIST_LANES];   // per lane: ParScan index (0xFFFFFFFF = idle), sub-sequence
    const uint32_t tid = threadIdx.x, j = blockIdx.x * blockDim.x + tid, count = *cnt_in;
    const uint32_t j0 = blockIdx.x * blockDim.x;
    CSH_PHASE_LOOP(3) {
        if (j0 >= count) continue;   // whole workgroup idle (uniform)
        if (phase == 0) {
            uint32_t si = 0xFFFFFFFFu, t = 0;
            if (j < count) {
                uint64_t e = list_in[j]; si = uint32_t(e >> 32); t = uint32_t(e);
                if (atomicExch(&claim[pss[si].sub_base + t], epoch) == epoch) si = 0xFFFFFFFFu;   // a walker of this launch already has it
            }
            d_scan[tid] = si; d_t[tid] = t;
            // the first entry's Huffman set is staged; lanes with another set read theirs from global memory
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&huffs[pss[uint32_t(list_in[j0] >> 32)].huff_set]);
            uint32_t *dst = reinterpret_cast<uint32_t *>(&lhs);
            for (uint32_t i = tid; i < sizeof(SET) / 4; i += CSH_LIST_LANES) dst[i] = src[i];
            continue;
        }
        if (phase == 1) {
            const uint32_t wave0 = tid & ~63u, lane = tid & 63u;
            for (uint32_t r = 0; r < 64; r++) {
                uint32_t si = d_scan[wave0 + r];
                if (si == 0xFFFFFFFFu || lane >= CSH_LROW_WORDS) continue;
                const ParScan &ps = pss[si];
                PReader g; g.base = clean + ps.bits_off; g.len = ps.clean_len;
                lbits[(wave0 + r) * CSH_LROW_STRIDE + lane] = g.word(d_t[wave0 + r] * (CSH_SUBSEQ_BYTES / 4) + lane);
            }
            continue;
        }
        if (j >= count || d_scan[tid] == 0xFFFFFFFFu) continue;
        const ParScan &ps = pss[d_scan[tid]];
        uint32_t t = d_t[tid];
        const size_t base = ps.sub_base + ps.par_index;
        PState st = unpack_state(state_rd[base + t]);
        uint32_t *row = lbits + tid * CSH_LROW_STRIDE;
        RowReader rd; rd.row = row;
        const ParCtx cx = make_ctx(ps, nullptr);
        const int staged_set = pss[uint32_t(list_in[j0] >> 32)].huff_set;
        const uint8_t *hb = ps.huff_set == staged_set ? reinterpret_cast<const uint8_t *>(&lhs) : reinterpret_cast<const uint8_t *>(&huffs[ps.huff_set]);
        PReader g; g.base = clean + ps.bits_off; g.len = ps.clean_len;
        for (int step = 0;; step++) {
            const uint32_t w0 = t * (CSH_SUBSEQ_BYTES / 4), stop = (t + 1) * CSH_SUBSEQ_BYTES * 8;
            const uint32_t n = CSH_SPAN(false, ps.kind, rd, w0, hb, uint32_t(sizeof(SET::root)), cx, st, stop, 0, nullptr, nullptr, nullptr);
            nblk[ps.sub_base + t] = n;
            const uint64_t e = pack_state(st);
            if (e == state[base + t + 1]) break;                            // in step with what is recorded: nothing downstream changes
            state[base + t + 1] = e;
            if ((t + 1) * CSH_SUBSEQ_BYTES >= ps.clean_len) break;          // past the data
            if (count < walk_below && step + 1 < CSH_LIST_WALK && atomicExch(&claim[ps.sub_base + t + 1], epoch) != epoch) {
                t++;                                                         // walk on: this lane's own row is refilled by the lane itself
                for (uint32_t i = 0; i < CSH_LROW_WORDS; i++) row[i] = g.word(t * (CSH_SUBSEQ_BYTES / 4) + i);
                continue;
            }
            list_out[atomicAdd(cnt_out, 1u)] = (uint64_t(ps.par_index) << 32) | (t + 1);
            break;
        }
    }
}

// Scans still listed after the list rounds are almost always stuck on the block-in-MCU LABEL m: with similar (optimised)
// or identical luma/chroma tables a wrong m no longer derails the bit position, so it creeps forward one sub-sequence per
// round.  For those scans only: k_dec_dense<3> decodes every sub-sequence once per possible entry label and records
// (exit label, block count); k_dec_chain then walks the cuts of the scan composing those maps -- a few thousand table
// look-ups -- which yields the exact labels and counts.  A hypothesis whose exit (position, zig-zag index) disagrees
// with what the next cut settled on ends the walk: that image goes to the sequential kernel.
__global__ void __launch_bounds__(256) k_dec_mark_pending(const ParScan *pss, const uint64_t *list_in, const uint32_t *cnt_in, uint32_t *scan_pending) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *cnt_in) return;
    scan_pending[uint32_t(list_in[j] >> 32)] = 1;
}
__global__ void k_dec_chain(const ParScan *pss, int nps, uint64_t *state, uint32_t *nblk, const uint16_t *hyp, const uint32_t *scan_pending, uint32_t *need_seq) {
    int si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si >= nps || !scan_pending[si] || !par_decoded(pss[si].kind)) return;
    const ParScan &ps = pss[si];
    if (ps.kind == CSH_PS_AC_FIRST) { need_seq[ps.image] = 2; return; }   // (no label to settle there: a scan still listed is not converging -- the sequential kernel takes the image)
    uint32_t nsub = (ps.bits_len + CSH_SUBSEQ_BYTES - 1) / CSH_SUBSEQ_BYTES;
    size_t base = ps.sub_base + ps.par_index;
    int m = 0;
    for (uint32_t t = 0; t < nsub && t * CSH_SUBSEQ_BYTES < ps.clean_len; t++) {
        PState st = unpack_state(state[base + t]);
        st.m = m;
        state[base + t] = pack_state(st);
        uint16_t h = hyp[(size_t(ps.sub_base) + t) * 10 + m];
        if (h == 0xFFFF) { need_seq[ps.image] = 2; return; }
        nblk[ps.sub_base + t] = h & 4095u;
        m = h >> 12;
    }
}

// ---- dense passes (every sub-sequence of every scan): a workgroup = 256 consecutive sub-sequences of one scan.
// Phase 0 stages what the lanes will hammer -- the scan's Huffman LUTs and the workgroup's 32 KiB slice of the stream --
// into LDS with coalesced loads (global reads by 64 lanes at a 128-byte stride would cost 64 line requests per load);
// phase 1 decodes out of LDS.  MODE 0: speculate from the guess state, 1: relax in place, 2: write coefficients,
// 3: label hypotheses (see k_dec_chain).
template <int MODE, class SET>
__global__ void __launch_bounds__(256) k_dec_dense(DenseArgs a) {
    CSH_SHARED uint32_t lbits[CSH_SWZ_WORDS];
    CSH_SHARED SET lhs;
    CSH_SHARED ParBlockInfo lbi[10];   // write pass: where block m of an MCU goes
    // The tiles must start at zero (the write pass stores non-zero coefficients only).  The speculation pass and the first relaxation pass store nothing and are bound by
    // their instructions, so their workgroups clear the tiles (half each) between them on the side, each BEHIND its own work (the stores of the early finishers run under the others' decoding) -- 12.85 GB
    // per 2048 files that a memset in front of the phase took 2.1 ms of HBM time for (round 6).  Every workgroup takes part, also those of scans that leave below.
    // (The emulation enters here once per phase: clearing twice is clearing.)
    auto clear_region = [&](uint8_t *ptr, uint64_t bytes) __attribute__((always_inline)) {
        if (!bytes) return;
        const uint64_t nwg = uint64_t(gridDim.x) * gridDim.y, wg = uint64_t(blockIdx.y) * gridDim.x + blockIdx.x;
        const uint64_t share = ((bytes / 16 + nwg - 1) / nwg) * 16, z0 = wg * share, z1 = z0 + share < bytes ? z0 + share : bytes;
        for (uint64_t off = z0 + uint64_t(threadIdx.x) * 16; off < z1; off += 256 * 16) {
#ifdef CSH_EMUL
            memset(ptr + off, 0, 16);
#else
            __builtin_nontemporal_store(0ull, reinterpret_cast<unsigned long long *>(ptr + off));
            __builtin_nontemporal_store(0ull, reinterpret_cast<unsigned long long *>(ptr + off + 8));
#endif
        }
    };
    auto clear_share = [&]() __attribute__((always_inline)) {
        if (MODE != 0 && MODE != 1) return;
        clear_region(a.zero_ptr, a.zero_bytes);
        clear_region(a.zero2_ptr, a.zero2_bytes);
    };
    const ParScan &ps = a.pss[blockIdx.y];
    if (!par_decoded(ps.kind)) { clear_share(); return; }   // listed for the unstuffing pass only (uniform for the workgroup, before any barrier)
    const uint32_t nsub = (ps.bits_len + CSH_SUBSEQ_BYTES - 1) / CSH_SUBSEQ_BYTES;
    const uint32_t t0 = blockIdx.x * 256, tid = threadIdx.x, t = t0 + tid;
    const bool wg_live = t0 < nsub && t0 * CSH_SUBSEQ_BYTES < ps.clean_len && !(MODE == 2 && a.need_seq[ps.image] == 1) &&
                         !(MODE == 3 && a.scan_pending[ps.par_index] == 0);
    PReader g; g.base = a.clean + ps.bits_off; g.len = ps.clean_len;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) {
            if (MODE == 1 && t < nsub && t * CSH_SUBSEQ_BYTES >= ps.clean_len) a.nblk[ps.sub_base + t] = 0;
            if (!wg_live) continue;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(&static_cast<const SET *>(a.huffs)[ps.huff_set]);
            uint32_t *dst = reinterpret_cast<uint32_t *>(&lhs);
            for (uint32_t i = tid; i < sizeof(SET) / 4; i += 256) dst[i] = src[i];
            if (MODE == 2 && int(tid) < ps.nb_mcu && tid < 10) make_block_info(ps, a.imgs[ps.image], int(tid), lbi[tid]);
            const uint32_t w_first = t0 * (CSH_SUBSEQ_BYTES / 4);
            for (uint32_t i = 0; i < CSH_SUBSEQ_BYTES / 4 + 1; i++) {  // 33 x 256 words cover the 256 rows + the look-ahead words of the last one
                uint32_t d = i * 256 + tid;
                if (d >= 256 * (CSH_SUBSEQ_BYTES / 4) + (CSH_LROW_WORDS - CSH_SUBSEQ_BYTES / 4)) break;
                lbits[SwzReader::index(d >> 5, d & 31u)] = g.word(w_first + d);
            }
            continue;
        }
        if (!wg_live || t >= nsub) continue;
        size_t base = ps.sub_base + ps.par_index;
        const bool live = t * CSH_SUBSEQ_BYTES < ps.clean_len;
        SwzReader rd; rd.lds = lbits; rd.tl = tid;
        const uint32_t w0 = t * (CSH_SUBSEQ_BYTES / 4), stop = (t + 1) * CSH_SUBSEQ_BYTES * 8;
        const uint8_t *hb = reinterpret_cast<const uint8_t *>(&lhs);
        const uint32_t sub_off = uint32_t(sizeof(SET::root));
        const ParCtx cx = make_ctx(ps, MODE == 2 ? &a.imgs[ps.image] : nullptr);
        if (MODE == 0) {
            PState st; st.pos = t * CSH_SUBSEQ_BYTES * 8; st.m = 0; st.k = ps.kind == CSH_PS_AC_FIRST ? ps.Ss : 0;
            if (t == 0) a.state[base] = pack_state(st);
            if (!live) { a.state[base + t + 1] = 0; continue; }
            CSH_SPAN(false, ps.kind, rd, w0, hb, sub_off, cx, st, stop, 0, nullptr, nullptr, nullptr);
            a.state[base + t + 1] = pack_state(st);
        } else if (MODE == 1) {
            if (!live) continue;
            PState st = unpack_state(a.state[base + t]);
            uint32_t n = CSH_SPAN(false, ps.kind, rd, w0, hb, sub_off, cx, st, stop, 0, nullptr, nullptr, nullptr);
            a.nblk[ps.sub_base + t] = n;
            uint64_t e = pack_state(st);
            if (e != a.state[base + t + 1]) {
                a.state[base + t + 1] = e;
                if ((t + 1) * CSH_SUBSEQ_BYTES < ps.clean_len) a.list_out[atomicAdd(a.cnt_out, 1u)] = (uint64_t(ps.par_index) << 32) | (t + 1);
            }
        } else if (MODE == 3) {
            // exit label and block count for EVERY possible entry label, from the settled (position, zig-zag index)
            if (!live) continue;
            const PState s0 = unpack_state(a.state[base + t]);
            const PState nx = unpack_state(a.state[base + t + 1]);
            const bool last = (t + 1) * CSH_SUBSEQ_BYTES >= ps.clean_len;
            for (int m0 = 0; m0 < ps.nb_mcu && m0 < 10; m0++) {
                PState st = s0; st.m = m0;
                uint32_t n = CSH_SPAN(false, ps.kind, rd, w0, hb, sub_off, cx, st, stop, 0, nullptr, nullptr, nullptr);
                const bool same = last || (st.pos == nx.pos && st.k == nx.k);
                a.hyp[(size_t(ps.sub_base) + t) * 10 + m0] = same ? uint16_t((st.m << 12) | (n > 4095 ? 4095 : n)) : uint16_t(0xFFFF);
            }
        } else {
            { const uint32_t ns = a.need_seq[ps.image]; if (ns && ns != 4u) continue; }   // (4: a progressive image -- its first scans are decoded here, its AC refinements by k_decode_prog)
            uint32_t ordinal = uint32_t(a.blk_off[ps.sub_base + t] - a.blk_off[ps.sub_base]);
            // a segment that yields fewer blocks than the frame needs has run out of data: its remaining blocks stay zero
            // (cut_block, set by the lane that crossed the end; k_dc_scatter gives those blocks a zero DC as well)
            if (!live || ordinal >= ps.total_blocks) continue;
            PState st = unpack_state(a.state[base + t]);
            const uint32_t cut = CSH_SPAN(true, ps.kind, rd, w0, hb, sub_off, cx, st, stop, ordinal, lbi, a.coef, a.dcdiff, a.need_seq + ps.image);
            if (cut != 0xFFFFFFFFu) atomicMin(&a.cut_block[ps.par_index], cut);
        }
    }
    clear_share();
}

// DC: prefix sums of the differences (scan order) -> absolute DC at zig-zag row 0 of the tiles
__global__ void __launch_bounds__(256) k_dc_scatter(const ParScan *pss, const ImgDesc *imgs, const uint64_t *dc_off, int16_t *coef, const uint32_t *need_seq,
                                                     const uint32_t *cut_block) {
    const ParScan &ps = pss[blockIdx.y];
    if (ps.kind != CSH_PS_SEQ && ps.kind != CSH_PS_DC_FIRST) return;
    { const uint32_t ns = need_seq[ps.image]; if (ns && ns != 4u) return; }
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;  // block ordinal in scan order
    if (j >= ps.total_blocks) return;
    const ImgDesc &im = imgs[ps.image];
    uint32_t mcu = j / uint32_t(ps.nb_mcu);
    int m = int(j - mcu * uint32_t(ps.nb_mcu));
    const CompGeom &g = im.in[ps.comp_of[m]];
    int by, bx;
    const int amcu = int(mcu + ps.first_mcu);   // position in the scan; `mcu` is relative to the segment (restart interval)
    if (ps.ncomp > 1) { int my = amcu / im.mcus_x, mx = amcu - my * im.mcus_x; by = my * g.v + ps.by_of[m]; bx = mx * g.h + ps.bx_of[m]; }
    else { by = amcu / g.real_bw; bx = amcu - by * g.real_bw; }
    uint32_t idx = ps.dc_base[m] + mcu * ps.dc_per_mcu[m] + ps.dc_idx[m];
    // inclusive prefix over this component's differences (two's-complement wrap-around is harmless)
    uint32_t v = uint32_t(dc_off[idx + 1] - dc_off[ps.dc_base[m]]);
    if (j >= cut_block[blockIdx.y]) v = 0;   // MCUs after the one in which the data ran out hold zeros, DC included
    coef[coef_index(g.tile_base, by * g.bw + bx, 0)] = int16_t(int32_t(v) * (1 << (ps.kind == CSH_PS_DC_FIRST ? ps.Al : 0)));   // (a progressive DC first scan carries the values >> Al)
}
// progressive DC refinement (T.81 G.1.2.1.1, libjpeg jdphuff.c decode_mcu_DC_refine): one raw bit per block in scan order -- bit j of the
// unstuffed segment is block j's; a set bit ORs 1 << Al into the DC value (k_dc_scatter wrote it: launched after).  Past the data: nothing.
__global__ void __launch_bounds__(256) k_dc_refine(const uint8_t *clean, const ParScan *pss, const ImgDesc *imgs, int16_t *coef, const uint32_t *need_seq) {
    const ParScan &ps = pss[blockIdx.y];
    if (ps.kind != CSH_PS_DC_REFINE || need_seq[ps.image] != 4u) return;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ps.total_blocks || (j >> 3) >= ps.clean_len) return;
    if (!((clean[ps.bits_off + (j >> 3)] >> (7u - (j & 7u))) & 1u)) return;
    const ImgDesc &im = imgs[ps.image];
    const uint32_t mcu = j / uint32_t(ps.nb_mcu);
    const int m = int(j - mcu * uint32_t(ps.nb_mcu));
    const CompGeom &g = im.in[ps.comp_of[m]];
    int by, bx;
    if (ps.ncomp > 1) { const int my = int(mcu) / im.mcus_x, mx = int(mcu) - my * im.mcus_x; by = my * g.v + ps.by_of[m]; bx = mx * g.h + ps.bx_of[m]; }
    else { by = int(mcu) / g.real_bw; bx = int(mcu) - by * g.real_bw; }
    // every DC refinement scan of an image is in this one grid (blockIdx.y = scan): two scans (Al = 1 and Al = 0 behind a first scan at Al = 2)
    // touch the same coefficient from different workgroups, so the OR is atomic, on the aligned word that holds the int16 (ORs of
    // different bits commute: the scans need no order among themselves)
    int16_t *p = coef + coef_index(g.tile_base, by * g.bw + bx, 0);
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    atomicOr(reinterpret_cast<uint32_t *>(a & ~uintptr_t(3)), (1u << ps.Al) << ((a & 2u) ? 16 : 0));
}

void launch_unstuff_count(hipStream_t st, const uint8_t *raw, const ParScan *ps, int nps, uint32_t nchunks, uint32_t *cnt) {
    if (nchunks) CSH_LAUNCH_PHASED(k_unstuff_count, 2, dim3((nchunks + 255) / 256), dim3(256), st, raw, ps, nps, nchunks, cnt);
}
void launch_unstuff_copy(hipStream_t st, const uint8_t *raw, uint8_t *clean, ParScan *ps, int nps, uint32_t nchunks, const uint64_t *off) {
    if (nchunks) CSH_LAUNCH_PHASED(k_unstuff_copy, 2, dim3((nchunks + 255) / 256), dim3(256), st, raw, clean, ps, nps, nchunks, off);
}
void launch_dec_dense(hipStream_t st, int mode, int nps, uint32_t max_sub, const DenseArgs &a) {
    if (!nps || !max_sub) return;   // max_sub == 0: only progressive scans are listed (a zero-sized grid is a launch error)
    dim3 grid((max_sub + 255) / 256, nps);
#define CSH_DENSE(M) do { if (a.compact) CSH_LAUNCH_PHASED((k_dec_dense<M, ParHuffSet4>), 2, grid, dim3(256), st, a); \
                          else CSH_LAUNCH_PHASED((k_dec_dense<M, ParHuffSet>), 2, grid, dim3(256), st, a); } while (0)
    if (mode == 0) CSH_DENSE(0);
    else if (mode == 1) CSH_DENSE(1);
    else if (mode == 2) CSH_DENSE(2);
    else CSH_DENSE(3);
#undef CSH_DENSE
}
#ifdef CSH_EMUL
int csh_emul_jacobi = 0;  // tests: make a list round read the states as they were BEFORE the launch (what concurrent lanes see at worst)
#endif
void launch_dec_relax_list(hipStream_t st, const uint8_t *clean, const ParScan *ps, uint32_t total_sub, const void *huffs, int compact, uint64_t *state, uint32_t *nblk,
                           const uint64_t *list_in, const uint32_t *cnt_in, uint64_t *list_out, uint32_t *cnt_out, size_t nstate, uint32_t *claim, uint32_t epoch) {
    // walking starts when the list is down to 0.1 % of the sub-sequences, or after ten launches whatever its size: files of
    // ordinary quality are through by then (their lists shrink by 60 % a launch and walking would only make their waves wait
    // for the walkers), what is left is slow-to-synchronise data that needs it
    const uint32_t walk_below = epoch > 10 ? 0xFFFFFFFFu : total_sub / 1024 + 1;
    if (!total_sub) return;
    const uint64_t *state_rd = state;
#ifdef CSH_EMUL
    uint64_t *snap = nullptr;
    if (csh_emul_jacobi) { snap = (uint64_t *)malloc(nstate * 8); memcpy(snap, state, nstate * 8); state_rd = snap; }
#else
    (void)nstate;
#endif
    if (compact) CSH_LAUNCH_PHASED(k_dec_relax_list<ParHuffSet4>, 3, dim3((total_sub + CSH_LIST_LANES - 1) / CSH_LIST_LANES), dim3(CSH_LIST_LANES), st, clean, ps, static_cast<const ParHuffSet4 *>(huffs), state, state_rd, nblk, list_in, cnt_in, list_out, cnt_out, claim, epoch, walk_below);
    else CSH_LAUNCH_PHASED(k_dec_relax_list<ParHuffSet>, 3, dim3((total_sub + CSH_LIST_LANES - 1) / CSH_LIST_LANES), dim3(CSH_LIST_LANES), st, clean, ps, static_cast<const ParHuffSet *>(huffs), state, state_rd, nblk, list_in, cnt_in, list_out, cnt_out, claim, epoch, walk_below);
#ifdef CSH_EMUL
    free(snap);
#endif
}
void launch_dec_mark_pending(hipStream_t st, const ParScan *ps, uint32_t total_sub, const uint64_t *list_in, const uint32_t *cnt_in, uint32_t *scan_pending) {
    if (total_sub) CSH_LAUNCH(k_dec_mark_pending, dim3((total_sub + 255) / 256), dim3(256), st, ps, list_in, cnt_in, scan_pending);
}
void launch_dec_chain(hipStream_t st, const ParScan *ps, int nps, uint64_t *state, uint32_t *nblk, const uint16_t *hyp, const uint32_t *scan_pending, uint32_t *need_seq) {
    if (nps) CSH_LAUNCH(k_dec_chain, dim3((nps + 63) / 64), dim3(64), st, ps, nps, state, nblk, hyp, scan_pending, need_seq);
}
void launch_dc_refine(hipStream_t st, const uint8_t *clean, const ParScan *ps, int nps, uint32_t max_blocks, const ImgDesc *imgs, int16_t *coef, const uint32_t *need_seq) {
    if (nps && max_blocks) CSH_LAUNCH(k_dc_refine, dim3((max_blocks + 255) / 256, nps), dim3(256), st, clean, ps, imgs, coef, need_seq);
}
void launch_dc_scatter(hipStream_t st, const ParScan *ps, int nps, uint32_t max_blocks, const ImgDesc *imgs, const uint64_t *dc_off, int16_t *coef,
                       const uint32_t *need_seq, const uint32_t *cut_block) {
    if (nps && max_blocks) CSH_LAUNCH(k_dc_scatter, dim3((max_blocks + 255) / 256, nps), dim3(256), st, ps, imgs, dc_off, coef, need_seq, cut_block);
}

}  // namespace csh
