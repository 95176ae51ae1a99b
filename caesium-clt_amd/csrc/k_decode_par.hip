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

// a stuffed zero = 0x00 that follows 0xFF inside the scan (T.81 F.1.2.3).  One lane per 64-byte chunk; the chunk and the
// byte in front of it are pulled into registers first, then a 64-bit "delete" mask drives both the count and the copy.
__device__ __forceinline__ static uint64_t stuffed_mask(const uint8_t *raw, uint32_t b0, uint32_t lo, uint32_t end, uint8_t bytes[64]) {
    uint64_t del = 0;
    unsigned prev = (b0 > lo) ? raw[b0 - 1] : 0u;
    for (int i = 0; i < 64; i++) {
        unsigned v = (b0 + i < end) ? raw[b0 + i] : 0x55u;
        bytes[i] = uint8_t(v);
        if (b0 + i < end && prev == 0xFFu && v == 0u) del |= 1ull << i;
        prev = v;
    }
    return del;
}

__global__ void __launch_bounds__(256) k_unstuff_count(const uint8_t *raw, const ParScan *ps, int nps, uint32_t nchunks, uint32_t *cnt) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t n = 0, b0 = c * 64;
    if (nps) {
        const ParScan &s = ps[scan_of_byte(ps, nps, b0)];
        uint32_t end = s.bits_off + s.bits_len;
        if (b0 < end) { uint8_t bytes[64]; n = uint32_t(__popcll(stuffed_mask(raw, b0, s.bits_off, end, bytes))); }
    }
    cnt[c] = n;
}

__global__ void __launch_bounds__(256) k_unstuff_copy(const uint8_t *raw, uint8_t *clean, ParScan *ps, int nps, uint32_t nchunks, const uint64_t *off) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks || !nps) return;
    uint32_t b0 = c * 64;
    int si = scan_of_byte(ps, nps, b0);
    const uint32_t lo = ps[si].bits_off, len = ps[si].bits_len, end = lo + len;
    if (b0 >= end) return;
    uint32_t removed = uint32_t(off[c] - off[lo >> 6]);
    uint8_t bytes[64];
    uint64_t del = stuffed_mask(raw, b0, lo, end, bytes);
    uint32_t o = b0 - removed;
    for (int i = 0; i < 64; i++) {
        if (b0 + i >= end) break;
        if ((del >> i) & 1) continue;
        clean[o++] = bytes[i];
    }
    if (b0 + 64 >= end) {  // last chunk of the scan: publish the unstuffed length
        uint32_t last_chunk = (end + 63) >> 6;
        ps[si].clean_len = len - uint32_t(off[last_chunk] - off[lo >> 6]);
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
    __device__ __forceinline__ uint32_t peek32(uint32_t pos) const {
        uint32_t wi = pos >> 5, o = pos & 31;
        uint64_t w = (uint64_t(word(wi)) << 32) | word(wi + 1);
        return uint32_t((w << o) >> 32);
    }
};

__device__ __forceinline__ static int huff_lookup(const DevHuff &h, uint32_t top16, int &len) {
    int e = h.look[top16 >> 7];
    if (e) { len = e >> 8; return e & 255; }
    for (int l = 10; l <= 16; l++) {
        int c = int(top16 >> (16 - l));
        if (c <= h.maxcode[l]) { len = l; return h.vals[(h.valptr[l] + c) & 255]; }
    }
    len = 16;
    return 0;
}
__device__ __forceinline__ static int extend_p(int r, int n) { return r < (1 << (n - 1)) ? r - (1 << n) + 1 : r; }

// decode from state `st` until st.pos >= stop_bit; returns #blocks completed.  WRITE: store coefficients.
template <bool WRITE>
__device__ static uint32_t decode_span(const PReader &rd, const DevHuffSet &hs, const ParScan &ps, PState &st, uint32_t stop_bit,
                                        uint32_t ordinal, const ImgDesc *im, int16_t *coef, int32_t *dcdiff) {
    uint32_t nblk = 0;
    int16_t *blk = nullptr;
    auto locate = [&](uint32_t ord) {
        if (!WRITE) return;
        blk = nullptr;
        if (ord >= ps.total_blocks) return;
        uint32_t mcu = ord / uint32_t(ps.nb_mcu);
        int m = int(ord - mcu * uint32_t(ps.nb_mcu));
        const CompGeom &g = im->in[ps.comp_of[m]];
        int by, bx;
        if (ps.ncomp > 1) { int my = int(mcu) / im->mcus_x, mx = int(mcu) - my * im->mcus_x; by = my * g.v + ps.by_of[m]; bx = mx * g.h + ps.bx_of[m]; }
        else { by = int(mcu) / g.real_bw; bx = int(mcu) - by * g.real_bw; }
        blk = coef + coef_index(g.tile_base, by * g.bw + bx, 0);
    };
    locate(ordinal);
    while (st.pos < stop_bit) {
        uint32_t w = rd.peek32(st.pos);
        int len;
        if (st.k == 0) {
            int t = huff_lookup(hs.dc[ps.dct[st.m]], w >> 16, len);
            if (WRITE && ordinal + nblk < ps.total_blocks) {
                int diff = 0;
                if (t) { uint32_t w2 = rd.peek32(st.pos + len); diff = extend_p(int(w2 >> (32 - t)), t); }
                dcdiff[ps.dc_base[st.m] + (ordinal + nblk) / uint32_t(ps.nb_mcu) * ps.dc_per_mcu[st.m] + ps.dc_idx[st.m]] = diff;
            }
            st.pos += len + t;
            st.k = 1;
        } else {
            int rs = huff_lookup(hs.ac[ps.act[st.m]], w >> 16, len);
            int r = rs >> 4, n = rs & 15;
            st.pos += len;
            if (n) {
                st.k += r;
                if (st.k > 63) st.k = 64;  // corrupt run: block ends (no extra bits consumed, as the sequential path)
                else {
                    if (WRITE && blk) { uint32_t w2 = rd.peek32(st.pos); blk[st.k << 6] = int16_t(extend_p(int(w2 >> (32 - n)), n)); }
                    st.pos += n;
                    st.k++;
                }
            } else if (r == 15) st.k += 16;
            else st.k = 64;
        }
        if (st.k >= 64) {
            st.k = 0;
            st.m = st.m + 1 == ps.nb_mcu ? 0 : st.m + 1;
            nblk++;
            locate(ordinal + nblk);
        }
    }
    return nblk;
}

// pass A: speculative state at every cut
__global__ void __launch_bounds__(256) k_dec_spec(const uint8_t *clean, const ParScan *pss, const DevHuffSet *huffs, uint64_t *state) {
    const ParScan &ps = pss[blockIdx.y];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nsub = (ps.bits_len + CSH_SUBSEQ_BYTES - 1) / CSH_SUBSEQ_BYTES;
    if (t >= nsub) return;
    uint64_t *s = state + ps.sub_base + ps.par_index;  // nsub+1 entries per scan
    PState st; st.pos = t * CSH_SUBSEQ_BYTES * 8; st.m = 0; st.k = 0;
    if (t == 0) s[0] = pack_state(st);
    if (t * CSH_SUBSEQ_BYTES >= ps.clean_len) { s[t + 1] = 0; return; }  // sub-sequence lies in the slack the unstuffing freed
    PReader rd; rd.base = clean + ps.bits_off; rd.len = ps.clean_len;
    uint32_t stop = (t + 1) * CSH_SUBSEQ_BYTES * 8;
    decode_span<false>(rd, huffs[ps.huff_set], ps, st, stop, 0, nullptr, nullptr, nullptr);
    s[t + 1] = pack_state(st);
}

// pass B: relaxation  s[t+1] = F_t(s[t]), in place.  Only the lane of sub-sequence t-1 ever writes s[t]; whenever it
// changes s[t] it appends t to the next work list, so t is re-evaluated in a LATER launch with the newest s[t].  An empty
// list therefore means s[t+1] == F_t(s[t]) for every t, i.e. the true state chain (s[0] is exact).  Reading a value that
// a neighbour updates during the same launch is harmless: it only decides whether this evaluation is already final.
__device__ __forceinline__ static void relax_one(const uint8_t *clean, const ParScan &ps, const DevHuffSet *huffs, uint64_t *state, uint32_t *nblk,
                                                  uint32_t t, uint64_t *list_out, uint32_t *cnt_out) {
    size_t base = ps.sub_base + ps.par_index;
    PState st = unpack_state(state[base + t]);
    PReader rd; rd.base = clean + ps.bits_off; rd.len = ps.clean_len;
    uint32_t n = decode_span<false>(rd, huffs[ps.huff_set], ps, st, (t + 1) * CSH_SUBSEQ_BYTES * 8, 0, nullptr, nullptr, nullptr);
    nblk[ps.sub_base + t] = n;
    uint64_t e = pack_state(st);
    if (e != state[base + t + 1]) {
        state[base + t + 1] = e;
        if ((t + 1) * CSH_SUBSEQ_BYTES < ps.clean_len) list_out[atomicAdd(cnt_out, 1u)] = (uint64_t(ps.par_index) << 32) | (t + 1);
    }
}

__global__ void __launch_bounds__(256) k_dec_relax_all(const uint8_t *clean, const ParScan *pss, const DevHuffSet *huffs, uint64_t *state, uint32_t *nblk,
                                                        uint64_t *list_out, uint32_t *cnt_out) {
    const ParScan &ps = pss[blockIdx.y];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nsub = (ps.bits_len + CSH_SUBSEQ_BYTES - 1) / CSH_SUBSEQ_BYTES;
    if (t >= nsub) return;
    if (t * CSH_SUBSEQ_BYTES >= ps.clean_len) { nblk[ps.sub_base + t] = 0; return; }
    relax_one(clean, ps, huffs, state, nblk, t, list_out, cnt_out);
}

__global__ void __launch_bounds__(256) k_dec_relax_list(const uint8_t *clean, const ParScan *pss, const DevHuffSet *huffs, uint64_t *state, uint32_t *nblk,
                                                         const uint64_t *list_in, const uint32_t *cnt_in, uint64_t *list_out, uint32_t *cnt_out) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *cnt_in) return;
    uint64_t e = list_in[j];
    relax_one(clean, pss[uint32_t(e >> 32)], huffs, state, nblk, uint32_t(e), list_out, cnt_out);
}

// whatever is still listed after the last launch has not reached the fixed point: sequential fallback for that image
__global__ void __launch_bounds__(256) k_dec_unconverged(const ParScan *pss, const uint64_t *list_in, const uint32_t *cnt_in, uint32_t *need_seq) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= *cnt_in) return;
    need_seq[pss[uint32_t(list_in[j] >> 32)].image] = 2;
}

// pass W: decode from the true states and store
__global__ void __launch_bounds__(256) k_dec_write(const uint8_t *clean, const ParScan *pss, const DevHuffSet *huffs, const uint64_t *state,
                                                    const uint64_t *blk_off, const ImgDesc *imgs, int16_t *coef, int32_t *dcdiff, uint32_t *need_seq) {
    const ParScan &ps = pss[blockIdx.y];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nsub = (ps.bits_len + CSH_SUBSEQ_BYTES - 1) / CSH_SUBSEQ_BYTES;
    if (t >= nsub) return;
    if (need_seq[ps.image]) return;
    PState st = unpack_state(state[ps.sub_base + ps.par_index + t]);
    uint32_t ordinal = uint32_t(blk_off[ps.sub_base + t] - blk_off[ps.sub_base]);
    if (t == nsub - 1) {  // the whole scan must have produced exactly its blocks
        uint32_t total = uint32_t(blk_off[ps.sub_base + nsub] - blk_off[ps.sub_base]);
        if (total < ps.total_blocks) need_seq[ps.image] = 2;
    }
    if (t * CSH_SUBSEQ_BYTES >= ps.clean_len || ordinal >= ps.total_blocks) return;
    PReader rd; rd.base = clean + ps.bits_off; rd.len = ps.clean_len;
    decode_span<true>(rd, huffs[ps.huff_set], ps, st, (t + 1) * CSH_SUBSEQ_BYTES * 8, ordinal, &imgs[ps.image], coef, dcdiff);
}

// DC: prefix sums of the differences (scan order) -> absolute DC at zig-zag row 0 of the tiles
__global__ void __launch_bounds__(256) k_dc_scatter(const ParScan *pss, const ImgDesc *imgs, const uint64_t *dc_off, int16_t *coef, const uint32_t *need_seq) {
    const ParScan &ps = pss[blockIdx.y];
    if (need_seq[ps.image]) return;
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;  // block ordinal in scan order
    if (j >= ps.total_blocks) return;
    const ImgDesc &im = imgs[ps.image];
    uint32_t mcu = j / uint32_t(ps.nb_mcu);
    int m = int(j - mcu * uint32_t(ps.nb_mcu));
    const CompGeom &g = im.in[ps.comp_of[m]];
    int by, bx;
    if (ps.ncomp > 1) { int my = int(mcu) / im.mcus_x, mx = int(mcu) - my * im.mcus_x; by = my * g.v + ps.by_of[m]; bx = mx * g.h + ps.bx_of[m]; }
    else { by = int(mcu) / g.real_bw; bx = int(mcu) - by * g.real_bw; }
    uint32_t idx = ps.dc_base[m] + mcu * ps.dc_per_mcu[m] + ps.dc_idx[m];
    // inclusive prefix over this component's differences (two's-complement wrap-around is harmless)
    uint32_t v = uint32_t(dc_off[idx + 1] - dc_off[ps.dc_base[m]]);
    coef[coef_index(g.tile_base, by * g.bw + bx, 0)] = int16_t(int32_t(v));
}

void launch_unstuff_count(hipStream_t st, const uint8_t *raw, const ParScan *ps, int nps, uint32_t nchunks, uint32_t *cnt) {
    if (nchunks) CSH_LAUNCH(k_unstuff_count, dim3((nchunks + 255) / 256), dim3(256), st, raw, ps, nps, nchunks, cnt);
}
void launch_unstuff_copy(hipStream_t st, const uint8_t *raw, uint8_t *clean, ParScan *ps, int nps, uint32_t nchunks, const uint64_t *off) {
    if (nchunks) CSH_LAUNCH(k_unstuff_copy, dim3((nchunks + 255) / 256), dim3(256), st, raw, clean, ps, nps, nchunks, off);
}
void launch_dec_spec(hipStream_t st, const uint8_t *clean, const ParScan *ps, int nps, uint32_t max_sub, const DevHuffSet *huffs, uint64_t *state) {
    if (nps) CSH_LAUNCH(k_dec_spec, dim3((max_sub + 255) / 256, nps), dim3(256), st, clean, ps, huffs, state);
}
void launch_dec_relax_all(hipStream_t st, const uint8_t *clean, const ParScan *ps, int nps, uint32_t max_sub, const DevHuffSet *huffs, uint64_t *state,
                          uint32_t *nblk, uint64_t *list_out, uint32_t *cnt_out) {
    if (nps) CSH_LAUNCH(k_dec_relax_all, dim3((max_sub + 255) / 256, nps), dim3(256), st, clean, ps, huffs, state, nblk, list_out, cnt_out);
}
void launch_dec_relax_list(hipStream_t st, const uint8_t *clean, const ParScan *ps, uint32_t total_sub, const DevHuffSet *huffs, uint64_t *state, uint32_t *nblk,
                           const uint64_t *list_in, const uint32_t *cnt_in, uint64_t *list_out, uint32_t *cnt_out) {
    if (total_sub) CSH_LAUNCH(k_dec_relax_list, dim3((total_sub + 255) / 256), dim3(256), st, clean, ps, huffs, state, nblk, list_in, cnt_in, list_out, cnt_out);
}
void launch_dec_unconverged(hipStream_t st, const ParScan *ps, uint32_t total_sub, const uint64_t *list_in, const uint32_t *cnt_in, uint32_t *need_seq) {
    if (total_sub) CSH_LAUNCH(k_dec_unconverged, dim3((total_sub + 255) / 256), dim3(256), st, ps, list_in, cnt_in, need_seq);
}
void launch_dec_write(hipStream_t st, const uint8_t *clean, const ParScan *ps, int nps, uint32_t max_sub, const DevHuffSet *huffs, const uint64_t *state,
                      const uint64_t *blk_off, const ImgDesc *imgs, int16_t *coef, int32_t *dcdiff, uint32_t *need_seq) {
    if (nps) CSH_LAUNCH(k_dec_write, dim3((max_sub + 255) / 256, nps), dim3(256), st, clean, ps, huffs, state, blk_off, imgs, coef, dcdiff, need_seq);
}
void launch_dc_scatter(hipStream_t st, const ParScan *ps, int nps, uint32_t max_blocks, const ImgDesc *imgs, const uint64_t *dc_off, int16_t *coef,
                       const uint32_t *need_seq) {
    if (nps) CSH_LAUNCH(k_dc_scatter, dim3((max_blocks + 255) / 256, nps), dim3(256), st, ps, imgs, dc_off, coef, need_seq);
}

}  // namespace csh
