// png_codes.h -- prefix-code construction and the bit packer shared by the DEFLATE coder (k_png_deflate.hip) and the lossless WebP coder
// (k_vp8l_enc.hip): both formats pack LSB first and store a code's bits from its most significant one, with lengths limited to 15.
#pragma once
#include "png_wave.h"

namespace csp {

// Huffman by repeated merge of the two least frequent (ties: the larger index first), limited by the bit-count adjustment
// of T.81 K.2; at least two symbols are coded (zlib's rule).  One lane does one code: the arrays live in scratch.
// The statement (oracle/png_oracle.c, libjpeg's jpeg_gen_optimal_table) finds the two least frequent by scanning all slots, merges the second tree
// into the first one's slot and bumps the code size of every leaf of both: O(m^2).  Here the same merges come out of a binary heap keyed by
// (frequency, slot descending) -- the keys are distinct, so "the least, the larger slot on a tie" is the heap's minimum, and a merged tree keeps the
// first one's slot exactly as the statement does -- and the code sizes are the leaves' depths, read off the parent links at the end: O(m log m).
// 286 symbols: 40 k instead of 800 k instructions per code (measured in DESIGN.md 7).
__device__ static void code_lengths(const uint32_t *freq_in, int n, int limit, uint8_t *len_out) {
    unsigned long long heap[288];     // frequency << 16 | (0xFFFF - slot)
    int16_t parent[2 * 288], node_of_slot[288], idx[288];
    uint8_t depth[2 * 288];
    int used = 0, m = 0;
    for (int i = 0; i < n; i++) used += freq_in[i] != 0;
    int forced = 2 - used;   // zero-frequency symbols that get a code anyway, lowest first
    for (int i = 0; i < n; i++) {
        uint32_t f = freq_in[i];
        if (!f && forced > 0) { f = 1; forced--; }
        len_out[i] = 0;
        if (f) { heap[m] = (static_cast<unsigned long long>(f) << 16) | static_cast<unsigned long long>(0xFFFF - m); idx[m] = int16_t(i); node_of_slot[m] = int16_t(m); m++; }
    }
    auto sift_down = [&](int at, int size) {
        const unsigned long long v = heap[at];
        for (;;) {
            int ch = 2 * at + 1;
            if (ch >= size) break;
            if (ch + 1 < size && heap[ch + 1] < heap[ch]) ch++;
            if (heap[ch] >= v) break;
            heap[at] = heap[ch]; at = ch;
        }
        heap[at] = v;
    };
    for (int i = m / 2 - 1; i >= 0; i--) sift_down(i, m);
    int size = m, next = m;   // nodes: the leaves 0 .. m-1 (slot order), then one per merge
    while (size > 1) {
        const unsigned long long k1 = heap[0];
        heap[0] = heap[--size]; sift_down(0, size);
        const unsigned long long k2 = heap[0];
        const int s1 = 0xFFFF - int(k1 & 0xFFFFu), s2 = 0xFFFF - int(k2 & 0xFFFFu);
        parent[node_of_slot[s1]] = int16_t(next); parent[node_of_slot[s2]] = int16_t(next);
        node_of_slot[s1] = int16_t(next++);
        heap[0] = (((k1 >> 16) + (k2 >> 16)) << 16) | (k1 & 0xFFFFu);   // the merged tree, in the first one's slot
        sift_down(0, size);
    }
    depth[next - 1] = 0;
    for (int v = next - 2; v >= 0; v--) { const int d = depth[parent[v]] + 1; depth[v] = uint8_t(d > 63 ? 63 : d); }   // (a parent is made after its children)
    int bits[64], first[64];
    for (int i = 0; i < 64; i++) bits[i] = 0;
    for (int i = 0; i < m; i++) bits[depth[i]]++;
    // the symbols by (code size, index): where each size's run starts -- taken before the adjustment moves the counts
    { int at = 0; for (int cs = 0; cs < 64; cs++) { first[cs] = at; at += bits[cs]; } }
    for (int i = 63; i > limit; i--)
        while (bits[i] > 0) {
            int j = i - 2; while (bits[j] == 0) j--;
            bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
        }
    // rank r (in that order) gets the r-th shortest of the adjusted lengths
    int16_t order[288];
    for (int i = 0; i < m; i++) order[first[depth[i]]++] = int16_t(i);
    int l = 1;
    for (int r = 0; r < m; r++) { while (bits[l] == 0) l++; bits[l]--; len_out[idx[order[r]]] = uint8_t(l); }
}
__device__ static void canonical(const uint8_t *len, int n, uint16_t *code) {   // bit-reversed, as deflate packs Huffman codes
    int count[16], next[16];
    for (int i = 0; i < 16; i++) count[i] = 0;
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    int cd = 0;
    next[0] = 0;
    for (int l = 1; l < 16; l++) { cd = (cd + count[l - 1]) << 1; next[l] = cd; }
    for (int i = 0; i < n; i++) {
        code[i] = 0;
        if (!len[i]) continue;
        const int v = next[len[i]]++;
        int r = 0;
        for (int b = 0; b < len[i]; b++) r |= ((v >> b) & 1) << (len[i] - 1 - b);
        code[i] = uint16_t(r);
    }
}

// bits are OR-ed into a window of LDS words (160 of them, zeroed by the caller); complete words leave for HBM after every put
struct BitOut {
    uint32_t *win;       // window: word 0 = output word `wbase`
    uint8_t *out;        // chunk's first byte
    uint64_t bitpos;     // bits written so far
    uint32_t wbase;
    // every lane appends nbits[l] (<= 48) bits of val[l], lane order
    __device__ void put(const LV<uint64_t> &val, const LV<uint32_t> &nbits) {
        uint32_t total;
        const LV<uint32_t> off = lscan(nbits, total);
        if (!total) return;
        LFOR(l) if (nbits[l]) {
            const uint64_t at = bitpos + off[l] - uint64_t(wbase) * 32u;
            const uint32_t w = uint32_t(at >> 5), sh = uint32_t(at & 31u);
            const uint64_t v = val[l] & ((1ull << nbits[l]) - 1ull);
            atomicOr(&win[w], uint32_t(v << sh));
            if (sh + nbits[l] > 32) atomicOr(&win[w + 1], uint32_t(v >> (32 - sh)));
            if (sh + nbits[l] > 64) atomicOr(&win[w + 2], uint32_t(v >> (64 - sh)));
        }
        bitpos += total;
        CSP_WAVE_SYNC();
        const uint32_t full = uint32_t(bitpos >> 5) - wbase;   // complete words
        if (full) {
            for (uint32_t w0 = 0; w0 < full; w0 += 64) LFOR(l) {
                const uint32_t w = w0 + uint32_t(l);
                if (w < full) { const uint32_t v = win[w]; uint8_t *o = out + uint64_t(wbase + w) * 4u; o[0] = uint8_t(v); o[1] = uint8_t(v >> 8); o[2] = uint8_t(v >> 16); o[3] = uint8_t(v >> 24); }
            }
            CSP_WAVE_SYNC();
            const uint32_t carry = win[full];
            CSP_WAVE_SYNC();
            LFOR(l) for (uint32_t w = uint32_t(l); w <= full; w += 64) win[w] = w == 0 ? carry : 0u;
            wbase += full;
            CSP_WAVE_SYNC();
        }
    }
    // the last partial word, byte by byte, up to the byte that holds bit bitpos-1
    __device__ void finish() {
        const uint32_t nbytes = uint32_t(((bitpos + 7) >> 3) - uint64_t(wbase) * 4u);
        LFOR(l) if (uint32_t(l) < nbytes) out[uint64_t(wbase) * 4u + uint32_t(l)] = uint8_t(win[0] >> (8 * l));
    }
};

}  // namespace csp
