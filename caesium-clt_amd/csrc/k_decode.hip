// k_decode.hip -- phase 0: Huffman / progressive entropy decode (T.81 F.2.2, G.2) into k-major
// coefficient tiles.  Replaces mozjpeg's jdhuff.c / jdphuff.c stage of libcaesium's JPEG path
// (reference call site /root/reference/src/compressor.rs:305).
//
// v1 mapping: one lane per image, scans in file order.  The bit-serial dependency is per scan; a batch
// supplies the parallelism.  (DESIGN.md lists the self-synchronising sub-sequence decoder as the next
// step for this kernel.)
#include "kernels.h"

namespace csh {

struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc;
    int nbits;
    int marker;  // first marker byte met (0 = none); zeros are fed from then on, as libjpeg does
    // libjpeg jdhuff.c `insufficient_data`: once more bits have been CONSUMED than the segment holds, the current MCU is
    // finished on zero bits and every later MCU up to the next restart is skipped (left as it is).  real = file bits in acc.
    int real, insufficient;
};

__device__ static inline void br_fill(BitReader &b) {
    while (b.nbits <= 56) {
        int c = 0;
        if (!b.marker && b.p < b.end) {
            c = *b.p;
            if (c == 0xFF) {
                int c2 = (b.p + 1 < b.end) ? b.p[1] : 0xD9;
                if (c2 == 0) b.p += 2;
                else { b.marker = c2; c = 0; }
            } else b.p++;
            if (!b.marker) b.real += 8;
        }
        b.acc |= uint64_t(c) << (56 - b.nbits);
        b.nbits += 8;
    }
}
__device__ static inline int br_peek16(BitReader &b) { if (b.nbits < 16) br_fill(b); return int(b.acc >> 48); }
__device__ static inline void br_skip(BitReader &b, int n) { b.acc <<= n; b.nbits -= n; if (n > b.real) { b.insufficient = 1; b.real = 0; } else b.real -= n; }
__device__ static inline int br_get(BitReader &b, int n) {
    if (n == 0) return 0;
    if (b.nbits < n) br_fill(b);
    int v = int(b.acc >> (64 - n));
    br_skip(b, n);
    return v;
}
__device__ static inline int huff_decode(BitReader &b, const DevHuff &h) {
    int v = br_peek16(b);
    int e = h.look[v >> 7];
    if (e) { br_skip(b, e >> 8); return e & 255; }
    for (int l = 10; l <= 16; l++) {
        int c = v >> (16 - l);
        if (c <= h.maxcode[l]) { br_skip(b, l); return h.vals[(h.valptr[l] + c) & 255]; }
    }
    br_skip(b, 16);
    return 0;
}
__device__ static inline int extend(int r, int n) { return r < (1 << (n - 1)) ? r - (1 << n) + 1 : r; }

__device__ static inline void restart(BitReader &b) {
    b.acc = 0; b.nbits = 0; b.real = 0;
    if (b.marker >= 0xD0 && b.marker <= 0xD7) { b.p += 2; b.marker = 0; b.insufficient = 0; }
    else {
        while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
        if (b.p + 1 < b.end) { b.p += 2; b.insufficient = 0; }
    }
}

struct BlockRef { int16_t *base; };  // &coef[tile][0][lane]; coefficient k lives at base[k*64]
__device__ static inline BlockRef block_ref(int16_t *coef, const CompGeom &g, int by, int bx) {
    int b = by * g.bw + bx;
    BlockRef r; r.base = coef + coef_index(g.tile_base, b, 0);
    return r;
}

__device__ static void decode_block(BitReader &br, const DecScan &sc, bool progressive, const DevHuff &dct, const DevHuff &act,
                                    int &pred, int &eobrun, BlockRef blk) {
    if (!progressive) {
        int t = huff_decode(br, dct);
        int diff = t ? extend(br_get(br, t), t) : 0;
        pred += diff;
        blk.base[0] = int16_t(pred);
        for (int k = 1; k < 64;) {
            int rs = huff_decode(br, act);
            int r = rs >> 4, n = rs & 15;
            if (n) { k += r; if (k > 63) break; blk.base[coef_off(k)] = int16_t(extend(br_get(br, n), n)); k++; }
            else { if (r == 15) k += 16; else break; }
        }
        return;
    }
    if (sc.Ss == 0) {
        if (sc.Ah == 0) {
            int t = huff_decode(br, dct);
            int diff = t ? extend(br_get(br, t), t) : 0;
            pred += diff;
            blk.base[0] = int16_t(pred * (1 << sc.Al));
        } else if (br_get(br, 1)) blk.base[0] = int16_t(blk.base[0] | (1 << sc.Al));
        return;
    }
    if (sc.Ah == 0) {
        if (eobrun > 0) { eobrun--; return; }
        for (int k = sc.Ss; k <= sc.Se; k++) {
            int rs = huff_decode(br, act);
            int r = rs >> 4, n = rs & 15;
            if (n) { k += r; if (k > 63) break; blk.base[coef_off(k)] = int16_t(extend(br_get(br, n), n) * (1 << sc.Al)); }
            else {
                if (r == 15) k += 15;
                else { eobrun = 1 << r; if (r) eobrun += br_get(br, r); eobrun--; break; }
            }
        }
        return;
    }
    // AC refinement (T.81 G.1.2.3)
    int p1 = 1 << sc.Al, m1 = -p1, k = sc.Ss;
    if (eobrun == 0) {
        for (; k <= sc.Se; k++) {
            int rs = huff_decode(br, act);
            int r = rs >> 4, n = rs & 15, val = 0;
            if (n) val = br_get(br, 1) ? p1 : m1;
            else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br_get(br, r); break; }
            do {
                int16_t *c = &blk.base[coef_off(k)];
                int cv = *c;
                if (cv != 0) { if (br_get(br, 1) && (cv & p1) == 0) *c = int16_t(cv >= 0 ? cv + p1 : cv + m1); }
                else if (--r < 0) break;
                k++;
            } while (k <= sc.Se);
            if (val && k <= 63) blk.base[coef_off(k)] = int16_t(val);
        }
    }
    if (eobrun > 0) {
        for (; k <= sc.Se; k++) {
            int16_t *c = &blk.base[coef_off(k)];
            int cv = *c;
            if (cv != 0 && br_get(br, 1) && (cv & p1) == 0) *c = int16_t(cv >= 0 ? cv + p1 : cv + m1);
        }
        eobrun--;
    }
}

__global__ void k_decode_seq(const uint8_t *bits, ImgDesc *imgs, const DecScan *scans, const DevHuffSet *huffs, int16_t *coef, int nimg,
                             const uint32_t *need_seq) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nimg) return;
    if (need_seq[i] == 0 || need_seq[i] == 4) return;  // the parallel decoder / the progressive wave decoder handled this image
    const ImgDesc &im = imgs[i];
    if (need_seq[i] >= 2)          // the parallel decoder gave up half-way (2: labels unresolved, 3: scan ended short): start from clean tiles
        for (int c = 0; c < im.ncomp; c++) {
            int16_t *p = coef + size_t(im.in[c].tile_base) * CSH_TILE_I16;
            for (size_t n = 0; n < size_t(im.in[c].ntiles) * CSH_TILE_I16; n++) p[n] = 0;
        }
    for (int s = 0; s < im.nscans_in; s++) {
        const DecScan &sc = scans[im.first_scan + s];
        const DevHuffSet &hs = huffs[sc.huff_set];
        BitReader br;
        br.p = bits + sc.bits_off; br.end = br.p + sc.bits_len; br.acc = 0; br.nbits = 0; br.marker = 0; br.real = 0; br.insufficient = 0;
        int pred[CSH_MAX_COMPS] = {0, 0, 0};
        int eobrun = 0;
        int ri = sc.restart_interval, todo = ri;
        bool prog = im.progressive_in != 0;
        if (sc.ncomp == 1) {
            const CompGeom &g = im.in[sc.comp[0]];
            const DevHuff &dct = hs.dc[sc.td[0] & 3], &act = hs.ac[sc.ta[0] & 3];
            for (int by = 0; by < g.real_bh; by++)
                for (int bx = 0; bx < g.real_bw; bx++) {
                    if (ri) { if (todo == 0) { restart(br); pred[0] = pred[1] = pred[2] = 0; eobrun = 0; todo = ri; } todo--; }
                    if (br.insufficient) continue;
                    decode_block(br, sc, prog, dct, act, pred[0], eobrun, block_ref(coef, g, by, bx));
                }
        } else {
            for (int my = 0; my < im.mcus_y; my++)
                for (int mx = 0; mx < im.mcus_x; mx++) {
                    if (ri) { if (todo == 0) { restart(br); pred[0] = pred[1] = pred[2] = 0; eobrun = 0; todo = ri; } todo--; }
                    if (br.insufficient) continue;
                    for (int c = 0; c < sc.ncomp; c++) {
                        const CompGeom &g = im.in[sc.comp[c]];
                        const DevHuff &dct = hs.dc[sc.td[c] & 3], &act = hs.ac[sc.ta[c] & 3];
                        for (int y = 0; y < g.v; y++)
                            for (int x = 0; x < g.h; x++)
                                decode_block(br, sc, prog, dct, act, pred[c], eobrun, block_ref(coef, g, my * g.v + y, mx * g.h + x));
                    }
                }
        }
    }
}

void launch_decode_seq(hipStream_t st, const uint8_t *bits, ImgDesc *imgs, const DecScan *scans, const DevHuffSet *huffs, int16_t *coef, int nimg,
                       const uint32_t *need_seq) {
    CSH_LAUNCH(k_decode_seq, dim3((nimg + 63) / 64), dim3(64), st, bits, imgs, scans, huffs, coef, nimg, need_seq);
}

}  // namespace csh
