// k_png_filter.hip -- row P3 of SURVEY.md 8a: the PNG row-filter search (oxipng's RowFilter 0-9 as restated in
// oracle/png_oracle.c filter_row() / filter_scores() / cso_png_filter()).
// Filtering reads RAW neighbours only, so every row is independent: the five fixed-filter streams are produced in one
// pass (slots 0..4), every (row, filter) candidate is scored from them, and the adaptive strategies (5 MinSum, 6 Entropy,
// 7 Bigrams, 8 BigEnt, 9 Brute) pick per row and gather their stream out of the five.
#include "png_kernels.h"
#include "png_lz.h"
#include "png_wave.h"

namespace csp {

__device__ __forceinline__ static int paeth_f(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
__device__ __forceinline__ static uint64_t ilog2i(uint64_t i) {   // i * log2(i) in integers (LodePNG)
    if (!i) return 0;
    const int l = 63 - __clzll((unsigned long long)i);
    return i * uint64_t(l) + ((i - (1ull << l)) << 1);
}

// ---- P2: reductions (oracle: cso_png_reduce)
__global__ void __launch_bounds__(256) k_png_analyze(const PngImg *imgs, const uint32_t *row_image, const uint8_t *pix, uint32_t *flags, const uint32_t *status) {
    const uint32_t row = blockIdx.x, image = row_image[row];
    if (status[image]) return;
    const PngImg &im = imgs[image];
    const uint32_t ch = im.channels, bps = im.bps;
    uint32_t keep = flags[image];   // only ever cleared: a stale read costs work, not correctness
    if (!keep) return;
    const uint32_t start = keep;
    const uint8_t *r = pix + im.pix_off + uint64_t(row - im.row_base) * im.rowbytes;
    if (keep & 64u) return;   // an 8-bit indexed image: k_png_used looks at it
    if (!bps) return;
    for (uint32_t x = threadIdx.x; x < im.width && keep; x += blockDim.x) {
        const uint8_t *px = r + uint64_t(x) * ch * bps;
        if (keep & 1u) for (uint32_t k = 0; k < ch; k++) if (px[2 * k] != px[2 * k + 1]) keep &= ~1u;
        if (keep & 2u) for (uint32_t b = 0; b < bps; b++) if (px[(ch - 1) * bps + b] != 0xFF) keep &= ~2u;
        if (keep & 4u) for (uint32_t b = 0; b < bps; b++) if (px[b] != px[bps + b] || px[b] != px[2 * bps + b]) keep &= ~4u;
        const uint32_t v = px[0];   // the grey level, if the image turns out grey (high byte of a 16-bit sample)
        if ((keep & 8u) && v % 17u) keep &= ~8u;
        if ((keep & 16u) && v % 85u) keep &= ~16u;
        if ((keep & 32u) && v % 255u) keep &= ~32u;
    }
    if (keep != start) atomicAnd(&flags[image], keep);
}
// which palette entries do the pixels of an 8-bit indexed image point at?  One workgroup per row, the row's 256-bit answer gathered in LDS first.
__global__ void __launch_bounds__(256) k_png_used(const PngImg *imgs, const uint32_t *row_image, const uint8_t *pix, const uint32_t *flags, uint32_t *used, const uint32_t *status) {
    CSH_SHARED uint32_t s_used[8];
    const uint32_t row = blockIdx.x, image = row_image[row];
    const bool mine = !status[image] && (flags[image] & 64u);
    CSH_PHASE_LOOP(3) {
        if (!mine) continue;
        if (phase == 0) { if (threadIdx.x < 8) s_used[threadIdx.x] = 0; continue; }
        if (phase == 1) {
            const PngImg &im = imgs[image];
            const uint8_t *r = pix + im.pix_off + uint64_t(row - im.row_base) * im.rowbytes;
            uint32_t last = 256;
            for (uint32_t x = threadIdx.x; x < im.width; x += blockDim.x) { const uint32_t v = r[x]; if (v != last) { atomicOr(&s_used[v >> 5], 1u << (v & 31u)); last = v; } }
            continue;
        }
        if (threadIdx.x < 8 && s_used[threadIdx.x]) atomicOr(&used[image * 8u + threadIdx.x], s_used[threadIdx.x]);
    }
}
__global__ void __launch_bounds__(256) k_png_repack(const PngImg *imgs, const ReduceJob *jobs, const uint8_t *src, uint8_t *dst, const uint8_t *remaps) {
    const ReduceJob j = jobs[blockIdx.y];
    const PngImg &im = imgs[j.image];   // already the new geometry
    const uint32_t y = blockIdx.x;
    if (y >= im.height) return;
    const uint32_t nbps = im.bps, nk = im.channels;
    const bool opaque = j.mask & 2u, grey = j.mask & 4u;
    const uint8_t *s = src + j.src_off + uint64_t(y) * j.old_rowbytes;
    uint8_t *d = dst + j.dst_off + uint64_t(y) * im.rowbytes;
    if (j.gdepth) {   // an 8-bit grey result packed to 4, 2 or 1 bit: one lane per byte of the new row
        // (bit 8 of gdepth: the samples are palette indices -- renumbered by the job's table and packed at 1, 2, 4 or 8 bits, not scaled)
        const bool index = (j.gdepth & 256u) != 0;
        const uint32_t gd = j.gdepth & 255u, per = 8u / gd, div = index ? 1u : 255u / ((1u << gd) - 1u);
        const uint8_t *map = remaps + uint64_t(j.remap) * 256u;
        for (uint32_t bx = threadIdx.x; bx < im.rowbytes; bx += blockDim.x) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < per; k++) {
                const uint32_t x = bx * per + k;
                if (x >= im.width) break;
                const uint32_t sv = s[uint64_t(x) * j.old_channels * j.old_bps];
                v |= (index ? uint32_t(map[sv]) : sv / div) << (8u - gd - k * gd);
            }
            d[bx] = uint8_t(v);
        }
        return;
    }
    for (uint32_t x = threadIdx.x; x < im.width; x += blockDim.x) {
        uint32_t k2 = 0;
        for (uint32_t k = 0; k < j.old_channels; k++) {
            if ((opaque && k == j.old_channels - 1) || (grey && (k == 1 || k == 2))) continue;
            for (uint32_t b = 0; b < nbps; b++) d[(uint64_t(x) * nk + k2) * nbps + b] = s[(uint64_t(x) * j.old_channels + k) * j.old_bps + b];
            k2++;
        }
    }
}
void launch_png_analyze(hipStream_t st, const PngImg *imgs, uint32_t total_rows, const uint32_t *row_image, const uint8_t *pix, uint32_t *flags, const uint32_t *status) {
    if (total_rows) CSH_LAUNCH(k_png_analyze, dim3(total_rows), dim3(256), st, imgs, row_image, pix, flags, status);
}
void launch_png_repack(hipStream_t st, const PngImg *imgs, const ReduceJob *jobs, int njobs, uint32_t max_height, const uint8_t *src, uint8_t *dst, const uint8_t *remaps) {
    if (njobs) CSH_LAUNCH(k_png_repack, dim3(max_height, njobs), dim3(256), st, imgs, jobs, src, dst, remaps);
}
void launch_png_used(hipStream_t st, const PngImg *imgs, uint32_t total_rows, const uint32_t *row_image, const uint8_t *pix, const uint32_t *flags, uint32_t *used, const uint32_t *status) {
    if (total_rows) CSH_LAUNCH_PHASED(k_png_used, 3, dim3(total_rows), dim3(256), st, imgs, row_image, pix, flags, used, status);
}

// ---- colour -> palette
__device__ __forceinline__ static uint32_t pixel_key(const uint8_t *px, uint32_t ch, uint32_t bps) {   // alpha, red, green, blue (high bytes)
    return (uint32_t(ch == 4 ? px[3 * bps] : 255u) << 24) | (uint32_t(px[0]) << 16) | (uint32_t(px[bps]) << 8) | px[2 * bps];
}
__device__ __forceinline__ static uint32_t key_slot(uint32_t key) { return (key * 0x9E3779B1u) >> 22; }
__global__ void __launch_bounds__(256) k_png_colors(const PngImg *imgs, const uint32_t *row_image, const uint8_t *pix, const uint32_t *cand, unsigned long long *keys, uint32_t *counts,
                                                    const uint32_t *status) {
    const uint32_t row = blockIdx.x, image = row_image[row];
    const uint32_t ch = cand[image];
    if (!ch || status[image]) return;
    const PngImg &im = imgs[image];
    const uint32_t bps = im.bps;
    const uint8_t *r = pix + im.pix_off + uint64_t(row - im.row_base) * im.rowbytes;
    unsigned long long *tab = keys + uint64_t(image) * CSP_PAL_SLOTS;
    // Too many colours is the common answer (every photograph): the rows of such an image must learn it at once -- a row that keeps
    // inserting fills the table, and a full table costs every later key a walk over all of its slots.  So the count is read afresh
    // (not from a register the compiler kept) on entry, with every new key, and every few probes.
    auto too_many = [&]() { return coherent_load(&counts[image]) > 256u; };
    if (too_many()) return;
    uint32_t prev = 0;
    bool have_prev = false;
    for (uint32_t x = threadIdx.x; x < im.width; x += blockDim.x) {
        const uint32_t key = pixel_key(r + uint64_t(x) * ch * bps, ch, bps);
        if (have_prev && key == prev) continue;
        prev = key; have_prev = true;
        uint32_t h = key_slot(key);
        for (int probe = 0; probe < int(CSP_PAL_SLOTS); probe++, h = (h + 1) & (CSP_PAL_SLOTS - 1)) {
            const unsigned long long seen = atomicCAS(&tab[h], ~0ull, (unsigned long long)key);
            if (seen == ~0ull) { if (atomicAdd(&counts[image], 1u) >= 256u) return; break; }
            if (seen == (unsigned long long)key) break;
            if ((probe & 7) == 7 && too_many()) return;
        }
    }
}
// one lane per byte of the indexed row: the index of every pixel by hash look-up (exact palette) or by nearest entry (lossy)
__global__ void __launch_bounds__(256) k_png_indexed(const PngImg *imgs, const PaletteJob *jobs, const unsigned long long *keys, const uint16_t *slot_index, const uint32_t *palettes,
                                                     const uint8_t *src, uint8_t *dst) {
    CSH_SHARED uint32_t pal[256];
    const PaletteJob j = jobs[blockIdx.y];
    const PngImg &im = imgs[j.image];   // already the new geometry
    const uint32_t y = blockIdx.x;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) { if (j.nearest && threadIdx.x < j.npal) pal[threadIdx.x] = palettes[j.pal_off + threadIdx.x]; continue; }
        if (y >= im.height || j.nearest == 2u) continue;   // (2: the quantiser's jobs go to k_png_dither)
        const unsigned long long *tab = keys + uint64_t(j.table) * CSP_PAL_SLOTS;
        const uint16_t *idx = slot_index + uint64_t(j.table) * CSP_PAL_SLOTS;
        const uint8_t *s = src + j.src_off + uint64_t(y) * j.old_rowbytes;
        uint8_t *d = dst + j.dst_off + uint64_t(y) * im.rowbytes;
        const uint32_t per = 8u / j.depth;
        for (uint32_t bx = threadIdx.x; bx < im.rowbytes; bx += blockDim.x) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < per; k++) {
                const uint32_t x = bx * per + k;
                if (x >= im.width) break;
                const uint32_t key = pixel_key(s + uint64_t(x) * j.old_channels * j.old_bps, j.old_channels, j.old_bps);
                uint32_t index;
                if (j.nearest) {   // squared distance over a, r, g, b; ties: the lower index
                    const int a = int(key >> 24), r = int((key >> 16) & 255u), g = int((key >> 8) & 255u), b = int(key & 255u);
                    uint32_t bd = ~0u;
                    index = 0;
                    for (uint32_t q = 0; q < j.npal; q++) {
                        const uint32_t pq = pal[q];
                        const int dr = r - int((pq >> 16) & 255u), dg = g - int((pq >> 8) & 255u), db = b - int(pq & 255u), da = a - int(pq >> 24);
                        const uint32_t dist = uint32_t(dr * dr + dg * dg + db * db + da * da);
                        if (dist < bd) { bd = dist; index = q; }
                    }
                } else {
                    uint32_t h = key_slot(key);
                    while (tab[h] != (unsigned long long)key) h = (h + 1) & (CSP_PAL_SLOTS - 1);   // every pixel's colour is in the table
                    index = idx[h];
                }
                v |= index << (8u - j.depth - k * j.depth);
            }
            d[bx] = uint8_t(v);
        }
    }
}
// ---- the lossy row's pixels to palette entries with Floyd-Steinberg error diffusion (oracle: quantize).  One workgroup per picture, a lane per ROW: row r is
// two pixels behind row r - 1 (the error it needs from above -- 3 / 5 / 1 sixteenths of the upper row's pixels x + 1, x, x - 1 -- was made one, two and three
// steps ago and sits in a four-deep ring in LDS), so DITHER_ROWS (512) rows advance together, a barrier a step.  A picture taller than that goes in bands: the last row
// of a band leaves what it hands down in a line buffer in HBM (two of them, taken in turns).  The nearest entry is a search over the whole palette (256 x 4
// channels) per pixel: ~2.5 k instructions a step, which is what a step costs; a 1080p picture takes three bands of 1920 + 1024 steps.
enum { DITHER_ROWS = CSP_DITHER_ROWS };
struct DitherLds { uint32_t pal[256]; uint32_t pal_rg[256], pal_ba[256]; int16_t ring[DITHER_ROWS][4][4]; };   // pal_rg / pal_ba: the entries as halves (r, g) and (b, a); ring[row][step & 3][r g b a]
// two 16-bit halves subtracted / multiplied and summed in one instruction each (v_pk_sub_i16, v_dot2_i32_i16): a palette entry's squared distance is four
// instructions, and four entries at a time keep four chains going -- the search is what a step of k_png_dither costs
#ifdef CSH_EMUL
__device__ __forceinline__ static uint32_t dpk_sub(uint32_t a, uint32_t b) { return ((a - b) & 0xFFFFu) | ((a & 0xFFFF0000u) - (b & 0xFFFF0000u)); }
__device__ __forceinline__ static int ddot2(uint32_t a, uint32_t k, int acc) {
    return int(uint32_t(acc) + uint32_t(int(int16_t(a)) * int(int16_t(k))) + uint32_t(int(int16_t(a >> 16)) * int(int16_t(k >> 16))));
}
#else
typedef short dpk16_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ static uint32_t dpk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(dpk16_t, a) - __builtin_bit_cast(dpk16_t, b)); }
__device__ __forceinline__ static int ddot2(uint32_t a, uint32_t k, int acc) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(dpk16_t, a), __builtin_bit_cast(dpk16_t, k), acc, false); }
#endif
__global__ void __launch_bounds__(DITHER_ROWS) k_png_dither(const PngImg *imgs, const PaletteJob *jobs, const uint32_t *palettes, const uint8_t *src, uint8_t *dst, int16_t *lines, int nsteps) {
    CSH_SHARED DitherLds S;
    CSH_PERSIST(int, left, 4);      // what this row's last pixel left for the next one
    CSH_PERSIST(int, prev, 8);      // the errors of this row's last two pixels (the band's last row: what goes into the line buffer)
    CSH_PERSIST(uint32_t, acc, 1);  // the byte of indices being filled
    CSH_PERSIST(uint32_t, nextkey, 1);   // the row's next pixel, asked for a step ahead (the lanes of a wave read 64 different rows: the load is a step's longest wait)
    const PaletteJob j = jobs[blockIdx.x];
    const PngImg &im = imgs[j.image];   // already the new geometry
    const uint32_t W = im.width, H = im.height, r = threadIdx.x;
    const uint32_t per_band = W + 2u * DITHER_ROWS;
    int16_t *line0 = lines + uint64_t(j.line_off) * 4u, *line1 = line0 + uint64_t(W) * 4u;
    CSH_PHASE_LOOP(nsteps + 1) {
        if (phase == 0) {
            if (r < 256u) {
                const uint32_t pq = r < j.npal ? palettes[j.pal_off + r] : 0u;
                S.pal[r] = pq;
                // (an entry past the palette: 0x4000 in every half -- farther than any colour, its squares still inside 32 bits)
                S.pal_rg[r] = r < j.npal ? ((pq >> 16) & 255u) | (((pq >> 8) & 255u) << 16) : 0x40004000u;
                S.pal_ba[r] = r < j.npal ? (pq & 255u) | ((pq >> 24) << 16) : 0x40004000u;
            }
            CSH_UNROLL for (int k = 0; k < 4; k++) CSH_UNROLL for (int c = 0; c < 4; c++) S.ring[r][k][c] = 0;
            continue;
        }
        if (j.nearest != 2u) continue;
        const uint32_t t_all = uint32_t(phase - 1), band = t_all / per_band, t = t_all % per_band;
        const uint32_t y = band * DITHER_ROWS + r;
        const int x = int(t) - 2 * int(r);
        int e[4] = {0, 0, 0, 0};
        if (y < H && x >= 0 && x < int(W)) {
            if (x == 0) { CSH_UNROLL for (int c = 0; c < 4; c++) { left[c] = 0; prev[c] = 0; prev[4 + c] = 0; } acc[0] = 0; }
            int below[4];
            if (r == 0) {   // from the band above, through the line buffer it wrote (nothing above the first band)
                const int16_t *in = (band & 1u) ? line1 : line0;
                CSH_UNROLL
                for (int c = 0; c < 4; c++) below[c] = band ? int(coherent_load(&in[uint64_t(x) * 4u + c])) : 0;
            } else {
                const int16_t (*up)[4] = S.ring[r - 1];
                CSH_UNROLL
                for (int c = 0; c < 4; c++) below[c] = 3 * up[(t + 3u) & 3u][c] + 5 * up[(t + 2u) & 3u][c] + up[(t + 1u) & 3u][c];   // steps t - 1, t - 2, t - 3
            }
            const uint8_t *rowp = src + j.src_off + uint64_t(y) * j.old_rowbytes;
            const uint32_t key = x == 0 ? pixel_key(rowp, j.old_channels, j.old_bps) : nextkey[0];
            if (x + 1 < int(W)) nextkey[0] = pixel_key(rowp + uint64_t(x + 1) * j.old_channels * j.old_bps, j.old_channels, j.old_bps);
            const int px[4] = {int((key >> 16) & 255u), int((key >> 8) & 255u), int(key & 255u), int(key >> 24)};   // r g b a
            int want[4];
            CSH_UNROLL
            for (int c = 0; c < 4; c++) { const int v = px[c] + ((7 * left[c] + below[c] + 8) >> 4); want[c] = v < 0 ? 0 : v > 255 ? 255 : v; }
            // squared distance over a, r, g, b; ties: the lower index.  Four entries a round, each with its own running minimum (entry q in chain q & 3)
            const uint32_t wrg = uint32_t(want[0]) | (uint32_t(want[1]) << 16), wba = uint32_t(want[2]) | (uint32_t(want[3]) << 16);
            uint32_t bdk[4] = {~0u, ~0u, ~0u, ~0u}, ixk[4] = {0, 1, 2, 3};
            for (uint32_t q0 = 0; q0 < j.npal; q0 += 4) {
                CSH_UNROLL
                for (int k = 0; k < 4; k++) {
                    const uint32_t drg = dpk_sub(wrg, S.pal_rg[q0 + k]), dba = dpk_sub(wba, S.pal_ba[q0 + k]);
                    const uint32_t dist = uint32_t(ddot2(drg, drg, ddot2(dba, dba, 0)));
                    if (dist < bdk[k]) { bdk[k] = dist; ixk[k] = q0 + uint32_t(k); }
                }
            }
            uint32_t bd = bdk[0], index = ixk[0];
            CSH_UNROLL
            for (int k = 1; k < 4; k++) if (bdk[k] < bd || (bdk[k] == bd && ixk[k] < index)) { bd = bdk[k]; index = ixk[k]; }
            const uint32_t pq = S.pal[index];
            e[0] = want[0] - int((pq >> 16) & 255u); e[1] = want[1] - int((pq >> 8) & 255u); e[2] = want[2] - int(pq & 255u); e[3] = want[3] - int(pq >> 24);
            // the band's last row (when rows follow below it): what pixel x - 1 of the row below gets is complete now
            if (r == DITHER_ROWS - 1 && y + 1 < H) {
                int16_t *out = (band & 1u) ? line0 : line1;
                if (x >= 1) { CSH_UNROLL for (int c = 0; c < 4; c++) out[uint64_t(x - 1) * 4u + c] = int16_t(prev[4 + c] + 5 * prev[c] + 3 * e[c]); }
                if (x == int(W) - 1) { CSH_UNROLL for (int c = 0; c < 4; c++) out[uint64_t(x) * 4u + c] = int16_t(prev[c] + 5 * e[c]); }
            }
            CSH_UNROLL
            for (int c = 0; c < 4; c++) { prev[4 + c] = prev[c]; prev[c] = e[c]; left[c] = e[c]; }
            // the index into its byte
            const uint32_t d = j.depth, bit = uint32_t(x) * d;
            acc[0] |= index << (8u - d - (bit & 7u));
            if (((bit + d) & 7u) == 0 || x == int(W) - 1) { dst[j.dst_off + uint64_t(y) * im.rowbytes + (bit >> 3)] = uint8_t(acc[0]); acc[0] = 0; }
        }
        CSH_UNROLL
        for (int c = 0; c < 4; c++) S.ring[r][t & 3u][c] = int16_t(e[c]);
        if (t + 1 == per_band) CSP_MEM_FENCE();   // the band's line buffer is complete before the next band's first row reads it (the barrier follows)
    }
}
// ---- one lane per pixel: any PNG format to interleaved 8-bit samples (RgbJob).  16-bit samples round as image-rs converts them
// ((v + 128) / 257 [UPSTREAM-RECALL]); sub-byte grey scales to the full range; a palette index past the PLTE decodes as black
__global__ void __launch_bounds__(256) k_png_rgb(const RgbJob *jobs, const uint8_t *plte, const uint8_t *work, uint8_t *rgb, const uint32_t *status) {
    const RgbJob j = jobs[blockIdx.y];
    const uint32_t y = blockIdx.x;
    if (y >= j.height || status[j.image]) return;
    const uint8_t *r = work + j.src_off + uint64_t(y) * j.rowbytes;
    const uint8_t *pal = plte + j.plte_off, *trns = plte + j.trns_off;
    const uint32_t nc = j.out_nc, in_nc = j.ctype == 2 ? 3u : j.ctype == 4 ? 2u : j.ctype == 6 ? 4u : 1u, bps = j.depth == 16 ? 2u : 1u;
    const bool alpha_out = nc == 2 || nc == 4;
    uint8_t *d = rgb + j.dst_off + uint64_t(y) * j.width * nc;
    auto sample = [&](const uint8_t *p) { return bps == 2 ? uint32_t(((((uint32_t(p[0]) << 8) | p[1]) + 128u) / 257u)) : uint32_t(p[0]); };
    auto raw16 = [&](const uint8_t *p) { return bps == 2 ? ((uint32_t(p[0]) << 8) | p[1]) : uint32_t(p[0]); };
    auto key = [&](int c) { return (uint32_t(trns[2 * c]) << 8) | trns[2 * c + 1]; };
    if (j.wide) {   // 16-bit grey / RGB + tRNS -> 16-bit grey + alpha / RGBA (what the png crate's EXPAND hands image-rs: La16 / Rgba16): samples as they are, alpha 0 at the key
        const uint32_t colour = j.ctype == 2 ? 3u : 1u;
        uint8_t *dw = rgb + j.dst_off + uint64_t(y) * j.width * nc * 2u;
        for (uint32_t x = threadIdx.x; x < j.width; x += blockDim.x) {
            const uint8_t *p = r + uint64_t(x) * colour * 2u;
            uint8_t *o = dw + uint64_t(x) * nc * 2u;
            bool hit = j.ntrns != 0;
            for (uint32_t c = 0; c < colour; c++) { o[2 * c] = p[2 * c]; o[2 * c + 1] = p[2 * c + 1]; hit = hit && raw16(p + 2 * c) == key(int(c)); }
            o[2 * colour] = o[2 * colour + 1] = hit ? 0 : 255;
        }
        return;
    }
    for (uint32_t x = threadIdx.x; x < j.width; x += blockDim.x) {
        uint8_t *o = d + uint64_t(x) * nc;
        if (j.ctype == 2 || j.ctype == 6 || j.ctype == 4 || (j.ctype == 0 && j.depth >= 8)) {
            const uint8_t *p = r + uint64_t(x) * in_nc * bps;
            const uint32_t colour = j.ctype == 2 || j.ctype == 6 ? 3u : 1u;
            for (uint32_t c = 0; c < colour; c++) o[c] = uint8_t(sample(p + c * bps));
            if (in_nc > colour) { if (alpha_out) o[colour] = uint8_t(sample(p + colour * bps)); }
            else if (alpha_out) {
                bool hit = j.ntrns != 0;
                for (uint32_t c = 0; c < colour && hit; c++) hit = raw16(p + c * bps) == key(int(c));
                o[colour] = hit ? 0 : 255;
            }
            continue;
        }
        const uint32_t per = 8u / j.depth, k = x % per;
        const uint32_t v = (uint32_t(r[x / per]) >> (8u - j.depth - k * j.depth)) & ((1u << j.depth) - 1u);   // depth 8 too: per = 1
        if (j.ctype == 3) {
            if (v < j.npal) { o[0] = pal[3 * v]; o[1] = pal[3 * v + 1]; o[2] = pal[3 * v + 2]; } else { o[0] = 0; o[1] = 0; o[2] = 0; }
            if (alpha_out) o[3] = v < j.ntrns ? trns[v] : 255;
        } else {
            o[0] = uint8_t(v * (255u / ((1u << j.depth) - 1u)));
            if (alpha_out) o[1] = (j.ntrns && v == key(0)) ? 0 : 255;
        }
    }
}
// ---- lossy PNG: colour bins
// colour bins of an image: count and channel sums per bin.  A workgroup takes QH_ROWS rows and gathers them in an LDS table first (neighbouring pixels share
// bins: a 1080p photograph's row touches a few hundred): five global atomics per DISTINCT bin of the rows instead of five per pixel (88 ms per 96
// 1080p files before).  A pixel that finds no slot within QH_PROBES steps goes to the global bins directly; all sums are integers, so neither the slot a bin lands in
// nor the order of the additions shows in the result.
enum { QH_ROWS = 8, QH_SLOTS = 2048, QH_PROBES = 16 };
__global__ void __launch_bounds__(256) k_png_qhist(const QuantJob *jobs, const uint8_t *work, uint32_t *bins) {
    CSH_SHARED uint32_t s_key[QH_SLOTS];
    CSH_SHARED uint32_t s_val[QH_SLOTS][5];
    const QuantJob j = jobs[blockIdx.y];
    const uint32_t y0 = blockIdx.x * QH_ROWS;
    uint32_t *b = bins + j.bins_off;
    CSH_PHASE_LOOP(3) {
        if (y0 >= j.height) continue;
        if (phase == 0) {
            for (uint32_t k = threadIdx.x; k < QH_SLOTS; k += blockDim.x) { s_key[k] = 0xFFFFFFFFu; for (int c = 0; c < 5; c++) s_val[k][c] = 0; }
            continue;
        }
        if (phase == 1) {
            for (uint32_t y = y0; y < y0 + QH_ROWS && y < j.height; y++) {
                const uint8_t *r = work + j.src_off + uint64_t(y) * j.rowbytes;
                for (uint32_t x = threadIdx.x; x < j.width; x += blockDim.x) {
                    const uint32_t key = pixel_key(r + uint64_t(x) * j.channels * j.bps, j.channels, j.bps);
                    const uint32_t a = key >> 24, rr = (key >> 16) & 255u, g = (key >> 8) & 255u, bb = key & 255u;
                    const uint32_t id = ((a >> 4) << 15) | ((rr >> 3) << 10) | ((g >> 3) << 5) | (bb >> 3);
                    uint32_t h = (id * 2654435761u) >> 21;
                    int probes = 0;
                    for (; probes < QH_PROBES; probes++, h = (h + 1u) & (QH_SLOTS - 1u)) {
                        const uint32_t seen = atomicCAS(&s_key[h], 0xFFFFFFFFu, id);
                        if (seen == 0xFFFFFFFFu || seen == id) break;
                    }
                    if (probes < QH_PROBES) { uint32_t *q = s_val[h]; atomicAdd(&q[0], 1u); atomicAdd(&q[1], rr); atomicAdd(&q[2], g); atomicAdd(&q[3], bb); atomicAdd(&q[4], a); }
                    else { uint32_t *q = b + uint64_t(id) * 5; atomicAdd(&q[0], 1u); atomicAdd(&q[1], rr); atomicAdd(&q[2], g); atomicAdd(&q[3], bb); atomicAdd(&q[4], a); }
                }
            }
            continue;
        }
        for (uint32_t k = threadIdx.x; k < QH_SLOTS; k += blockDim.x) {
            if (s_key[k] == 0xFFFFFFFFu) continue;
            uint32_t *q = b + uint64_t(s_key[k]) * 5;
            for (int c = 0; c < 5; c++) atomicAdd(&q[c], s_val[k][c]);
        }
    }
}
__global__ void __launch_bounds__(256) k_png_qcompact(const QuantJob *jobs, const uint32_t *bins, QBin *list, uint32_t *nlist) {
    const QuantJob j = jobs[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CSP_QBINS) return;
    const uint32_t *q = bins + j.bins_off + uint64_t(i) * 5;
    if (!q[0]) return;
    const uint32_t slot = atomicAdd(&nlist[blockIdx.y], 1u);
    QBin o; o.id = i; o.cnt = q[0]; o.s[0] = q[1]; o.s[1] = q[2]; o.s[2] = q[3]; o.s[3] = q[4];
    list[j.list_off + slot] = o;
}
// ---- lossy PNG: the median cut over the colour bins of one image (oracle: median_cut), one workgroup per image.
// A box is a set of bins: a contiguous range of an index array (the order inside does not matter).  A split = the most populous live box, its widest channel
// (of the bin means), and the weighted median of the box's bins in the order (mean on that channel, bin id): the lower part ends with the bin at which it
// first holds half the box's pixels (never with the box's last bin).  No sort: the 256-level histogram of the channel finds the level the median falls
// in, a histogram over 1024 ranges of bin ids inside that level the range, and the range's 512 ids are slots of their own -- three counting passes; the
// split then moves the box's indices into the other index array, lower part from the front, upper part from the back.  Every step is a pass of all
// lanes over the BOX's bins (so the 255 splits together visit each bin about ten times, not 255 times) between decisions that a prefix per lane takes
// (lane l sums the entries up to l; the one lane at which the sum crosses half the box writes the answer).  All sums are integers: the order of the
// atomics does not matter, nor does the order of the bin list.  The cut stops at 256 boxes, when nothing can be split, or -- imagequant's
// set_quality(0, q) -- when the squared error of the bin means against the rounded box means is within the bound of the quality.
// Phases: 5 of set-up (0..4), then MC_STEP per iteration (a split, or the retirement of a box whose bins all share one mean), at most 255 + 256 iterations.
enum { MC_LANES = 1024, MC_STEP = 14, MC_ITER = 511, MC_PHASES = 5 + MC_STEP * MC_ITER };
static_assert(CSP_QBINS == 1024 * 512, "the id ranges of the median search");
struct McRec { uint32_t mean, cnt, id, pad; };            // a bin as the passes read it: its four channel means in one word, its pixels, its id
struct McLds {
    unsigned long long cnt[256], err[256], sum[256][4];   // per box: pixels, error, channel sums
    uint32_t lo[256], hi[256];                            // per box: its range of the index array ...
    uint32_t buf[256];                                    // ... and which of the two arrays holds it
    uint32_t dead[256];
    unsigned long long lev[256];                          // pixels per level of the axis, in the picked box
    unsigned long long rng[1024];                         // ... per range of 512 bin ids, inside the median's level
    unsigned long long slot[512];                         // ... per bin id, inside the median's range
    unsigned long long psum[2][4], pcnt[2], perr[2];      // the two parts of a split
    uint32_t pnb[2], pmean[2][4];
    uint32_t mn[4], mx[4];
    unsigned long long total, total_err;
    unsigned long long below, below_range;                // pixels in front of the median's level; ... and of its id range (two words: the lanes of the range search read the first while one of them writes the second)
    unsigned long long best;                              // the picked box's pixels
    int nbox, pick, axis, cut, done, skip;                // cut: the median's level
    int range, cut_id, cut_incl, last_slot;               // the median's id range; the lower part ends with bin cut_id (cut_incl) or just in front of it
};
__global__ void __launch_bounds__(MC_LANES) k_png_mediancut(const QuantJob *jobs, const QBin *list, const uint32_t *nlist, uint4 *recs, uint32_t *order, uint64_t order_stride, int quality,
                                                          unsigned long long bound, uint32_t *pal_out, uint32_t *npal_out) {
    CSH_SHARED McLds L;
    const QuantJob j = jobs[blockIdx.x];
    const uint32_t n = nlist[blockIdx.x], tid = threadIdx.x;
    const QBin *bins = list + j.list_off;
    McRec *rec = reinterpret_cast<McRec *>(recs + j.list_off);
    uint32_t *ord[2] = {order + j.list_off, order + order_stride + j.list_off};
    auto mean_of = [&](uint32_t m, int c) -> uint32_t { return (m >> (8 * c)) & 255u; };
    // a bin's error in a part: its mean against the part's rounded mean
    auto error_of = [&](int part, const McRec &r) {
        unsigned long long d2 = 0;
        for (int c = 0; c < 4; c++) { const long long d = (long long)mean_of(r.mean, c) - (long long)L.pmean[part][c]; d2 += (unsigned long long)(d * d); }
        return d2 * r.cnt;
    };
    CSH_PHASE_LOOP(MC_PHASES + 1) {
        if (phase == 0) {
            if (tid < 256) { L.cnt[tid] = 0; L.err[tid] = 0; L.lo[tid] = 0; L.hi[tid] = 0; L.buf[tid] = 0; L.dead[tid] = 0; for (int c = 0; c < 4; c++) L.sum[tid][c] = 0; }
            if (tid == 0) { L.nbox = 1; L.done = n == 0 ? 1 : 0; L.skip = 0; L.perr[0] = L.perr[1] = 0; }
            continue;
        }
        if (phase == 1) {   // the bins as records; box 0 = everything
            unsigned long long c0 = 0, s0[4] = {0, 0, 0, 0};
            for (uint32_t i = tid; i < n; i += MC_LANES) {
                const QBin q = bins[i];
                McRec r;
                r.mean = (q.s[0] / q.cnt) | ((q.s[1] / q.cnt) << 8) | ((q.s[2] / q.cnt) << 16) | ((q.s[3] / q.cnt) << 24); r.cnt = q.cnt; r.id = q.id; r.pad = 0;
                rec[i] = r;
                ord[0][i] = i;
                c0 += q.cnt; for (int c = 0; c < 4; c++) s0[c] += q.s[c];
            }
            if (c0) { atomicAdd(&L.cnt[0], c0); for (int c = 0; c < 4; c++) atomicAdd(&L.sum[0][c], s0[c]); }
            continue;
        }
        if (phase == 2) {
            if (tid == 0 && n) { L.hi[0] = n; L.total = L.cnt[0]; for (int c = 0; c < 4; c++) L.pmean[0][c] = uint32_t((2 * L.sum[0][c] + L.cnt[0]) / (2 * L.cnt[0])); }
            continue;
        }
        if (phase == 3) {
            unsigned long long e = 0;
            for (uint32_t i = tid; i < n; i += MC_LANES) e += error_of(0, rec[i]);
            if (e) atomicAdd(&L.perr[0], e);
            continue;
        }
        if (phase == 4) { if (tid == 0) { L.err[0] = L.perr[0]; L.total_err = L.perr[0]; L.best = 0; L.pick = -1; } continue; }
        if (phase == MC_PHASES) {   // palette entries (unsorted: the host sorts them and merges equal ones)
            if (tid < 256) {
                uint32_t v = 0;
                if (int(tid) < L.nbox && n) {
                    uint32_t m[4];
                    for (int c = 0; c < 4; c++) m[c] = uint32_t((2 * L.sum[tid][c] + L.cnt[tid]) / (2 * L.cnt[tid]));
                    v = (m[3] << 24) | (m[0] << 16) | (m[1] << 8) | m[2];
                }
                pal_out[blockIdx.x * 256 + tid] = v;
            }
            if (tid == 0) npal_out[blockIdx.x] = n ? uint32_t(L.nbox) : 0u;
            continue;
        }
        if (L.done) continue;
        const int sub = (phase - 5) % MC_STEP;
        if (sub == 0) {          // good enough?  otherwise the most populous box that can be split (the first of equals): every box bids its pixels
            if (tid < 256) L.lev[tid] = 0;
            L.rng[tid] = 0;
            if (tid < 512) L.slot[tid] = 0;
            if (tid < 4) { L.mn[tid] = 255; L.mx[tid] = 0; }
            const int q = quality < 0 ? 0 : quality > 100 ? 100 : quality;
            const bool enough = L.nbox >= 256 || (L.nbox >= 2 && (q == 0 || L.total_err * 1024ull <= bound * L.total));
            if (tid == 0) { L.skip = 0; L.pick = -1; L.best = 0; if (enough) L.done = 1; }
            continue;
        }
        if (sub == 1) {          // (the bids: a box that can be split has pixels, so 0 means none)
            if (int(tid) < L.nbox && !L.dead[tid] && L.hi[tid] - L.lo[tid] > 1) atomicMax(&L.best, L.cnt[tid]);
            continue;
        }
        if (sub == 2) {          // the first box with that many pixels
            if (int(tid) < L.nbox && !L.dead[tid] && L.hi[tid] - L.lo[tid] > 1 && L.cnt[tid] == L.best) {
                bool first = true;
                for (int k = 0; k < int(tid); k++) if (!L.dead[k] && L.hi[k] - L.lo[k] > 1 && L.cnt[k] == L.best) { first = false; break; }
                if (first) L.pick = int(tid);
            }
            continue;
        }
        const int pick = L.pick;
        if (pick < 0) { if (tid == 0) L.done = 1; continue; }
        if (L.skip) continue;
        const uint32_t lo = L.lo[pick], hi = L.hi[pick];
        const uint32_t *src = ord[L.buf[pick]];
        if (sub == 3) {          // the box's extent in every channel
            uint32_t mn[4] = {255, 255, 255, 255}, mx[4] = {0, 0, 0, 0};
            bool any = false;
            for (uint32_t k = lo + tid; k < hi; k += MC_LANES) {
                const uint32_t m = rec[src[k]].mean;
                any = true;
                for (int c = 0; c < 4; c++) { const uint32_t v = mean_of(m, c); mn[c] = min(mn[c], v); mx[c] = max(mx[c], v); }
            }
            if (any) for (int c = 0; c < 4; c++) { atomicMin(&L.mn[c], mn[c]); atomicMax(&L.mx[c], mx[c]); }
            continue;
        }
        if (sub == 4) {          // the widest channel (the first of equals); a box of one colour is retired
            if (tid == 0) {
                int axis = 0, range = -1;
                for (int c = 0; c < 4; c++) if (int(L.mx[c]) - int(L.mn[c]) > range) { range = int(L.mx[c]) - int(L.mn[c]); axis = c; }
                L.axis = axis;
                if (range == 0) { L.dead[pick] = 1; L.skip = 1; }
            }
            continue;
        }
        const int axis = L.axis;
        if (sub == 5) {          // pixels per level of that channel
            for (uint32_t k = lo + tid; k < hi; k += MC_LANES) { const McRec r = rec[src[k]]; atomicAdd(&L.lev[mean_of(r.mean, axis)], (unsigned long long)r.cnt); }
            continue;
        }
        const unsigned long long half = L.cnt[pick];   // compared with twice a running sum
        if (sub == 6) {          // the level the median falls in: the first at which the lower levels + this one hold half the box's pixels
            if (tid < 256 && L.lev[tid]) {
                unsigned long long run = 0;
                for (uint32_t l = 0; l < tid; l++) run += L.lev[l];
                if (2 * run < half && 2 * (run + L.lev[tid]) >= half) { L.cut = int(tid); L.below = run; }
            }
            continue;
        }
        const int cut = L.cut;
        if (sub == 7) {          // inside that level: pixels per range of bin ids
            for (uint32_t k = lo + tid; k < hi; k += MC_LANES) { const McRec r = rec[src[k]]; if (int(mean_of(r.mean, axis)) == cut) atomicAdd(&L.rng[r.id >> 9], (unsigned long long)r.cnt); }
            continue;
        }
        if (sub == 8) {
            if (L.rng[tid]) {
                unsigned long long run = L.below;
                for (uint32_t r = 0; r < tid; r++) run += L.rng[r];
                if (2 * run < half && 2 * (run + L.rng[tid]) >= half) { L.range = int(tid); L.below_range = run; }
            }
            continue;
        }
        if (sub == 9) {          // inside that range: every bin id is a slot
            const uint32_t rg = uint32_t(L.range);
            for (uint32_t k = lo + tid; k < hi; k += MC_LANES) { const McRec r = rec[src[k]]; if (int(mean_of(r.mean, axis)) == cut && (r.id >> 9) == rg) L.slot[r.id & 511u] = r.cnt; }
            if (tid == 0) L.last_slot = -1;
            continue;
        }
        if (sub == 10) {         // the bin the lower part ends with -- or, if that is the box's last bin, ends in front of
            if (tid < 512 && L.slot[tid]) {
                unsigned long long run = L.below_range;
                for (uint32_t k = 0; k < tid; k++) run += L.slot[k];
                if (2 * run < half && 2 * (run + L.slot[tid]) >= half) {
                    bool is_last = true;                       // of the range; of the box if nothing lies behind the range and the level either
                    for (uint32_t k = tid + 1; k < 512 && is_last; k++) if (L.slot[k]) is_last = false;
                    for (int r = L.range + 1; r < 1024 && is_last; r++) if (L.rng[r]) is_last = false;
                    for (int l = cut + 1; l < 256 && is_last; l++) if (L.lev[l]) is_last = false;
                    L.cut_id = L.range * 512 + int(tid); L.cut_incl = is_last ? 0 : 1;
                }
            }
            if (tid == 0) for (int p = 0; p < 2; p++) { L.pcnt[p] = 0; L.perr[p] = 0; L.pnb[p] = 0; for (int c = 0; c < 4; c++) L.psum[p][c] = 0; }
            continue;
        }
        uint32_t *dst = ord[L.buf[pick] ^ 1u];
        if (sub == 11) {         // the split: indices into the other array (lower part from the front of the range, upper part from its back); both parts' sums
            const uint32_t cut_id = uint32_t(L.cut_id), incl = uint32_t(L.cut_incl);
            unsigned long long pc[2] = {0, 0}, ps[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            for (uint32_t k = lo + tid; k < hi; k += MC_LANES) {
                const uint32_t i = src[k];
                const McRec r = rec[i];
                const int lv = int(mean_of(r.mean, axis));
                const int part = (lv > cut || (lv == cut && (incl ? r.id > cut_id : r.id >= cut_id))) ? 1 : 0;
                const uint32_t at = atomicAdd(&L.pnb[part], 1u);
                dst[part ? hi - 1 - at : lo + at] = i;
                const QBin q = bins[i];
                pc[part] += q.cnt;
                for (int c = 0; c < 4; c++) ps[part][c] += q.s[c];
            }
            for (int p = 0; p < 2; p++) if (pc[p]) { atomicAdd(&L.pcnt[p], pc[p]); for (int c = 0; c < 4; c++) atomicAdd(&L.psum[p][c], ps[p][c]); }
            continue;
        }
        if (sub == 12) {         // both parts' errors (every lane works the parts' rounded means out for itself from the finished sums)
            uint32_t pm[2][4];
            for (int p = 0; p < 2; p++) for (int c = 0; c < 4; c++) pm[p][c] = uint32_t((2 * L.psum[p][c] + L.pcnt[p]) / (2 * L.pcnt[p]));
            const uint32_t nl = L.pnb[0];
            unsigned long long e[2] = {0, 0};
            for (uint32_t k = lo + tid; k < hi; k += MC_LANES) {
                const McRec r = rec[dst[k]];
                const int part = k - lo < nl ? 0 : 1;
                unsigned long long d2 = 0;
                for (int c = 0; c < 4; c++) { const long long d = (long long)mean_of(r.mean, c) - (long long)pm[part][c]; d2 += (unsigned long long)(d * d); }
                e[part] += d2 * r.cnt;
            }
            for (int p = 0; p < 2; p++) if (e[p]) atomicAdd(&L.perr[p], e[p]);
            continue;
        }
        if (tid == 0) {          // sub == 13: the box table
            const int nbox = L.nbox;
            const uint32_t mid = lo + L.pnb[0], nb = L.buf[pick] ^ 1u;
            L.total_err = L.total_err - L.err[pick] + L.perr[0] + L.perr[1];
            L.cnt[pick] = L.pcnt[0]; L.err[pick] = L.perr[0]; L.hi[pick] = mid; L.buf[pick] = nb;
            L.cnt[nbox] = L.pcnt[1]; L.err[nbox] = L.perr[1]; L.lo[nbox] = mid; L.hi[nbox] = hi; L.buf[nbox] = nb;
            for (int c = 0; c < 4; c++) { L.sum[pick][c] = L.psum[0][c]; L.sum[nbox][c] = L.psum[1][c]; }
            L.nbox = nbox + 1;
        }
    }
}
void launch_png_mediancut(hipStream_t st, const QuantJob *jobs, int njobs, const QBin *list, const uint32_t *nlist, uint4 *recs, uint32_t *order, uint64_t order_stride, int quality,
                          unsigned long long bound, uint32_t *pal, uint32_t *npal) {
    if (njobs) CSH_LAUNCH_PHASED(k_png_mediancut, MC_PHASES + 1, dim3(unsigned(njobs)), dim3(MC_LANES), st, jobs, list, nlist, recs, order, order_stride, quality, bound, pal, npal);
}
void launch_png_colors(hipStream_t st, const PngImg *imgs, uint32_t total_rows, const uint32_t *row_image, const uint8_t *pix, const uint32_t *cand, unsigned long long *keys,
                       uint32_t *counts, const uint32_t *status) {
    if (total_rows) CSH_LAUNCH(k_png_colors, dim3(total_rows), dim3(256), st, imgs, row_image, pix, cand, keys, counts, status);
}
void launch_png_indexed(hipStream_t st, const PngImg *imgs, const PaletteJob *jobs, int njobs, uint32_t max_height, const unsigned long long *keys, const uint16_t *slot_index,
                        const uint32_t *palettes, const uint8_t *src, uint8_t *dst) {
    if (njobs) CSH_LAUNCH_PHASED(k_png_indexed, 2, dim3(max_height, njobs), dim3(256), st, imgs, jobs, keys, slot_index, palettes, src, dst);
}
// nsteps: the largest (bands x steps per band) of the jobs that dither; lines: two rows of width x 4 int16 per such job (PaletteJob::line_off, in pixels)
void launch_png_dither(hipStream_t st, const PngImg *imgs, const PaletteJob *jobs, int njobs, int nsteps, const uint32_t *palettes, const uint8_t *src, uint8_t *dst, int16_t *lines) {
    if (njobs && nsteps) CSH_LAUNCH_PHASED(k_png_dither, nsteps + 1, dim3(unsigned(njobs)), dim3(DITHER_ROWS), st, imgs, jobs, palettes, src, dst, lines, nsteps);
}
void launch_png_rgb(hipStream_t st, const RgbJob *jobs, int njobs, uint32_t max_height, const uint8_t *plte, const uint8_t *work, uint8_t *rgb, const uint32_t *status) {
    if (njobs) CSH_LAUNCH(k_png_rgb, dim3(max_height, njobs), dim3(256), st, jobs, plte, work, rgb, status);
}
void launch_png_qhist(hipStream_t st, const QuantJob *jobs, int njobs, uint32_t max_height, const uint8_t *work, uint32_t *bins) {
    if (njobs) CSH_LAUNCH_PHASED(k_png_qhist, 3, dim3((max_height + QH_ROWS - 1) / QH_ROWS, njobs), dim3(256), st, jobs, work, bins);
}
void launch_png_qcompact(hipStream_t st, const QuantJob *jobs, int njobs, const uint32_t *bins, QBin *list, uint32_t *nlist) {
    if (njobs) CSH_LAUNCH(k_png_qcompact, dim3(CSP_QBINS / 256, njobs), dim3(256), st, jobs, bins, list, nlist);
}

// one workgroup per batch row: all five filtered versions of the row
__global__ void __launch_bounds__(256) k_png_filter5(FilterCtx c) {
    const uint32_t row = blockIdx.x;
    const uint32_t image = c.row_image[row];
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint32_t y = row - im.row_base, W = im.rowbytes, bpp = im.bpp;
    const uint8_t *cur = c.pix + im.pix_off + uint64_t(y) * W, *up = y ? cur - W : nullptr;
    uint8_t *dst = c.streams + im.stream_off + uint64_t(y) * (W + 1);
    if (threadIdx.x < 5) dst[uint64_t(threadIdx.x) * im.stream_stride] = uint8_t(threadIdx.x);
    for (uint32_t x = threadIdx.x; x < W; x += blockDim.x) {
        const int v = cur[x], a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, cc = (up && x >= bpp) ? up[x - bpp] : 0;
        uint8_t *o = dst + 1 + x;
        o[0] = uint8_t(v);
        o[im.stream_stride] = uint8_t(v - a);
        o[2 * im.stream_stride] = uint8_t(v - b);
        o[3 * im.stream_stride] = uint8_t(v - ((a + b) >> 1));
        o[4 * im.stream_stride] = uint8_t(v - paeth_f(a, b, cc));
    }
}

// one workgroup per batch row; per filter three barrier-separated steps: the byte histogram (+ MinSum); the histogram's entropy (256 lanes) together
// with the pair counts; the pairs read back.  The 65536 pair counters are 16 bits wide (a row has at most 2^16 - 1 pairs: rows longer than that take the
// two-halves path below), two to a 32-bit LDS word -- 128 KiB, the whole key space in one pass -- and are read back by exchanging the WORD with zero: whoever
// gets there first accounts for both of its counters and leaves the table clean for the next filter.  Every filter has accumulators of its own, written
// out in one last step.  (Round 6: 17 steps a row instead of 36 -- the steps are short, eleven bytes per lane for a 4K row, and a step is a barrier.)
#ifndef CSP_SCORE_THREADS
#define CSP_SCORE_THREADS 1024
#endif
enum { SCORE_THREADS = CSP_SCORE_THREADS };
struct ScoreLds {
    uint32_t pair[32768];
    uint32_t hist[256];
    unsigned long long acc[5][4];   // per filter: minsum, entropy, distinct, bigent
};
template <bool WIDE>   // WIDE: some row of the batch has more than 65536 bytes -- counts may pass 16 bits: 32-bit counters, the key space in two halves, six steps per filter
__global__ void __launch_bounds__(SCORE_THREADS) k_png_scores(FilterCtx c) {
    constexpr int SCORE_STEPS = WIDE ? 6 : 3;
    CSH_SHARED ScoreLds S;
    const uint32_t row = blockIdx.x;
    const uint32_t image = c.row_image[row];
    const PngImg &im = c.imgs[image];
    const uint32_t y = row - im.row_base, W = im.rowbytes, n = W + 1;   // n bytes: type byte + data
    const bool dead = c.status[image] != 0;
    CSH_PHASE_LOOP(2 + 5 * SCORE_STEPS) {
        if (phase == 0) {
            for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) S.pair[i] = 0;
            if (threadIdx.x < 256) S.hist[threadIdx.x] = 0;
            if (threadIdx.x < 20) S.acc[threadIdx.x >> 2][threadIdx.x & 3] = 0;
            continue;
        }
        if (dead) continue;
        if (phase == 1 + 5 * SCORE_STEPS) {
            if (threadIdx.x < 20) c.scores[(uint64_t(row) * 5 + (threadIdx.x >> 2)) * 5 + (threadIdx.x & 3)] = S.acc[threadIdx.x >> 2][threadIdx.x & 3];
            continue;
        }
        const int f = (phase - 1) / SCORE_STEPS, step = (phase - 1) % SCORE_STEPS;
        const uint8_t *r = c.streams + im.stream_off + uint64_t(f) * im.stream_stride + uint64_t(y) * n;
        if (step == 0) {
            unsigned long long ms = 0;
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                const uint32_t b = r[i];
                atomicAdd(&S.hist[b], 1u);
                if (i) ms += b < 128 ? b : 256 - b;
            }
            if (ms) atomicAdd(&S.acc[f][0], ms);
        } else if (step == 1) {
            if (threadIdx.x < 256) { const uint32_t h = S.hist[threadIdx.x]; S.hist[threadIdx.x] = 0; if (h) atomicAdd(&S.acc[f][1], (unsigned long long)ilog2i(h)); }
            if (!WIDE)
                for (uint32_t i = threadIdx.x; i + 1 < n; i += blockDim.x) {
                    const uint32_t k = (uint32_t(r[i]) << 8) | r[i + 1];
                    atomicAdd(&S.pair[k >> 1], (k & 1u) ? 65536u : 1u);
                }
        } else if (!WIDE) {
            unsigned long long d = 0, e = 0;
            for (uint32_t i = threadIdx.x; i + 1 < n; i += blockDim.x) {
                const uint32_t k = (uint32_t(r[i]) << 8) | r[i + 1];
                const uint32_t both = atomicExch(&S.pair[k >> 1], 0u);
                if (both & 0xFFFFu) { d++; e += ilog2i(both & 0xFFFFu); }
                if (both >> 16) { d++; e += ilog2i(both >> 16); }
            }
            if (d) { atomicAdd(&S.acc[f][2], d); atomicAdd(&S.acc[f][3], e); }
        } else if (step == 2 || step == 4) {
            const uint32_t half = step == 2 ? 0u : 1u;
            for (uint32_t i = threadIdx.x; i + 1 < n; i += blockDim.x) {
                const uint32_t k = (uint32_t(r[i]) << 8) | r[i + 1];
                if ((k >> 15) == half) atomicAdd(&S.pair[k & 32767u], 1u);
            }
        } else {
            const uint32_t half = step == 3 ? 0u : 1u;
            unsigned long long d = 0, e = 0;
            for (uint32_t i = threadIdx.x; i + 1 < n; i += blockDim.x) {
                const uint32_t k = (uint32_t(r[i]) << 8) | r[i + 1];
                if ((k >> 15) == half) { const uint32_t cnt = atomicExch(&S.pair[k & 32767u], 0u); if (cnt) { d++; e += ilog2i(cnt); } }
            }
            if (d) { atomicAdd(&S.acc[f][2], d); atomicAdd(&S.acc[f][3], e); }
        }
    }
}

// Brute score (strategy 9): one wave per (row, filter) tokenizes the candidate row as a chunk of its own and estimates
// its size under a code of its own: n log n - sum c log c over both alphabets, plus the extra bits.
struct BruteLds { LzLds lz; uint32_t hist[CSP_NSYM]; };
struct BruteSink {
    uint32_t *hist;
    LV<uint32_t> extra, nl, nd;
    __device__ __forceinline__ void tile(uint64_t, uint32_t, uint64_t taken, const LV<uint32_t> &mlen, const LV<uint32_t> &mdist, const LV<uint32_t> &lit) {
        LFOR(l) if ((taken >> l) & 1) {
            nl[l]++;
            if (mlen[l]) {
                const uint32_t lc = len_code_of(mlen[l]), dc = dist_code_of(mdist[l]);
                atomicAdd(&hist[257 + lc], 1u); atomicAdd(&hist[CSP_NLIT + dc], 1u);
                extra[l] += len_extra_of(lc) + dist_extra_of(dc); nd[l]++;
            } else
                atomicAdd(&hist[lit[l]], 1u);
        }
    }
};
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_png_brute(FilterCtx c) {
    CSH_SHARED BruteLds S;
    const uint32_t row = blockIdx.x, f = blockIdx.y;
    const uint32_t image = c.row_image[row];
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint32_t y = row - im.row_base;
    uint64_t n = uint64_t(im.rowbytes) + 1;
    const uint8_t *r = c.streams + im.stream_off + uint64_t(f) * im.stream_stride + uint64_t(y) * n;
    if (n > CSP_CHUNK) n = CSP_CHUNK;
    LFOR(l) for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) S.hist[i] = 0;
    BruteSink sink; sink.hist = S.hist;
    LFOR(l) { sink.extra[l] = 0; sink.nl[l] = 0; sink.nd[l] = 0; }
    CSP_WAVE_SYNC();
    lz_chunk(r, n, 0, n, S.lz, sink);
    CSP_WAVE_SYNC();
    LV<uint64_t> part;
    LFOR(l) { uint64_t s = 0; for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) s += ilog2i(S.hist[i]); part[l] = s; }
    const uint64_t sub = lsum(part);
    LFOR(l) part[l] = sink.extra[l];
    const uint64_t extra = lsum(part);
    LFOR(l) part[l] = sink.nl[l];
    const uint64_t nl = lsum(part);
    LFOR(l) part[l] = sink.nd[l];
    const uint64_t nd = lsum(part);
    LFOR(l) if (l == 0) c.scores[(uint64_t(row) * 5 + f) * 5 + 4] = ilog2i(nl) + ilog2i(nd) + extra - sub;
}

// per row: the adaptive strategies' choices (ties: the lower filter), and their streams gathered out of the fixed five
__global__ void __launch_bounds__(256) k_png_pick(FilterCtx c) {
    const uint32_t row = blockIdx.x;
    const uint32_t image = c.row_image[row];
    if (c.status[image]) return;
    const PngImg &im = c.imgs[image];
    const uint32_t y = row - im.row_base, n = im.rowbytes + 1;
    const uint64_t *sc = c.scores + uint64_t(row) * 25;
    for (int a = 0; a < c.plan.nadaptive; a++) {
        const int s = c.plan.adaptive_strategy[a], k = s - 5;
        const bool more_wins = s == 6 || s == 8;
        int pick = 0;
        uint64_t best = sc[k];
        for (int f = 1; f < 5; f++) { const uint64_t v = sc[f * 5 + k]; if (more_wins ? v > best : v < best) { best = v; pick = f; } }
        if (threadIdx.x == 0) c.choice[uint64_t(a) * c.total_rows + row] = uint8_t(pick);
        const uint8_t *src = c.streams + im.stream_off + uint64_t(pick) * im.stream_stride + uint64_t(y) * n;
        uint8_t *dst = c.streams + im.stream_off + uint64_t(5 + a) * im.stream_stride + uint64_t(y) * n;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

void launch_png_filter5(hipStream_t st, const FilterCtx &c) { if (c.total_rows) CSH_LAUNCH(k_png_filter5, dim3(c.total_rows), dim3(256), st, c); }
void launch_png_scores(hipStream_t st, const FilterCtx &c) {
    if (!c.total_rows) return;
    if (c.max_rowbytes + 1 > 65536u) CSH_LAUNCH_PHASED(k_png_scores<true>, 2 + 5 * 6, dim3(c.total_rows), dim3(SCORE_THREADS), st, c);
    else CSH_LAUNCH_PHASED(k_png_scores<false>, 2 + 5 * 3, dim3(c.total_rows), dim3(SCORE_THREADS), st, c);
}
void launch_png_brute(hipStream_t st, const FilterCtx &c) { if (c.total_rows) CSH_LAUNCH(k_png_brute, dim3(c.total_rows, 5), dim3(CSP_WAVE_THREADS), st, c); }
void launch_png_pick(hipStream_t st, const FilterCtx &c) { if (c.total_rows && c.plan.nadaptive) CSH_LAUNCH(k_png_pick, dim3(c.total_rows), dim3(256), st, c); }

}  // namespace csp
