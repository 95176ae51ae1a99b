// png_kernels.h -- host-callable launchers of the PNG kernels (k_png_*.hip)
#pragma once
#include "gpu_rt.h"
#include "png_types.h"
#include "types.h"

namespace csp {

// P1: IDAT zlib stream -> filtered rows -> pixels (k_png_inflate.hip)
void launch_png_inflate(hipStream_t st, const PngImg *imgs, int nimg, const uint8_t *idat, uint8_t *raw, uint64_t *matches, uint32_t *nmatch, uint32_t *status);
void launch_png_unfilter(hipStream_t st, const PngPass *jobs, int njobs, uint32_t max_height, uint8_t *work, uint32_t *status);   // one job per image, seven per Adam7 image
void launch_png_deinterlace(hipStream_t st, const PngImg *imgs, const PngAdam7 *jobs, int njobs, uint64_t max_items, uint8_t *work, const uint32_t *status);

// P2: reductions (k_png_filter.hip).  flags[image] starts as the reductions the format allows (1: 16 -> 8 bits, 2: drop an
// opaque alpha channel, 4: colour -> grey; 8 / 16 / 32: the first sample of every pixel is a multiple of 17 / 85 / 255, i.e. a grey result
// fits 4 / 2 / 1 bits); the analysis clears what a pixel contradicts; the host decides and rewrites the
// descriptor; the repack moves the surviving bytes (from `src` at the old geometry to `dst` at the new one).
struct ReduceJob { uint32_t image, mask, old_rowbytes, old_channels, old_bps, gdepth, remap, pad_; uint64_t src_off, dst_off; };   // gdepth: 0, or the depth (1, 2, 4) an 8-bit grey result is packed to; | 256: an 8-bit indexed image's indices, renumbered by table `remap` (256 bytes each) and packed at that depth (1, 2, 4, 8)
void launch_png_analyze(hipStream_t st, const PngImg *imgs, uint32_t total_rows, const uint32_t *row_image, const uint8_t *pix, uint32_t *flags, const uint32_t *status);
void launch_png_repack(hipStream_t st, const PngImg *imgs, const ReduceJob *jobs, int njobs, uint32_t max_height, const uint8_t *src, uint8_t *dst, const uint8_t *remaps);
// 8-bit indexed images (flag bit 64): used[image * 8 + w] |= the palette entries their pixels point at
void launch_png_used(hipStream_t st, const PngImg *imgs, uint32_t total_rows, const uint32_t *row_image, const uint8_t *pix, const uint32_t *flags, uint32_t *used, const uint32_t *status);

// colour -> palette (oracle: to_palette).  cand[image] = channels (3 / 4) of an 8- or 16-bit truecolour image that may become indexed,
// else 0.  The distinct pixels (alpha, red, green, blue of the high bytes in one 32-bit key) are collected in a 1024-slot open
// hash table per image; counts above 256 mean "not a palette image".  The host sorts the palette and sends back, per slot, the
// index of its colour; the conversion packs the indices (depth 1, 2, 4 or 8).
enum { CSP_PAL_SLOTS = 1024 };
struct PaletteJob {
    uint32_t image, old_rowbytes, old_channels, old_bps, depth, table;   // table: the image's slot in the hash tables (exact) ...
    uint32_t nearest, npal, pal_off;                                      // ... or, lossy (nearest 2): palette[pal_off, pal_off + npal), Floyd-Steinberg error diffusion (k_png_dither)
    uint32_t line_off, pad_;                                              // k_png_dither: the job's two hand-down lines start at this pixel of the line buffer
    uint64_t src_off, dst_off;
};
void launch_png_colors(hipStream_t st, const PngImg *imgs, uint32_t total_rows, const uint32_t *row_image, const uint8_t *pix, const uint32_t *cand, unsigned long long *keys,
                       uint32_t *counts, const uint32_t *status);
void launch_png_indexed(hipStream_t st, const PngImg *imgs, const PaletteJob *jobs, int njobs, uint32_t max_height, const unsigned long long *keys, const uint16_t *slot_index,
                        const uint32_t *palettes, const uint8_t *src, uint8_t *dst);
enum { CSP_DITHER_ROWS = 512 };   // rows of a picture that advance together in k_png_dither (two waves per SIMD: 256 rows took 12 160 steps for a 1080p picture at the same time per step, 1024 rows 7 936 at twice the time)
void launch_png_dither(hipStream_t st, const PngImg *imgs, const PaletteJob *jobs, int njobs, int nsteps, const uint32_t *palettes, const uint8_t *src, uint8_t *dst, int16_t *lines);

// resize of a PNG source (k_png_resize.hip): interleaved samples of bps bytes (1, or 2 big-endian), nc per pixel; src / dst are byte
// offsets, tmp a float offset
struct PngResize { uint32_t width, height, nc, nw, nh, vtap_base, htap_base, bps; uint64_t src_off, tmp_off, dst_off; };
// hjobs: the host's copy of jobs.  Fused (every source row fits LDS): one kernel, tmp is not touched and need not exist
bool png_resize_is_fused(const PngResize *hjobs, int njobs);
void launch_png_resize(hipStream_t st, const PngResize *jobs, const PngResize *hjobs, int njobs, const csh::ResizeTap *taps, const float *weights, const uint8_t *src, float *tmp, uint8_t *dst,
                       uint64_t max_tmp, uint64_t max_dst);

// the decoded pixels of an 8-bit-or-less PNG format (16-bit: narrowed) as interleaved 8-bit samples.  out_nc 1 / 3: grey or RGB for
// k_webp_yuv (opaque sources).  out_nc 2 / 4 (or 1 / 3 with no tRNS): what the png crate's EXPAND transformation gives image-rs before
// a resize -- palette entries looked up, sub-byte grey scaled, tRNS turned into an alpha channel (trns: per-index alpha for a palette,
// else the transparent sample value(s) as 16-bit big-endian numbers)
// wide != 0: a 16-bit grey / RGB image with a tRNS chunk -- its samples stay 16-bit (big-endian, as in the file) and the colour key becomes a 16-bit alpha sample
struct RgbJob { uint32_t image, width, height, rowbytes, ctype, depth, plte_off, npal, out_nc, trns_off, ntrns, wide; uint64_t src_off, dst_off; };
void launch_png_rgb(hipStream_t st, const RgbJob *jobs, int njobs, uint32_t max_height, const uint8_t *plte, const uint8_t *work, uint8_t *rgb, const uint32_t *status);

// lossy PNG (oracle: quantize): colour bins of 4 + 5 + 5 + 5 bits (a, r, g, b) with count and channel sums, compacted to a list
// k_png_mediancut works on
enum { CSP_QBINS = 1 << 19 };
struct QBin { uint32_t id, cnt, s[4]; };
struct QuantJob { uint32_t image, channels, bps, rowbytes, width, height; uint64_t src_off, bins_off, list_off; };   // bins_off: in uint32 units (5 per bin); list_off: in QBin units
void launch_png_qhist(hipStream_t st, const QuantJob *jobs, int njobs, uint32_t max_height, const uint8_t *work, uint32_t *bins);
void launch_png_qcompact(hipStream_t st, const QuantJob *jobs, int njobs, const uint32_t *bins, QBin *list, uint32_t *nlist);
// the median cut of every job's bin list: pal[job][256] (unsorted, equal entries possible), npal[job]; scratch: recs (one 16-byte entry per list
// element), order (two arrays of one uint32 per list element, order_stride apart)
void launch_png_mediancut(hipStream_t st, const QuantJob *jobs, int njobs, const QBin *list, const uint32_t *nlist, uint4 *recs, uint32_t *order, uint64_t order_stride, int quality,
                          unsigned long long bound, uint32_t *pal, uint32_t *npal);

// P3: row-filter search (k_png_filter.hip)
struct FilterCtx {
    const PngImg *imgs;
    int nimg;
    uint32_t total_rows;          // rows of the whole batch
    uint32_t max_rowbytes;        // the longest row of the batch (k_png_scores: 16-bit pair counters while a row has fewer than 65536 pairs)
    const uint32_t *row_image;    // [total_rows] image of a batch row
    const uint8_t *pix;
    uint8_t *streams;
    uint64_t *scores;             // [total_rows][5 filters][5 scores]: MinSum, Entropy, Bigrams, BigEnt, Brute
    uint8_t *choice;              // [5 strategies][total_rows]
    PngPlan plan;
    const uint32_t *status;
};
void launch_png_filter5(hipStream_t st, const FilterCtx &c);   // the five fixed-filter streams (slots 0..4)
void launch_png_scores(hipStream_t st, const FilterCtx &c);    // scores 0..3 of every (row, filter)
void launch_png_brute(hipStream_t st, const FilterCtx &c);     // score 4
void launch_png_pick(hipStream_t st, const FilterCtx &c);      // per strategy and row: best filter; gathers the adaptive streams (slots 5..)

// P4: deflate (k_png_deflate.hip)
struct DeflateCtx {
    const PngImg *imgs;
    int nimg;
    uint32_t total_chunks;        // chunks of one stream slot over the whole batch
    const uint32_t *chunk_image;  // [total_chunks] image of a batch chunk; its index inside the image = chunk - imgs[image].chunk0
    const uint32_t *chunk_first;  // [nimg] first batch chunk of the image
    uint32_t total_groups;        // the tokenizer passes take CSP_GROUP consecutive chunks per wave
    const uint32_t *group_image;  // [total_groups]
    const uint32_t *group_first;  // [nimg] first batch group of the image
    const uint8_t *streams;
    PngChunk *chunks;             // [(image chunk_base) + slot * nchunks + c]
    PngPlan plan;
    uint64_t *trial_bytes;        // [nimg][CSP_MAX_STREAMS] zlib stream size per trial
    int32_t *winner;              // [nimg] winning trial
    uint8_t *trial_live;          // [nimg][CSP_MAX_STREAMS] 1: the trial's greedy stream is close enough to the smallest for the min-cost-path parse to matter
    uint32_t *adler_parts;        // [total_chunks][2]
    uint8_t *out;
    const uint8_t *fixed;         // prefix/suffix bytes
    uint32_t *file_len;           // [nimg]
    uint32_t *crc_parts;          // per KiB piece of the IDAT chunk
    uint32_t *status;
    uint8_t *deep_scratch;        // [deep_slots][CSP_DEEP_SCRATCH]: one area per workgroup of the min-cost-path kernels (png_parse.h)
    uint32_t deep_slots;
    uint32_t *deep_queue;         // [4] work counters of the two kernels, the length of deep_list (zeroed by launch_png_deep)
    uint32_t *deep_list;          // [total_chunks * ntrials] the (trial, chunk) items whose counts the parse replaced
    int deep_iters;               // passes of that parse over the chunks that qualify: CSP_DEEP_ITERS, CSP_DEEP_ITERS_ZOPFLI with png.force_zopfli
};
void launch_png_hist(hipStream_t st, const DeflateCtx &c);     // tokenizer pass 1: symbol counts of every (trial, chunk)
void launch_png_codes(hipStream_t st, const DeflateCtx &c, int only_deep = 0);    // code lengths, codes, header, block size
void launch_png_choose(hipStream_t st, const DeflateCtx &c);   // per image: stream sizes, the winner, chunk byte offsets
void launch_png_emit(hipStream_t st, const DeflateCtx &c);     // tokenizer pass 2 on the winner: the blocks, in place
void launch_png_deep(hipStream_t st, const DeflateCtx &c);        // behind choose: the marked chunks of the live trials through the min-cost-path parse, codes and choose again
void launch_png_deep_emit(hipStream_t st, const DeflateCtx &c);   // (inside launch_png_emit) the winner's marked chunks
void launch_png_finish(hipStream_t st, const DeflateCtx &c, uint32_t max_pieces);   // max_pieces: KiB pieces of the largest IDAT chunk; crc_parts holds nimg * max_pieces   // zlib header + Adler-32, IDAT framing + CRC-32, carried chunks

}  // namespace csp
