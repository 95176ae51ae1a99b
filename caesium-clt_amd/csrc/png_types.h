// png_types.h -- device-visible descriptors of the lossless PNG row: what
// caesium::compress_in_memory does for a PNG when png.optimize is set (/root/reference/src/compressor.rs:305, :427-429).
// The statement of the algorithms is oracle/png_oracle.c; every kernel must equal it byte for byte.
#pragma once
#include <cstdint>

namespace csp {

enum : uint32_t {
    CSP_CHUNK = 32768,        // bytes of filtered stream per deflate block (one wave codes one chunk)
    CSP_HASH_BITS = 9,        // the greedy match finder: 512 buckets x 4 positions (4 KiB of LDS per wave).  What it is for is telling the chunks with matches in
                              // them from the ones without -- the former get the min-cost-path parse and its own, larger tables (png_parse.h); the latter lose
                              // nothing to a small table (measured on the synthetic set at 2048 / 1024 / 512 / 256 buckets), and its waves fit a CU four times over
    CSP_WAYS = 4,
    CSP_GROUP = 1,            // consecutive chunks one wave codes in a row (every chunk seeds its own match finder: carrying the
                              // table from chunk to chunk was measured at -3 % on the tokenizer passes, not worth its complexity)
    CSP_NLIT = 286, CSP_NDIST = 30, CSP_NSYM = 316, CSP_NCL = 19,
    CSP_RAW_SLACK = 4096,     // the inflate kernel flushes whole KiB and a last match may overshoot the expected size
    CSP_MAX_STREAMS = 10,     // 5 fixed filters + up to 5 adaptive strategies
};

// error codes a kernel can leave in status[image] (CCSResult.code of that file)
enum : uint32_t { CSP_OK = 0, CSP_ERR_BAD_PNG = 30100, CSP_ERR_POOL = 20200 };

struct PngImg {
    uint64_t idat_off;        // concatenated IDAT payload (the zlib stream) in the input pool
    uint32_t idat_len;
    uint32_t width, height, rowbytes, bpp;   // bpp: filter unit in bytes (1..8)
    uint64_t raw_len;         // height * (1 + rowbytes): bytes of a filtered stream
    uint64_t inflate_off;     // inflate output: the input's own filtered stream (Adam7: the seven passes back to back)
    uint64_t inflate_len;     // its length in bytes
    uint64_t raw_off;         // a second region of the image's size (plain images: the inflate output's; Adam7: the pass buffers'): the
                              // reductions repack into it
    uint64_t pix_off;         // unfiltered rows, height * rowbytes
    uint64_t stream_off;      // stream slot s of this image: stream_off + s * stream_stride
    uint64_t stream_stride;
    uint64_t match_off;       // k_png_huff's match records of this image (8-byte units, in the stream pool: the slots are free until the first trial)
    uint32_t row_base;        // first row of the image in the per-row arrays
    uint32_t nchunks;         // ceil(raw_len / CSP_CHUNK)
    uint32_t chunk_base;      // chunk record of (slot s, chunk c): chunk_base + s * chunk_stride + c
    uint32_t chunk_stride;    // records reserved per slot (nchunks of the image as it came in; a reduction only shrinks it)
    uint32_t channels, bps;   // samples per pixel, bytes per sample (0 for palette / sub-byte images: no reductions)
    uint32_t prefix_len;      // bytes in front of the IDAT chunk in the output file (signature, IHDR, carried chunks)
    uint32_t suffix_len;      // bytes after it (carried chunks, IEND)
    uint64_t fix_off;         // prefix bytes then suffix bytes in the `fixed` pool
    uint64_t out_off;         // output file region
    uint64_t out_cap;
};

// one reconstruction job of k_png_unfilter: a plain image, or one pass of an Adam7 image (PNG spec 8.2)
struct PngPass {
    uint32_t image, rowbytes, height, bpp;
    uint64_t src_off, dst_off;   // filtered rows (1 + rowbytes each) -> pixel rows (rowbytes each), both in the work buffer
};
// an Adam7 image: where its seven reconstructed passes are, for the gather that puts the pixels in place
struct PngAdam7 {
    uint32_t image, bits;        // bits per pixel
    uint64_t base[7];            // pass pixel rows in the work buffer
    uint32_t prb[7];             // their row lengths in bytes (0: the pass is empty)
};

// per (image, stream slot, chunk): what the coder knows about one deflate block
struct PngChunk {
    uint32_t freq[CSP_NSYM];      // literal/length [0,286) then distance [286,316) counts
    uint32_t extra_bits;          // sum of the extra bits of its length and distance symbols
    uint8_t len[CSP_NSYM];        // code lengths
    uint16_t code[CSP_NSYM];      // bit-reversed canonical codes
    uint8_t cl_len[CSP_NCL];
    uint16_t cl_code[CSP_NCL];
    uint16_t hlit, hdist, hclen;  // counts as the header states them (nl, nd, ncl)
    uint16_t nhdr;                // header symbols
    uint8_t hdr_sym[CSP_NSYM], hdr_extra[CSP_NSYM];
    uint64_t bits;                // size of the block in bits (header + data + end-of-block)
    uint32_t bytes;               // bytes this chunk contributes to the zlib stream (block + sync marker / final padding)
    uint32_t deep;                // 1: the chunk takes the min-cost-path parse (k_png_hist decides, png_parse.h)
};

// trial plan of a batch (the same for every image: one --png-opt-level per call)
struct PngPlan {
    int ntrials;
    int trial_slot[CSP_MAX_STREAMS];      // stream slot of trial t
    int trial_strategy[CSP_MAX_STREAMS];  // its oxipng filter number 0..9
    int nadaptive;
    int adaptive_strategy[5];             // strategies 5..9 that are needed, slot 5 + i
    int need_brute;
};

}  // namespace csp
