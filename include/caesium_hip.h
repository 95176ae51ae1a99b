/*
 * caesium_hip.h -- C ABI of libcaesium_hip.so, the MI355X-native replacement for the three
 * libcaesium entry points caesium-clt calls on its per-image hot path.
 *
 * Reference interface being replaced (Rust API, called from a rayon par_iter):
 *   caesium::compress_in_memory          /root/reference/src/compressor.rs:305
 *   caesium::compress_to_size_in_memory  /root/reference/src/compressor.rs:295,298
 *   caesium::convert_in_memory           /root/reference/src/compressor.rs:289,300
 *   parameter mapping (CSParameters)     /root/reference/src/compressor.rs:411-446, 503-536
 *   SupportedFileTypes order             /root/reference/src/compressor.rs:589-598
 *   the par_iter that becomes the batch  /root/reference/src/compressor.rs:74-101
 * The struct layout mirrors libcaesium's own C interface (CCSParameters / CCSResult / CByteArray,
 * SURVEY.md 2b) so a cgo/FFI binding written for libcaesium maps 1:1; INTEGRATION.md shows the
 * Rust-side `extern "C"` block a maintainer would add.
 *
 * Conventions: nothing throws across this ABI; every call is thread-safe; inputs are borrowed for
 * the duration of the call; outputs are callee-allocated and released with cs_free_bytes /
 * cs_free_result.  The compute stages run on the GPU only: without a usable gfx950 device every
 * compute entry point fails with code CS_ERR_NO_DEVICE -- there is no CPU fallback.
 */
#ifndef CAESIUM_HIP_H
#define CAESIUM_HIP_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* caesium::SupportedFileTypes, in the order map_supported_formats() names them */
enum { CS_TYPE_JPEG = 0, CS_TYPE_PNG = 1, CS_TYPE_GIF = 2, CS_TYPE_WEBP = 3, CS_TYPE_TIFF = 4, CS_TYPE_UNKN = 5 };

/* error codes (CCSResult.code); 0 = success */
enum {
    CS_OK = 0,
    CS_ERR_NO_DEVICE = 10001,      /* no gfx950 device / HIP runtime failure */
    CS_ERR_UNKNOWN_TYPE = 10200,   /* magic bytes not recognised */
    CS_ERR_UNSUPPORTED = 10201,    /* recognised, but this build has no device path for it yet */
    CS_ERR_BAD_JPEG = 20100,       /* malformed / truncated JPEG */
    CS_ERR_JPEG_FEATURE = 20101,   /* arithmetic coding, 12-bit, CMYK, unsupported sampling ... */
    CS_ERR_POOL_OVERFLOW = 20200,  /* internal device pool too small even after retry */
    CS_ERR_BAD_PNG = 30100,        /* malformed / truncated PNG (chunk walk, zlib stream, filter bytes) */
    CS_ERR_BAD_WEBP = 40100,       /* malformed / truncated WebP (RIFF chunk walk, VP8 frame header, bool-coded data running out) */
    CS_ERR_SAME_FORMAT = 10407,    /* convert_in_memory: source type == target type */
    CS_ERR_TOO_BIG = 10500         /* compress_to_size: cannot reach max_output_size */
};

typedef struct {
    bool keep_metadata;
    uint32_t jpeg_quality;
    uint32_t jpeg_chroma_subsampling; /* 444, 422, 420, 411, 0 = Auto */
    bool jpeg_progressive;
    bool jpeg_optimize;               /* true = lossless coefficient transcode (--lossless) */
    bool jpeg_preserve_icc;
    uint32_t png_quality;
    uint32_t png_optimization_level;
    bool png_force_zopfli;            /* --zopfli (with png_optimize): fifteen passes of the DEFLATE coder's cost model instead of five; refused until round 6 */
    bool png_optimize;
    uint32_t gif_quality;
    uint32_t webp_quality;
    bool webp_lossless;
    uint32_t tiff_compression;
    uint32_t tiff_deflate_level;
    uint32_t width;
    uint32_t height;
} CCSParameters;

typedef struct {
    bool success;
    uint32_t code;
    const char *error_message; /* callee-owned; release with cs_free_result */
} CCSResult;

typedef struct {
    uint8_t *data; /* callee-allocated; release with cs_free_bytes */
    size_t length;
} CByteArray;

/* CSParameters::new() defaults (quality 80, progressive, Auto subsampling, png level 3 ...) */
void cs_default_parameters(CCSParameters *p);

/* replaces caesium::compress_in_memory (compressor.rs:305) */
CCSResult cs_compress_in_memory(const uint8_t *in, size_t n, const CCSParameters *p, CByteArray *out);
/* replaces caesium::compress_to_size_in_memory (compressor.rs:295,298); *p is in-out (quality fields
   are overwritten while bisecting, as the reference's &mut CSParameters is) */
CCSResult cs_compress_to_size_in_memory(const uint8_t *in, size_t n, CCSParameters *p, size_t max_output_size,
                                        bool return_smallest, CByteArray *out);
/* replaces caesium::convert_in_memory (compressor.rs:289,300).  Built: JPEG -> WebP and PNG -> WebP (lossy, or lossless with webp_lossless; a PNG's transparency is kept: an ALPH chunk / ARGB), JPEG -> PNG (png_optimize: lossless trials, else the quantiser), PNG -> JPEG (alpha dropped), WebP -> JPEG / PNG; every other pair answers CS_ERR_UNSUPPORTED (or CS_ERR_SAME_FORMAT) */
CCSResult cs_convert_in_memory(const uint8_t *in, size_t n, const CCSParameters *p, uint32_t format, CByteArray *out);
/* the batch form of it: results[i] / outputs[i] correspond to inputs[i] */
int cs_batch_convert(const CByteArray *inputs, size_t count, const CCSParameters *p, uint32_t format, int device, CByteArray *outputs, CCSResult *results);

/* start_compression's par_iter (compressor.rs:74-101) turned into a device batch queue:
   results[i] / outputs[i] correspond to inputs[i] (order preserved, as compressor.rs:789-792 asserts).
   device = HIP device ordinal.  Returns the number of failed items (per-item status in results). */
int cs_batch_compress(const CByteArray *inputs, size_t count, const CCSParameters *p, int device,
                      CByteArray *outputs, CCSResult *results);
/* the same for --max-size (compressor.rs:297-299): every file runs libcaesium's quality bisection; the batch is decoded
   and transformed once, each round only re-quantises and re-codes the files still searching */
int cs_batch_compress_to_size(const CByteArray *inputs, size_t count, CCSParameters *p, size_t max_output_size, bool return_smallest,
                              int device, CByteArray *outputs, CCSResult *results);

/* how many of the leading `count` inputs one device batch takes: at most 2048 files, 2 GiB of input bytes (scan offsets are 32-bit)
   and ~96 GiB of estimated device pools (from each file's declared size: a header probe, no decode).  Always >= 1 when count >= 1.
   cs_batch_* use it themselves; it is exported for callers that drive csh_batch_create directly (the CLI, bench.py) */
size_t cs_batch_extent(const CByteArray *inputs, size_t count);

void cs_free_bytes(CByteArray *b);
void cs_free_result(CCSResult *r);

/* ------------------------------------------------------------------------------------------------
 * Device-resident batch interface: what bench.py times (inputs already in HBM) and what the stage
 * parity tests poke at.  create = parse headers + upload entropy segments; run = the whole hot path
 * on the device (entropy decode, pixel-domain transcode, entropy encode, byte assembly); fetch =
 * copy the finished files back.
 */
typedef struct csh_batch csh_batch;

enum { CSH_NPHASES = 8, CSH_NKERNELS = 36 };
typedef struct {
    float total_ms;               /* hipEvent time around the whole run, on the batch's stream */
    float phase_ms[CSH_NPHASES];  /* 0 decode, 1 pixel transcode, 2 masks+flags+runs, 3 stats+tables,
                                     4 size+scan, 5 pack, 6 stuff+assemble, 7 reserved */
    float kernel_ms[CSH_NKERNELS];/* hipEvent time of every launch (names: csh_kernel_name) */
    uint64_t in_bytes, out_bytes; /* entropy-coded bytes in, file bytes out */
    uint64_t pixels;              /* source pixels processed */
    uint64_t coef_bytes;          /* bytes of coefficient planes (one direction) */
    uint32_t n_images, n_failed;
    uint32_t n_seq_decoded;       /* images (re)done by the sequential decode kernel: progressive / DRI inputs, or
                                     parallel-decoder fallbacks */
    uint32_t n_par_fallback;      /* of those, images the parallel decoder started and gave up on */
    uint32_t n_par_short;         /* of those, because the scan produced fewer blocks than the frame needs (truncated data) */
    uint32_t n_prog_decoded;      /* progressive inputs decoded by the wave-per-chain kernel (not counted in n_seq_decoded) */
    uint32_t n_refine_chains;     /* chains of AC refinement scans (progressive inputs) taken by the parse + apply kernels (k_decode_refine.hip) */
    uint32_t n_search_extra;      /* conditional stages of the scan search this run needed (0..3: luma at Al 3, the fourth and the fifth frequency split) */
} csh_timing;

int csh_device_count(void);
const char *csh_last_error(void);
void csh_warmup(int device);            /* optional: start the runtime, the device context and the code objects now (call it from a thread of its own while the files are being read) */
void csh_release_cached_memory(void);   /* hand the cached device pools and pinned blocks of finished batches back to the driver (they are kept for the next batch otherwise) */
const char *csh_kernel_name(int slot);
const char *csh_kernel_name_webp(int slot);   /* the same for a csh_batch_create_webp batch: behind the resize slot come the VP8 tail's kernels */
int csh_batch_create(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csh_batch **out);
/* the same batch with WebP as the target container (convert_in_memory to WebP, compressor.rs:289,300): decode, optional
   resize, then the VP8 encoder; run / fetch / destroy as for any csh_batch */
int csh_batch_create_webp(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csh_batch **out);
/* the same decode and resize with nothing behind them (the front half of convert_in_memory to PNG): after csh_batch_run the image
   of input i is in device memory -- interleaved 8-bit samples, 1 (grey) or 3 (RGB) per pixel, rows back to back -- until the batch is
   destroyed.  csh_batch_pixels returns the CCSResult code of that file (0: pointers set); csh_batch_fetch does not apply */
int csh_batch_create_pixels(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csh_batch **out);
int csh_batch_pixels(csh_batch *b, size_t image, const uint8_t **device_pixels, uint32_t *width, uint32_t *height, uint32_t *channels, const char **message);
/* pixels in, JPEG out (the back half of convert_in_memory to JPEG): 8-bit grey or RGB in device memory (csp_pixels below; copied at
   creation), then the resize and the encoder of the JPEG path with p's quality / progressive / chroma parameters.  A side of more than
   65535 pixels or another channel count is answered per file */
struct csp_pixels_s;
int csh_batch_create_from_pixels(const struct csp_pixels_s *sources, size_t count, const CCSParameters *p, int device, csh_batch **out);
int csh_batch_run(csh_batch *b, csh_timing *t);
int csh_batch_fetch(csh_batch *b, CByteArray *outputs, CCSResult *results);
void csh_batch_destroy(csh_batch *b);

/* size targeting on the device (what compress_to_size_in_memory's bisection needs per try): retain the unquantised DCT
   of a full run, re-target the quality of individual images (quality[i] == 0 keeps it; same indexing as the inputs), and
   re-run only re-quantisation + entropy coding + assembly.  Results are identical to a full run at that quality. */
int csh_batch_retain_dct(csh_batch *b, int on);
int csh_batch_set_quality(csh_batch *b, const uint32_t *quality);
int csh_batch_rerun_encode(csh_batch *b, csh_timing *t);

/* stage taps for the parity tests (copy device intermediates to host after csh_batch_run):
   which = 0: decoded coefficients, 1: re-quantised coefficients.  dst receives the component's
   [bh][bw][64] int16 blocks in ZIG-ZAG order.  Returns 0, or -1 (see csh_last_error). */
int csh_batch_geometry(csh_batch *b, size_t image, int comp, int which, int *bw, int *bh, int *real_bw, int *real_bh);
int csh_batch_read_coefs(csh_batch *b, size_t image, int comp, int which, int16_t *dst);

/* ------------------------------------------------------------------------------------------------
 * Lossless PNG row (png.optimize = true, the `--lossless` flag: compressor.rs:427-429; level = --png-opt-level,
 * src/options.rs:63-65): the device-resident batch interface of the PNG pipeline.  create = chunk walk + upload of the IDAT
 * streams; run = inflate, unfilter, row-filter search, deflate trials, assembly, all on the device; fetch = copy the
 * finished files back (a file that did not get smaller comes back unchanged, as oxipng does).  cs_batch_compress routes
 * PNG inputs here.
 */
typedef struct csp_batch csp_batch;
enum { CSP_NKERNELS = 16 };
typedef struct {
    float total_ms;
    float kernel_ms[CSP_NKERNELS];   /* names: csp_kernel_name */
    uint64_t in_bytes, out_bytes;    /* IDAT bytes in, file bytes out (device result, before the "not smaller" rule) */
    uint64_t pixels, raw_bytes;      /* pixels; bytes of one filtered stream per image, summed */
    uint32_t n_images, n_failed, n_trials;
} csp_timing;
const char *csp_kernel_name(int slot);
int csp_batch_create(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csp_batch **out);
/* the same decode stages feeding the VP8 encoder (caesium::convert_in_memory with a PNG source and SupportedFileTypes::WebP,
   /root/reference/src/compressor.rs:289-299): opaque PNGs only; p->webp_quality applies.  run / fetch / destroy as above;
   the stage taps of the PNG coder do not exist for such a batch */
int csp_batch_create_webp(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csp_batch **out);
/* the PNG coder over pixels that are already in device memory (caesium::convert_in_memory to PNG: the decoded source goes through
   png::compress, /root/reference/src/compressor.rs:289-305): 8-bit samples, rows of width * channels bytes back to back, channels 1
   (grey), 2 (grey + alpha), 3 (RGB) or 4 (RGBA).  The pixels are copied at creation (device to device; the source must be complete:
   synchronise the stream that produced it first).  p selects the lossless (png_optimize) or the quantising form, as for a PNG file */
typedef struct csp_pixels_s { const uint8_t *device_pixels; uint32_t width, height, channels; } csp_pixels;
int csp_batch_create_pixels(const csp_pixels *sources, size_t count, const CCSParameters *p, int device, csp_batch **out);
/* PNG in, JPEG out in one call (what cs_batch_convert does for such files; convert_in_memory to JPEG, compressor.rs:289-299): the PNG
   decode stages, the pixels as 8-bit grey / RGB (alpha dropped), then the JPEG resize + encoder with p's parameters.  Returns the
   number of failed files; outputs / results in input order */
int csp_png_to_jpeg(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results);
/* PNG -> lossless WebP (`--format webp --lossless` on a PNG; compressor.rs:289-305 with webp.lossless): the same decode, the VP8L coder behind it;
   an alpha channel or a tRNS chunk stays (grey + alpha / RGBA pixels into the coder); such a picture with p->width / height answers CS_ERR_UNSUPPORTED */
int csp_png_to_lossless_webp(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, CByteArray *outputs, CCSResult *results);
int csp_batch_run(csp_batch *b, csp_timing *t);
int csp_batch_fetch(csp_batch *b, CByteArray *outputs, CCSResult *results);
void csp_batch_destroy(csp_batch *b);
/* stage taps for the parity tests (after csp_batch_run): the unfiltered rows (height * rowbytes), the filtered stream of
   one oxipng strategy 0..9 (height * (1 + rowbytes); only strategies the level needs exist), and per trial the size of
   its zlib stream with the winner */
int csp_batch_geometry(csp_batch *b, size_t image, uint32_t *width, uint32_t *height, uint32_t *rowbytes);
int csp_batch_read_rows(csp_batch *b, size_t image, uint8_t *dst);
int csp_batch_read_stream(csp_batch *b, size_t image, int strategy, uint8_t *dst);
/* scores of every (row, filter) candidate: dst[height][5 filters][5: MinSum, Entropy, Bigrams, BigEnt, Brute]; *have = bit k
   set when score k was computed for this level's plan */
int csp_batch_read_scores(csp_batch *b, size_t image, uint64_t *dst, int *have);
int csp_batch_trials(csp_batch *b, size_t image, int *strategies, uint64_t *zlib_bytes, int *ntrials, int *winner);
/* size in bits of every deflate block of one trial (dst[*nchunks]; capacity `cap` entries) */
int csp_batch_chunk_bits(csp_batch *b, size_t image, int trial, uint64_t *dst, size_t cap, size_t *nchunks);

/* ------------------------------------------------------------------------------------------------
 * WebP INPUTS (libcaesium decodes them with libwebp before webp::compress / convert_in_memory, compressor.rs:289-305): the RIFF
 * container is walked on the host, the VP8 key frame is decoded on the device (k_webp_dec.hip), the RGB stays in HBM
 * (cswd_batch_pixels: the csp_pixels the encoders take).  Built: lossy (VP8) and lossless (VP8L) still pictures, transparency (ALPH chunk, non-opaque
 * VP8L) included -- cswd_batch_alpha; animation answers CS_ERR_UNSUPPORTED per file.  cs_batch_compress and cs_batch_convert route WebP files here themselves.
 */
/* lossless WebP OUTPUT (webp.lossless; libcaesium: webp::compress with libwebp's lossless coder, compressor.rs:427-429, 289-305):
 * 8-bit grey (channels 1), grey + alpha (2), RGB (3) or RGBA (4) pictures that are in device memory -> one VP8L file each (k_vp8l_enc.hip).
 * outputs / results as cs_batch_compress; returns the number of failed items.  cs_batch_compress (WebP sources) and cs_batch_convert (JPEG and PNG
 * sources) call it when p->webp_lossless is set; the lossy PNG -> WebP path calls it for the ALPH chunk of a transparent picture (channels 16 + n: the
 * last of n samples per pixel coded as a grey picture). */
int csl_encode_pixels(const struct csp_pixels_s *sources, size_t count, int device, CByteArray *outputs, CCSResult *results);
/* a lossy WebP file + its alpha plane coded by csl_encode_pixels (as a grey picture) -> the extended-format file with an ALPH chunk (VP8X, ALPH, VP8);
   lossy is replaced in place.  0 ok. */
int csl_attach_alpha(CByteArray *lossy, const CByteArray *alpha_vp8l, uint32_t width, uint32_t height);

typedef struct cswd_batch cswd_batch;
int cswd_batch_create(const CByteArray *inputs, size_t count, int device, cswd_batch **out);
int cswd_batch_run(cswd_batch *b);
int cswd_batch_pixels(cswd_batch *b, size_t image, const uint8_t **device_pixels, uint32_t *width, uint32_t *height, uint32_t *channels, const char **message);
int cswd_batch_read_pixels(cswd_batch *b, size_t image, uint8_t *dst /* width * height * 3 */);
/* a picture with transparency (ALPH chunk / a VP8L picture that is not opaque): its RGBA (4 bytes per pixel) and its alpha plane in device memory; both
   NULL for an opaque picture.  cswd_batch_pixels keeps handing out the RGB of either kind. */
int cswd_batch_alpha(cswd_batch *b, size_t image, const uint8_t **device_rgba, const uint8_t **device_alpha);
int cswd_batch_read_rgba(cswd_batch *b, size_t image, uint8_t *dst /* width * height * 4; returns 1 for an opaque picture */);
void cswd_batch_destroy(cswd_batch *b);
/* the two halves of a picture with transparency after their resize (RGB, 3 bytes per pixel, and the alpha plane, both in device memory) joined into one
   RGBA picture in device memory of its own: what the PNG coder and the lossless WebP coder take (image-rs resamples the four channels alike,
   /root/reference/src/compressor.rs:289-300 through libcaesium's resize).  0 ok. */
typedef struct cswd_rgba cswd_rgba;
int cswd_rgba_join(const uint8_t *device_rgb, const uint8_t *device_alpha, uint32_t width, uint32_t height, int device, cswd_rgba **out, const uint8_t **device_rgba);
void cswd_rgba_destroy(cswd_rgba *r);
/* pixels in, WebP out: the lossy WebP encoder behind the same stand-in as csh_batch_create_from_pixels */
int csh_batch_create_webp_from_pixels(const struct csp_pixels_s *sources, size_t count, const CCSParameters *p, int device, csh_batch **out);
/* pixels in, resized pixels out (csh_batch_pixels): the Lanczos branch alone */
int csh_batch_create_from_pixels_rgb(const struct csp_pixels_s *sources, size_t count, const CCSParameters *p, int device, csh_batch **out);

#ifdef __cplusplus
}
#endif
#endif
