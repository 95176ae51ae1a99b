/*
 * jpeg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the JPEG arithmetic that sits behind the reference's
 * hot path (reference call sites: /root/reference/src/compressor.rs:287-306 ->
 * libcaesium 0.20.3 `compress_in_memory` -> mozjpeg-sys 2.2.1, pinned in
 * /root/reference/Cargo.lock:892-913 and :1035-1044).  None of that third-party
 * source is present under /root/reference, so this file restates the *published*
 * algorithms (ITU-T T.81 + the libjpeg ISLOW integer pipeline) and is pinned by
 *   (a) byte-for-byte agreement with libjpeg-turbo 3.1.4.1 (through Pillow), and
 *   (b) byte-identical entropy round trips of the reference's own fixtures
 *       samples/j0.JPG and samples/level_1_0/j1.jpg,
 * see tests/test_oracle_*.py.  The scan-script search is pinned by j0.JPG; parity with real
 * mozjpeg's trellis quantiser and deringing (cso_enc_params.trellis / .deringing) is UNPINNED
 * (DESIGN.md "Parity tiers").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in oracle/.
 */
#ifndef CSO_JPEG_ORACLE_H
#define CSO_JPEG_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CSO_MAX_COMPS 4
#define CSO_MAX_SCANS 64

typedef struct {
    int id, h, v, tq;          /* SOF: component id, sampling factors, quant table id */
    int comp_w, comp_h;        /* real downsampled size in samples */
    int real_bw, real_bh;      /* ceil(comp/8): block grid of non-interleaved scans */
    int bw, bh;                /* MCU-padded block grid (interleaved scans) */
    int16_t *coef;             /* [bh][bw][64], NATURAL order, quantised */
} cso_comp;

typedef struct {
    int ncomp_in_scan;
    int comp_idx[CSO_MAX_COMPS];
    int Ss, Se, Ah, Al;
} cso_scan;

typedef struct {
    int width, height, ncomp, precision;
    int progressive;           /* SOF2 */
    int hmax, vmax, mcus_x, mcus_y;
    int restart_interval;
    cso_comp comp[CSO_MAX_COMPS];
    uint16_t qt[4][64];        /* natural order */
    int qt_present[4];
    int nscans;                /* scan script as found in the file */
    cso_scan scans[CSO_MAX_SCANS];
    /* marker segments kept for metadata carry-over: concatenated raw segments
       (FF Mx len.. payload) of APPn/COM in file order */
    uint8_t *meta; size_t meta_len;
    int saw_jfif, adobe_transform;
} cso_image;

typedef struct {
    int quality;               /* 0..100 (libjpeg scaling) */
    int progressive;           /* 1: SOF2 + scan script; 0: sequential, one scan */
    int subsampling;           /* 444, 422, 420; 0 = auto (420 for 3-comp, none for gray) */
    int qtable_profile;        /* 3 = mozjpeg table #3 (JCP_MAX_COMPRESSION, pinned by j0.JPG);
                                  0 = Annex K (stock libjpeg) */
    int marker_style;          /* 1 = mozjpeg (merged DQT / merged DHT per scan), 0 = libjpeg */
    int scan_script;           /* 0 = stock jpeg_simple_progression; 1 = the 8-scan script found in j0.JPG;
                                  2 = mozjpeg's scan search (optimize_scans): the script is chosen per image from the sizes of its candidate scans */
    int keep_metadata;         /* copy APPn/COM (except the encoder's own JFIF) */
    int force_baseline;        /* clamp quant entries to 255 */
    int preserve_icc;          /* keep APP2 "ICC_PROFILE" segments even when keep_metadata is 0; drop them when 0
                                  (libcaesium jpeg.preserve_icc = !--strip-icc, compressor.rs:425) [UPSTREAM-RECALL] */
    /* the quantiser half of mozjpeg's JCP_MAX_COMPRESSION profile -- what `-q N` runs in libcaesium (compressor.rs:415,427 ->
       jpeg_set_defaults).  [UPSTREAM-RECALL] of mozjpeg-sys 2.2.1 (mozjpeg 4.1) jcdctmgr.c / jccoefct.c / jcmaster.c, UNPINNED: no
       golden bytes exist in the reference and the crate cannot be built here (tests/golden/make_reference_goldens.sh is the recipe). */
    int trellis;               /* 1: trellis quantisation of the AC coefficients (quantize_trellis) and, as mozjpeg's default couples it, of the DC
                                  coefficients (trellis_quant_dc); rates from the optimal Huffman table of a statistics pass over the scalar result */
    int deringing;             /* 1: overshoot deringing (preprocess_deringing) on the level-shifted samples in front of the forward DCT */
} cso_enc_params;

/* ---- decode ---- */
int  cso_decode(const uint8_t *data, size_t n, cso_image **out);   /* parse + entropy decode */
void cso_image_free(cso_image *im);
/* IDCT + upsample; out_color_space = jpeg_color_space (no colour conversion).
   out: H*W*ncomp interleaved u8. */
int  cso_decode_pixels(const cso_image *im, uint8_t *out);
/* per-component decoded plane at component resolution (IDCT + range limit, cropped
   to comp_w x comp_h).  out: comp_h*comp_w. */
int  cso_decode_plane(const cso_image *im, int ci, uint8_t *out);

/* ---- encode ---- */
/* full-resolution interleaved samples (already in the JPEG colour space) -> new
   coefficient image (edge expansion, downsample, jfdctint, plain quantiser, dummy blocks) */
int  cso_forward(const uint8_t *pix, int w, int h, int ncomp,
                 const cso_enc_params *p, const uint16_t *qt_override /* [2][64] natural or NULL */,
                 cso_image **out);
/* entropy-code a coefficient image (optimal Huffman tables) */
int  cso_encode(const cso_image *im, const cso_enc_params *p,
                const cso_scan *script, int nscans /* NULL/0 = per params */,
                uint8_t **out, size_t *out_len);
void cso_free(void *p);
/* the scan script mozjpeg's optimize_scans search picks for these coefficients ([UPSTREAM-RECALL], pinned by samples/j0.JPG) */
int  cso_search_script(const cso_image *im, const cso_enc_params *p, cso_scan *out);

/* what libcaesium's jpeg::compress_in_memory does, plain profile:
   lossless=0: decode -> pixels -> forward(quality) -> encode
   lossless=1: decode -> encode (coefficients untouched) */
int  cso_jpeg_compress(const uint8_t *in, size_t n, const cso_enc_params *p, int lossless,
                       uint8_t **out, size_t *out_len);

/* resize path (libcaesium: decode -> image-rs resize_exact(Lanczos3) -> re-encode; reference parameter mapping
   /root/reference/src/compressor.rs:503-536).  Restated pieces, all [UPSTREAM-RECALL] (image 0.25.9 imageops::sample,
   SURVEY.md B.11) -- parity with the real crate is UNPINNED:
     cso_compute_dimensions  libcaesium resize.rs compute_dimensions (f32 ratio, round half away)
     cso_lanczos3_resize     vertical_sample -> f32 image -> horizontal_sample, weights sinc(x)sinc(x/3) normalised in f32,
                             separate multiply and add (no FMA), clamp [0,255], round half away
     cso_ycc_to_rgb / cso_rgb_to_ycc   libjpeg jdcolor.c / jccolor.c 16-bit fixed point
   cso_jpeg_compress_resized = decode -> YCbCr->RGB -> Lanczos3 -> RGB->YCbCr -> forward(quality) -> encode.
   (the real chain decodes with zune-jpeg and re-encodes once more with image-rs before mozjpeg sees it) */
void cso_compute_dimensions(int ow, int oh, int dw, int dh, int *nw, int *nh);
void cso_lanczos3_resize16(const uint16_t *src, int w, int h, int nch, int nw, int nh, uint16_t *dst);
int cso_pixels_to_jpeg(const uint8_t *pix, int W, int H, int nc, const cso_enc_params *p, int width, int height, uint8_t **out, size_t *out_len);
void cso_lanczos3_resize(const uint8_t *src, int w, int h, int nch, int nw, int nh, uint8_t *dst);
void cso_ycc_to_rgb(const uint8_t *ycc, size_t npix, uint8_t *rgb);
void cso_rgb_to_ycc(const uint8_t *rgb, size_t npix, uint8_t *ycc);
int  cso_jpeg_compress_resized(const uint8_t *in, size_t n, const cso_enc_params *p, int width, int height,
                               uint8_t **out, size_t *out_len);

/* ---- pieces exported for stage-level parity tests ---- */
void cso_quality_tables(int quality, int profile, int force_baseline, uint16_t out[2][64]);
void cso_fdct_islow(const uint8_t *samples8x8 /* row stride 8 */, int32_t out[64]);
void cso_dering_block(int32_t level_shifted[64] /* natural order, in place */, int dc_quant);   /* mozjpeg preprocess_deringing [UPSTREAM-RECALL] */
void cso_trellis_tables(const cso_image *im, int ci, uint8_t aclen[256], uint8_t dclen[17]);   /* rate tables of component ci's trellis pass */
void cso_idct_islow(const int16_t coef[64], const uint16_t qt[64], uint8_t out[64]);
int  cso_stock_script(int ncomp, int which, cso_scan *out); /* returns nscans */
/* optimal Huffman table: freq[257] -> bits[17], huffval[256]; returns #symbols */
int  cso_gen_optimal_table(const long freq_in[257], uint8_t bits[17], uint8_t huffval[256]);
const char *cso_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
