/* png_oracle.h -- CPU oracle of the lossless PNG row (TEST INFRASTRUCTURE ONLY; see png_oracle.c) */
#ifndef PNG_ORACLE_H
#define PNG_ORACLE_H
#include <stddef.h>
#include <stdint.h>
enum { CSO_PNG_BAD = 30100, CSO_PNG_UNSUPPORTED = 10201 };
typedef struct {
    uint32_t width, height;
    int depth, ctype, interlace, channels, bpp, nplte;
    size_t rowbytes;
    uint8_t *pix;            /* height * rowbytes, unfiltered */
    uint8_t *chunks;         /* the chunks that are carried over (whole: length, type, data, crc), in file order */
    size_t chunks_len, idat_at; /* idat_at: offset in `chunks` where the first IDAT stood */
    int no_reduce;           /* a carried chunk (tRNS, bKGD, sBIT) is tied to the colour type: the image keeps its format */
    int pal_tied;            /* a carried chunk (bKGD, sBIT, hIST) counts on the palette as it is: an indexed image keeps its depth */
} cso_png;
uint32_t cso_crc32(uint32_t crc, const uint8_t *p, size_t n);
uint32_t cso_adler32(const uint8_t *p, size_t n);
int cso_inflate_zlib(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *produced);
int cso_png_decode(const uint8_t *in, size_t n, int keep_metadata, cso_png **out);
void cso_png_free(cso_png *p);
int cso_png_scores(const cso_png *P, uint64_t *out);
int cso_png_reduce(cso_png *P);   /* P2: returns a bit mask of what was applied (1: 16->8 bits, 2: alpha dropped, 4: colour->grey, 8: colour->palette, 32: grey depth, 64: index depth) */
int cso_png_quantize(cso_png *P, int quality);   /* lossy: truecolour with more than 256 colours -> indexed (median cut); returns 16 when applied */
int cso_png_lossy(const uint8_t *in, size_t n, int level, int keep_metadata, int quality, uint8_t **out, size_t *out_len);
int cso_png_to_rgb(const cso_png *P, uint8_t *rgb);   /* width * height * 3 bytes; CSO_PNG_UNSUPPORTED for an image with transparency */
int cso_png_to_webp(const uint8_t *in, size_t n, int quality, uint8_t **out, size_t *out_len);
int cso_png_filter(const cso_png *P, int strategy, uint8_t *out, uint8_t *choice);
int cso_deflate_zlib(const uint8_t *data, size_t n, uint8_t **out, size_t *out_len);
int cso_deflate_zlib_iters(const uint8_t *data, size_t n, int iters, uint8_t **out, size_t *out_len);
int cso_png_optimize_zopfli(const uint8_t *in, size_t n, int level, int keep_metadata, uint8_t **out, size_t *out_len, int *chosen);
int cso_png_optimize_iters(const uint8_t *in, size_t n, int level, int iters, uint8_t **out, size_t *out_len);
void cso_png_deep_div(int v);
int cso_png_trials(int level, int *set);
int cso_png_optimize(const uint8_t *in, size_t n, int level, int keep_metadata, uint8_t **out, size_t *out_len, int *chosen);
/* lossless WebP output (VP8L) as the device coder writes it (k_vp8l_enc.hip): px = width * height * channels bytes (1 grey, 2 grey + alpha, 3 RGB, 4 RGBA) */
int cso_vp8l_encode(const uint8_t *px, int width, int height, int channels, uint8_t **out, size_t *out_len);
#endif
