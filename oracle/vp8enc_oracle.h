/* vp8enc_oracle.h -- CPU oracle of libwebp's lossy encoder at its defaults (TEST INFRASTRUCTURE ONLY; see vp8enc_oracle.c) */
#ifndef VP8ENC_ORACLE_H
#define VP8ENC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

/* per-macroblock record shared by the encoder's trace and the stream parser: what a VP8 key frame says about one macroblock */
typedef struct {
    uint8_t segment, is_i4, ymode, uvmode;   /* ymode / uvmode in libwebp's numbering: 0 DC, 1 TM, 2 V, 3 H */
    uint8_t bmodes[16];                      /* sub-block modes (0 DC 1 TM 2 VE 3 HE 4 RD 5 VR 6 LD 7 VL 8 HD 9 HU); for an i16 macroblock: ymode sixteen times */
    uint8_t skip, alpha, pad[2];             /* alpha: the analysis pass's susceptibility (encoder trace only) */
    int16_t levels[25][16];                  /* Y2, sixteen luma, four U, four V blocks; scan order */
} cso_vp8_mb;

typedef struct {
    int width, height, mbw, mbh;
    int num_segments, update_map, seg_quant[4], seg_filter[4], seg_probs[3];
    int filter_simple, filter_level, filter_sharpness;
    int num_parts_log2, base_quant, dq[5];   /* y1dc y2dc y2ac uvdc uvac */
    int use_skip, skip_proba;
    uint8_t probas[4 * 8 * 3 * 11];
    size_t part0_size, vp8_size;
    /* encoder trace only */
    int alpha_avg, uv_alpha_avg, seg_alpha[4], seg_beta[4], seg_max_edge[4];
} cso_vp8_frame;

/* quality as libwebp's WebPConfig.quality; method 4, segments 4, sns 50, filter strength 60 / sharpness 0 / strong, one pass, one
   partition, no preprocessing.  yp/up/vp: planes padded to whole macroblocks (cso_webp_rgb_to_yuv).  frame / mbs optional (mbw * mbh records). */
int cso_vp8enc_encode_yuv(const uint8_t *yp, const uint8_t *up, const uint8_t *vp, int width, int height, float quality,
                          uint8_t **out, size_t *out_len, cso_vp8_frame *frame, cso_vp8_mb *mbs);
int cso_vp8enc_encode_rgb(const uint8_t *rgb, int width, int height, float quality, uint8_t **out, size_t *out_len);
/* parses any lossy WebP / bare VP8 key frame (no reconstruction): header fields and every macroblock's modes and levels */
int cso_vp8_parse(const uint8_t *data, size_t n, cso_vp8_frame *frame, cso_vp8_mb *mbs, size_t mbs_cap);
#endif
