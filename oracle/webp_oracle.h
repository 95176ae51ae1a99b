/* webp_oracle.h -- CPU oracle of the lossy WebP row (TEST INFRASTRUCTURE ONLY; see webp_oracle.c and vp8enc_oracle.c) */
#ifndef WEBP_ORACLE_H
#define WEBP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
void cso_webp_rgb_to_yuv(const uint8_t *rgb, int w, int h, uint8_t *yp, uint8_t *up, uint8_t *vp);   /* planes padded to whole macroblocks */
int cso_webp_encode_rgb(const uint8_t *rgb, int width, int height, int quality, uint8_t **out, size_t *out_len);   /* = cso_vp8enc_encode_rgb: libwebp's encoder restated */
#endif
