/* webp_oracle.h -- CPU oracle of the lossy WebP row (TEST INFRASTRUCTURE ONLY; see webp_oracle.c) */
#ifndef WEBP_ORACLE_H
#define WEBP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
void cso_webp_rgb_to_yuv(const uint8_t *rgb, int w, int h, uint8_t *yp, uint8_t *up, uint8_t *vp);   /* planes padded to whole macroblocks */
int cso_webp_quality_to_qi(int quality);
int cso_webp_encode_yuv(const uint8_t *yp, const uint8_t *up, const uint8_t *vp, int width, int height, int qi, uint8_t **out, size_t *out_len,
                        uint8_t *ry, uint8_t *ru, uint8_t *rv);   /* ry/ru/rv: optional, the encoder's own reconstruction (padded planes) */
int cso_webp_bmode_cost(int m, int top, int left, int from_table);   /* sub-block mode cost in 1/256 bit: by the formula, or from kVp8BModeCost */
int cso_webp_encode_rgb(const uint8_t *rgb, int width, int height, int quality, uint8_t **out, size_t *out_len);
#endif
