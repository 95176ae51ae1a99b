/*
 * vp8enc_oracle.c -- CPU ORACLE of SURVEY.md 8a row W2/W3: libwebp's lossy (VP8) encoder as the reference reaches it.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, tools/, bench.py's cpu_baseline leg): nothing under caesium-clt_amd/ links or calls this.
 *
 * The reference's JPEG/PNG -> WebP conversions and its WebP recompression end in libwebp's WebPEncode with a default WebPConfig at the
 * requested quality (/root/reference/src/compressor.rs:289, :300, :417, :429 -> crate webp 0.3.1 -> libwebp-sys 0.9.5, Cargo.lock:956,
 * :1814): method 4, 4 segments, sns_strength 50, filter_strength 60 (strong filter, sharpness 0), 1 pass, 1 token partition.  libwebp's
 * source is not under /root/reference; this file restates the published algorithm of its src/enc (analysis_enc.c, quant_enc.c,
 * frame_enc.c, cost_enc.c, token_enc.c, syntax_enc.c, iterator_enc.c, tree_enc.c, dsp/enc.c, utils/bit_writer_utils.c) in this repo's
 * own frame-level form (phases over whole planes instead of a macroblock iterator), which is also the form the device takes.
 *
 * PARITY IS PINNED: libwebp itself is executable in this container in three versions (1.2.0, 1.2.2, 1.6.0: they agree byte for byte at
 * these settings, so does whatever 0.9.5 vendors in between), and tests/test_oracle_vp8enc.py requires this file's bytes to equal
 * WebPEncode's on synthetic and fixture pictures (live when a libwebp is present, and against committed digests otherwise).
 *
 * Order of the computation (what depends on what; the device keeps it):
 *   A  analysis: per macroblock, from SOURCE samples only: alpha = how peaked the histogram of its transformed DC / TM residual is
 *   B  frame set-up: alpha histogram -> 4-means -> segment map; per segment: quantiser index (SNS), matrices, lambdas, filter strength
 *   C  the macroblock loop in raster order: i16 (4 modes) vs i4 (16 x 10 modes) vs chroma (4 modes) by rate-distortion score with
 *      libwebp's cost tables, reconstruction, token statistics.  The level-cost tables are rebuilt from the statistics so far
 *      every max(96, mbs / 8) (+1) macroblocks: the only coupling besides the spatial one (left / above / above-right).
 *   D  final coefficient probabilities; partition 0 (headers + modes); the token partition; RIFF.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vp8_cost_tables.h"
#include "../include/vp8_tables.h"
#include "vp8enc_oracle.h"
#include "webp_oracle.h"

typedef int64_t score_t;
#define MAX_SCORE ((score_t)0x7fffffffffffffLL)
enum { NTYPES = 4, NBANDS = 8, NCTX = 3, NPROBAS = 11, NSLOTS = NTYPES * NBANDS * NCTX * NPROBAS, MAX_VAR_LEVEL = 67, MAX_LEVEL = 2047 };
enum { QFIX = 17, FLAT_I16 = 0, FLAT_I4 = 3, FLAT_UV = 2, FLAT_PENALTY = 140, RD_MULT = 256 };

static int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static int clipi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static int bit_cost(int bit, int p) { return kVp8EntropyCost[bit ? 255 - p : p]; }

/* ------------------------------------------------------------------------------------------------ transforms (dsp/enc.c) */
static void fdct4(const uint8_t *src, int ss, const uint8_t *ref, int rs, int16_t *out) {
    int t[16];
    for (int i = 0; i < 4; i++, src += ss, ref += rs) {
        const int d0 = src[0] - ref[0], d1 = src[1] - ref[1], d2 = src[2] - ref[2], d3 = src[3] - ref[3];
        const int a0 = d0 + d3, a1 = d1 + d2, a2 = d1 - d2, a3 = d0 - d3;
        t[0 + i * 4] = (a0 + a1) * 8;
        t[1 + i * 4] = (a2 * 2217 + a3 * 5352 + 1812) >> 9;
        t[2 + i * 4] = (a0 - a1) * 8;
        t[3 + i * 4] = (a3 * 2217 - a2 * 5352 + 937) >> 9;
    }
    for (int i = 0; i < 4; i++) {
        const int a0 = t[0 + i] + t[12 + i], a1 = t[4 + i] + t[8 + i], a2 = t[4 + i] - t[8 + i], a3 = t[0 + i] - t[12 + i];
        out[0 + i] = (int16_t)((a0 + a1 + 7) >> 4);
        out[4 + i] = (int16_t)(((a2 * 2217 + a3 * 5352 + 12000) >> 16) + (a3 != 0));
        out[8 + i] = (int16_t)((a0 - a1 + 7) >> 4);
        out[12 + i] = (int16_t)((a3 * 2217 - a2 * 5352 + 51000) >> 16);
    }
}
static void fwht(const int16_t *dc, int16_t *out) {   /* dc: the sixteen block DCs in block raster order */
    int t[16];
    for (int i = 0; i < 4; i++) {
        const int a0 = dc[i * 4 + 0] + dc[i * 4 + 2], a1 = dc[i * 4 + 1] + dc[i * 4 + 3], a2 = dc[i * 4 + 1] - dc[i * 4 + 3], a3 = dc[i * 4 + 0] - dc[i * 4 + 2];
        t[0 + i * 4] = a0 + a1; t[1 + i * 4] = a3 + a2; t[2 + i * 4] = a3 - a2; t[3 + i * 4] = a0 - a1;
    }
    for (int i = 0; i < 4; i++) {
        const int a0 = t[0 + i] + t[8 + i], a1 = t[4 + i] + t[12 + i], a2 = t[4 + i] - t[12 + i], a3 = t[0 + i] - t[8 + i];
        out[0 + i] = (int16_t)((a0 + a1) >> 1); out[4 + i] = (int16_t)((a3 + a2) >> 1); out[8 + i] = (int16_t)((a3 - a2) >> 1); out[12 + i] = (int16_t)((a0 - a1) >> 1);
    }
}
static void iwht(const int16_t *in, int16_t *dc) {
    int t[16];
    for (int i = 0; i < 4; i++) {
        const int a0 = in[0 + i] + in[12 + i], a1 = in[4 + i] + in[8 + i], a2 = in[4 + i] - in[8 + i], a3 = in[0 + i] - in[12 + i];
        t[0 + i] = a0 + a1; t[8 + i] = a0 - a1; t[4 + i] = a3 + a2; t[12 + i] = a3 - a2;
    }
    for (int i = 0; i < 4; i++) {
        const int d = t[0 + i * 4] + 3, a0 = d + t[3 + i * 4], a1 = t[1 + i * 4] + t[2 + i * 4], a2 = t[1 + i * 4] - t[2 + i * 4], a3 = d - t[3 + i * 4];
        dc[i * 4 + 0] = (int16_t)((a0 + a1) >> 3); dc[i * 4 + 1] = (int16_t)((a3 + a2) >> 3); dc[i * 4 + 2] = (int16_t)((a0 - a1) >> 3); dc[i * 4 + 3] = (int16_t)((a3 - a2) >> 3);
    }
}
#define M1(a) ((((a) * 20091) >> 16) + (a))
#define M2(a) (((a) * 35468) >> 16)
static void idct4_add(const int16_t *in, const uint8_t *ref, int rs, uint8_t *dst, int ds) {
    int t[16];
    for (int i = 0; i < 4; i++) {
        const int a = in[0 + i] + in[8 + i], b = in[0 + i] - in[8 + i], c = M2(in[4 + i]) - M1(in[12 + i]), d = M1(in[4 + i]) + M2(in[12 + i]);
        t[0 + i * 4] = a + d; t[1 + i * 4] = b + c; t[2 + i * 4] = b - c; t[3 + i * 4] = a - d;
    }
    for (int i = 0; i < 4; i++) {
        const int dc = t[0 + i] + 4, a = dc + t[8 + i], b = dc - t[8 + i], c = M2(t[4 + i]) - M1(t[12 + i]), d = M1(t[4 + i]) + M2(t[12 + i]);
        dst[i * ds + 0] = (uint8_t)clip8(ref[i * rs + 0] + ((a + d) >> 3));
        dst[i * ds + 1] = (uint8_t)clip8(ref[i * rs + 1] + ((b + c) >> 3));
        dst[i * ds + 2] = (uint8_t)clip8(ref[i * rs + 2] + ((b - c) >> 3));
        dst[i * ds + 3] = (uint8_t)clip8(ref[i * rs + 3] + ((a - d) >> 3));
    }
}
/* distortions: squared error, and the "texture" term: difference of the two blocks' weighted Hadamard spectra */
static int sse(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h) {
    int s = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const int d = a[y * as + x] - b[y * bs + x]; s += d * d; }
    return s;
}
static int hadamard_w(const uint8_t *in, int s) {
    int t[16], sum = 0;
    for (int i = 0; i < 4; i++, in += s) {
        const int a0 = in[0] + in[2], a1 = in[1] + in[3], a2 = in[1] - in[3], a3 = in[0] - in[2];
        t[0 + i * 4] = a0 + a1; t[1 + i * 4] = a3 + a2; t[2 + i * 4] = a3 - a2; t[3 + i * 4] = a0 - a1;
    }
    for (int i = 0; i < 4; i++) {
        const int a0 = t[0 + i] + t[8 + i], a1 = t[4 + i] + t[12 + i], a2 = t[4 + i] - t[12 + i], a3 = t[0 + i] - t[8 + i];
        sum += kVp8WeightY[0 + i] * abs(a0 + a1) + kVp8WeightY[4 + i] * abs(a3 + a2) + kVp8WeightY[8 + i] * abs(a3 - a2) + kVp8WeightY[12 + i] * abs(a0 - a1);
    }
    return sum;
}
static int tdisto4(const uint8_t *a, int as, const uint8_t *b, int bs) { return abs(hadamard_w(b, bs) - hadamard_w(a, as)) >> 5; }
static int tdisto16(const uint8_t *a, int as, const uint8_t *b, int bs) {
    int d = 0;
    for (int y = 0; y < 16; y += 4) for (int x = 0; x < 16; x += 4) d += tdisto4(a + y * as + x, as, b + y * bs + x, bs);
    return d;
}

/* ------------------------------------------------------------------------------------------------ intra predictors (encoder's edge rules) */
/* N x N (16 luma, 8 chroma) from the samples around position r of plane stride rs; has_left = mx > 0, has_top = my > 0.  Without
   a neighbour: V predicts 127, H 129, TM degenerates to the other one (129 with neither), DC doubles the side it has (128 with neither). */
static void predN(int mode, const uint8_t *r, int rs, int N, int has_left, int has_top, uint8_t *out) {
    if (mode == 0) {
        int dc = 0;
        const int sh = N == 16 ? 5 : 4, rnd = N;
        if (has_top) { for (int i = 0; i < N; i++) dc += r[i - rs]; if (has_left) for (int i = 0; i < N; i++) dc += r[i * rs - 1]; else dc += dc; dc = (dc + rnd) >> sh; }
        else if (has_left) { for (int i = 0; i < N; i++) dc += r[i * rs - 1]; dc += dc; dc = (dc + rnd) >> sh; }
        else dc = 128;
        memset(out, dc, (size_t)N * N);
        return;
    }
    if (mode == 1 && has_left && has_top) {
        const int tl = r[-rs - 1];
        for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) out[y * N + x] = (uint8_t)clip8(r[x - rs] + r[y * rs - 1] - tl);
        return;
    }
    if (mode == 1) mode = has_left ? 3 : has_top ? 2 : 4;   /* 4: nothing around: 129 */
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) out[y * N + x] = (uint8_t)(mode == 2 ? (has_top ? r[x - rs] : 127) : mode == 3 ? (has_left ? r[y * rs - 1] : 129) : 129);
}
/* 4x4 from the thirteen edge samples e[] = L K J I X A B C D E F G H (vp8_tables.h) */
static void pred4(int mode, const uint8_t *e, uint8_t *out) {
    if (mode == 0) { memset(out, (e[5] + e[6] + e[7] + e[8] + e[3] + e[2] + e[1] + e[0] + 4) >> 3, 16); return; }
    for (int k = 0; k < 16; k++) {
        if (mode == 1) out[k] = (uint8_t)clip8(e[3 - (k >> 2)] + e[5 + (k & 3)] - e[4]);
        else { const unsigned t = kVp8Pred4Taps[(mode - 2) * 16 + k]; out[k] = (uint8_t)((e[t & 15] + e[(t >> 4) & 15] + e[(t >> 8) & 15] + e[t >> 12] + 2) >> 2); }
    }
}

/* ------------------------------------------------------------------------------------------------ boolean coder (bit_writer_utils.c) */
typedef struct { uint8_t *buf; size_t pos, cap; int32_t range, value; int run, nb_bits; } boolenc;
static void be_init(boolenc *e) { memset(e, 0, sizeof *e); e->range = 255 - 1; e->nb_bits = -8; }
static void be_room(boolenc *e, size_t n) { if (e->pos + n > e->cap) { e->cap = (e->pos + n) * 2 + 256; e->buf = (uint8_t *)realloc(e->buf, e->cap); } }
static void be_out(boolenc *e) {
    const int s = 8 + e->nb_bits;
    const int32_t bits = e->value >> s;
    e->value -= bits << s;
    e->nb_bits -= 8;
    if ((bits & 0xff) != 0xff) {
        be_room(e, (size_t)e->run + 1);
        if ((bits & 0x100) && e->pos > 0) e->buf[e->pos - 1]++;
        if (e->run > 0) { const uint8_t v = (bits & 0x100) ? 0x00 : 0xff; for (; e->run > 0; --e->run) e->buf[e->pos++] = v; }
        e->buf[e->pos++] = (uint8_t)(bits & 0xff);
    } else
        e->run++;
}
static int be_put(boolenc *e, int bit, int prob) {
    const int32_t split = (e->range * prob) >> 8;
    if (bit) { e->value += split + 1; e->range -= split + 1; } else e->range = split;
    if (e->range < 127) {
        const int shift = __builtin_clz((unsigned)(e->range + 1)) - 24;
        e->range = ((e->range + 1) << shift) - 1;
        e->value <<= shift;
        e->nb_bits += shift;
        if (e->nb_bits > 0) be_out(e);
    }
    return bit;
}
static void be_bits(boolenc *e, uint32_t v, int n) { while (n--) be_put(e, (v >> n) & 1, 128); }
static void be_sbits(boolenc *e, int v, int n) { if (!be_put(e, v != 0, 128)) return; if (v < 0) be_bits(e, ((uint32_t)(-v) << 1) | 1, n + 1); else be_bits(e, (uint32_t)v << 1, n + 1); }
static void be_finish(boolenc *e) { be_bits(e, 0, 9 - e->nb_bits); e->nb_bits = 0; be_out(e); }

/* ------------------------------------------------------------------------------------------------ quantiser set-up (quant_enc.c) */
typedef struct { int q[2], iq[2], bias[2], zthresh[2], sharpen[16]; } qmat;   /* [0] DC, [1] AC; sharpen in raster order */
typedef struct {
    qmat y1, y2, uv;
    int quant, fstrength, alpha, beta, max_edge, min_disto;
    int lambda_i16, lambda_i4, lambda_uv, lambda_mode, tlambda;
} segment;
static int expand(qmat *m, int type) {
    for (int i = 0; i < 2; i++) {
        m->iq[i] = (1 << QFIX) / m->q[i];
        m->bias[i] = kVp8BiasMatrices[type * 2 + i] << (QFIX - 8);
        m->zthresh[i] = ((1 << QFIX) - 1 - m->bias[i]) / m->iq[i];
    }
    for (int j = 0; j < 16; j++) m->sharpen[j] = type == 0 ? (kVp8FreqSharpening[j] * m->q[j > 0]) >> 11 : 0;
    return (m->q[0] + 15 * m->q[1] + 8) >> 4;
}
static int lam(int v) { return v < 1 ? 1 : v; }
static void setup_matrices(segment *s, int dq_uv_dc, int dq_uv_ac, int sns) {
    const int q = s->quant;
    s->y1.q[0] = kVp8DcQ[clipi(q, 0, 127)];
    s->y1.q[1] = kVp8AcQ[clipi(q, 0, 127)];
    s->y2.q[0] = kVp8DcQ[clipi(q, 0, 127)] * 2;
    s->y2.q[1] = kVp8AcTable2[clipi(q, 0, 127)];
    s->uv.q[0] = kVp8DcQ[clipi(q + dq_uv_dc, 0, 117)];
    s->uv.q[1] = kVp8AcQ[clipi(q + dq_uv_ac, 0, 127)];
    const int q4 = expand(&s->y1, 0), q16 = expand(&s->y2, 1), quv = expand(&s->uv, 2);
    s->lambda_i4 = lam((3 * q4 * q4) >> 7);
    s->lambda_i16 = lam(3 * q16 * q16);
    s->lambda_uv = lam((3 * quv * quv) >> 6);
    s->lambda_mode = lam((1 * q4 * q4) >> 7);
    s->tlambda = (sns * q4) >> 5;   /* method >= 4: the texture-distortion weight scales with the SNS strength */
    s->min_disto = 20 * s->y1.q[0];
    s->max_edge = 0;
}
/* one coefficient block: in[] raster (overwritten with the dequantised values), out[] scan order; returns "has a non-zero level" */
static int quantize_block(int16_t *in, int16_t *out, const qmat *m) {
    int last = -1;
    for (int n = 0; n < 16; n++) {
        const int j = kVp8Zigzag[n], k = j > 0, sign = in[j] < 0;
        const uint32_t coeff = (uint32_t)((sign ? -in[j] : in[j]) + m->sharpen[j]);
        if (coeff > (uint32_t)m->zthresh[k]) {
            int level = (int)((coeff * (uint32_t)m->iq[k] + (uint32_t)m->bias[k]) >> QFIX);
            if (level > MAX_LEVEL) level = MAX_LEVEL;
            if (sign) level = -level;
            in[j] = (int16_t)(level * m->q[k]);
            out[n] = (int16_t)level;
            if (level) last = n;
        } else { out[n] = 0; in[j] = 0; }
    }
    return last >= 0;
}
/* chroma DC with error diffusion: quantise one value, return what was lost (halved for storage) */
static int quantize_single(int16_t *v, const qmat *m) {
    int V = *v;
    const int sign = V < 0;
    if (sign) V = -V;
    if (V > m->zthresh[0]) {
        const int qV = (int)(((uint32_t)V * (uint32_t)m->iq[0] + (uint32_t)m->bias[0]) >> QFIX) * m->q[0], err = V - qV;
        *v = (int16_t)(sign ? -qV : qV);
        return (sign ? -err : err) >> 1;
    }
    *v = 0;
    return (sign ? -V : V) >> 1;
}

/* ------------------------------------------------------------------------------------------------ costs (cost_enc.c) */
typedef struct {
    uint8_t coeffs[NSLOTS];                      /* current probabilities [type][band][ctx][11] */
    uint32_t stats[NSLOTS];                      /* hi 16: events, lo 16: ones -- with libwebp's halving on overflow */
    uint16_t level_cost[NTYPES][NBANDS][NCTX][MAX_VAR_LEVEL + 1];
    int dirty;
} probas;
static int slot(int t, int b, int c) { return ((t * NBANDS + b) * NCTX + c) * NPROBAS; }
static void level_costs(probas *P) {
    if (!P->dirty) return;
    for (int t = 0; t < NTYPES; t++)
        for (int b = 0; b < NBANDS; b++)
            for (int c = 0; c < NCTX; c++) {
                const uint8_t *p = P->coeffs + slot(t, b, c);
                uint16_t *tab = P->level_cost[t][b][c];
                const int cost0 = c > 0 ? bit_cost(1, p[0]) : 0, base = bit_cost(1, p[1]) + cost0;
                tab[0] = (uint16_t)(bit_cost(0, p[1]) + cost0);
                for (int v = 1; v <= MAX_VAR_LEVEL; v++) {
                    int pattern = kVp8LevelCodes[(v - 1) * 2], bits = kVp8LevelCodes[(v - 1) * 2 + 1], cost = 0;
                    for (int i = 2; pattern; i++, bits >>= 1, pattern >>= 1) if (pattern & 1) cost += bit_cost(bits & 1, p[i]);
                    tab[v] = (uint16_t)(base + cost);
                }
            }
    P->dirty = 0;
}
static int level_cost1(const uint16_t *tab, int v) { return kVp8LevelFixedCosts[v] + tab[v > MAX_VAR_LEVEL ? MAX_VAR_LEVEL : v]; }
/* bits (1/256) of one block's levels given the context ctx0 of its first coefficient */
static int residual_cost(const probas *P, int type, int first, int ctx0, const int16_t *lv) {
    int last = -1, n = first;
    for (int i = 15; i >= 0; i--) if (lv[i]) { last = i; break; }   /* all sixteen positions are looked at, whatever `first` */
    const int p0 = P->coeffs[slot(type, kVp8Bands[n], ctx0)];
    const uint16_t *t = P->level_cost[type][kVp8Bands[n]][ctx0];
    int cost = ctx0 == 0 ? bit_cost(1, p0) : 0;
    if (last < 0) return bit_cost(0, p0);
    for (; n < last; n++) {
        const int v = abs(lv[n]), ctx = v >= 2 ? 2 : v;
        cost += level_cost1(t, v);
        t = P->level_cost[type][kVp8Bands[n + 1]][ctx];
    }
    {
        const int v = abs(lv[n]);
        cost += level_cost1(t, v);
        if (n < 15) cost += bit_cost(0, P->coeffs[slot(type, kVp8Bands[n + 1], v == 1 ? 1 : 2)]);
    }
    return cost;
}
static void record(uint32_t *s, int bit) {
    uint32_t p = *s;
    if (p >= 0xfffe0000u) p = ((p + 1u) >> 1) & 0x7fff7fffu;
    *s = p + 0x00010000u + (uint32_t)bit;
}
/* the frame's probabilities from the statistics so far (FinalizeTokenProbas) */
static void finalize_probas(probas *P) {
    int changed = 0;
    for (int i = 0; i < NSLOTS; i++) {
        const int nb = (int)(P->stats[i] & 0xffff), total = (int)(P->stats[i] >> 16), up = kVp8CoefUpdateProbs[i], oldp = kVp8CoefProbs[i];
        const int newp = nb ? 255 - nb * 255 / total : 255;
        const int old_cost = nb * bit_cost(1, oldp) + (total - nb) * bit_cost(0, oldp) + bit_cost(0, up);
        const int new_cost = nb * bit_cost(1, newp) + (total - nb) * bit_cost(0, newp) + bit_cost(1, up) + 8 * 256;
        if (old_cost > new_cost) { P->coeffs[i] = (uint8_t)newp; changed |= newp != oldp; } else P->coeffs[i] = (uint8_t)oldp;
    }
    P->dirty = changed;
}

/* ------------------------------------------------------------------------------------------------ tokens (token_enc.c) */
typedef struct { uint16_t *t; size_t n, cap; } tokbuf;   /* bit 15: the bit; bit 14: fixed probability in the low byte; else the slot index */
static void tk(tokbuf *b, uint16_t v) { if (b->n == b->cap) { b->cap = b->cap * 2 + 4096; b->t = (uint16_t *)realloc(b->t, b->cap * 2); } b->t[b->n++] = v; }
static int tok(tokbuf *b, probas *P, int bit, int idx) { tk(b, (uint16_t)((bit << 15) | idx)); record(&P->stats[idx], bit); return bit; }
/* libwebp counts the second category bit of the two big categories (probability 10) in the statistics of probability 9; the token itself carries 10 */
static void tok10(tokbuf *b, probas *P, int bit, int idx) { tk(b, (uint16_t)((bit << 15) | idx)); record(&P->stats[idx - 1], bit); }
static void tokc(tokbuf *b, int bit, int prob) { tk(b, (uint16_t)((bit << 15) | (1 << 14) | prob)); }
static int record_block(tokbuf *b, probas *P, int type, int first, int ctx, const int16_t *lv) {
    int last = -1, n = first;
    for (int i = 15; i >= 0; i--) if (lv[i]) { last = i; break; }
    int s = slot(type, kVp8Bands[n], ctx);
    if (!tok(b, P, last >= 0, s + 0)) return 0;
    while (n < 16) {
        const int c = lv[n++], sign = c < 0;
        const unsigned v = (unsigned)(sign ? -c : c);
        if (!tok(b, P, v != 0, s + 1)) { s = slot(type, kVp8Bands[n], 0); continue; }
        if (!tok(b, P, v > 1, s + 2)) s = slot(type, kVp8Bands[n], 1);
        else {
            if (!tok(b, P, v > 4, s + 3)) { if (tok(b, P, v != 2, s + 4)) tok(b, P, v == 4, s + 5); }
            else if (!tok(b, P, v > 10, s + 6)) {
                if (!tok(b, P, v > 6, s + 7)) tokc(b, v == 6, 159);
                else { tokc(b, v >= 9, 165); tokc(b, !(v & 1), 145); }
            } else {
                unsigned residue = v - 3, mask;
                const uint8_t *tab;
                if (residue < (8 << 1)) { tok(b, P, 0, s + 8); tok(b, P, 0, s + 9); residue -= 8 << 0; mask = 1 << 2; tab = kVp8Cat3; }
                else if (residue < (8 << 2)) { tok(b, P, 0, s + 8); tok(b, P, 1, s + 9); residue -= 8 << 1; mask = 1 << 3; tab = kVp8Cat4; }
                else if (residue < (8 << 3)) { tok(b, P, 1, s + 8); tok10(b, P, 0, s + 10); residue -= 8 << 2; mask = 1 << 4; tab = kVp8Cat5; }
                else { tok(b, P, 1, s + 8); tok10(b, P, 1, s + 10); residue -= 8 << 3; mask = 1 << 10; tab = kVp8Cat6; }
                for (; mask; mask >>= 1) tokc(b, !!(residue & mask), *tab++);
            }
            s = slot(type, kVp8Bands[n], 2);
        }
        tokc(b, sign, 128);
        if (n == 16 || !tok(b, P, n <= last, s + 0)) return 1;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------ the encoder */
typedef struct {
    int mbw, mbh, ys, cs;
    const uint8_t *sy, *su, *sv;   /* source planes, padded */
    uint8_t *ry, *ru, *rv;         /* reconstruction (unfiltered: what intra prediction sees) */
    segment seg[4];
    probas P;
    uint32_t *nz;                  /* per macroblock: bits 0-15 luma blocks, 16-19 U, 20-23 V, 24 Y2 (an i4 macroblock passes the one above it on) */
    uint8_t *bm;                   /* sub-block mode context: (4 mbw) x (4 mbh), an i16 macroblock counts as its mode */
    int8_t (*top_derr)[2][2];      /* chroma DC error diffusion, per column [U / V][2] */
} enc_t;

/* A: susceptibility of one macroblock, from source samples only (neighbours included) */
static int alpha_of(const int *hist) {
    int maxv = 0, last = 1;
    for (int k = 0; k <= 31; k++) if (hist[k] > 0) { if (hist[k] > maxv) maxv = hist[k]; last = k; }
    return maxv > 1 ? 510 * last / maxv : 0;
}
static void hist_blocks(const uint8_t *src, int ss, const uint8_t *pred, int N, int *hist) {
    for (int by = 0; by < N; by += 4)
        for (int bx = 0; bx < N; bx += 4) {
            int16_t c[16];
            fdct4(src + by * ss + bx, ss, pred + by * N + bx, N, c);
            for (int k = 0; k < 16; k++) { const int v = abs(c[k]) >> 3; hist[v > 31 ? 31 : v]++; }
        }
}
static void analyse_mb(const enc_t *E, int mx, int my, int *alpha_out, int *uv_alpha_out) {
    const uint8_t *sy = E->sy + (size_t)my * 16 * E->ys + mx * 16, *su = E->su + (size_t)my * 8 * E->cs + mx * 8, *sv = E->sv + (size_t)my * 8 * E->cs + mx * 8;
    int best = -1, best_uv = -1;
    for (int mode = 0; mode < 2; mode++) {   /* DC and TM only */
        uint8_t p[256];
        int hist[32] = {0};
        predN(mode, sy, E->ys, 16, mx > 0, my > 0, p);
        hist_blocks(sy, E->ys, p, 16, hist);
        const int a = alpha_of(hist);
        if (a > best) best = a;
    }
    for (int mode = 0; mode < 2; mode++) {
        uint8_t p[64];
        int hist[32] = {0};
        predN(mode, su, E->cs, 8, mx > 0, my > 0, p);
        hist_blocks(su, E->cs, p, 8, hist);
        predN(mode, sv, E->cs, 8, mx > 0, my > 0, p);
        hist_blocks(sv, E->cs, p, 8, hist);
        const int a = alpha_of(hist);
        if (a > best_uv) best_uv = a;
    }
    *alpha_out = clipi(255 - ((3 * best + best_uv + 2) >> 2), 0, 255);
    *uv_alpha_out = best_uv;
}

/* B: 4-means over the alpha histogram (analysis_enc.c AssignSegments), then each segment's alpha / beta */
static void assign_segments(const int *hist, int nb, int *map, int *centers, int *seg_alpha, int *seg_beta) {
    int min_a, max_a, n, wavg = 0;
    for (n = 0; n <= 255 && hist[n] == 0; n++) {}
    min_a = n;
    for (n = 255; n > min_a && hist[n] == 0; n--) {}
    max_a = n;
    const int range = max_a - min_a;
    for (int k = 0, m = 1; k < nb; k++, m += 2) centers[k] = min_a + (m * range) / (2 * nb);
    for (int it = 0; it < 6; it++) {
        int accum[4] = {0}, dist[4] = {0}, displaced = 0, total = 0;
        n = 0;
        for (int a = min_a; a <= max_a; a++)
            if (hist[a]) {
                while (n + 1 < nb && abs(a - centers[n + 1]) < abs(a - centers[n])) n++;
                map[a] = n;
                dist[n] += a * hist[a];
                accum[n] += hist[a];
            }
        wavg = 0;
        for (n = 0; n < nb; n++)
            if (accum[n]) {
                const int c = (dist[n] + accum[n] / 2) / accum[n];
                displaced += abs(centers[n] - c);
                centers[n] = c;
                wavg += c * accum[n];
                total += accum[n];
            }
        wavg = (wavg + total / 2) / total;
        if (displaced < 5) break;
    }
    int mn = centers[0], mx = centers[0];
    if (nb > 1) for (n = 0; n < nb; n++) { if (mn > centers[n]) mn = centers[n]; if (mx < centers[n]) mx = centers[n]; }
    if (mx == mn) mx = mn + 1;
    for (n = 0; n < nb; n++) {
        seg_alpha[n] = clipi(255 * (centers[n] - wavg) / (mx - mn), -127, 127);
        seg_beta[n] = clipi(255 * (centers[n] - mn) / (mx - mn), 0, 255);
    }
}

/* per macroblock result of the mode decision */
typedef struct { score_t D, SD, R, H, score; uint32_t nz; } rd_t;
static void rd_score(rd_t *r, int lambda) { r->score = (r->R + r->H) * lambda + (score_t)RD_MULT * (r->D + r->SD); }
static void rd_add(rd_t *d, const rd_t *s) { d->D += s->D; d->SD += s->SD; d->R += s->R; d->H += s->H; d->nz |= s->nz; d->score += s->score; }
static int flat_levels(const int16_t *lv, int nblocks, int thresh) {
    int score = 0;
    for (; nblocks-- > 0; lv += 16) for (int i = 1; i < 16; i++) { score += lv[i] != 0; if (score > thresh) return 0; }
    return 1;
}

/* the non-zero context bits of the macroblocks above and to the left, unpacked (index 0-3 luma, 4-5 U, 6-7 V, 8 Y2) */
static void nz_import(const enc_t *E, int mx, int my, int left_y2, int *top, int *left) {
    const uint32_t t = my > 0 ? E->nz[(size_t)(my - 1) * E->mbw + mx] : 0, l = mx > 0 ? E->nz[(size_t)my * E->mbw + mx - 1] : 0;
    top[0] = (t >> 12) & 1; top[1] = (t >> 13) & 1; top[2] = (t >> 14) & 1; top[3] = (t >> 15) & 1;
    top[4] = (t >> 18) & 1; top[5] = (t >> 19) & 1; top[6] = (t >> 22) & 1; top[7] = (t >> 23) & 1; top[8] = (t >> 24) & 1;
    left[0] = (l >> 3) & 1; left[1] = (l >> 7) & 1; left[2] = (l >> 11) & 1; left[3] = (l >> 15) & 1;
    left[4] = (l >> 17) & 1; left[5] = (l >> 19) & 1; left[6] = (l >> 21) & 1; left[7] = (l >> 23) & 1; left[8] = left_y2;
}

int cso_vp8enc_encode_yuv(const uint8_t *yp, const uint8_t *up, const uint8_t *vp, int width, int height, float quality,
                          uint8_t **out, size_t *out_len, cso_vp8_frame *frame, cso_vp8_mb *mbs_out) {
    if (width < 1 || height < 1 || width > 16383 || height > 16383) return -1;
    const int SNS = 50, FSTRENGTH = 60, NSEG = 4;
    const int diffuse = quality <= 98.f;   /* chroma DC error diffusion: not at the top qualities */
    enc_t E;
    memset(&E, 0, sizeof E);
    const int mbw = E.mbw = (width + 15) >> 4, mbh = E.mbh = (height + 15) >> 4, ys = E.ys = mbw * 16, cs = E.cs = mbw * 8, nmb = mbw * mbh;
    E.sy = yp; E.su = up; E.sv = vp;
    E.ry = (uint8_t *)calloc((size_t)ys * mbh * 16, 1); E.ru = (uint8_t *)calloc((size_t)cs * mbh * 8, 1); E.rv = (uint8_t *)calloc((size_t)cs * mbh * 8, 1);
    E.nz = (uint32_t *)calloc((size_t)nmb, 4);
    E.bm = (uint8_t *)calloc((size_t)nmb * 16, 1);
    E.top_derr = (int8_t(*)[2][2])calloc((size_t)mbw, sizeof *E.top_derr);
    cso_vp8_mb *mbs = mbs_out ? mbs_out : (cso_vp8_mb *)calloc((size_t)nmb, sizeof *mbs);
    memset(mbs, 0, (size_t)nmb * sizeof *mbs);
    if (quality < 0.f) quality = 0.f;
    if (quality > 100.f) quality = 100.f;

    /* ---- A: analysis */
    int hist[256] = {0}, alpha_sum = 0, uv_alpha_sum = 0;
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            int a, ua;
            analyse_mb(&E, mx, my, &a, &ua);
            mbs[my * mbw + mx].alpha = (uint8_t)a;
            hist[a]++;
            alpha_sum += a;
            uv_alpha_sum += ua;
        }
    const int uv_alpha = uv_alpha_sum / nmb;

    /* ---- B: segments, quantisers, filter strengths */
    int amap[256], centers[4], nseg = NSEG, remap[4] = {0, 1, 2, 3};
    memset(amap, 0, sizeof amap);
    {
        int sa[4], sb[4];
        assign_segments(hist, nseg, amap, centers, sa, sb);
        for (int i = 0; i < nseg; i++) { E.seg[i].alpha = sa[i]; E.seg[i].beta = sb[i]; }
        for (int i = 0; i < nmb; i++) mbs[i].segment = (uint8_t)amap[mbs[i].alpha];
    }
    const double amp = 0.9 * SNS / 100. / 128., Q = quality / 100.;
    const double lin = Q < 0.75 ? Q * (2. / 3.) : 2. * Q - 1., c_base = pow(lin, 1 / 3.);
    for (int i = 0; i < nseg; i++) {
        const double expn = 1. - amp * E.seg[i].alpha, c = pow(c_base, expn);
        E.seg[i].quant = clipi((int)(127. * (1. - c)), 0, 127);
    }
    const int base_quant = E.seg[0].quant;
    const int dq_uv_ac = clipi((uv_alpha - 64) * (6 - -4) / (100 - 30) * SNS / 100, -4, 6), dq_uv_dc = clipi(-4 * SNS / 100, -15, 15);
    for (int i = 0; i < 4; i++) {   /* loop-filter strength from the AC step and the segment's complexity */
        segment *s = &E.seg[i];
        const int qstep = kVp8AcQ[clipi(s->quant, 0, 127)] >> 2, base = kVp8LevelsFromDelta[qstep > 63 ? 63 : qstep], f = base * (5 * FSTRENGTH) / (256 + s->beta);
        s->fstrength = f < 2 ? 0 : f > 63 ? 63 : f;
    }
    int filter_level = E.seg[0].fstrength;
    if (nseg > 1) {   /* merge segments that ended up alike (quantiser and filter strength) */
        int nfinal = 1;
        for (int s1 = 1; s1 < nseg; s1++) {
            int s2, found = 0;
            for (s2 = 0; s2 < nfinal; s2++) if (E.seg[s1].quant == E.seg[s2].quant && E.seg[s1].fstrength == E.seg[s2].fstrength) { found = 1; break; }
            remap[s1] = s2;
            if (!found) { if (nfinal != s1) E.seg[nfinal] = E.seg[s1]; nfinal++; }
        }
        if (nfinal < nseg) {
            for (int i = 0; i < nmb; i++) mbs[i].segment = (uint8_t)remap[mbs[i].segment];
            for (int i = nfinal; i < nseg; i++) E.seg[i] = E.seg[nfinal - 1];
            nseg = nfinal;
        }
    }
    for (int i = 0; i < nseg; i++) setup_matrices(&E.seg[i], dq_uv_dc, dq_uv_ac, SNS);
    /* segment-id probabilities */
    int seg_probs[3] = {255, 255, 255}, update_map = 0;
    {
        int cnt[4] = {0};
        for (int i = 0; i < nmb; i++) cnt[mbs[i].segment]++;
        if (nseg > 1) {
#define GETP(a, b) (((a) + (b)) == 0 ? 255 : (255 * (a) + ((a) + (b)) / 2) / ((a) + (b)))
            seg_probs[0] = GETP(cnt[0] + cnt[1], cnt[2] + cnt[3]); seg_probs[1] = GETP(cnt[0], cnt[1]); seg_probs[2] = GETP(cnt[2], cnt[3]);
            update_map = seg_probs[0] != 255 || seg_probs[1] != 255 || seg_probs[2] != 255;
            if (!update_map) for (int i = 0; i < nmb; i++) mbs[i].segment = 0;
        }
    }

    /* ---- C: the macroblock loop */
    probas *P = &E.P;
    memcpy(P->coeffs, kVp8CoefProbs, NSLOTS);
    P->dirty = 1;
    level_costs(P);
    tokbuf T = {NULL, 0, 0};
    int max_count = nmb >> 3, cnt;
    if (max_count < 96) max_count = 96;
    cnt = max_count;
    for (int my = 0; my < mbh; my++) {
        int left_y2 = 0;
        int8_t left_derr[2][2] = {{0, 0}, {0, 0}};
        for (int mx = 0; mx < mbw; mx++) {
            cso_vp8_mb *M = &mbs[my * mbw + mx];
            segment *S = &E.seg[M->segment];
            const uint8_t *sy = E.sy + (size_t)my * 16 * ys + mx * 16, *su = E.su + (size_t)my * 8 * cs + mx * 8, *sv = E.sv + (size_t)my * 8 * cs + mx * 8;
            uint8_t *ry = E.ry + (size_t)my * 16 * ys + mx * 16, *ru = E.ru + (size_t)my * 8 * cs + mx * 8, *rv = E.rv + (size_t)my * 8 * cs + mx * 8;
            const int hl = mx > 0, ht = my > 0;
            int top[9], left[9];
            if (--cnt < 0) { finalize_probas(P); level_costs(P); cnt = max_count; }

            /* --- i16: four modes, full reconstruction each */
            rd_t rd = {0, 0, 0, 0, MAX_SCORE, 0};
            int best16 = -1;
            int16_t lv16[17][16];          /* [0] Y2, [1 + b] luma block b */
            uint8_t rec16[256];
            int src_flat = 1;
            for (int y = 0; y < 16 && src_flat; y++) for (int x = 0; x < 16; x++) if (sy[y * ys + x] != sy[0]) { src_flat = 0; break; }
            for (int mode = 0; mode < 4; mode++) {
                uint8_t pred[256], rec[256];
                int16_t coef[16][16], dcs[16], y2[16], lv[17][16];
                rd_t cur = {0, 0, 0, 0, 0, 0};
                predN(mode, ry, ys, 16, hl, ht, pred);
                for (int b = 0; b < 16; b++) { fdct4(sy + (b >> 2) * 4 * ys + (b & 3) * 4, ys, pred + (b >> 2) * 64 + (b & 3) * 4, 16, coef[b]); dcs[b] = coef[b][0]; }
                fwht(dcs, y2);
                cur.nz |= (uint32_t)quantize_block(y2, lv[0], &S->y2) << 24;
                for (int b = 0; b < 16; b++) { coef[b][0] = 0; cur.nz |= (uint32_t)quantize_block(coef[b], lv[1 + b], &S->y1) << b; }
                iwht(y2, dcs);
                for (int b = 0; b < 16; b++) { coef[b][0] = dcs[b]; idct4_add(coef[b], pred + (b >> 2) * 64 + (b & 3) * 4, 16, rec + (b >> 2) * 64 + (b & 3) * 4, 16); }
                cur.D = sse(sy, ys, rec, 16, 16, 16);
                cur.SD = S->tlambda ? (S->tlambda * tdisto16(sy, ys, rec, 16) + 128) >> 8 : 0;
                cur.H = kVp8FixedCostsI16[mode];
                nz_import(&E, mx, my, left_y2, top, left);
                cur.R = residual_cost(P, 1, 0, top[8] + left[8], lv[0]);
                for (int y = 0; y < 4; y++)
                    for (int x = 0; x < 4; x++) {
                        cur.R += residual_cost(P, 0, 1, top[x] + left[y], lv[1 + y * 4 + x]);
                        top[x] = left[y] = (cur.nz >> (y * 4 + x)) & 1;
                    }
                if (src_flat) {   /* a flat source whose levels are flat too: distortion counts double */
                    src_flat = flat_levels(lv[1], 16, FLAT_I16);
                    if (src_flat) { cur.D *= 2; cur.SD *= 2; }
                }
                rd_score(&cur, S->lambda_i16);
                if (mode == 0 || cur.score < rd.score) { rd = cur; best16 = mode; memcpy(lv16, lv, sizeof lv); memcpy(rec16, rec, 256); }
            }
            rd_score(&rd, S->lambda_mode);
            if ((rd.nz & 0x100ffff) == 0x1000000 && rd.D > S->min_disto) {   /* only DCs, yet distorted: blocky -> remember the step for the loop filter */
                const int v0 = abs(lv16[0][1]), v1 = abs(lv16[0][2]), v2 = abs(lv16[0][4]);
                int m = v1 > v0 ? v1 : v0;
                if (v2 > m) m = v2;
                if (m > S->max_edge) S->max_edge = m;
            }

            /* --- i4: sixteen sub-blocks in raster order, ten modes each; abandoned as soon as its running score passes the i16 one */
            int use_i4 = 0;
            int16_t lv4[16][16];
            uint8_t rec4[256], modes4[16];
            {
                rd_t best = {0, 0, 0, 211, 0, 0};   /* 211: the cost of the "not i16" flag */
                rd_score(&best, S->lambda_mode);
                uint8_t tr[4];                  /* the four samples above-right of the macroblock */
                for (int i = 0; i < 4; i++) tr[i] = !ht ? 127 : mx + 1 < mbw ? ry[-ys + 16 + i] : ry[-ys + 15];
                nz_import(&E, mx, my, left_y2, top, left);
                int ok = 1, header_bits = 0;
                for (int k = 0; k < 16 && ok; k++) {
                    const int bx = k & 3, by = k >> 2;
                    uint8_t e[13];
                    /* edge samples: from this macroblock's sub-blocks so far (rec4), else from the frame's reconstruction, else the frame-edge constants */
                    for (int i = 0; i < 4; i++) {
                        e[3 - i] = bx ? rec4[(by * 4 + i) * 16 + bx * 4 - 1] : hl ? ry[(by * 4 + i) * ys - 1] : 129;
                        e[5 + i] = by ? rec4[(by * 4 - 1) * 16 + bx * 4 + i] : ht ? ry[-ys + bx * 4 + i] : 127;
                        e[9 + i] = bx == 3 ? tr[i] : by ? rec4[(by * 4 - 1) * 16 + bx * 4 + 4 + i] : ht ? ry[-ys + bx * 4 + 4 + i] : 127;
                    }
                    e[4] = bx && by ? rec4[(by * 4 - 1) * 16 + bx * 4 - 1] : by ? (hl ? ry[(by * 4 - 1) * ys - 1] : 129) : bx ? (ht ? ry[-ys + bx * 4 - 1] : 127) : (ht ? (hl ? ry[-ys - 1] : 129) : 127);
                    const int tmode = by ? modes4[k - 4] : ht ? E.bm[(size_t)(my * 4 - 1) * mbw * 4 + mx * 4 + bx] : 0;
                    const int lmode = bx ? modes4[k - 1] : hl ? E.bm[(size_t)(my * 4 + by) * mbw * 4 + mx * 4 - 1] : 0;
                    const uint16_t *mode_cost = kVp8FixedCostsI4 + (tmode * 10 + lmode) * 10;
                    const uint8_t *src = sy + by * 4 * ys + bx * 4;
                    rd_t bi = {0, 0, 0, 0, MAX_SCORE, 0};
                    int bmode = -1;
                    uint8_t brec[16];
                    for (int mode = 0; mode < 10; mode++) {
                        uint8_t p4[16], r4[16];
                        int16_t c[16], l[16];
                        rd_t t = {0, 0, 0, 0, 0, 0};
                        pred4(mode, e, p4);
                        fdct4(src, ys, p4, 4, c);
                        t.nz = (uint32_t)quantize_block(c, l, &S->y1) << k;
                        idct4_add(c, p4, 4, r4, 4);
                        t.D = sse(src, ys, r4, 4, 4, 4);
                        t.SD = S->tlambda ? (S->tlambda * tdisto4(src, ys, r4, 4) + 128) >> 8 : 0;
                        t.H = mode_cost[mode];
                        t.R = mode > 0 && flat_levels(l, 1, FLAT_I4) ? FLAT_PENALTY : 0;
                        rd_score(&t, S->lambda_i4);
                        if (bmode >= 0 && t.score >= bi.score) continue;
                        t.R += residual_cost(P, 3, 0, top[bx] + left[by], l);
                        rd_score(&t, S->lambda_i4);
                        if (bmode < 0 || t.score < bi.score) { bi = t; bmode = mode; memcpy(brec, r4, 16); memcpy(lv4[k], l, 32); }
                    }
                    rd_score(&bi, S->lambda_mode);
                    rd_add(&best, &bi);
                    if (best.score >= rd.score) { ok = 0; break; }
                    header_bits += (int)bi.H;
                    if (header_bits > 256 * 16 * 16) { ok = 0; break; }
                    for (int y = 0; y < 4; y++) memcpy(rec4 + (by * 4 + y) * 16 + bx * 4, brec + y * 4, 4);
                    modes4[k] = (uint8_t)bmode;
                    top[bx] = left[by] = bi.nz ? 1 : 0;
                }
                if (ok) { use_i4 = 1; rd = best; }
            }

            /* --- chroma: four modes, both planes together; the DC of each block absorbs part of its neighbours' quantisation error */
            rd_t rduv = {0, 0, 0, 0, MAX_SCORE, 0};
            int bestuv = -1;
            int16_t lvuv[8][16];
            uint8_t recuv[2][64];
            int8_t derr[2][3] = {{0, 0, 0}, {0, 0, 0}};
            for (int mode = 0; mode < 4; mode++) {
                uint8_t pred[2][64], rec[2][64];
                int16_t coef[8][16], lv[8][16];
                int8_t de[2][3];
                rd_t cur = {0, 0, 0, 0, 0, 0};
                for (int pl = 0; pl < 2; pl++) {
                    const uint8_t *s = pl ? sv : su;
                    predN(mode, pl ? rv : ru, cs, 8, hl, ht, pred[pl]);
                    for (int b = 0; b < 4; b++) fdct4(s + (b >> 1) * 4 * cs + (b & 1) * 4, cs, pred[pl] + (b >> 1) * 32 + (b & 1) * 4, 8, coef[pl * 4 + b]);
                }
                memset(de, 0, sizeof de);
                for (int pl = 0; pl < 2 && diffuse; pl++) {   /* error diffusion over the 2 x 2 DCs: 7/16 from above, 8/16 from the left */
                    int16_t(*c)[16] = &coef[pl * 4];
                    const int8_t *tp = E.top_derr[mx][pl], *lf = left_derr[pl];
                    c[0][0] = (int16_t)(c[0][0] + ((7 * tp[0] + 8 * lf[0]) >> 3));
                    const int e0 = quantize_single(&c[0][0], &S->uv);
                    c[1][0] = (int16_t)(c[1][0] + ((7 * tp[1] + 8 * e0) >> 3));
                    const int e1 = quantize_single(&c[1][0], &S->uv);
                    c[2][0] = (int16_t)(c[2][0] + ((7 * e0 + 8 * lf[1]) >> 3));
                    const int e2 = quantize_single(&c[2][0], &S->uv);
                    c[3][0] = (int16_t)(c[3][0] + ((7 * e1 + 8 * e2) >> 3));
                    const int e3 = quantize_single(&c[3][0], &S->uv);
                    de[pl][0] = (int8_t)e1; de[pl][1] = (int8_t)e2; de[pl][2] = (int8_t)e3;
                }
                for (int b = 0; b < 8; b++) cur.nz |= (uint32_t)quantize_block(coef[b], lv[b], &S->uv) << (16 + b);
                for (int pl = 0; pl < 2; pl++)
                    for (int b = 0; b < 4; b++) idct4_add(coef[pl * 4 + b], pred[pl] + (b >> 1) * 32 + (b & 1) * 4, 8, rec[pl] + (b >> 1) * 32 + (b & 1) * 4, 8);
                cur.D = sse(su, cs, rec[0], 8, 8, 8) + sse(sv, cs, rec[1], 8, 8, 8);
                cur.H = kVp8FixedCostsUV[mode];
                nz_import(&E, mx, my, left_y2, top, left);
                for (int ch = 0; ch <= 2; ch += 2)
                    for (int y = 0; y < 2; y++)
                        for (int x = 0; x < 2; x++) {
                            const int b = ch * 2 + y * 2 + x;
                            cur.R += residual_cost(P, 2, 0, top[4 + ch + x] + left[4 + ch + y], lv[b]);
                            top[4 + ch + x] = left[4 + ch + y] = (cur.nz >> (16 + b)) & 1;
                        }
                if (mode > 0 && flat_levels(lv[0], 8, FLAT_UV)) cur.R += FLAT_PENALTY * 8;
                rd_score(&cur, S->lambda_uv);
                if (mode == 0 || cur.score < rduv.score) { rduv = cur; bestuv = mode; memcpy(lvuv, lv, sizeof lv); memcpy(recuv, rec, sizeof rec); memcpy(derr, de, sizeof de); }
            }
            rd_add(&rd, &rduv);
            for (int pl = 0; pl < 2; pl++) {   /* errors handed on: e1 to the right, e2 below, e3 split 3/4 right, 1/4 below */
                left_derr[pl][0] = derr[pl][0];
                left_derr[pl][1] = (int8_t)((3 * derr[pl][2]) >> 2);
                E.top_derr[mx][pl][0] = derr[pl][1];
                E.top_derr[mx][pl][1] = (int8_t)(derr[pl][2] - left_derr[pl][1]);
            }

            /* --- commit: reconstruction, modes, levels */
            M->is_i4 = (uint8_t)use_i4; M->ymode = (uint8_t)best16; M->uvmode = (uint8_t)bestuv; M->skip = rd.nz == 0;
            for (int y = 0; y < 16; y++) memcpy(ry + (size_t)y * ys, (use_i4 ? rec4 : rec16) + y * 16, 16);
            for (int y = 0; y < 8; y++) { memcpy(ru + (size_t)y * cs, recuv[0] + y * 8, 8); memcpy(rv + (size_t)y * cs, recuv[1] + y * 8, 8); }
            for (int k = 0; k < 16; k++) {
                M->bmodes[k] = use_i4 ? modes4[k] : (uint8_t)best16;
                E.bm[(size_t)(my * 4 + (k >> 2)) * mbw * 4 + mx * 4 + (k & 3)] = M->bmodes[k];
            }
            if (use_i4) { memset(M->levels[0], 0, 32); memcpy(M->levels[1], lv4, sizeof lv4); } else memcpy(M->levels[0], lv16, sizeof lv16);
            memcpy(M->levels[17], lvuv, sizeof lvuv);

            /* --- tokens + statistics, with the real contexts */
            nz_import(&E, mx, my, left_y2, top, left);
            if (!use_i4) top[8] = left[8] = left_y2 = record_block(&T, P, 1, 0, top[8] + left[8], M->levels[0]);
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) top[x] = left[y] = record_block(&T, P, use_i4 ? 3 : 0, use_i4 ? 0 : 1, top[x] + left[y], M->levels[1 + y * 4 + x]);
            for (int ch = 0; ch <= 2; ch += 2)
                for (int y = 0; y < 2; y++)
                    for (int x = 0; x < 2; x++) top[4 + ch + x] = left[4 + ch + y] = record_block(&T, P, 2, 0, top[4 + ch + x] + left[4 + ch + y], M->levels[17 + ch * 2 + y * 2 + x]);
            E.nz[(size_t)my * mbw + mx] = ((uint32_t)top[0] << 12) | ((uint32_t)top[1] << 13) | ((uint32_t)top[2] << 14) | ((uint32_t)top[3] << 15) | ((uint32_t)top[4] << 18) |
                                          ((uint32_t)top[5] << 19) | ((uint32_t)top[6] << 22) | ((uint32_t)top[7] << 23) | ((uint32_t)top[8] << 24) | ((uint32_t)left[0] << 3) |
                                          ((uint32_t)left[1] << 7) | ((uint32_t)left[2] << 11) | ((uint32_t)left[4] << 17) | ((uint32_t)left[6] << 21);
        }
    }

    /* ---- D: final probabilities, filter strength, partitions */
    finalize_probas(P);
    for (int s = 0; s < 4; s++) {   /* blocky DC-only macroblocks ask for at least this much filtering */
        segment *S = &E.seg[s];
        const int delta = (S->max_edge * S->y2.q[1]) >> 3, level = kVp8LevelsFromDelta[delta > 63 ? 63 : delta];
        if (level > S->fstrength) S->fstrength = level;
    }
    {
        int m = 0;
        for (int s = 0; s < 4; s++) if (m < E.seg[s].fstrength) m = E.seg[s].fstrength;
        filter_level = m;
    }
    boolenc h;
    be_init(&h);
    be_bits(&h, 0, 1);   /* colour space */
    be_bits(&h, 0, 1);   /* clamping */
    if (be_put(&h, nseg > 1, 128)) {
        be_bits(&h, (uint32_t)update_map, 1);
        be_bits(&h, 1, 1);   /* segment data follows ... */
        be_bits(&h, 1, 1);   /* ... as absolute values */
        for (int s = 0; s < 4; s++) be_sbits(&h, E.seg[s].quant, 7);
        for (int s = 0; s < 4; s++) be_sbits(&h, E.seg[s].fstrength, 6);
        if (update_map) for (int s = 0; s < 3; s++) if (be_put(&h, seg_probs[s] != 255, 128)) be_bits(&h, (uint32_t)seg_probs[s], 8);
    }
    be_bits(&h, 0, 1);   /* normal (not simple) loop filter */
    be_bits(&h, (uint32_t)filter_level, 6);
    be_bits(&h, 0, 3);   /* sharpness */
    be_bits(&h, 0, 1);   /* no filter deltas */
    be_bits(&h, 0, 2);   /* one token partition */
    be_bits(&h, (uint32_t)base_quant, 7);
    be_sbits(&h, 0, 4); be_sbits(&h, 0, 4); be_sbits(&h, 0, 4); be_sbits(&h, dq_uv_dc, 4); be_sbits(&h, dq_uv_ac, 4);
    be_bits(&h, 0, 1);   /* refresh_entropy_probs */
    for (int i = 0; i < NSLOTS; i++) if (be_put(&h, P->coeffs[i] != kVp8CoefProbs[i], kVp8CoefUpdateProbs[i])) be_bits(&h, P->coeffs[i], 8);
    be_bits(&h, 0, 1);   /* no skip flags */
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            const cso_vp8_mb *M = &mbs[my * mbw + mx];
            if (update_map) { const int s = M->segment; if (be_put(&h, s >= 2, seg_probs[0])) be_put(&h, s & 1, seg_probs[2]); else be_put(&h, s & 1, seg_probs[1]); }
            if (be_put(&h, !M->is_i4, 145)) {
                if (be_put(&h, M->ymode == 1 || M->ymode == 3, 156)) be_put(&h, M->ymode == 1, 128); else be_put(&h, M->ymode == 2, 163);
            } else
                for (int k = 0; k < 16; k++) {
                    const int bx = k & 3, by = k >> 2, m = M->bmodes[k];
                    const int tmode = my * 4 + by > 0 ? E.bm[(size_t)(my * 4 + by - 1) * mbw * 4 + mx * 4 + bx] : 0, lmode = mx * 4 + bx > 0 ? E.bm[(size_t)(my * 4 + by) * mbw * 4 + mx * 4 + bx - 1] : 0;
                    const uint8_t *pr = kVp8BModeProbs + (tmode * 10 + lmode) * 9;
                    if (be_put(&h, m != 0, pr[0]) && be_put(&h, m != 1, pr[1]) && be_put(&h, m != 2, pr[2])) {
                        if (!be_put(&h, m >= 6, pr[3])) { if (be_put(&h, m != 3, pr[4])) be_put(&h, m != 4, pr[5]); }
                        else if (be_put(&h, m != 6, pr[6]) && be_put(&h, m != 7, pr[7])) be_put(&h, m != 8, pr[8]);
                    }
                }
            if (be_put(&h, M->uvmode != 0, 142) && be_put(&h, M->uvmode != 2, 114)) be_put(&h, M->uvmode != 3, 183);
        }
    be_finish(&h);
    boolenc t;
    be_init(&t);
    for (size_t i = 0; i < T.n; i++) { const uint16_t v = T.t[i]; be_put(&t, v >> 15, (v & 0x4000) ? (v & 0xff) : P->coeffs[v & 0x3fff]); }
    be_finish(&t);

    size_t vp8 = 10 + h.pos + t.pos;
    const size_t pad = vp8 & 1;
    vp8 += pad;
    const size_t total = 12 + 8 + vp8;
    uint8_t *o = (uint8_t *)calloc(total, 1), *w = o;
    memcpy(w, "RIFF", 4); w[4] = (uint8_t)(total - 8); w[5] = (uint8_t)((total - 8) >> 8); w[6] = (uint8_t)((total - 8) >> 16); w[7] = (uint8_t)((total - 8) >> 24);
    memcpy(w + 8, "WEBPVP8 ", 8); w[16] = (uint8_t)vp8; w[17] = (uint8_t)(vp8 >> 8); w[18] = (uint8_t)(vp8 >> 16); w[19] = (uint8_t)(vp8 >> 24);
    w += 20;
    const uint32_t tag = ((uint32_t)h.pos << 5) | (1u << 4) | (0u << 1) | 0u;   /* key frame, profile 0 (normal filter), shown */
    w[0] = (uint8_t)tag; w[1] = (uint8_t)(tag >> 8); w[2] = (uint8_t)(tag >> 16);
    w[3] = 0x9D; w[4] = 0x01; w[5] = 0x2A;
    w[6] = (uint8_t)width; w[7] = (uint8_t)(width >> 8); w[8] = (uint8_t)height; w[9] = (uint8_t)(height >> 8);
    memcpy(w + 10, h.buf, h.pos);
    memcpy(w + 10 + h.pos, t.buf, t.pos);
    *out = o; *out_len = total;

    if (frame) {
        memset(frame, 0, sizeof *frame);
        frame->width = width; frame->height = height; frame->mbw = mbw; frame->mbh = mbh;
        frame->num_segments = nseg; frame->update_map = update_map;
        for (int s = 0; s < 4; s++) { frame->seg_quant[s] = E.seg[s].quant; frame->seg_filter[s] = E.seg[s].fstrength; frame->seg_alpha[s] = E.seg[s].alpha; frame->seg_beta[s] = E.seg[s].beta; frame->seg_max_edge[s] = E.seg[s].max_edge; }
        for (int s = 0; s < 3; s++) frame->seg_probs[s] = seg_probs[s];
        frame->filter_level = filter_level; frame->base_quant = base_quant; frame->dq[3] = dq_uv_dc; frame->dq[4] = dq_uv_ac;
        memcpy(frame->probas, P->coeffs, NSLOTS);
        frame->part0_size = h.pos; frame->vp8_size = vp8;
        frame->alpha_avg = alpha_sum / nmb; frame->uv_alpha_avg = uv_alpha;
    }
    free(h.buf); free(t.buf); free(T.t);
    free(E.ry); free(E.ru); free(E.rv); free(E.nz); free(E.bm); free(E.top_derr);
    if (!mbs_out) free(mbs);
    return 0;
}

int cso_vp8enc_encode_rgb(const uint8_t *rgb, int width, int height, float quality, uint8_t **out, size_t *out_len) {
    const int mbw = (width + 15) >> 4, mbh = (height + 15) >> 4;
    uint8_t *yp = (uint8_t *)malloc((size_t)mbw * mbh * 256), *up = (uint8_t *)malloc((size_t)mbw * mbh * 64), *vp = (uint8_t *)malloc((size_t)mbw * mbh * 64);
    cso_webp_rgb_to_yuv(rgb, width, height, yp, up, vp);
    const int rc = cso_vp8enc_encode_yuv(yp, up, vp, width, height, quality, out, out_len, NULL, NULL);
    free(yp); free(up); free(vp);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ stream parser (RFC 6386 9, 13, 19) */
typedef struct { const uint8_t *p, *end; uint32_t value, range; int count; } booldec;
static uint32_t bd_byte(booldec *d) { return d->p < d->end ? *d->p++ : 0; }
static void bd_init(booldec *d, const uint8_t *p, size_t n) { d->p = p; d->end = p + n; d->value = bd_byte(d) << 8; d->value |= bd_byte(d); d->range = 255; d->count = 0; }
static int bd_get(booldec *d, int prob) {
    const uint32_t split = 1 + (((d->range - 1) * (uint32_t)prob) >> 8), big = split << 8;
    int bit;
    if (d->value >= big) { bit = 1; d->range -= split; d->value -= big; } else { bit = 0; d->range = split; }
    while (d->range < 128) {
        d->value <<= 1;
        d->range <<= 1;
        if (++d->count == 8) { d->count = 0; d->value |= bd_byte(d); }
    }
    return bit;
}
static int bd_lit(booldec *d, int n) { int v = 0; while (n--) v = (v << 1) | bd_get(d, 128); return v; }
static int bd_slit(booldec *d, int n) { const int v = bd_lit(d, n); return bd_get(d, 128) ? -v : v; }
static int parse_block(booldec *d, const uint8_t *probs, int type, int ctx, int first, int16_t *lv) {
    int n = first;
    const uint8_t *p = probs + slot(type, kVp8Bands[n], ctx);
    memset(lv, 0, 32);
    if (!bd_get(d, p[0])) return 0;
    while (n < 16) {
        if (!bd_get(d, p[1])) { p = probs + slot(type, kVp8Bands[++n], 0); continue; }
        int v;
        if (!bd_get(d, p[2])) { v = 1; p = probs + slot(type, kVp8Bands[n + 1], 1); }
        else {
            if (!bd_get(d, p[3])) { if (!bd_get(d, p[4])) v = 2; else v = 3 + bd_get(d, p[5]); }
            else if (!bd_get(d, p[6])) {
                if (!bd_get(d, p[7])) v = 5 + bd_get(d, 159);
                else { v = 7 + 2 * bd_get(d, 165); v += bd_get(d, 145); }
            } else {
                const int b1 = bd_get(d, p[8]), b0 = bd_get(d, p[9 + b1]), cat = 2 * b1 + b0;
                const uint8_t *tab = cat == 0 ? kVp8Cat3 : cat == 1 ? kVp8Cat4 : cat == 2 ? kVp8Cat5 : kVp8Cat6;
                v = 0;
                for (; *tab; tab++) v = v + v + bd_get(d, *tab);
                v += 3 + (8 << cat);
            }
            p = probs + slot(type, kVp8Bands[n + 1], 2);
        }
        lv[n++] = (int16_t)(bd_get(d, 128) ? -v : v);
        if (n == 16 || !bd_get(d, p[0])) return 1;
    }
    return 1;
}
int cso_vp8_parse(const uint8_t *data, size_t n, cso_vp8_frame *F, cso_vp8_mb *mbs, size_t cap) {
    if (n >= 20 && !memcmp(data, "RIFF", 4) && !memcmp(data + 8, "WEBPVP8 ", 8)) { data += 20; n -= 20; }
    if (n < 10) return -1;
    memset(F, 0, sizeof *F);
    const uint32_t tag = data[0] | (data[1] << 8) | ((uint32_t)data[2] << 16);
    if (tag & 1) return -2;
    F->part0_size = tag >> 5;
    F->width = (data[6] | (data[7] << 8)) & 0x3fff; F->height = (data[8] | (data[9] << 8)) & 0x3fff;
    const int mbw = F->mbw = (F->width + 15) >> 4, mbh = F->mbh = (F->height + 15) >> 4;
    if ((size_t)mbw * mbh > cap || 10 + F->part0_size > n) return -3;
    booldec h;
    bd_init(&h, data + 10, F->part0_size);
    bd_lit(&h, 2);
    F->num_segments = 1;
    F->seg_probs[0] = F->seg_probs[1] = F->seg_probs[2] = 255;
    int segmented = bd_lit(&h, 1);
    if (segmented) {
        F->num_segments = 4;
        F->update_map = bd_lit(&h, 1);
        if (bd_lit(&h, 1)) {
            bd_lit(&h, 1);
            for (int s = 0; s < 4; s++) F->seg_quant[s] = bd_lit(&h, 1) ? bd_slit(&h, 7) : 0;
            for (int s = 0; s < 4; s++) F->seg_filter[s] = bd_lit(&h, 1) ? bd_slit(&h, 6) : 0;
        }
        if (F->update_map) for (int s = 0; s < 3; s++) F->seg_probs[s] = bd_lit(&h, 1) ? bd_lit(&h, 8) : 255;
    }
    F->filter_simple = bd_lit(&h, 1); F->filter_level = bd_lit(&h, 6); F->filter_sharpness = bd_lit(&h, 3);
    if (bd_lit(&h, 1) && bd_lit(&h, 1)) { for (int i = 0; i < 8; i++) if (bd_lit(&h, 1)) bd_slit(&h, 6); }
    F->num_parts_log2 = bd_lit(&h, 2);
    F->base_quant = bd_lit(&h, 7);
    for (int i = 0; i < 5; i++) F->dq[i] = bd_lit(&h, 1) ? bd_slit(&h, 4) : 0;
    bd_lit(&h, 1);
    memcpy(F->probas, kVp8CoefProbs, NSLOTS);
    for (int i = 0; i < NSLOTS; i++) if (bd_get(&h, kVp8CoefUpdateProbs[i])) F->probas[i] = (uint8_t)bd_lit(&h, 8);
    F->use_skip = bd_lit(&h, 1);
    if (F->use_skip) F->skip_proba = bd_lit(&h, 8);
    const int nparts = 1 << F->num_parts_log2;
    const uint8_t *pp = data + 10 + F->part0_size + 3 * (nparts - 1);
    booldec td[8];
    {
        const uint8_t *sz = data + 10 + F->part0_size, *q = pp;
        for (int p = 0; p < nparts; p++) {
            size_t len = p + 1 < nparts ? (size_t)(sz[0] | (sz[1] << 8) | (sz[2] << 16)) : (size_t)(data + n - q);
            if (q + len > data + n) return -4;
            bd_init(&td[p], q, len);
            q += len; sz += 3;
        }
    }
    uint8_t *tbm = (uint8_t *)calloc((size_t)mbw * 4, 1), *tnz = (uint8_t *)calloc((size_t)mbw, 9);
    for (int my = 0; my < mbh; my++) {
        uint8_t lbm[4] = {0, 0, 0, 0}, lnz[9] = {0};
        booldec *d = &td[my & (nparts - 1)];
        for (int mx = 0; mx < mbw; mx++) {
            cso_vp8_mb *M = &mbs[my * mbw + mx];
            memset(M, 0, sizeof *M);
            if (F->update_map) M->segment = (uint8_t)(bd_get(&h, F->seg_probs[0]) ? 2 + bd_get(&h, F->seg_probs[2]) : bd_get(&h, F->seg_probs[1]));
            if (F->use_skip) M->skip = (uint8_t)bd_get(&h, F->skip_proba);
            uint8_t *tb = tbm + mx * 4;
            if (bd_get(&h, 145)) {
                M->ymode = (uint8_t)(bd_get(&h, 156) ? (bd_get(&h, 128) ? 1 : 3) : (bd_get(&h, 163) ? 2 : 0));
                for (int k = 0; k < 16; k++) M->bmodes[k] = M->ymode;
                for (int k = 0; k < 4; k++) tb[k] = lbm[k] = M->ymode;
            } else {
                M->is_i4 = 1;
                for (int k = 0; k < 16; k++) {
                    const uint8_t *pr = kVp8BModeProbs + (tb[k & 3] * 10 + lbm[k >> 2]) * 9;
                    int m;
                    if (!bd_get(&h, pr[0])) m = 0; else if (!bd_get(&h, pr[1])) m = 1; else if (!bd_get(&h, pr[2])) m = 2;
                    else if (!bd_get(&h, pr[3])) m = !bd_get(&h, pr[4]) ? 3 : !bd_get(&h, pr[5]) ? 4 : 5;
                    else m = !bd_get(&h, pr[6]) ? 6 : !bd_get(&h, pr[7]) ? 7 : !bd_get(&h, pr[8]) ? 8 : 9;
                    M->bmodes[k] = tb[k & 3] = lbm[k >> 2] = (uint8_t)m;
                }
            }
            M->uvmode = (uint8_t)(!bd_get(&h, 142) ? 0 : !bd_get(&h, 114) ? 2 : !bd_get(&h, 183) ? 3 : 1);
            uint8_t *tp = tnz + mx * 9;
            if (M->skip) { if (!M->is_i4) tp[8] = lnz[8] = 0; for (int k = 0; k < 8; k++) tp[k] = lnz[k] = 0; continue; }
            if (!M->is_i4) tp[8] = lnz[8] = (uint8_t)parse_block(d, F->probas, 1, tp[8] + lnz[8], 0, M->levels[0]);
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) tp[x] = lnz[y] = (uint8_t)parse_block(d, F->probas, M->is_i4 ? 3 : 0, tp[x] + lnz[y], M->is_i4 ? 0 : 1, M->levels[1 + y * 4 + x]);
            for (int ch = 0; ch <= 2; ch += 2)
                for (int y = 0; y < 2; y++)
                    for (int x = 0; x < 2; x++) tp[4 + ch + x] = lnz[4 + ch + y] = (uint8_t)parse_block(d, F->probas, 2, tp[4 + ch + x] + lnz[4 + ch + y], 0, M->levels[17 + ch * 2 + y * 2 + x]);
        }
    }
    free(tbm); free(tnz);
    F->vp8_size = n;
    return 0;
}
