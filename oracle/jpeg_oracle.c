/*
 * jpeg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See jpeg_oracle.h.
 *
 * Restates, in plain C, the published algorithms that libcaesium 0.20.3 reaches through
 * mozjpeg-sys 2.2.1 for the JPEG hot path (reference call sites
 * /root/reference/src/compressor.rs:287-306; dependency pins /root/reference/Cargo.lock:892-913,
 * 1035-1044).  mozjpeg's source is NOT under /root/reference; every function below names the
 * public algorithm it restates (ITU-T T.81 clause / libjpeg module behaviour as summarised in
 * SURVEY.md Appendix B) and is pinned by tests/test_oracle_*.py against libjpeg-turbo 3.1.4.1
 * (via Pillow) and against the reference's own fixtures samples/j0.JPG, samples/level_1_0/j1.jpg.
 *
 * Profiles: "plain" = ISLOW integer DCTs, scalar quantiser, optimal Huffman tables, stock progression
 * (or a supplied script) -- byte-pinned to libjpeg-turbo; + mozjpeg's scan search (scan_script 2, pinned by
 * samples/j0.JPG); + mozjpeg's trellis quantiser and overshoot deringing (cso_enc_params.trellis /
 * .deringing: written down from memory of mozjpeg 4.1's jcdctmgr.c, PARITY UNPINNED -- DESIGN.md).
 *
 * Provenance of the trellis routine: quantize_trellis_row below is NOT an independent restatement of a published
 * algorithm.  It follows mozjpeg's quantize_trellis (jcdctmgr.c, BSD-3-Clause / IJG licence) identifier for identifier
 * (accumulated_zero_dist, accumulated_cost, run_start, dc_cost_backtrack, lambda_table, the candidate loops and the float
 * evaluation order), because the product has to reproduce that routine's choices bit for bit and its behaviour is defined
 * by nothing but its source.  mozjpeg is not under /root/reference; this file is test infrastructure only and nothing of it
 * is compiled into, linked with or executed by the product (caesium-clt_amd/csrc/k_trellis.hip is a different programme: one
 * wave per block, predecessor sets in registers -- DESIGN.md 4.6).
 */
#include "jpeg_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char *cso_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* T.81 Figure A.6: zig-zag index -> natural index */
static const uint8_t ZZ[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void cso_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------ */
/* byte vector                                                                                */
typedef struct { uint8_t *p; size_t n, cap; } bvec;
static void bv_reserve(bvec *b, size_t extra) {
    if (b->n + extra <= b->cap) return;
    size_t c = b->cap ? b->cap * 2 : 4096;
    while (c < b->n + extra) c *= 2;
    b->p = (uint8_t *)realloc(b->p, c); b->cap = c;
}
static void bv_put(bvec *b, int v) { bv_reserve(b, 1); b->p[b->n++] = (uint8_t)v; }
static void bv_put2(bvec *b, int v) { bv_put(b, v >> 8); bv_put(b, v & 255); }
static void bv_write(bvec *b, const void *s, size_t n) { bv_reserve(b, n); memcpy(b->p + b->n, s, n); b->n += n; }

/* ------------------------------------------------------------------------------------------ */
/* geometry                                                                                   */
static int ceil_div(int a, int b) { return (a + b - 1) / b; }

static int setup_geometry(cso_image *im) {
    im->hmax = im->vmax = 1;
    for (int c = 0; c < im->ncomp; c++) {
        if (im->comp[c].h < 1 || im->comp[c].h > 4 || im->comp[c].v < 1 || im->comp[c].v > 4) FAIL("bad sampling factor");
        if (im->comp[c].h > im->hmax) im->hmax = im->comp[c].h;
        if (im->comp[c].v > im->vmax) im->vmax = im->comp[c].v;
    }
    im->mcus_x = ceil_div(im->width, 8 * im->hmax);
    im->mcus_y = ceil_div(im->height, 8 * im->vmax);
    for (int c = 0; c < im->ncomp; c++) {
        cso_comp *k = &im->comp[c];
        k->comp_w = ceil_div(im->width * k->h, im->hmax);
        k->comp_h = ceil_div(im->height * k->v, im->vmax);
        k->real_bw = ceil_div(k->comp_w, 8);
        k->real_bh = ceil_div(k->comp_h, 8);
        k->bw = im->mcus_x * k->h;
        k->bh = im->mcus_y * k->v;
        k->coef = (int16_t *)calloc((size_t)k->bw * k->bh * 64, sizeof(int16_t));
        if (!k->coef) FAIL("out of memory");
    }
    return 0;
}

void cso_image_free(cso_image *im) {
    if (!im) return;
    for (int c = 0; c < CSO_MAX_COMPS; c++) free(im->comp[c].coef);
    free(im->meta);
    free(im);
}

/* ------------------------------------------------------------------------------------------ */
/* Huffman decode tables (T.81 Annex C generation + F.2.2.3 decode, with a 9-bit lookahead)   */
typedef struct {
    int present;
    uint8_t bits[17], huffval[256];
    int32_t maxcode[18], valptr[17];
    uint16_t look[512]; /* (len<<8)|sym, 0 = slow path */
} dhuff;

static int build_dhuff(dhuff *h) {
    int code = 0, p = 0;
    uint16_t huffcode[257]; uint8_t huffsize[257];
    for (int l = 1; l <= 16; l++) for (int i = 0; i < h->bits[l]; i++) { if (p >= 256) FAIL("bad DHT"); huffsize[p++] = (uint8_t)l; }
    int n = p; p = 0;
    int si = huffsize[0];
    while (p < n) {
        while (p < n && huffsize[p] == si) huffcode[p++] = (uint16_t)code++;
        if (code > (1 << si)) FAIL("bad DHT codes");
        code <<= 1; si++;
    }
    p = 0;
    for (int l = 1; l <= 16; l++) {
        if (h->bits[l]) { h->valptr[l] = p - huffcode[p]; p += h->bits[l]; h->maxcode[l] = huffcode[p - 1]; }
        else h->maxcode[l] = -1;
    }
    h->maxcode[17] = 0xFFFFF;
    memset(h->look, 0, sizeof h->look);
    p = 0;
    for (int l = 1; l <= 9; l++)
        for (int i = 0; i < h->bits[l]; i++, p++) {
            int base = huffcode[p] << (9 - l);
            for (int k = 0; k < (1 << (9 - l)); k++) h->look[base + k] = (uint16_t)((l << 8) | h->huffval[p]);
        }
    return 0;
}

/* bit reader over one entropy-coded segment, T.81 F.2.2.5 (0xFF00 unstuffing) */
typedef struct {
    const uint8_t *p, *end;
    uint64_t acc; int nbits;
    int hit_marker; /* marker code seen (0 = none) */
    /* libjpeg jdhuff.c "insufficient_data": once the decoder has CONSUMED more bits than the segment holds (zeros are fed
       for the rest of that MCU), every following MCU up to the next restart is skipped, i.e. left as it is (zero blocks
       in a sequential or first scan).  `real` = bits of acc that came from the file. */
    int real, insufficient;
} breader;

static void br_fill(breader *b) {
    while (b->nbits <= 56) {
        int c = 0;
        if (!b->hit_marker && b->p < b->end) {
            c = *b->p;
            if (c == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0x00) b->p += 2;
                else { /* marker (or truncated): feed zeros from here on */
                    b->hit_marker = (b->p + 1 < b->end) ? b->p[1] : 0xD9;
                    c = 0;
                }
            } else b->p++;
            if (!b->hit_marker) b->real += 8;
        }
        b->acc |= (uint64_t)c << (56 - b->nbits);
        b->nbits += 8;
    }
}
static inline int br_peek(breader *b, int n) { if (b->nbits < n) br_fill(b); return (int)(b->acc >> (64 - n)); }
static inline void br_skip(breader *b, int n) { b->acc <<= n; b->nbits -= n; if (n > b->real) { b->insufficient = 1; b->real = 0; } else b->real -= n; }
static inline int br_get(breader *b, int n) { if (n == 0) return 0; int v = br_peek(b, n); br_skip(b, n); return v; }
static inline int hdecode(breader *b, const dhuff *h) {
    int v = br_peek(b, 16);
    int e = h->look[v >> 7];
    if (e) { br_skip(b, e >> 8); return e & 255; }
    for (int l = 10; l <= 16; l++) {
        int c = v >> (16 - l);
        if (c <= h->maxcode[l]) { br_skip(b, l); return h->huffval[(h->valptr[l] + c) & 255]; }
    }
    br_skip(b, 16);
    return 0; /* corrupt: libjpeg also substitutes zero */
}
static inline int extend(int r, int n) { return r < (1 << (n - 1)) ? r - (1 << n) + 1 : r; }

/* ------------------------------------------------------------------------------------------ */
/* scan decode: sequential (T.81 F.2.2) and progressive (T.81 G.2)            */
typedef struct {
    cso_image *im;
    const cso_scan *sc;
    dhuff *dc[CSO_MAX_COMPS], *ac[CSO_MAX_COMPS]; /* per scan component */
    int pred[CSO_MAX_COMPS];
    int eobrun;
    breader br;
} sdec;

static void dec_block_seq(sdec *s, int si, int16_t *blk) {
    breader *b = &s->br;
    int t = hdecode(b, s->dc[si]);
    int diff = t ? extend(br_get(b, t), t) : 0;
    s->pred[si] += diff;
    blk[0] = (int16_t)s->pred[si];
    for (int k = 1; k < 64;) {
        int rs = hdecode(b, s->ac[si]);
        int r = rs >> 4, n = rs & 15;
        if (n) { k += r; if (k > 63) break; blk[ZZ[k]] = (int16_t)extend(br_get(b, n), n); k++; }
        else { if (r == 15) k += 16; else break; }
    }
}
static void dec_block_dc_first(sdec *s, int si, int16_t *blk) {
    breader *b = &s->br;
    int t = hdecode(b, s->dc[si]);
    int diff = t ? extend(br_get(b, t), t) : 0;
    s->pred[si] += diff;
    blk[0] = (int16_t)(s->pred[si] * (1 << s->sc->Al));
}
static void dec_block_dc_refine(sdec *s, int16_t *blk) {
    if (br_get(&s->br, 1)) blk[0] |= (int16_t)(1 << s->sc->Al);
}
static void dec_block_ac_first(sdec *s, int16_t *blk) {
    breader *b = &s->br;
    const cso_scan *sc = s->sc;
    if (s->eobrun > 0) { s->eobrun--; return; }
    for (int k = sc->Ss; k <= sc->Se; k++) {
        int rs = hdecode(b, s->ac[0]);
        int r = rs >> 4, n = rs & 15;
        if (n) { k += r; if (k > 63) break; blk[ZZ[k]] = (int16_t)(extend(br_get(b, n), n) * (1 << sc->Al)); }
        else {
            if (r == 15) k += 15;
            else { s->eobrun = (1 << r); if (r) s->eobrun += br_get(b, r); s->eobrun--; break; }
        }
    }
}
static void dec_block_ac_refine(sdec *s, int16_t *blk) {
    breader *b = &s->br;
    const cso_scan *sc = s->sc;
    int p1 = 1 << sc->Al, m1 = -p1;
    int k = sc->Ss;
    if (s->eobrun == 0) {
        for (; k <= sc->Se; k++) {
            int rs = hdecode(b, s->ac[0]);
            int r = rs >> 4, n = rs & 15, val = 0;
            if (n) { val = br_get(b, 1) ? p1 : m1; }
            else if (r != 15) {
                s->eobrun = 1 << r; if (r) s->eobrun += br_get(b, r);
                break; /* to EOB tail */
            }
            do {
                int16_t *c = &blk[ZZ[k]];
                if (*c != 0) {
                    if (br_get(b, 1)) { if ((*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1); }
                } else { if (--r < 0) break; }
                k++;
            } while (k <= sc->Se);
            if (val && k <= 63) blk[ZZ[k]] = (int16_t)val;
        }
    }
    if (s->eobrun > 0) {
        for (; k <= sc->Se; k++) {
            int16_t *c = &blk[ZZ[k]];
            if (*c != 0 && br_get(b, 1)) { if ((*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1); }
        }
        s->eobrun--;
    }
}

static void dec_restart(sdec *s) {
    breader *b = &s->br;
    /* discard partial byte + any fill, then consume RSTn */
    b->acc = 0; b->nbits = 0; b->real = 0;
    if (b->hit_marker >= 0xD0 && b->hit_marker <= 0xD7) { b->p += 2; b->hit_marker = 0; b->insufficient = 0; }
    else {
        /* bit reader may not have touched the marker yet: scan forward */
        while (b->p + 1 < b->end && !(b->p[0] == 0xFF && b->p[1] >= 0xD0 && b->p[1] <= 0xD7)) b->p++;
        if (b->p + 1 < b->end) { b->p += 2; b->insufficient = 0; }
    }
    for (int i = 0; i < CSO_MAX_COMPS; i++) s->pred[i] = 0;
    s->eobrun = 0;
}

static int decode_scan(cso_image *im, const cso_scan *sc, dhuff dctab[4], dhuff actab[4],
                       const int td[], const int ta[], const uint8_t *p, const uint8_t *end) {
    sdec s; memset(&s, 0, sizeof s);
    s.im = im; s.sc = sc; s.br.p = p; s.br.end = end;
    int dc_scan = sc->Ss == 0;
    for (int i = 0; i < sc->ncomp_in_scan; i++) {
        int need_dc = im->progressive ? (dc_scan && sc->Ah == 0) : 1;
        int need_ac = im->progressive ? (!dc_scan) : 1;
        if ((need_dc && td[i] > 3) || (need_ac && ta[i] > 3)) FAIL("SOS table id out of range");
        s.dc[i] = &dctab[td[i] & 3]; s.ac[i] = &actab[ta[i] & 3];
        if (need_dc && !s.dc[i]->present) FAIL("missing DC Huffman table %d", td[i]);
        if (need_ac && !s.ac[i]->present) FAIL("missing AC Huffman table %d", ta[i]);
    }
    if (im->progressive) {   /* libjpeg jdphuff.c start_pass_phuff_decoder: JERR_BAD_PROGRESSION */
        if (dc_scan) { if (sc->Se != 0) FAIL("bad progressive DC scan"); }
        else if (sc->ncomp_in_scan != 1 || sc->Se < sc->Ss || sc->Se > 63) FAIL("bad progressive AC scan");
        if ((sc->Ah != 0 && sc->Al != sc->Ah - 1) || sc->Al > 13) FAIL("bad progressive parameters");
    }
    int ri = im->restart_interval, todo = ri;
    if (sc->ncomp_in_scan == 1) {
        cso_comp *k = &im->comp[sc->comp_idx[0]];
        for (int by = 0; by < k->real_bh; by++)
            for (int bx = 0; bx < k->real_bw; bx++) {
                if (ri) { if (todo == 0) { dec_restart(&s); todo = ri; } todo--; }
                if (s.br.insufficient) continue;
                int16_t *blk = k->coef + ((size_t)by * k->bw + bx) * 64;
                if (!im->progressive) dec_block_seq(&s, 0, blk);
                else if (dc_scan) { if (sc->Ah == 0) dec_block_dc_first(&s, 0, blk); else dec_block_dc_refine(&s, blk); }
                else { if (sc->Ah == 0) dec_block_ac_first(&s, blk); else dec_block_ac_refine(&s, blk); }
            }
    } else {
        for (int my = 0; my < im->mcus_y; my++)
            for (int mx = 0; mx < im->mcus_x; mx++) {
                if (ri) { if (todo == 0) { dec_restart(&s); todo = ri; } todo--; }
                if (s.br.insufficient) continue;
                for (int i = 0; i < sc->ncomp_in_scan; i++) {
                    cso_comp *k = &im->comp[sc->comp_idx[i]];
                    for (int y = 0; y < k->v; y++)
                        for (int x = 0; x < k->h; x++) {
                            int16_t *blk = k->coef + ((size_t)(my * k->v + y) * k->bw + (mx * k->h + x)) * 64;
                            if (!im->progressive) dec_block_seq(&s, i, blk);
                            else if (sc->Ah == 0) dec_block_dc_first(&s, i, blk);
                            else dec_block_dc_refine(&s, blk);
                        }
                }
            }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* marker parser (T.81 Annex B)                                                               */
int cso_decode(const uint8_t *d, size_t n, cso_image **out) {
    *out = NULL;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8 || d[2] != 0xFF) FAIL("not a JPEG (no SOI + marker: libcaesium sniffs FF D8 FF)");
    cso_image *im = (cso_image *)calloc(1, sizeof *im);
    dhuff *dctab = (dhuff *)calloc(4, sizeof(dhuff)), *actab = (dhuff *)calloc(4, sizeof(dhuff));
    bvec meta = {0};
    int rc = -1, have_sof = 0;
    im->adobe_transform = -1;
    size_t i = 2;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) { i++; continue; }
        int m = d[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) { i += 2; continue; }
        size_t L = ((size_t)d[i + 2] << 8) | d[i + 3];
        if (L < 2 || i + 2 + L > n) { snprintf(g_err, sizeof g_err, "truncated marker segment"); goto done; }
        const uint8_t *s = d + i + 4; size_t sl = L - 2;
        if (m == 0xDB) {
            size_t j = 0;
            while (j < sl) {
                int pq = s[j] >> 4, tq = s[j] & 15; j++;
                if (tq > 3 || j + (pq ? 128 : 64) > sl) { snprintf(g_err, sizeof g_err, "bad DQT"); goto done; }
                for (int k = 0; k < 64; k++) {
                    int v = pq ? ((s[j] << 8) | s[j + 1]) : s[j];
                    j += pq ? 2 : 1;
                    im->qt[tq][ZZ[k]] = (uint16_t)v;
                }
                im->qt_present[tq] = 1;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (have_sof || sl < 6) { snprintf(g_err, sizeof g_err, "bad SOF"); goto done; }
            im->progressive = (m == 0xC2);
            im->precision = s[0];
            im->height = (s[1] << 8) | s[2]; im->width = (s[3] << 8) | s[4]; im->ncomp = s[5];
            if (im->precision != 8 || im->ncomp < 1 || im->ncomp > 4 || !im->width || !im->height || sl < 6 + 3u * im->ncomp) {
                snprintf(g_err, sizeof g_err, "unsupported SOF"); goto done; }
            for (int c = 0; c < im->ncomp; c++) {
                im->comp[c].id = s[6 + 3 * c]; im->comp[c].h = s[7 + 3 * c] >> 4; im->comp[c].v = s[7 + 3 * c] & 15;
                im->comp[c].tq = s[8 + 3 * c];   /* libjpeg keeps the byte and fails on a table number >= 4 when the table is needed (JERR_NO_QUANT_TABLE) */
            }
            if (setup_geometry(im)) goto done;
            have_sof = 1;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            snprintf(g_err, sizeof g_err, "unsupported JPEG process SOF%d", m - 0xC0); goto done;
        } else if (m == 0xC4) {
            size_t j = 0;
            while (j + 17 <= sl) {
                int tc = s[j] >> 4, th = s[j] & 15;
                if (tc > 1 || th > 3) { snprintf(g_err, sizeof g_err, "bad DHT"); goto done; }
                dhuff *h = tc ? &actab[th] : &dctab[th];
                memset(h, 0, sizeof *h);
                int cnt = 0;
                for (int l = 1; l <= 16; l++) { h->bits[l] = s[j + l]; cnt += s[j + l]; }
                j += 17;
                if (cnt > 256 || j + cnt > sl) { snprintf(g_err, sizeof g_err, "bad DHT"); goto done; }
                memcpy(h->huffval, s + j, cnt); j += cnt;
                if (!tc) for (int i = 0; i < cnt; i++) if (h->huffval[i] > 15) { snprintf(g_err, sizeof g_err, "bad DHT (DC symbol > 15)"); goto done; }  /* libjpeg JERR_BAD_HUFF_TABLE */
                if (build_dhuff(h)) goto done;
                h->present = 1;
            }
        } else if (m == 0xDD) {
            if (sl >= 2) im->restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof) { snprintf(g_err, sizeof g_err, "SOS before SOF"); goto done; }
            int ns = s[0];
            if (ns < 1 || ns > 4 || sl < 1u + 2 * ns + 3) { snprintf(g_err, sizeof g_err, "bad SOS"); goto done; }
            if (im->nscans >= CSO_MAX_SCANS) { snprintf(g_err, sizeof g_err, "too many scans"); goto done; }
            cso_scan *sc = &im->scans[im->nscans];
            int td[4], ta[4];
            sc->ncomp_in_scan = ns;
            for (int k = 0; k < ns; k++) {
                int cid = s[1 + 2 * k], ci = -1;
                for (int c = 0; c < im->ncomp; c++) if (im->comp[c].id == cid) ci = c;
                if (ci < 0) { snprintf(g_err, sizeof g_err, "SOS names unknown component"); goto done; }
                for (int q = 0; q < k; q++) if (sc->comp_idx[q] == ci) { snprintf(g_err, sizeof g_err, "SOS names a component twice"); goto done; }  /* libjpeg JERR_BAD_COMPONENT_ID */
                sc->comp_idx[k] = ci; td[k] = s[2 + 2 * k] >> 4; ta[k] = s[2 + 2 * k] & 15;   /* a number >= 4 fails when the table is needed (libjpeg: JERR_NO_HUFF_TABLE) */
            }
            sc->Ss = s[1 + 2 * ns]; sc->Se = s[2 + 2 * ns]; sc->Ah = s[3 + 2 * ns] >> 4; sc->Al = s[3 + 2 * ns] & 15;
            if (!im->progressive) { sc->Ss = 0; sc->Se = 63; sc->Ah = sc->Al = 0; }
            im->nscans++;
            /* entropy-coded segment runs to the next non-RST marker */
            size_t e = i + 2 + L;
            while (e + 1 < n) {
                if (d[e] == 0xFF && d[e + 1] != 0x00 && !(d[e + 1] >= 0xD0 && d[e + 1] <= 0xD7) && d[e + 1] != 0xFF) break;
                e++;
            }
            if (e + 1 >= n) e = n;
            if (decode_scan(im, sc, dctab, actab, td, ta, d + i + 2 + L, d + e)) goto done;
            i = e; continue;
        } else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            if (m == 0xE0 && sl >= 5 && !memcmp(s, "JFIF\0", 5)) im->saw_jfif = 1;
            if (m == 0xEE && sl >= 12 && !memcmp(s, "Adobe", 5)) im->adobe_transform = s[11];
            bv_write(&meta, d + i, 2 + L);
        }
        i += 2 + L;
    }
    if (!have_sof || im->nscans == 0) { snprintf(g_err, sizeof g_err, "no image data"); goto done; }
    for (int c = 0; c < im->ncomp; c++)   /* libjpeg JERR_NO_QUANT_TABLE: at decompression start, and at emit_dqt when transcoding */
        if (im->comp[c].tq < 0 || im->comp[c].tq > 3 || !im->qt_present[im->comp[c].tq]) { snprintf(g_err, sizeof g_err, "missing quantisation table"); goto done; }
    im->meta = meta.p; im->meta_len = meta.n; meta.p = NULL;
    rc = 0;
done:
    free(meta.p); free(dctab); free(actab);
    if (rc) { cso_image_free(im); return -1; }
    *out = im;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* jidctint / jfdctint constants: FIX(x) = round(x * 2^13)               */
#define CONST_BITS 13
#define PASS1_BITS 2
#define F_0_298 2446
#define F_0_390 3196
#define F_0_541 4433
#define F_0_765 6270
#define F_0_899 7373
#define F_1_175 9633
#define F_1_501 12299
#define F_1_847 15137
#define F_1_961 16069
#define F_2_053 16819
#define F_2_562 20995
#define F_3_072 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

/* libjpeg jidctint.c behaviour (ISLOW inverse DCT, column pass then row pass) + range limit */
void cso_idct_islow(const int16_t coef[64], const uint16_t qt[64], uint8_t out[64]) {
    int32_t ws[64];
    for (int c = 0; c < 8; c++) {
        int32_t x0 = coef[c] * qt[c], x1 = coef[8 + c] * qt[8 + c], x2 = coef[16 + c] * qt[16 + c], x3 = coef[24 + c] * qt[24 + c];
        int32_t x4 = coef[32 + c] * qt[32 + c], x5 = coef[40 + c] * qt[40 + c], x6 = coef[48 + c] * qt[48 + c], x7 = coef[56 + c] * qt[56 + c];
        int32_t z1 = (x2 + x6) * F_0_541, tmp2 = z1 - x6 * F_1_847, tmp3 = z1 + x2 * F_0_765;
        int32_t tmp0 = (x0 + x4) * (1 << CONST_BITS), tmp1 = (x0 - x4) * (1 << CONST_BITS);
        int32_t t10 = tmp0 + tmp3, t13 = tmp0 - tmp3, t11 = tmp1 + tmp2, t12 = tmp1 - tmp2;
        int32_t a0 = x7, a1 = x5, a2 = x3, a3 = x1;
        int32_t y1 = a0 + a3, y2 = a1 + a2, y3 = a0 + a2, y4 = a1 + a3, y5 = (y3 + y4) * F_1_175;
        a0 *= F_0_298; a1 *= F_2_053; a2 *= F_3_072; a3 *= F_1_501;
        y1 *= -F_0_899; y2 *= -F_2_562; y3 = y3 * -F_1_961 + y5; y4 = y4 * -F_0_390 + y5;
        a0 += y1 + y3; a1 += y2 + y4; a2 += y2 + y3; a3 += y1 + y4;
        ws[c] = DESCALE(t10 + a3, CONST_BITS - PASS1_BITS); ws[56 + c] = DESCALE(t10 - a3, CONST_BITS - PASS1_BITS);
        ws[8 + c] = DESCALE(t11 + a2, CONST_BITS - PASS1_BITS); ws[48 + c] = DESCALE(t11 - a2, CONST_BITS - PASS1_BITS);
        ws[16 + c] = DESCALE(t12 + a1, CONST_BITS - PASS1_BITS); ws[40 + c] = DESCALE(t12 - a1, CONST_BITS - PASS1_BITS);
        ws[24 + c] = DESCALE(t13 + a0, CONST_BITS - PASS1_BITS); ws[32 + c] = DESCALE(t13 - a0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; r++) {
        const int32_t *w = ws + 8 * r;
        int32_t x0 = w[0], x1 = w[1], x2 = w[2], x3 = w[3], x4 = w[4], x5 = w[5], x6 = w[6], x7 = w[7];
        int32_t z1 = (x2 + x6) * F_0_541, tmp2 = z1 - x6 * F_1_847, tmp3 = z1 + x2 * F_0_765;
        int32_t tmp0 = (x0 + x4) * (1 << CONST_BITS), tmp1 = (x0 - x4) * (1 << CONST_BITS);
        int32_t t10 = tmp0 + tmp3, t13 = tmp0 - tmp3, t11 = tmp1 + tmp2, t12 = tmp1 - tmp2;
        int32_t a0 = x7, a1 = x5, a2 = x3, a3 = x1;
        int32_t y1 = a0 + a3, y2 = a1 + a2, y3 = a0 + a2, y4 = a1 + a3, y5 = (y3 + y4) * F_1_175;
        a0 *= F_0_298; a1 *= F_2_053; a2 *= F_3_072; a3 *= F_1_501;
        y1 *= -F_0_899; y2 *= -F_2_562; y3 = y3 * -F_1_961 + y5; y4 = y4 * -F_0_390 + y5;
        a0 += y1 + y3; a1 += y2 + y4; a2 += y2 + y3; a3 += y1 + y4;
        int32_t o[8];
        const int SH = CONST_BITS + PASS1_BITS + 3;
        o[0] = DESCALE(t10 + a3, SH); o[7] = DESCALE(t10 - a3, SH);
        o[1] = DESCALE(t11 + a2, SH); o[6] = DESCALE(t11 - a2, SH);
        o[2] = DESCALE(t12 + a1, SH); o[5] = DESCALE(t12 - a1, SH);
        o[3] = DESCALE(t13 + a0, SH); o[4] = DESCALE(t13 - a0, SH);
        for (int c = 0; c < 8; c++) { int v = o[c] + 128; out[8 * r + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
    }
}

/* libjpeg jfdctint.c behaviour (ISLOW forward DCT, rows then columns; output scaled by 8), on level-shifted samples
   (the -128 belongs to the sample-conversion step in front: mozjpeg's deringing sits between the two). */
static void fdct_islow_ls(const int32_t *s, int32_t d[64]) {
    for (int r = 0; r < 8; r++) {
        const int32_t *p = s + 8 * r; int32_t *o = d + 8 * r;
        int32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4], d5 = p[5], d6 = p[6], d7 = p[7];
        int32_t tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
        int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        o[0] = (tmp10 + tmp11) * (1 << PASS1_BITS); o[4] = (tmp10 - tmp11) * (1 << PASS1_BITS);
        int32_t z1 = (tmp12 + tmp13) * F_0_541;
        o[2] = DESCALE(z1 + tmp13 * F_0_765, CONST_BITS - PASS1_BITS);
        o[6] = DESCALE(z1 - tmp12 * F_1_847, CONST_BITS - PASS1_BITS);
        int32_t y1 = tmp4 + tmp7, y2 = tmp5 + tmp6, y3 = tmp4 + tmp6, y4 = tmp5 + tmp7, y5 = (y3 + y4) * F_1_175;
        tmp4 *= F_0_298; tmp5 *= F_2_053; tmp6 *= F_3_072; tmp7 *= F_1_501;
        y1 *= -F_0_899; y2 *= -F_2_562; y3 = y3 * -F_1_961 + y5; y4 = y4 * -F_0_390 + y5;
        o[7] = DESCALE(tmp4 + y1 + y3, CONST_BITS - PASS1_BITS); o[5] = DESCALE(tmp5 + y2 + y4, CONST_BITS - PASS1_BITS);
        o[3] = DESCALE(tmp6 + y2 + y3, CONST_BITS - PASS1_BITS); o[1] = DESCALE(tmp7 + y1 + y4, CONST_BITS - PASS1_BITS);
    }
    for (int c = 0; c < 8; c++) {
        int32_t *o = d + c;
        int32_t d0 = o[0], d1 = o[8], d2 = o[16], d3 = o[24], d4 = o[32], d5 = o[40], d6 = o[48], d7 = o[56];
        int32_t tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
        int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        o[0] = DESCALE(tmp10 + tmp11, PASS1_BITS); o[32] = DESCALE(tmp10 - tmp11, PASS1_BITS);
        int32_t z1 = (tmp12 + tmp13) * F_0_541;
        o[16] = DESCALE(z1 + tmp13 * F_0_765, CONST_BITS + PASS1_BITS);
        o[48] = DESCALE(z1 - tmp12 * F_1_847, CONST_BITS + PASS1_BITS);
        int32_t y1 = tmp4 + tmp7, y2 = tmp5 + tmp6, y3 = tmp4 + tmp6, y4 = tmp5 + tmp7, y5 = (y3 + y4) * F_1_175;
        tmp4 *= F_0_298; tmp5 *= F_2_053; tmp6 *= F_3_072; tmp7 *= F_1_501;
        y1 *= -F_0_899; y2 *= -F_2_562; y3 = y3 * -F_1_961 + y5; y4 = y4 * -F_0_390 + y5;
        o[56] = DESCALE(tmp4 + y1 + y3, CONST_BITS + PASS1_BITS); o[40] = DESCALE(tmp5 + y2 + y4, CONST_BITS + PASS1_BITS);
        o[24] = DESCALE(tmp6 + y2 + y3, CONST_BITS + PASS1_BITS); o[8] = DESCALE(tmp7 + y1 + y4, CONST_BITS + PASS1_BITS);
    }
}
void cso_fdct_islow(const uint8_t *s, int32_t d[64]) {
    int32_t ls[64];
    for (int i = 0; i < 64; i++) ls[i] = (int32_t)s[i] - 128;
    fdct_islow_ls(ls, d);
}

/* mozjpeg jcdctmgr.c preprocess_deringing + catmull_rom ("overshoot deringing", on by default in the JCP_MAX_COMPRESSION profile)
   [UPSTREAM-RECALL, UNPINNED].  On the level-shifted samples of one block, walked in zig-zag order as one line: every run of samples at
   the top of the range (>= 127) is replaced by a Catmull-Rom arc through the slopes either side of it, so that the clipped plateau
   overshoots (by at most min(31, 2 * DC quantiser, the headroom the block's mean leaves)) instead of ringing.  float arithmetic, no
   contraction (x86-64 build of the crate), DCTELEM truncation where the C source passes a float to a DCTELEM parameter. */
static float dering_catmull_rom(int value1, int value2, int value3, int value4, float t, int size) {
    const int tan1 = (value3 - value1) * size, tan2 = (value4 - value2) * size;
    const float t2 = t * t, t3 = t2 * t;
    const float f1 = 2.f * t3 - 3.f * t2 + 1.f, f2 = -2.f * t3 + 3.f * t2, f3 = t3 - 2.f * t2 + t, f4 = t3 - t2;
    return value2 * f1 + tan1 * f3 + value3 * f2 + tan2 * f4;
}
void cso_dering_block(int32_t data[64], int dc_quant) {
    const int maxsample = 255 - 128, size = 64;
    int sum = 0, cnt = 0;
    for (int i = 0; i < size; i++) { sum += data[i]; if (data[i] >= maxsample) cnt++; }
    if (!cnt || cnt == size) return;
    int over = 2 * dc_quant < 31 ? 2 * dc_quant : 31, room = (maxsample * size - sum) / cnt;
    const int maxovershoot = maxsample + (over < room ? over : room);
    int n = 0;
    do {
        if (data[ZZ[n]] < maxsample) { n++; continue; }
        const int start = n;
        while (++n < size && data[ZZ[n]] >= maxsample) {}
        const int end = n;
        const float f1 = (float)data[ZZ[start >= 1 ? start - 1 : 0]], f2 = (float)data[ZZ[start >= 2 ? start - 2 : 0]];
        const float l1 = (float)data[ZZ[end < size - 1 ? end : size - 1]], l2 = (float)data[ZZ[end < size - 2 ? end + 1 : size - 1]];
        float fslope = f1 - f2 > maxsample - f1 ? f1 - f2 : maxsample - f1;
        float lslope = l1 - l2 > maxsample - l1 ? l1 - l2 : maxsample - l1;
        if (start == 0) fslope = lslope;
        if (end == size) lslope = fslope;
        const int length = end - start;
        const float step = 1.f / (float)(length + 1);
        float position = step;
        for (int i = start; i < end; i++, position += step) {
            const int tmp = (int)ceilf(dering_catmull_rom((int)(maxsample - fslope), maxsample, maxsample, (int)(maxsample - lslope), position, length));
            data[ZZ[i]] = tmp < maxovershoot ? tmp : maxovershoot;
        }
        n++;
    } while (n < size);
}

/* ------------------------------------------------------------------------------------------ */
/* decode to samples                                                                          */
int cso_decode_plane(const cso_image *im, int ci, uint8_t *out) {
    const cso_comp *k = &im->comp[ci];
    if (!im->qt_present[k->tq]) FAIL("missing quantisation table %d", k->tq);
    uint8_t px[64];
    for (int by = 0; by < k->real_bh; by++)
        for (int bx = 0; bx < k->real_bw; bx++) {
            cso_idct_islow(k->coef + ((size_t)by * k->bw + bx) * 64, im->qt[k->tq], px);
            for (int y = 0; y < 8; y++) {
                int yy = by * 8 + y; if (yy >= k->comp_h) break;
                for (int x = 0; x < 8; x++) { int xx = bx * 8 + x; if (xx < k->comp_w) out[(size_t)yy * k->comp_w + xx] = px[8 * y + x]; }
            }
        }
    return 0;
}

/* libjpeg jdsample.c behaviour: h2v2 / h2v1 "fancy" (triangle) upsampling on the REAL
   downsampled plane; plain replication otherwise. */
static void upsample_plane(const uint8_t *in, int cw, int ch, int hx, int vx, uint8_t *out, int W, int H) {
    if (hx == 1 && vx == 1) {
        for (int y = 0; y < H; y++) memcpy(out + (size_t)y * W, in + (size_t)y * cw, W);
    } else if (hx == 2 && vx == 2 && cw > 2) {
        int32_t *cs = (int32_t *)malloc(sizeof(int32_t) * cw);
        uint8_t *row = (uint8_t *)malloc(2 * (size_t)cw);
        for (int oy = 0; oy < H; oy++) {
            int y = oy >> 1;
            const uint8_t *nr = in + (size_t)y * cw;
            int fy = (oy & 1) ? y + 1 : y - 1;
            if (fy < 0) fy = 0; if (fy > ch - 1) fy = ch - 1;
            const uint8_t *fr = in + (size_t)fy * cw;
            for (int x = 0; x < cw; x++) cs[x] = 3 * nr[x] + fr[x];
            row[0] = (uint8_t)((cs[0] * 4 + 8) >> 4);
            row[1] = (uint8_t)((cs[0] * 3 + cs[1] + 7) >> 4);
            for (int x = 1; x < cw - 1; x++) {
                row[2 * x] = (uint8_t)((cs[x] * 3 + cs[x - 1] + 8) >> 4);
                row[2 * x + 1] = (uint8_t)((cs[x] * 3 + cs[x + 1] + 7) >> 4);
            }
            row[2 * (cw - 1)] = (uint8_t)((cs[cw - 1] * 3 + cs[cw - 2] + 8) >> 4);
            row[2 * (cw - 1) + 1] = (uint8_t)((cs[cw - 1] * 4 + 7) >> 4);
            memcpy(out + (size_t)oy * W, row, W);
        }
        free(cs); free(row);
    } else if (hx == 2 && vx == 1 && cw > 2) {
        uint8_t *row = (uint8_t *)malloc(2 * (size_t)cw);
        for (int y = 0; y < H; y++) {
            const uint8_t *r = in + (size_t)y * cw;
            row[0] = r[0];
            row[1] = (uint8_t)((r[0] * 3 + r[1] + 2) >> 2);
            for (int x = 1; x < cw - 1; x++) {
                row[2 * x] = (uint8_t)((r[x] * 3 + r[x - 1] + 1) >> 2);
                row[2 * x + 1] = (uint8_t)((r[x] * 3 + r[x + 1] + 2) >> 2);
            }
            row[2 * (cw - 1)] = (uint8_t)((r[cw - 1] * 3 + r[cw - 2] + 1) >> 2);
            row[2 * (cw - 1) + 1] = r[cw - 1];
            memcpy(out + (size_t)y * W, row, W);
        }
        free(row);
    } else if (hx == 1 && vx == 2) {
        /* libjpeg-turbo h1v2_fancy_upsample (unvalidated here: Pillow cannot write 4:4:0) */
        for (int oy = 0; oy < H; oy++) {
            int y = oy >> 1, fy = (oy & 1) ? y + 1 : y - 1, bias = (oy & 1) ? 2 : 1;
            if (fy < 0) fy = 0; if (fy > ch - 1) fy = ch - 1;
            for (int x = 0; x < W; x++) out[(size_t)oy * W + x] = (uint8_t)((3 * in[(size_t)y * cw + x] + in[(size_t)fy * cw + x] + bias) >> 2);
        }
    } else {
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) out[(size_t)y * W + x] = in[(size_t)(y / vx) * cw + x / hx];
    }
    (void)ch;
}

int cso_decode_pixels(const cso_image *im, uint8_t *out) {
    int W = im->width, H = im->height;
    uint8_t *full = (uint8_t *)malloc((size_t)W * H);
    for (int c = 0; c < im->ncomp; c++) {
        const cso_comp *k = &im->comp[c];
        if (im->hmax % k->h || im->vmax % k->v) { free(full); FAIL("fractional sampling ratio unsupported"); }
        uint8_t *pl = (uint8_t *)malloc((size_t)k->comp_w * k->comp_h);
        if (cso_decode_plane(im, c, pl)) { free(pl); free(full); return -1; }
        upsample_plane(pl, k->comp_w, k->comp_h, im->hmax / k->h, im->vmax / k->v, full, W, H);
        for (size_t i = 0; i < (size_t)W * H; i++) out[i * im->ncomp + c] = full[i];
        free(pl);
    }
    free(full);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* quality -> tables (libjpeg jcparam.c jpeg_quality_scaling + jpeg_add_quant_table)*/
static const uint16_t BASE_ANNEXK_L[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint16_t BASE_ANNEXK_C[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
/* mozjpeg base table index 3, luma == chroma; pinned 64/64 by the DQT of /root/reference/samples/j0.JPG */
static const uint16_t BASE_MOZ3[64] = {16, 16, 16, 18, 25, 37, 56, 85, 16, 17, 20, 27, 34, 40, 53, 75, 16, 20, 24, 31, 43, 62, 91, 135, 18, 27, 31, 40, 53, 74, 106, 156, 25, 34, 43, 53, 69, 94, 131, 189, 37, 40, 62, 74, 94, 124, 169, 238, 56, 53, 91, 106, 131, 169, 226, 311, 85, 75, 135, 156, 189, 238, 311, 418};

void cso_quality_tables(int q, int profile, int force_baseline, uint16_t out[2][64]) {
    if (q <= 0) q = 1; if (q > 100) q = 100;
    int s = q < 50 ? 5000 / q : 200 - 2 * q;
    const uint16_t *bl = profile == 3 ? BASE_MOZ3 : BASE_ANNEXK_L, *bc = profile == 3 ? BASE_MOZ3 : BASE_ANNEXK_C;
    for (int i = 0; i < 64; i++) {
        long a = ((long)bl[i] * s + 50) / 100, b = ((long)bc[i] * s + 50) / 100;
        if (a <= 0) a = 1; if (a > 32767) a = 32767; if (force_baseline && a > 255) a = 255;
        if (b <= 0) b = 1; if (b > 32767) b = 32767; if (force_baseline && b > 255) b = 255;
        out[0][i] = (uint16_t)a; out[1][i] = (uint16_t)b;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* forward path: libjpeg jcprepct/jcsample/jcdctmgr/jccoefct behaviour   */
/* blocks that exist only to complete an MCU (jccoefct.c): zero AC, DC of the last real block in the row (right edge) / of block
   (h-1) of the MCU in the row above (bottom) */
static void make_dummy_blocks(cso_image *im, int ci) {
    cso_comp *k = &im->comp[ci];
    for (int by = 0; by < k->real_bh; by++)
        for (int bx = k->real_bw; bx < k->bw; bx++) {
            int16_t *o = k->coef + ((size_t)by * k->bw + bx) * 64;
            memset(o, 0, 128); o[0] = o[-64];
        }
    for (int by = k->real_bh; by < k->bh; by++)
        for (int m = 0; m < im->mcus_x; m++) {
            int16_t last = k->coef[((size_t)(by - 1) * k->bw + m * k->h + k->h - 1) * 64];
            for (int x = 0; x < k->h; x++) { int16_t *o = k->coef + ((size_t)by * k->bw + m * k->h + x) * 64; memset(o, 0, 128); o[0] = last; }
        }
}
static void forward_component(const uint8_t *full, int W, int H, cso_image *im, int ci, int deringing, int16_t *raw /* unquantised DCT per padded block, or NULL */) {
    cso_comp *k = &im->comp[ci];
    int hx = im->hmax / k->h, vx = im->vmax / k->v;
    int pw = k->real_bw * 8, ph = k->bh * 8;         /* sample plane fed to the DCT */
    int in_cols = pw * hx;                           /* expand_right_edge target */
    int in_rows = ceil_div(H, im->vmax) * im->vmax;  /* bottom padded to a row group */
    int out_rows = in_rows / vx;                     /* downsampled rows that exist before bottom replication */
    uint8_t *pl = (uint8_t *)malloc((size_t)pw * ph);
    uint8_t *r0 = (uint8_t *)malloc(in_cols), *r1 = (uint8_t *)malloc(in_cols);
    for (int oy = 0; oy < out_rows; oy++) {
        uint8_t *o = pl + (size_t)oy * pw;
        for (int v = 0; v < vx && v < 2; v++) {
            int y = oy * vx + v; if (y > H - 1) y = H - 1;
            uint8_t *r = v ? r1 : r0;
            memcpy(r, full + (size_t)y * W, W);
            for (int x = W; x < in_cols; x++) r[x] = r[W - 1];
        }
        if (hx == 1 && vx == 1) memcpy(o, r0, pw);
        else if (hx == 2 && vx == 1) { int bias = 0; for (int x = 0; x < pw; x++) { o[x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + bias) >> 1); bias ^= 1; } }
        else if (hx == 2 && vx == 2) { int bias = 1; for (int x = 0; x < pw; x++) { o[x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + bias) >> 2); bias ^= 3; } }
        else { /* int_downsample: box average with rounding */
            int np = hx * vx;
            for (int x = 0; x < pw; x++) {
                long sum = 0;
                for (int v = 0; v < vx; v++) { int y = oy * vx + v; if (y > H - 1) y = H - 1; for (int h = 0; h < hx; h++) { int xx = x * hx + h; if (xx > W - 1) xx = W - 1; sum += full[(size_t)y * W + xx]; } }
                o[x] = (uint8_t)((sum + np / 2) / np);
            }
        }
    }
    for (int oy = out_rows; oy < ph; oy++) memcpy(pl + (size_t)oy * pw, pl + (size_t)(out_rows - 1) * pw, pw);
    free(r0); free(r1);

    const uint16_t *qt = im->qt[k->tq];
    int32_t s[64], d[64];
    for (int by = 0; by < k->real_bh; by++) {
        for (int bx = 0; bx < k->real_bw; bx++) {
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) s[8 * y + x] = (int32_t)pl[(size_t)(by * 8 + y) * pw + bx * 8 + x] - 128;
            if (deringing) cso_dering_block(s, qt[0]);
            fdct_islow_ls(s, d);
            int16_t *o = k->coef + ((size_t)by * k->bw + bx) * 64;
            if (raw) for (int i = 0; i < 64; i++) raw[((size_t)by * k->bw + bx) * 64 + i] = (int16_t)d[i];
            for (int i = 0; i < 64; i++) {
                int32_t qv = (int32_t)qt[i] << 3, t = d[i];
                if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
                else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
                o[i] = (int16_t)t;
            }
        }
    }
    make_dummy_blocks(im, ci);
    free(pl);
}

static void trellis_image(cso_image *im, int16_t *const raw[], const cso_enc_params *p);
int cso_forward(const uint8_t *pix, int w, int h, int ncomp, const cso_enc_params *p, const uint16_t *qto, cso_image **out) {
    *out = NULL;
    if (ncomp != 1 && ncomp != 3) FAIL("forward: only 1 or 3 components");
    cso_image *im = (cso_image *)calloc(1, sizeof *im);
    im->width = w; im->height = h; im->ncomp = ncomp; im->precision = 8; im->progressive = p->progressive; im->adobe_transform = -1;
    int ss = p->subsampling ? p->subsampling : 420;
    for (int c = 0; c < ncomp; c++) { im->comp[c].id = c + 1; im->comp[c].h = im->comp[c].v = 1; im->comp[c].tq = c ? 1 : 0; }
    if (ncomp == 3) {
        if (ss == 420) { im->comp[0].h = 2; im->comp[0].v = 2; }
        else if (ss == 422) { im->comp[0].h = 2; im->comp[0].v = 1; }
        else if (ss == 411) { im->comp[0].h = 4; im->comp[0].v = 1; }
        else if (ss != 444) { cso_image_free(im); FAIL("bad subsampling %d", ss); }
    }
    if (setup_geometry(im)) { cso_image_free(im); return -1; }
    uint16_t t[2][64];
    if (qto) memcpy(t, qto, sizeof t); else cso_quality_tables(p->quality, p->qtable_profile, p->force_baseline, t);
    memcpy(im->qt[0], t[0], 128); memcpy(im->qt[1], t[1], 128); im->qt_present[0] = 1; im->qt_present[1] = ncomp > 1;
    uint8_t *full = (uint8_t *)malloc((size_t)w * h);
    int16_t *raw[CSO_MAX_COMPS] = {0};
    for (int c = 0; c < ncomp; c++) {
        for (size_t i = 0; i < (size_t)w * h; i++) full[i] = pix[i * ncomp + c];
        if (p->trellis) raw[c] = (int16_t *)calloc((size_t)im->comp[c].bw * im->comp[c].bh * 64, sizeof(int16_t));
        forward_component(full, w, h, im, c, p->deringing, raw[c]);
    }
    free(full);
    if (p->trellis) { trellis_image(im, raw, p); for (int c = 0; c < ncomp; c++) free(raw[c]); }
    *out = im;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* optimal Huffman table: libjpeg jpeg_gen_optimal_table behaviour (T.81 K.2 + libjpeg's      */
/* pseudo-symbol 256 and 16-bit length limiting)                               */
int cso_gen_optimal_table(const long freq_in[257], uint8_t bits_out[17], uint8_t huffval[256]) {
    long freq[257]; int codesize[257], others[257]; uint8_t bits[33];
    memcpy(freq, freq_in, sizeof freq);
    memset(bits, 0, sizeof bits); memset(codesize, 0, sizeof codesize);
    for (int i = 0; i < 257; i++) others[i] = -1;
    freq[256] = 1;
    for (;;) {
        int c1 = -1, c2 = -1; long v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2]; freq[c2] = 0;
        codesize[c1]++; while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++; while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i <= 256; i++) if (codesize[i]) { if (codesize[i] > 32) return -1; bits[codesize[i]]++; }
    for (int i = 32; i > 16; i--)
        while (bits[i] > 0) {
            int j = i - 2; while (bits[j] == 0) j--;
            bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
        }
    int i = 16; while (bits[i] == 0) i--;
    bits[i]--;
    memcpy(bits_out, bits, 17); bits_out[0] = 0;
    int p = 0;
    for (int l = 1; l <= 32; l++) for (int s = 0; s <= 255; s++) if (codesize[s] == l) huffval[p++] = (uint8_t)s;
    return p;
}

/* ------------------------------------------------------------------------------------------ */
/* entropy encoder: tokens first (symbol + raw bits), then statistics, tables, bit packing.   */
/* Sequential: libjpeg jchuff.c behaviour; progressive: jcphuff.c behaviour. */
typedef struct { uint8_t tbl; /* 0-3 DC id, 4-7 AC id, 255 raw */ uint8_t sym, nbits; uint16_t bits; } token;
typedef struct { token *t; size_t n, cap; } tvec;
static void tv_push(tvec *v, int tbl, int sym, int nbits, unsigned bits) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 65536; v->t = (token *)realloc(v->t, v->cap * sizeof(token)); }
    token k; k.tbl = (uint8_t)tbl; k.sym = (uint8_t)sym; k.nbits = (uint8_t)nbits; k.bits = (uint16_t)(bits & ((1u << nbits) - 1));
    v->t[v->n++] = k;
}
static inline int bitlen(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

typedef struct {
    tvec *tv; int actbl; int Ss, Se, Al;
    unsigned eobrun; int be; uint8_t bebuf[1000 + 64];
} penc;

static void emit_buffered(penc *e, const uint8_t *buf, int n) { for (int i = 0; i < n; i++) tv_push(e->tv, 255, 0, 1, buf[i]); }
static void emit_eobrun(penc *e) {
    if (e->eobrun > 0) {
        int nb = bitlen(e->eobrun) - 1;
        tv_push(e->tv, 4 + e->actbl, nb << 4, nb, e->eobrun);
        e->eobrun = 0;
        emit_buffered(e, e->bebuf, e->be); e->be = 0;
    }
}
static void enc_ac_first(penc *e, const int16_t *blk) {
    int r = 0;
    for (int k = e->Ss; k <= e->Se; k++) {
        int t = blk[ZZ[k]], t2;
        if (t == 0) { r++; continue; }
        if (t < 0) { t = -t; t >>= e->Al; t2 = ~t; } else { t >>= e->Al; t2 = t; }
        if (t == 0) { r++; continue; }
        if (e->eobrun > 0) emit_eobrun(e);
        while (r > 15) { tv_push(e->tv, 4 + e->actbl, 0xF0, 0, 0); r -= 16; }
        int nb = bitlen((unsigned)t);
        tv_push(e->tv, 4 + e->actbl, (r << 4) + nb, nb, (unsigned)t2);
        r = 0;
    }
    if (r > 0) { e->eobrun++; if (e->eobrun == 0x7FFF) emit_eobrun(e); }
}
static void enc_ac_refine(penc *e, const int16_t *blk) {
    int absv[64], EOB = 0, r = 0, BR = 0;
    uint8_t *brbuf = e->bebuf + e->be;
    for (int k = e->Ss; k <= e->Se; k++) {
        int t = blk[ZZ[k]]; if (t < 0) t = -t; t >>= e->Al; absv[k] = t; if (t == 1) EOB = k;
    }
    for (int k = e->Ss; k <= e->Se; k++) {
        int t = absv[k];
        if (t == 0) { r++; continue; }
        while (r > 15 && k <= EOB) {
            emit_eobrun(e);
            tv_push(e->tv, 4 + e->actbl, 0xF0, 0, 0); r -= 16;
            emit_buffered(e, brbuf, BR); brbuf = e->bebuf; BR = 0;
        }
        if (t > 1) { brbuf[BR++] = (uint8_t)(t & 1); continue; }
        emit_eobrun(e);
        tv_push(e->tv, 4 + e->actbl, (r << 4) + 1, 1, blk[ZZ[k]] < 0 ? 0 : 1);
        emit_buffered(e, brbuf, BR); brbuf = e->bebuf; BR = 0; r = 0;
    }
    if (r > 0 || BR > 0) {
        e->eobrun++; e->be += BR;
        if (e->eobrun == 0x7FFF || e->be > (1000 - 64 + 1)) emit_eobrun(e);
    }
}

static void tokenize_scan(const cso_image *im, const cso_scan *sc, tvec *tv) {
    int pred[CSO_MAX_COMPS] = {0};
    int seq = !im->progressive;
    if (sc->ncomp_in_scan == 1) {
        const cso_comp *k = &im->comp[sc->comp_idx[0]];
        int dctbl = sc->comp_idx[0] ? 1 : 0, actbl = dctbl;
        penc e; memset(&e, 0, sizeof e); e.tv = tv; e.actbl = actbl; e.Ss = sc->Ss; e.Se = sc->Se; e.Al = sc->Al;
        for (int by = 0; by < k->real_bh; by++)
            for (int bx = 0; bx < k->real_bw; bx++) {
                const int16_t *blk = k->coef + ((size_t)by * k->bw + bx) * 64;
                if (seq || sc->Ss == 0) {
                    if (!seq && sc->Ah) { tv_push(tv, 255, 0, 1, (unsigned)(blk[0] >> sc->Al) & 1); }
                    else {
                        int t2 = seq ? blk[0] : (blk[0] >> sc->Al);
                        int t = t2 - pred[0]; pred[0] = t2; t2 = t; if (t < 0) { t = -t; t2--; }
                        int nb = bitlen((unsigned)t); tv_push(tv, dctbl, nb, nb, (unsigned)t2);
                    }
                    if (seq) {
                        int r = 0;
                        for (int kk = 1; kk < 64; kk++) {
                            int t = blk[ZZ[kk]]; if (t == 0) { r++; continue; }
                            while (r > 15) { tv_push(tv, 4 + actbl, 0xF0, 0, 0); r -= 16; }
                            int t2 = t; if (t < 0) { t = -t; t2--; }
                            int nb = bitlen((unsigned)t); tv_push(tv, 4 + actbl, (r << 4) + nb, nb, (unsigned)t2); r = 0;
                        }
                        if (r > 0) tv_push(tv, 4 + actbl, 0, 0, 0);
                    }
                } else if (sc->Ah == 0) enc_ac_first(&e, blk);
                else enc_ac_refine(&e, blk);
            }
        emit_eobrun(&e);
    } else {
        for (int my = 0; my < im->mcus_y; my++)
            for (int mx = 0; mx < im->mcus_x; mx++)
                for (int i = 0; i < sc->ncomp_in_scan; i++) {
                    int ci = sc->comp_idx[i]; const cso_comp *k = &im->comp[ci];
                    int dctbl = ci ? 1 : 0, actbl = dctbl;
                    for (int y = 0; y < k->v; y++)
                        for (int x = 0; x < k->h; x++) {
                            const int16_t *blk = k->coef + ((size_t)(my * k->v + y) * k->bw + mx * k->h + x) * 64;
                            if (!seq && sc->Ah) { tv_push(tv, 255, 0, 1, (unsigned)(blk[0] >> sc->Al) & 1); continue; }
                            int t2 = seq ? blk[0] : (blk[0] >> sc->Al);
                            int t = t2 - pred[ci]; pred[ci] = t2; t2 = t; if (t < 0) { t = -t; t2--; }
                            int nb = bitlen((unsigned)t); tv_push(tv, dctbl, nb, nb, (unsigned)t2);
                            if (seq) {
                                int r = 0;
                                for (int kk = 1; kk < 64; kk++) {
                                    int tt = blk[ZZ[kk]]; if (tt == 0) { r++; continue; }
                                    while (r > 15) { tv_push(tv, 4 + actbl, 0xF0, 0, 0); r -= 16; }
                                    int tt2 = tt; if (tt < 0) { tt = -tt; tt2--; }
                                    int nb2 = bitlen((unsigned)tt); tv_push(tv, 4 + actbl, (r << 4) + nb2, nb2, (unsigned)tt2); r = 0;
                                }
                                if (r > 0) tv_push(tv, 4 + actbl, 0, 0, 0);
                            }
                        }
                }
    }
}

typedef struct { uint8_t bits[17], huffval[256]; int nsym; uint16_t code[256]; uint8_t size[256]; int used; } ehuff;
static void derive_ehuff(ehuff *h) {
    int p = 0, code = 0;
    memset(h->size, 0, sizeof h->size);
    for (int l = 1; l <= 16; l++) { for (int i = 0; i < h->bits[l]; i++, p++) { h->code[h->huffval[p]] = (uint16_t)code++; h->size[h->huffval[p]] = (uint8_t)l; } code <<= 1; }
}

/* ------------------------------------------------------------------------------------------ */
/* mozjpeg's trellis quantiser (jcdctmgr.c quantize_trellis, driven by jccoefct.c compress_trellis_pass and the pass sequence of
   jcmaster.c) [UPSTREAM-RECALL of mozjpeg 4.1.x as pinned by mozjpeg-sys 2.2.1, /root/reference/Cargo.lock:1035-1044; UNPINNED].
   Profile constants (jcparam.c, JCP_MAX_COMPRESSION): trellis_quant, trellis_quant_dc on; trellis_eob_opt, trellis_q_opt,
   use_scans_in_trellis off; trellis_num_loops 1; lambda_log_scale1 14.75, lambda_log_scale2 16.5; trellis_delta_dc_weight 0.
   What the pass sequence amounts to (optimize_coding on): per component, in order,
     (1) a statistics pass over its scalar-quantised coefficients with the entropy coder of the output mode -- progressive: ONE scan
         of the component alone, Ss 1, Se 63, Ah = Al = 0 (EOBRUN symbols included), which yields the optimal AC table; sequential
         (--jpeg-baseline): a one-component sequential scan, which yields optimal DC and AC tables;
     (2) the trellis pass: every row of blocks is re-quantised from the unquantised DCT with that table's code lengths as rates.  The DC
         table in progressive mode is still the Annex K table jpeg_set_defaults installed (no DC statistics exist yet).
   Inside quantize_trellis (`mode = 1` is hard-wired there, so the CSF weight table is overridden by 1 / q^2 and lambda_base is 1):
     norm   = mean of the block's 63 squared AC DCT values (float accumulation, natural order)
     lambda = 2^14.75 / (2^16.5 + norm)                                      (double arithmetic, stored as float)
     AC:  dynamic programme over zig-zag positions; a non-zero scalar level v offers the candidates 1, 3, 7, ... (2^k - 1 < v) and v;
          cost = bits of the (run, size) symbol + size + ZRL bits + squared error * lambda / q^2 (+ the squared values of the zeros skipped);
          then the cheapest last coefficient (an EOB costs the length of symbol 0x00), everything behind it zero
     DC:  per row of blocks a Viterbi path over up to 9 levels around the rounded one (min(9, (2 + 60 / q_dc) | 1)), rate = size +
          the DC code of the difference to the previous block's candidate; the predictor of a row's first block is the last DC of
          the row before it inside the same iMCU row, 0 for the first row of an iMCU row
   All cost arithmetic is float in the order written below (gcc, x86-64, no contraction). */
#define TRELLIS_MAX_COEF_BITS 10
#define TRELLIS_DC_MAX_CAND 9
static const uint8_t STD_DC_LEN[2][12] = {{2, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9}, {2, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}};   /* T.81 Tables K.3, K.4 */
#define TRELLIS_LAMBDA_C1 0x1.ae89f995ad3adp+14   /* pow(2.0, 14.75) */
#define TRELLIS_LAMBDA_C2 0x1.6a09e667f3bcdp+16   /* pow(2.0, 16.5) */

static void quantize_trellis_row(const uint8_t dclen[17], const uint8_t aclen[256], int16_t *coef_blocks, const int16_t *src, int num_blocks,
                                 const uint16_t *qt /* natural order */, int trellis_dc, int16_t *last_dc_val) {
    float accumulated_zero_dist[64], accumulated_cost[64];
    int run_start[64] = {0};
    float lambda_table[64];
    const int Ss = 1, Se = 63;
    int dcn = 2 + 60 / qt[0]; dcn |= 1; if (dcn > TRELLIS_DC_MAX_CAND) dcn = TRELLIS_DC_MAX_CAND;
    const int dc_trellis_candidates = dcn;
    float *accumulated_dc_cost[TRELLIS_DC_MAX_CAND]; int *dc_cost_backtrack[TRELLIS_DC_MAX_CAND]; int16_t *dc_candidate[TRELLIS_DC_MAX_CAND];
    for (int i = 0; i < TRELLIS_DC_MAX_CAND; i++) {
        accumulated_dc_cost[i] = (float *)malloc(sizeof(float) * (size_t)num_blocks);
        dc_cost_backtrack[i] = (int *)malloc(sizeof(int) * (size_t)num_blocks);
        dc_candidate[i] = (int16_t *)malloc(sizeof(int16_t) * (size_t)num_blocks);
    }
    for (int i = 0; i < 64; i++) lambda_table[i] = (float)(1.0 / (double)((int)qt[i] * (int)qt[i]));
    for (int bi = 0; bi < num_blocks; bi++) {
        const int16_t *sb = src + (size_t)bi * 64;
        int16_t *cb = coef_blocks + (size_t)bi * 64;
        float norm = 0.0f;
        for (int i = 1; i < 64; i++) norm += (float)((int)sb[i] * (int)sb[i]);
        norm = (float)((double)norm / 63.0);
        const float lambda = (float)(TRELLIS_LAMBDA_C1 * 1.0 / (TRELLIS_LAMBDA_C2 + (double)norm));
        const float lambda_dc = lambda * lambda_table[0];
        accumulated_zero_dist[Ss - 1] = 0.0f;
        accumulated_cost[Ss - 1] = 0.0f;
        if (trellis_dc) {
            const int sign = sb[0] >> 31, x = abs(sb[0]), q = 8 * qt[0];
            const int qval = (x + q / 2) / q;
            for (int k = 0; k < dc_trellis_candidates; k++) {
                int cand = qval - dc_trellis_candidates / 2 + k;
                if (cand >= (1 << TRELLIS_MAX_COEF_BITS)) cand = (1 << TRELLIS_MAX_COEF_BITS) - 1;
                if (cand <= -(1 << TRELLIS_MAX_COEF_BITS)) cand = -(1 << TRELLIS_MAX_COEF_BITS) + 1;
                const int delta = cand * q - x;
                const float dc_candidate_dist = (float)(delta * delta) * lambda_dc;
                cand *= 1 + 2 * sign;
                dc_candidate[k][bi] = (int16_t)cand;
                if (bi == 0) {
                    int dc_delta = abs(cand - *last_dc_val), bits = 0;
                    while (dc_delta) { dc_delta >>= 1; bits++; }
                    const float cost = (float)(bits + dclen[bits]) + dc_candidate_dist;
                    accumulated_dc_cost[k][0] = cost;
                    dc_cost_backtrack[k][0] = -1;
                } else {
                    for (int l = 0; l < dc_trellis_candidates; l++) {
                        int dc_delta = abs(cand - dc_candidate[l][bi - 1]), bits = 0;
                        while (dc_delta) { dc_delta >>= 1; bits++; }
                        const float cost = (float)(bits + dclen[bits]) + dc_candidate_dist + accumulated_dc_cost[l][bi - 1];
                        if (l == 0 || cost < accumulated_dc_cost[k][bi]) { accumulated_dc_cost[k][bi] = cost; dc_cost_backtrack[k][bi] = l; }
                    }
                }
            }
        }
        for (int i = Ss; i <= Se; i++) {
            const int z = ZZ[i];
            const int sign = sb[z] >> 31, x = abs(sb[z]), q = 8 * qt[z];
            int candidate[16], candidate_bits[16];
            float candidate_dist[16];
            accumulated_zero_dist[i] = (float)(x * x) * lambda * lambda_table[z] + accumulated_zero_dist[i - 1];
            int qval = (x + q / 2) / q;
            if (qval == 0) { cb[z] = 0; accumulated_cost[i] = 1e38f; continue; }
            if (qval >= (1 << TRELLIS_MAX_COEF_BITS)) qval = (1 << TRELLIS_MAX_COEF_BITS) - 1;
            const int num_candidates = bitlen((unsigned)qval);
            for (int k = 0; k < num_candidates; k++) {
                candidate[k] = (k < num_candidates - 1) ? (2 << k) - 1 : qval;
                const int delta = candidate[k] * q - x;
                candidate_bits[k] = k + 1;
                candidate_dist[k] = (float)(delta * delta) * lambda * lambda_table[z];
            }
            accumulated_cost[i] = 1e38f;
            for (int j = Ss - 1; j < i; j++) {
                if (j != Ss - 1 && cb[ZZ[j]] == 0) continue;
                int zero_run = i - 1 - j;
                if ((zero_run >> 4) && aclen[0xF0] == 0) continue;
                const int run_bits = (zero_run >> 4) * aclen[0xF0];
                zero_run &= 15;
                for (int k = 0; k < num_candidates; k++) {
                    const int coef_bits = aclen[16 * zero_run + candidate_bits[k]];
                    if (coef_bits == 0) continue;
                    const int rate = coef_bits + candidate_bits[k] + run_bits;
                    float cost = (float)rate + candidate_dist[k];
                    cost += accumulated_zero_dist[i - 1] - accumulated_zero_dist[j] + accumulated_cost[j];
                    if (cost < accumulated_cost[i]) {
                        cb[z] = (int16_t)((candidate[k] ^ sign) - sign);
                        accumulated_cost[i] = cost;
                        run_start[i] = j;
                    }
                }
            }
        }
        int last_coeff_idx = Ss - 1;
        float best_cost = accumulated_zero_dist[Se] + (float)aclen[0];
        for (int i = Ss; i <= Se; i++) {
            if (cb[ZZ[i]] != 0) {
                float cost = accumulated_cost[i] + accumulated_zero_dist[Se] - accumulated_zero_dist[i];
                if (i < Se) cost += (float)aclen[0];
                if (cost < best_cost) { best_cost = cost; last_coeff_idx = i; }
            }
        }
        /* zero out coefficients that are part of runs */
        for (int i = Se; i >= Ss;) {
            while (i > last_coeff_idx) { cb[ZZ[i]] = 0; i--; }
            last_coeff_idx = run_start[i];
            i--;
        }
    }
    if (trellis_dc) {
        int j = 0;
        for (int i = 1; i < dc_trellis_candidates; i++)
            if (accumulated_dc_cost[i][num_blocks - 1] < accumulated_dc_cost[j][num_blocks - 1]) j = i;
        for (int bi = num_blocks - 1; bi >= 0; bi--) { coef_blocks[(size_t)bi * 64] = dc_candidate[j][bi]; j = dc_cost_backtrack[j][bi]; }
        *last_dc_val = coef_blocks[(size_t)(num_blocks - 1) * 64];
    }
    for (int i = 0; i < TRELLIS_DC_MAX_CAND; i++) { free(accumulated_dc_cost[i]); free(dc_cost_backtrack[i]); free(dc_candidate[i]); }
}

/* code lengths (0 = symbol unused) of the optimal table for these counts */
static void optimal_lengths(const long freq[257], uint8_t len[256]) {
    ehuff h; memset(&h, 0, sizeof h);
    h.nsym = cso_gen_optimal_table(freq, h.bits, h.huffval);
    derive_ehuff(&h);
    memcpy(len, h.size, 256);
}
/* the rate tables one component's trellis pass works with (exported for the stage-level device tests): aclen[256], dclen[17] */
void cso_trellis_tables(const cso_image *im, int ci, uint8_t aclen[256], uint8_t dclen[17]) {
    cso_scan sc; memset(&sc, 0, sizeof sc);
    sc.ncomp_in_scan = 1; sc.comp_idx[0] = ci; sc.Ss = im->progressive ? 1 : 0; sc.Se = 63;
    tvec tv = {0};
    tokenize_scan(im, &sc, &tv);
    long freq[8][257]; memset(freq, 0, sizeof freq);
    for (size_t i = 0; i < tv.n; i++) if (tv.t[i].tbl != 255) freq[tv.t[i].tbl][tv.t[i].sym]++;
    free(tv.t);
    const int id = ci ? 1 : 0;
    optimal_lengths(freq[4 + id], aclen);
    memset(dclen, 0, 17);
    if (im->progressive) memcpy(dclen, STD_DC_LEN[id], 12);
    else { uint8_t l[256]; optimal_lengths(freq[id], l); memcpy(dclen, l, 17); }
}
static void trellis_image(cso_image *im, int16_t *const raw[], const cso_enc_params *p) {
    (void)p;
    for (int ci = 0; ci < im->ncomp; ci++) {
        cso_comp *k = &im->comp[ci];
        uint8_t aclen[256], dclen[17];
        cso_trellis_tables(im, ci, aclen, dclen);
        int16_t last_dc = 0;
        for (int by = 0; by < k->real_bh; by++) {
            if (by % k->v == 0) last_dc = 0;   /* compress_trellis_pass: the predictor starts at 0 in every iMCU row */
            quantize_trellis_row(dclen, aclen, k->coef + (size_t)by * k->bw * 64, raw[ci] + (size_t)by * k->bw * 64, k->real_bw, im->qt[k->tq], 1, &last_dc);
        }
        make_dummy_blocks(im, ci);
    }
}

/* stock progression scripts: libjpeg jcparam.c jpeg_simple_progression behaviour;
   which=1: the 8-scan script read out of /root/reference/samples/j0.JPG */
int cso_stock_script(int ncomp, int which, cso_scan *o) {
    int n = 0;
#define SCAN1(c, ss, se, ah, al) do { o[n].ncomp_in_scan = 1; o[n].comp_idx[0] = c; o[n].Ss = ss; o[n].Se = se; o[n].Ah = ah; o[n].Al = al; n++; } while (0)
#define SCANDC(ah, al) do { o[n].ncomp_in_scan = ncomp; for (int c_ = 0; c_ < ncomp; c_++) o[n].comp_idx[c_] = c_; o[n].Ss = 0; o[n].Se = 0; o[n].Ah = ah; o[n].Al = al; n++; } while (0)
    if (ncomp == 3 && which == 1) {
        SCANDC(0, 0); SCAN1(0, 1, 2, 0, 1); SCAN1(0, 3, 63, 0, 1); SCAN1(1, 1, 63, 0, 1); SCAN1(2, 1, 63, 0, 1);
        SCAN1(0, 1, 63, 1, 0); SCAN1(1, 1, 63, 1, 0); SCAN1(2, 1, 63, 1, 0);
    } else if (ncomp == 3) {
        SCANDC(0, 1); SCAN1(0, 1, 5, 0, 2); SCAN1(2, 1, 63, 0, 1); SCAN1(1, 1, 63, 0, 1); SCAN1(0, 6, 63, 0, 2);
        SCAN1(0, 1, 63, 2, 1); SCANDC(1, 0); SCAN1(2, 1, 63, 1, 0); SCAN1(1, 1, 63, 1, 0); SCAN1(0, 1, 63, 1, 0);
    } else {
        /* generic: per component, as jpeg_simple_progression does for non-YCbCr */
        if (ncomp == 1) { SCANDC(0, 1); SCAN1(0, 1, 5, 0, 2); SCAN1(0, 6, 63, 0, 2); SCAN1(0, 1, 63, 2, 1); SCANDC(1, 0); SCAN1(0, 1, 63, 1, 0); }
        else {
            SCANDC(0, 1);
            for (int c = 0; c < ncomp; c++) SCAN1(c, 1, 5, 0, 2);
            for (int c = 0; c < ncomp; c++) SCAN1(c, 6, 63, 0, 2);
            for (int c = 0; c < ncomp; c++) SCAN1(c, 1, 63, 2, 1);
            SCANDC(1, 0);
            for (int c = 0; c < ncomp; c++) SCAN1(c, 1, 63, 1, 0);
        }
    }
    return n;
}

static void put_marker(bvec *b, int m) { bv_put(b, 0xFF); bv_put(b, m); }

static void write_dqt(bvec *b, const cso_image *im, int style) {
    int seen[4] = {0}, ids[4], n = 0;
    for (int c = 0; c < im->ncomp; c++) if (!seen[im->comp[c].tq]) { seen[im->comp[c].tq] = 1; ids[n++] = im->comp[c].tq; }
    int prec[4];
    for (int i = 0; i < n; i++) { prec[i] = 0; for (int k = 0; k < 64; k++) if (im->qt[ids[i]][k] > 255) prec[i] = 1; }
    if (style == 1) { /* mozjpeg: one DQT segment holding every table */
        int len = 2; for (int i = 0; i < n; i++) len += 1 + (prec[i] ? 128 : 64);
        put_marker(b, 0xDB); bv_put2(b, len);
        for (int i = 0; i < n; i++) { bv_put(b, (prec[i] << 4) | ids[i]); for (int k = 0; k < 64; k++) { int v = im->qt[ids[i]][ZZ[k]]; if (prec[i]) bv_put(b, v >> 8); bv_put(b, v & 255); } }
    } else {
        for (int i = 0; i < n; i++) {
            put_marker(b, 0xDB); bv_put2(b, 2 + 1 + (prec[i] ? 128 : 64));
            bv_put(b, (prec[i] << 4) | ids[i]); for (int k = 0; k < 64; k++) { int v = im->qt[ids[i]][ZZ[k]]; if (prec[i]) bv_put(b, v >> 8); bv_put(b, v & 255); }
        }
    }
}
static void write_dht_group(bvec *b, ehuff *const tabs[], const int cls_id[], int n, int style) {
    if (n == 0) return;
    if (style == 1) {
        int len = 2; for (int i = 0; i < n; i++) len += 17 + tabs[i]->nsym;
        put_marker(b, 0xC4); bv_put2(b, len);
        for (int i = 0; i < n; i++) { bv_put(b, cls_id[i]); bv_write(b, tabs[i]->bits + 1, 16); bv_write(b, tabs[i]->huffval, tabs[i]->nsym); }
    } else {
        for (int i = 0; i < n; i++) { put_marker(b, 0xC4); bv_put2(b, 2 + 17 + tabs[i]->nsym); bv_put(b, cls_id[i]); bv_write(b, tabs[i]->bits + 1, 16); bv_write(b, tabs[i]->huffval, tabs[i]->nsym); }
    }
}

static int cso_search_progression(const cso_image *hdr, const cso_enc_params *p, cso_scan *out);
/* one scan as it goes into the file: DHT group, SOS, entropy-coded data (optimal tables, byte stuffing, final byte padded with 1-bits).
   mozjpeg's scan search measures exactly these bytes per candidate (jcmaster.c: every candidate scan is written, header included, into
   its own memory buffer; scan_size[] is that buffer's length) */
static void encode_one_scan(bvec *bp, const cso_image *hdr, const cso_enc_params *p, const cso_scan *sc, tvec *tvp) {
    bvec b = *bp;
    tvec tv = *tvp;
    const cso_image *im = hdr;
    {
        tv.n = 0;
        tokenize_scan(hdr, sc, &tv);
        long freq[8][257]; memset(freq, 0, sizeof freq);
        for (size_t i = 0; i < tv.n; i++) if (tv.t[i].tbl != 255) freq[tv.t[i].tbl][tv.t[i].sym]++;
        ehuff tabs[8]; memset(tabs, 0, sizeof tabs);
        ehuff *grp[8]; int cls[8], ng = 0;
        /* table emission order: per scan component, DC then AC (libjpeg write_scan_header) */
        for (int i = 0; i < sc->ncomp_in_scan; i++) {
            int id = sc->comp_idx[i] ? 1 : 0;
            int want_dc = p->progressive ? (sc->Ss == 0 && sc->Ah == 0) : 1;
            int want_ac = p->progressive ? (sc->Ss != 0) : 1;
            if (want_dc && !tabs[id].used) { tabs[id].used = 1; tabs[id].nsym = cso_gen_optimal_table(freq[id], tabs[id].bits, tabs[id].huffval); derive_ehuff(&tabs[id]); grp[ng] = &tabs[id]; cls[ng++] = id; }
            if (want_ac && !tabs[4 + id].used) { tabs[4 + id].used = 1; tabs[4 + id].nsym = cso_gen_optimal_table(freq[4 + id], tabs[4 + id].bits, tabs[4 + id].huffval); derive_ehuff(&tabs[4 + id]); grp[ng] = &tabs[4 + id]; cls[ng++] = 0x10 | id; }
        }
        write_dht_group(&b, grp, cls, ng, p->marker_style);
        put_marker(&b, 0xDA); bv_put2(&b, 6 + 2 * sc->ncomp_in_scan); bv_put(&b, sc->ncomp_in_scan);
        for (int i = 0; i < sc->ncomp_in_scan; i++) {
            int id = sc->comp_idx[i] ? 1 : 0, td = id, ta = id;
            if (p->progressive) { if (sc->Ss == 0) { ta = 0; if (sc->Ah != 0) td = 0; } else td = 0; }
            bv_put(&b, im->comp[sc->comp_idx[i]].id); bv_put(&b, (td << 4) | ta);
        }
        bv_put(&b, sc->Ss); bv_put(&b, sc->Se); bv_put(&b, (sc->Ah << 4) | sc->Al);
        /* bit packing, MSB first, 0xFF -> 0xFF00, pad with 1-bits (T.81 F.1.2.3) */
        uint64_t acc = 0; int nb = 0;
        for (size_t i = 0; i < tv.n; i++) {
            const token *t = &tv.t[i];
            if (t->tbl != 255) { const ehuff *h = &tabs[t->tbl]; acc = (acc << h->size[t->sym]) | h->code[t->sym]; nb += h->size[t->sym]; }
            if (t->nbits) { acc = (acc << t->nbits) | t->bits; nb += t->nbits; }
            while (nb >= 8) { int c = (int)((acc >> (nb - 8)) & 255); bv_put(&b, c); if (c == 255) bv_put(&b, 0); nb -= 8; }
        }
        if (nb > 0) { int c = (int)(((acc << (8 - nb)) | ((1u << (8 - nb)) - 1)) & 255); bv_put(&b, c); if (c == 255) bv_put(&b, 0); }
    }
    *bp = b; *tvp = tv;
}

int cso_encode(const cso_image *im, const cso_enc_params *p, const cso_scan *script, int nscans, uint8_t **out, size_t *out_len) {
    cso_scan local[CSO_MAX_SCANS];
    cso_image hdr = *im; /* shallow: only flags differ */
    hdr.progressive = p->progressive;
    if (!script || !nscans) {
        if (p->progressive) { nscans = cso_stock_script(im->ncomp, p->scan_script, local); script = local; }
        else { local[0].ncomp_in_scan = im->ncomp; for (int c = 0; c < im->ncomp; c++) local[0].comp_idx[c] = c; local[0].Ss = 0; local[0].Se = 63; local[0].Ah = local[0].Al = 0; nscans = 1; script = local; }
    }
    bvec b = {0};
    put_marker(&b, 0xD8);
    /* libjpeg write_file_header: JFIF APP0 for YCbCr / grayscale */
    if (im->ncomp == 1 || im->ncomp == 3) {
        static const uint8_t jfif[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
        bv_write(&b, jfif, sizeof jfif);
    }
    /* metadata carry-over: APPn/COM when keep_metadata; ICC (APP2 "ICC_PROFILE\0") governed separately by preserve_icc */
    for (size_t mo = 0; mo + 4 <= im->meta_len;) {
        size_t L = ((size_t)im->meta[mo + 2] << 8) | im->meta[mo + 3];
        int is_icc = im->meta[mo + 1] == 0xE2 && L >= 14 && !memcmp(im->meta + mo + 4, "ICC_PROFILE\0", 12);
        int keep = is_icc ? p->preserve_icc : p->keep_metadata;
        if (keep) bv_write(&b, im->meta + mo, 2 + L);
        mo += 2 + L;
    }
    write_dqt(&b, im, p->marker_style);
    int is_baseline = !p->progressive;
    for (int c = 0; c < im->ncomp; c++) for (int k = 0; k < 64; k++) if (im->qt[im->comp[c].tq][k] > 255) is_baseline = 0;
    put_marker(&b, p->progressive ? 0xC2 : (is_baseline ? 0xC0 : 0xC1));
    bv_put2(&b, 8 + 3 * im->ncomp); bv_put(&b, 8); bv_put2(&b, im->height); bv_put2(&b, im->width); bv_put(&b, im->ncomp);
    for (int c = 0; c < im->ncomp; c++) { bv_put(&b, im->comp[c].id); bv_put(&b, (im->comp[c].h << 4) | im->comp[c].v); bv_put(&b, im->comp[c].tq); }

    cso_scan chosen[CSO_MAX_SCANS];
    if (p->progressive && p->scan_script == 2 && (script == local)) { nscans = cso_search_progression(&hdr, p, chosen); script = chosen; }
    tvec tv = {0};
    for (int s = 0; s < nscans; s++) encode_one_scan(&b, &hdr, p, &script[s], &tv);
    free(tv.t);
    put_marker(&b, 0xD9);
    *out = b.p; *out_len = b.n;
    return 0;
}

/* ---- mozjpeg's scan-script search (optimize_scans; reached through jpeg_simple_progression under the JCP_MAX_COMPRESSION profile,
   which libcaesium uses on both its lossy and its lossless JPEG path: /root/reference/src/compressor.rs:415,434 -> Cargo.lock:1035-1044).
   [UPSTREAM-RECALL] of jcparam.c jpeg_search_progression (the candidate list) and jcmaster.c select_scans (the decisions), with
   dc_scan_opt_mode = 0 (one DC scan for all components: what the reference's own output samples/j0.JPG shows).  PINNED by that file:
   run on j0's coefficients it must come back with j0's 8-scan script (tests/test_oracle_jpeg.py).
   Candidates (index: scan), YCbCr -- grey images have the first 23 only:
     0 DC of every component | 1, 2 Y 1-8 / 9-63 at Al 0 | for Al = 0, 1, 2: 3+3Al Y 1-63 refinement Ah=Al+1 -> Al, then Y 1-8 / 9-63 at Al+1
     12 Y 1-63 | 13.. five splits {2, 8, 5, 12, 18}: Y 1-s, Y s+1-63 (12..22 at the Al chosen for luma)
     23-25 chroma DC variants (written by mozjpeg, never used in this mode) | 26-29 Cb 1-8, 9-63, Cr 1-8, 9-63 at Al 0
     for Al = 0, 1: 30+6Al Cb, Cr refinements, then the four band scans at Al+1 | 42, 43 Cb, Cr 1-63 | 44.. five splits x (Cb lo, Cb hi, Cr lo, Cr hi)
   A candidate's cost is the size of everything it puts into the file: DHT, SOS and the stuffed entropy-coded bytes. */
static size_t scan_cost(const cso_image *hdr, const cso_enc_params *p, const cso_scan *sc, tvec *tv) {
    bvec b = {0};
    encode_one_scan(&b, hdr, p, sc, tv);
    size_t n = b.n;
    free(b.p);
    return n;
}
static int cso_search_progression(const cso_image *hdr, const cso_enc_params *p, cso_scan *out) {
    static const int split[5] = {2, 8, 5, 12, 18};
    const int ncomp = hdr->ncomp;
    if (ncomp != 1 && ncomp != 3) return cso_stock_script(ncomp, 0, out);
    cso_scan L[64];
    int n = 0;
#define S1(c, ss, se, ah, al) do { L[n].ncomp_in_scan = 1; L[n].comp_idx[0] = c; L[n].Ss = ss; L[n].Se = se; L[n].Ah = ah; L[n].Al = al; n++; } while (0)
    L[n].ncomp_in_scan = ncomp; for (int c = 0; c < ncomp; c++) L[n].comp_idx[c] = c; L[n].Ss = L[n].Se = L[n].Ah = L[n].Al = 0; n++;
    S1(0, 1, 8, 0, 0); S1(0, 9, 63, 0, 0);
    for (int Al = 0; Al < 3; Al++) { S1(0, 1, 63, Al + 1, Al); S1(0, 1, 8, 0, Al + 1); S1(0, 9, 63, 0, Al + 1); }
    S1(0, 1, 63, 0, 0);
    for (int i = 0; i < 5; i++) { S1(0, 1, split[i], 0, 0); S1(0, split[i] + 1, 63, 0, 0); }
    const int nluma = n;   /* 23 */
    if (ncomp == 3) {
        L[n].ncomp_in_scan = 2; L[n].comp_idx[0] = 1; L[n].comp_idx[1] = 2; L[n].Ss = L[n].Se = L[n].Ah = L[n].Al = 0; n++;
        S1(1, 0, 0, 0, 0); S1(2, 0, 0, 0, 0);
        S1(1, 1, 8, 0, 0); S1(1, 9, 63, 0, 0); S1(2, 1, 8, 0, 0); S1(2, 9, 63, 0, 0);
        for (int Al = 0; Al < 2; Al++) { S1(1, 1, 63, Al + 1, Al); S1(2, 1, 63, Al + 1, Al); S1(1, 1, 8, 0, Al + 1); S1(1, 9, 63, 0, Al + 1); S1(2, 1, 8, 0, Al + 1); S1(2, 9, 63, 0, Al + 1); }
        S1(1, 1, 63, 0, 0); S1(2, 1, 63, 0, 0);
        for (int i = 0; i < 5; i++) { S1(1, 1, split[i], 0, 0); S1(1, split[i] + 1, 63, 0, 0); S1(2, 1, split[i], 0, 0); S1(2, split[i] + 1, 63, 0, 0); }
    }
#undef S1
    const int nscans = n, luma_split0 = 12, chroma_dc = 3, chroma_base = nluma + chroma_dc, chroma_split0 = nluma + chroma_dc + 16;
    size_t size[64];
    memset(size, 0, sizeof size);
    tvec tv = {0};
    int best_Al_luma = 0, best_Al_chroma = 0, best_split_luma = 0, best_split_chroma = 0;
    size_t best_cost = 0;
    for (int sn = 0; sn < nscans;) {
        cso_scan sc = L[sn];
        if (sn >= luma_split0 && sn < nluma) sc.Al = best_Al_luma;                 /* the frequency-split candidates are coded at the Al chosen before */
        if (sn >= chroma_split0) sc.Al = best_Al_chroma;
        L[sn] = sc;
        if (!(sn >= nluma && sn < chroma_base)) size[sn] = scan_cost(hdr, p, &sc, &tv);   /* the chroma DC variants play no part with one DC scan for all */
        const int next = sn + 1;   /* scans done */
        int jump = -1;
        if (next > 1 && next <= luma_split0) {
            if ((next - 1) % 3 == 2) {
                const int Al = (next - 1) / 3;
                size_t cost = size[next - 2] + size[next - 1];
                for (int i = 0; i < Al; i++) cost += size[3 + 3 * i];
                if (Al == 0 || cost < best_cost) { best_cost = cost; best_Al_luma = Al; }
                else jump = luma_split0;
            }
        } else if (next > luma_split0 && next <= nluma) {
            if (next == luma_split0 + 1) { best_split_luma = 0; best_cost = size[next - 1]; }
            else if ((next - luma_split0) % 2 == 1) {
                const int idx = (next - luma_split0) >> 1;
                const size_t cost = size[next - 2] + size[next - 1];
                if (cost < best_cost) { best_cost = cost; best_split_luma = idx; }
                if ((idx == 2 && best_split_luma == 0) || (idx == 3 && best_split_luma != 2) || (idx == 4 && best_split_luma != 4)) jump = nluma;
            }
        } else if (nscans > nluma) {
            if (next > chroma_base && next <= chroma_split0) {
                if ((next - chroma_base) % 6 == 4) {
                    const int Al = (next - chroma_base) / 6;
                    size_t cost = size[next - 4] + size[next - 3] + size[next - 2] + size[next - 1];
                    for (int i = 0; i < Al; i++) cost += size[chroma_base + 4 + 6 * i] + size[chroma_base + 5 + 6 * i];
                    if (Al == 0 || cost < best_cost) { best_cost = cost; best_Al_chroma = Al; }
                    else jump = chroma_split0;
                }
            } else if (next > chroma_split0 && next <= nscans) {
                if (next == chroma_split0 + 2) { best_split_chroma = 0; best_cost = size[next - 2] + size[next - 1]; }
                else if ((next - chroma_split0) % 4 == 2) {
                    const int idx = (next - chroma_split0) >> 2;
                    const size_t cost = size[next - 4] + size[next - 3] + size[next - 2] + size[next - 1];
                    if (cost < best_cost) { best_cost = cost; best_split_chroma = idx; }
                    if ((idx == 2 && best_split_chroma == 0) || (idx == 3 && best_split_chroma != 2) || (idx == 4 && best_split_chroma != 4)) jump = nscans;
                }
            }
        }
        sn = jump >= 0 ? jump : sn + 1;
    }
    free(tv.t);
    /* the file: DC, luma bands, luma refinements down to the Al both share, chroma bands, chroma refinements down to it, then the
       shared refinements luma first */
    int m = 0;
    const int min_Al = ncomp == 3 ? (best_Al_luma < best_Al_chroma ? best_Al_luma : best_Al_chroma) : best_Al_luma;
    out[m++] = L[0];
    if (best_split_luma == 0) out[m++] = L[luma_split0];
    else { out[m++] = L[luma_split0 + 2 * (best_split_luma - 1) + 1]; out[m++] = L[luma_split0 + 2 * (best_split_luma - 1) + 2]; }
    for (int Al = best_Al_luma - 1; Al >= min_Al; Al--) out[m++] = L[3 + 3 * Al];
    if (ncomp == 3) {
        if (best_split_chroma == 0) { out[m++] = L[chroma_split0]; out[m++] = L[chroma_split0 + 1]; }
        else for (int i = 2; i <= 5; i++) out[m++] = L[chroma_split0 + 4 * (best_split_chroma - 1) + i];
        for (int Al = best_Al_chroma - 1; Al >= min_Al; Al--) { out[m++] = L[chroma_base + 6 * Al + 4]; out[m++] = L[chroma_base + 6 * Al + 5]; }
    }
    for (int Al = min_Al - 1; Al >= 0; Al--) {
        out[m++] = L[3 + 3 * Al];
        if (ncomp == 3) { out[m++] = L[chroma_base + 6 * Al + 4]; out[m++] = L[chroma_base + 6 * Al + 5]; }
    }
    return m;
}
/* the script the search picks for an image's coefficients (tests; the device's choice is compared with it) */
int cso_search_script(const cso_image *im, const cso_enc_params *p, cso_scan *out) {
    cso_image hdr = *im;
    hdr.progressive = 1;
    return cso_search_progression(&hdr, p, out);
}

int cso_jpeg_compress(const uint8_t *in, size_t n, const cso_enc_params *p, int lossless, uint8_t **out, size_t *out_len) {
    cso_image *src = NULL, *dst = NULL;
    if (cso_decode(in, n, &src)) return -1;
    int rc = -1;
    if (lossless) { rc = cso_encode(src, p, NULL, 0, out, out_len); cso_image_free(src); return rc; }
    if (src->ncomp != 1 && src->ncomp != 3) { cso_image_free(src); FAIL("unsupported component count %d", src->ncomp); }
    uint8_t *pix = (uint8_t *)malloc((size_t)src->width * src->height * src->ncomp);
    if (cso_decode_pixels(src, pix) == 0 && cso_forward(pix, src->width, src->height, src->ncomp, p, NULL, &dst) == 0) {
        dst->meta = src->meta; dst->meta_len = src->meta_len;
        rc = cso_encode(dst, p, NULL, 0, out, out_len);
        dst->meta = NULL; dst->meta_len = 0;
    }
    free(pix); cso_image_free(src); cso_image_free(dst);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* resize path -- see jpeg_oracle.h.  Compile with -ffp-contract=off: image-rs does not fuse. */
void cso_compute_dimensions(int ow, int oh, int dw, int dh, int *nw, int *nh) {
    if (dw > 0 && dh > 0) { *nw = dw; *nh = dh; return; }
    float ratio = (float)ow / (float)oh;
    if (dw > 0) { *nw = dw; *nh = (int)roundf((float)dw / ratio); }
    else if (dh > 0) { *nh = dh; *nw = (int)roundf((float)dh * ratio); }
    else { *nw = ow; *nh = oh; }
    if (*nw < 1) *nw = 1;
    if (*nh < 1) *nh = 1;
}
static float sincf_(float t) { float a = t * 3.14159265358979323846f; return t == 0.0f ? 1.0f : sinf(a) / a; }
static float lanczos3f(float x) { return fabsf(x) < 3.0f ? sincf_(x) * sincf_(x / 3.0f) : 0.0f; }
/* weights of one output coordinate (image-rs horizontal_sample / vertical_sample) */
static int lanczos_taps(int in_size, int out_size, int o, int *left_out, float *ws) {
    float ratio = (float)in_size / (float)out_size;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float support = 3.0f * sratio;
    float center = ((float)o + 0.5f) * ratio;
    long left = (long)floorf(center - support); if (left < 0) left = 0; if (left > in_size - 1) left = in_size - 1;
    long right = (long)ceilf(center + support); if (right < left + 1) right = left + 1; if (right > in_size) right = in_size;
    center = center - 0.5f;
    float sum = 0.0f; int n = 0;
    for (long i = left; i < right; i++) { float w = lanczos3f(((float)i - center) / sratio); ws[n++] = w; sum += w; }
    for (int i = 0; i < n; i++) ws[i] /= sum;
    *left_out = (int)left;
    return n;
}
void cso_lanczos3_resize(const uint8_t *src, int w, int h, int nch, int nw, int nh, uint8_t *dst) {
    if (nw == w && nh == h) { memcpy(dst, src, (size_t)w * h * nch); return; }
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * nh * nch);
    float *ws = (float *)malloc(sizeof(float) * (size_t)((h > w ? h : w) + 8));
    for (int oy = 0; oy < nh; oy++) {               /* vertical pass -> f32 */
        int left, n = lanczos_taps(h, nh, oy, &left, ws);
        for (int x = 0; x < w; x++)
            for (int c = 0; c < nch; c++) {
                float t = 0.0f;
                for (int i = 0; i < n; i++) t += (float)src[((size_t)(left + i) * w + x) * nch + c] * ws[i];
                tmp[((size_t)oy * w + x) * nch + c] = t;
            }
    }
    for (int ox = 0; ox < nw; ox++) {               /* horizontal pass -> u8 */
        int left, n = lanczos_taps(w, nw, ox, &left, ws);
        for (int y = 0; y < nh; y++)
            for (int c = 0; c < nch; c++) {
                float t = 0.0f;
                for (int i = 0; i < n; i++) t += tmp[((size_t)y * w + left + i) * nch + c] * ws[i];
                t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
                dst[((size_t)y * nw + ox) * nch + c] = (uint8_t)roundf(t);
            }
    }
    free(tmp); free(ws);
}
/* the same resample over 16-bit samples (image-rs keeps L16 / La16 / Rgb16 / Rgba16 images at 16 bits: clamp to 0..65535) */
void cso_lanczos3_resize16(const uint16_t *src, int w, int h, int nch, int nw, int nh, uint16_t *dst) {
    if (nw == w && nh == h) { memcpy(dst, src, (size_t)w * h * nch * 2); return; }
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * nh * nch);
    float *ws = (float *)malloc(sizeof(float) * (size_t)((h > w ? h : w) + 8));
    for (int oy = 0; oy < nh; oy++) {
        int left, n = lanczos_taps(h, nh, oy, &left, ws);
        for (int x = 0; x < w; x++)
            for (int c = 0; c < nch; c++) {
                float t = 0.0f;
                for (int i = 0; i < n; i++) t += (float)src[((size_t)(left + i) * w + x) * nch + c] * ws[i];
                tmp[((size_t)oy * w + x) * nch + c] = t;
            }
    }
    for (int ox = 0; ox < nw; ox++) {
        int left, n = lanczos_taps(w, nw, ox, &left, ws);
        for (int y = 0; y < nh; y++)
            for (int c = 0; c < nch; c++) {
                float t = 0.0f;
                for (int i = 0; i < n; i++) t += tmp[((size_t)y * w + left + i) * nch + c] * ws[i];
                t = t < 0.0f ? 0.0f : (t > 65535.0f ? 65535.0f : t);
                dst[((size_t)y * nw + ox) * nch + c] = (uint16_t)roundf(t);
            }
    }
    free(tmp); free(ws);
}
/* libjpeg jdcolor.c ycc_rgb_convert: SCALEBITS 16, Cr=>R 1.40200, Cb=>B 1.77200, Cr=>G -0.71414, Cb=>G -0.34414 */
void cso_ycc_to_rgb(const uint8_t *ycc, size_t npix, uint8_t *rgb) {
    for (size_t i = 0; i < npix; i++) {
        int y = ycc[3 * i], cb = ycc[3 * i + 1] - 128, cr = ycc[3 * i + 2] - 128;
        int r = y + ((91881 * cr + 32768) >> 16);
        int b = y + ((116130 * cb + 32768) >> 16);
        int g = y + ((-22554 * cb + (-46802 * cr + 32768)) >> 16);
        rgb[3 * i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        rgb[3 * i + 1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
        rgb[3 * i + 2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
    }
}
/* libjpeg jccolor.c rgb_ycc_convert */
void cso_rgb_to_ycc(const uint8_t *rgb, size_t npix, uint8_t *ycc) {
    for (size_t i = 0; i < npix; i++) {
        int r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        ycc[3 * i] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
        ycc[3 * i + 1] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
        ycc[3 * i + 2] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
    }
}
/* pixels in, JPEG out (the back half of convert_in_memory to JPEG): 8-bit grey (nc 1) or RGB (nc 3) -> optional image-rs Lanczos3 ->
   jccolor -> the same forward path and encoder as a resized JPEG; no metadata */
int cso_pixels_to_jpeg(const uint8_t *pix, int W, int H, int nc, const cso_enc_params *p, int width, int height, uint8_t **out, size_t *out_len) {
    if (nc != 1 && nc != 3) FAIL("unsupported channel count %d", nc);
    int nw = W, nh = H, rc = -1;
    if (width || height) cso_compute_dimensions(W, H, width, height, &nw, &nh);
    cso_image *dst = NULL;
    uint8_t *rs = (uint8_t *)malloc((size_t)nw * nh * nc);
    cso_lanczos3_resize(pix, W, H, nc, nw, nh, rs);
    if (nc == 3) cso_rgb_to_ycc(rs, (size_t)nw * nh, rs);
    if (cso_forward(rs, nw, nh, nc, p, NULL, &dst) == 0) rc = cso_encode(dst, p, NULL, 0, out, out_len);
    free(rs); cso_image_free(dst);
    return rc;
}
int cso_jpeg_compress_resized(const uint8_t *in, size_t n, const cso_enc_params *p, int width, int height, uint8_t **out, size_t *out_len) {
    cso_image *src = NULL, *dst = NULL;
    if (cso_decode(in, n, &src)) return -1;
    if (src->ncomp != 1 && src->ncomp != 3) { cso_image_free(src); FAIL("unsupported component count %d", src->ncomp); }
    int W = src->width, H = src->height, nc = src->ncomp, nw, nh, rc = -1;
    cso_compute_dimensions(W, H, width, height, &nw, &nh);
    uint8_t *pix = (uint8_t *)malloc((size_t)W * H * nc), *rs = (uint8_t *)malloc((size_t)nw * nh * nc);
    if (cso_decode_pixels(src, pix) == 0) {
        if (nc == 3) cso_ycc_to_rgb(pix, (size_t)W * H, pix);
        cso_lanczos3_resize(pix, W, H, nc, nw, nh, rs);
        if (nc == 3) cso_rgb_to_ycc(rs, (size_t)nw * nh, rs);
        if (cso_forward(rs, nw, nh, nc, p, NULL, &dst) == 0) {
            dst->meta = src->meta; dst->meta_len = src->meta_len;
            rc = cso_encode(dst, p, NULL, 0, out, out_len);
            dst->meta = NULL; dst->meta_len = 0;
        }
    }
    free(pix); free(rs); cso_image_free(src); cso_image_free(dst);
    return rc;
}
