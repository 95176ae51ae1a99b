// xform_block_check.cpp -- the pixel kernels' block arithmetic (caesium-clt_amd/csrc/k_pixel.hip, compiled here as the CSH_EMUL build compiles it: the
// same statements) against the oracle's plain ISLOW routines (oracle/jpeg_oracle.c cso_fdct_islow / cso_idct_islow) on the inputs where a packed
// 16-bit formulation could go wrong: every sample at an end of its range in every sign pattern of the DCT basis, the deringing overshoot's
// largest values (level-shifted 158 = 286 uncentred), DC-only blocks at the largest legal DC, random blocks.  The forward transform is checked
// three ways: packed (centred and uncentred entry) == the multiply-add form kept in the same file == the oracle; the quantiser against the
// integer statement of libjpeg's rule.  Built and run by tests/test_xform_block.py.  Test infrastructure: links the oracle.
#include "../caesium-clt_amd/csrc/k_pixel.hip"
extern "C" {
void cso_fdct_islow(const uint8_t *samples8x8, int32_t out[64]);
void cso_idct_islow(const int16_t coef[64], const uint16_t qt[64], uint8_t out[64]);
}
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;   // the emulation runtime's lane coordinates (pipeline.cpp defines them for the library)
thread_local int csh_emul_phase = 0;
int csh_emul_reverse = 0;
using namespace csh;
static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static DevQuant make_q(const uint16_t nat[64]) {
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    DevQuant q;
    memset(&q, 0, sizeof q);
    for (int k = 0; k < 64; k++) { q.q[k] = nat[zz[k]]; q.div[k] = int32_t(q.q[k]) * 8; q.rcp[k] = float((1.0 / double(q.div[k])) * (1.0 + 1.0 / 524288.0)); }
    return q;
}
static long checked = 0, bad = 0;
static int16_t tile[CSH_TILE_I16], rawbuf[64 * 64];
static int16_t dr[64][256];
// samples: level-shifted, natural order (may exceed 127 by the deringing overshoot)
static void check_fdct(const int centred[64], const DevQuant &q) {
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    int want[64];
    for (int i = 0; i < 64; i++) want[i] = centred[i];
    for (int r = 0; r < 8; r++) fdct1d<true>(want[8 * r], want[8 * r + 1], want[8 * r + 2], want[8 * r + 3], want[8 * r + 4], want[8 * r + 5], want[8 * r + 6], want[8 * r + 7]);
    for (int c = 0; c < 8; c++) fdct1d<false>(want[c], want[8 + c], want[16 + c], want[24 + c], want[32 + c], want[40 + c], want[48 + c], want[56 + c]);
    bool in8 = true;
    for (int i = 0; i < 64; i++) in8 = in8 && centred[i] <= 127;
    if (in8) {   // the oracle takes 8-bit samples
        uint8_t s8[64]; int32_t o[64];
        for (int i = 0; i < 64; i++) s8[i] = uint8_t(centred[i] + 128);
        cso_fdct_islow(s8, o);
        for (int i = 0; i < 64; i++) if (o[i] != want[i]) { if (bad++ < 5) printf("multiply-add form != oracle at %d: %d vs %d\n", i, want[i], o[i]); }
    }
    for (int variant = 0; variant < 2; variant++) {
        int x[64];
        for (int i = 0; i < 64; i++) x[i] = centred[i] + (variant ? 128 : 0);
        memset(tile, 0x55, sizeof tile); memset(rawbuf, 0x55, sizeof rawbuf);
        if (variant) fdct_quant_store<false, false>(x, q, tile, rawbuf, dr); else fdct_quant_store<false, true>(x, q, tile, rawbuf, dr);
        for (int k = 0; k < 64; k++) {
            const int got = rawbuf[(k >> 3) * CSH_RAW_OCT + (k & 7)], w = want[zz[k]];
            if (got != int16_t(w)) { if (bad++ < 5) printf("packed transform (variant %d) != multiply-add form at zig-zag %d: %d vs %d\n", variant, k, got, w); }
            const int a = w < 0 ? -w : w, d = q.div[k], lv = (a + (d >> 1)) / d, wantq = w < 0 ? -lv : lv;
            const int gotq = tile[coef_off(k)];
            if (gotq != wantq) { if (bad++ < 5) printf("quantiser at zig-zag %d: %d vs %d (coefficient %d, divisor %d)\n", k, gotq, wantq, w, d); }
            checked++;
        }
    }
}
static void check_idct(const int16_t coef_zz[64], const DevQuant &q) {
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    int16_t nat[64]; uint16_t qn[64]; uint8_t want[64];
    for (int k = 0; k < 64; k++) { nat[zz[k]] = coef_zz[k]; qn[zz[k]] = q.q[k]; tile[coef_off(k)] = coef_zz[k]; }
    cso_idct_islow(nat, qn, want);
    int x[64];
    load_idct<true>(tile, q, x);
    for (int i = 0; i < 64; i++) { checked++; if (x[i] + 128 != int(want[i])) { if (bad++ < 5) printf("inverse transform at %d: %d vs %d\n", i, x[i] + 128, want[i]); } }
}
int main() {
    uint16_t flat8[64], q80[64], big[64];
    static const uint8_t base[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                     18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    for (int i = 0; i < 64; i++) { flat8[i] = 1; q80[i] = uint16_t((base[i] * 40 + 50) / 100 < 1 ? 1 : (base[i] * 40 + 50) / 100); big[i] = uint16_t(base[i] * 50); }
    const DevQuant Q1 = make_q(flat8), Q80 = make_q(q80), QB = make_q(big);
    const DevQuant *qs[3] = {&Q1, &Q80, &QB};
    int s[64];
    // every basis function's sign pattern at full swing, with and without the deringing overshoot on the positive side
    for (int hi : {127, 158})
        for (int u = 0; u < 8; u++) for (int v = 0; v < 8; v++) for (int neg = 0; neg < 2; neg++) {
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
                const double c = cos((2 * x + 1) * u * M_PI / 16) * cos((2 * y + 1) * v * M_PI / 16);
                s[8 * y + x] = ((c >= 0) != (neg != 0)) ? hi : -128;
            }
            for (const DevQuant *q : qs) check_fdct(s, *q);
        }
    for (int v : {-128, -1, 0, 1, 127, 158}) { for (int i = 0; i < 64; i++) s[i] = v; check_fdct(s, Q80); }
    for (int t = 0; t < 20000; t++) {   // random blocks: full range, sparse extremes, smooth
        const int kind = t % 3;
        for (int i = 0; i < 64; i++) s[i] = kind == 0 ? int(rnd() % 287) - 128 : kind == 1 ? ((rnd() & 7) ? int(rnd() % 17) - 8 : ((rnd() & 1) ? 158 : -128)) : int(rnd() % 33) - 16 + (i / 8) * 9 - 30;
        check_fdct(s, *qs[t % 3]);
    }
    // inverse transform: DC-only blocks up to the largest legal DC, single coefficients at the range a stream of 8-bit samples can hold, random sparse blocks
    int16_t c[64];
    for (int dc = -1024; dc <= 1024; dc += 1) { memset(c, 0, sizeof c); c[0] = int16_t(dc); check_idct(c, Q1); }
    for (int k = 0; k < 64; k++) for (int v : {-1023, -512, -1, 1, 512, 1023}) { memset(c, 0, sizeof c); c[k] = int16_t(v); check_idct(c, Q1); c[0] = 300; check_idct(c, Q1); }
    for (int t = 0; t < 20000; t++) {
        memset(c, 0, sizeof c);
        const DevQuant &q = *qs[t % 2];
        c[0] = int16_t((int(rnd() % 2049) - 1024) / q.q[0]);
        const int n = int(rnd() % 20);
        for (int j = 0; j < n; j++) { const int k = 1 + int(rnd() % 63); c[k] = int16_t((int(rnd() % 401) - 200) / int(q.q[k])); }
        check_idct(c, q);
    }
    printf("checked=%ld bad=%ld\n", checked, bad);
    return bad ? 1 : 0;
}
