/* The pixel kernels' scalar quantiser (caesium-clt_amd/csrc/k_pixel.hip quant_one; the reciprocal: pipeline.cpp make_quant) against the integer
   statement of libjpeg's rule, sign(t) * ((|t| + d / 2) / d) with d = 8 q: every 16-bit table value, every |t| <= 2^15.
   glibc's fmaf is the correctly rounded fused multiply-add v_fma_f32 computes.  Run by tests/test_quant_reciprocal.py. */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
int main() {
    long bad = 0;
    for (int q = 1; q <= 65535; q++) {
        const int d = q * 8;
        const float r = (float)((1.0 / (double)d) * (1.0 + 1.0 / 524288.0));
        for (int t = -32768; t <= 32768; t++) {
            const int a = t < 0 ? -t : t;
            const int want = (a + (d >> 1)) / d;
            const int ws = t < 0 ? -want : want;
            float f = fmaf((float)t, r, 12582912.0f);
            uint32_t b; memcpy(&b, &f, 4);
            const int16_t got = (int16_t)(b & 0xFFFF);
            if (got != (int16_t)ws) { if (bad < 10) printf("q=%d t=%d want=%d got=%d\n", q, t, ws, got); bad++; }
        }
    }
    printf("bad=%ld\n", bad);
    return 0;
}
