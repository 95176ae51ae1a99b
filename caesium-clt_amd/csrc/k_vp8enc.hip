// k_vp8enc.hip -- SURVEY.md 8a row W2 on the device: libwebp's lossy encoder at its defaults (method 4, 4 segments, SNS 50, filter strength 60),
// byte for byte.  Statement: oracle/vp8enc_oracle.c (pinned to libwebp itself); the phases are the oracle's:
//   k_vp8_analyse    A  one 16-lane row per macroblock: the susceptibility (alpha) of its DC / TM residual spectra, from source samples only
//   k_vp8_segments   B  one wave per picture: 4-means over the alpha histogram, segment quantisers (SNS), matrices, lambdas, filter strengths
//   k_vp8_loop       C  ONE WORKGROUP (two waves) WALKS ONE PICTURE through the rate-distortion mode decision.  A macroblock needs the reconstruction and the modes /
//                       non-zero flags of its left, upper and upper-right neighbours, and the level-cost tables of its CHUNK (libwebp rebuilds them from the token
//                       statistics every max(96, mbs / 8) (+1) macroblocks): the host lays the macroblocks of a picture size out in steps (plan_class) and the kernel is
//                       a phased loop over them with a workgroup barrier between steps.  mb_batch: four macroblocks to a wave, sixteen lanes each -- the sixteen luma
//                       blocks of an i16 candidate, the ten modes of an i4 sub-block, two chroma modes x eight blocks.  chunk_stats (wave 0, between two chunks): the
//                       chunk's token statistics in libwebp's 16-bit form (its halving on overflow depends on the ORDER of the events: counted in groups of 64
//                       macroblocks, a group that straddles a halving point is recounted in order), the frame's probabilities so far and the level-cost tables of the
//                       next chunk, all in LDS; after the last chunk the final probabilities and the filter level
// The coder back end (decision streams, boolean coder, RIFF) is k_webp.hip.
#include "vp8enc_dev.h"
#include "devmem.hpp"
#include "kernels.h"
#include <cmath>

namespace csw {
using csh::DevBuf;

__device__ __forceinline__ static int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
__device__ __forceinline__ static int clipi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
__device__ __forceinline__ static int iabs(int v) { return v < 0 ? -v : v; }

// ---- per-lane arrays that live across LFOR regions (emulation: one per lane)
template <class T, int N>
struct LVA {
#ifdef CSH_EMUL
    T v[64][N];
    __device__ T (&operator[](int l))[N] { return v[l]; }
#else
    T v[N];
    __device__ __forceinline__ T (&operator[](int))[N] { return v; }
#endif
};

// ---- rows of sixteen lanes: sums, minima, broadcasts (product: the VALU's DPP path)
__device__ __forceinline__ static LV<int> rowsum(const LV<int> &x) {
    LV<int> r;
#ifdef CSH_EMUL
    for (int g = 0; g < 4; g++) { int s = 0; for (int k = 0; k < 16; k++) s += x.v[g * 16 + k]; for (int k = 0; k < 16; k++) r.v[g * 16 + k] = s; }
#else
    int v = x.v;
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    r.v = v;
#endif
    return r;
}
__device__ __forceinline__ static LV<int> halfsum(const LV<int> &x) {   // over each eight lanes
    LV<int> r;
#ifdef CSH_EMUL
    for (int g = 0; g < 8; g++) { int s = 0; for (int k = 0; k < 8; k++) s += x.v[g * 8 + k]; for (int k = 0; k < 8; k++) r.v[g * 8 + k] = s; }
#else
    int v = x.v;
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    r.v = v;
#endif
    return r;
}
__device__ __forceinline__ static LV<uint64_t> rowmin64(const LV<uint64_t> &x) {
    LV<uint64_t> r;
#ifdef CSH_EMUL
    for (int g = 0; g < 4; g++) { uint64_t m = ~0ull; for (int k = 0; k < 16; k++) m = x.v[g * 16 + k] < m ? x.v[g * 16 + k] : m; for (int k = 0; k < 16; k++) r.v[g * 16 + k] = m; }
#else
    uint64_t v = x.v;
    auto step = [&](auto dpp) {
        const uint32_t lo = uint32_t(dpp(int(uint32_t(v)))), hi = uint32_t(dpp(int(uint32_t(v >> 32))));
        const uint64_t o = (uint64_t(hi) << 32) | lo;
        v = o < v ? o : v;
    };
    step([](int a) { return __builtin_amdgcn_update_dpp(a, a, 0xB1, 0xf, 0xf, false); });
    step([](int a) { return __builtin_amdgcn_update_dpp(a, a, 0x4E, 0xf, 0xf, false); });
    step([](int a) { return __builtin_amdgcn_update_dpp(a, a, 0x141, 0xf, 0xf, false); });
    step([](int a) { return __builtin_amdgcn_update_dpp(a, a, 0x140, 0xf, 0xf, false); });
    r.v = v;
#endif
    return r;
}
__device__ __forceinline__ static LV<uint64_t> halfmin64(const LV<uint64_t> &x) {
    LV<uint64_t> r;
#ifdef CSH_EMUL
    for (int g = 0; g < 8; g++) { uint64_t m = ~0ull; for (int k = 0; k < 8; k++) m = x.v[g * 8 + k] < m ? x.v[g * 8 + k] : m; for (int k = 0; k < 8; k++) r.v[g * 8 + k] = m; }
#else
    uint64_t v = x.v;
    auto step = [&](auto dpp) {
        const uint32_t lo = uint32_t(dpp(int(uint32_t(v)))), hi = uint32_t(dpp(int(uint32_t(v >> 32))));
        const uint64_t o = (uint64_t(hi) << 32) | lo;
        v = o < v ? o : v;
    };
    step([](int a) { return __builtin_amdgcn_update_dpp(a, a, 0xB1, 0xf, 0xf, false); });
    step([](int a) { return __builtin_amdgcn_update_dpp(a, a, 0x4E, 0xf, 0xf, false); });
    step([](int a) { return __builtin_amdgcn_update_dpp(a, a, 0x141, 0xf, 0xf, false); });
    r.v = v;
#endif
    return r;
}

// ---- transforms (oracle: fdct4 / fwht / iwht / idct4_add / hadamard_w)
__device__ __forceinline__ static void fdct4(const int (&d)[16], int (&out)[16]) {   // d = src - pred, row-major
    int t[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = d[4 * i] + d[4 * i + 3], a1 = d[4 * i + 1] + d[4 * i + 2], a2 = d[4 * i + 1] - d[4 * i + 2], a3 = d[4 * i] - d[4 * i + 3];
        t[0 + i * 4] = (a0 + a1) * 8;
        t[1 + i * 4] = (__mul24(a2, 2217) + __mul24(a3, 5352) + 1812) >> 9;
        t[2 + i * 4] = (a0 - a1) * 8;
        t[3 + i * 4] = (__mul24(a3, 2217) - __mul24(a2, 5352) + 937) >> 9;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = t[0 + i] + t[12 + i], a1 = t[4 + i] + t[8 + i], a2 = t[4 + i] - t[8 + i], a3 = t[0 + i] - t[12 + i];
        out[0 + i] = (a0 + a1 + 7) >> 4;
        out[4 + i] = ((__mul24(a2, 2217) + __mul24(a3, 5352) + 12000) >> 16) + (a3 != 0);
        out[8 + i] = (a0 - a1 + 7) >> 4;
        out[12 + i] = (__mul24(a3, 2217) - __mul24(a2, 5352) + 51000) >> 16;
    }
}
__device__ __forceinline__ static void fwht(const int (&dc)[16], int (&out)[16]) {
    int t[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = dc[i * 4 + 0] + dc[i * 4 + 2], a1 = dc[i * 4 + 1] + dc[i * 4 + 3], a2 = dc[i * 4 + 1] - dc[i * 4 + 3], a3 = dc[i * 4 + 0] - dc[i * 4 + 2];
        t[0 + i * 4] = a0 + a1; t[1 + i * 4] = a3 + a2; t[2 + i * 4] = a3 - a2; t[3 + i * 4] = a0 - a1;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = t[0 + i] + t[8 + i], a1 = t[4 + i] + t[12 + i], a2 = t[4 + i] - t[12 + i], a3 = t[0 + i] - t[8 + i];
        out[0 + i] = (a0 + a1) >> 1; out[4 + i] = (a3 + a2) >> 1; out[8 + i] = (a3 - a2) >> 1; out[12 + i] = (a0 - a1) >> 1;
    }
}
__device__ __forceinline__ static void iwht(const int (&in)[16], int (&dc)[16]) {
    int t[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = in[0 + i] + in[12 + i], a1 = in[4 + i] + in[8 + i], a2 = in[4 + i] - in[8 + i], a3 = in[0 + i] - in[12 + i];
        t[0 + i] = a0 + a1; t[8 + i] = a0 - a1; t[4 + i] = a3 + a2; t[12 + i] = a3 - a2;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int d = t[0 + i * 4] + 3, a0 = d + t[3 + i * 4], a1 = t[1 + i * 4] + t[2 + i * 4], a2 = t[1 + i * 4] - t[2 + i * 4], a3 = d - t[3 + i * 4];
        dc[i * 4 + 0] = (a0 + a1) >> 3; dc[i * 4 + 1] = (a3 + a2) >> 3; dc[i * 4 + 2] = (a0 - a1) >> 3; dc[i * 4 + 3] = (a3 - a2) >> 3;
    }
}
// (the 24-bit multiplier runs at full rate, the 32-bit one at a quarter; every operand here is far inside 24 bits and the low 32 bits of the product are the same)
__device__ __forceinline__ static int mul1(int a) { return (__mul24(a, 20091) >> 16) + a; }
__device__ __forceinline__ static int mul2(int a) { return __mul24(a, 35468) >> 16; }
__device__ __forceinline__ static void idct4_add(const int (&in)[16], const int (&pred)[16], int (&px)[16]) {
    int t[16];
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a = in[0 + i] + in[8 + i], b = in[0 + i] - in[8 + i], c = mul2(in[4 + i]) - mul1(in[12 + i]), d = mul1(in[4 + i]) + mul2(in[12 + i]);
        t[0 + i * 4] = a + d; t[1 + i * 4] = b + c; t[2 + i * 4] = b - c; t[3 + i * 4] = a - d;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int dc = t[0 + i] + 4, a = dc + t[8 + i], b = dc - t[8 + i], c = mul2(t[4 + i]) - mul1(t[12 + i]), d = mul1(t[4 + i]) + mul2(t[12 + i]);
        px[i * 4 + 0] = clip8(pred[i * 4 + 0] + ((a + d) >> 3)); px[i * 4 + 1] = clip8(pred[i * 4 + 1] + ((b + c) >> 3));
        px[i * 4 + 2] = clip8(pred[i * 4 + 2] + ((b - c) >> 3)); px[i * 4 + 3] = clip8(pred[i * 4 + 3] + ((a - d) >> 3));
    }
}
__device__ __forceinline__ static int hadamard_w(const int (&p)[16]) {   // weighted Hadamard spectrum of a 4x4 block of samples
    int t[16], sum = 0;
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = p[4 * i] + p[4 * i + 2], a1 = p[4 * i + 1] + p[4 * i + 3], a2 = p[4 * i + 1] - p[4 * i + 3], a3 = p[4 * i] - p[4 * i + 2];
        t[0 + i * 4] = a0 + a1; t[1 + i * 4] = a3 + a2; t[2 + i * 4] = a3 - a2; t[3 + i * 4] = a0 - a1;
    }
    CSH_UNROLL
    for (int i = 0; i < 4; i++) {
        const int a0 = t[0 + i] + t[8 + i], a1 = t[4 + i] + t[12 + i], a2 = t[4 + i] - t[12 + i], a3 = t[0 + i] - t[8 + i];
        sum += __mul24(int(kVp8WeightY[0 + i]), iabs(a0 + a1)) + __mul24(int(kVp8WeightY[4 + i]), iabs(a3 + a2)) + __mul24(int(kVp8WeightY[8 + i]), iabs(a3 - a2)) + __mul24(int(kVp8WeightY[12 + i]), iabs(a0 - a1));
    }
    return sum;
}
__device__ __forceinline__ static int sse16(const int (&a)[16], const int (&b)[16]) { int s = 0; CSH_UNROLL for (int k = 0; k < 16; k++) { const int d = a[k] - b[k]; s += __mul24(d, d); } return s; }

// ---- quantiser (oracle: quantize_block / quantize_single)
struct QM { int q0, q1, iq0, iq1, b0, b1, z0, z1; };
__device__ __forceinline__ static QM load_qm(const Vp8SegDev &S, int t) { return QM{S.q[t][0], S.q[t][1], S.iq[t][0], S.iq[t][1], S.bias[t][0], S.bias[t][1], S.zth[t][0], S.zth[t][1]}; }
// c: raster coefficients in, their dequantised values out; lv: levels in scan order.  SHARP: luma AC coefficients are boosted before the threshold
template <bool SHARP>
__device__ __forceinline__ static int quant_block(int (&c)[16], int (&lv)[16], const QM &m) {
    int any = 0;
    CSH_UNROLL
    for (int n = 0; n < 16; n++) {
        const int j = kVp8Zigzag[n];
        const int q = j ? m.q1 : m.q0, iq = j ? m.iq1 : m.iq0, b = j ? m.b1 : m.b0, z = j ? m.z1 : m.z0;
        const int sign = c[j] < 0;
        const uint32_t coeff = uint32_t(iabs(c[j]) + (SHARP ? __mul24(int(kVp8FreqSharpening[j]), q) >> 11 : 0));
        (void)z;   // (coeff <= zthresh <=> the division below gives 0: that is how zthresh is defined)
        int level = int((__umul24(coeff, uint32_t(iq)) + uint32_t(b)) >> 17);   // coeff < 2^16, iq <= 2^15
        if (level > 2047) level = 2047;
        if (sign) level = -level;
        c[j] = __mul24(level, q);
        lv[n] = level;
        any |= level;
    }
    return any != 0;
}
__device__ __forceinline__ static int quant_single(int &v, const QM &m) {   // chroma DC with error diffusion: returns what was lost, halved
    int V = v;
    const int sign = V < 0;
    if (sign) V = -V;
    if (V > m.z0) {
        const int qV = int((uint32_t(V) * uint32_t(m.iq0) + uint32_t(m.b0)) >> 17) * m.q0, err = V - qV;
        v = sign ? -qV : qV;
        return (sign ? -err : err) >> 1;
    }
    v = 0;
    return (sign ? -V : V) >> 1;
}

// =================================================================================================== A: analysis
// grid (ceil(nmb / 4), images); a row of sixteen lanes per macroblock: lane = luma block, then lanes 0..7 = the U and V blocks.  DC and TM predictions from
// SOURCE samples around the macroblock (oracle: analyse_mb); per prediction the histogram of |coefficient| >> 3 over its blocks; alpha = 510 * last / max
struct AnaLds { uint32_t hist[4][32]; };   // luma DC, luma TM, chroma DC, chroma TM
__device__ __forceinline__ static int alpha_of(const uint32_t *h) {
    int maxv = 0, last = 1;
    for (int k = 0; k < 32; k++) if (h[k] > 0) { if (int(h[k]) > maxv) maxv = int(h[k]); last = k; }
    return maxv > 1 ? 510 * last / maxv : 0;
}
// one N x N prediction of the encoder (oracle: predN) for the 4x4 block at (bx, by): top / left: the block's four neighbours above / to the left (already 127 / 129
// where the frame ends), tl the macroblock's corner, dcv the DC value
__device__ __forceinline__ static void pred_block(int mode, const int (&top)[4], const int (&left)[4], int tl, int dcv, bool hl, bool ht, int (&pred)[16]) {
    CSH_UNROLL
    for (int k = 0; k < 16; k++) {
        const int x = k & 3, y = k >> 2;
        int v;
        if (mode == 0) v = dcv;
        else if (mode == 2) v = top[x];
        else if (mode == 3) v = left[y];
        else v = (hl && ht) ? clip8(top[x] + left[y] - tl) : !ht ? left[y] : top[x];   // TM without a neighbour degenerates to H / V (129 with neither: left[] is 129 then)
        pred[k] = v;
    }
}
__device__ __forceinline__ static int dc_value(int sum, bool hl, bool ht, int N) {   // sum: the neighbours that exist
    const int sh = N == 16 ? 5 : 4;
    if (hl && ht) return (sum + N) >> sh;
    if (hl || ht) return (2 * sum + N) >> sh;
    return 128;
}
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_vp8_analyse(const WebpImg *imgs, const uint8_t *work, int16_t *levels, Vp8FrameDev *frames) {
    CSH_SHARED AnaLds s[4];
    const WebpImg im = imgs[blockIdx.y];
    const int mbw = int(im.mbw), nmb = mbw * int(im.mbh), ys = mbw * 16, cs = mbw * 8;
    if (int(blockIdx.x) * 4 >= nmb) return;
    LFOR(l) for (int i = l & 15; i < 128; i += 16) (&s[l >> 4].hist[0][0])[i] = 0;
    CSP_WAVE_SYNC();
    LV<int> edge;
    LFOR(l) {
        const int n = int(blockIdx.x) * 4 + (l >> 4), i = l & 15;
        edge[l] = 0;
        if (n < nmb) {
            const int my = n / mbw, mx = n - my * mbw;
            const uint8_t *sy = work + im.y_off + size_t(my * 16) * ys + mx * 16;
            edge[l] = (my ? int(sy[i - ys]) : 0) + (mx ? int(sy[size_t(i) * ys - 1]) : 0);
        }
    }
    const LV<int> ysum = rowsum(edge);
    LFOR(l) {
        const int n = int(blockIdx.x) * 4 + (l >> 4), i = l & 15, k = i & 7;
        edge[l] = 0;
        if (n < nmb) {
            const int my = n / mbw, mx = n - my * mbw;
            const uint8_t *sc = work + (i < 8 ? im.u_off : im.v_off) + size_t(my * 8) * cs + mx * 8;
            edge[l] = (my ? int(sc[k - cs]) : 0) + (mx ? int(sc[size_t(k) * cs - 1]) : 0);
        }
    }
    const LV<int> csum = halfsum(edge);
    LV<int> usum, vsum;   // the U / V sums of the row, taken while every lane is active (a cross-lane read of an inactive lane returns 0)
    LFOR(l) {
#ifdef CSH_EMUL
        usum[l] = csum.v[l & 48]; vsum[l] = csum.v[(l & 48) + 8];
#else
        usum[l] = __shfl(csum.v, l & 48, 64); vsum[l] = __shfl(csum.v, (l & 48) + 8, 64);
#endif
    }
    LFOR(l) {
        const int g = l >> 4, n = int(blockIdx.x) * 4 + g, i = l & 15;
        if (n < nmb) {
            const int my = n / mbw, mx = n - my * mbw;
            const bool hl = mx > 0, ht = my > 0;
            for (int pass = 0; pass < 2; pass++) {   // 0: the luma block of this lane; 1: lanes 0..7 a chroma block
                if (pass && i >= 8) break;
                const int N = pass ? 8 : 16, stride = pass ? cs : ys, bx = pass ? i & 1 : i & 3, by = pass ? (i >> 1) & 1 : i >> 2;
                const uint8_t *mb = work + (pass ? (i < 4 ? im.u_off : im.v_off) : im.y_off) + size_t(my * N) * stride + mx * N, *b = mb + size_t(by * 4) * stride + bx * 4;
                int src[16], top[4], left[4];
                CSH_UNROLL
                for (int k = 0; k < 16; k++) src[k] = b[size_t(k >> 2) * stride + (k & 3)];
                CSH_UNROLL
                for (int k = 0; k < 4; k++) { top[k] = ht ? int(mb[bx * 4 + k - stride]) : 127; left[k] = hl ? int(mb[size_t(by * 4 + k) * stride - 1]) : 129; }
                const int tl = (hl && ht) ? int(mb[-stride - 1]) : 0;
                const int dcv = !pass ? dc_value(ysum[l], hl, ht, 16) : dc_value(i < 4 ? usum[l] : vsum[l], hl, ht, 8);
                for (int mode = 0; mode < 2; mode++) {
                    int pred[16], d[16], c[16];
                    pred_block(mode, top, left, tl, dcv, hl, ht, pred);
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) d[k] = src[k] - pred[k];
                    fdct4(d, c);
                    // (most coefficients land in bin 0: counted in a register and added once per block, or every lane of the row would queue at one LDS word)
                    uint32_t zeros = 0;
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) { const int v = iabs(c[k]) >> 3; if (v == 0) zeros++; else atomicAdd(&s[g].hist[pass * 2 + mode][v > 31 ? 31 : v], 1u); }
                    if (zeros) atomicAdd(&s[g].hist[pass * 2 + mode][0], zeros);
                }
            }
        }
    }
    CSP_WAVE_SYNC();
    LFOR(l) {
        const int g = l >> 4, n = int(blockIdx.x) * 4 + g;
        if ((l & 15) == 0 && n < nmb) {
            const int a0 = alpha_of(s[g].hist[0]), a1 = alpha_of(s[g].hist[1]), u0 = alpha_of(s[g].hist[2]), u1 = alpha_of(s[g].hist[3]);
            const int best = a1 > a0 ? a1 : a0, best_uv = u1 > u0 ? u1 : u0;
            const int alpha = clipi(255 - ((3 * best + best_uv + 2) >> 2), 0, 255);
            levels[im.lev_off + size_t(n) * WEBP_MB_REC + MB_ALPHA] = int16_t(alpha);
            Vp8FrameDev *F = frames + blockIdx.y;
            atomicAdd(&F->hist[alpha], 1u);
            atomicAdd(&F->alpha_sum, alpha);
            atomicAdd(&F->uv_alpha_sum, best_uv);
        }
    }
}

// =================================================================================================== B: segments, quantisers, cost tables
// level_cost tables from the current probabilities (oracle: level_costs): lanes share the 96 x 68 entries
__device__ static void make_level_costs(uint16_t *lc, const uint8_t *coeffs) {
    LFOR(l)
        for (int e = l; e < 96 * (VP8_MAXLV + 1); e += 64) {
            const int tbc = e / (VP8_MAXLV + 1), v = e - tbc * (VP8_MAXLV + 1), c = tbc % 3;
            const uint8_t *p = coeffs + tbc * 11;
            const int cost0 = c > 0 ? vp8_bitcost(1, p[0]) : 0;
            int cost;
            if (v == 0) cost = vp8_bitcost(0, p[1]) + cost0;
            else {
                int pattern = kVp8LevelCodes[(v - 1) * 2], bits = kVp8LevelCodes[(v - 1) * 2 + 1];
                cost = vp8_bitcost(1, p[1]) + cost0;
                for (int i = 2; pattern; i++, bits >>= 1, pattern >>= 1) if (pattern & 1) cost += vp8_bitcost(bits & 1, p[i]);
            }
            lc[e] = uint16_t(cost);
        }
}
__device__ static void expand_matrix(Vp8SegDev &S, int t, int type_bias) {
    for (int i = 0; i < 2; i++) {
        S.iq[t][i] = (1 << 17) / S.q[t][i];
        S.bias[t][i] = int(kVp8BiasMatrices[type_bias * 2 + i]) << 9;
        S.zth[t][i] = ((1 << 17) - 1 - S.bias[t][i]) / S.iq[t][i];
    }
}
// qtab: the segment quantiser index for each alpha -127..127 at the pictures' quality (the host evaluates libwebp's two pow() calls with the C library's pow)
__global__ void __launch_bounds__(CSP_WAVE_THREADS) k_vp8_segments(const WebpImg *imgs, Vp8FrameDev *frames, const uint8_t *qtabs) {
    const WebpImg im = imgs[blockIdx.x];
    Vp8FrameDev *F = frames + blockIdx.x;
    const int nmb = int(im.mbw * im.mbh), SNS = 50, FSTRENGTH = 60;
    const uint8_t *qtab = qtabs + size_t(im.qtab) * 256;
    LFOR(l) {
        if (l == 0) {
            // 4-means over the histogram (oracle: assign_segments)
            int nb = 4, amap[256], centers[4], min_a, max_a, n, wavg = 0;
            const uint32_t *hist = F->hist;
            for (n = 0; n <= 255 && hist[n] == 0; n++) {}
            min_a = n;
            for (n = 255; n > min_a && hist[n] == 0; n--) {}
            max_a = n;
            const int range = max_a - min_a;
            for (int k = 0, m = 1; k < nb; k++, m += 2) centers[k] = min_a + (m * range) / (2 * nb);
            for (int k = 0; k < 256; k++) amap[k] = 0;
            for (int it = 0; it < 6; it++) {
                int accum[4] = {0, 0, 0, 0}, dist[4] = {0, 0, 0, 0}, displaced = 0, total = 0;
                n = 0;
                for (int a = min_a; a <= max_a; a++)
                    if (hist[a]) {
                        while (n + 1 < nb && iabs(a - centers[n + 1]) < iabs(a - centers[n])) n++;
                        amap[a] = n;
                        dist[n] += a * int(hist[a]);
                        accum[n] += int(hist[a]);
                    }
                wavg = 0;
                for (n = 0; n < nb; n++)
                    if (accum[n]) {
                        const int c = (dist[n] + accum[n] / 2) / accum[n];
                        displaced += iabs(centers[n] - c);
                        centers[n] = c;
                        wavg += c * accum[n];
                        total += accum[n];
                    }
                wavg = (wavg + total / 2) / total;
                if (displaced < 5) break;
            }
            int mn = centers[0], mx = centers[0];
            for (n = 0; n < nb; n++) { if (mn > centers[n]) mn = centers[n]; if (mx < centers[n]) mx = centers[n]; }
            if (mx == mn) mx = mn + 1;
            Vp8SegDev seg[4];
            for (n = 0; n < 4; n++) {
                Vp8SegDev &S = seg[n];
                memset(&S, 0, sizeof S);
                S.alpha = clipi(255 * (centers[n] - wavg) / (mx - mn), -127, 127);
                S.beta = clipi(255 * (centers[n] - mn) / (mx - mn), 0, 255);
                S.quant = qtab[S.alpha + 127];
                const int qstep = int(kVp8AcQ[S.quant]) >> 2, base = kVp8LevelsFromDelta[qstep > 63 ? 63 : qstep], f = base * (5 * FSTRENGTH) / (256 + S.beta);
                S.fstrength = f < 2 ? 0 : f > 63 ? 63 : f;
            }
            const int uv_alpha = F->uv_alpha_sum / nmb;
            const int dq_uv_ac = clipi((uv_alpha - 64) * (6 - -4) / (100 - 30) * SNS / 100, -4, 6), dq_uv_dc = clipi(-4 * SNS / 100, -15, 15);
            F->base_quant = seg[0].quant; F->dq_uv_ac = dq_uv_ac; F->dq_uv_dc = dq_uv_dc;
            // segments that ended up alike are merged (oracle: the "nfinal" loop)
            int nseg = 4, remap[4] = {0, 1, 2, 3}, nfinal = 1;
            for (int s1 = 1; s1 < nseg; s1++) {
                int s2, found = 0;
                for (s2 = 0; s2 < nfinal; s2++) if (seg[s1].quant == seg[s2].quant && seg[s1].fstrength == seg[s2].fstrength) { found = 1; break; }
                remap[s1] = s2;
                if (!found) { if (nfinal != s1) seg[nfinal] = seg[s1]; nfinal++; }
            }
            if (nfinal < nseg) { for (int i = nfinal; i < nseg; i++) seg[i] = seg[nfinal - 1]; nseg = nfinal; } else for (int k = 0; k < 4; k++) remap[k] = k;
            for (n = 0; n < nseg; n++) {
                Vp8SegDev &S = seg[n];
                const int q = S.quant;
                S.q[0][0] = kVp8DcQ[q]; S.q[0][1] = kVp8AcQ[q];
                S.q[1][0] = int(kVp8DcQ[q]) * 2; S.q[1][1] = kVp8AcTable2[q];
                S.q[2][0] = kVp8DcQ[clipi(q + dq_uv_dc, 0, 117)]; S.q[2][1] = kVp8AcQ[clipi(q + dq_uv_ac, 0, 127)];
                expand_matrix(S, 0, 0); expand_matrix(S, 1, 1); expand_matrix(S, 2, 2);
                const int q4 = (S.q[0][0] + 15 * S.q[0][1] + 8) >> 4, q16 = (S.q[1][0] + 15 * S.q[1][1] + 8) >> 4, quv = (S.q[2][0] + 15 * S.q[2][1] + 8) >> 4;
                auto lam = [](int v) { return v < 1 ? 1 : v; };
                S.lambda_i4 = lam((3 * q4 * q4) >> 7); S.lambda_i16 = lam(3 * q16 * q16); S.lambda_uv = lam((3 * quv * quv) >> 6); S.lambda_mode = lam((q4 * q4) >> 7);
                S.tlambda = (SNS * q4) >> 5;
                S.min_disto = 20 * S.q[0][0];
                S.max_edge = 0;
            }
            // macroblocks per segment, the segment-id probabilities
            int cnt[4] = {0, 0, 0, 0};
            for (int a = 0; a < 256; a++) if (hist[a]) cnt[remap[amap[a]]] += int(hist[a]);
            int sp[3] = {255, 255, 255}, update_map = 0;
            if (nseg > 1) {
                auto getp = [](int a, int b) { return a + b == 0 ? 255 : (255 * a + (a + b) / 2) / (a + b); };
                sp[0] = getp(cnt[0] + cnt[1], cnt[2] + cnt[3]); sp[1] = getp(cnt[0], cnt[1]); sp[2] = getp(cnt[2], cnt[3]);
                update_map = sp[0] != 255 || sp[1] != 255 || sp[2] != 255;
            }
            for (int a = 0; a < 256; a++) F->alpha_seg[a] = uint8_t(update_map ? remap[amap[a]] : 0);
            for (n = 0; n < 4; n++) F->seg[n] = seg[n];
            F->nseg = nseg; F->update_map = update_map; F->seg_probs[0] = sp[0]; F->seg_probs[1] = sp[1]; F->seg_probs[2] = sp[2];
            F->filter_level = seg[0].fstrength;
            F->dirty = 0;
            F->diffuse = im.quality <= 98;
        }
        for (int i = l; i < VP8_NSLOT; i += 64) { F->coeffs[i] = kVp8CoefProbs[i]; F->stats[i] = 0; }
    }
}

// =================================================================================================== C: the macroblock loop
// the host's plan for pictures of one size: steps[step_off + s] .. steps[step_off + s + 1]: the macroblocks of step s in items[item_off + ..] (mx | my << 16);
// chunk j ends before macroblock chunk_end[chunk_off + j] and its statistics step is chunk_step[chunk_off + j]
struct Vp8Class { uint32_t mbw, mbh, nsteps, step_off, item_off, nchunks, chunk_off, pad; };

struct MbLds {
    uint32_t cbw[17 * 8];          // luma with its edges: 17 rows of 32 bytes; row 0 = the row above, byte 3 = the column to the left, bytes 4..19 the macroblock, 20..23 above-right
    uint32_t srcw[64];             // luma source 16 x 16
    uint32_t csrcw[2][16];         // chroma source 8 x 8 x 2
    uint8_t ce[2][20];             // chroma edges: [0..7] above, [8..15] left, [16] corner
    int32_t dcs[16];               // i16: the sixteen block DCs
    int32_t hsrc[16];              // the weighted Hadamard spectrum of each luma source block
    int16_t y2best[16];            // i16: the best mode's Y2 levels
    int16_t lv4[16][16];           // i4: the chosen levels
    int16_t lvuv[8][16];           // chroma: the best mode's levels
    uint32_t recuv[2][16];         // chroma: its reconstruction
    int32_t cdc[4][4];             // chroma: the four DCs of one (mode & 1, plane) for the error diffusion
    int32_t cerr[4][4];            // and the three errors it hands on
    uint8_t e[16];                 // i4: the edge line L K J I X A B C D E F G H of the current sub-block
    int32_t win[4];                // i4: the winning candidate's D, SD, R, H
    uint8_t bm[16], nz4[16];
    int8_t derr[2][4];
};
// the picture's tables, one copy per wave (all four macroblocks of a wave belong to one picture): level costs, probabilities, the fixed tables, the segments
struct MbTables {
    uint16_t lc[96 * (VP8_MAXLV + 1)];
    uint16_t fixed[2048];            // kVp8LevelFixedCosts
    uint16_t ent[256];
    uint8_t coeffs[VP8_NSLOT];
    Vp8SegDev seg[4];
};
__device__ __forceinline__ static int tb_bitcost(const MbTables &T, int bit, int p) { return T.ent[bit ? 255 - p : p]; }
// bits (1/256) of one block's levels given the context of its first coefficient (oracle: residual_cost).  Every table look-up is addressed from the levels
// alone, so the sixteen steps are independent loads
__device__ __forceinline__ static int block_cost(const MbTables &T, int type, int first, int ctx0, const int (&lv)[16]) {
    int last = -1;
    CSH_UNROLL
    for (int i = 0; i < 16; i++) if (lv[i]) last = i;
    const int p0 = T.coeffs[vp8_slot(type, first, ctx0)];   // (band of position 0 / 1 = 0 / 1)
    int cost = ctx0 == 0 ? tb_bitcost(T, 1, p0) : 0, ctx = ctx0;
    CSH_UNROLL
    for (int n = 0; n < 16; n++) {
        const int v = iabs(lv[n]), on = (n >= first && n <= last) ? 1 : 0;
        const int c = int(T.fixed[v]) + int(T.lc[((type * 8 + kVp8Bands[n]) * 3 + ctx) * (VP8_MAXLV + 1) + (v > VP8_MAXLV ? VP8_MAXLV : v)]);
        cost += on ? c : 0;
        ctx = n >= first ? (v >= 2 ? 2 : v) : ctx;
    }
    int vl = 0;
    CSH_UNROLL
    for (int n = 0; n < 16; n++) vl = n == last ? iabs(lv[n]) : vl;
    const int tail = last < 15 ? tb_bitcost(T, 0, T.coeffs[vp8_slot(type, kVp8Bands[(last < 0 ? 0 : last) + 1], vl == 1 ? 1 : 2)]) : 0;
    return last < 0 ? tb_bitcost(T, 0, p0) : cost + tail;
}
__device__ __forceinline__ static uint32_t pack4(const int (&p)[16], int r) { return uint32_t(p[r * 4]) | (uint32_t(p[r * 4 + 1]) << 8) | (uint32_t(p[r * 4 + 2]) << 16) | (uint32_t(p[r * 4 + 3]) << 24); }

// the macroblocks items[lo .. min(lo + 4, hi)) of one picture, a row of sixteen lanes each
__device__ __forceinline__ static void mb_batch(MbLds (&s)[4], const MbTables &T, const WebpImg &im, Vp8FrameDev *F, const uint32_t *items, uint32_t lo, uint32_t hi, uint8_t *work, int16_t *levels) {
    const int mbw = int(im.mbw), ys = mbw * 16, cs = mbw * 8;
    // ---- which macroblock each row of sixteen lanes works on
    LV<int> ok, vmx, vmy;
    LFOR(l) {
        const uint32_t it = lo + uint32_t(l >> 4);
        ok[l] = it < hi;
        const uint32_t xy = ok[l] ? items[it] : 0u;
        vmx[l] = int(xy & 0xFFFFu); vmy[l] = int(xy >> 16);
    }

#define MB_PROLOGUE                                                                                                                             \
    const int g = l >> 4, i = l & 15, mx = vmx[l], my = vmy[l];                                                                                 \
    const bool hl = mx > 0, ht = my > 0;                                                                                                        \
    int16_t *L = levels + im.lev_off + (size_t(my) * mbw + mx) * WEBP_MB_REC;                                                                    \
    uint8_t *cb = reinterpret_cast<uint8_t *>(s[g].cbw), *srcb = reinterpret_cast<uint8_t *>(s[g].srcw);                                        \
    (void)i; (void)hl; (void)ht; (void)L; (void)cb; (void)srcb

    // ---- source samples, the edges of the reconstruction, the neighbours' contexts
    LV<int> segv;
    LV<uint32_t> topm, leftm;
    LFOR(l) {
        segv[l] = 0; topm[l] = 0; leftm[l] = 0;
        if (ok[l]) {
            MB_PROLOGUE;
            const uint8_t *sy = work + im.y_off + size_t(my * 16) * ys + mx * 16, *ry = work + im.ry_off + size_t(my * 16) * ys + mx * 16;
            const uint4 row = *reinterpret_cast<const uint4 *>(sy + size_t(i) * ys);
            s[g].srcw[i * 4] = row.x; s[g].srcw[i * 4 + 1] = row.y; s[g].srcw[i * 4 + 2] = row.z; s[g].srcw[i * 4 + 3] = row.w;
            {
                const int pl = i >> 3, r = i & 7;
                const uint8_t *sc = work + (pl ? im.v_off : im.u_off) + size_t(my * 8 + r) * cs + mx * 8, *rc = work + (pl ? im.rv_off : im.ru_off) + size_t(my * 8) * cs + mx * 8;
                const uint2 crow = *reinterpret_cast<const uint2 *>(sc);
                s[g].csrcw[pl][r * 2] = crow.x;
                s[g].csrcw[pl][r * 2 + 1] = crow.y;
                s[g].ce[pl][r] = ht ? rc[r - cs] : uint8_t(127);
                s[g].ce[pl][8 + r] = hl ? rc[size_t(r) * cs - 1] : uint8_t(129);
                if (r == 0) s[g].ce[pl][16] = (hl && ht) ? rc[-cs - 1] : uint8_t(0);
            }
            cb[4 + i] = ht ? ry[i - ys] : uint8_t(127);
            cb[(i + 1) * 32 + 3] = hl ? ry[size_t(i) * ys - 1] : uint8_t(129);
            if (i < 4) cb[20 + i] = !ht ? uint8_t(127) : (mx + 1 < mbw ? ry[16 + i - ys] : ry[15 - ys]);
            if (i == 0) cb[3] = ht ? (hl ? ry[-ys - 1] : uint8_t(129)) : uint8_t(127);
            segv[l] = F->alpha_seg[L[MB_ALPHA] & 255];
            topm[l] = ht ? nz_mask(L - size_t(mbw) * WEBP_MB_REC) : 0u;
            leftm[l] = hl ? nz_mask(L - WEBP_MB_REC) : 0u;
        }
    }
    CSP_WAVE_SYNC();

    // ---- DC values; the Hadamard spectrum of every luma source block (the texture term compares it with the reconstruction's)
    LV<int> t1, t2;
    LFOR(l) {
        t1[l] = 0; t2[l] = 0;
        if (ok[l]) {
            MB_PROLOGUE;
            t1[l] = (ht ? int(cb[4 + i]) : 0) + (hl ? int(cb[(i + 1) * 32 + 3]) : 0);
            t2[l] = (ht ? int(s[g].ce[i >> 3][i & 7]) : 0) + (hl ? int(s[g].ce[i >> 3][8 + (i & 7)]) : 0);
            int src[16];
            CSH_UNROLL
            for (int k = 0; k < 16; k++) src[k] = srcb[((i >> 2) * 4 + (k >> 2)) * 16 + (i & 3) * 4 + (k & 3)];
            s[g].hsrc[i] = hadamard_w(src);
        }
    }
    const LV<int> ydc_sum = rowsum(t1), cdc_sum = halfsum(t2);   // cdc_sum: lanes 0..7 the U sum, 8..15 the V sum
    LV<int> usum, vsum;
    LFOR(l) {
#ifdef CSH_EMUL
        usum[l] = cdc_sum.v[l & 48]; vsum[l] = cdc_sum.v[(l & 48) + 8];
#else
        usum[l] = __shfl(cdc_sum.v, l & 48, 64); vsum[l] = __shfl(cdc_sum.v, (l & 48) + 8, 64);
#endif
    }
    CSP_WAVE_SYNC();

    // =========================================================================================== i16: four modes, lane = luma block
    LVA<int, 8> best_lv;         // the best mode's levels of this lane's block, two to a register
    LVA<uint32_t, 4> best_rec;   // and its reconstruction, a row to a register
    LV<int64_t> sc16;            // the best i16 candidate's score for the i16 / i4 decision (lambda_mode)
    LV<int> best16, nz16, srcflat, D16;
    {
        LFOR(l) {
            best16[l] = -1; nz16[l] = 0; D16[l] = 0; sc16[l] = 0; t1[l] = 1;
            if (ok[l]) {
                MB_PROLOGUE;
                const uint32_t rep = (s[g].srcw[0] & 255u) * 0x01010101u;
                int same = 1;
                CSH_UNROLL
                for (int k = 0; k < 4; k++) same &= s[g].srcw[(((i >> 2) * 4 + k) * 4) + (i & 3)] == rep;
                t1[l] = same;
            }
        }
        const uint64_t nf = lballot([&](int l) { return t1[l] == 0; });
        LFOR(l) srcflat[l] = ((nf >> (l & 48)) & 0xFFFFu) == 0;
        LV<int64_t> best_score;
        LFOR(l) best_score[l] = 0;
        for (int mode = 0; mode < 4; mode++) {
            LVA<int, 16> coef;
            LFOR(l) if (ok[l]) {
                MB_PROLOGUE;
                const int bx = i & 3, by = i >> 2;
                int top[4], left[4], pred[16], d[16];
                CSH_UNROLL
                for (int k = 0; k < 4; k++) { top[k] = cb[4 + bx * 4 + k]; left[k] = cb[(by * 4 + k + 1) * 32 + 3]; }
                pred_block(mode, top, left, cb[3], dc_value(ydc_sum[l], hl, ht, 16), hl, ht, pred);
                CSH_UNROLL
                for (int k = 0; k < 16; k++) d[k] = int(srcb[(by * 4 + (k >> 2)) * 16 + bx * 4 + (k & 3)]) - pred[k];
                fdct4(d, coef[l]);
                s[g].dcs[i] = coef[l][0];
            }
            CSP_WAVE_SYNC();
            LVA<int, 16> lv;
            LVA<uint32_t, 4> rec4;
            LVA<int, 16> lv2;
            LV<int> vD, vSD, vnz, vnz2;
            LFOR(l) {
                vD[l] = 0; vSD[l] = 0; vnz[l] = 0; vnz2[l] = 0;
                if (ok[l]) {
                    MB_PROLOGUE;
                    const int bx = i & 3, by = i >> 2;
                    const Vp8SegDev &S = T.seg[segv[l]];
                    int dcs[16], y2[16], top[4], left[4], pred[16], src[16], rec[16];
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) dcs[k] = s[g].dcs[k];
                    fwht(dcs, y2);
                    vnz2[l] = quant_block<false>(y2, lv2[l], load_qm(S, 1));
                    iwht(y2, dcs);
                    int mine = 0;
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) mine = i == k ? dcs[k] : mine;
                    coef[l][0] = 0;
                    vnz[l] = quant_block<true>(coef[l], lv[l], load_qm(S, 0));
                    coef[l][0] = mine;
                    CSH_UNROLL
                    for (int k = 0; k < 4; k++) { top[k] = cb[4 + bx * 4 + k]; left[k] = cb[(by * 4 + k + 1) * 32 + 3]; }
                    pred_block(mode, top, left, cb[3], dc_value(ydc_sum[l], hl, ht, 16), hl, ht, pred);
                    idct4_add(coef[l], pred, rec);
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) src[k] = srcb[(by * 4 + (k >> 2)) * 16 + bx * 4 + (k & 3)];
                    vD[l] = sse16(src, rec);
                    vSD[l] = iabs(hadamard_w(rec) - s[g].hsrc[i]) >> 5;
                    CSH_UNROLL
                    for (int r = 0; r < 4; r++) rec4[l][r] = pack4(rec, r);
                }
            }
            const uint64_t nzb = lballot([&](int l) { return vnz[l] != 0; });
            LV<int> vR;
            LFOR(l) {
                vR[l] = 0;
                if (ok[l]) {
                    MB_PROLOGUE;
                    const uint32_t cur = (uint32_t((nzb >> (l & 48)) & 0xFFFFu) << 1);
                    int type, first, ctx;
                    block_info(1 + i, cur, topm[l], leftm[l], false, type, first, ctx);
                    vR[l] = block_cost(T, type, first, ctx, lv[l]);
                    if (i == 0) { block_info(0, cur, topm[l], leftm[l], false, type, first, ctx); vR[l] += block_cost(T, type, first, ctx, lv2[l]); }
                }
            }
            const LV<int> sD = rowsum(vD), sSD = rowsum(vSD), sR = rowsum(vR);
            LFOR(l) if (ok[l]) {
                MB_PROLOGUE;
                const Vp8SegDev &S = T.seg[segv[l]];
                const uint32_t acnz = uint32_t((nzb >> (l & 48)) & 0xFFFFu);
                int64_t D = sD[l], SD = S.tlambda ? (int64_t(S.tlambda) * sSD[l] + 128) >> 8 : 0;
                if (srcflat[l]) { srcflat[l] = acnz == 0; if (srcflat[l]) { D *= 2; SD *= 2; } }   // a flat source whose levels are flat too: distortion counts double
                const int64_t RH = int64_t(sR[l]) + kVp8FixedCostsI16[mode], score = RH * S.lambda_i16 + 256 * (D + SD);
                if (mode == 0 || score < best_score[l]) {
                    best_score[l] = score; best16[l] = mode;
                    nz16[l] = int(acnz) | (vnz2[l] << 24);
                    D16[l] = int(D);
                    sc16[l] = RH * S.lambda_mode + 256 * (D + SD);
                    CSH_UNROLL
                    for (int k = 0; k < 8; k++) best_lv[l][k] = (lv[l][2 * k] & 0xFFFF) | (lv[l][2 * k + 1] << 16);
                    CSH_UNROLL
                    for (int r = 0; r < 4; r++) best_rec[l][r] = rec4[l][r];
                    int mine = 0;
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) mine = i == k ? lv2[l][k] : mine;
                    s[g].y2best[i] = int16_t(mine);
                }
            }
            CSP_WAVE_SYNC();
        }
        // only DCs, yet distorted: blocky -> remember the step for the loop filter (oracle: max_edge)
        LFOR(l) if (ok[l] && (l & 15) == 0) {
            MB_PROLOGUE;
            const Vp8SegDev &S = T.seg[segv[l]];
            if ((uint32_t(nz16[l]) & 0x100ffffu) == 0x1000000u && D16[l] > S.min_disto) {
                const int v0 = iabs(s[g].y2best[1]), v1 = iabs(s[g].y2best[2]), v2 = iabs(s[g].y2best[4]);
                int m = v1 > v0 ? v1 : v0;
                if (v2 > m) m = v2;
                atomicMax(&F->seg[segv[l]].max_edge, m);
            }
        }
    }

    // =========================================================================================== i4: sixteen sub-blocks in order, lane = mode
    LV<int> use4, live4, hbits;
    LV<int64_t> sc4;     // running i4 score (lambda_mode)
    {
        LFOR(l) {
            use4[l] = 0; live4[l] = ok[l]; hbits[l] = 0; sc4[l] = 0;
            if (ok[l]) sc4[l] = int64_t(211) * T.seg[segv[l]].lambda_mode;   // the cost of the "not i16" flag
        }
        for (int k = 0; k < 16; k++) {
            const int bx = k & 3, by = k >> 2;
            if (lballot([&](int l) { return live4[l] != 0; }) == 0) break;
            LV<uint64_t> key;
            LVA<int, 16> lv;
            LVA<uint32_t, 4> rec4;
            LV<int> vD, vSD, vR, vH, vnz;
            LFOR(l) if (live4[l] && (l & 15) < 13) {
                MB_PROLOGUE;
                const uint8_t *d = cb + (by * 4 + 1) * 32 + 4 + bx * 4;   // the sub-block's first sample
                s[g].e[i] = i < 4 ? d[(3 - i) * 32 - 1] : i == 4 ? d[-32 - 1] : i < 9 ? d[-32 + (i - 5)] : bx == 3 ? cb[20 + (i - 9)] : d[-32 + 4 + (i - 9)];
            }
            CSP_WAVE_SYNC();
            LFOR(l) {
                key[l] = ~0ull; vD[l] = 0; vSD[l] = 0; vR[l] = 0; vH[l] = 0; vnz[l] = 0;
                if (live4[l] && (l & 15) < 10) {
                    MB_PROLOGUE;
                    const int mode = i;
                    const Vp8SegDev &S = T.seg[segv[l]];
                    const uint8_t *e = s[g].e;
                    int pred[16], dd[16], c[16], src[16], rec[16];
                    if (mode == 0) {
                        const int v = (int(e[5]) + e[6] + e[7] + e[8] + e[3] + e[2] + e[1] + e[0] + 4) >> 3;
                        CSH_UNROLL
                        for (int t = 0; t < 16; t++) pred[t] = v;
                    } else if (mode == 1) {
                        CSH_UNROLL
                        for (int t = 0; t < 16; t++) pred[t] = clip8(int(e[3 - (t >> 2)]) + int(e[5 + (t & 3)]) - int(e[4]));
                    } else {
                        CSH_UNROLL
                        for (int t = 0; t < 16; t++) {
                            const uint32_t tp = kVp8Pred4Taps[(mode - 2) * 16 + t];
                            pred[t] = (int(e[tp & 15u]) + int(e[(tp >> 4) & 15u]) + int(e[(tp >> 8) & 15u]) + int(e[tp >> 12]) + 2) >> 2;
                        }
                    }
                    CSH_UNROLL
                    for (int t = 0; t < 16; t++) { src[t] = srcb[(by * 4 + (t >> 2)) * 16 + bx * 4 + (t & 3)]; dd[t] = src[t] - pred[t]; }
                    fdct4(dd, c);
                    vnz[l] = quant_block<true>(c, lv[l], load_qm(S, 0));
                    idct4_add(c, pred, rec);
                    vD[l] = sse16(src, rec);
                    vSD[l] = S.tlambda ? (S.tlambda * (iabs(hadamard_w(rec) - s[g].hsrc[k]) >> 5) + 128) >> 8 : 0;
                    CSH_UNROLL
                    for (int r = 0; r < 4; r++) rec4[l][r] = pack4(rec, r);
                    const int tmode = by ? int(s[g].bm[k - 4]) : (ht ? int((L - size_t(mbw) * WEBP_MB_REC)[MB_INFO + 4 + 12 + bx]) : 0);
                    const int lmode = bx ? int(s[g].bm[k - 1]) : (hl ? int((L - WEBP_MB_REC)[MB_INFO + 4 + by * 4 + 3]) : 0);
                    vH[l] = kVp8FixedCostsI4[(tmode * 10 + lmode) * 10 + mode];
                    int nzac = 0;
                    CSH_UNROLL
                    for (int t = 1; t < 16; t++) nzac += lv[l][t] != 0;
                    int R = (mode > 0 && nzac <= 3) ? 140 : 0;   // flatness penalty: a flat block should not be predicted by a complex mode
                    const int tctx = by ? int(s[g].nz4[k - 4]) : int((topm[l] >> (13 + bx)) & 1u), lctx = bx ? int(s[g].nz4[k - 1]) : int((leftm[l] >> (4 + by * 4)) & 1u);
                    R += block_cost(T, 3, 0, tctx + lctx, lv[l]);
                    vR[l] = R;
                    const int64_t score = int64_t(R + vH[l]) * S.lambda_i4 + 256 * int64_t(vD[l] + vSD[l]);
                    key[l] = (uint64_t(score) << 4) | uint64_t(mode);
                }
            }
            const LV<uint64_t> best = rowmin64(key);
            LFOR(l) if (live4[l] && key[l] == best[l] && key[l] != ~0ull) {
                MB_PROLOGUE;
                CSH_UNROLL
                for (int r = 0; r < 4; r++) s[g].cbw[(by * 4 + r + 1) * 8 + 1 + bx] = rec4[l][r];
                CSH_UNROLL
                for (int t = 0; t < 16; t++) s[g].lv4[k][t] = int16_t(lv[l][t]);
                s[g].bm[k] = uint8_t(i);
                s[g].nz4[k] = uint8_t(vnz[l]);
                s[g].win[0] = vD[l]; s[g].win[1] = vSD[l]; s[g].win[2] = vR[l]; s[g].win[3] = vH[l];
            }
            CSP_WAVE_SYNC();
            LFOR(l) if (live4[l]) {
                const int g = l >> 4;
                const Vp8SegDev &S = T.seg[segv[l]];
                sc4[l] += int64_t(s[g].win[2] + s[g].win[3]) * S.lambda_mode + 256 * int64_t(s[g].win[0] + s[g].win[1]);
                hbits[l] += s[g].win[3];
                if (sc4[l] >= sc16[l] || hbits[l] > 256 * 16 * 16) live4[l] = 0;
                else if (k == 15) use4[l] = 1;
            }
            CSP_WAVE_SYNC();
        }
    }

    // =========================================================================================== chroma: four modes, eight blocks; two modes to a pass
    LV<int> bestuv, nzuv;
    {
        LV<int64_t> best_score;
        LFOR(l) { bestuv[l] = -1; nzuv[l] = 0; best_score[l] = 0; }
        for (int pass = 0; pass < 2; pass++) {
            LVA<int, 16> c, pred;
            LFOR(l) if (ok[l]) {
                MB_PROLOGUE;
                const int mode = pass * 2 + (i >> 3), b = i & 7, pl = b >> 2, bx = b & 1, by = (b >> 1) & 1;
                const uint8_t *cs8 = reinterpret_cast<const uint8_t *>(s[g].csrcw[pl]);
                int top[4], left[4], d[16];
                CSH_UNROLL
                for (int k = 0; k < 4; k++) { top[k] = s[g].ce[pl][bx * 4 + k]; left[k] = s[g].ce[pl][8 + by * 4 + k]; }
                pred_block(mode, top, left, s[g].ce[pl][16], dc_value(pl ? vsum[l] : usum[l], hl, ht, 8), hl, ht, pred[l]);
                CSH_UNROLL
                for (int k = 0; k < 16; k++) d[k] = int(cs8[(by * 4 + (k >> 2)) * 8 + bx * 4 + (k & 3)]) - pred[l][k];
                fdct4(d, c[l]);
                s[g].cdc[(i >> 3) * 2 + pl][b & 3] = c[l][0];
            }
            CSP_WAVE_SYNC();
            // error diffusion over the 2 x 2 DCs of a plane: one lane per (mode, plane)
            LFOR(l) if (ok[l] && (l & 3) == 0) {
                MB_PROLOGUE;
                const int slot4 = i >> 2, pl = slot4 & 1;    // slot4 = (mode & 1) * 2 + plane
                if (F->diffuse) {
                    const QM m = load_qm(T.seg[segv[l]], 2);
                    const int16_t *Lt = L - size_t(mbw) * WEBP_MB_REC, *Ll = L - WEBP_MB_REC;
                    const int tp0 = ht ? int(int8_t(uint16_t(Lt[MB_DERR_TOP + pl]) & 255u)) : 0, tp1 = ht ? int(int8_t(uint16_t(Lt[MB_DERR_TOP + pl]) >> 8)) : 0;
                    const int lf0 = hl ? int(int8_t(uint16_t(Ll[MB_DERR_LEFT + pl]) & 255u)) : 0, lf1 = hl ? int(int8_t(uint16_t(Ll[MB_DERR_LEFT + pl]) >> 8)) : 0;
                    int *d = s[g].cdc[slot4];
                    int v0 = int(int16_t(d[0] + ((7 * tp0 + 8 * lf0) >> 3)));
                    const int e0 = quant_single(v0, m);
                    int v1 = int(int16_t(d[1] + ((7 * tp1 + 8 * e0) >> 3)));
                    const int e1 = quant_single(v1, m);
                    int v2 = int(int16_t(d[2] + ((7 * e0 + 8 * lf1) >> 3)));
                    const int e2 = quant_single(v2, m);
                    int v3 = int(int16_t(d[3] + ((7 * e1 + 8 * e2) >> 3)));
                    const int e3 = quant_single(v3, m);
                    d[0] = v0; d[1] = v1; d[2] = v2; d[3] = v3;
                    s[g].cerr[slot4][0] = int(int8_t(e1)); s[g].cerr[slot4][1] = int(int8_t(e2)); s[g].cerr[slot4][2] = int(int8_t(e3));   // kept until the best mode is known
                } else { s[g].cerr[slot4][0] = 0; s[g].cerr[slot4][1] = 0; s[g].cerr[slot4][2] = 0; }
            }
            CSP_WAVE_SYNC();
            LVA<int, 16> lv;
            LVA<uint32_t, 4> rec4;
            LV<int> vD, vnz, vac;
            LFOR(l) {
                vD[l] = 0; vnz[l] = 0; vac[l] = 0;
                if (ok[l]) {
                    MB_PROLOGUE;
                    const int b = i & 7, pl = b >> 2, bx = b & 1, by = (b >> 1) & 1;
                    const uint8_t *cs8 = reinterpret_cast<const uint8_t *>(s[g].csrcw[pl]);
                    int rec[16], src[16];
                    c[l][0] = s[g].cdc[(i >> 3) * 2 + pl][b & 3];
                    vnz[l] = quant_block<false>(c[l], lv[l], load_qm(T.seg[segv[l]], 2));
                    idct4_add(c[l], pred[l], rec);
                    CSH_UNROLL
                    for (int k = 0; k < 16; k++) src[k] = cs8[(by * 4 + (k >> 2)) * 8 + bx * 4 + (k & 3)];
                    vD[l] = sse16(src, rec);
                    CSH_UNROLL
                    for (int r = 0; r < 4; r++) rec4[l][r] = pack4(rec, r);
                    int nzac = 0;
                    CSH_UNROLL
                    for (int t = 1; t < 16; t++) nzac += lv[l][t] != 0;
                    vac[l] = nzac;
                }
            }
            const uint64_t nzb = lballot([&](int l) { return vnz[l] != 0; });
            LV<int> vR;
            LFOR(l) {
                vR[l] = 0;
                if (ok[l]) {
                    const int i = l & 15, b = i & 7;
                    const uint32_t cur = uint32_t((nzb >> ((l & 48) + (i & 8))) & 0xFFu) << 17;
                    int type, first, ctx;
                    block_info(17 + b, cur, topm[l], leftm[l], false, type, first, ctx);
                    vR[l] = block_cost(T, type, first, ctx, lv[l]);
                }
            }
            const LV<int> sD = halfsum(vD), sR = halfsum(vR), sAC = halfsum(vac);
            LV<int64_t> sc;
            LFOR(l) {
                sc[l] = 0;
                if (ok[l]) {
                    const int i = l & 15, mode = pass * 2 + (i >> 3);
                    int R = sR[l];
                    if (mode > 0 && sAC[l] <= 2) R += 140 * 8;
                    sc[l] = int64_t(R + kVp8FixedCostsUV[mode]) * T.seg[segv[l]].lambda_uv + 256 * int64_t(sD[l]);
                }
            }
            // the better of the pass's two modes (the lower mode on a tie), then against the best so far
            LV<int64_t> other;
            LFOR(l) {
#ifdef CSH_EMUL
                other[l] = sc.v[l ^ 8];
#else
                other[l] = int64_t((uint64_t(uint32_t(__shfl(int(uint32_t(uint64_t(sc.v) >> 32)), l ^ 8, 64))) << 32) | uint32_t(__shfl(int(uint32_t(uint64_t(sc.v))), l ^ 8, 64)));
#endif
            }
            LV<int> take;
            LFOR(l) {
                take[l] = 0;
                if (ok[l]) {
                    const int i = l & 15, mode = pass * 2 + (i >> 3);
                    const int64_t lo2 = (i & 8) ? other[l] : sc[l], hi2 = (i & 8) ? sc[l] : other[l];   // scores of modes 2 pass and 2 pass + 1
                    const int pick = hi2 < lo2 ? 1 : 0;
                    const int64_t psc = pick ? hi2 : lo2;
                    const int pmode = pass * 2 + pick;
                    if (bestuv[l] < 0 || psc < best_score[l]) {
                        const uint32_t rownz = uint32_t((nzb >> (l & 48)) & 0xFFFFu);
                        best_score[l] = psc; bestuv[l] = pmode; take[l] = mode == pmode;
                        nzuv[l] = int(pick ? (rownz >> 8) & 0xFFu : rownz & 0xFFu);
                    }
                }
            }
            LFOR(l) if (ok[l] && take[l]) {
                const int g = l >> 4, i = l & 15, b = i & 7, pl = b >> 2, bx = b & 1, by = (b >> 1) & 1;
                CSH_UNROLL
                for (int t = 0; t < 16; t++) s[g].lvuv[b][t] = int16_t(lv[l][t]);
                CSH_UNROLL
                for (int r = 0; r < 4; r++) s[g].recuv[pl][(by * 4 + r) * 2 + bx] = rec4[l][r];
                if ((b & 3) == 0) { const int slot4 = (i >> 3) * 2 + pl; s[g].derr[pl][0] = int8_t(s[g].cerr[slot4][0]); s[g].derr[pl][1] = int8_t(s[g].cerr[slot4][1]); s[g].derr[pl][2] = int8_t(s[g].cerr[slot4][2]); }
            }
            CSP_WAVE_SYNC();
        }
    }

    // =========================================================================================== commit: levels, modes, flags, reconstruction
    LFOR(l) if (ok[l]) {
        MB_PROLOGUE;
        const bool i4 = use4[l] != 0;
        uint8_t *ry = work + im.ry_off + size_t(my * 16) * ys + mx * 16;
        // luma reconstruction and levels: lane = block (i16) / the sub-blocks' buffer (i4)
        {
            const int bx = i & 3, by = i >> 2;
            CSH_UNROLL
            for (int r = 0; r < 4; r++) *reinterpret_cast<uint32_t *>(ry + size_t(by * 4 + r) * ys + bx * 4) = i4 ? s[g].cbw[(by * 4 + r + 1) * 8 + 1 + bx] : best_rec[l][r];
            uint32_t *o = reinterpret_cast<uint32_t *>(L + (1 + i) * 16);
            const uint32_t *o4 = reinterpret_cast<const uint32_t *>(s[g].lv4[i]);
            CSH_UNROLL
            for (int t = 0; t < 8; t++) o[t] = i4 ? o4[t] : uint32_t(best_lv[l][t]);
            L[i] = i4 ? int16_t(0) : s[g].y2best[i];
        }
        // chroma: lanes 0..7 a block's levels; every lane two words of the reconstruction
        if (i < 8) {
            uint32_t *o = reinterpret_cast<uint32_t *>(L + (17 + i) * 16);
            const uint32_t *ouv = reinterpret_cast<const uint32_t *>(s[g].lvuv[i]);
            CSH_UNROLL
            for (int t = 0; t < 8; t++) o[t] = ouv[t];
        }
        {
            const int pl = i >> 3, r = i & 7;
            uint8_t *rc = work + (pl ? im.rv_off : im.ru_off) + size_t(my * 8 + r) * cs + mx * 8;
            *reinterpret_cast<uint32_t *>(rc) = s[g].recuv[pl][r * 2];
            *reinterpret_cast<uint32_t *>(rc + 4) = s[g].recuv[pl][r * 2 + 1];
        }
        int16_t *I = L + MB_INFO;
        I[4 + i] = int16_t(i4 ? int(s[g].bm[i]) : best16[l]);
        if (i == 0) {
            uint32_t luma = 0;
            if (i4) { for (int k = 0; k < 16; k++) luma |= uint32_t(s[g].nz4[k] ? 1u : 0u) << k; } else luma = uint32_t(nz16[l]) & 0xFFFFu;
            const uint32_t y2 = (uint32_t(nz16[l]) >> 24) & 1u, left_y2 = leftm[l] & 1u, top_y2 = (topm[l] >> 25) & 1u;
            const uint32_t mask = (i4 ? left_y2 : y2) | (luma << 1) | ((uint32_t(nzuv[l]) & 0xFFu) << 17) | ((i4 ? top_y2 : y2) << 25);
            I[0] = int16_t(mask & 0xFFFFu); I[1] = int16_t(mask >> 16); I[2] = int16_t(i4 ? 4 : best16[l]); I[3] = int16_t(bestuv[l]);
            I[21] = int16_t(segv[l]);
            // chroma DC errors handed on: e1 to the right, e2 below, e3 split 3/4 right, 1/4 below
            for (int pl = 0; pl < 2; pl++) {
                const int e1 = s[g].derr[pl][0], e2 = s[g].derr[pl][1], e3 = s[g].derr[pl][2];
                const int l0 = e1, l1 = int(int8_t((3 * e3) >> 2)), t0 = e2, t1b = int(int8_t(e3 - l1));
                I[22 + pl] = int16_t(uint16_t(uint8_t(t0)) | (uint16_t(uint8_t(t1b)) << 8));
                I[24 + pl] = int16_t(uint16_t(uint8_t(l0)) | (uint16_t(uint8_t(l1)) << 8));
            }
        }
    }
#undef MB_PROLOGUE
}

// =================================================================================================== C': statistics between chunks
// the statistics walk of one block: every adaptive decision counted in the group's LDS counters (events and ones), and the block's number of decisions
// (adaptive and fixed) for the decision stream of the coder back end
struct StatSink {
    uint32_t *cnt, *ones;
    uint32_t nd;
    __device__ __forceinline__ void ad(int bit, int idx) { atomicAdd(&cnt[idx], 1u); if (bit) atomicAdd(&ones[idx], 1u); nd++; }
    __device__ __forceinline__ void ad10(int bit, int idx) { ad(bit, idx - 1); }   // libwebp's books: the second bit of the two big categories counts for the slot before
    __device__ __forceinline__ void fx(int, int) { nd++; }
};
// the same walk watching ONE slot in order: events, ones, and the ones among the first `limit` events
struct WatchSink {
    int slot;
    uint32_t t, o, limit, o_lim;
    __device__ __forceinline__ void ad(int bit, int idx) { if (idx == slot) { if (t < limit) o_lim += uint32_t(bit); t++; o += uint32_t(bit); } }
    __device__ __forceinline__ void ad10(int bit, int idx) { ad(bit, idx - 1); }
    __device__ __forceinline__ void fx(int, int) {}
};
__device__ __forceinline__ static uint32_t sink_nd(const StatSink &s) { return s.nd; }
__device__ __forceinline__ static uint32_t sink_nd(const WatchSink &) { return 0; }
template <class S>
__device__ static void walk_mb(S &sink, const int16_t *L, int mbw, int mx, int my, uint16_t *blk_cnt) {
    const uint32_t cur = nz_mask(L), top = my ? nz_mask(L - size_t(mbw) * WEBP_MB_REC) : 0u, left = mx ? nz_mask(L - WEBP_MB_REC) : 0u;
    const bool i4 = L[MB_INFO + 2] == 4;
    for (int k = i4 ? 1 : 0; k < 25; k++) {
        int type, first, ctx;
        block_info(k, cur, top, left, i4, type, first, ctx);
        const uint32_t before = sink_nd(sink);
        put_coeffs(sink, type, ctx, L + k * 16, first);
        if (blk_cnt) blk_cnt[k] = uint16_t(sink_nd(sink) - before);
    }
    if (blk_cnt && i4) blk_cnt[0] = 0;
}

struct StatLds { uint32_t cnt[VP8_NSLOT], ones[VP8_NSLOT]; };
// macroblocks n0 .. n1 of one picture (a chunk), one wave.  mb_cnt / blk_cnt (already offset to the picture's first macroblock): decisions per macroblock / block (32 slots
// a macroblock).  Leaves the frame's probabilities so far in T.coeffs and, when any differs from the defaults, the level-cost tables made from them in T.lc
__device__ __forceinline__ static void chunk_stats(StatLds &A, MbTables &T, const WebpImg &im, Vp8FrameDev *F, const int16_t *lev, int n0, int n1, bool final_chunk, uint32_t *mb_cnt, uint16_t *blk_cnt) {
    uint32_t *s_cnt = A.cnt, *s_ones = A.ones;
    uint8_t *s_coeffs = T.coeffs;
    const int mbw = int(im.mbw);
    for (int base = n0; base < n1; base += 64) {
        LFOR(l) for (int i = l; i < VP8_NSLOT; i += 64) { s_cnt[i] = 0; s_ones[i] = 0; }
        CSP_WAVE_SYNC();
        LFOR(l) {
            const int n = base + l;
            if (n < n1) {
                const int my = n / mbw, mx = n - my * mbw;
                StatSink sink{s_cnt, s_ones, 0u};
                walk_mb(sink, lev + size_t(n) * WEBP_MB_REC, mbw, mx, my, blk_cnt + size_t(n) * 32);
                mb_cnt[n] = sink.nd;
            }
        }
        CSP_WAVE_SYNC();
        // into libwebp's 16-bit books; a slot that would pass 0xfffe events inside this group is recounted in order below (lane l looks after the slots
        // l, l + 64, ..: bit k of `pend` = its k-th slot is waiting for that)
        LV<uint32_t> pend;
        LFOR(l) {
            pend[l] = 0;
            for (int i = l, k = 0; i < VP8_NSLOT; i += 64, k++) {
                const uint32_t cn = s_cnt[i];
                if (!cn) continue;
                const uint32_t p = coherent_load(&F->stats[i]);
                if ((p >> 16) + cn <= 0xfffeu) coherent_store(&F->stats[i], p + (cn << 16) + s_ones[i]);
                else pend[l] |= 1u << k;
            }
        }
        for (;;) {
            const uint64_t waiting = lballot([&](int l) { return pend[l] != 0; });
            if (!waiting) break;
            const uint32_t lane = uint32_t(__builtin_ctzll(waiting)), bits = csh::lget(pend, lane), kbit = uint32_t(__builtin_ctz(bits));
            csh::lset(pend, lane, bits & (bits - 1u));
            const int slot = int(lane + 64u * kbit);
            const uint32_t p = coherent_load(&F->stats[slot]), k = 0xfffeu - (p >> 16);   // the halving comes after k more events
            LV<uint32_t> tl, ol;
            LFOR(l) {
                tl[l] = 0; ol[l] = 0;
                const int n = base + l;
                if (n < n1) {
                    const int my = n / mbw, mx = n - my * mbw;
                    WatchSink w{slot, 0u, 0u, 0xFFFFFFFFu, 0u};
                    walk_mb(w, lev + size_t(n) * WEBP_MB_REC, mbw, mx, my, nullptr);
                    tl[l] = w.t; ol[l] = w.o;
                }
            }
            uint32_t tsum, osum;
            const LV<uint32_t> tex = lscan(tl, tsum);
            (void)lscan(ol, osum);
            LV<uint32_t> part;
            LFOR(l) {
                part[l] = 0;
                if (tex[l] + tl[l] <= k) part[l] = ol[l];                 // wholly before the halving point
                else if (tex[l] < k) {                                      // the macroblock it falls into: its first k - tex events
                    const int n = base + l, my = n / mbw, mx = n - my * mbw;
                    WatchSink w{slot, 0u, 0u, k - tex[l], 0u};
                    walk_mb(w, lev + size_t(n) * WEBP_MB_REC, mbw, mx, my, nullptr);
                    part[l] = w.o_lim;
                }
            }
            uint32_t before;
            (void)lscan(part, before);
            LFOR(l) if (l == 0) {
                const uint32_t nk = (p & 0xffffu) + before, halved = ((nk + 1u) >> 1) & 0x7fffu;
                coherent_store(&F->stats[slot], ((0x7fffu + (tsum - k)) << 16) | (halved + (osum - before)));
            }
        }
        CSP_WAVE_SYNC();
    }
    // the frame's probabilities from the books so far (oracle: finalize_probas); cost tables for the next chunk when any differs from the defaults
    LV<int> chg;
    LFOR(l) {
        chg[l] = 0;
        for (int i = l; i < VP8_NSLOT; i += 64) {
            const uint32_t st = coherent_load(&F->stats[i]);
            const int nb = int(st & 0xffffu), total = int(st >> 16), up = kVp8CoefUpdateProbs[i], oldp = kVp8CoefProbs[i];
            const int newp = nb ? 255 - nb * 255 / total : 255;
            const int old_cost = nb * vp8_bitcost(1, oldp) + (total - nb) * vp8_bitcost(0, oldp) + vp8_bitcost(0, up);
            const int new_cost = nb * vp8_bitcost(1, newp) + (total - nb) * vp8_bitcost(0, newp) + vp8_bitcost(1, up) + 8 * 256;
            const int pnew = old_cost > new_cost ? newp : oldp;
            chg[l] |= pnew != oldp;
            s_coeffs[i] = uint8_t(pnew);
            F->coeffs[i] = uint8_t(pnew);
        }
    }
    CSP_WAVE_SYNC();
    const bool dirty = lballot([&](int l) { return chg[l] != 0; }) != 0;
    if (dirty) make_level_costs(T.lc, s_coeffs);
    if (!final_chunk) return;
    // after the last macroblock: blocky DC-only macroblocks ask for at least this much filtering (oracle: the max_edge loop)
    LFOR(l) if (l == 0) {
        int m = 0;
        for (int sg = 0; sg < 4; sg++) {
            Vp8SegDev &S = F->seg[sg];
            const int delta = (S.max_edge * S.q[1][1]) >> 3, level = kVp8LevelsFromDelta[delta > 63 ? 63 : delta];
            if (level > S.fstrength) S.fstrength = level;
            if (m < S.fstrength) m = S.fstrength;
        }
        F->filter_level = m;
    }
}

// =================================================================================================== C + C': one workgroup walks one picture
// Two waves per picture; phase p of the loop is step p - 1 of the picture's plan (phase 0 sets the tables up): in a macroblock step wave w takes the macroblocks 4w .. 4w + 3
// (then 8 + 4w .., for steps wider than eight), in a statistics step wave 0 brings the books, the probabilities and the level-cost tables up to date.  The tables live in
// LDS for the whole walk; what a step leaves for the next (levels, modes, flags, reconstruction) goes through global memory, fenced at the barrier between the phases.
// Pictures walk independently of each other: no launch per step, no lock-step between the pictures of a batch.
union WaveLds { MbLds mb[4]; StatLds st; };
__global__ void __launch_bounds__(2 * CSP_WAVE_THREADS, 2) k_vp8_loop(const WebpImg *imgs, const Vp8Class *classes, const uint32_t *steps, const uint32_t *items, const uint32_t *chunk_step,
                                                                    const uint32_t *chunk_end, uint8_t *work, int16_t *levels, Vp8FrameDev *frames, const uint64_t *mb_base, uint32_t *mb_cnt,
                                                                    uint16_t *blk_cnt, int nph) {
    CSH_SHARED MbTables T;
    CSH_SHARED WaveLds s[2];
    const WebpImg im = imgs[blockIdx.x];
    const Vp8Class cls = classes[im.cls];
    Vp8FrameDev *F = frames + blockIdx.x;
    const int wave = int(threadIdx.x / CSP_WAVE_THREADS);
    CSH_PHASE_LOOP(nph) {
        CSP_ACQUIRE_FENCE();
        if (phase == 0) {
            if (wave == 0) {
                LFOR(l) {
                    for (int i = l; i < VP8_NSLOT; i += 64) T.coeffs[i] = kVp8CoefProbs[i];
                    for (int i = l; i < 256; i += 64) T.ent[i] = kVp8EntropyCost[i];
                    for (int i = l; i < 2048; i += 64) T.fixed[i] = kVp8LevelFixedCosts[i];
                    for (int i = l; i < int(sizeof(T.seg) / 4); i += 64) reinterpret_cast<uint32_t *>(T.seg)[i] = reinterpret_cast<const uint32_t *>(F->seg)[i];
                }
                make_level_costs(T.lc, kVp8CoefProbs);
            }
        } else if (uint32_t(phase - 1) < cls.nsteps) {
            const uint32_t step = uint32_t(phase - 1), lo = steps[cls.step_off + step], hi = steps[cls.step_off + step + 1];
            if (lo < hi) {
                for (uint32_t at = lo + uint32_t(wave) * 4u; at < hi; at += 8u) mb_batch(s[wave].mb, T, im, F, items + cls.item_off, at, hi, work, levels);
            } else if (wave == 0) {
                int j = -1;
                for (uint32_t q = 0; q < cls.nchunks; q++) if (chunk_step[cls.chunk_off + q] == step) j = int(q);
                if (j >= 0) {
                    const int n0 = j ? int(chunk_end[cls.chunk_off + uint32_t(j) - 1]) : 0, n1 = int(chunk_end[cls.chunk_off + uint32_t(j)]);
                    chunk_stats(s[1].st, T, im, F, levels + im.lev_off, n0, n1, uint32_t(j) + 1 == cls.nchunks, mb_cnt + mb_base[blockIdx.x], blk_cnt + mb_base[blockIdx.x] * 32);
                }
            }
        }
        CSP_MEM_FENCE();
    }
}

// =================================================================================================== the host's plan and the launches
// pictures of one size share a plan: T(mb) = the step in which it can run = 1 + max over its left and upper-right (upper in the last column) neighbours, and
// not before the statistics step of its chunk, which follows the last macroblock of the chunk before
struct Vp8Plan { std::vector<Vp8Class> classes; std::vector<uint32_t> steps, items, chunk_step, chunk_end; std::vector<uint32_t> width_at, stats_at; uint32_t nsteps = 0; };
static uint32_t plan_class(Vp8Plan &P, uint32_t mbw, uint32_t mbh) {
    for (size_t i = 0; i < P.classes.size(); i++) if (P.classes[i].mbw == mbw && P.classes[i].mbh == mbh) return uint32_t(i);
    const uint32_t nmb = mbw * mbh, max_count = std::max<uint32_t>(96u, nmb >> 3);
    std::vector<int> T(nmb);
    Vp8Class c{};
    c.mbw = mbw; c.mbh = mbh; c.step_off = uint32_t(P.steps.size()); c.item_off = uint32_t(P.items.size()); c.chunk_off = uint32_t(P.chunk_step.size());
    int maxT = -1, base = 0;
    uint32_t next_refresh = max_count;
    for (uint32_t n = 0; n < nmb; n++) {
        const uint32_t mx = n % mbw, my = n / mbw;
        if (n == next_refresh) { P.chunk_step.push_back(uint32_t(maxT + 1)); P.chunk_end.push_back(n); base = maxT + 2; next_refresh = n + max_count + 1; c.nchunks++; }
        int t = base;
        if (mx) t = std::max(t, T[n - 1] + 1);
        if (my) t = std::max(t, T[n - mbw + (mx + 1 < mbw ? 1 : 0)] + 1);
        T[n] = t;
        maxT = std::max(maxT, t);
    }
    P.chunk_step.push_back(uint32_t(maxT + 1)); P.chunk_end.push_back(nmb); c.nchunks++;
    c.nsteps = uint32_t(maxT + 2);
    std::vector<uint32_t> count(c.nsteps + 1, 0);
    for (uint32_t n = 0; n < nmb; n++) count[size_t(T[n]) + 1]++;
    for (uint32_t t = 0; t < c.nsteps; t++) count[t + 1] += count[t];
    for (uint32_t t = 0; t <= c.nsteps; t++) P.steps.push_back(count[t]);
    std::vector<uint32_t> at(count.begin(), count.end() - 1), it(nmb);
    for (uint32_t n = 0; n < nmb; n++) it[at[size_t(T[n])]++] = (n % mbw) | ((n / mbw) << 16);
    P.items.insert(P.items.end(), it.begin(), it.end());
    if (P.width_at.size() < c.nsteps) { P.width_at.resize(c.nsteps, 0); P.stats_at.resize(c.nsteps, 0); }
    for (uint32_t t = 0; t < c.nsteps; t++) P.width_at[t] = std::max(P.width_at[t], count[t + 1] - count[t]);
    for (uint32_t q = 0; q < c.nchunks; q++) P.stats_at[P.chunk_step[c.chunk_off + q]] = 1;
    P.nsteps = std::max(P.nsteps, c.nsteps);
    P.classes.push_back(c);
    return uint32_t(P.classes.size() - 1);
}

// libwebp's quality -> quantiser index per segment alpha (VP8SetSegmentParams): two pow() calls in double, evaluated here with the C library as libwebp does
static void quality_table(int quality, uint8_t *tab) {
    const int SNS = 50;
    const float qf = float(quality < 0 ? 0 : quality > 100 ? 100 : quality);
    const double amp = 0.9 * SNS / 100. / 128., Q = qf / 100.;
    const double lin = Q < 0.75 ? Q * (2. / 3.) : 2. * Q - 1., c_base = pow(lin, 1 / 3.);
    for (int a = -127; a <= 127; a++) {
        const double expn = 1. - amp * a, c = pow(c_base, expn);
        const int q = int(127. * (1. - c));
        tab[a + 127] = uint8_t(q < 0 ? 0 : q > 127 ? 127 : q);
    }
    tab[255] = 0;
}

int launch_webp_encode(hipStream_t st, WebpImg *himgs, int nimg, WebpImg *d_imgs, uint8_t *work, int16_t *levels, uint8_t *scratch, uint32_t *part_size, uint8_t *out,
                       uint32_t *img_size, uint32_t *status, hipEvent_t mid) {
    if (!nimg) return 0;
    Vp8Plan P;
    std::vector<int> quals;
    std::vector<uint8_t> qtabs;
    std::vector<uint64_t> base(size_t(nimg) + 1);
    uint64_t nmb = 0;
    uint32_t max_nmb = 0;
    for (int i = 0; i < nimg; i++) {
        WebpImg &wi = himgs[i];
        wi.cls = plan_class(P, wi.mbw, wi.mbh);
        size_t q = 0;
        for (; q < quals.size(); q++) if (quals[q] == wi.quality) break;
        if (q == quals.size()) { quals.push_back(wi.quality); qtabs.resize(qtabs.size() + 256); quality_table(wi.quality, qtabs.data() + q * 256); }
        wi.qtab = uint32_t(q);
        base[size_t(i)] = nmb;
        nmb += uint64_t(wi.mbw) * wi.mbh;
        max_nmb = std::max(max_nmb, wi.mbw * wi.mbh);
    }
    base[size_t(nimg)] = nmb;
    DevBuf<Vp8FrameDev> d_frames;
    DevBuf<Vp8Class> d_classes;
    DevBuf<uint32_t> d_steps, d_items, d_cstep, d_cend, d_cnt;
    DevBuf<uint8_t> d_qtabs;
    DevBuf<uint64_t> d_base;
    DevBuf<uint16_t> d_blk;
    CSH_CHECK(hipMemcpyAsync(d_imgs, himgs, size_t(nimg) * sizeof(WebpImg), hipMemcpyHostToDevice, st));
    if (d_frames.alloc(size_t(nimg)) || d_frames.zero(st) || d_classes.upload(P.classes, st) || d_steps.upload(P.steps, st) || d_items.upload(P.items, st) ||
        d_cstep.upload(P.chunk_step, st) || d_cend.upload(P.chunk_end, st) || d_qtabs.upload(qtabs, st) || d_base.upload(base, st) || d_cnt.alloc(2 * nmb + size_t(nimg) + 2) || d_blk.alloc((nmb + 1) * 32))
        return -1;
    CSH_LAUNCH(k_vp8_analyse, dim3((max_nmb + 3) / 4, unsigned(nimg)), dim3(CSP_WAVE_THREADS), st, d_imgs, work, levels, d_frames.p);
    CSH_LAUNCH(k_vp8_segments, dim3(unsigned(nimg)), dim3(CSP_WAVE_THREADS), st, d_imgs, d_frames.p, d_qtabs.p);
    CSH_LAUNCH_PHASED(k_vp8_loop, int(P.nsteps) + 1, dim3(unsigned(nimg)), dim3(2 * CSP_WAVE_THREADS), st, d_imgs, d_classes.p, d_steps.p, d_items.p, d_cstep.p, d_cend.p, work, levels, d_frames.p,
                      d_base.p, d_cnt.p, d_blk.p, int(P.nsteps) + 1);
    if (mid) CSH_CHECK(hipEventRecord(mid, st));
    return launch_webp_backend(st, d_imgs, himgs, nimg, levels, d_frames.p, base, d_base.p, d_cnt, d_blk.p, scratch, part_size, out, img_size, status);
}

}  // namespace csw
