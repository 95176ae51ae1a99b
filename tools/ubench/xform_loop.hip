// xform_loop.hip -- the block arithmetic of k_xform_direct (dequantise, IDCT, clamp, FDCT, quantise: the functions of k_pixel.hip themselves) run
// REPS times per lane from L2-resident tiles: what the arithmetic costs per block when no wave waits for HBM.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../caesium-clt_amd/csrc -o /tmp/xform_loop xform_loop.hip
#include "../../caesium-clt_amd/csrc/k_pixel.hip"
#include <vector>
using namespace csh;

template <bool DERING>
__global__ void __launch_bounds__(256) k_loop(const DevQuant *__restrict__ quant, const int16_t *__restrict__ coef_in, int16_t *__restrict__ coef_out, int reps, int ntiles) {
    CSH_SHARED int16_t s_dr[64][256];
    int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < reps; r++) {
        const int t = (tile + r * 977) % ntiles;
        int x[64];
        load_idct<true>(coef_in + coef_index(t, lane, 0), quant[0], x);
        fdct_quant_store<DERING, true>(x, quant[1], coef_out + coef_index(t, lane, 0), nullptr, s_dr);
    }
}

int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int ntiles = 2048;   // 16 MiB of coefficient tiles: L2 / MALL resident
    std::vector<int16_t> h(size_t(ntiles) * CSH_TILE_I16);
    uint32_t s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = int16_t((s >> 24) % 7) - 3; }
    for (int t = 0; t < ntiles; t++) for (int b = 0; b < 64; b++) h[coef_index(t, b, 0)] = int16_t((b * 7 + t) % 200 - 100);
    DevQuant q[2];
    for (int k = 0; k < 64; k++) {
        q[0].q[k] = 2 + k / 8; q[1].q[k] = 4 + k / 4;
        for (int j = 0; j < 2; j++) { q[j].div[k] = q[j].q[k] * 8; q[j].rcp[k] = float((1.0 / double(q[j].div[k])) * (1.0 + 1.0 / 524288.0)); q[j].mul[k] = 0; q[j].sh[k] = 0; q[j].lt[k] = 0; }
    }
    int16_t *din, *dout; DevQuant *dq;
    (void)hipMalloc(&din, h.size() * 2); (void)hipMalloc(&dout, h.size() * 2); (void)hipMalloc(&dq, sizeof(q));
    (void)hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dq, q, sizeof(q), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double mhz = prop.clockRate / 1000.0;
    for (int dering = 0; dering < 2; dering++)
        for (int wg_per_cu : {1, 2, 3, 4, 5}) {
            const int blocks = prop.multiProcessorCount * wg_per_cu;
            float ms[2];
            for (int pass = 0; pass < 2; pass++) {
                const int reps = pass ? 40 : 8;
                float best = 1e30f;
                for (int it = 0; it < 3; it++) {
                    (void)hipEventRecord(e0);
                    if (dering) hipLaunchKernelGGL(k_loop<true>, dim3(blocks), dim3(256), 0, 0, dq, din, dout, reps, ntiles);
                    else hipLaunchKernelGGL(k_loop<false>, dim3(blocks), dim3(256), 0, 0, dq, din, dout, reps, ntiles);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float m; (void)hipEventElapsedTime(&m, e0, e1);
                    if (m < best) best = m;
                }
                ms[pass] = best;
            }
            const double cyc = (ms[1] - ms[0]) * 1e-3 * mhz * 1e6 / 32.0;   // per repetition, all waves of a SIMD side by side
            printf("dering=%d waves/SIMD=%d: %.0f cycles per repetition = %.0f cycles per block-wave per SIMD\n", dering, wg_per_cu, cyc, cyc / wg_per_cu);
        }
    return 0;
}
