// Sustained MFMA issue rate of this MI355X under its power limit: every wave issues v_mfma_f32_16x16x32_f16 from
// registers only (no LDS, no memory).  Prices the conv kernels against what the matrix pipe actually sustains rather
// than the 2.5 PFLOP/s (2.4 GHz) data-sheet figure.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_spin(float* out, int iters) {
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc : {4, 8, 16}) {                      // waves per CU: 1, 2, 4 per SIMD
        const int threads = wpc >= 8 ? 512 : 256, blocks = 256 * wpc * 64 / threads;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_spin<16>, dim3(blocks), dim3(threads), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * (threads / 64) * iters * 16 * 16384.0;
            if (rep == 2) printf("waves/CU %2d: %.1f ms, %.0f TFLOP/s (%.2f of 2500), implied clock at 1024 FLOP/clk/SIMD: %.2f GHz\n", wpc, ms,
                                 flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0, flop / ms * 1e-9 * 1e12 / (256.0 * 4 * 1024) * 1e-9);
        }
    }
    return 0;
}
