// Issue rate of v_cvt_pk_f16_f32 against v_cvt_f16_f32 x 2 + v_pack_b32_f16 on gfx950 (one wave, eight independent chains).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/cvt_rate.hip -o /tmp/cvt_rate && /tmp/cvt_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 64
template <int MODE> __global__ void k(float* io, long long* cyc) {
    float a[8], b[8];
    unsigned r[8];
    for (int i = 0; i < 8; ++i) { a[i] = io[threadIdx.x + 64 * i]; b[i] = io[threadIdx.x + 64 * (i + 8)]; r[i] = 0; }
    const long long t0 = clock64();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int u = 0; u < N / 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned o;
                if (MODE == 0) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(a[i]), "v"(b[i]));
                else if (MODE == 1) {
                    unsigned x, y;
                    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(x) : "v"(a[i]));
                    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(y) : "v"(b[i]));
                    asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(o) : "v"(x), "v"(y));
                } else if (MODE == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(o) : "v"(a[i]), "v"(b[i]));
                else asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(o) : "v"(a[i]), "v"(b[i]));
                r[i] ^= o;
            }
        }
    }
    const long long t1 = clock64();
    unsigned x = 0;
    for (int i = 0; i < 8; ++i) x ^= r[i];
    io[threadIdx.x] = (float)x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* io; long long* c;
    hipMalloc(&io, 64 * 16 * 4); hipMalloc(&c, 8);
    hipMemset(io, 0, 64 * 16 * 4);
    const char* names[4] = {"v_cvt_pk_f16_f32", "2 x v_cvt_f16_f32 + v_pack_b32_f16", "v_cvt_pkrtz_f16_f32", "v_pk_max_i16 (reference: a full-rate VOP3P)"};
    for (int m = 0; m < 4; ++m) {
        long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, io, c);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, io, c);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, io, c);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, io, c);
            hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        }
        // clock64 = s_memtime: 100 MHz constant clock on gfx9?  report raw ticks per packed result and let the reference row scale it
        printf("%-48s %8.3f ticks per packed pair (incl. the xor)\n", names[m], (double)h / (1000.0 * N));
    }
    return 0;
}
