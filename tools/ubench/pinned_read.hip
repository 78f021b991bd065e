// How fast does the CPU read memory a kernel has just written through a pinned host pointer?  (the tracker's per-frame cost rows)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/pinned_read.hip -o /tmp/pinned_read && /tmp/pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void fill(float* p, int n, float v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int n = 160 * 31 * 3;                       // one frame's three [T, N] matrices
    std::vector<float> dst(n);
    const unsigned flags[3] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocCoherent};
    const char* names[3] = {"default", "non-coherent", "coherent"};
    for (int f = 0; f < 3; ++f) {
        float* h = nullptr;
        if (hipHostMalloc((void**)&h, n * 4, flags[f]) != hipSuccess) { printf("%s: alloc failed\n", names[f]); continue; }
        float* d = nullptr;
        hipHostGetDevicePointer((void**)&d, h, 0);
        double tk = 0, tc = 0;
        for (int it = 0; it < 200; ++it) {
            const double t0 = now();
            hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, (float)it);
            hipStreamSynchronize(0);
            const double t1 = now();
            std::memcpy(dst.data(), h, n * 4);
            const double t2 = now();
            if (dst[5] != (float)it + 5) { printf("%s: stale data at iteration %d\n", names[f], it); break; }
            if (it >= 20) tk += t1 - t0, tc += t2 - t1;
        }
        printf("%-13s kernel + sync %.1f us, CPU copy of %d KB %.2f us (%.1f GB/s)\n", names[f], 1e6 * tk / 180, n * 4 / 1024, 1e6 * tc / 180, n * 4 / (tc / 180) / 1e9);
        hipHostFree(h);
    }
    return 0;
}
