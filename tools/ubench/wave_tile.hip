// What can ONE wave per SIMD sustain when it feeds its own MFMAs?  (VERDICT r4 #2b: "a one-wave-per-SIMD 128 x 128 per-wave tile with the
// 256 accumulators in AGPRs" against the ping-pong kernels' two waves per SIMD with 128 x 64 tiles.)
// A K-step of the candidate kernel, stripped of everything but its instruction mix: 64 v_mfma_f32_16x16x32_f16 (8 x 8 tiles of 16 x 16,
// 256 accumulator registers), NREAD ds_read_b128 of the NEXT step's fragments interleaved with them, NDMA global_load_lds_dwordx4 (the
// step's weights / patch pass, L2-resident source), one counted s_waitcnt vmcnt + one s_barrier.  Random operands (real power draw).
// Reported per (NREAD, NDMA): shader cycles per K-step (1 024 = the matrix pipe's own time), TFLOP/s over all CUs, implied clock.
// The same loop with the tile of the ping-pong kernels (MT 8 x NT 4, 8 waves per block, no ping-pong schedule -- all waves in step) is
// printed beside it as the baseline an unsophisticated 2-waves-per-SIMD kernel gets.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wave_tile.hip -o tools/ubench/wave_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// MT x NT tiles per wave, NW waves per block, NREAD fragment reads and NDMA LDS-DMA issues per K-step
template <int MT, int NT, int NW, int NDMA, bool PIPE>
__global__ __launch_bounds__(NW * 64) void step_kernel(const _Float16* __restrict__ g, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    // fill LDS with data (the DMA below keeps refreshing one region of it)
    for (int i = t; i < 96 * 1024 / 16; i += NW * 64) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(g)[i & 4095];
    __syncthreads();
    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 xf[2][MT], wf[2][NT];
    // fragment addresses: conflict-free 256-byte runs per 16-lane group, different tiles 1 KB apart
    const int fa = (lane & 15) * 16 + (lane >> 4) * 4096 + wv * 64;
#pragma unroll
    for (int i = 0; i < MT; ++i) xf[0][i] = *reinterpret_cast<const half8*>(smem + ((fa + i * 1024) & 0xffff));
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[0][j] = *reinterpret_cast<const half8*>(smem + 65536 + ((fa + j * 1024) & 0x7fff));
    const _Float16* src = g + (size_t)(blockIdx.x & 63) * 32768 + t * 8;
    char* const dst = smem + 98304 + wv * 1024;
    unsigned long long c0 = 0;
    if (t == 0) c0 = clock64();
    auto step = [&](int it, auto curc) {
        constexpr int cur = PIPE ? decltype(curc)::value : 0, nxt = PIPE ? cur ^ 1 : 0;
        const int so = (it & 3) * 2048;
        if constexpr (!PIPE) {      // the ping-pong kernels' LOAD segment without a partner: reads, DMA, wait, barrier, THEN the MFMAs
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[0][i] = *reinterpret_cast<const half8*>(smem + ((fa + i * 1024 + so) & 0xffff));
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[0][j] = *reinterpret_cast<const half8*>(smem + 65536 + ((fa + j * 1024 + so) & 0x7fff));
        }
#pragma unroll
        for (int d = 0; d < NDMA; ++d) {
            const _Float16* s2 = src + ((it * NDMA + d) & 7) * 2048;
            asm volatile("" : "+v"(s2));
            __builtin_amdgcn_global_load_lds((gptr_t)s2, (lptr_t)(dst + ((it + d) & 3) * 8192), 16, 0, 0);
        }
        if constexpr (!PIPE) {
            wait_vm<NDMA>();
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (PIPE) {      // next step's fragments requested here, consumed next iteration: the reads overlap this step's MFMAs
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[nxt][i] = *reinterpret_cast<const half8*>(smem + ((fa + i * 1024 + so) & 0xffff));
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[nxt][j] = *reinterpret_cast<const half8*>(smem + 65536 + ((fa + j * 1024 + so) & 0x7fff));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cur][j], xf[cur][i], acc[i][j], 0, 0, 0);
        if constexpr (PIPE) {
            // interleave: one ds_read per (MT NT / (MT + NT)) MFMAs (mask 0x100 = DS read, 0x008 = MFMA)
#pragma unroll
            for (int k = 0; k < MT + NT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, (MT * NT) / (MT + NT), 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            wait_vm<NDMA>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    };
    for (int it = 0; it < iters; it += 2) {
        step(it, std::integral_constant<int, 0>{});
        step(it + 1, std::integral_constant<int, 1>{});
    }
    if (t == 0 && blockIdx.x < 1024) clk[blockIdx.x] = clock64() - c0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[0] = s;
}

// The same 128 x 64 per-wave tile with v_mfma_f32_32x32x16_f16 (VERDICT r4 #2b: "build the 32x32x16 twin"): 4 x 2 tiles of 32 x 32, a K-step of 32 =
// two K-slices of 16; per K-step 16 MFMAs of 32 cycles and (4 + 2) * 2 = 12 fragment reads -- the SAME LDS bytes as the 16 x 16 x 32 form (a wave
// reads (128 + 64) rows x 64 bytes of operands per K-step whatever the instruction shape), half the MFMA issue slots.
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NDMA>
__global__ __launch_bounds__(512) void step32_kernel(const _Float16* __restrict__ g, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8, MT = 4, NT = 2;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < 96 * 1024 / 16; i += NW * 64) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(g)[i & 4095];
    __syncthreads();
    floatx16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    half8 xf[2][2][MT], wf[2][2][NT];          // [set][K slice][tile]
    const int fa = (lane & 15) * 16 + (lane >> 4) * 4096 + wv * 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[0][s][i] = *reinterpret_cast<const half8*>(smem + ((fa + (2 * i + s) * 1024) & 0xffff));
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[0][s][j] = *reinterpret_cast<const half8*>(smem + 65536 + ((fa + (2 * j + s) * 1024) & 0x7fff));
    }
    const _Float16* src = g + (size_t)(blockIdx.x & 63) * 32768 + t * 8;
    char* const dst = smem + 98304 + wv * 1024;
    unsigned long long c0 = 0;
    if (t == 0) c0 = clock64();
    auto step = [&](int it, auto curc) {
        constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
        const int so = (it & 3) * 2048;
#pragma unroll
        for (int d = 0; d < NDMA; ++d) {
            const _Float16* s2 = src + ((it * NDMA + d) & 7) * 2048;
            asm volatile("" : "+v"(s2));
            __builtin_amdgcn_global_load_lds((gptr_t)s2, (lptr_t)(dst + ((it + d) & 3) * 8192), 16, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[nxt][s][i] = *reinterpret_cast<const half8*>(smem + ((fa + (2 * i + s) * 1024 + so) & 0xffff));
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[nxt][s][j] = *reinterpret_cast<const half8*>(smem + 65536 + ((fa + (2 * j + s) * 1024 + so) & 0x7fff));
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][s][j], xf[cur][s][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {              // 16 MFMAs, 12 reads: 1 + 1 + 2 MFMAs, a read behind each group
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        wait_vm<NDMA>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int it = 0; it < iters; it += 2) {
        step(it, std::integral_constant<int, 0>{});
        step(it + 1, std::integral_constant<int, 1>{});
    }
    if (t == 0 && blockIdx.x < 1024) clk[blockIdx.x] = clock64() - c0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[0] = s;
}

template <int NDMA>
static void run32(const char* name, const _Float16* g, float* out, unsigned long long* clk, int iters) {
    auto k = step32_kernel<NDMA>;
    const int lds = 98304 + 4 * 8192 + 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, g, out, clk, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (auto v : h) cyc += (double)v;
    cyc /= 256.0 * iters;
    const double flop = 256.0 * 8 * iters * 16 * 32768.0;
    printf("%-44s %7.0f shader clocks per K-step (matrix pipe needs 1024: %.2f busy), %6.0f TFLOP/s, %.3f ms\n", name, cyc, 1024.0 / cyc, flop / best * 1e-9, best);
}

template <int MT, int NT, int NW, int NDMA, bool PIPE>
static void run(const char* name, const _Float16* g, float* out, unsigned long long* clk, int iters) {
    auto k = step_kernel<MT, NT, NW, NDMA, PIPE>;
    const int lds = 98304 + 4 * 8192 + 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(NW * 64), lds, 0, g, out, clk, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (auto v : h) cyc += (double)v;
    cyc /= 256.0 * iters;
    const double flop = 256.0 * NW * iters * MT * NT * 16384.0;
    const double mfma_cyc = (double)MT * NT * 16 * (NW / 4);        // matrix-pipe cycles per K-step per SIMD
    printf("%-44s %7.0f shader clocks per K-step (matrix pipe needs %4.0f: %.2f busy), %6.0f TFLOP/s, %.3f ms\n", name, cyc, mfma_cyc, mfma_cyc / cyc,
           flop / best * 1e-9, best);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    _Float16* g; float* out; unsigned long long* clk;
    std::vector<_Float16> h(64 * 32768 + 65536);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) * 0.001f);
    CK(hipMalloc(&g, h.size() * 2)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&clk, 1024 * 8));
    CK(hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    printf("one wave per SIMD, 128 x 128 per-wave tile (MT 8 x NT 8), fragments of step k+1 read under the MFMAs of step k:\n");
    run<8, 8, 4, 0, true>("  16 ds_read_b128, 0 LDS-DMA", g, out, clk, iters);
    run<8, 8, 4, 2, true>("  16 ds_read_b128, 2 LDS-DMA", g, out, clk, iters);
    run<8, 8, 4, 4, true>("  16 ds_read_b128, 4 LDS-DMA", g, out, clk, iters);
    run<8, 8, 4, 5, true>("  16 ds_read_b128, 5 LDS-DMA", g, out, clk, iters);
    run<8, 8, 4, 6, true>("  16 ds_read_b128, 6 LDS-DMA", g, out, clk, iters);
    printf("two waves per SIMD, 128 x 64 tiles (MT 8 x NT 4), software-pipelined the same way (no register room for it in the real kernel):\n");
    run<8, 4, 8, 0, true>("  12 ds_read_b128, 0 LDS-DMA", g, out, clk, iters);
    run<8, 4, 8, 2, true>("  12 ds_read_b128, 2 LDS-DMA", g, out, clk, iters);
    run<8, 4, 8, 3, true>("  12 ds_read_b128, 3 LDS-DMA", g, out, clk, iters);
    printf("two waves per SIMD, 128 x 64 tiles as 4 x 2 tiles of v_mfma_f32_32x32x16_f16, software-pipelined the same way:\n");
    run32<0>("  12 ds_read_b128, 0 LDS-DMA", g, out, clk, iters);
    run32<2>("  12 ds_read_b128, 2 LDS-DMA", g, out, clk, iters);
    run32<3>("  12 ds_read_b128, 3 LDS-DMA", g, out, clk, iters);
    printf("two waves per SIMD, 128 x 64 tiles, serial LOAD then COMPUTE in every wave, all waves in step (no ping-pong offset):\n");
    run<8, 4, 8, 2, false>("  12 ds_read_b128, 2 LDS-DMA", g, out, clk, iters);
    run<8, 4, 8, 3, false>("  12 ds_read_b128, 3 LDS-DMA", g, out, clk, iters);
    return 0;
}
