// kernels_elt.hip -- the small NHWC graph ops (SPPF pooling, 2x upsample, max/avg pool, L2 norm) and input converters.
#include "conv_common.hpp"

namespace aic {

// ------------------------------------------------------------------------------------------------
// Small NHWC ops. One thread per 16-byte channel chunk (8 halves / 4 floats); HBM/L2-bound.
template <typename T> struct Vec;
template <> struct Vec<half_t> { typedef half8 type; static constexpr int N = 8; };
template <> struct Vec<float> { typedef floatx4 type; static constexpr int N = 4; };

template <typename T>
__device__ __forceinline__ typename Vec<T>::type vmax(typename Vec<T>::type a, typename Vec<T>::type b) {
    typename Vec<T>::type o;
#pragma unroll
    for (int e = 0; e < Vec<T>::N; ++e) o[e] = a[e] > b[e] ? a[e] : b[e];
    return o;
}

template <typename T>
__global__ void sppf_pool_kernel(const EltArgs a) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const int cv = a.c / VN;
    const long total = (long)a.n * a.h * a.w * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = (int)(idx % cv);
    long p = idx / cv;
    const int x = (int)(p % a.w); p /= a.w;
    const int y = (int)(p % a.h);
    const int img = (int)(p / a.h);
    const T* src = reinterpret_cast<const T*>(a.src);
    T* dst = reinterpret_cast<T*>(a.dst);
    V m5, m9, m13;
    const T lowest = (T)(-65504.0f);
#pragma unroll
    for (int e = 0; e < VN; ++e) m5[e] = m9[e] = m13[e] = lowest;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= a.h) continue;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= a.w) continue;
            const V v = *reinterpret_cast<const V*>(src + ((size_t)(img * a.h + yy) * a.w + xx) * a.s_cs + a.s_coff + cc * VN);
            m13 = vmax<T>(m13, v);
            const int ad = max(abs(dy), abs(dx));
            if (ad <= 4) m9 = vmax<T>(m9, v);
            if (ad <= 2) m5 = vmax<T>(m5, v);
        }
    }
    T* o = dst + ((size_t)(img * a.h + y) * a.w + x) * a.d_cs + a.d_coff + cc * VN;
    *reinterpret_cast<V*>(o) = m5;
    *reinterpret_cast<V*>(o + a.c) = m9;
    *reinterpret_cast<V*>(o + 2 * a.c) = m13;
}

// Separable form of the three cascaded 5x5 max-pools of SPPF (windows 5, 9, 13 with -inf padding): one block per
// (image, 16-byte channel chunk), the h x w map in LDS, a horizontal then a vertical pass of 13 taps each (26 LDS
// reads per pixel instead of 169 global ones).
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_sep_kernel(const EltArgs a) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    V* tile = reinterpret_cast<V*>(sm_raw);          // [4][h*w]: input, then the three horizontal maxima
    const int cv = a.c / VN;
    const int img = blockIdx.x / cv, cc = blockIdx.x - img * cv;
    const int hw = a.h * a.w;
    const T* src = reinterpret_cast<const T*>(a.src) + (size_t)img * hw * a.s_cs + a.s_coff + cc * VN;
    T* dst = reinterpret_cast<T*>(a.dst) + (size_t)img * hw * a.d_cs + a.d_coff + cc * VN;
    for (int p = threadIdx.x; p < hw; p += 256) tile[p] = *reinterpret_cast<const V*>(src + (size_t)p * a.s_cs);
    __syncthreads();
    for (int p = threadIdx.x; p < hw; p += 256) {
        const int y = p / a.w, x = p - y * a.w;
        V m5 = tile[p], m9, m13;
#pragma unroll
        for (int d = 1; d <= 2; ++d) {
            if (x - d >= 0) m5 = vmax<T>(m5, tile[p - d]);
            if (x + d < a.w) m5 = vmax<T>(m5, tile[p + d]);
        }
        m9 = m5;
#pragma unroll
        for (int d = 3; d <= 4; ++d) {
            if (x - d >= 0) m9 = vmax<T>(m9, tile[p - d]);
            if (x + d < a.w) m9 = vmax<T>(m9, tile[p + d]);
        }
        m13 = m9;
#pragma unroll
        for (int d = 5; d <= 6; ++d) {
            if (x - d >= 0) m13 = vmax<T>(m13, tile[p - d]);
            if (x + d < a.w) m13 = vmax<T>(m13, tile[p + d]);
        }
        tile[hw + p] = m5, tile[2 * hw + p] = m9, tile[3 * hw + p] = m13;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < hw; p += 256) {
        const int y = p / a.w;
        V m5 = tile[hw + p], m9 = tile[2 * hw + p], m13 = tile[3 * hw + p];
#pragma unroll
        for (int d = 1; d <= 6; ++d) {
            const bool up = y - d >= 0, dn = y + d < a.h;
            if (d <= 2) {
                if (up) m5 = vmax<T>(m5, tile[hw + p - d * a.w]);
                if (dn) m5 = vmax<T>(m5, tile[hw + p + d * a.w]);
            }
            if (d <= 4) {
                if (up) m9 = vmax<T>(m9, tile[2 * hw + p - d * a.w]);
                if (dn) m9 = vmax<T>(m9, tile[2 * hw + p + d * a.w]);
            }
            if (up) m13 = vmax<T>(m13, tile[3 * hw + p - d * a.w]);
            if (dn) m13 = vmax<T>(m13, tile[3 * hw + p + d * a.w]);
        }
        T* o = dst + (size_t)p * a.d_cs;
        *reinterpret_cast<V*>(o) = m5;
        *reinterpret_cast<V*>(o + a.c) = m9;
        *reinterpret_cast<V*>(o + 2 * a.c) = m13;
    }
}

template <typename T>
__global__ void upsample2x_kernel(const EltArgs a) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const int cv = a.c / VN;
    const int oh = 2 * a.h, ow = 2 * a.w;
    const long total = (long)a.n * oh * ow * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = (int)(idx % cv);
    long p = idx / cv;
    const int x = (int)(p % ow); p /= ow;
    const int y = (int)(p % oh);
    const int img = (int)(p / oh);
    const T* src = reinterpret_cast<const T*>(a.src);
    T* dst = reinterpret_cast<T*>(a.dst);
    const V v = *reinterpret_cast<const V*>(src + ((size_t)(img * a.h + (y >> 1)) * a.w + (x >> 1)) * a.s_cs + a.s_coff + cc * VN);
    *reinterpret_cast<V*>(dst + ((size_t)(img * oh + y) * ow + x) * a.d_cs + a.d_coff + cc * VN) = v;
}

template <typename T>
__global__ void maxpool3s2_kernel(const EltArgs a) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const int cv = a.c / VN;
    const int oh = (a.h + 2 - 3) / 2 + 1, ow = (a.w + 2 - 3) / 2 + 1;
    const long total = (long)(a.n_dev ? min(a.n, a.n_dev[0]) : a.n) * oh * ow * cv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = (int)(idx % cv);
    long p = idx / cv;
    const int x = (int)(p % ow); p /= ow;
    const int y = (int)(p % oh);
    const int img = (int)(p / oh);
    const T* src = reinterpret_cast<const T*>(a.src);
    T* dst = reinterpret_cast<T*>(a.dst);
    V m;
#pragma unroll
    for (int e = 0; e < VN; ++e) m[e] = (T)(-65504.0f);
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = 2 * y + dy;
        if (yy < 0 || yy >= a.h) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = 2 * x + dx;
            if (xx < 0 || xx >= a.w) continue;
            m = vmax<T>(m, *reinterpret_cast<const V*>(src + ((size_t)(img * a.h + yy) * a.w + xx) * a.s_cs + a.s_coff + cc * VN));
        }
    }
    *reinterpret_cast<V*>(dst + ((size_t)(img * oh + y) * ow + x) * a.d_cs + a.d_coff + cc * VN) = m;
}

// global average pool: one thread per (item, channel); h*w is 32 for the ReID trunk.
template <typename T>
__global__ void avgpool_kernel(const EltArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)(a.n_dev ? min(a.n, a.n_dev[0]) : a.n) * a.c) return;
    const int ch = (int)(idx % a.c);
    const int img = (int)(idx / a.c);
    const T* src = reinterpret_cast<const T*>(a.src) + (size_t)img * a.h * a.w * a.s_cs + a.s_coff + ch;
    float sum = 0.f;
    const int hw = a.h * a.w;
    for (int p = 0; p < hw; ++p) sum += (float)src[(size_t)p * a.s_cs];
    reinterpret_cast<T*>(a.dst)[(size_t)img * a.d_cs + a.d_coff + ch] = (T)(sum / (float)hw);
}

// L2 normalise: one wavefront per item, fp32 output.
template <typename T>
__global__ void l2norm_kernel(const EltArgs a) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (item >= (a.n_dev ? min(a.n, a.n_dev[0]) : a.n)) return;
    const T* src = reinterpret_cast<const T*>(a.src) + (size_t)item * a.s_cs + a.s_coff;
    float ss = 0.f;
    for (int c = lane; c < a.c; c += 64) { const float v = (float)src[c]; ss += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float* dst = reinterpret_cast<float*>(a.dst) + (size_t)item * a.d_cs + a.d_coff;
    for (int c = lane; c < a.c; c += 64) dst[c] = (float)src[c] / nrm;
}

template <typename T>
__global__ void nchw_to_nhwc8_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int h, int w) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)h * w;
    if (idx >= (long)n * hw) return;
    const long img = idx / hw, p = idx - img * hw;
    const float* s = src + img * 3 * hw + p;
    T o[8];
    o[0] = (T)s[0]; o[1] = (T)s[hw]; o[2] = (T)s[2 * hw];
#pragma unroll
    for (int e = 3; e < 8; ++e) o[e] = (T)0.f;
    T* d = dst + idx * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = o[e];
}

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

#define ELT_LAUNCH(kernel, total)                                                                  \
    do {                                                                                           \
        const long _tot = (total);                                                                 \
        if (_tot <= 0) return;                                                                     \
        if (dtype == AIC_F16) hipLaunchKernelGGL(kernel<half_t>, dim3(ceil_div(_tot, 256)), dim3(256), 0, s, a); \
        else hipLaunchKernelGGL(kernel<float>, dim3(ceil_div(_tot, 256)), dim3(256), 0, s, a);     \
        KCHECK();                                                                                  \
    } while (0)

static inline int vecn(int dtype) { return dtype == AIC_F16 ? 8 : 4; }

void launch_sppf_pool(int dtype, const EltArgs& a, hipStream_t s) {
    const size_t lds = (size_t)4 * a.h * a.w * 16;
    static const bool direct = getenv("AICAM_SPPF_DIRECT") != nullptr;
    if (!direct && lds <= 64 * 1024 && a.n > 0) {      // in-place safe: a block reads its whole (image, chunk) map before it writes
        const int blocks = a.n * (a.c / vecn(dtype));
        if (dtype == AIC_F16) hipLaunchKernelGGL(sppf_pool_sep_kernel<half_t>, dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL(sppf_pool_sep_kernel<float>, dim3(blocks), dim3(256), lds, s, a);
        KCHECK();
        return;
    }
    ELT_LAUNCH(sppf_pool_kernel, (long)a.n * a.h * a.w * (a.c / vecn(dtype)));
}
void launch_upsample2x(int dtype, const EltArgs& a, hipStream_t s) { ELT_LAUNCH(upsample2x_kernel, (long)a.n * 4 * a.h * a.w * (a.c / vecn(dtype))); }
void launch_maxpool3s2(int dtype, const EltArgs& a, hipStream_t s) {
    const int oh = (a.h - 1) / 2 + 1, ow = (a.w - 1) / 2 + 1;
    ELT_LAUNCH(maxpool3s2_kernel, (long)a.n * oh * ow * (a.c / vecn(dtype)));
}
// fp16, channels in groups of eight: one 16-byte load per pixel instead of eight 2-byte ones (the scalar form moved the ReID trunk's
// 252 MB at 1.7 TB/s: 150 us per 15 360 crops); every channel is still summed over p = 0 .. hw-1 in that order: same bits
__global__ void avgpool8_kernel(const EltArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = a.c / 8;
    if (idx >= (long)(a.n_dev ? min(a.n, a.n_dev[0]) : a.n) * c8) return;
    const int cg = (int)(idx % c8);
    const int img = (int)(idx / c8);
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const half_t* src = reinterpret_cast<const half_t*>(a.src) + (size_t)img * a.h * a.w * a.s_cs + a.s_coff + 8 * cg;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int hw = a.h * a.w;
#pragma unroll 4
    for (int p = 0; p < hw; ++p) {
        const h8 v = *reinterpret_cast<const h8*>(src + (size_t)p * a.s_cs);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] += (float)v[e];
    }
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)(sum[e] / (float)hw);
    *reinterpret_cast<h8*>(reinterpret_cast<half_t*>(a.dst) + (size_t)img * a.d_cs + a.d_coff + 8 * cg) = o;
}

void launch_avgpool(int dtype, const EltArgs& a, hipStream_t s) {
    if (dtype == AIC_F16 && a.n > 0 && a.c % 8 == 0 && ((a.s_cs | a.s_coff | a.d_cs | a.d_coff) % 8) == 0) {
        const long tot = (long)a.n * (a.c / 8);
        hipLaunchKernelGGL(avgpool8_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, s, a);
        KCHECK();
        return;
    }
    ELT_LAUNCH(avgpool_kernel, (long)a.n * a.c);
}
void launch_l2norm(int dtype, const EltArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    if (dtype == AIC_F16) hipLaunchKernelGGL(l2norm_kernel<half_t>, dim3(ceil_div(a.n, 4)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(l2norm_kernel<float>, dim3(ceil_div(a.n, 4)), dim3(256), 0, s, a);
    KCHECK();
}
void launch_nchw_to_nhwc8(int dtype, const float* src, void* dst, int n, int h, int w, hipStream_t s) {
    const long tot = (long)n * h * w;
    if (tot <= 0) return;
    if (dtype == AIC_F16) hipLaunchKernelGGL(nchw_to_nhwc8_kernel<half_t>, dim3(ceil_div(tot, 256)), dim3(256), 0, s, src, (half_t*)dst, n, h, w);
    else hipLaunchKernelGGL(nchw_to_nhwc8_kernel<float>, dim3(ceil_div(tot, 256)), dim3(256), 0, s, src, (float*)dst, n, h, w);
    KCHECK();
}
void launch_copy_f32(const float* src, float* dst, size_t count, hipStream_t s) {
    if (!count) return;
    hipLaunchKernelGGL(copy_f32_kernel, dim3(ceil_div((long)count, 256)), dim3(256), 0, s, src, dst, count);
    KCHECK();
}

}  // namespace aic
