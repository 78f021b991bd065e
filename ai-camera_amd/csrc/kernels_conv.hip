// kernels_conv.hip -- implicit-GEMM convolution on the gfx950 matrix cores + the small graph ops.
//
// conv_igemm: NHWC activations, weights packed [Cout][kh][kw][cin] so both GEMM operands are
// contiguous along K.  GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k] with
// K = kh*kw*cin walked tap-major; the MFMA "A" operand is the weight tile and the "B" operand
// the gathered pixel tile, so each lane ends up with 4 CONSECUTIVE output channels of one
// pixel (row = 4*(lane>>4)+i, col = lane&15 of v_mfma_f32_16x16x*) and the epilogue stores
// them as one 8-byte (fp16) / 16-byte (fp32) NHWC vector: no transpose through LDS.
//
//   fp16 mode: v_mfma_f32_16x16x32_f16, fp32 accumulate  (throughput mode)
//   fp32 mode: v_mfma_f32_16x16x4_f32, exact fp32 fmaf chain (parity mode)
//
// Per K-step each tile row holds 64 B (32 halves / 16 floats).  LDS image: row*64 +
// 16*(chunk ^ ((row>>1)&3)) -- an XOR swizzle that is conflict-free for the ds_read_b128
// lane groups of gfx950 (checked exhaustively against the group table of
// MI355X_MICROARCH.md §LDS) and for the staging ds_write_b128.  Global->register->LDS staging
// with the next step's loads issued before the current step's MFMAs (register double buffer):
// zero-padding, image borders and the K tail are resolved per 16-byte chunk at load time.
#include "kernels.hpp"

#include <cstdlib>
#include <type_traits>

namespace aic {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return v / (1.0f + __expf(-v));   // SiLU
    if (act == 2) return fmaxf(v, 0.0f);             // ReLU
    return v;
}

template <typename T> struct Frag;
template <> struct Frag<half_t> {
    typedef half8 type;
    static __device__ __forceinline__ floatx4 mma(const half8& a, const half8& b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Frag<float> {
    typedef floatx4 type;
    static __device__ __forceinline__ floatx4 mma(const floatx4& a, const floatx4& b, floatx4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

// swz(row) = ((row>>1)&3) ^ ((row>>3)&2): conflict-free for 16 consecutive rows (pixel tiles, identity weight tiles) AND
// for the permuted weight rows {c + 8k + s} of perm_row() (brute-forced over the ds_read_b128 lane groups).
__device__ __forceinline__ int lds_swz(int row) { return ((row >> 1) & 3) ^ ((row >> 3) & 2); }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + 16 * (chunk ^ lds_swz(row)); }

// 4 waves per block arranged WM x WN; each wave owns MT x NT tiles of 16 pixels x 16 channels.
template <typename T, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int CH = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int BKE = 4 * CH;                // K elements per step
    constexpr int BM = WM * MT * 16;           // pixels per block
    constexpr int BN = WN * NT * 16;           // output channels per block
    constexpr int A_PER = BM / 64;             // 16-byte chunks of the pixel tile per thread
    constexpr int B_PER = (BN + 63) / 64;      // ... of the weight tile
    constexpr int TILE = (BM + BN) * 64;       // bytes per stage
    static_assert(WM * WN == 4 && BM % 64 == 0, "block is 4 waves; pixel tile a multiple of 64");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int kc = t & 3;        // which 16-byte chunk of the 64-byte K-step this thread stages
    const int r0 = t >> 2;       // first tile row this thread stages
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);

    // ---- per-thread description of the pixel rows it gathers
    size_t pix_base[A_PER];
    int ih0[A_PER], iw0[A_PER];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + r0 + 64 * i;
        if (m < a.M) {
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int oh = rem / a.Wo;
            const int ow = rem - oh * a.Wo;
            pix_base[i] = (size_t)img * a.H * a.W;
            ih0[i] = oh * a.stride - a.pad;
            iw0[i] = ow * a.stride - a.pad;
        } else {
            pix_base[i] = 0;
            ih0[i] = -(1 << 28);   // never in range -> zero rows
            iw0[i] = 0;
        }
    }
    // ---- position of this thread's chunk inside K: (kh, kw, c)
    int c_in = kc * CH, kw = 0, kh = 0;
    while (c_in >= a.Cin) {
        c_in -= a.Cin;
        if (++kw == a.KW) { kw = 0; ++kh; }
    }

    uint4 a_reg[A_PER], b_reg[B_PER];
    const int nsteps = a.Kp / BKE;

    auto load_step = [&](int step) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int ih = ih0[i] + kh, iw = iw0[i] + kw;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kh < a.KH && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
                const size_t off = (pix_base[i] + (size_t)ih * a.W + iw) * a.x_cs + a.x_coff + c_in;
                v = *reinterpret_cast<const uint4*>(xg + off);
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int row = r0 + 64 * j;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < BN) v = *reinterpret_cast<const uint4*>(wg + (size_t)(n0 + row) * a.Kp + step * BKE + kc * CH);
            b_reg[j] = v;
        }
        // advance (kh, kw, c) by one K-step
        c_in += BKE;
        while (c_in >= a.Cin) {
            c_in -= a.Cin;
            if (++kw == a.KW) { kw = 0; ++kh; }
        }
    };
    auto store_step = [&](int stage) {
        char* base = smem + stage * TILE;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int row = r0 + 64 * i;
            *reinterpret_cast<uint4*>(base + lds_off(row, kc)) = a_reg[i];
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int row = r0 + 64 * j;
            if (row < BN) *reinterpret_cast<uint4*>(base + BM * 64 + lds_off(row, kc)) = b_reg[j];
        }
    };

    const int lane = t & 63, wv = t >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    load_step(0);
    store_step(0);
    __syncthreads();

    typedef typename Frag<T>::type frag_t;
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const bool more = step + 1 < nsteps;
        if (more) load_step(step + 1);   // global loads in flight under the MFMAs below
        const char* base = smem + cur * TILE;
        frag_t xf[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = (wm * MT + i) * 16 + r;
            xf[i] = *reinterpret_cast<const frag_t*>(base + lds_off(row, q));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int row = (wn * NT + j) * 16 + r;
            wf[j] = *reinterpret_cast<const frag_t*>(base + BM * 64 + lds_off(row, q));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = Frag<T>::mma(wf[j], xf[i], acc[i][j]);
        if (more) store_step(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, residual, activation; 4 consecutive channels per lane
    const float* __restrict__ bias = a.bias;
    const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wm * MT + i) * 16 + r;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + (wn * NT + j) * 16 + 4 * q;
            if (n >= a.Cout) continue;
            const floatx4 b4 = *reinterpret_cast<const floatx4*>(bias + n);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[e];
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.res_mode) {
                const T* rp = rg + (size_t)m * a.r_cs + a.r_coff + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[e] = (n + e < a.Cout) ? (float)rp[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e];
                if (a.res_mode == 1) x += rv[e];
                x = act_apply(x, a.act);
                if (a.res_mode == 2) x += rv[e];
                v[e] = x;
            }
            const size_t yoff = (size_t)m * a.y_cs + a.y_coff + n;
            if (n + 4 <= a.Cout) {
                if (a.out_f32) {
                    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(a.y) + yoff) = floatx4{v[0], v[1], v[2], v[3]};
                } else if constexpr (sizeof(T) == 2) {
                    half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(a.y) + yoff) = h;
                } else {
                    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(a.y) + yoff) = floatx4{v[0], v[1], v[2], v[3]};
                }
            } else {
                for (int e = 0; e < 4 && n + e < a.Cout; ++e) {
                    if (a.out_f32 || sizeof(T) == 4) reinterpret_cast<float*>(a.y)[yoff + e] = v[e];
                    else reinterpret_cast<half_t*>(a.y)[yoff + e] = (half_t)v[e];
                }
            }
        }
    }
}


// ---- shared by the v2 kernel: cheap index math and a specialised epilogue ------------------------
// m -> (m / d, m % d) with one reciprocal multiply and a +-1 fix-up (m < 2^26 here).
__device__ __forceinline__ void fast_divmod(int m, int d, float inv, int& q, int& r) {
    q = (int)(__int2float_rz(m) * inv);
    r = m - q * d;
    if (r >= d) { r -= d; ++q; }
    if (r < 0) { r += d; --q; }
}

template <int ACT> __device__ __forceinline__ float act_fast(float v) {
    if constexpr (ACT == 1) {   // SiLU = v * sigmoid(v); v_exp_f32 + v_rcp_f32 (<= 1 ulp each)
        return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
    } else if constexpr (ACT == 2) {
        return fmaxf(v, 0.0f);
    } else {
        return v;
    }
}

// One lane owns, per (i, j) tile, 4 consecutive output channels of one pixel.
// ACT / RES / F32OUT are compile-time so the unrolled body carries no branches.
template <typename T, int MT, int NT, int ACT, int RES, bool F32OUT>
__device__ __forceinline__ void epilogue_fast(const ConvArgs& a, floatx4 (&acc)[MT][NT], const int (&mrow)[MT], int n_base, int q) {
    const float* __restrict__ bias = a.bias;
    const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
    floatx4 b4[NT];
    bool ncol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n_base + j * 16 + 4 * q;
        ncol[j] = n < a.Cout;                       // Cout % 4 == 0 on this path: all four or none
        b4[j] = *reinterpret_cast<const floatx4*>(bias + n);   // bias is padded to cout_pad
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mrow[i];
        if (m < 0) continue;
        const size_t ybase = (size_t)m * a.y_cs + a.y_coff;
        const size_t rbase = RES ? (size_t)m * a.r_cs + a.r_coff : 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (!ncol[j]) continue;
            const int n = n_base + j * 16 + 4 * q;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[j][e];
            if constexpr (RES != 0) {
                float rv[4];
                if constexpr (sizeof(T) == 2) {
                    const half4 h = *reinterpret_cast<const half4*>(rg + rbase + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[e] = (float)h[e];
                } else {
                    const floatx4 h = *reinterpret_cast<const floatx4*>(rg + rbase + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[e] = h[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = RES == 1 ? act_fast<ACT>(v[e] + rv[e]) : act_fast<ACT>(v[e]) + rv[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fast<ACT>(v[e]);
            }
            if constexpr (F32OUT || sizeof(T) == 4) {
                *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(a.y) + ybase + n) = floatx4{v[0], v[1], v[2], v[3]};
            } else {
                const half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(a.y) + ybase + n) = h;
            }
        }
    }
}

// Channel permutation of the v2+ kernels: MFMA tile j, row rho (= 4q + e on the output side) of a wave's
// NT-tile channel block carries output channel  32*(j>>1) + 8*(rho>>2) + 4*(j&1) + (rho&3)  (tiles taken in
// pairs; an odd last tile keeps the identity 16j + rho).  A lane (r, q) then owns, per tile pair, EIGHT
// consecutive channels of its pixel: one 16-byte fp16 store (two for fp32) instead of two 8-byte ones, and
// the four q-lanes of a pixel write 64 contiguous bytes per instruction.  The A-operand (weight) rows are
// fetched from LDS through the same map (perm_row), so the arithmetic per output is unchanged.
template <int NT> __device__ __forceinline__ int perm_ch(int j, int q, int e) {
    return j < (NT & ~1) ? 32 * (j >> 1) + 8 * q + 4 * (j & 1) + e : 16 * j + 4 * q + e;
}
template <int NT> __device__ __forceinline__ int perm_row(int j, int r) { return perm_ch<NT>(j, r >> 2, r & 3); }

template <typename T, int MT, int NT, int ACT, int RES, bool F32OUT>
__device__ __forceinline__ void epilogue_wide(const ConvArgs& a, floatx4 (&acc)[MT][NT], const int (&mrow)[MT], int n_base, int q) {
    constexpr int NP = NT / 2;
    const float* __restrict__ bias = a.bias;
    const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
    floatx4 b4[NT];
    bool pcol[NP + 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int n = n_base + 32 * p + 8 * q;
        pcol[p] = n < a.Cout;                       // Cout % 8 == 0 on this path: all eight or none
        b4[2 * p] = *reinterpret_cast<const floatx4*>(bias + n);          // bias is padded to cout_pad
        b4[2 * p + 1] = *reinterpret_cast<const floatx4*>(bias + n + 4);
    }
    if constexpr (NT & 1) {
        const int n = n_base + 16 * (NT - 1) + 4 * q;
        pcol[NP] = n < a.Cout;
        b4[NT - 1] = *reinterpret_cast<const floatx4*>(bias + n);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mrow[i];
        if (m < 0) continue;
        const size_t ybase = (size_t)m * a.y_cs + a.y_coff;
        const size_t rbase = RES ? (size_t)m * a.r_cs + a.r_coff : 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (!pcol[p]) continue;
            const int n = n_base + 32 * p + 8 * q;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * p + (e >> 2)][e & 3] + b4[2 * p + (e >> 2)][e & 3];
            if constexpr (RES != 0) {
                float rv[8];
                if constexpr (sizeof(T) == 2) {
                    const half8 h = *reinterpret_cast<const half8*>(rg + rbase + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] = (float)h[e];
                } else {
                    const floatx4 h0 = *reinterpret_cast<const floatx4*>(rg + rbase + n);
                    const floatx4 h1 = *reinterpret_cast<const floatx4*>(rg + rbase + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rv[e] = h0[e]; rv[4 + e] = h1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = RES == 1 ? act_fast<ACT>(v[e] + rv[e]) : act_fast<ACT>(v[e]) + rv[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_fast<ACT>(v[e]);
            }
            if constexpr (F32OUT || sizeof(T) == 4) {
                float* yp = reinterpret_cast<float*>(a.y) + ybase + n;
                *reinterpret_cast<floatx4*>(yp) = floatx4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<floatx4*>(yp + 4) = floatx4{v[4], v[5], v[6], v[7]};
            } else {
                const half8 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
                *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(a.y) + ybase + n) = h;
            }
        }
        if constexpr (NT & 1) {
            if (pcol[NP]) {
                constexpr int j = NT - 1;
                const int n = n_base + 16 * j + 4 * q;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[j][e];
                if constexpr (RES != 0) {
                    float rv[4];
                    if constexpr (sizeof(T) == 2) {
                        const half4 h = *reinterpret_cast<const half4*>(rg + rbase + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) rv[e] = (float)h[e];
                    } else {
                        const floatx4 h = *reinterpret_cast<const floatx4*>(rg + rbase + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) rv[e] = h[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = RES == 1 ? act_fast<ACT>(v[e] + rv[e]) : act_fast<ACT>(v[e]) + rv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fast<ACT>(v[e]);
                }
                if constexpr (F32OUT || sizeof(T) == 4) {
                    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(a.y) + ybase + n) = floatx4{v[0], v[1], v[2], v[3]};
                } else {
                    const half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(a.y) + ybase + n) = h;
                }
            }
        }
    }
}

// Generic (any Cout, any mode) fallback: runtime branches, scalar tail.
template <typename T, int MT, int NT, bool PERM>
__device__ __forceinline__ void epilogue_generic(const ConvArgs& a, floatx4 (&acc)[MT][NT], const int (&mrow)[MT], int n_base, int q) {
    // (inlined and fully unrolled on purpose: a call would force `a` and `acc` into scratch memory)
    const float* __restrict__ bias = a.bias;
    const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mrow[i];
        if (m < 0) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n_base + (PERM ? perm_ch<NT>(j, q, 0) : j * 16 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= a.Cout) continue;
                float x = acc[i][j][e] + bias[n + e];
                const float rv = a.res_mode ? (float)rg[(size_t)m * a.r_cs + a.r_coff + n + e] : 0.f;
                if (a.res_mode == 1) x += rv;
                x = act_apply(x, a.act);
                if (a.res_mode == 2) x += rv;
                const size_t yo = (size_t)m * a.y_cs + a.y_coff + n + e;
                if (a.out_f32 || sizeof(T) == 4) reinterpret_cast<float*>(a.y)[yo] = x;
                else reinterpret_cast<half_t*>(a.y)[yo] = (half_t)x;
            }
        }
    }
}

template <typename T, int MT, int NT, bool PERM = false>
__device__ __forceinline__ void epilogue_dispatch(const ConvArgs& a, floatx4 (&acc)[MT][NT], const int (&mrow)[MT], int n_base, int q) {
    const int key = (a.Cout & (PERM ? 7 : 3)) ? -1 : (a.act | (a.res_mode << 2) | (a.out_f32 << 4));
#define AIC_EPI(ACT, RES, F32) do { if constexpr (PERM) epilogue_wide<T, MT, NT, ACT, RES, F32>(a, acc, mrow, n_base, q); \
                                    else epilogue_fast<T, MT, NT, ACT, RES, F32>(a, acc, mrow, n_base, q); } while (0)
    switch (key) {
        case 1: AIC_EPI(1, 0, false); break;             // SiLU
        case 1 | (2 << 2): AIC_EPI(1, 2, false); break;  // SiLU then +res (C2f bottleneck)
        case 2: AIC_EPI(2, 0, false); break;             // ReLU
        case 2 | (1 << 2): AIC_EPI(2, 1, false); break;  // relu(x + res) (BasicBlock)
        case 0: AIC_EPI(0, 0, false); break;             // linear (downsample, FC)
        case 0 | (1 << 4): AIC_EPI(0, 0, true); break;   // linear fp32 (detect head)
        default: epilogue_generic<T, MT, NT, PERM>(a, acc, mrow, n_base, q); break;
    }
#undef AIC_EPI
}

// ------------------------------------------------------------------------------------------------
// v2: same tiling and MFMA mapping, but the operand tiles travel HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR staging and no ds_write pass) into an
// NSTAGE-deep ring, NSTAGE-1 K-steps in flight.  Per K-step: counted s_waitcnt vmcnt (never 0 in
// the loop) -> one raw s_barrier -> issue the loads of step+NSTAGE-1 -> MFMAs of the current step.
//  * the LDS image of a wave-instruction must be lane-linear, so the XOR swizzle is applied to the
//    SOURCE: the thread that fills LDS slot s of row r fetches K-chunk s ^ ((r>>1)&3)
//    (cdna_hip_programming.md §5.4 rule 21); the ds_read side uses the same involution;
//  * zero padding / image borders / K tail / rows past M or Cout: the lane's source address is a
//    64-byte page of zeros in HBM, so every wave issues exactly LPS loads per stage and the vmcnt
//    arithmetic is uniform (also in the drain iterations, which load zeros nobody reads);
//  * im2col address = per-row base pointer + one per-thread tap offset; in-bounds is a precomputed
//    bit per (row, tap).
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else static_assert(N < 0, "add this vmcnt literal");
}

template <typename T, int MT, int NT, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_dma_kernel(const ConvArgs a) {
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int BKE = 4 * CH;
    constexpr int NTHR = 64 * WM * WN;         // 4 or 8 waves
    constexpr int RP = NTHR / 4;               // tile rows staged per pass (one 16-byte chunk per thread)
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int BNP = (BN + RP - 1) / RP * RP;   // weight rows padded so every wave issues the same loads
    constexpr int A_PER = BM / RP;
    constexpr int B_PER = BNP / RP;
    constexpr int LPS = A_PER + B_PER;         // LDS-DMA instructions per stage per wave
    constexpr int STAGE = (BM + BNP) * 64;
    static_assert((WM * WN == 4 || WM * WN == 8) && BM % RP == 0 && NSTAGE >= 2, "geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);         // K-chunk this thread fetches (source-side swizzle)
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    const T* rowp[A_PER];
    unsigned vmask[A_PER];
    const int HoWo = a.Ho * a.Wo;
    const int ntap = a.KH * a.KW;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + r0 + RP * i;
        unsigned mk = 0;
        const T* rp = zero;
        if (m < a.M) {
            int img, rem, oh, ow;
            fast_divmod(m, HoWo, inv_howo, img, rem);
            fast_divmod(rem, a.Wo, inv_wo, oh, ow);
            const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
            rp = xg + (((long)img * a.H + ih0) * a.W + iw0) * a.x_cs + a.x_coff;
            // in-bounds taps, loop-free: columns [lo_w, hi_w) x rows [lo_h, hi_h) of the KH x KW window
            const int lo_w = max(0, -iw0), hi_w = min(a.KW, a.W - iw0);
            const int lo_h = max(0, -ih0), hi_h = min(a.KH, a.H - ih0);
            if (hi_w > lo_w && hi_h > lo_h) {
                const unsigned vw = ((1u << hi_w) - 1u) & ~((1u << lo_w) - 1u);
                const unsigned rows = (((1u << (hi_h * a.KW)) - 1u) & ~((1u << (lo_h * a.KW)) - 1u)) & a.tap_rows;
                mk = vw * rows;                     // replicate the column bits at every valid row
            }
        }
        rowp[i] = rp;
        vmask[i] = mk;
    }
    const int nsteps = a.Kp / BKE;
    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // LDS offsets of this lane's operand chunks inside a stage (stage base added as an immediate below)
    int xoff[MT], woff[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xoff[i] = lds_off((wm * MT + i) * 16 + r, q);
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = BM * 64 + lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);
    typedef typename Frag<T>::type frag_t;
    auto compute = [&](int stage) {
        const char* base = smem + stage * STAGE;
        frag_t xf[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(base + xoff[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(base + woff[j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = Frag<T>::mma(wf[j], xf[i], acc[i][j]);
    };
    char* const sdst = smem + (16 * wv) * 64;   // this wave's 16 rows inside a staging pass

    if (a.Cin % BKE == 0) {
        // ---- fast path: a K-step never straddles a tap, so the tap is uniform. Source pointers are set up
        // once per tap (validity bit, tap offset) and then only incremented: ~2 VALU per LDS-DMA instead of ~12.
        const int csteps = a.Cin / BKE;
        const T* aptr[A_PER];
        int ainc[A_PER];
        const T* wptr[B_PER];
        int winc[B_PER];
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const bool okr = r0 + RP * j < BN;
            wptr[j] = okr ? wg + (size_t)(n0 + r0 + RP * j) * a.Kp + kc * CH : zero;   // weights carry NSTAGE K-steps of slack
            winc[j] = okr ? BKE : 0;
        }
        int tap = 0, kh = 0, kw = 0, cc = 0;
        auto set_tap = [&] {
            const long toff = ((long)kh * a.W + kw) * a.x_cs + kc * CH;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const bool ok = tap < ntap && ((vmask[i] >> (tap & 31)) & 1u);
                aptr[i] = ok ? rowp[i] + toff : zero;
                ainc[i] = ok ? BKE : 0;
            }
        };
        set_tap();
        auto issue_fast = [&](int stage) {
            char* sbase = sdst + stage * STAGE;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                __builtin_amdgcn_global_load_lds((gptr_t)aptr[i], (lptr_t)(sbase + i * (RP * 64)), 16, 0, 0);
                aptr[i] += ainc[i];
            }
#pragma unroll
            for (int j = 0; j < B_PER; ++j) {
                __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(sbase + BM * 64 + j * (RP * 64)), 16, 0, 0);
                wptr[j] += winc[j];
            }
            if (++cc == csteps) {                      // uniform: next tap
                cc = 0;
                ++tap;
                if (++kw == a.KW) { kw = 0; ++kh; }
                set_tap();
            }
        };
#pragma unroll
        for (int st = 0; st < NSTAGE - 1; ++st) issue_fast(st);
        for (int step0 = 0; step0 < nsteps; step0 += NSTAGE) {
#pragma unroll
            for (int u = 0; u < NSTAGE; ++u) {         // stage index is a compile-time constant inside the body
                if (step0 + u < nsteps) {
                    wait_vmcnt<(NSTAGE - 2) * LPS>();
                    __builtin_amdgcn_s_barrier();
                    issue_fast((u + NSTAGE - 1) % NSTAGE);
                    compute(u);
                }
            }
        }
    } else {
        // ---- generic path (Cin < K-step or not a multiple of it: 3-channel stems, 16/48/80-channel layers)
        const T* wp[B_PER];
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int row = r0 + RP * j;
            wp[j] = row < BN ? wg + (size_t)(n0 + row) * a.Kp + kc * CH : nullptr;
        }
        int c_in = kc * CH, kw_ = 0, kh_ = 0;
        while (c_in >= a.Cin) {
            c_in -= a.Cin;
            if (++kw_ == a.KW) { kw_ = 0; ++kh_; }
        }
        int issued = 0;
        auto issue = [&](int stage) {
            const int tap = kh_ * a.KW + kw_;
            const bool in_k = tap < ntap;
            const long toff = ((long)kh_ * a.W + kw_) * a.x_cs + c_in;
            char* sbase = sdst + stage * STAGE;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const bool ok = in_k && ((vmask[i] >> (tap & 31)) & 1u);
                const T* src = ok ? rowp[i] + toff : zero;
                asm volatile("" : "+v"(src));   // one select, ONE LDS-DMA instruction per wave: keeps vmcnt counting uniform
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + i * (RP * 64)), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < B_PER; ++j) {
                const T* src = (wp[j] != nullptr && issued < nsteps) ? wp[j] + (size_t)issued * BKE : zero;
                asm volatile("" : "+v"(src));
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + BM * 64 + j * (RP * 64)), 16, 0, 0);
            }
            ++issued;
            c_in += BKE;
            while (c_in >= a.Cin) {
                c_in -= a.Cin;
                if (++kw_ == a.KW) { kw_ = 0; ++kh_; }
            }
        };
#pragma unroll
        for (int st = 0; st < NSTAGE - 1; ++st) issue(st);
        int cur = 0;
        for (int step = 0; step < nsteps; ++step) {
            wait_vmcnt<(NSTAGE - 2) * LPS>();      // this wave's loads of `step` have landed
            __builtin_amdgcn_s_barrier();          // ... everyone's have, and everyone finished step-1
            int nxt = cur + NSTAGE - 1;
            if (nxt >= NSTAGE) nxt -= NSTAGE;
            issue(nxt);                            // refills the buffer that step-1 just released
            compute(cur);
            if (++cur == NSTAGE) cur = 0;
        }
    }
    wait_vmcnt<0>();   // drain the zero-page loads of the tail before the LDS goes away

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wm * MT + i) * 16 + r;
        mrow[i] = m < a.M ? m : -1;
    }
    epilogue_dispatch<T, MT, NT, true>(a, acc, mrow, n0 + wn * NT * 16, q);
}

// ------------------------------------------------------------------------------------------------
// v4 "ping-pong": the v2 tile, operand ring and source-side swizzle, but every K-step is split into a LOAD
// segment (issue the LDS-DMA of step k+NSTAGE-2, ds_read the fragments of step k) and a COMPUTE segment (the
// MFMAs of step k), each closed by a raw s_barrier, and waves 4..7 run ONE barrier behind waves 0..3.  Each SIMD
// hosts one wave of either half, so while one half's MFMAs own the matrix pipe the other half is reading LDS
// and issuing DMA (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md T3/T5).  In v2 all eight
// waves leave the barrier together, read together and then fight for the pipe together.
// Hazards, in program segments (L_k = 2k, C_k = 2k+1; a wave of the late half executes segment s one global
// barrier after the early half):
//   RAW  step j is waited for (counted vmcnt) in L_{j-1} and read in L_j: two barriers later, so the late half's
//        waits have also passed a barrier every reader has passed;
//   WAR  the ring slot read in L_k (data in registers by C_k) is re-filled by the DMA issued in L_{k+2}: three
//        segments after the read was issued, hence after the late half's C_k.
// Needs Cin % K-step == 0 (uniform tap per K-step) and 8 waves; one block per CU (LDS: NSTAGE stages).
// Optional per-block phase timestamps (100 MHz wall clock) for tools/conv_bench.py: AICAM_PP_TIMES=1.
__device__ unsigned long long g_pp_times[4 * 4096];
__device__ int g_pp_times_on;
#define PP_STAMP(k) do { if (g_pp_times_on && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) g_pp_times[4 * blockIdx.x + (k)] = wall_clock64(); } while (0)

template <typename T, int MT, int NT, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(512) void conv_igemm_pp_kernel(const ConvArgs a) {
    PP_STAMP(0);
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int BKE = 4 * CH;
    constexpr int NTHR = 512;
    constexpr int RP = NTHR / 4;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int BNP = (BN + RP - 1) / RP * RP;
    constexpr int A_PER = BM / RP;
    constexpr int B_PER = BNP / RP;
    constexpr int LPS = A_PER + B_PER;
    constexpr int STAGE = (BM + BNP) * 64;
    static_assert(WM * WN == 8 && BM % RP == 0 && NSTAGE >= 4, "geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const bool late = wv >= 4;

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    const T* rowp[A_PER];
    unsigned vmask[A_PER];
    const int HoWo = a.Ho * a.Wo;
    const int ntap = a.KH * a.KW;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + r0 + RP * i;
        unsigned mk = 0;
        const T* rp = zero;
        if (m < a.M) {
            int img, rem, oh, ow;
            fast_divmod(m, HoWo, inv_howo, img, rem);
            fast_divmod(rem, a.Wo, inv_wo, oh, ow);
            const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
            rp = xg + (((long)img * a.H + ih0) * a.W + iw0) * a.x_cs + a.x_coff;
            const int lo_w = max(0, -iw0), hi_w = min(a.KW, a.W - iw0);
            const int lo_h = max(0, -ih0), hi_h = min(a.KH, a.H - ih0);
            if (hi_w > lo_w && hi_h > lo_h) {
                const unsigned vw = ((1u << hi_w) - 1u) & ~((1u << lo_w) - 1u);
                const unsigned rows = (((1u << (hi_h * a.KW)) - 1u) & ~((1u << (lo_h * a.KW)) - 1u)) & a.tap_rows;
                mk = vw * rows;
            }
        }
        rowp[i] = rp;
        vmask[i] = mk;
    }
    const int nsteps = a.Kp / BKE;
    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    int xoff[MT], woff[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xoff[i] = lds_off((wm * MT + i) * 16 + r, q);
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = BM * 64 + lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);
    typedef typename Frag<T>::type frag_t;
    char* const sdst = smem + (16 * wv) * 64;

    const int csteps = a.Cin / BKE;
    const T* aptr[A_PER];
    int ainc[A_PER];
    const T* wptr[B_PER];
    int winc[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const bool okr = r0 + RP * j < BN;
        wptr[j] = okr ? wg + (size_t)(n0 + r0 + RP * j) * a.Kp + kc * CH : zero;
        winc[j] = okr ? BKE : 0;
    }
    int tap = 0, kh = 0, kw = 0, cc = 0;
    auto set_tap = [&] {
        const long toff = ((long)kh * a.W + kw) * a.x_cs + kc * CH;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const bool ok = tap < ntap && ((vmask[i] >> (tap & 31)) & 1u);
            aptr[i] = ok ? rowp[i] + toff : zero;
            ainc[i] = ok ? BKE : 0;
        }
    };
    set_tap();
    auto issue = [&](int stage) {
        char* sbase = sdst + stage * STAGE;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)aptr[i], (lptr_t)(sbase + i * (RP * 64)), 16, 0, 0);
            aptr[i] += ainc[i];
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(sbase + BM * 64 + j * (RP * 64)), 16, 0, 0);
            wptr[j] += winc[j];
        }
        if (++cc == csteps) {
            cc = 0;
            ++tap;
            if (++kw == a.KW) { kw = 0; ++kh; }
            set_tap();
        }
    };
#pragma unroll
    for (int st = 0; st < NSTAGE - 2; ++st) issue(st);
    wait_vmcnt<(NSTAGE - 3) * LPS>();          // step 0 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();              // ... everyone's
    PP_STAMP(1);
    if (late) __builtin_amdgcn_s_barrier();    // waves 4..7 now run one segment behind

    for (int step0 = 0; step0 < nsteps; step0 += NSTAGE) {
#pragma unroll
        for (int u = 0; u < NSTAGE; ++u) {
            if (step0 + u < nsteps) {
                // ---- LOAD segment
                issue((u + NSTAGE - 2) % NSTAGE);
                const char* base = smem + u * STAGE;
                frag_t xf[MT], wf[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(base + woff[j]);
#pragma unroll
                for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(base + xoff[i]);
                wait_vmcnt<(NSTAGE - 3) * LPS>();      // step+1 has landed (this wave's part)
                __builtin_amdgcn_s_barrier();
                // ---- COMPUTE segment
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = Frag<T>::mma(wf[j], xf[i], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    if (!late) __builtin_amdgcn_s_barrier();   // every wave executes the same number of barriers
    wait_vmcnt<0>();
    PP_STAMP(2);

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wm * MT + i) * 16 + r;
        mrow[i] = m < a.M ? m : -1;
    }
    epilogue_dispatch<T, MT, NT, true>(a, acc, mrow, n0 + wn * NT * 16, q);
    if (g_pp_times_on) { wait_vmcnt<0>(); PP_STAMP(3); }
}

static void pp_times_report(hipStream_t s, int nblk) {
    static std::vector<unsigned long long> h(4 * 4096);
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pp_times), sizeof(unsigned long long) * 4 * 4096));
    nblk = std::min(nblk, 4096);
    unsigned long long t0 = ~0ull, t3 = 0;
    double d[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int b = 0; b < nblk; ++b) {
        t0 = std::min(t0, h[4 * b]);
        t3 = std::max(t3, h[4 * b + 3]);
        for (int k = 0; k < 3; ++k) {
            const double v = (double)(h[4 * b + k + 1] - h[4 * b + k]) * 0.01;
            d[k] += v / nblk;
            mx[k] = std::max(mx[k], v);
        }
    }
    double start_spread = 0, end_spread = 0;
    for (int b = 0; b < nblk; ++b) {
        start_spread = std::max(start_spread, (double)(h[4 * b] - t0) * 0.01);
        end_spread = std::max(end_spread, (double)(t3 - h[4 * b + 3]) * 0.01);
    }
    fprintf(stderr, "[pp_times] blocks %d: prologue %.2f (max %.2f) us, k-loop %.2f (max %.2f), epilogue+drain %.2f (max %.2f); first start -> last end %.2f us; start spread %.2f, end spread %.2f\n",
            nblk, d[0], mx[0], d[1], mx[1], d[2], mx[2], (double)(t3 - t0) * 0.01, start_spread, end_spread);
}

template <typename T, int MT, int NT, int WM, int WN, int NSTAGE>
static void launch_pp(const ConvArgs& a, hipStream_t s) {
    constexpr int RP = 128;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + RP - 1) / RP * RP;
    dim3 grid(ceil_div(a.M, BM), ceil_div(a.Cout, BN));
    constexpr size_t lds = (size_t)NSTAGE * (BM + BNP) * 64;
    static_assert(lds <= 160 * 1024, "ring does not fit the LDS");
    auto kfn = conv_igemm_pp_kernel<T, MT, NT, WM, WN, NSTAGE>;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    static const bool times = getenv("AICAM_PP_TIMES") != nullptr;
    if (times) {
        const int on = 1;
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_pp_times_on), &on, sizeof(int)));
    }
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a);
    KCHECK();
    if (times) pp_times_report(s, (int)grid.x);
}

// ------------------------------------------------------------------------------------------------
// v5 "ping-pong patch" for 3x3 / stride 1 / pad 1 with Cin a multiple of the K-step: the v4 schedule (load / compute
// segments, late half one barrier behind, NSTAGE-deep weight ring), but the pixel operand no longer travels as an
// im2col tile (every input chunk fetched 9x through L2 -> LDS: 80 % of v4's LDS-DMA bytes on ReID layer2).  K is walked
// (channel chunk, tap): for one chunk of BKE input channels the (TH+2) x (TW+2) halo patches of the tile's NI images
// sit in LDS and the nine taps are ds_read at shifted addresses; the next chunk's patch streams into the second
// buffer one LDS-DMA per thread per K-step while this one is consumed.
//  * patch image: four PLANES (one per 16-byte K sub-chunk q), plane q holds chunk q of every patch pixel at
//    pixel*16: a lane group (16 consecutive pixels, fixed q) reads 256 contiguous bytes -- conflict-free with no
//    swizzle -- and the tap shift (kh*PW + kw)*16 is a ds_read immediate: ONE address register per pixel tile;
//  * LDS-DMA stays lane-linear: wave w of a pass writes plane w&3, 64 consecutive pixels;
//  * every L segment issues exactly B_PER weight loads + 1 patch load (a zero-page load into a dummy slot when no
//    patch pass is due), so the counted vmcnt of v4 is unchanged.
// Hazards (segments as in v4): patch passes of chunk c+1 are issued in L_{9c+1} .. L_{9c+NPASS} (NPASS <= 7): the
// buffer was last read in L_{9c-1} (WAR: 4 segments), and the last pass is waited for in L_{9c+8}, read in L_{9c+9}.
template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE>
__global__ __launch_bounds__(512) void conv3x3_pp_patch_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int BKE = 4 * CH;
    constexpr int RP = 128;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int BNP = (BN + RP - 1) / RP * RP;
    constexpr int B_PER = BNP / RP;
    constexpr int LPS = B_PER + 1;
    constexpr int TPIX = TH * TW, NI = BM / TPIX;
    constexpr int PW = TW + 2, PH = TH + 2, IPIX = PW * PH, NPIX = NI * IPIX;
    constexpr int NPASS = (NPIX + 127) / 128, NPIXP = NPASS * 128;
    constexpr int PLANE = NPIXP * 16, PBUF = 4 * PLANE, DUMMY = 8192, WSTAGE = BNP * 64;
    constexpr int RING = 2 * PBUF + DUMMY;
    static_assert(WM * WN == 8 && BM % TPIX == 0 && NPASS <= 11 - NSTAGE && TW % 4 == 0 && NSTAGE >= 4, "geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool late = wv >= 4;
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img0 = (bx / tiles_y) * NI;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int n0 = blockIdx.y * BN;
    const int n_img = a.M / (a.Ho * a.Wo);

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    // ---- patch passes: pass i, wave w -> plane w&3, pixels i*128 + (w>>2)*64 + lane
    const int plane = wv & 3;
    int poff[NPASS];                       // element offset of this thread's chunk at channel chunk 0, or -1 (zero page)
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int p = i * 128 + (wv >> 2) * 64 + lane;
        const int il = p / IPIX, rem = p - il * IPIX;
        const int py = rem / PW, px = rem - py * PW;
        const int img = img0 + il, iy = oy0 + py - 1, ix = ox0 + px - 1;
        const bool ok = p < NPIX && img < n_img && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        poff[i] = ok ? (((img * a.H + iy) * a.W + ix) * a.x_cs + a.x_coff + plane * CH) : -1;
    }
    char* const pdst = smem + plane * PLANE + (wv >> 2) * 1024;     // + buffer*PBUF + pass*2048 (+ lane*16 by the DMA)
    auto issue_patch = [&](int i, int buf, int chunk_off) {        // i: compile-time pass index
        const T* src = poff[i] >= 0 ? xg + poff[i] + chunk_off : zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(pdst + buf * PBUF + i * 2048), 16, 0, 0);
    };
    auto issue_dummy = [&] {
        const T* src = zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + 2 * PBUF + wv * 1024), 16, 0, 0);
    };

    // ---- weight stream: row r0 + RP*j of the block's channel tile, K offset of step (chunk c, tap) = tap*Cin + c*BKE
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    int wofs[B_PER];                          // element offset of this thread's weight chunk at K = 0, or -1 (zero page)
#pragma unroll
    for (int j = 0; j < B_PER; ++j) wofs[j] = (r0 + RP * j < BN) ? (n0 + r0 + RP * j) * a.Kp + kc * CH : -1;
    const int nchunks = a.Cin / BKE, nsteps = 9 * nchunks;
    char* const wdst = smem + RING + (16 * wv) * 64;
    int is_c = 0, is_tap = 0, is_k = 0, is_st = 0;   // the step whose weights are fetched next (and its ring stage)
    auto issue_w = [&] {
        const int koff = is_tap * a.Cin + is_c * BKE;
        char* sbase = wdst + is_st * WSTAGE;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const T* src = (wofs[j] >= 0 && is_k < nsteps) ? wg + wofs[j] + koff : zero;
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + j * (RP * 64)), 16, 0, 0);
        }
        ++is_k;
        if (++is_st == NSTAGE) is_st = 0;
        if (++is_tap == 9) { is_tap = 0; ++is_c; }
    };

    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;
    // LDS byte address of (this lane's pixel of tile 0, tap (0,0)) in buffer 0; tile i sits a compile-time distance away
    // because a wave's MT*16 pixels either tile whole images or lie inside one (static_assert below)
    static_assert((MT * 16) % TPIX == 0 || TPIX % (MT * 16) == 0, "a wave's pixels must not straddle images irregularly");
    static_assert(TW % 16 == 0 || 16 % TW == 0, "a 16-pixel MFMA tile is whole rows or a piece of one row");
    auto patch_pix = [](int m) constexpr { return (m / TPIX) * IPIX + ((m % TPIX) / TW) * PW + (m % TPIX) % TW; };
    int xa0;
    {
        const int ml = wm * MT * 16 + r;
        const int il = ml / TPIX, rem = ml - il * TPIX;
        const int ly = rem / TW, lx = rem - ly * TW;
        xa0 = q * PLANE + (il * IPIX + ly * PW + lx) * 16;
    }
    // weight fragment addresses: tiles j and j+2 are 32 rows (2048 B) apart, j and j+1 differ in the swizzle term
    int woff2[2];
#pragma unroll
    for (int j = 0; j < 2 && j < NT; ++j) woff2[j] = RING + lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);
    static_assert(NT % 2 == 0, "tile pairs");

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    typedef typename Frag<T>::type frag_t;

    // ---- prologue: patch chunk 0, weights of steps 0 and 1
#pragma unroll
    for (int i = 0; i < NPASS; ++i) issue_patch(i, 0, 0);
#pragma unroll
    for (int st = 0; st < NSTAGE - 2; ++st) {
        issue_w();
        if (st) issue_dummy();                 // every set in flight has LPS loads: the counted waits below rely on it
    }
    wait_vmcnt<(NSTAGE - 3) * LPS>();          // patch chunk 0 and the weights of step 0 have landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();

    int rd_st = 0;                             // ring stage of the step being computed
    auto chunk = [&](int c, auto bufc) {       // bufc: compile-time parity of the patch buffer read in this chunk
        constexpr int BUF = decltype(bufc)::value;
        const bool more = c + 1 < nchunks;
        const int noff = (c + 1) * BKE;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // ---- LOAD segment
            issue_w();
            if (tap >= 1 && tap <= NPASS && more) issue_patch(tap >= 1 && tap <= NPASS ? tap - 1 : 0, BUF ^ 1, noff);
            else issue_dummy();
            const int so = rd_st * WSTAGE;
            if (++rd_st == NSTAGE) rd_st = 0;
            const int tapoff = BUF * PBUF + ((tap / 3) * PW + tap % 3) * 16;
            frag_t xf[MT], wf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(smem + woff2[j & 1] + so + (j >> 1) * 2048);
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(smem + xa0 + tapoff + patch_pix(16 * i) * 16);
            wait_vmcnt<(NSTAGE - 3) * LPS>();
            __builtin_amdgcn_s_barrier();
            // ---- COMPUTE segment
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = Frag<T>::mma(wf[j], xf[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    };
    for (int c = 0; c < nchunks; c += 2) {     // Cin / BKE is even for every layer that reaches this kernel
        chunk(c, std::integral_constant<int, 0>{});
        chunk(c + 1, std::integral_constant<int, 1>{});
    }
    if (!late) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ml = (wm * MT + i) * 16 + r;
        const int il = ml / TPIX, rem = ml - il * TPIX;
        const int ly = rem / TW, lx = rem - ly * TW;
        const int img = img0 + il;
        mrow[i] = img < n_img ? (img * a.Ho + oy0 + ly) * a.Wo + ox0 + lx : -1;
    }
    epilogue_dispatch<T, MT, NT, true>(a, acc, mrow, n0 + wn * NT * 16, q);
}

template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE>
static bool launch_pp_patch(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + 127) / 128 * 128;
    constexpr int NI = BM / (TH * TW), NPIX = NI * (TH + 2) * (TW + 2), NPASS = (NPIX + 127) / 128;
    constexpr size_t lds = (size_t)2 * 4 * NPASS * 128 * 16 + 8192 + (size_t)NSTAGE * BNP * 64;
    static_assert(lds <= 160 * 1024, "does not fit the LDS");
    if (a.H % TH || a.W % TW || a.Ho != a.H || a.Wo != a.W) return false;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    const int n_img = a.M / (a.Ho * a.Wo);
    auto kfn = conv3x3_pp_patch_kernel<T, MT, NT, WM, WN, TH, TW, NSTAGE>;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    dim3 grid(ceil_div(n_img, NI) * tiles_x * tiles_y, ceil_div(a.Cout, BN));
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a, tiles_x, tiles_y);
    KCHECK();
    return true;
}

// 3x3/s1/p1 layers whose map tiles exactly: pick the tile by map shape and Cout (ReID layer1..4 shapes and their multiples).
template <typename T>
static bool try_pp_patch(const ConvArgs& a, hipStream_t s) {
    // bit 0: Cout 64 (slower than the 4-wave patch kernel: 16 MFMAs per segment), 1: Cout 128, 2: Cout % 256, 3: deeper ring (no gain)
    static const int mode = [] { const char* e = getenv("AICAM_PPP"); return e ? atoi(e) : 6; }();
    static const int pp_min = [] { const char* e = getenv("AICAM_PP_MIN"); return e ? atoi(e) : 200; }();
    constexpr int BKE = 64 / (int)sizeof(T);
    if (!mode || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % (2 * BKE)) return false;
    if ((long)a.M * a.x_cs >= (1l << 31)) return false;                       // 32-bit element offsets inside the kernel
    const int c = a.Cout;
    const bool deep = mode & 8;
    if (c == 64 && (mode & 1) && a.M / 512 >= pp_min) return deep ? launch_pp_patch<T, 4, 4, 8, 1, 16, 32, 6>(a, s) : launch_pp_patch<T, 4, 4, 8, 1, 16, 32, 4>(a, s);
    if (c == 128 && (mode & 2) && a.M / 512 >= pp_min) return deep ? launch_pp_patch<T, 8, 4, 4, 2, 32, 16, 6>(a, s) : launch_pp_patch<T, 8, 4, 4, 2, 32, 16, 4>(a, s);
    if (c % 256 == 0 && (mode & 4) && (long)(a.M / 256) * (c / 256) >= pp_min) {
        if (a.H % 16 == 0 && a.W % 8 == 0) return deep ? launch_pp_patch<T, 8, 4, 2, 4, 16, 8, 6>(a, s) : launch_pp_patch<T, 8, 4, 2, 4, 16, 8, 4>(a, s);
        return deep ? launch_pp_patch<T, 8, 4, 2, 4, 8, 4, 5>(a, s) : launch_pp_patch<T, 8, 4, 2, 4, 8, 4, 4>(a, s);
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// Direct 3x3 (pad 1, stride 1 or 2) for 16 input channels, fp16, SiLU: YOLOv8n's P1/P2 layers (1.conv, 2.c2f.m0.cv1/cv2).
// Through the implicit GEMM these run 3-4x above their HBM floor: K = 144 straddles taps inside a K-step (generic
// gather path, ~12 VALU per 16-byte LDS-DMA) and every input pixel is fetched 9 times for 16-32 output channels.
// Here a block owns 8 x 32 output pixels of one image: the input patch is read once into LDS, the weights (<= 9 KB)
// live in registers as MFMA A fragments, K is the natural (tap, 16 ch) order so one v_mfma_f32_16x16x32_f16 eats
// two taps and the B fragment of a lane is ONE aligned 16-byte ds_read (its tap's channel half).
// LDS entry (row, column parity p, channel half h, column c2) = 16 bytes; for stride 2 even and odd input columns
// are kept apart so that 16 consecutive output pixels read 16 consecutive entries (conflict-free).
template <int COUT, int S>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int TH = 8, TW = 32, NCT = COUT / 16;
    constexpr int PR = (TH - 1) * S + 3, PC = (TW - 1) * S + 3;
    constexpr int NPAR = S, PCP = (PC + NPAR - 1) / NPAR;
    static_assert((S == 1 || S == 2) && (COUT == 16 || COUT == 32), "variants");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 15, q = lane >> 4;
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const half_t* xg = reinterpret_cast<const half_t*>(a.x) + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;

    for (int idx = t; idx < PR * PC * 2; idx += 256) {
        const int h = idx & 1, pp = idx >> 1;
        const int pr = pp / PC, pc = pp - pr * PC;
        const int iy = iy0 + pr, ix = ix0 + pc;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
            v = *reinterpret_cast<const uint4*>(xg + ((size_t)iy * a.W + ix) * a.x_cs + h * 8);
        const int par = pc % NPAR, c2 = pc / NPAR;
        *reinterpret_cast<uint4*>(smem + (((pr * NPAR + par) * 2 + h) * PCP + c2) * 16) = v;
    }

    // A fragments: MFMA m covers K = 32m .. 32m+31 = taps 2m, 2m+1 x 16 channels (Kp = 160: k >= 144 are zero rows)
    const half_t* wg = reinterpret_cast<const half_t*>(a.w);
    half8 wa[NCT][5];
    floatx4 bi[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const half_t* wr = wg + (size_t)perm_row<NCT>(ct, r) * a.Kp + 8 * q;
#pragma unroll
        for (int m = 0; m < 5; ++m) wa[ct][m] = *reinterpret_cast<const half8*>(wr + 32 * m);
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[ct][e] = a.bias[perm_ch<NCT>(ct, q, e)];
    }
    int d[5];                                  // LDS offset of this lane's (tap, channel half) relative to its output pixel
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int tap = min(2 * m + (q >> 1), 8), kh = tap / 3, kw = tap - 3 * kh, h = q & 1;
        d[m] = (((kh * NPAR + kw % NPAR) * 2 + h) * PCP + kw / NPAR) * 16;
    }
    __syncthreads();

    half_t* yg = reinterpret_cast<half_t*>(a.y);
    const half_t* rg = reinterpret_cast<const half_t*>(a.res);
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int oyl = 2 * wv + (tile >> 1), oxl = (tile & 1) * 16 + r;
        const int base = (oyl * S * NPAR * 2 * PCP + oxl) * 16;
        half8 xb[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) xb[m] = *reinterpret_cast<const half8*>(smem + base + d[m]);
        const size_t pix = ((size_t)img * a.Ho + oy0 + oyl) * a.Wo + ox0 + oxl;
        float v[NCT][4];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            floatx4 acc = bi[ct];
#pragma unroll
            for (int m = 0; m < 5; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ct][m], xb[m], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[ct][e] = act_fast<1>(acc[e]);
        }
        if constexpr (NCT == 1) {
            const int ch = 4 * q;
            if (a.res_mode == 2) {
                const half4 h = *reinterpret_cast<const half4*>(rg + pix * a.r_cs + a.r_coff + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[0][e] += (float)h[e];
            }
            const half4 o = {(half_t)v[0][0], (half_t)v[0][1], (half_t)v[0][2], (half_t)v[0][3]};
            *reinterpret_cast<half4*>(yg + pix * a.y_cs + a.y_coff + ch) = o;
        } else {
            const int ch = 8 * q;                                       // perm_ch<2>: tiles 0, 1 -> channels 8q + 4*(ct) + e
            if (a.res_mode == 2) {
                const half8 h = *reinterpret_cast<const half8*>(rg + pix * a.r_cs + a.r_coff + ch);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] += (float)h[e];
            }
            const half8 o = {(half_t)v[0][0], (half_t)v[0][1], (half_t)v[0][2], (half_t)v[0][3],
                             (half_t)v[1][0], (half_t)v[1][1], (half_t)v[1][2], (half_t)v[1][3]};
            *reinterpret_cast<half8*>(yg + pix * a.y_cs + a.y_coff + ch) = o;
        }
    }
}

template <int COUT, int S>
static void launch_c16(const ConvArgs& a, hipStream_t s) {
    constexpr int PR = 7 * S + 3, PC = 31 * S + 3, PCP = (PC + S - 1) / S;
    constexpr size_t lds = (size_t)PR * S * 2 * PCP * 16;
    const int tiles_x = a.Wo / 32, tiles_y = a.Ho / 8, n_img = a.M / (a.Ho * a.Wo);
    hipLaunchKernelGGL((conv3x3_c16_kernel<COUT, S>), dim3(n_img * tiles_x * tiles_y), dim3(256), lds, s, a, tiles_x, tiles_y);
    KCHECK();
}

static bool try_c16(const ConvArgs& a, hipStream_t s) {
    static const bool off = getenv("AICAM_NO_C16") != nullptr;
    if (off || a.KH != 3 || a.KW != 3 || a.pad != 1 || a.Cin != 16 || a.act != 1 || a.out_f32 || (a.res_mode != 0 && a.res_mode != 2)) return false;
    if (a.Wo % 32 || a.Ho % 8 || a.Kp != 160 || (a.x_cs | a.x_coff | a.y_cs | a.y_coff | a.r_cs | a.r_coff) % 8) return false;
    if (a.stride == 1 && (a.Ho != a.H || a.Wo != a.W)) return false;
    if (a.stride == 2 && (a.Ho != (a.H + 1) / 2 || a.Wo != (a.W + 1) / 2)) return false;
    if (a.Cout == 16 && a.stride == 1) launch_c16<16, 1>(a, s);
    else if (a.Cout == 32 && a.stride == 2) launch_c16<32, 2>(a, s);
    else if (a.Cout == 16 && a.stride == 2) launch_c16<16, 2>(a, s);
    else if (a.Cout == 32 && a.stride == 1) launch_c16<32, 1>(a, s);
    else return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Persistent, weights-resident 3x3 / stride 1 / pad 1 for Cin = Cout = 64, fp16 (ReID layer1: 21 % of the FLOPs, tensors
// of 1 GB per 128-frame launch group, K = 576 only).  A tile's K loop is too short to amortise a block's prologue and
// epilogue (conv3x3_patch_kernel: 38 % MFMA busy), and every block re-fetches the 72 KB of weights through L2 -> LDS.
// Here one 8-wave block per CU walks many 8 x 32-pixel tiles:
//  * the weights never touch LDS: wave w keeps the A fragments of its 32 output channels (half w>>2) for all 18
//    K-steps in 144 VGPRs, loaded once per kernel;
//  * the input patch (10 x 34 pixels x 64 channels, eight 16-byte planes, conflict-free, tap shift = ds_read
//    immediate) is triple-buffered: the LDS-DMA of tile t+2 is issued right after the one barrier of tile t and has
//    two tiles of MFMAs (144 per wave each) to land; no barrier and no global load inside the K loop;
//  * waves w and w+4 share a SIMD and a pixel group (same B fragments, other channel half), so one wave's ds_reads
//    and epilogue sit under its partner's MFMAs.
template <int ACT, int RES>
__global__ __launch_bounds__(512) void conv3x3_c64_resident_kernel(const ConvArgs a, int n_tiles, int tiles_x, int tiles_y) {
    constexpr int TH = 8, TW = 32, PW = TW + 2, PH = TH + 2, NPIX = PW * PH, NPASS = (NPIX + 63) / 64, NPIXP = NPASS * 64;
    constexpr int PLANE = NPIXP * 16, PBUF = 8 * PLANE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 15, q = lane >> 4;
    const int pg = wv & 3, ch = wv >> 2;
    const half_t* __restrict__ xg = reinterpret_cast<const half_t*>(a.x);
    const half_t* __restrict__ wg = reinterpret_cast<const half_t*>(a.w);
    const half_t* zero = reinterpret_cast<const half_t*>(a.zero);

    // ---- weights: A fragments of channels 32*ch + perm_row<2>(j, rho) for K-step s = (tap, channel half cc): k = 32 s + 8 q
    half8 wreg[18][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const half_t* wr = wg + (size_t)(32 * ch + perm_row<2>(j, r)) * a.Kp + 8 * q;
#pragma unroll
        for (int s2 = 0; s2 < 18; ++s2) wreg[s2][j] = *reinterpret_cast<const half8*>(wr + 32 * s2);
    }
    floatx4 bi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[j][e] = a.bias[32 * ch + perm_ch<2>(j, q, e)];

    // tile order: the tiles of one image stay on one XCD (blocks b, b+8, ... share an L2) so halo rows are L2 hits
    const int tpi = tiles_x * tiles_y;
    auto tile_of = [&](int k) -> int {
        if (tpi == 8 && (gridDim.x & 63) == 0) {
            const int xcd = blockIdx.x & 7, sl = blockIdx.x >> 3, per = gridDim.x >> 6;     // images in flight per XCD
            const int im = ((sl >> 3) + per * k) * 8 + xcd;
            return im * 8 + (sl & 7);
        }
        return blockIdx.x + k * gridDim.x;
    };
    auto issue_patch = [&](int tile, int buf) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / tpi;
        const int oy0 = ty * TH, ox0 = tx * TW;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int p = i * 64 + lane;
            const int py = p / PW, px = p - py * PW;
            const int iy = oy0 + py - 1, ix = ox0 + px - 1;
            const bool ok = p < NPIX && tile < n_tiles && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const half_t* src = ok ? xg + ((size_t)(img * a.H + iy) * a.W + ix) * a.x_cs + a.x_coff + wv * 8 : zero;
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * PBUF + wv * PLANE + i * 1024), 16, 0, 0);
        }
    };

    int xa[4];                                  // this lane's pixel of MFMA tile i in plane q of buffer 0, tap (0,0)
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = q * PLANE + ((2 * pg + (i >> 1)) * PW + (i & 1) * 16 + r) * 16;

    half_t* yg = reinterpret_cast<half_t*>(a.y);
    const half_t* rg = reinterpret_cast<const half_t*>(a.res);
    const int trips = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;      // same trip count for every block
    issue_patch(tile_of(0), 0);
    issue_patch(tile_of(1), 1);                                             // (zero page when past the end)
    int buf = 0;                                                            // k % 3
    for (int k = 0; k < trips; ++k) {
        const int tile = tile_of(k);
        // In flight, oldest first: patch k | stores k-2 | patch k+1 | stores k-1.  Leaving NPASS operations
        // outstanding retires patch k for certain (a conservative count: it also retires the head of patch k+1).
        wait_vmcnt<NPASS>();
        __builtin_amdgcn_s_barrier();           // patch k complete for everyone; everyone is done reading buffer (k+2) % 3
        {
            int nb = buf + 2; if (nb >= 3) nb -= 3;
            issue_patch(tile_of(k + 2), nb);
        }
        if (tile < n_tiles) {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / tpi;
            const size_t pix0 = ((size_t)img * a.Ho + ty * TH) * a.Wo + tx * TW;
            const int boff = buf * PBUF;
            half8 rv[4];                        // residual vectors: requested now, they land under the K loop
            if constexpr (RES == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t pix = pix0 + (size_t)(2 * pg + (i >> 1)) * a.Wo + (i & 1) * 16 + r;
                    rv[i] = *reinterpret_cast<const half8*>(rg + pix * a.r_cs + a.r_coff + 32 * ch + 8 * q);
                }
            }
            floatx4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = bi[j];
#pragma unroll
            for (int s2 = 0; s2 < 18; ++s2) {
                const int tap = s2 >> 1, cc = s2 & 1, kh = tap / 3, kw = tap - 3 * kh;
                const int off = cc * 4 * PLANE + (kh * PW + kw) * 16;
                half8 xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const half8*>(smem + xa[i] + boff + off);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[s2][j], xf[i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t pix = pix0 + (size_t)(2 * pg + (i >> 1)) * a.Wo + (i & 1) * 16 + r;
                const int n = 32 * ch + 8 * q;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[i][e >> 2][e & 3];
                if constexpr (RES == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[i][e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_fast<ACT>(v[e]);
                const half8 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
                *reinterpret_cast<half8*>(yg + pix * a.y_cs + a.y_coff + n) = o;
            }
        }
        if (++buf == 3) buf = 0;
    }
    wait_vmcnt<0>();
}

static bool try_c64_resident(const ConvArgs& a, hipStream_t s) {
    static const int on = [] { const char* e = getenv("AICAM_C64R"); return e ? atoi(e) : 1; }();   // 0: off, 1 (default): layers without residual (+6 % on them), 2: also with residual (slower: its loads are exposed)
    if (!on || (a.res_mode != 0 && on < 2) || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin != 64 || a.Cout != 64 || a.out_f32 || a.Kp != 576) return false;
    if (a.W % 32 || a.H % 8 || a.Ho != a.H || a.Wo != a.W || a.M < 1500000 || (long)a.M * a.x_cs >= (1l << 31)) return false;
    if ((a.x_cs | a.x_coff | a.y_cs | a.y_coff | a.r_cs | a.r_coff) % 8) return false;
    const int tiles_x = a.W / 32, tiles_y = a.H / 8, n_img = a.M / (a.H * a.W), n_tiles = n_img * tiles_x * tiles_y;
    constexpr size_t lds = (size_t)3 * 8 * 384 * 16;
    auto launch = [&](auto kfn) {
        static bool attr = false;
        if (!attr) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        hipLaunchKernelGGL(kfn, dim3(256), dim3(512), lds, s, a, n_tiles, tiles_x, tiles_y);
        KCHECK();
    };
    if (a.act == 2 && a.res_mode == 0) launch(conv3x3_c64_resident_kernel<2, 0>);
    else if (a.act == 2 && a.res_mode == 1) launch(conv3x3_c64_resident_kernel<2, 1>);
    else return false;
    return true;
}

static int conv_impl() {   // AICAM_CONV=v1 selects the register-staged kernel (A/B and fallback)
    static int v = [] { const char* e = getenv("AICAM_CONV"); return (e && e[0] == 'v' && e[1] == '1') ? 1 : 2; }();
    return v;
}

// ------------------------------------------------------------------------------------------------
// v3 for 3x3 / stride 1 / pad 1: the im2col gather of v2 fetches every input chunk 9 times (once per
// tap) through L2 -> LDS.  Here a block owns a TH x TW tile of output pixels of ONE image; the
// (TH+2) x (TW+2) x Cin input patch (with its halo; zero page outside the image) is pulled into LDS
// once by LDS-DMA and the nine taps are generated from LDS at shifted addresses.  Only the weights
// stream through the NSTAGE ring.  K order = tap-major, Cin/BKE steps per tap (needs Cin % BKE == 0).
// Patch image: pixel p, 16-byte chunk j stored at chunk slot j ^ swz(p) (source-side swizzle again);
// swz(p) = p & (CPP-1) (CPP = chunks per pixel >= 8) or (p>>1)&3 (CPP == 4): conflict-free /
// <= 2-way for the ds_read_b128 lane groups (16 consecutive pixels x 4 consecutive chunks).
template <int CPP> __device__ __forceinline__ int patch_swz(int p) { return CPP == 4 ? ((p >> 1) & 3) : (p & (CPP - 1)); }

// Everything the hot loop needs is a compile-time constant or a precomputed register:
//  * LGCPP: log2 of the 16-byte chunks per pixel (Cin fixed per instantiation), CSTEPS = CPP/4 K-steps per tap;
//  * patch rows are padded to PWP pixels, a multiple of max(8, CPP): the swizzle term of a patch pixel
//    then depends on its column only, so the LDS address of (tile i, tap column kw, chunk cc) is one of
//    3*CSTEPS*MT precomputed VGPRs and the tap row kh is a ds_read immediate;
//  * taps, chunks and ring stages are fully unrolled; the weight stream is a pointer increment.
// VALU per MFMA drops from ~12 to <1 (SQ_INSTS_VALU / SQ_INSTS_MFMA, profiles/).
template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE, int LGCPP>
__global__ __launch_bounds__(64 * WM * WN) void conv3x3_patch_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int BKE = 4 * CH;
    constexpr int NTHR = 64 * WM * WN;
    constexpr int RP = NTHR / 4;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int BNP = (BN + RP - 1) / RP * RP;
    constexpr int B_PER = BNP / RP;
    constexpr int WSTAGE = BNP * 64;
    constexpr int CPP = 1 << LGCPP, CSTEPS = CPP / 4, NSTEPS = 9 * CSTEPS;
    constexpr int PAL = CPP >= 8 ? CPP : 8;
    constexpr int PH = TH + 2, PWP = (TW + 2 + PAL - 1) / PAL * PAL;
    constexpr int TOTAL = PH * PWP * CPP;
    constexpr int PATCH_BYTES = (TOTAL + NTHR - 1) / NTHR * NTHR * 16;
    constexpr int ROWB = PWP * CPP * 16;                    // bytes per patch row
    static_assert(BM == TH * TW && TW % 16 == 0 && (TW & (TW - 1)) == 0 && B_PER == 1, "tile geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem + PATCH_BYTES;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int n0 = blockIdx.y * BN;

    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);
    const T* ximg = reinterpret_cast<const T*>(a.x) + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;

    // ---- the input patch, once (pad columns and out-of-image pixels come from the zero page)
#pragma unroll 2
    for (int base = 0; base < TOTAL; base += NTHR) {
        const int L = base + t;
        const int p = L >> LGCPP, sl = L & (CPP - 1);
        const int j = sl ^ patch_swz<CPP>(p);
        const int py = p / PWP, px = p - py * PWP;
        const int iy = oy0 + py - 1, ix = ox0 + px - 1;
        const bool ok = L < TOTAL && px < TW + 2 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const T* src = ok ? ximg + ((size_t)iy * a.W + ix) * a.x_cs + j * CH : zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (size_t)(base + 64 * wv) * 16), 16, 0, 0);
    }

    // ---- weight stream: one 16-byte chunk per thread per K-step; rows past Cout read the zero page with stride 0
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    const bool wrow_ok = r0 < BN;
    const T* wptr = wrow_ok ? wg + (size_t)(n0 + r0) * a.Kp + kc * CH : zero;
    const int winc = wrow_ok ? BKE : 0;
    char* wdst = ring + (16 * wv) * 64;
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) {
        __builtin_amdgcn_global_load_lds((gptr_t)wptr, (lptr_t)(wdst + st * WSTAGE), 16, 0, 0);
        wptr += winc;
    }

    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;
    // LDS byte address of this lane's 16-byte operand chunk for (tap column kw, K-chunk cc, pixel tile i), tap row 0
    int xaddr[3][CSTEPS][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pt = (wm * MT + i) * 16 + r;
        const int ly = pt / TW, lx = pt % TW;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int p0 = ly * PWP + lx + kw;
            const int sw = patch_swz<CPP>(p0);
#pragma unroll
            for (int cc = 0; cc < CSTEPS; ++cc) xaddr[kw][cc][i] = (p0 * CPP + ((cc * 4 + q) ^ sw)) * 16;
        }
    }
    int woff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = PATCH_BYTES + lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    typedef typename Frag<T>::type frag_t;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
            for (int cc = 0; cc < CSTEPS; ++cc) {
                constexpr int dummy = 0; (void)dummy;
                const int step = (kh * 3 + kw) * CSTEPS + cc;       // compile-time after unrolling
                const int cur = step % NSTAGE, nxt = (step + NSTAGE - 1) % NSTAGE;
                wait_vmcnt<(NSTAGE - 2) * B_PER>();
                __builtin_amdgcn_s_barrier();
                {   // refill the stage that step-1 released (zero page once the real K-steps are exhausted)
                    const T* src = (step + NSTAGE - 1 < NSTEPS) ? wptr : zero;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wdst + nxt * WSTAGE), 16, 0, 0);
                    wptr += winc;
                }
                frag_t xf[MT], wf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(smem + xaddr[kw][cc][i] + kh * ROWB);
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(smem + woff[j] + cur * WSTAGE);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = Frag<T>::mma(wf[j], xf[i], acc[i][j]);
            }
        }
    }
    wait_vmcnt<0>();

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pt = (wm * MT + i) * 16 + r;
        const int oy = oy0 + pt / TW, ox = ox0 + pt % TW;
        mrow[i] = (oy < a.Ho && ox < a.Wo) ? (img * a.Ho + oy) * a.Wo + ox : -1;
    }
    epilogue_dispatch<T, MT, NT, true>(a, acc, mrow, n0 + wn * NT * 16, q);
}

template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE, int LGCPP>
static bool launch_patch(const ConvArgs& a, hipStream_t s) {
    constexpr int CH = 16 / (int)sizeof(T), NTHR = 64 * WM * WN, RP = NTHR / 4;
    constexpr int BN = WN * NT * 16, BNP = (BN + RP - 1) / RP * RP;
    constexpr int CPP = 1 << LGCPP, PAL = CPP >= 8 ? CPP : 8, PWP = (TW + 2 + PAL - 1) / PAL * PAL;
    constexpr int TOTAL = (TH + 2) * PWP * CPP;
    constexpr size_t lds = (size_t)(TOTAL + NTHR - 1) / NTHR * NTHR * 16 + (size_t)NSTAGE * BNP * 64;
    static_assert(lds <= 160 * 1024, "patch does not fit the LDS");
    if (a.Cin != CPP * CH) return false;
    const int tiles_x = ceil_div(a.Wo, TW), tiles_y = ceil_div(a.Ho, TH);
    const int n_img = a.M / (a.Ho * a.Wo);
    auto kfn = conv3x3_patch_kernel<T, MT, NT, WM, WN, TH, TW, NSTAGE, LGCPP>;
    static bool attr = false;
    if (lds > 64 * 1024 && !attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    dim3 grid(n_img * tiles_x * tiles_y, ceil_div(a.Cout, BN));
    hipLaunchKernelGGL(kfn, grid, dim3(NTHR), lds, s, a, tiles_x, tiles_y);
    KCHECK();
    return true;
}

// 3x3/s1/p1 with Cin a multiple of the K-step: tile shape by output width.
template <typename T>
static bool try_patch(const ConvArgs& a, hipStream_t s) {
    // Measured on MI355X (profiles/): the patch form wins where Cout is small and M is large (ReID layer1);
    // for Cout >= 128 the 8-wave im2col tile is faster, and small maps are launch-bound either way.
    static const bool off = getenv("AICAM_NO_PATCH") != nullptr;
    static const bool all = getenv("AICAM_PATCH_ALL") != nullptr;
    static const bool c32 = getenv("AICAM_NO_PATCH_C32") == nullptr;   // Cin = Cout = 32 (YOLOv8n P3 bottlenecks): 244 -> 460 TFLOP/s
    if (off || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Wo < 16 || a.Ho < 8) return false;
    if (a.M < 200000 && !all) return false;
    const bool wide = a.Wo % 32 == 0 || (a.Wo % 16 != 0 && a.Wo >= 32);   // 8 x 32 tiles unless 16 x 16 tiles cover the map exactly
    if (a.Cout == 64) {
        constexpr int LG64 = sizeof(T) == 2 ? 3 : 4;    // Cin = 64: 8 chunks (fp16) / 16 chunks (fp32) per pixel
        if (wide) return launch_patch<T, 4, 4, 4, 1, 8, 32, 3, LG64>(a, s);
        return launch_patch<T, 4, 4, 4, 1, 16, 16, 3, LG64>(a, s);
    }
    if (a.Cout == 32 && c32) {
        constexpr int LG32 = sizeof(T) == 2 ? 2 : 3;    // Cin = 32
        if (wide) return launch_patch<T, 4, 2, 4, 1, 8, 32, 3, LG32>(a, s);
        return launch_patch<T, 4, 2, 4, 1, 16, 16, 3, LG32>(a, s);
    }
    return false;
}

template <typename T, int MT, int NT, int WM, int WN, int NSTAGE>
static void launch_dma(const ConvArgs& a, hipStream_t s) {
    constexpr int RP = 16 * WM * WN;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + RP - 1) / RP * RP;
    dim3 grid(ceil_div(a.M, BM), ceil_div(a.Cout, BN));
    const size_t lds = (size_t)NSTAGE * (BM + BNP) * 64;
    auto kfn = conv_igemm_dma_kernel<T, MT, NT, WM, WN, NSTAGE>;
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(64 * WM * WN), lds, s, a);
    KCHECK();
}

template <typename T, int MT, int NT, int WM, int WN>
static void launch_variant(const ConvArgs& a, hipStream_t s) {
    if (conv_impl() == 2) {
        launch_dma<T, MT, NT, WM, WN, 4>(a, s);
        return;
    }
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    dim3 grid(ceil_div(a.M, BM), ceil_div(a.Cout, BN));
    const size_t lds = 2 * (size_t)(BM + BN) * 64;
    hipLaunchKernelGGL((conv_igemm_kernel<T, MT, NT, WM, WN>), grid, dim3(256), lds, s, a);
    KCHECK();
}

template <typename T>
static void launch_conv_t(const ConvArgs& a, hipStream_t s) {
    const int c = a.Cout;
    if (conv_impl() == 2 && try_pp_patch<T>(a, s)) return;
    if (conv_impl() == 2 && try_patch<T>(a, s)) return;
    const long blocks128 = (long)ceil_div(a.M, 128);
    if (c % 128 == 0 || c > 160) {
        static const bool t256 = getenv("AICAM_NO_T256") == nullptr;   // +12% on ReID layer3/4 over 256x128 (profiles/)
        // Ping-pong kernels (one block per CU) where the K loop is long enough to amortise the tile's prologue/epilogue:
        // measured on MI355X (tools/conv_bench.py, profiles/): +17..19% on ReID layer3/4, +14% on layer2, a loss at K < 512.
        static const bool pp = getenv("AICAM_NO_PP") == nullptr;
        static const int pp_min = [] { const char* e = getenv("AICAM_PP_MIN"); return e ? atoi(e) : 200; }();
        constexpr int BKE_ = 64 / (int)sizeof(T);
        if (pp && conv_impl() == 2 && a.Cin % BKE_ == 0 && (a.Kp >= 16 * BKE_ || pp_min == 0)) {
            if (c % 256 == 0 && (long)ceil_div(a.M, 256) * (c / 256) >= pp_min) { launch_pp<T, 8, 4, 2, 4, 4>(a, s); return; }   // 256 px x 256 ch
            static const int pp128_k = [] { const char* e = getenv("AICAM_PP128_K"); return e ? atoi(e) : 32; }();
            if (c == 128 && (a.Kp >= pp128_k * BKE_ || pp_min == 0) && ceil_div(a.M, 512) >= pp_min) { launch_pp<T, 8, 4, 4, 2, 4>(a, s); return; }                      // 512 px x 128 ch
        }
        if (t256 && conv_impl() == 2 && c % 256 == 0 && (long)ceil_div(a.M, 256) * (c / 256) >= 200) launch_dma<T, 8, 4, 2, 4, 4>(a, s);   // 8 waves: 256 px x 256 ch
        else if (conv_impl() == 2 && (blocks128 / 2) * ceil_div(c, 128) >= 384) launch_dma<T, 4, 4, 4, 2, 3>(a, s);   // 8 waves: 256 px x 128 ch
        else if (blocks128 * ceil_div(c, 128) >= 128) launch_variant<T, 4, 4, 2, 2>(a, s);   // 128 px x 128 ch
        else launch_variant<T, 2, 2, 2, 2>(a, s);                                       // 64 px x 64 ch (small maps)
    } else if (c % 80 == 0) {
        launch_variant<T, 2, 5, 4, 1>(a, s);                                            // 128 px x 80 ch
    } else if (c % 64 == 0) {
        if (blocks128 >= 512) launch_variant<T, 4, 4, 4, 1>(a, s);                      // 256 px x 64 ch
        else launch_variant<T, 2, 4, 4, 1>(a, s);                                       // 128 px x 64 ch
    } else if (c % 48 == 0) {
        launch_variant<T, 2, 3, 4, 1>(a, s);                                            // 128 px x 48 ch
    } else if (c % 32 == 0 || c > 16) {
        launch_variant<T, 4, 2, 4, 1>(a, s);                                            // 256 px x 32 ch
    } else {
        launch_variant<T, 4, 1, 4, 1>(a, s);                                            // 256 px x 16 ch
    }
}

void launch_conv_igemm(int dtype, const ConvArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    if (dtype == AIC_F16 && try_c16(a, s)) return;
    if (dtype == AIC_F16 && try_c64_resident(a, s)) return;
    if (dtype == AIC_F16) launch_conv_t<half_t>(a, s);
    else launch_conv_t<float>(a, s);
}

// ------------------------------------------------------------------------------------------------
// Fused ReID stem: conv 3x3/1 (3 -> 64) + bias + ReLU + max-pool 3x3/2 (pad 1) in one kernel, fp16.
// The unfused pair writes and re-reads a [N,128,64,64] tensor (1 MB per crop) for 14 MMAC of work;
// here a block owns 4 pooled rows of one crop: the 11x66 input patch (RGB0) and the 9x64x64 conv
// tile live in LDS only, K = 27 is padded to one v_mfma_f32_16x16x32_f16 per 16 px x 16 ch tile
// (the im2col fragment is gathered from the patch), and only the pooled [N,64,32,64] tensor
// reaches HBM.  PyTorch semantics: conv zero-pads its input, the pool ignores out-of-image taps.
struct StemArgs {
    const void* x; const void* w; const float* bias; void* y;
    int n, H, W, Kp, y_cs, y_coff;   // input [n][H][W][8]; output [n][H/2][W/2][y_cs]
};

__global__ __launch_bounds__(256) void reid_stem_pool_kernel(const StemArgs a) {
    constexpr int PT = 4, CR = 2 * PT + 1, IR = 2 * PT + 3, CW = 64, PW = CW + 2, CO = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2* patch = reinterpret_cast<uint2*>(smem);                        // [IR][PW] pixels x 4 halves
    char* convbuf = smem + ((IR * PW * 8 + 15) / 16) * 16;               // [CR][CW] pixels x 128 B (swizzled chunks)
    const half_t* patch_h = reinterpret_cast<const half_t*>(smem);

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 15, q = lane >> 4;
    const int Hp = a.H / 2, Wp = a.W / 2;
    const int groups = Hp / PT;
    const int img = blockIdx.x / groups, rg = blockIdx.x - img * groups;
    const int oy0 = rg * PT, cr0 = 2 * oy0 - 1, ir0 = cr0 - 1;
    const half_t* xg = reinterpret_cast<const half_t*>(a.x) + (size_t)img * a.H * a.W * 8;

    for (int idx = t; idx < IR * PW; idx += 256) {
        const int iy = idx / PW, ix = idx - iy * PW;
        const int gy = ir0 + iy, gx = ix - 1;
        uint2 v = make_uint2(0u, 0u);
        if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
            v = *reinterpret_cast<const uint2*>(xg + ((size_t)gy * a.W + gx) * 8);
        patch[idx] = v;
    }
    // weight fragments (A operand): lane (r, q) of channel tile ct holds w[16ct + r][k = 8q .. 8q+7]
    const half_t* wg = reinterpret_cast<const half_t*>(a.w);
    half8 wf[4];
    int poff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        const int tap = k / 3, ci = k - 3 * tap, kh = tap / 3, kw = tap - 3 * kh;
        poff[j] = k < 27 ? (kh * PW + kw) * 4 + ci : -1;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            wf[ct][j] = k < 27 ? wg[(size_t)(16 * ct + r) * a.Kp + tap * 8 + ci] : (half_t)0.f;
    }
    floatx4 b4[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) b4[ct] = *reinterpret_cast<const floatx4*>(a.bias + 16 * ct + 4 * q);
    __syncthreads();

    for (int tile = wv; tile < CR * (CW / 16); tile += 4) {
        const int cr = tile / (CW / 16), cx = (tile - cr * (CW / 16)) * 16 + r;
        const int base = (cr * PW + cx) * 4;
        half8 xf;
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = poff[j] >= 0 ? patch_h[base + poff[j]] : (half_t)0.f;
        char* dst = convbuf + (size_t)(cr * CW + cx) * 128 + (q & 1) * 8;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct], xf, acc, 0, 0, 0);
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)fmaxf(acc[e] + b4[ct][e], 0.f);
            const int chunk = 2 * ct + (q >> 1);                         // 16-byte chunk of the pixel's 64 channels
            *reinterpret_cast<half4*>(dst + ((chunk ^ (cx & 7)) * 16)) = h;
        }
    }
    __syncthreads();

    half_t* yg = reinterpret_cast<half_t*>(a.y);
    for (int o = t; o < PT * Wp * (CO / 8); o += 256) {
        const int g = o & 7, px = (o >> 3) % Wp, py = (o >> 3) / Wp;
        half8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (half_t)0.f;                  // post-ReLU values are >= 0
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int cr = 2 * py + dy;
            if ((unsigned)(cr0 + cr) >= (unsigned)a.H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int cc = 2 * px - 1 + dx;
                if ((unsigned)cc >= (unsigned)a.W) continue;
                const half8 v = *reinterpret_cast<const half8*>(convbuf + (size_t)(cr * CW + cc) * 128 + ((g ^ (cc & 7)) * 16));
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *reinterpret_cast<half8*>(yg + ((size_t)(img * Hp + oy0 + py) * Wp + px) * a.y_cs + a.y_coff + g * 8) = m;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused ReID stem, second form (default): one block = one crop, 8 waves, wave w owns pooled rows [Hp/8*w, +Hp/8).
//  * K is laid out (tap, RGB0): taps 0..7 = one v_mfma_f32_16x16x32_f16 whose B fragment is two aligned 8-byte
//    patch pixels per lane, tap 8 = a second one with zero weights outside (q = 0, j < 3): the im2col fragment is
//    3 ds_read_b64, no scalar gathers (the first form spent 71 VALU per MFMA on them).  [v_mfma_f32_16x16x16_f16
//    for tap 8 returned stale accumulator halves under hipcc 7.2: the first two results were read too early];
//  * the bias rides in as the accumulator's initial value, ReLU is a packed fp16 max after the conversion;
//  * the 3x3/2 max-pool never touches LDS: vertical max of three conv rows in registers (v_pk_max_f16),
//    horizontal max over lane neighbours by DPP row shifts inside the 16-pixel tile (lane 0 takes pixel 15 of
//    the tile to its left by row_ror), out-of-image taps are 0 = the identity of max over post-ReLU values;
//  * with the channel permutation of perm_ch() a lane owns 8 consecutive channels per tile pair; odd lanes take
//    the second pair of their even neighbour, so one 16-byte store instruction writes 8 pooled pixels x 128 B.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));
// max of packed fp16 pairs as SIGNED 16-bit integers (v_pk_max_i16): exact for the values met here -- non-negative
// halves order like their bit patterns, and against 0 it is ReLU (any negative half, -0 included, has the sign bit
// set and loses to 0).  The fp16 form would add a canonicalising v_pk_max_f16 v,v,v per operand.
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
    const short2_t m = __builtin_elementwise_max(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b));
    return __builtin_bit_cast(unsigned, m);
}
template <int CTRL, bool BOUND> __device__ __forceinline__ unsigned dpp(unsigned old, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, BOUND);
}
struct Row8 { unsigned u[8]; };   // one pixel's 16 output channels x 2 tile pairs, packed fp16: u[4p + i]

__global__ __launch_bounds__(512) void reid_stem_pool2_kernel(const StemArgs a) {
    constexpr int CW = 64, PW = CW + 2, NTX = CW / 16;
    constexpr int ROW_SHL1 = 0x101, ROW_SHR1 = 0x111, ROW_ROR1 = 0x121;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2* patch = reinterpret_cast<uint2*>(smem);                        // [H + 2][PW] pixels x RGB0 halves, zero border

    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 15, q = lane >> 4;
    const int H = a.H, Hp = H / 2, Wp = CW / 2, rows_per_wave = Hp / 8;
    const int img = blockIdx.x;
    const half_t* xg = reinterpret_cast<const half_t*>(a.x) + (size_t)img * H * CW * 8;

    for (int idx = t; idx < (H + 2) * PW; idx += 512) {
        const int iy = idx / PW, ix = idx - iy * PW;
        const int gy = iy - 1, gx = ix - 1;
        uint2 v = make_uint2(0u, 0u);
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)CW) v = *reinterpret_cast<const uint2*>(xg + ((size_t)gy * CW + gx) * 8);
        patch[idx] = v;
    }

    // A operands: MFMA row rho of channel tile ct carries channel perm_row<4>(ct, rho); lane (rho = r, q) holds k = 8q..8q+7
    const half_t* wg = reinterpret_cast<const half_t*>(a.w);
    half8 wa[4], wb[4];
    floatx4 bi[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const half_t* wr = wg + (size_t)perm_row<4>(ct, r) * a.Kp;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 2 * q + (j >> 2), ci = j & 3;
            wa[ct][j] = ci < 3 ? wr[tap * 8 + ci] : (half_t)0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) wb[ct][j] = (q == 0 && j < 3) ? wr[8 * 8 + j] : (half_t)0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[ct][e] = a.bias[perm_ch<4>(ct, q, e)];
    }
    // patch offsets (in pixels) of this lane's taps relative to (conv row y, tile pixel): taps 2q, 2q+1 and tap 8
    const int t0 = 2 * q, t1 = 2 * q + 1;
    const int off0 = (t0 / 3) * PW + t0 % 3 + r, off1 = (t1 / 3) * PW + t1 % 3 + r, off2 = 2 * PW + 2 + r;
    __syncthreads();

    auto conv_tile = [&](int y, int tx) -> Row8 {   // conv + bias + ReLU of 16 pixels (row y, columns 16tx..) x 64 channels
        Row8 o;
        if ((unsigned)y >= (unsigned)H) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o.u[i] = 0u;
            return o;
        }
        const uint2* pp = patch + y * PW + 16 * tx;
        const uint2 x0 = pp[off0], x1 = pp[off1], x2 = pp[off2];
        const uint4 xa4 = make_uint4(x0.x, x0.y, x1.x, x1.y), xb4 = make_uint4(x2.x, x2.y, 0u, 0u);
        const half8 xa = __builtin_bit_cast(half8, xa4), xb = __builtin_bit_cast(half8, xb4);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            floatx4 acc = bi[ct];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ct], xa, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ct], xb, acc, 0, 0, 0);
            const half2_t h01 = {(half_t)acc[0], (half_t)acc[1]}, h23 = {(half_t)acc[2], (half_t)acc[3]};
            o.u[2 * ct] = pk_max(__builtin_bit_cast(unsigned, h01), 0u);
            o.u[2 * ct + 1] = pk_max(__builtin_bit_cast(unsigned, h23), 0u);
        }
        return o;
    };

    half_t* yg = reinterpret_cast<half_t*>(a.y);
    const int py0 = wv * rows_per_wave;
    Row8 prev[NTX];
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) prev[tx] = conv_tile(2 * py0 - 1, tx);
    for (int py = py0; py < py0 + rows_per_wave; ++py) {
        Row8 vleft;
#pragma unroll
        for (int i = 0; i < 8; ++i) vleft.u[i] = 0u;
#pragma unroll
        for (int tx = 0; tx < NTX; ++tx) {
            const Row8 b = conv_tile(2 * py, tx), c = conv_tile(2 * py + 1, tx);
            Row8 v, hm;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v.u[i] = pk_max(pk_max(prev[tx].u[i], b.u[i]), c.u[i]);
                prev[tx].u[i] = c.u[i];
                const unsigned rot = dpp<ROW_ROR1, false>(0u, vleft.u[i]);          // lane 0 <- pixel 15 of the tile to the left (0 at tx = 0)
                const unsigned lf = dpp<ROW_SHR1, false>(rot, v.u[i]);               // lane r <- pixel r-1 (lane 0 keeps rot)
                const unsigned rt = dpp<ROW_SHL1, true>(0u, v.u[i]);                 // lane r <- pixel r+1 (only even r are used)
                hm.u[i] = pk_max(pk_max(lf, v.u[i]), rt);
            }
            vleft = v;
            // even lane 2u: pooled pixel 8tx+u, channels of pair 0; odd lane 2u+1: same pixel, pair 1 (taken from lane 2u)
            uint4 out;
            unsigned* op = reinterpret_cast<unsigned*>(&out);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned nb = dpp<ROW_SHR1, true>(0u, hm.u[4 + i]);
                op[i] = (r & 1) ? nb : hm.u[i];
            }
            const size_t pix = ((size_t)img * Hp + py) * Wp + 8 * tx + (r >> 1);
            *reinterpret_cast<uint4*>(yg + pix * a.y_cs + a.y_coff + (r & 1) * 32 + 8 * q) = out;
        }
    }
}

void launch_reid_stem_pool(const void* x, const void* w, const float* bias, void* y, int n, int H, int W, int Kp, int y_cs,
                           int y_coff, hipStream_t s) {
    if (n <= 0) return;
    StemArgs a{x, w, bias, y, n, H, W, Kp, y_cs, y_coff};
    static const bool v1 = [] { const char* e = getenv("AICAM_STEM"); return e && e[0] == 'v' && e[1] == '1'; }();
    const size_t lds2 = (size_t)(H + 2) * 66 * 8;
    if (!v1 && W == 64 && H % 16 == 0 && lds2 <= 160 * 1024) {
        static bool attr2 = false;
        if (!attr2) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(reid_stem_pool2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr2 = true;
        }
        hipLaunchKernelGGL(reid_stem_pool2_kernel, dim3(n), dim3(512), lds2, s, a);
        KCHECK();
        return;
    }
    const size_t lds = ((11 * 66 * 8 + 15) / 16) * 16 + (size_t)9 * 64 * 128;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(reid_stem_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(reid_stem_pool_kernel, dim3(n * (H / 2 / 4)), dim3(256), lds, s, a);
    KCHECK();
}

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
    if (idx >= (long)a.n * a.c) return;
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
    if (item >= a.n) return;
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

void launch_sppf_pool(int dtype, const EltArgs& a, hipStream_t s) { ELT_LAUNCH(sppf_pool_kernel, (long)a.n * a.h * a.w * (a.c / vecn(dtype))); }
void launch_upsample2x(int dtype, const EltArgs& a, hipStream_t s) { ELT_LAUNCH(upsample2x_kernel, (long)a.n * 4 * a.h * a.w * (a.c / vecn(dtype))); }
void launch_maxpool3s2(int dtype, const EltArgs& a, hipStream_t s) {
    const int oh = (a.h - 1) / 2 + 1, ow = (a.w - 1) / 2 + 1;
    ELT_LAUNCH(maxpool3s2_kernel, (long)a.n * oh * ow * (a.c / vecn(dtype)));
}
void launch_avgpool(int dtype, const EltArgs& a, hipStream_t s) { ELT_LAUNCH(avgpool_kernel, (long)a.n * a.c); }
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
