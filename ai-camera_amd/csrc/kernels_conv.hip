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
#include "conv_common.hpp"

namespace aic {

static int g_conv_cus = [] { const char* e = getenv("AICAM_CONV_CUS"); return e ? std::max(1, atoi(e)) : 256; }();
int conv_cu_budget() { return g_conv_cus; }
void set_conv_cu_budget(int cus) { if (getenv("AICAM_CONV_CUS") == nullptr) g_conv_cus = std::max(1, cus); }


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
    const int M = a.n_dev ? min(a.M, min(a.n_dev[0], a.M / (a.Ho * a.Wo)) * (a.Ho * a.Wo)) : a.M;
    if (m0 >= M) return;

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);

    // ---- per-thread description of the pixel rows it gathers
    size_t pix_base[A_PER];
    int ih0[A_PER], iw0[A_PER];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + r0 + 64 * i;
        if (m < M) {
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
        mma_tiles<T, MT, NT>(acc, wf, xf);
        if (more) store_step(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, residual, activation; 4 consecutive channels per lane
    const float* __restrict__ bias = a.bias;
    const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wm * MT + i) * 16 + r;
        if (m >= M) continue;
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
//  * TAIL: the conv's output feeds a 1x1 conv (a.w_tail) that runs in this kernel's epilogue (tail_1x1, conv_common.hpp)
template <typename T, int MT, int NT, int WM, int WN, int NSTAGE, bool TAIL = false>
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
    const int M = a.n_dev ? min(a.M, min(a.n_dev[0], a.M / (a.Ho * a.Wo)) * (a.Ho * a.Wo)) : a.M;     // device-side item count: the grid was sized for a bound
    int tbx, tby;
    if (!xcd_tile_xy_live(a.xcd_map, (M + BM - 1) / BM, tbx, tby)) return;
    const int m0 = tbx * BM;
    const int n0 = tby * BN;

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
        if (m < M) {
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
    if (a.bias_init) {                         // bias first (ConvArgs::bias_init): the epilogue's `bias` is a page of zeros
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            floatx4 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = a.bias_init[n0 + wn * NT * 16 + perm_ch<NT>(j, q, e)];
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = b;
        }
    }

    // LDS offsets of this lane's operand chunks inside a stage (stage base added as an immediate below)
    int xoff[MT], woff[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xoff[i] = lds_off((wm * MT + i) * 16 + r, q);
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = BM * 64 + lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);
    typedef typename Frag<T>::type frag_t;
    floatx4 mid[sizeof(T) == 4 ? MT : 1][sizeof(T) == 4 ? NT : 1];      // fp32 only: the mid-level accumulators ...
    doublex4 accd[sizeof(T) == 4 ? MT : 1][sizeof(T) == 4 ? NT : 1];    // ... and the top level (it starts from `acc`: zero, or the bias)
    int mid_steps = 0;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                mid[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) accd[i][j][e] = acc[i][j][e];
            }
    }
    auto compute = [&](int stage) {
        const char* base = smem + stage * STAGE;
        frag_t xf[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(base + xoff[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(base + woff[j]);
        if constexpr (sizeof(T) == 4) {         // fp32: three-level summation (conv_common.hpp)
            static_assert(sizeof(T) == 2 || MT * NT <= 8, "fp32 tiles carry a mid-level set beside the accumulators");
            mma_tiles_mid<MT, NT>(mid, wf, xf);
            if ((++mid_steps & (MID_STEPS - 1)) == 0) flush_mid<MT, NT>(accd, mid);
        } else {
            mma_tiles<T, MT, NT>(acc, wf, xf);
        }
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
        int tap = 0, kh = 0, kw = 0, cc = 0, ti3 = 0;
        const int kord = a.k_order;                 // != 0: K-steps in another order than memory's, source pointers rebuilt every step
        const bool cmaj = kord != 0;
        // split source (ConvArgs::xs; 1x1 / 1 / 0, memory order only): K-steps 0 .. ssteps - 1 read the half-resolution tensor
        const int ssteps = (a.xs && !cmaj) ? a.Cs / BKE : 0;
        auto xs_row = [&](int i) -> const T* {
            const int m = m0 + r0 + RP * i;
            if (m >= M) return zero;
            int img, rem, oh, ow;
            fast_divmod(m, HoWo, inv_howo, img, rem);
            fast_divmod(rem, a.Wo, inv_wo, oh, ow);
            return reinterpret_cast<const T*>(a.xs) + (((long)img * a.Hs + (oh >> 1)) * a.Ws + (ow >> 1)) * a.xs_cs + a.xs_coff + kc * CH;
        };
        // second source (ConvArgs::x2, chunk-major walks only): its channel chunk e is accumulated right after tap (0, 0) of the window's chunk
        // e + 1 -- the place conv3x3_pp_patch_kernel has for it; every kernel walks the same order.  xs: the step being set up is that chunk, e = cc - 1
        const int csteps2 = (a.x2 && kord == 1) ? a.Cin2 / BKE : 0;
        bool xs = false;
        auto set_tap = [&] {
            if (xs) {
                const int c2 = cc - 1;
#pragma unroll
                for (int i = 0; i < A_PER; ++i) {
                    const int m = m0 + r0 + RP * i;
                    const T* p = zero;
                    if (m < M) {
                        int img, rem, oh, ow;
                        fast_divmod(m, HoWo, inv_howo, img, rem);
                        fast_divmod(rem, a.Wo, inv_wo, oh, ow);
                        p = reinterpret_cast<const T*>(a.x2) + (((long)img * a.H2 + oh * a.s2) * a.W2 + ow * a.s2) * a.x2_cs + a.x2_coff + kc * CH + c2 * BKE;
                    }
                    aptr[i] = p, ainc[i] = 0;
                }
                return;
            }
            const long toff = ((long)kh * a.W + kw) * a.x_cs + kc * CH + (cmaj ? cc * BKE : 0);
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const bool ok = tap < ntap && cc < csteps && ((vmask[i] >> (tap & 31)) & 1u);
                aptr[i] = ok ? rowp[i] + toff : zero;
                ainc[i] = (ok && !cmaj) ? BKE : 0;
            }
        };
        set_tap();
        if (ssteps) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) { aptr[i] = xs_row(i); ainc[i] = aptr[i] == zero ? 0 : BKE; }
        }
        auto issue_fast = [&](int stage) {
            char* sbase = sdst + stage * STAGE;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                __builtin_amdgcn_global_load_lds((gptr_t)aptr[i], (lptr_t)(sbase + i * (RP * 64)), 16, 0, 0);
                aptr[i] += ainc[i];
            }
            if (cmaj) {
                const int koff = xs ? ntap * a.Cin + (cc - 1) * BKE : tap * a.Cin + cc * BKE;
#pragma unroll
                for (int j = 0; j < B_PER; ++j) {
                    const T* src = (winc[j] && cc < csteps) ? wptr[j] + koff : zero;
                    asm volatile("" : "+v"(src));
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + BM * 64 + j * (RP * 64)), 16, 0, 0);
                }
                if (kord == 1) {                        // (cc, kh, kw), the second source's chunk cc - 1 behind tap (0, 0) of chunks 1 .. csteps2
                    if (!xs && kh == 0 && kw == 0 && cc >= 1 && cc <= csteps2 && cc < csteps) xs = true;
                    else {
                        xs = false;
                        if (++kw == a.KW) { kw = 0; if (++kh == a.KH) { kh = 0; ++cc; } }
                    }
                } else if (kord == 3) {                 // (cc, then the nine taps plane by plane: 0 2 6 8 | 1 7 | 3 5 | 4 -- conv3x3s2_sp_patch_kernel's order)
                    if (++ti3 == 9) { ti3 = 0; ++cc; }
                    const int tp = (int)((0x453718620ull >> (4 * ti3)) & 15);
                    kh = tp / 3, kw = tp - 3 * kh;
                } else {                                // (kw, cc, kh)
                    if (++kh == a.KH) { kh = 0; if (++cc == csteps) { cc = 0; ++kw; } }
                    if (kw == a.KW) { kw = 0; cc = csteps; }          // past the last step: zero page from here on
                }
                tap = kh * a.KW + kw;
                set_tap();
                return;
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
            } else if (cc == ssteps && ssteps) {       // the split source's channels are through: on in the concat buffer, at channel Cs
#pragma unroll
                for (int i = 0; i < A_PER; ++i) {
                    const bool ok = (vmask[i] & 1u) != 0;
                    aptr[i] = ok ? rowp[i] + kc * CH + cc * BKE : zero;
                    ainc[i] = ok ? BKE : 0;
                }
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
    if constexpr (sizeof(T) == 4) {                                // the last (partial) block of mid-level steps
        flush_mid<MT, NT>(accd, mid);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = (float)accd[i][j][e];
    }

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wm * MT + i) * 16 + r;
        mrow[i] = m < M ? m : -1;
    }
    if constexpr (TAIL) {
        static_assert(sizeof(T) == 2 && WN == 1, "the tail needs fp16 and a wave that owns every channel of its pixels");
        tail_1x1<MT, NT>(a, acc, mrow, lane);
    } else {
        epilogue_dispatch<T, MT, NT, true, WM * WN == 8>(a, acc, mrow, n0 + wn * NT * 16, q);
    }
}

template <typename T, int MT, int NT, int WM, int WN, int NSTAGE, bool TAIL = false>
static void launch_dma(const ConvArgs& a, hipStream_t s) {
    constexpr int RP = 16 * WM * WN;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + RP - 1) / RP * RP;
    dim3 grid(ceil_div(a.M, BM), ceil_div(a.Cout, BN));
    const size_t lds = (size_t)NSTAGE * (BM + BNP) * 64;
    auto kfn = conv_igemm_dma_kernel<T, MT, NT, WM, WN, NSTAGE, TAIL>;
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
    // (A lean K-step for these small-tile launches -- kernels_conv_sp.hip's: per-lane row pointers + one scalar offset per step, next step's
    //  fragments read under this step's MFMAs, immediates for ring stage and fragment set; bit-identical to this kernel, 31 net tests green --
    //  was built and measured in round 5: the per-frame plugin loop's conv time 1 070 -> 1 030 us per frame (4 or 8 ring stages alike), but
    //  16- / 64-frame launch groups 7 100 -> 6 320 and 10 300 -> 9 900 frames/s and the headline -0.4 %.  Removed.  Found on the way: with
    //  inline-asm MFMAs and early loop exits the allocator copies accumulators between bodies -- VALU reads of matrix-pipe results the
    //  compiler does not know are such -- and the ReID layers came out non-deterministic; in-place asm MFMAs are safe only where the
    //  accumulators provably stay put, as in kernels_conv_sp.hip.)
    if constexpr (sizeof(T) == 2 && WM * WN == 4) {
        // launches of a few tiles (the per-frame plugin loop): one synchronisation per group of K-steps, bit-identical (kernels_conv_wide.hip)
        if (conv_impl() == 2 && conv_try_wide<MT, NT, WM, WN>(a, s)) return;
    }
    if (conv_impl() == 2) {
        // ring depth by K: a layer whose whole K is 2 .. 6 steps (YOLOv8n's 1x1 convs: K = 64 .. 192) gains nothing from a 4-deep ring, and
        // the LDS it costs halves the blocks a CU holds (80 KB per 256 x 64 tile: 2 blocks; 2 stages: 40 KB, 4 blocks).  AICAM_DMA_NSTAGE=n
        // forces a depth (A/B), AICAM_DMA_NS_K=k sets the largest K-step count that takes the shallow ring
        static const int force = [] { const char* e = getenv("AICAM_DMA_NSTAGE"); return e ? atoi(e) : 0; }();
        static const int ns_k = [] { const char* e = getenv("AICAM_DMA_NS_K"); return e ? atoi(e) : 0; }();
        const int nsteps = a.Kp / (sizeof(T) == 2 ? 32 : 16);
        // (A deeper ring for the small launches -- 8 stages where a 64 x 64 tile's K loop runs 0.41 us per step -- was measured in round 5 on the
        //  per-frame plugin loop: 1 101 against 1 075 us of conv time per frame.  Their K loop is not waiting for memory: it is ~90 instructions
        //  per step for four MFMAs -- the chunk-major walks rebuild every row pointer every step.)
        const int ns = force ? force : (nsteps <= ns_k ? 2 : 4);
        if (ns == 2) launch_dma<T, MT, NT, WM, WN, 2>(a, s);
        else if (ns == 3) launch_dma<T, MT, NT, WM, WN, 3>(a, s);
        else launch_dma<T, MT, NT, WM, WN, 4>(a, s);
        return;
    }
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    dim3 grid(ceil_div(a.M, BM), ceil_div(a.Cout, BN));
    const size_t lds = 2 * (size_t)(BM + BN) * 64;
    hipLaunchKernelGGL((conv_igemm_kernel<T, MT, NT, WM, WN>), grid, dim3(256), lds, s, a);
    KCHECK();
}

template <typename T>
static void launch_conv_t(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    if (a.k_order == 2) {                      // the order AND the bias placement of the weights-resident kernels (Cout = 64: one 256-byte zero page covers the epilogue's reads)
        a.bias_init = a.bias;
        a.bias = reinterpret_cast<const float*>(a.zero);
    }
    const int c = a.Cout;
    constexpr int DT = sizeof(T) == 2 ? AIC_F16 : AIC_F32;
    const long blocks128 = (long)ceil_div(a.M, 128);
    if constexpr (sizeof(T) == 4) {
        // fp32 engines (the parity mode) run ONE kernel family since round 5: the LDS-DMA implicit GEMM on tiles of at most 8 MFMA tiles per wave,
        // whose three-level summation carries a mid-level accumulator set (conv_common.hpp).  The patch,
        // ping-pong and 8-wave forms are fp16 only.  (--dtype fp32 throughput: 1 167 frames/s with the two-level kernels of rounds 2-4.)
        if (c % 128 == 0 || c > 160) launch_variant<T, 2, 4, 2, 2>(a, s);               // 64 px x 128 ch
        else if (c % 80 == 0) launch_variant<T, 1, 5, 4, 1>(a, s);                      // 64 px x 80 ch
        else if (c % 64 == 0) launch_variant<T, 2, 4, 4, 1>(a, s);                      // 128 px x 64 ch
        else if (c % 48 == 0) launch_variant<T, 2, 3, 4, 1>(a, s);                      // 128 px x 48 ch
        else if (c % 32 == 0 || c > 16) launch_variant<T, 4, 2, 4, 1>(a, s);            // 256 px x 32 ch
        else launch_variant<T, 4, 1, 4, 1>(a, s);                                       // 256 px x 16 ch
        return;
    } else {
    if (conv_impl() == 2 && conv_try_pp_patch(DT, a, s)) return;
    if (conv_impl() == 2 && !a.x2 && DT == AIC_F16 && conv_try_pm_patch(a, s)) return;
    if (conv_impl() == 2 && !a.x2 && conv_try_patch(DT, a, s)) return;        // (a second source: the ping-pong patch kernel above or the LDS-DMA implicit GEMMs below)
    if (c % 128 == 0 || c > 160) {
        static const bool t256 = getenv("AICAM_NO_T256") == nullptr;   // +12% on ReID layer3/4 over 256x128 (profiles/)
        if (conv_impl() == 2 && conv_try_pp(DT, a, s)) return;   // one-block-per-CU ping-pong kernels (kernels_conv_pp.hip)
        if (t256 && conv_impl() == 2 && c % 256 == 0 && (long)ceil_div(a.M, 256) * (c / 256) >= 200) launch_dma<T, 8, 4, 2, 4, 4>(a, s);   // 8 waves: 256 px x 256 ch
        else if (conv_impl() == 2 && (blocks128 / 2) * ceil_div(c, 128) >= 384) launch_dma<T, 4, 4, 4, 2, 3>(a, s);   // 8 waves: 256 px x 128 ch
        else if (blocks128 * ceil_div(c, 128) >= 128) launch_variant<T, 4, 4, 2, 2>(a, s);   // 128 px x 128 ch
        else if ((long)ceil_div(a.M, 64) * ceil_div(c, 64) > 256 && conv_impl() == 2 && conv_try_wide<4, 4, 2, 2>(a, s)) return;   // too many 64 x 64 tiles for the wide-step kernel, few enough 128 x 128 ones
        else launch_variant<T, 2, 2, 2, 2>(a, s);                                       // 64 px x 64 ch (small maps)
    } else if (c == 144 && conv_impl() == 2) {
        // the merged first convs of a YOLOv8 detect level (64 box + 80 class channels, Model::Model; fp16 only): one 144-wide tile,
        // the map is read once.  4 waves, one per SIMD: 36 accumulator tiles per wave on the 256-pixel tile need the whole register file
        if (ceil_div(a.M, 256) >= 512) launch_dma<T, 4, 9, 4, 1, 4>(a, s);          // 256 px x 144 ch
        else if (!conv_try_wide<2, 9, 4, 1>(a, s)) launch_dma<T, 2, 9, 4, 1, 4>(a, s);   // 128 px x 144 ch (a few tiles: kernels_conv_wide.hip)
    } else if (c % 80 == 0) {
        // YOLOv8's class branches (Cout = nc = 80).  512 px x 80 ch on 8 waves once there are tiles for every CU:
        // 428 -> 499 TFLOP/s on cls0.1 (80 -> 80, 3x3 at 80 x 80), +7..16 % on the others (tools/conv_bench.py); AICAM_C80=0: off
        static const bool big80 = [] { const char* e = getenv("AICAM_C80"); return !e || atoi(e) != 0; }();
        if (big80 && conv_impl() == 2 && ceil_div(a.M, 512) >= 256) launch_dma<T, 4, 5, 8, 1, 3>(a, s);
        else launch_variant<T, 2, 5, 4, 1>(a, s);                                       // 128 px x 80 ch
    } else if (c % 64 == 0) {
        if (blocks128 >= 512) launch_variant<T, 4, 4, 4, 1>(a, s);                      // 256 px x 64 ch
        else if (blocks128 * ceil_div(c, 64) > 256 && conv_impl() == 2 && conv_try_wide<4, 4, 4, 1>(a, s)) return;   // (as above: 256 px tiles where the 128 px grid is too large for the wide-step kernel)
        else launch_variant<T, 2, 4, 4, 1>(a, s);                                       // 128 px x 64 ch
    } else if (c % 48 == 0) {
        launch_variant<T, 2, 3, 4, 1>(a, s);                                            // 128 px x 48 ch
    } else if (c % 32 == 0 || c > 16) {
        launch_variant<T, 4, 2, 4, 1>(a, s);                                            // 256 px x 32 ch
    } else {
        launch_variant<T, 4, 1, 4, 1>(a, s);                                            // 256 px x 16 ch
    }
    }
}

// Lead conv of a (conv, 1x1) pair with the 1x1 in its epilogue.  Only the kernels whose waves own all channels of their pixels:
// the 3x3 patch kernel where it applies (Cout 64), else the LDS-DMA implicit GEMM with NT = Cout / 16 and WN = 1 -- the same
// choices the conv gets on its own.
bool conv_tail_supported(int dtype, const ConvArgs& lead, const ConvArgs& tail) {
    static const bool off = getenv("AICAM_NO_TAIL") != nullptr;
    if (off || dtype != AIC_F16 || conv_impl() != 2) return false;
    if ((lead.Cout != 64 && lead.Cout != 80) || lead.act != 1 || lead.res_mode != 0 || lead.out_f32) return false;
    if (lead.xs || lead.x2) return false;                          // split / second sources are walked by the plain kernels only
    if (tail.KH != 1 || tail.KW != 1 || tail.stride != 1 || tail.pad != 0 || tail.res_mode != 0) return false;
    if (tail.x != lead.y || tail.x_cs != lead.y_cs || tail.x_coff != lead.y_coff || tail.M != lead.M || tail.Cin != lead.Cout) return false;
    if (tail.Cout > lead.Cout || tail.Kp != 32 * ((lead.Cout + 31) / 32) || tail.cout_pad < lead.Cout) return false;
    if (lead.cout_pad < lead.Cout || (tail.y_cs | tail.y_coff) % 8) return false;
    return true;
}

// A split source is walked by the memory-order fast path of conv_igemm_dma_kernel only: a 1x1 / 1 / 0 conv without tail whose Cout
// keeps it away from the ping-pong kernels (K of these layers is short anyway) and from the direct kernels.
bool conv_xs_supported(int dtype, const ConvArgs& a, int cs) {
    static const bool off = getenv("AICAM_NO_XS") != nullptr;
    const int bke = dtype == AIC_F16 ? 32 : 16;
    if (off || conv_impl() != 2) return false;
    if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.w_tail || a.x2 || a.Cin % bke || cs <= 0 || cs % bke || cs >= a.Cin) return false;
    if (a.H % 2 || a.W % 2) return false;
    return a.Kp < 16 * bke;                                       // (conv_try_pp takes K >= 16 steps: it has no split-source walk)
}

// A second source rides on the chunk-major walk of conv_igemm_dma_kernel / conv_igemm_pp_kernel: the layer must be one that every
// batch size sends there in that order -- a ping-pong-patch SHAPE (k_order 1) whose Cout takes the 128-multiple branch of launch_conv_t.
bool conv_x2_supported(int dtype, const ConvArgs& a, int cin2) {
    static const bool off = getenv("AICAM_NO_X2") != nullptr;
    const int bke = dtype == AIC_F16 ? 32 : 16;
    if (off || conv_impl() != 2 || getenv("AICAM_K_TAP_MAJOR")) return false;
    const int shape = conv_pp_patch_shape(dtype, a);             // 2: 512 x 128 tile, 3 / 4: 256 x 256 on 16 x 8 / 8 x 4 maps (shape 1, Cout 64, has no such kernel)
    if (shape < 2 || a.Cout % 128 || a.w_tail || a.out_f32 || cin2 <= 0 || cin2 % bke) return false;
    return cin2 / bke < a.Cin / bke;                             // its chunk e rides behind the window's chunk e + 1
}

static void launch_conv_tail(const ConvArgs& a, hipStream_t s) {
    if (a.Cout == 64) {
        if (conv_try_c32s2_tail(a, s)) return;
        if (conv_try_pm_patch_tail(a, s)) return;
        if (conv_try_patch_tail(a, s)) return;
        if ((long)ceil_div(a.M, 128) >= 512) launch_dma<half_t, 4, 4, 4, 1, 4, true>(a, s);   // 256 px x 64 ch
        else if (!conv_try_wide_tail<2, 4>(a, s)) launch_dma<half_t, 2, 4, 4, 1, 4, true>(a, s);   // 128 px x 64 ch (a few tiles: kernels_conv_wide.hip)
    } else {                                                                                  // 80
        if (conv_try_pm_patch_tail(a, s)) return;
        if (ceil_div(a.M, 512) >= 256) launch_dma<half_t, 4, 5, 8, 1, 3, true>(a, s);          // 512 px x 80 ch
        else if (!conv_try_wide_tail<2, 5>(a, s)) launch_dma<half_t, 2, 5, 4, 1, 4, true>(a, s);   // 128 px x 80 ch
    }
}

void launch_conv_igemm(int dtype, const ConvArgs& a0, hipStream_t s) {
    if (a0.M <= 0) return;
    ConvArgs a = a0;
    a.xcd_map = xcd_map_on();
    static const bool tap_major_everywhere = getenv("AICAM_K_TAP_MAJOR") != nullptr;   // A/B: the pre-round-3 behaviour (batch-dependent bits)
    a.k_order = 0;
    if (conv_impl() == 2 && !tap_major_everywhere) {
        // the 64-channel weights-resident kernels (fp16): any 3x3 / 1 / 1 layer with Cin = Cout = 64 whose map they tile
        // (ReLU, with or without the BasicBlock's residual: the only forms those kernels have)
        const bool c64 = dtype == AIC_F16 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 64 && a.Cout == 64 && a.Kp == 576 &&
                         a.act == 2 && a.res_mode <= 1 && !a.out_f32 && !a.w_tail &&
                         a.Ho == a.H && a.Wo == a.W && ((a.W % 32 == 0 && a.H % 8 == 0) || (a.W == 32 && a.H % 4 == 0));
        a.k_order = conv_pp_patch_shape(dtype, a) ? 1 : (c64 ? 2 : (conv_s2_patch_shape(a) ? 3 : 0));      // 3: the stride-2 patch kernel's order (kernels_conv_sp.hip)
    }
    if (a.x2) AIC_REQUIRE(a.k_order == 1 && a.Cout % 128 == 0 && !a.w_tail, AIC_ERR_INVALID, "conv with a second source: unsupported shape (check conv_x2_supported)");
    if (a.xs) AIC_REQUIRE(a.k_order == 0 && a.KH == 1 && a.KW == 1 && !a.w_tail && a.Kp < 16 * (dtype == AIC_F16 ? 32 : 16), AIC_ERR_INVALID,
                          "conv with a split source: unsupported shape (check conv_xs_supported)");
    if (a.w_tail) {
        AIC_REQUIRE(dtype == AIC_F16 && (a.Cout == 64 || a.Cout == 80) && a.act == 1 && a.res_mode == 0, AIC_ERR_INVALID,
                    "conv with a 1x1 tail: unsupported lead (check conv_tail_supported before setting w_tail)");
        launch_conv_tail(a, s);
        return;
    }
    if (dtype == AIC_F16 && conv_try_c16(a, s)) return;
    if (dtype == AIC_F16 && conv_try_1x1_stream(a, s)) return;
    if (dtype == AIC_F16 && conv_try_c64_resident(a, s)) return;
    if (dtype == AIC_F16) launch_conv_t<half_t>(a, s);
    else launch_conv_t<float>(a, s);
}


}  // namespace aic
