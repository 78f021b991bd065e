// kernels_conv_direct.hip -- conv kernels that keep the input PATCH in LDS instead of gathering im2col tiles: the 4-wave
// patch kernel (v3), the 16-input-channel direct kernel, the persistent weights-resident kernel, the fused ReID stem.
#include "conv_common.hpp"
#include "pre_math.hpp"

namespace aic {

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
    int bx = xcd_tile((int)blockIdx.x, (int)gridDim.x, a.xcd_map);
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const half_t* xg = reinterpret_cast<const half_t*>(a.x) + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;

    // all of a thread's patch loads in flight before the first store (rolled, the loop was nine dependent HBM round trips per block)
    constexpr int NSLOT = PR * PC * 2, NIT = (NSLOT + 255) / 256;
    {
        uint4 v[NIT];
        int dst[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = t + u * 256;
            const int h = idx & 1, pp = idx >> 1;
            const int pr = pp / PC, pc = pp - pr * PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (idx < NSLOT && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v[u] = *reinterpret_cast<const uint4*>(xg + ((size_t)iy * a.W + ix) * a.x_cs + h * 8);
            const int par = pc % NPAR, c2 = pc / NPAR;
            dst[u] = idx < NSLOT ? (((pr * NPAR + par) * 2 + h) * PCP + c2) * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u)
            if (dst[u] >= 0) *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
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
    // The wave's four output vectors are formed first and stored afterwards, each from registers of its own (and the residual
    // vectors are requested up front): stored tile by tile, one set of data registers is recycled and every store is fenced with
    // s_waitcnt vmcnt(0) -- four exposed store round trips per wave (conv_common.hpp, epilogue_wide_phased)
    typedef typename std::conditional<NCT == 1, half4, half8>::type ovec_t;
    constexpr int NE = NCT == 1 ? 4 : 8;
    const int ch = NCT == 1 ? 4 * q : 8 * q;                            // perm_ch<2>: tiles 0, 1 -> channels 8q + 4*(ct) + e
    ovec_t o[4];
    size_t pix[4];
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) pix[tile] = ((size_t)img * a.Ho + oy0 + 2 * wv + (tile >> 1)) * a.Wo + ox0 + (tile & 1) * 16 + r;
    const bool res = a.res_mode == 2;
    if (res) {
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) o[tile] = *reinterpret_cast<const ovec_t*>(rg + pix[tile] * a.r_cs + a.r_coff + ch);
    }
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int oyl = 2 * wv + (tile >> 1), oxl = (tile & 1) * 16 + r;
        const int base = (oyl * S * NPAR * 2 * PCP + oxl) * 16;
        half8 xb[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) xb[m] = *reinterpret_cast<const half8*>(smem + base + d[m]);
        float v[NE];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            floatx4 acc = bi[ct];
#pragma unroll
            for (int m = 0; m < 5; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ct][m], xb[m], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * ct + e] = act_fast<1>(acc[e]);
        }
        if (res) {
#pragma unroll
            for (int e = 0; e < NE; ++e) v[e] += (float)o[tile][e];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) o[tile][e] = (half_t)v[e];
    }
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) asm volatile("" : "+v"(o[tile]));
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) *reinterpret_cast<ovec_t*>(yg + pix[tile] * a.y_cs + a.y_coff + ch) = o[tile];
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
// Direct 3x3 / stride 2 / pad 1 for 32 -> 64 channels, fp16, SiLU, WITH the 1x1 conv behind it in its epilogue (ConvArgs::w_tail):
// YOLOv8n's `3.conv` + `4.c2f.cv1` at large batch (160 x 160 -> 80 x 80, round 5).  Through the LDS-DMA implicit GEMM the pair took
// 845 us per 512 frames at 7.8 % MFMA busy -- four times its HBM floor (839 MB in, 419 MB out): a stride-2 im2col K-step gathers 64-byte
// half lines and every input pixel comes through the L2 2.25 times.  Here a block owns 16 x 16 output pixels of one image:
//  * the 33 x 33 x 32-channel input patch is read ONCE into LDS as [row][column parity][8-channel chunk][column / 2] x 16 bytes (pitch 18
//    columns): the 16 lanes that share a chunk of a fragment read -- 16 output pixels of one tap -- take 256 consecutive bytes, and the
//    16 lanes of a store (four pixels x four chunks, 288 bytes from chunk to chunk, 1 152 from parity to parity) hit every bank once.
//    (Pixel-major, 64 bytes per pixel, the same 16 lanes read 64 bytes apart: a 4-way conflict, 45 % of the kernel's LDS cycles);
//  * K in memory order (tap, channel): one v_mfma_f32_16x16x32_f16 per tap and channel tile, nine per accumulator, from zero, in tap
//    order; bias + SiLU + fp16 rounding and the 1x1 in tail_1x1() (conv_common.hpp) -- the products, their order and the roundings of
//    conv_igemm_dma_kernel<.., TAIL> and of the wide-step kernel small launches get: bit-identical;
//  * wave w owns output rows 4w .. 4w + 3 (four 16-pixel tiles) and ALL 64 channels of them (the tail needs that), in two passes of 32
//    channels: 18 weight fragments (72 VGPRs) at a time, fetched from L2 once per pass; the pixel fragments are read twice.
//  * NW waves per block (2: 8 x 16 outputs, a 17 x 33 patch, four blocks per CU; 4: 16 x 16 outputs, two blocks per CU): what a block
//    spends its time on is latency -- the patch, the weights of each pass, the tail's weights -- and the CU hides it behind OTHER blocks.
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_c32s2_tail_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int TH = 4 * NW, TW = 16, PRH = 2 * TH + 1, PRW = 2 * TW + 1, PCP = 18, ROWB = 2 * PCP * 64;   // patch rows x columns; columns per parity (padded); bytes per patch row
    constexpr int NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 15, q = lane >> 4;
    int bx = xcd_tile((int)blockIdx.x, (int)gridDim.x, a.xcd_map);
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const half_t* xg = reinterpret_cast<const half_t*>(a.x) + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;
    const half_t* wg = reinterpret_cast<const half_t*>(a.w);

    // the first pass's weights are requested FIRST: their L2 round trip runs beside the patch's
    half8 wf[2][9];                                                                  // channel tiles 2h, 2h + 1: rows perm_row<4>(j, r), K = 32 tap + 8 q
    auto load_w = [&](int h) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const half_t* wr = wg + (size_t)perm_row<4>(2 * h + jj, r) * a.Kp + 8 * q;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wf[jj][tap] = *reinterpret_cast<const half8*>(wr + 32 * tap);
        }
    };
    load_w(0);

    // patch -> LDS: a wave instruction = 16 consecutive pixels of one row x 4 chunks = 1 KB of consecutive input bytes; nine loads in flight
    // per thread (as a rolled load-store loop a block spent ~20 us in eighteen dependent HBM round trips)
    constexpr int NSLOT = PRH * PRW * 4, NB = 9, NBATCH = (NSLOT + NB * NTHR - 1) / (NB * NTHR);
#pragma unroll
    for (int b = 0; b < NBATCH; ++b) {
        uint4 v[NB];
        int dst[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = t + (b * NB + u) * NTHR;
            const int pr = idx / (PRW * 4), rem = idx - pr * (PRW * 4), pc = rem >> 2, c = rem & 3;
            const int iy = iy0 + pr, ix = ix0 + pc;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (idx < NSLOT && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v[u] = *reinterpret_cast<const uint4*>(xg + ((size_t)iy * a.W + ix) * a.x_cs + c * 8);
            dst[u] = idx < NSLOT ? pr * ROWB + (((pc & 1) * 4 + c) * PCP + (pc >> 1)) * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (dst[u] >= 0) *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
    }
    // this lane's fragment of tap (kh, kw), output row oyl, pixel r: patch pixel (2 oyl + kh, 2 r + kw), chunk q
    int toff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        toff[tap] = kh * ROWB + (((kw & 1) * 4 + q) * PCP + (kw >> 1) + r) * 16;
    }
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    static_for<2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        if constexpr (h == 1) {
            asm volatile("" ::: "memory");                                           // the second pass's weights are fetched HERE, not ahead of the first pass (144 + 64 + 36 registers: one wave per SIMD)
            __builtin_amdgcn_sched_barrier(0);
            load_w(1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* base = smem + (2 * (4 * wv + i)) * ROWB;
            half8 xf[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) xf[tap] = *reinterpret_cast<const half8*>(base + toff[tap]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[i][2 * h + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj][tap], xf[tap], acc[i][2 * h + jj], 0, 0, 0);
        }
    });
    int mrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mrow[i] = (img * a.Ho + oy0 + 4 * wv + i) * a.Wo + ox0 + r;
    tail_1x1<4, 4>(a, acc, mrow, lane);
}

template <int NW>
static void launch_c32s2_tail(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)(8 * NW + 1) * 2 * 18 * 64;
    const int tiles_x = a.Wo / 16, tiles_y = a.Ho / (4 * NW), n_img = a.M / (a.Ho * a.Wo);
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32s2_tail_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(conv3x3_c32s2_tail_kernel<NW>, dim3((unsigned)(n_img * tiles_x * tiles_y)), dim3(64 * NW), lds, s, a, tiles_x, tiles_y);
    KCHECK();
}

static bool try_c32s2_tail(const ConvArgs& a, hipStream_t s) {
    static const int mode = [] { const char* e = getenv("AICAM_C32S2"); return e ? atoi(e) : 2; }();   // waves per block: 2 (default) or 4; 0: off (AICAM_NO_C32S2=1 as well)
    static const bool off = getenv("AICAM_NO_C32S2") != nullptr;
    if (off || mode <= 0 || !a.w_tail || a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.Cin != 32 || a.Cout != 64 || a.Kp != 288) return false;
    if (a.act != 1 || a.res_mode != 0 || a.out_f32 || a.k_order != 0 || a.xs || a.x2 || a.n_dev || a.t_max || a.t_box) return false;
    if (a.t_cout > 64 || a.t_cout % 8 || a.t_kp != 64 || a.cout_pad < 64) return false;
    if (a.Ho % 16 || a.Wo % 16 || a.Ho != (a.H + 1) / 2 || a.Wo != (a.W + 1) / 2 || (a.x_cs | a.x_coff | a.t_y_cs | a.t_y_coff) % 8) return false;
    const long blocks = (long)(a.M / (a.Ho * a.Wo)) * (a.Wo / 16) * (a.Ho / 16);
    if (blocks < 512 || (long)a.M * std::max(a.t_y_cs, 1) >= (1l << 31)) return false;     // a few tiles: the wide-step kernel (one block per CU there)
    if (mode == 4) launch_c32s2_tail<4>(a, s);
    else launch_c32s2_tail<2>(a, s);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Streaming 1x1 conv for 64 output channels and at most 128 input channels, fp16 (YOLOv8n's `4.c2f.cv2` and `15.c2f.cv2` at large batch:
// 1 GB in + out per 512 frames each, 2x their HBM floor on the LDS-DMA implicit GEMM, whose K loop is three or four steps long).  No LDS, no
// barrier: a wave keeps the whole weight matrix as MFMA A fragments (KS x 4 tiles) and walks 16-pixel tiles; a lane fetches ITS 16 bytes of
// every K-step of a tile straight from memory (16 pixels x 64 contiguous bytes per instruction), the next tile's under this tile's MFMAs and
// epilogue.  K in memory order from zero, the shared epilogue: the bits of the implicit GEMM.
// (Measured late in round 5: two tiles ahead instead of one -- 8 KB in flight per wave -- and a grid of exactly the resident blocks (768 at 152
//  registers; the 1 024 launched are 1.33 rounds): 4.c2f.cv2 347 / 335 us against 339, 15.c2f.cv2 290 / 267 against 252.  It is not
//  bytes in flight that this kernel waits for.  Not kept.)
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv1x1_stream_kernel(const ConvArgs a, int n_tiles) {
    const int t = threadIdx.x, lane = t & 63, r = lane & 15, q = lane >> 4;
    const int wave = (int)blockIdx.x * 4 + (t >> 6), n_waves = (int)gridDim.x * 4;
    const half_t* wg = reinterpret_cast<const half_t*>(a.w);
    half8 wf[4][KS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[j][s] = *reinterpret_cast<const half8*>(wg + (size_t)perm_row<4>(j, r) * a.Kp + 32 * s + 8 * q);
    // a load instruction = 16 pixels x the 64 bytes of one K-step, lane L = (pixel L >> 2, 16-byte chunk L & 3): four consecutive lanes read
    // 64 consecutive bytes.  (Fetched in the MFMA operand's own lane order -- lane (r, q) = (pixel r, chunk q), consecutive lanes a pixel
    // pitch apart -- every lane is a memory request of its own: measured no faster than the implicit GEMM.)  The operand order is
    // restored inside the wave: lane (r, q) takes the four dwords of lane 4 r + q (ds_bpermute).
    const char* xb = reinterpret_cast<const char*>(a.x) + ((size_t)a.x_coff + 8 * (lane & 3)) * 2;
    const size_t pitch = (size_t)a.x_cs * 2;
    const int src_lane4 = (4 * r + q) * 4;
    auto fetch = [&](int tile, int4 (&xr)[KS]) {
        const int m = min(tile * 16 + (lane >> 2), a.M - 1);   // (rows past the end: any pixel; nothing of them is stored)
        const char* p = xb + (size_t)m * pitch;
#pragma unroll
        for (int s = 0; s < KS; ++s) xr[s] = *reinterpret_cast<const int4*>(p + 64 * s);
    };
    int4 xr[KS], xn[KS];
    int tile = wave;
    if (tile < n_tiles) fetch(tile, xr);
    for (; tile < n_tiles; tile += n_waves) {
        const int nt = tile + n_waves;
        if (nt < n_tiles) fetch(nt, xn);
        floatx4 acc[1][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            int4 g;
            g.x = __builtin_amdgcn_ds_bpermute(src_lane4, xr[s].x), g.y = __builtin_amdgcn_ds_bpermute(src_lane4, xr[s].y);
            g.z = __builtin_amdgcn_ds_bpermute(src_lane4, xr[s].z), g.w = __builtin_amdgcn_ds_bpermute(src_lane4, xr[s].w);
            const half8 xf = __builtin_bit_cast(half8, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j][s], xf, acc[0][j], 0, 0, 0);
        }
        const int m = tile * 16 + r;
        const int mrow[1] = {m < a.M ? m : -1};
        epilogue_dispatch<half_t, 1, 4, true>(a, acc, mrow, 0, q);
#pragma unroll
        for (int s = 0; s < KS; ++s) xr[s] = xn[s];
    }
}

template <int KS>
static void launch_1x1_stream(const ConvArgs& a, hipStream_t s) {
    const int n_tiles = (a.M + 15) / 16;
    const int blocks = std::min((n_tiles + 3) / 4, 256 * 4);           // 3 - 4 waves per SIMD resident: every wave walks a strided run of tiles
    hipLaunchKernelGGL(conv1x1_stream_kernel<KS>, dim3(blocks), dim3(256), 0, s, a, n_tiles);
    KCHECK();
}

static bool try_1x1_stream(const ConvArgs& a, hipStream_t s) {
    static const bool off = getenv("AICAM_NO_1X1_STREAM") != nullptr;
    if (off || a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.Cout != 64 || a.Cin % 32 || a.Cin > 128 || a.Cin < 64 || a.Kp != a.Cin) return false;
    if (a.res_mode != 0 || a.out_f32 || a.k_order != 0 || a.xs || a.x2 || a.w_tail || a.n_dev || a.bias_init || a.cout_pad < 64) return false;
    if (a.M < 150000 || (a.x_cs | a.x_coff | a.y_cs | a.y_coff) % 8 || (long)a.M * std::max(a.x_cs, a.y_cs) >= (1l << 31)) return false;
    if (a.Cin == 128) launch_1x1_stream<4>(a, s);
    else if (a.Cin == 96) launch_1x1_stream<3>(a, s);
    else launch_1x1_stream<2>(a, s);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Persistent, weights-resident 3x3 / stride 1 / pad 1 for Cin = Cout = 64, fp16 (ReID layer1: 21 % of the FLOPs, tensors
// of 1 GB per 128-frame launch group, K = 576 only).  A tile's K loop is too short to amortise a block's prologue and
// epilogue (conv3x3_patch_kernel: 38 % MFMA busy), and every block re-fetches the 72 KB of weights through L2 -> LDS.
// Here one 8-wave block per CU walks many 8 x 32-pixel tiles:
//  * the weights never touch LDS: wave w keeps the A fragments of its 32 output channels (half w>>2) for all 18
//    K-steps in 144 VGPRs, loaded once per kernel;
//  * the input patch (10 x 34 pixels x 64 channels, pixel-major: 128 bytes per pixel, the eight 16-byte channel chunks
//    XOR-swizzled by the column so a fragment read is conflict-free) is triple-buffered; a wave's LDS-DMA instruction
//    fetches 8 pixels x 128 contiguous bytes (8 cache lines; the planar layout used before touched 64 per
//    instruction); no global load inside the K loop;
//  * a wave owns 4 output rows x 16 pixels: an input-row fragment is read once per (tap column, channel half) and
//    feeds up to three output rows (36 ds_reads per tile instead of 72), software-pipelined by one group of 24 MFMAs
//    with hand-placed lgkmcnt waits;
//  * waves w and w+4 share a SIMD and a pixel group (other channel half) and run half a tile apart (two barriers per
//    tile), so one wave's barrier wait, DMA issue and epilogue sit under its partner's MFMAs.
// Measured (tools/conv_bench.py, 7680 crops): 859 TFLOP/s without / 774 with residual; without stores 1150, with
// L2-resident input and no stores 1275 -- the remaining gap is the 2 GB of output writes, which do not overlap.
template <int ACT, int RES>
__global__ __launch_bounds__(512) void conv3x3_c64_resident_kernel(const ConvArgs a, int n_tiles_bound, int tiles_x, int tiles_y, int nblk) {
    const int n_tiles = a.n_dev ? min(n_tiles_bound, min(a.n_dev[0], n_tiles_bound / (tiles_x * tiles_y)) * tiles_x * tiles_y) : n_tiles_bound;   // device-side item count
    // nblk = gridDim.x as an argument: read through the dispatch packet it is a scalar load inside the tile loop, and a
    // scalar load in flight makes every compiler-made LDS wait an lgkmcnt(0)
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
#pragma unroll
    for (int s2 = 0; s2 < 18; ++s2) asm volatile("" :: "v"(wreg[s2][0]), "v"(wreg[s2][1]));    // weights landed: no compiler-made vmcnt(0) inside the loop
    // bias in this wave's accumulator order, parked in LDS behind the patch buffers (8 VGPRs the K loop needs more)
    float* bias_l = reinterpret_cast<float*>(smem + 3 * PBUF) + (ch * 4 + q) * 8;
    if (pg == 0 && r == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias_l[4 * j + e] = a.bias[32 * ch + perm_ch<2>(j, q, e)];
    }
    // tile order: the tiles of one image stay on one XCD (blocks b, b+8, ... share an L2) so halo rows are L2 hits
    const int tpi = tiles_x * tiles_y;
    auto tile_of = [&](int k) -> int {
        if (tpi == 8 && (nblk & 63) == 0) {
            const int xcd = blockIdx.x & 7, sl = blockIdx.x >> 3, per = nblk >> 6;     // images in flight per XCD
            const int im = ((sl >> 3) + per * k) * 8 + xcd;
            return im * 8 + (sl & 7);
        }
        return blockIdx.x + k * nblk;
    };
    auto issue_patch = [&](int tile, int buf) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / tpi;
        const int oy0 = ty * TH, ox0 = tx * TW;
        int ln = lane;
        asm volatile("" : "+v"(ln));            // recompute the per-pass lane terms here: hoisted out of the tile loop they cost 12+ VGPRs
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int sl = i * 512 + wv * 64 + ln;              // 16-byte slot of the patch image: pixel sl / 8, slot sl % 8
            const int p = sl >> 3, j = sl & 7;
            const int py = p / PW, px = p - py * PW;
            const int c = j ^ (px & 7);                         // source-side swizzle: slot j of a pixel holds channel chunk j ^ (column & 7)
            const int iy = oy0 + py - 1, ix = ox0 + px - 1;
            const bool ok = p < NPIX && tile < n_tiles && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const half_t* src = ok ? xg + ((size_t)(img * a.H + iy) * a.W + ix) * a.x_cs + a.x_coff + c * 8 : zero;
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * PBUF + i * 8192 + wv * 1024), 16, 0, 0);
        }
    };

    // A wave owns 4 output rows x 16 pixels (row group rg, column half cx): an input-row fragment is read from LDS
    // once per (tap column, channel half) and feeds the up-to-three output rows it is a tap row of -- 36 ds_reads per
    // tile instead of 72, which takes the LDS pipe (shared by the CU's 8 waves) off the critical path.
    const int rg = pg >> 1, cx = pg & 1;
    int xk[3];                                  // byte address of (patch row 4 rg, column cx 16 + r + kw, channel chunk q) in buffer 0
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = cx * 16 + r + kw;
        xk[kw] = ((4 * rg * PW + px) * 8 + (q ^ (px & 7))) * 16;
    }

    half_t* yg = reinterpret_cast<half_t*>(a.y);
    const half_t* resg = reinterpret_cast<const half_t*>(a.res);
    const int trips = (n_tiles + nblk - 1) / nblk;                          // same trip count for every block
    issue_patch(tile_of(0), 0);
    issue_patch(tile_of(1), 1);                                             // (zero page when past the end)

    floatx4 acc[4][2];
    half8 rv[4];
    size_t pix0 = 0;
    // The residual vectors are fetched with inline-asm loads: an ordinary load inside this loop makes the compiler
    // place s_waitcnt vmcnt(0) in front of the first ds_read of every tile (it then orders LDS reads behind the
    // in-flight LDS-DMA), which exposes the whole DMA latency.  They are requested at the start of a tile and waited
    // for (res_wait, counted like the patches) just before the epilogue.
    // addresses = uniform 64-bit tile base (SGPRs) + one 32-bit lane offset (VGPR) + a uniform row step
    const unsigned yl0 = (unsigned)(((4 * rg * a.Wo + cx * 16 + r) * a.y_cs + a.y_coff + 32 * ch + 8 * q) * 2);
    const unsigned rl0 = (unsigned)(((4 * rg * a.Wo + cx * 16 + r) * a.r_cs + a.r_coff + 32 * ch + 8 * q) * 2);
    const unsigned ystep = (unsigned)(a.Wo * a.y_cs * 2), rstep = (unsigned)(a.Wo * a.r_cs * 2);
    auto begin = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, img = tile / tpi;
        pix0 = ((size_t)img * a.Ho + ty * TH) * a.Wo + tx * TW;
        if constexpr (RES == 1) {
            const char* rb = reinterpret_cast<const char*>(resg) + pix0 * a.r_cs * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rv[i]) : "v"(rl0 + i * rstep), "s"(rb) : "memory");
        }
    };
    auto init_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = *reinterpret_cast<const floatx4*>(bias_l + 4 * j);
    };
    // K loop: six (tap column, channel half) groups of six input-row fragments.  Fragment ir of group g feeds the
    // output rows ir - kh (2, 4, 6, 6, 4, 2 MFMAs) and is refilled with group g+1's fragment as soon as those are
    // issued, so every ds_read has a whole group of MFMAs (24) to land under, with 24 VGPRs of fragments in all.
    // The reads and their waits are inline asm: with a global_load_lds in flight the compiler turns every wait it
    // places itself into lgkmcnt(0) / vmcnt(0) (it treats the LDS-DMA as a FLAT access that may touch both), i.e.
    // one fully exposed LDS round trip per group.  LDS returns in order and the rotation leaves exactly five newer
    // reads behind the fragment about to be used (5 - ir in the last group, which refills nothing).
    half8 xf[6];
    int xb[3];                                  // xk + buffer offset of the current tile
    auto xread = [&xf, &xb](auto g, auto ir) {
        constexpr int gg = decltype(g)::value, ii = decltype(ir)::value, kw = gg >> 1;
        const int ad = (gg & 1) ? (xb[kw] ^ 64) : xb[kw];      // channel half 1: chunk q + 4, i.e. slot ^ 4
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[ii]) : "v"(ad), "n"(ii * PW * 128));
    };
    auto xwait = [&xf](auto n, auto ir) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[decltype(ir)::value]) : "n"(decltype(n)::value));
    };
    auto kfirst = [&]() {
        static_for<6>([&](auto ir) { xread(std::integral_constant<int, 0>{}, ir); });
    };
    auto kpart = [&](auto part) {               // groups 3 part .. 3 part + 2: 72 MFMAs
        static_for<3>([&](auto gi) {
            constexpr int g = 3 * decltype(part)::value + decltype(gi)::value, kw = g >> 1, cc = g & 1;
            static_for<6>([&](auto ir_c) {
                constexpr int ir = decltype(ir_c)::value;
                xwait(std::integral_constant<int, (g == 5 ? 5 - ir : 5)>{}, ir_c);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int i = ir - kh;
                    if (i < 0 || i > 3) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[(kh * 3 + kw) * 2 + cc][j], xf[ir], acc[i][j], 0, 0, 0);
                }
                if constexpr (g + 1 < 6) xread(std::integral_constant<int, g + 1>{}, ir_c);
                __builtin_amdgcn_sched_barrier(0);          // keep the read / MFMA interleave as written
            });
        });
    };
    auto finish = [&](bool valid) {
        char* yb = reinterpret_cast<char*>(yg) + pix0 * a.y_cs * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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
            if (valid) *reinterpret_cast<half8*>(yb + (size_t)(yl0 + i * ystep)) = o;
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // Two barriers per tile; the waves of channel half 1 run HALF A TILE behind those of half 0 (they pass one extra
    // barrier first, the early ones one extra at the end).  A SIMD holds one wave of each group, so one wave's barrier
    // wait, LDS-DMA issue, first ds_reads and epilogue sit under its partner's 72-MFMA half instead of both waves
    // leaving the matrix pipe idle together.  Barrier 2k: early group starts tile k, late group is mid tile k-1;
    // barrier 2k+1: late group starts tile k (patch k is complete since barrier 2k), early group is mid tile k and
    // nobody reads buffer (k+2) % 3 any more: the late group issues its planes of patch k+2 right there, the early
    // group after its epilogue (no accumulators live while the addresses are formed).
    // vmcnt, oldest first -- early group at the top of tile k: patch k | stores k-1 | patch k+1; late group mid tile
    // k: patch k+1 | stores k-1 | (residual k |) patch k+2: the patch that must have landed before the barrier is the
    // oldest, NPASS + 4 may stay in flight (NPASS in the first trip, which has no stores yet).
    const bool late = ch == 1;
    if (late) {
        wait_vmcnt<NPASS>();
        __builtin_amdgcn_s_barrier();
    }
    int buf = 0;                                                            // k % 3
    for (int k = 0; k < trips; ++k) {
        const bool valid = tile_of(k) < n_tiles;            // a tile past the end (last trip only) computes on the zero
        const int tile = valid ? tile_of(k) : n_tiles - 1;  // page and stores nothing: no divergent control flow around the barriers
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xb[kw] = xk[kw] + buf * PBUF;
        int nb = buf + 2; if (nb >= 3) nb -= 3;
        if (!late) {
            if (k == 0) wait_vmcnt<NPASS>(); else wait_vmcnt<NPASS + 4>();
            __builtin_amdgcn_s_barrier();
            begin(tile); kfirst(); init_acc(); kpart(P0{});
            __builtin_amdgcn_s_barrier();
            kpart(P1{});
            if constexpr (RES == 1)                         // residual landed (behind it only patch k+1, a whole tile old)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]) :: "memory");
            finish(valid);
            issue_patch(tile_of(k + 2), nb);
        } else {
            __builtin_amdgcn_s_barrier();
            begin(tile);
            issue_patch(tile_of(k + 2), nb);
            kfirst(); init_acc(); kpart(P0{});
            if (k == 0) wait_vmcnt<NPASS>(); else wait_vmcnt<NPASS + 4>();
            __builtin_amdgcn_s_barrier();
            kpart(P1{});
            if constexpr (RES == 1)                         // residual landed; patch k+2 (issued after it) may stay in flight
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]) : "n"(NPASS) : "memory");
            finish(valid);
        }
        if (++buf == 3) buf = 0;
    }
    if (!late) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();
}

static bool try_c64_resident(const ConvArgs& a, hipStream_t s) {
    static const int on = [] { const char* e = getenv("AICAM_C64R"); return e ? atoi(e) : 2; }();   // 0: off, 1: layers without residual only, 2 (default): also with residual (612 -> 774 TFLOP/s against the 4-wave patch kernel)
    if (!on || (a.res_mode != 0 && on < 2) || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin != 64 || a.Cout != 64 || a.out_f32 || a.Kp != 576) return false;
    if (a.W % 32 || a.H % 8 || a.Ho != a.H || a.Wo != a.W || a.M < 1500000 || (long)a.M * a.x_cs >= (1l << 31) ||
        (long)a.M * a.y_cs >= (1l << 31) || (a.res_mode != 0 && (long)a.M * a.r_cs >= (1l << 31))) return false;
    if ((a.x_cs | a.x_coff | a.y_cs | a.y_coff | a.r_cs | a.r_coff) % 8) return false;
    const int tiles_x = a.W / 32, tiles_y = a.H / 8, n_img = a.M / (a.H * a.W), n_tiles = n_img * tiles_x * tiles_y;
    constexpr size_t lds = (size_t)3 * 8 * 384 * 16 + 256;
    auto launch = [&](auto kfn) {
        static bool attr = false;
        if (!attr) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        static const int split = [] { const char* e = getenv("AICAM_C64R_SPLIT"); return e ? std::max(1, atoi(e)) : 1; }();
        const int nblk = conv_cu_budget() * split;
        hipLaunchKernelGGL(kfn, dim3(nblk), dim3(512), lds, s, a, n_tiles, tiles_x, tiles_y, nblk);
        KCHECK();
    };
    if (a.act == 2 && a.res_mode == 0) launch(conv3x3_c64_resident_kernel<2, 0>);
    else if (a.act == 2 && a.res_mode == 1) launch(conv3x3_c64_resident_kernel<2, 1>);
    else return false;
    return true;
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
// TAIL: a 1x1 conv (a.w_tail) runs in the epilogue on the tile in registers (tail_1x1, conv_common.hpp).
// KORD: 0 = K-steps in memory order (kh, kw, cc); 2 = (kw, cc, kh), the accumulation order of the weights-resident 64-channel
// kernels (ConvArgs::k_order): the same layer then gives the same bits below and above their batch threshold.
template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE, int LGCPP, bool TAIL = false, int KORD = 0>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu((NT == 5 || TAIL) ? 2 : 1))) void conv3x3_patch_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
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
    static_assert(BM == TH * TW && TW % 16 == 0 && (TW & (TW - 1)) == 0 && B_PER <= 2, "tile geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem + PATCH_BYTES;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int bx, tby;
    if (!xcd_tile_xy_live(a.xcd_map, a.n_dev ? min((int)gridDim.x, min(a.n_dev[0], (int)gridDim.x / (tiles_x * tiles_y)) * tiles_x * tiles_y) : (int)gridDim.x, bx, tby)) return;   // device-side item count: the grid was sized for a bound
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int n0 = tby * BN;

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

    // ---- weight stream: B_PER 16-byte chunks per thread per K-step (rows r0, r0 + RP); rows past Cout read the zero page with stride 0
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    const T* wptr[B_PER];
    int winc[B_PER];
#pragma unroll
    for (int jb = 0; jb < B_PER; ++jb) {
        const bool wrow_ok = r0 + RP * jb < BN;
        wptr[jb] = wrow_ok ? wg + (size_t)(n0 + r0 + RP * jb) * a.Kp + kc * CH : zero;
        winc[jb] = wrow_ok ? BKE : 0;
    }
    char* wdst = ring + (16 * wv) * 64;
    // step s of the walk -> (kh, kw, cc) and the K-step's position in the packed weights (memory order is (kh, kw, cc))
    auto step_kh = [](int s) constexpr { return KORD == 2 ? s % 3 : s / (3 * CSTEPS); };
    auto step_kw = [](int s) constexpr { return KORD == 2 ? s / (3 * CSTEPS) : (s / CSTEPS) % 3; };
    auto step_cc = [](int s) constexpr { return KORD == 2 ? (s / 3) % CSTEPS : s % CSTEPS; };
    auto step_mem = [=](int s) constexpr { return (step_kh(s) * 3 + step_kw(s)) * CSTEPS + step_cc(s); };
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) {
#pragma unroll
        for (int jb = 0; jb < B_PER; ++jb)
            __builtin_amdgcn_global_load_lds((gptr_t)(wptr[jb] + step_mem(st) * winc[jb]), (lptr_t)(wdst + st * WSTAGE + jb * (RP * 64)), 16, 0, 0);
    }
    const T* wnext[B_PER];                                   // memory order (KORD 0): the stream is a pointer increment
#pragma unroll
    for (int jb = 0; jb < B_PER; ++jb) wnext[jb] = wptr[jb] + (NSTAGE - 1) * winc[jb];

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
    if constexpr (KORD == 2) {                 // bias first (ConvArgs::bias_init): the epilogue's `bias` is a page of zeros
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            floatx4 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = a.bias_init[n0 + wn * NT * 16 + perm_ch<NT>(j, q, e)];
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = b;
        }
    }

    typedef typename Frag<T>::type frag_t;
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {                     // fully unrolled: kh / kw / cc / ring stages are compile-time
        const int kh = step_kh(step), kw = step_kw(step), cc = step_cc(step);
        const int cur = step % NSTAGE, nxt = (step + NSTAGE - 1) % NSTAGE;
        wait_vmcnt<(NSTAGE - 2) * B_PER>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int jb = 0; jb < B_PER; ++jb) {   // refill the stage that step-1 released (zero page once the real K-steps are exhausted)
            const T* src = zero;
            if constexpr (KORD == 0) {
                if (step + NSTAGE - 1 < NSTEPS) src = wnext[jb];
                wnext[jb] += winc[jb];
            } else {
                if (step + NSTAGE - 1 < NSTEPS) src = wptr[jb] + step_mem(step + NSTAGE - 1 < NSTEPS ? step + NSTAGE - 1 : 0) * winc[jb];
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wdst + nxt * WSTAGE + jb * (RP * 64)), 16, 0, 0);
        }
        frag_t xf[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(smem + xaddr[kw][cc][i] + kh * ROWB);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(smem + woff[j] + cur * WSTAGE);
        mma_tiles<T, MT, NT>(acc, wf, xf);
    }
    wait_vmcnt<0>();

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pt = (wm * MT + i) * 16 + r;
        const int oy = oy0 + pt / TW, ox = ox0 + pt % TW;
        mrow[i] = (oy < a.Ho && ox < a.Wo) ? (img * a.Ho + oy) * a.Wo + ox : -1;
    }
    if constexpr (TAIL) {
        static_assert(sizeof(T) == 2 && WN == 1, "the tail needs fp16 and a wave that owns every channel of its pixels");
        tail_1x1<MT, NT>(a, acc, mrow, lane);
    } else {
        epilogue_dispatch<T, MT, NT, true>(a, acc, mrow, n0 + wn * NT * 16, q);
    }
}

template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE, int LGCPP, bool TAIL = false, int KORD = 0>
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
    auto kfn = conv3x3_patch_kernel<T, MT, NT, WM, WN, TH, TW, NSTAGE, LGCPP, TAIL, KORD>;
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
    if (a.k_order == 1) return false;                                       // (cc, kh, kw): only the implicit-GEMM kernels walk K that way
    const bool wide = a.Wo % 32 == 0 || (a.Wo % 16 != 0 && a.Wo >= 32);   // 8 x 32 tiles unless 16 x 16 tiles cover the map exactly
    if (a.k_order == 2) {                                                   // fp16, Cin = Cout = 64, W % 32 == 0 (launch_conv_igemm): the resident kernels' order
        if constexpr (sizeof(T) == 2) return a.Cout == 64 && wide && launch_patch<T, 4, 4, 4, 1, 8, 32, 3, 3, false, 2>(a, s);
        return false;
    }
    if (a.Cout == 64) {
        constexpr int LG64 = sizeof(T) == 2 ? 3 : 4;    // Cin = 64: 8 chunks (fp16) / 16 chunks (fp32) per pixel
        if (wide) return launch_patch<T, 4, 4, 4, 1, 8, 32, 3, LG64>(a, s);
        return launch_patch<T, 4, 4, 4, 1, 16, 16, 3, LG64>(a, s);
    }
    if (a.Cout == 80) {                                 // YOLOv8n's 22.cls0.0 (64 -> 80 at 80 x 80; round 5): 461 TFLOP/s on the 512 x 80 implicit-GEMM tile
        static const bool c80 = getenv("AICAM_NO_PATCH_C80") == nullptr;
        if constexpr (sizeof(T) == 2) {
            if (!c80) return false;
            if (wide) return launch_patch<T, 4, 5, 4, 1, 8, 32, 3, 3>(a, s);
            return launch_patch<T, 4, 5, 4, 1, 16, 16, 3, 3>(a, s);
        }
        return false;
    }
    if (a.Cout == 32 && c32) {
        constexpr int LG32 = sizeof(T) == 2 ? 2 : 3;    // Cin = 32
        if (wide) return launch_patch<T, 4, 2, 4, 1, 8, 32, 3, LG32>(a, s);
        return launch_patch<T, 4, 2, 4, 1, 16, 16, 3, LG32>(a, s);
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// The patch form with a PIXEL-MAJOR patch, for the detect branches' second convs with the 1x1 behind them (YOLOv8n's `22.cls{l}.1` + `.2`,
// `22.box{l}.1` + `.2` at launches of 50 000 pixels and more; round 5).  80 channels are ten 16-byte chunks per pixel -- not a power of two,
// not a multiple of the 32-element K-step -- so conv3x3_patch_kernel (chunk index by shift and mask, XOR swizzle, one tap per K-step) does
// not take the class branch and it ran on the implicit GEMM's generic gather path (880 us per 512 frames at level 0, 23 % MFMA busy; the
// patch kernel runs the 64 -> 80 conv beside it at 750 TFLOP/s).  Here:
//  * the 18 x 18 x Cin patch of a 16 x 16 tile goes into LDS ONCE by LDS-DMA, pixel-major, PITCH chunks per pixel, unswizzled: 80 channels at
//    160 bytes per pixel (the 16 lanes of a fragment read are 160 bytes apart: a 2-way conflict on four reads per 20 MFMAs; a pitch of 176
//    would be conflict-free and would not leave room for two blocks per CU), 64 channels at 144 (128 would be an 8-way conflict);
//  * K in memory order (tap, channel) in 32-element steps FROM k = 0, as the implicit GEMM walks it: step s is the flattened chunks
//    4 s .. 4 s + 3 = (tap, chunk) (kc / CPP, kc % CPP) -- with ten chunks a step straddles taps, so a lane's (tap, chunk) depends on its q:
//    per-lane offsets, worked out once; the last step's chunks 90, 91 meet the zero columns of the padded weights (Kp = 736) and read chunk 0;
//  * weights through the 3-stage ring of conv3x3_patch_kernel (80 channels: 128 padded rows, two 16-byte chunks per thread and step), the
//    epilogue is tail_1x1; tiles may hang over the map's edge (a 40 x 40 map is 3 x 3 tiles);
//  * bit-identical to conv_igemm_dma_kernel<.., TAIL> / conv3x3_patch_kernel<.., TAIL> (child-process test at 32 frames).
// 22.cls0.1 + .2: 883 -> 487 us per 512 frames (860 TFLOP/s), 22.cls1.1 + .2: 250 -> 190; the 64-channel form against conv3x3_patch_kernel's
// tail form: 513 -> 500 and 191 -> 177 (without a tail it is no faster than that kernel: measured, not kept).
// CPP: 16-byte chunks per input pixel (Cin / 8); PITCH: chunks a pixel takes in LDS (CPP, or CPP + 1 where CPP chunks would put the 16 lanes
// of a fragment read on the same banks: 8 chunks = 128 bytes is an 8-way conflict, 9 is none); NT: channel tiles (Cout / 16); TAIL: a.w_tail's 1x1.
// TH x TW: the block's output tile, 64 MT pixels, walked row-major in 16-pixel MFMA tiles (TW = 16: a tile row each; TW = 8: two rows of eight --
// a 40 x 40 map is five 40 x 8 strips exactly, where 16 x 16 tiles cover 1.44 maps and conv3x3_patch_kernel's 8 x 32 tiles 1.6).
template <int CPP, int PITCH, int NT, bool TAIL, int TH = 16, int TW = 16, int MT = 4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT >= 9 ? 1 : 2))) void conv3x3_pm_patch_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    typedef half_t T;
    constexpr int PW = TW + 2, PH = TH + 2, NSTAGE = 3;
    static_assert(TH * TW == 64 * MT && (TW == 16 || TW == 8), "tile geometry");
    constexpr int NTHR = 256, RP = 64, BN = NT * 16, BNP = (BN + RP - 1) / RP * RP, B_PER = BNP / RP, WSTAGE = BNP * 64, BKE = 32, CH = 8;
    constexpr int TOTAL = PH * PW * PITCH, PATCH_BYTES = (TOTAL + NTHR - 1) / NTHR * NTHR * 16, NSTEPS = (9 * CPP + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem + PATCH_BYTES;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int bx = xcd_tile((int)blockIdx.x, (int)gridDim.x, a.xcd_map);
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);
    const T* ximg = reinterpret_cast<const T*>(a.x) + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;

    // ---- the input patch, once (out-of-image pixels come from the zero page)
#pragma unroll 2
    for (int base = 0; base < TOTAL; base += NTHR) {
        const int L = base + t;
        const int p = L / PITCH, j = L - p * PITCH;
        const int py = p / PW, px = p - py * PW;
        const int iy = oy0 + py - 1, ix = ox0 + px - 1;
        const bool ok = L < TOTAL && j < CPP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const T* src = ok ? ximg + ((size_t)iy * a.W + ix) * a.x_cs + j * CH : zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (size_t)(base + 64 * wv) * 16), 16, 0, 0);
    }
    // ---- weight stream: two 16-byte chunks per thread per K-step (rows r0, r0 + 64); rows past Cout read the zero page with stride 0
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    const T* wnext[B_PER];
    int winc[B_PER];
#pragma unroll
    for (int jb = 0; jb < B_PER; ++jb) {
        const bool wrow_ok = r0 + RP * jb < BN;
        wnext[jb] = wrow_ok ? wg + (size_t)(r0 + RP * jb) * a.Kp + kc * CH : zero;
        winc[jb] = wrow_ok ? BKE : 0;
    }
    char* wdst = ring + (16 * wv) * 64;
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
#pragma unroll
        for (int jb = 0; jb < B_PER; ++jb) {
            __builtin_amdgcn_global_load_lds((gptr_t)wnext[jb], (lptr_t)(wdst + st * WSTAGE + jb * (RP * 64)), 16, 0, 0);
            wnext[jb] += winc[jb];
        }

    const int q = lane >> 4, r = lane & 15;
    int koff[NSTEPS];                           // this lane's (tap, chunk) of every K-step, as a byte offset from its pixel's tap (0, 0), chunk 0
#pragma unroll
    for (int s2 = 0; s2 < NSTEPS; ++s2) {
        int c = 4 * s2 + q;
        if (c >= 9 * CPP) c = 0;
        const int tap = c / CPP, ch = c - tap * CPP, kh = tap / 3, kw = tap - 3 * kh;
        koff[s2] = ((kh * PW + kw) * PITCH + ch) * 16;
    }
    int xbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {             // pixel tile (wave, i): pixels 16 (wv MT + i) .. + 15 of the tile in row-major order
        const int pt = (wv * MT + i) * 16 + r;
        xbase[i] = ((pt / TW) * PW + pt % TW) * (PITCH * 16);
    }
    int woff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = PATCH_BYTES + lds_off(perm_row<NT>(j, r), q);

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {                     // fully unrolled: ring stages are compile-time
        const int cur = step % NSTAGE, nxt = (step + NSTAGE - 1) % NSTAGE;
        wait_vmcnt<(NSTAGE - 2) * B_PER>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int jb = 0; jb < B_PER; ++jb) {    // refill the stage that step - 1 released (zero page once the real K-steps are exhausted)
            const T* src = step + NSTAGE - 1 < NSTEPS ? wnext[jb] : zero;
            wnext[jb] += winc[jb];
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wdst + nxt * WSTAGE + jb * (RP * 64)), 16, 0, 0);
        }
        half8 xf[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const half8*>(smem + xbase[i] + koff[step]);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const half8*>(smem + woff[j] + cur * WSTAGE);
        mma_tiles<T, MT, NT>(acc, wf, xf);
    }
    wait_vmcnt<0>();

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {             // (tiles may hang over the map's edge -- a 40 x 40 map is 3 x 3 of them: those pixels are computed on zeros and not stored)
        const int pt = (wv * MT + i) * 16 + r;
        const int oy = oy0 + pt / TW, ox = ox0 + pt % TW;
        mrow[i] = (oy < a.Ho && ox < a.Wo) ? (img * a.Ho + oy) * a.Wo + ox : -1;
    }
    if constexpr (TAIL) tail_1x1<MT, NT>(a, acc, mrow, lane);
    else epilogue_dispatch<T, MT, NT, true>(a, acc, mrow, 0, q);
}

template <int CPP, int PITCH, int NT, bool TAIL, int TH = 16, int TW = 16, int MT = 4>
static bool launch_pm_patch(const ConvArgs& a, hipStream_t s) {
    constexpr int BNP = (NT * 16 + 63) / 64 * 64;
    constexpr size_t lds = (size_t)(((TH + 2) * (TW + 2) * PITCH + 255) / 256 * 256) * 16 + (size_t)3 * BNP * 64;
    static_assert((NT >= 9 ? 1 : 2) * lds <= 160 * 1024, "two blocks per CU (the 144-channel form: one, its patch alone is 112 KB)");
    auto kfn = conv3x3_pm_patch_kernel<CPP, PITCH, NT, TAIL, TH, TW, MT>;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const int tiles_x = ceil_div(a.Wo, TW), tiles_y = ceil_div(a.Ho, TH), n_img = a.M / (a.Ho * a.Wo);
    hipLaunchKernelGGL(kfn, dim3(n_img * tiles_x * tiles_y), dim3(256), lds, s, a, tiles_x, tiles_y);
    KCHECK();
    return true;
}

// shapes every form shares: 3x3 / 1 / 1 on whole maps, fp16, memory K order, launches of 50 000 pixels and more (below: a few tiles, the
// wide-step kernel or v2)
static bool pm_patch_shape(const ConvArgs& a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.k_order != 0 || a.xs || a.x2 || a.n_dev || a.out_f32 || a.bias_init) return false;
    return a.Ho == a.H && a.Wo == a.W && a.M >= 50000 && (a.x_cs | a.x_coff | a.y_cs | a.y_coff | a.r_cs | a.r_coff) % 8 == 0;
}
// 16 x 16 tiles cover the map with at most half as many pixels again hanging over the edge (40 x 40: 1.44; 20 x 20 would be 2.56)
static bool pm_cover16(const ConvArgs& a) {
    const long cover = (long)ceil_div(a.Wo, 16) * 16 * ceil_div(a.Ho, 16) * 16;
    return 2 * cover <= 3 * (long)a.Wo * a.Ho;
}

// lead + 1x1 tail: 80 -> 80 (the class branches), 64 -> 64 (the box branches)
bool conv_try_pm_patch_tail(const ConvArgs& a, hipStream_t s) {
    static const bool off = getenv("AICAM_NO_PATCH_C80") != nullptr;
    if (off || !a.w_tail || !pm_patch_shape(a) || a.res_mode != 0 || a.act != 1) return false;
    if (a.Cin == 80 && a.Cout == 80 && a.Kp == 736 && a.cout_pad >= 128 && pm_cover16(a)) return launch_pm_patch<10, 10, 5, true>(a, s);
    if (a.Cin == 64 && a.Cout == 64 && a.Kp == 576 && a.cout_pad >= 64) {
        if (a.Ho == 40 && a.Wo % 8 == 0) return launch_pm_patch<8, 9, 4, true, 40, 8, 5>(a, s);
        if (pm_cover16(a)) return launch_pm_patch<8, 9, 4, true>(a, s);
    }
    return false;
}
// without a tail: 64 -> 64 on 40-row maps in 40 x 8 strips (YOLOv8n's P4 bottlenecks: eight layers on conv3x3_patch_kernel's 8 x 32 tiles,
// which cover 1.6 maps).  On maps its 16 x 16 tiles cover exactly this form is no faster than that kernel (measured: 22.box0.0 348 against 354 us).
bool conv_try_pm_patch(const ConvArgs& a, hipStream_t s) {
    static const bool off = getenv("AICAM_NO_PATCH_C80") != nullptr || getenv("AICAM_NO_PM_STRIPS") != nullptr;
    if (off || a.w_tail || !pm_patch_shape(a) || a.act != 1 || (a.res_mode != 0 && a.res_mode != 2)) return false;
    if (a.Cin == 64 && a.Cout == 64 && a.Kp == 576 && a.cout_pad >= 64 && a.Ho == 40 && a.Wo % 8 == 0) return launch_pm_patch<8, 9, 4, false, 40, 8, 5>(a, s);
    // (32 -> 32 on 80 x 80 maps -- YOLOv8n's P3 bottlenecks, which conv3x3_patch_kernel's 8 x 32 tiles cover 1.2 times -- as
    //  launch_pm_patch<4, 5, 2, false> on 16 x 16 tiles that cover the map exactly: built, bit-identical, measured 843 against 830 us for the
    //  four layers of 4.c2f.  They do not wait for their tiles: they read and write 64-byte slices of a 256-byte-pitch concat buffer.  Not kept.)
    // 128 -> 144 on 40-row maps: the merged first convs of YOLOv8n's 40 x 40 detect level (22.box1.0 + 22.cls1.0), which the implicit GEMM runs on a
    // 256 x 144 tile with 36 accumulator tiles per wave, one wave per SIMD.  Sixteen chunks per pixel at a pitch of 17 (272 bytes: the 16 lanes of a
    // fragment read sit four banks apart), nine channel tiles (the odd last one keeps the identity map), one block per CU.  AICAM_NO_PM144=1: off
    static const bool no144 = getenv("AICAM_NO_PM144") != nullptr;
    if (!no144 && a.res_mode == 0 && a.Cin == 128 && a.Cout == 144 && a.Kp == 1152 && a.cout_pad >= 144 && a.Ho == 40 && a.Wo % 8 == 0)
        return launch_pm_patch<16, 17, 9, false, 40, 8, 5>(a, s);
    return false;
}

// the Cout = 64 patch kernel with a 1x1 tail (same eligibility as try_patch)
bool conv_try_patch_tail(const ConvArgs& a, hipStream_t s) {
    static const bool off = getenv("AICAM_NO_PATCH") != nullptr;
    if (off || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Wo < 16 || a.Ho < 8 || a.M < 200000 || a.Cout != 64 || a.k_order != 0) return false;
    const bool wide = a.Wo % 32 == 0 || (a.Wo % 16 != 0 && a.Wo >= 32);
    if (wide) return launch_patch<half_t, 4, 4, 4, 1, 8, 32, 3, 3, true>(a, s);
    return launch_patch<half_t, 4, 4, 4, 1, 16, 16, 3, 3, true>(a, s);
}

bool conv_try_patch(int dtype, const ConvArgs& a, hipStream_t s) {
    return dtype == AIC_F16 ? try_patch<half_t>(a, s) : false;          // (fp32 engines: the LDS-DMA implicit GEMM only, kernels_conv.hip)
}
bool conv_try_c16(const ConvArgs& a, hipStream_t s) { return try_c16(a, s); }
bool conv_try_c32s2_tail(const ConvArgs& a, hipStream_t s) { return try_c32s2_tail(a, s); }
bool conv_try_1x1_stream(const ConvArgs& a, hipStream_t s) { return try_1x1_stream(a, s); }
bool conv_try_c64_resident(const ConvArgs& a, hipStream_t s) { return try_c64_resident(a, s); }

// ------------------------------------------------------------------------------------------------
// Fused ReID stem: conv 3x3/1 (3 -> 64) + bias + ReLU + max-pool 3x3/2 (pad 1) in one kernel, fp16.
// The unfused pair writes and re-reads a [N,128,64,64] tensor (1 MB per crop) for 14 MMAC of work;
// here a block owns 4 pooled rows of one crop: the 11x66 input patch (RGB0) and the 9x64x64 conv
// tile live in LDS only, K = 27 is padded to one v_mfma_f32_16x16x32_f16 per 16 px x 16 ch tile
// (the im2col fragment is gathered from the patch), and only the pooled [N,64,32,64] tensor
// reaches HBM.  PyTorch semantics: conv zero-pads its input, the pool ignores out-of-image taps.
struct StemArgs {
    const void* x; const void* w; const float* bias; void* y;
    int n, H, W, Kp, y_cs, y_coff;   // input [n][H][W][in_stride]; output [n][H/2][W/2][y_cs]
    int in_stride;                   // halves per input pixel: 8 (NHWC8) or 4 (NHWC4, RGB0)
    // fused crop (second stem form only): frames != NULL = the block resamples its crop from the u8 frame itself
    // (_extract_image_crops + preprocess_reid_input, deepsort_tracker.py:143-159 / image_processing.py:105-138) instead of reading x
    const uint8_t* frames; int fh, fw; const float* boxes; const int* frame_of; int* valid;
    const int* n_dev;                // optional device-side crop count (n is then the bound the grid was sized for)
    int dbg;                         // AICAM_STEM_DBG (timing only, wrong outputs): 1 = leave after the patch is built, 2 = do not build it
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
    if (a.n_dev && img >= a.n_dev[0]) return;
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
//  * the bias rides in as the accumulator's initial value, ReLU is one packed max on the POOLED vector;
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

// (two blocks per CU = four waves per SIMD: 128 registers.  A build that took 132 ran one block per CU and 28 % slower)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void reid_stem_pool2_kernel(const StemArgs a) {
    constexpr int CW = 64, PW = CW + 2, NTX = CW / 16;
    constexpr int ROW_SHL1 = 0x101, ROW_SHR1 = 0x111, ROW_ROR1 = 0x121;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2* patch = reinterpret_cast<uint2*>(smem);                        // [H + 2][PW] pixels x RGB0 halves, zero border

    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 15, q = lane >> 4;
    const int H = a.H, Hp = H / 2, Wp = CW / 2, rows_per_wave = Hp / 8;
    const int img = blockIdx.x;
    if (a.n_dev && img >= a.n_dev[0]) return;               // device-side crop count: the grid was sized for a bound
    const half_t* xg = reinterpret_cast<const half_t*>(a.x) + (size_t)img * H * CW * a.in_stride;

    if (a.dbg == 2) {
    } else if (a.frames == nullptr) {
        for (int idx = t; idx < (H + 2) * PW; idx += 512) {
            const int iy = idx / PW, ix = idx - iy * PW;
            const int gy = iy - 1, gx = ix - 1;
            uint2 v = make_uint2(0u, 0u);
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)CW) v = *reinterpret_cast<const uint2*>(xg + ((size_t)gy * CW + gx) * a.in_stride);
            patch[idx] = v;
        }
    } else {
        // crop + resize + normalise straight into the patch: the arithmetic of crop_resize_kernel (kernels_pre.hip), pixel for pixel --
        // int() truncation + clamp of the box, cv2 taps (fp64 coordinates), 11-bit fixed point, (v/255 - mean)/std in fp32, fp16 RGB0.
        // The 1 GB crop tensor of a 15 360-crop launch group is never written or read back.
        __shared__ Taps xt[CW];
        __shared__ Taps yt[256];
        __shared__ float lut[3][256];
        const float* bb = a.boxes + (size_t)img * 4;
        const float lim = 1.0e9f;
        int x1 = (int)fminf(fmaxf(bb[0], -lim), lim), y1 = (int)fminf(fmaxf(bb[1], -lim), lim);
        int x2 = (int)fminf(fmaxf(bb[2], -lim), lim), y2 = (int)fminf(fmaxf(bb[3], -lim), lim);
        x1 = max(0, x1); y1 = max(0, y1); x2 = min(a.fw, x2); y2 = min(a.fh, y2);
        const bool ok = x1 < x2 && y1 < y2;
        if (t == 0 && a.valid) a.valid[img] = ok ? 1 : 0;
        const int sw = x2 - x1, sh = y2 - y1;
        const bool area2 = ok && is_area2(sw, sh, CW, H);
        if (ok && !area2) {
            if (t < CW) xt[t] = taps_x(t, 1.0 / ((double)CW / (double)sw), sw);
            else if (t - CW < H) yt[t - CW] = taps_y(t - CW, 1.0 / ((double)H / (double)sh), sh);
        }
        if (t < 256) {
            const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
            for (int c = 0; c < 3; ++c) lut[c][t] = ((float)t / 255.0f - mean[c]) / stdv[c];      // image_processing.py:126-131 (fp32)
        }
        __syncthreads();
        const uint8_t* f = a.frames + (size_t)(a.frame_of ? a.frame_of[img] : 0) * a.fh * a.fw * 3;
        const int pitch = a.fw * 3;
        // the convolution's zero padding: rows -1 and H, columns -1 and CW of the patch
        for (int idx = t; idx < 2 * PW + 2 * H; idx += 512) {
            const int e = idx < 2 * PW ? (idx < PW ? idx : (H + 1) * PW + idx - PW) : (1 + ((idx - 2 * PW) >> 1)) * PW + ((idx & 1) ? PW - 1 : 0);
            patch[e] = make_uint2(0u, 0u);
        }
        // A thread keeps ONE column (ox = t & 63: its horizontal taps, byte offset and weights are loop constants) and walks the rows
        // wv, wv + 8, ...: a wave is one row, so the vertical taps are wave-uniform.  (The form before walked the patch linearly, halo
        // included: a division by 66, two tap-table reads and the 64-bit address arithmetic per pixel -- 150 VALU instructions of which
        // 60 are left; same loads, same integer arithmetic, same table: same bits.)
        const int ox = t & 63;
        const uintptr_t fb = reinterpret_cast<uintptr_t>(f);
        const uint8_t* fa = reinterpret_cast<const uint8_t*>(fb & ~(uintptr_t)3);           // frame base rounded down to 4 bytes ...
        const unsigned fd = (unsigned)(fb & 3);                                               // ... and what was cut off
        if (!ok) {
            for (int oy = wv; oy < H; oy += 8) patch[(oy + 1) * PW + ox + 1] = make_uint2(0u, 0u);          // an empty crop is all zeros
        } else if (area2) {
            const unsigned c0 = (unsigned)(x1 + 2 * ox) * 3u + fd;
#pragma unroll 2
            for (int oy = wv; oy < H; oy += 8) {
                const uint8_t* p0 = fa + ((unsigned)(y1 + 2 * oy) * (unsigned)pitch + c0);
                const uint8_t* p1 = p0 + pitch;
                int px[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) px[c] = ((int)p0[c] + (int)p0[3 + c] + (int)p1[c] + (int)p1[3 + c] + 2) >> 2;
                const half4 hv = {(half_t)lut[0][px[2]], (half_t)lut[1][px[1]], (half_t)lut[2][px[0]], (half_t)0.f};     // BGR -> RGB
                patch[(oy + 1) * PW + ox + 1] = __builtin_bit_cast(uint2, hv);
            }
        } else {
            const Taps tx = xt[ox];
            const bool two = tx.i1 != tx.i0;
            const unsigned c0 = (unsigned)(x1 + tx.i0) * 3u + fd;
#pragma unroll 2
            for (int oy = wv; oy < H; oy += 8) {
                const Taps ty = yt[oy];
                int b0[2][3], b1[2][3];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    // both taps of a row are 6 consecutive bytes: one aligned 12-byte load per row (the frame ring has >= 16 bytes of slack)
                    const unsigned o = __umul24((unsigned)(y1 + (rr ? ty.i1 : ty.i0)), (unsigned)pitch) + c0;      // row < 2^24, pitch < 2^24 (launch_reid_stem_pool)
                    const uint3 w3 = *reinterpret_cast<const uint3*>(fa + (o & ~3u));
                    const unsigned shb = o & 3u;
                    const unsigned q0 = __builtin_amdgcn_alignbyte(w3.y, w3.x, shb), q1 = __builtin_amdgcn_alignbyte(w3.z, w3.y, shb);
                    b0[rr][0] = q0 & 255u, b0[rr][1] = (q0 >> 8) & 255u, b0[rr][2] = (q0 >> 16) & 255u;
                    b1[rr][0] = two ? (q0 >> 24) : b0[rr][0], b1[rr][1] = two ? (q1 & 255u) : b0[rr][1], b1[rr][2] = two ? ((q1 >> 8) & 255u) : b0[rr][2];
                }
                int px[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // bytes <= 255, weights <= 2048, h >> 4 < 2^15: every factor fits 24 bits, so these are full-rate v_mul / v_mad_i32_i24
                    // (a 32-bit v_mul_lo_u32 is a quarter-rate instruction and there are twelve of them per pixel) -- same integers
                    const int h0 = __mul24(b0[0][c], tx.w0) + __mul24(b1[0][c], tx.w1);
                    const int h1 = __mul24(b0[1][c], tx.w0) + __mul24(b1[1][c], tx.w1);
                    px[c] = ((__mul24(ty.w0, h0 >> 4) >> 16) + (__mul24(ty.w1, h1 >> 4) >> 16) + 2) >> 2;
                }
                const half4 hv = {(half_t)lut[0][px[2]], (half_t)lut[1][px[1]], (half_t)lut[2][px[0]], (half_t)0.f};     // BGR -> RGB
                patch[(oy + 1) * PW + ox + 1] = __builtin_bit_cast(uint2, hv);
            }
        }
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
    if (a.dbg == 1) return;

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
            // (written as a vector conversion every pair becomes a v_cvt_pk_f16_f32 and the kernel needs 132 registers: one block per CU,
            //  28 % slower; held to 128 it spills.  This form compiles to a mix of packed and scalar conversions inside 128)
            const half2_t h01 = {(half_t)acc[0], (half_t)acc[1]}, h23 = {(half_t)acc[2], (half_t)acc[3]};
            o.u[2 * ct] = __builtin_bit_cast(unsigned, h01);        // (ReLU: once, on the pooled vector -- see the store)
            o.u[2 * ct + 1] = __builtin_bit_cast(unsigned, h23);
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
                // ReLU after the pool: max(relu(a), relu(b), ...) = relu(max(a, b, ...)), and the signed-integer max above is still exact for
                // it -- a positive half beats every negative one (sign bit = negative integer) and the larger of two positives wins; if all
                // nine are negative the winner is SOME negative half, which the max against 0 here turns into the same 0.  Out-of-image
                // taps stay 0.  Four v_pk_max per 16 x 64 outputs instead of the sixteen that clamped both conv rows first.
                op[i] = pk_max((r & 1) ? nb : hm.u[i], 0u);
            }
            const size_t pix = ((size_t)img * Hp + py) * Wp + 8 * tx + (r >> 1);
            *reinterpret_cast<uint4*>(yg + pix * a.y_cs + a.y_coff + (r & 1) * 32 + 8 * q) = out;
        }
    }
}

bool reid_stem2_usable(int H, int W) {
    static const bool v1 = [] { const char* e = getenv("AICAM_STEM"); return e && e[0] == 'v' && e[1] == '1'; }();
    return !v1 && W == 64 && H % 16 == 0 && (size_t)(H + 2) * 66 * 8 <= 150 * 1024;
}

void launch_reid_stem_pool(const void* x, const void* w, const float* bias, void* y, int n, int H, int W, int Kp, int y_cs,
                           int y_coff, int in_stride, hipStream_t s, const CropSrc* crop, const int* n_dev) {
    if (n <= 0) return;
    static const int dbg = [] { const char* e = getenv("AICAM_STEM_DBG"); return e ? atoi(e) : 0; }();
    StemArgs a{x, w, bias, y, n, H, W, Kp, y_cs, y_coff, in_stride, nullptr, 0, 0, nullptr, nullptr, nullptr, n_dev, dbg};
    if (crop && crop->frames) {
        AIC_REQUIRE(reid_stem2_usable(H, W) && H <= 256 - 64, AIC_ERR_INVALID, "fused crop needs the second stem form");
        AIC_REQUIRE(crop->fh < (1 << 23) && crop->fw * 3 < (1 << 23) && (size_t)crop->fh * crop->fw * 3 < ((size_t)1 << 31), AIC_ERR_INVALID, "fused crop: frame too large for its 24-bit row arithmetic");
        a.frames = crop->frames, a.fh = crop->fh, a.fw = crop->fw, a.boxes = crop->boxes, a.frame_of = crop->frame_of, a.valid = crop->valid;
    }
    AIC_REQUIRE(in_stride == 8 || (in_stride == 4 && reid_stem2_usable(H, W)), AIC_ERR_INVALID, "NHWC4 input needs the second stem form");
    const size_t lds2 = (size_t)(H + 2) * 66 * 8;
    if (reid_stem2_usable(H, W)) {
        static bool attr2 = false;
        if (!attr2) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(reid_stem_pool2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));   // + 8 KB of static tap tables
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


}  // namespace aic
