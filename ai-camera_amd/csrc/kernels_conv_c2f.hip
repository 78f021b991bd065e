// kernels_conv_c2f.hip -- a whole C2f block of YOLOv8n's 160 x 160 stage (Ultralytics yolov8.yaml layer 2: c1 = c2 = 32, n = 1,
// shortcut) as ONE kernel: cv1 (1x1, 32 -> 32) -> split -> m.cv1 (3x3, 16 -> 16) -> m.cv2 (3x3, 16 -> 16, + its input) -> concat ->
// cv2 (1x1, 48 -> 32), every conv + folded-BN bias + SiLU, fp16 in / fp32 accumulate / fp16 out.
//
// Why: as four launches the stage moves 17.5 MB per frame through HBM (the 48-channel concat buffer and the 16-channel
// intermediate are written and read back) for 0.44 GFLOP -- 2.2 ms per 512-frame launch group at 40-180 TFLOP/s, four times its
// HBM floor.  Fused, a block reads the 32-channel input of its tile once (with a halo of 2) and writes the 32-channel output once:
// 4.9 MB per frame; the concat buffer and the intermediate never leave the CU.
//
// A block owns 8 x 32 output pixels of one image (4 waves).  LDS: every tensor as PLANES of 8 channels, [plane][pixel] x 16 bytes, so
// the 16 lanes that share a K slice read 16 consecutive pixels = 256 contiguous bytes (pixel-major 64- or 32-byte pixels put
// lanes r and r + 4 / r + 8 on the same banks: 63 % LDS bank-conflict cycles measured, SQ_LDS_BANK_CONFLICT):
//   X   4 planes [12 x 36]  the input tile with halo 2 (zeros outside the image)
//   Y0a 2 planes [ 8 x 32]  cv1's output, pass-through half: only cv2 reads it, and only at the tile's own pixels
//   Y0b 2 planes [12 x 36]  cv1's output, the bottleneck's input, with halo 2; ZERO outside the image (what the 3x3 convs' zero
//                           padding sees -- NOT cv1 applied to padding)
//   T   2 planes [10 x 34]  m.cv1's output with halo 1, zero outside the image           } in X's place: X is dead once cv1 has run
//   Y1  2 planes [ 8 x 32]  m.cv2's output + its input (the shortcut)                     }
// 49.7 KB in all: three blocks per CU (the first form kept all four tensors apart and cv1's pass-through half with its halo, 74.7 KB, two
// blocks per CU -- the kernel waits on its own latencies, SQ_WAIT_ANY 49 %, so residency is what it lacks).
// MFMA conventions are those of conv3x3_c16_kernel (kernels_conv_direct.hip): weights = A operand (lane (r, q): row r, K elements
// 8q..8q+7 of a 32-deep step), pixels = B operand (one aligned 16-byte ds_read per lane and step), a 3x3 over 16 channels =
// 5 steps of two taps each with the bias as accumulator init; the 1x1 convs add the bias after the accumulation like the generic
// epilogue does.  Same K order per output as the unfused kernels.
#include "conv_common.hpp"

namespace aic {

namespace {

constexpr int TH = 8, TW = 32;
constexpr int XR = TH + 4, XC = TW + 4;          // 12 x 36: cv1 region
constexpr int TR = TH + 2, TC = TW + 2;          // 10 x 34: m.cv1 region
// bytes of one 8-channel plane of X / Y0, T, Y1.  Every plane pitch is a multiple of 256 bytes (all 64 banks): a ds_read_b128 lane group
// mixes lanes of two planes (q and q + 1), and with equal bank phases the 16 pixels it covers fall on 16 disjoint bank quads.  T's
// natural pitch (10 x 34 x 16 = 5 440 B = 16 banks off) put the two planes' pixels on top of each other: 2-way conflicts on every
// tap read of m.cv2 (SQ_LDS_BANK_CONFLICT 48 % of the kernel's LDS cycles in round 2).
#ifdef AICAM_C2F_OLD_LAYOUT
constexpr int PX = XR * XC * 16, PT = TR * TC * 16, PY = TH * TW * 16;
#else
constexpr int PX = XR * XC * 16, PT = (TR * TC * 16 + 255) / 256 * 256, PY = TH * TW * 16;
#endif
static_assert(PX % 256 == 0 && PY % 256 == 0, "plane pitch");
// X is written by S1 four planes of a pixel at a time (consecutive lanes = the four 16-byte channel groups of one pixel: a coalesced
// 64-byte global read) -- with equal bank phases that is a 4-way store conflict.  Plane g therefore keeps pixel p in slot
// p ^ 2g (inside its aligned group of 8): a store group (2 pixels x 4 planes) hits 8 distinct slots, and a read of 16 consecutive
// pixels of plane q (S2) still covers the same aligned 16 slots, each once.
__device__ __forceinline__ int x_slot(int p, int g) {
#ifdef AICAM_C2F_OLD_LAYOUT
    return p;
#else
    return (p & ~7) | ((p & 7) ^ (2 * g));
#endif
}
static_assert((XR * XC) % 8 == 0, "swizzle group");
constexpr int LDS_X = 0, LDS_T = 0, LDS_Y1 = LDS_T + 2 * PT, LDS_Y0A = 4 * PX, LDS_Y0B = LDS_Y0A + 2 * PY;
constexpr int LDS_BYTES = LDS_Y0B + 2 * PX;
static_assert(LDS_Y1 + 2 * PY <= 4 * PX, "T and Y1 live in X's place");

}  // namespace

struct C2fArgs {
    const half_t* x; half_t* y;
    const half_t *w1, *w2, *w3, *w4;
    const float *b1, *b2, *b3, *b4;
    int x_cs, x_coff, y_cs, y_coff, H, W, n_img, xcd_map;
};

__global__ __launch_bounds__(256) void c2f16_fused_kernel(const C2fArgs a, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 15, q = lane >> 4;
    int bx = xcd_tile((int)blockIdx.x, (int)gridDim.x, a.xcd_map);
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const half_t* xg = a.x + (size_t)img * a.H * a.W * a.x_cs + a.x_coff;

    // ---- S1: input tile with halo 2 -> LDS (four 16-byte channel groups per pixel)
    {   // (all of a thread's loads in flight before the first store: rolled, the loop was seven dependent HBM round trips per block)
        constexpr int NSLOT = XR * XC * 4, NIT = (NSLOT + 255) / 256;
        uint4 v[NIT];
        int dst[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = t + u * 256;
            const int g = idx & 3, p = idx >> 2;
            const int pr = p / XC, pc = p - pr * XC;
            const int iy = oy0 - 2 + pr, ix = ox0 - 2 + pc;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (idx < NSLOT && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v[u] = *reinterpret_cast<const uint4*>(xg + ((size_t)iy * a.W + ix) * a.x_cs + g * 8);
            dst[u] = idx < NSLOT ? LDS_X + g * PX + x_slot(p, g) * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u)
            if (dst[u] >= 0) *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
    }
    // weights as A fragments
    half8 w1[2], w2[5], w3[5], w4[2][2];
    floatx4 bi1[2], bi2, bi3, bi4[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        w1[ct] = *reinterpret_cast<const half8*>(a.w1 + (size_t)(16 * ct + r) * 32 + 8 * q);
#pragma unroll
        for (int m = 0; m < 2; ++m) w4[ct][m] = *reinterpret_cast<const half8*>(a.w4 + (size_t)perm_row<2>(ct, r) * 64 + 32 * m + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bi1[ct][e] = a.b1[16 * ct + 4 * q + e]; bi4[ct][e] = a.b4[perm_ch<2>(ct, q, e)]; }
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        w2[m] = *reinterpret_cast<const half8*>(a.w2 + (size_t)r * 160 + 32 * m + 8 * q);
        w3[m] = *reinterpret_cast<const half8*>(a.w3 + (size_t)r * 160 + 32 * m + 8 * q);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { bi2[e] = a.b2[4 * q + e]; bi3[e] = a.b3[4 * q + e]; }
    __syncthreads();

    // ---- S2: cv1.  Channels 16-31 (the bottleneck's input) on the 12 x 36 region (27 tiles of 16 pixels): Y0b = SiLU(W1 x + b1), zero
    // outside the image; channels 0-15 (pass-through) on the tile's own 8 x 32 pixels only
    for (int tile = wv; tile < (XR * XC) / 16; tile += 4) {
        const int p = tile * 16 + r;
        const half8 xb = *reinterpret_cast<const half8*>(smem + LDS_X + q * PX + x_slot(p, q) * 16);
        const int pr = p / XC, pc = p - pr * XC;
        const bool inside = (unsigned)(oy0 - 2 + pr) < (unsigned)a.H && (unsigned)(ox0 - 2 + pc) < (unsigned)a.W;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[1], xb, acc, 0, 0, 0);
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = inside ? (half_t)act_fast<1>(acc[e] + bi1[1][e]) : (half_t)0.f;
        *reinterpret_cast<half4*>(smem + LDS_Y0B + (q >> 1) * PX + p * 16 + (q & 1) * 8) = o;
    }
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int oyl = 2 * wv + (tile >> 1), oxl = (tile & 1) * 16 + r;
        const int p = (oyl + 2) * XC + oxl + 2;
        const half8 xb = *reinterpret_cast<const half8*>(smem + LDS_X + q * PX + x_slot(p, q) * 16);
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[0], xb, acc, 0, 0, 0);
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)act_fast<1>(acc[e] + bi1[0][e]);
        *reinterpret_cast<half4*>(smem + LDS_Y0A + (q >> 1) * PY + (oyl * TW + oxl) * 16 + (q & 1) * 8) = o;
    }
    __syncthreads();

    // ---- S3: m.cv1 (3x3 over Y0's channels 16-31) on the 10 x 34 region: T = SiLU(W2 * Y0b + b2), zero outside the image
    {
        int d[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const int tap = min(2 * m + (q >> 1), 8), kh = tap / 3, kw = tap - 3 * kh;
            d[m] = (q & 1) * PX + (kh * XC + kw) * 16;
        }
        for (int tile = wv; tile < (TR * TC + 15) / 16; tile += 4) {
            const int p = min(tile * 16 + r, TR * TC - 1);
            const int pr = p / TC, pc = p - pr * TC;
            const int base = LDS_Y0B + (pr * XC + pc) * 16;         // top-left tap of this output pixel
            floatx4 acc = bi2;
#pragma unroll
            for (int m = 0; m < 5; ++m)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[m], *reinterpret_cast<const half8*>(smem + base + d[m]), acc, 0, 0, 0);
            const bool inside = (unsigned)(oy0 - 1 + pr) < (unsigned)a.H && (unsigned)(ox0 - 1 + pc) < (unsigned)a.W;
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = inside ? (half_t)act_fast<1>(acc[e]) : (half_t)0.f;
            if (tile * 16 + r < TR * TC) *reinterpret_cast<half4*>(smem + LDS_T + (q >> 1) * PT + p * 16 + (q & 1) * 8) = o;
        }
    }
    __syncthreads();

    // ---- S4: m.cv2 (3x3 over T) on the 8 x 32 outputs, + the bottleneck's input (shortcut): Y1 = SiLU(W3 * T + b3) + Y0b
    {
        int d[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const int tap = min(2 * m + (q >> 1), 8), kh = tap / 3, kw = tap - 3 * kh;
            d[m] = (q & 1) * PT + (kh * TC + kw) * 16;
        }
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const int oyl = 2 * wv + (tile >> 1), oxl = (tile & 1) * 16 + r;
            const int base = LDS_T + (oyl * TC + oxl) * 16;
            floatx4 acc = bi3;
#pragma unroll
            for (int m = 0; m < 5; ++m)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[m], *reinterpret_cast<const half8*>(smem + base + d[m]), acc, 0, 0, 0);
            const half4 res = *reinterpret_cast<const half4*>(smem + LDS_Y0B + (q >> 1) * PX + ((oyl + 2) * XC + oxl + 2) * 16 + (q & 1) * 8);
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)(act_fast<1>(acc[e]) + (float)res[e]);
            *reinterpret_cast<half4*>(smem + LDS_Y1 + (q >> 1) * PY + (oyl * TW + oxl) * 16 + (q & 1) * 8) = o;
        }
    }
    __syncthreads();

    // ---- S5: cv2 (1x1 over the 48 concatenated channels) -> global: out = SiLU(W4 [Y0a, Y0b, Y1] + b4)
    half_t* yg = a.y + (size_t)img * a.H * a.W * a.y_cs + a.y_coff;
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int oyl = 2 * wv + (tile >> 1), oxl = (tile & 1) * 16 + r;
        const int a0 = q < 2 ? LDS_Y0A + q * PY + (oyl * TW + oxl) * 16 : LDS_Y0B + (q - 2) * PX + ((oyl + 2) * XC + oxl + 2) * 16;
        const half8 x0 = *reinterpret_cast<const half8*>(smem + a0);
        const half8 x1 = *reinterpret_cast<const half8*>(smem + LDS_Y1 + (q & 1) * PY + (oyl * TW + oxl) * 16);   // k 32..47; k 48..63 meet zero weights
        float v[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w4[ct][0], x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w4[ct][1], x1, acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[ct][e] = act_fast<1>(acc[e] + bi4[ct][e]);
        }
        const half8 o = {(half_t)v[0][0], (half_t)v[0][1], (half_t)v[0][2], (half_t)v[0][3],
                         (half_t)v[1][0], (half_t)v[1][1], (half_t)v[1][2], (half_t)v[1][3]};
        *reinterpret_cast<half8*>(yg + ((size_t)(oy0 + oyl) * a.W + ox0 + oxl) * a.y_cs + 8 * q) = o;   // perm_ch<2>: lane owns channels 8q..8q+7
    }
}

// cv1 -> m.cv1 -> m.cv2 (+ shortcut) -> cv2 of a C2f with 16-channel halves, when the four convs are wired as engine_file.py's c2f()
// wires them; false = pattern / geometry not supported, nothing launched.
bool conv_try_c2f16(const ConvArgs& c1, const ConvArgs& m1, const ConvArgs& m2, const ConvArgs& c2, hipStream_t s) {
    static const bool off = getenv("AICAM_NO_C2F") != nullptr;
    if (off) return false;
    auto one = [](const ConvArgs& c, int cin, int cout, int kp) {
        return c.KH == 1 && c.KW == 1 && c.stride == 1 && c.pad == 0 && c.Cin == cin && c.Cout == cout && c.Kp == kp && c.act == 1 &&
               c.res_mode == 0 && !c.out_f32;
    };
    auto three = [](const ConvArgs& c) {
        return c.KH == 3 && c.KW == 3 && c.stride == 1 && c.pad == 1 && c.Cin == 16 && c.Cout == 16 && c.Kp == 160 && c.act == 1 && !c.out_f32;
    };
    if (!one(c1, 32, 32, 32) || !one(c2, 48, 32, 64) || !three(m1) || !three(m2) || m1.res_mode != 0 || m2.res_mode != 2) return false;
    const int H = c1.H, W = c1.W;
    for (const ConvArgs* c : {&c1, &m1, &m2, &c2})
        if (c->H != H || c->W != W || c->Ho != H || c->Wo != W || c->M != c1.M) return false;
    if (H % TH || W % TW) return false;
    // wiring: cv1 writes cat[0:32]; m.cv1 reads cat[16:32] -> tmp; m.cv2 reads tmp, adds cat[16:32], writes cat[32:48]; cv2 reads cat[0:48]
    const void* cat = c1.y;
    if (c1.y_coff != 0 || m1.x != cat || m1.x_coff != 16 || m1.x_cs != c1.y_cs || m2.x != m1.y || m2.x_coff != m1.y_coff || m2.x_cs != m1.y_cs ||
        m2.y != cat || m2.y_coff != 32 || m2.res != cat || m2.r_coff != 16 || m2.r_cs != c1.y_cs || c2.x != cat || c2.x_coff != 0 ||
        c2.x_cs != c1.y_cs || c1.y_cs < 48)
        return false;
    if ((c1.x_cs | c1.x_coff | c2.y_cs | c2.y_coff) % 8) return false;
    C2fArgs a{};
    a.x = reinterpret_cast<const half_t*>(c1.x), a.y = reinterpret_cast<half_t*>(c2.y);
    a.w1 = reinterpret_cast<const half_t*>(c1.w), a.w2 = reinterpret_cast<const half_t*>(m1.w);
    a.w3 = reinterpret_cast<const half_t*>(m2.w), a.w4 = reinterpret_cast<const half_t*>(c2.w);
    a.b1 = c1.bias, a.b2 = m1.bias, a.b3 = m2.bias, a.b4 = c2.bias;
    a.x_cs = c1.x_cs, a.x_coff = c1.x_coff, a.y_cs = c2.y_cs, a.y_coff = c2.y_coff, a.H = H, a.W = W;
    a.n_img = c1.M / (H * W), a.xcd_map = xcd_map_on();
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(c2f16_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr = true;
    }
    const int tiles_x = W / TW, tiles_y = H / TH;
    hipLaunchKernelGGL(c2f16_fused_kernel, dim3(a.n_img * tiles_x * tiles_y), dim3(256), (size_t)LDS_BYTES, s, a, tiles_x, tiles_y);
    KCHECK();
    return true;
}

}  // namespace aic
