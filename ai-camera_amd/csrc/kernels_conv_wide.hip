// kernels_conv_wide.hip -- the implicit GEMM for launches of a FEW tiles (fp16, round 5): one frame of YOLOv8n, thirty ReID crops -- the
// per-frame plugin loop (src/aicamera_tracker.py:169-207 calls the engines one frame at a time).
//
// At those sizes a layer is 6 .. 150 blocks on 256 CUs, one block per CU, one wave per SIMD, and its time is its K loop run once:
// conv_igemm_dma_kernel (v2) takes ~0.43 us per 32-deep K-step whatever its instruction count (a lean K-step: -4 %) and whatever its ring
// depth (8 stages: no change) -- every step ends in "my loads have landed" + a block barrier, and with nothing else on the CU that round trip
// is paid in full, 144 times for a ReID layer4 conv (62 us for 0.13 GFLOP).  This kernel keeps v2's tile, LDS row layout, K orders and MFMA
// sequence per accumulator -- bit-identical outputs -- and synchronises once per GROUP of G K-steps: the ring holds NG groups of G
// sub-stages, a group's LDS-DMA is issued NG - 1 groups ahead, one wait + one barrier per group.  With one wave per SIMD every instruction is
// paid for (a 64 x 64 tile's step is four MFMAs: a K walk of ~50 scalar instructions per step ran 0.22 us per step, one of ~110 -- with the
// second source's state machine -- 0.36): the walk is therefore a TABLE, one 16-byte entry per K-step (pixel byte offset, weight byte offset,
// tap-validity bit, source select), built on the host once per layer shape and read with scalar loads a group ahead.  3x3 (pad 1) and 1x1 (pad 0) convs of any
// stride with Cin a multiple of 32 in memory order or the chunk-major orders (ConvArgs::k_order 0 / 1 / 3), with v2's second source
// (ConvArgs::x2: the folded 1x1 downsample of a BasicBlock, its chunk e behind tap (0, 0) of the window's chunk e + 1) and v2's 1x1 tail
// (TAIL: the detect branches' last conv in the epilogue); no split source, no device-side item count, no bias-first (those stay on v2).
#include "conv_common.hpp"

#include <array>
#include <map>
#include <mutex>
#include <vector>

namespace aic {

// s_waitcnt vmcnt(N) alone, any N < 64 (gfx9 encoding: vmcnt in bits 3:0 and 15:14; expcnt 6:4 and lgkmcnt 11:8 left at their maxima)
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70); }

template <int MT, int NT, int WM, int WN, int G, int NG, bool TAIL, bool X2>
__global__ __launch_bounds__(256) void conv_wide_kernel(const ConvArgs a, const uint4* __restrict__ tab, int nsteps) {
    typedef half_t T;
    constexpr int CH = 8, BKE = 32, RP = 64;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + RP - 1) / RP * RP;
    constexpr int A_PER = BM / RP, B_PER = BNP / RP, LPS = A_PER + B_PER;
    constexpr int STAGE = (BM + BNP) * 64;
    static_assert(WM * WN == 4 && BM % RP == 0 && NG >= 2 && NG * G * STAGE <= 160 * 1024 && LPS * G * (NG - 1) < 64, "geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    const int M = a.M;
    int tbx, tby;
    xcd_tile_xy(a.xcd_map, tbx, tby);
    const int m0 = tbx * BM, n0 = tby * BN;

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);
    const int HoWo = a.Ho * a.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;

    // rows of the pixel operand this thread fetches: pointer at tap (0, 0), channel 0 (+ its 16-byte slot), and the taps inside the image
    const char* rowp[A_PER];
    unsigned vmask[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int m = m0 + r0 + RP * i;
        unsigned mk = 0;
        const char* rp = reinterpret_cast<const char*>(zero);
        if (m < M) {
            int img, rem, oh, ow;
            fast_divmod(m, HoWo, inv_howo, img, rem);
            fast_divmod(rem, a.Wo, inv_wo, oh, ow);
            const int ih0 = oh * a.stride - a.pad, iw0 = ow * a.stride - a.pad;
            rp = reinterpret_cast<const char*>(xg + (((long)img * a.H + ih0) * a.W + iw0) * a.x_cs + a.x_coff + kc * CH);
            const int lo_w = max(0, -iw0), hi_w = min(a.KW, a.W - iw0);
            const int lo_h = max(0, -ih0), hi_h = min(a.KH, a.H - ih0);
            if (hi_w > lo_w && hi_h > lo_h) {
                const unsigned vw = ((1u << hi_w) - 1u) & ~((1u << lo_w) - 1u);
                const unsigned rows = (((1u << (hi_h * a.KW)) - 1u) & ~((1u << (lo_h * a.KW)) - 1u)) & a.tap_rows;
                mk = vw * rows;
            }
        }
        rowp[i] = rp;
        vmask[i] = mk | (m < M ? 1u << 16 : 0u);              // bit 16: the row exists (the second source has no window to fall out of)
    }
    const char* wptr[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
        const bool okr = r0 + RP * j < BN && n0 + r0 + RP * j < a.cout_pad;      // rows past the channel tile / the padded weight matrix: any in-bounds row (never read by an MFMA whose result is stored)
        wptr[j] = reinterpret_cast<const char*>(wg + (size_t)(okr ? n0 + r0 + RP * j : 0) * a.Kp + kc * CH);
    }
    char* const sdst = smem + (16 * wv) * 64;
    // second source (ConvArgs::x2): the rows of its strided pixels, channel 0 (+ this thread's 16-byte slot)
    const char* x2row[X2 ? A_PER : 1];
    if constexpr (X2) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int m = m0 + r0 + RP * i;
            const char* rp = reinterpret_cast<const char*>(zero);
            if (m < M) {
                int img, rem, oh, ow;
                fast_divmod(m, HoWo, inv_howo, img, rem);
                fast_divmod(rem, a.Wo, inv_wo, oh, ow);
                rp = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.x2) + (((long)img * a.H2 + oh * a.s2) * a.W2 + ow * a.s2) * a.x2_cs + a.x2_coff + kc * CH);
            }
            x2row[i] = rp;
        }
    }
    // ---- the LDS-DMA stream: K-step s_k's entry of the layer's table = (pixel byte offset from the row's tap (0, 0) pointer, weight byte offset
    // in the row, the bit of `vmask` that says whether this row has the step's tap -- bit 16 for a second-source step --, 1 if second source);
    // entries past the last step have no bit set (zero page) and the first step's weights (in bounds, never multiplied)
    int s_k = 0;
    auto issue = [&](int st, const uint4 e) {
        char* const dst = sdst + st * STAGE;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const char* base = rowp[i];
            if constexpr (X2) base = (e.w & 1u) ? x2row[i] : rowp[i];
            const T* src = (vmask[i] & e.z) ? reinterpret_cast<const T*>(base + (int)e.x) : zero;
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + i * (RP * 64)), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const T* src = reinterpret_cast<const T*>(wptr[j] + (int)e.y);
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + BM * 64 + j * (RP * 64)), 16, 0, 0);
        }
    };
    auto issue_group = [&](int gslot) {
        uint4 e[G];
#pragma unroll
        for (int g = 0; g < G; ++g) e[g] = tab[s_k + g];          // (uniform address: scalar loads)
#pragma unroll
        for (int g = 0; g < G; ++g) issue(gslot * G + g, e[g]);
        s_k += G;
    };

    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;
    int xoff[MT], woff[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xoff[i] = lds_off((wm * MT + i) * 16 + r, q);
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = BM * 64 + lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (a.bias_init) {                         // bias first (ConvArgs::bias_init, k_order 2: the weights-resident kernels' placement); the epilogue's `bias` is a page of zeros
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            floatx4 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = a.bias_init[n0 + wn * NT * 16 + perm_ch<NT>(j, q, e)];
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = b;
        }
    }

    const int ngroups = (nsteps + G - 1) / G;
#pragma unroll
    for (int p = 0; p < NG - 1; ++p) issue_group(p);
    wait_vm<LPS * G * (NG - 2)>();           // group 0 has landed (this wave's part) ...
    __builtin_amdgcn_s_barrier();               // ... and every other wave's
    int gs = 0;                                 // ring slot of the group being multiplied
    for (int S = 0; S < ngroups; ++S) {
        issue_group(gs == 0 ? NG - 1 : gs - 1); // group S + NG - 1 into the slot group S - 1 left (every wave is past the barrier that closed it)
        const char* base = smem + gs * (G * STAGE);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (S * G + g < nsteps) {           // (uniform; a sub-step past the end holds zeros x step 0's weights: not multiplied, -0 stays -0)
                half8 xf[MT], wf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const half8*>(base + g * STAGE + xoff[i]);
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const half8*>(base + g * STAGE + woff[j]);
                mma_tiles<T, MT, NT>(acc, wf, xf);
            }
        }
        wait_vm<LPS * G * (NG - 2)>();       // group S + 1 has landed; the NG - 2 groups behind it may stay in flight
        __builtin_amdgcn_s_barrier();
        gs = gs == NG - 1 ? 0 : gs + 1;
    }
    wait_vmcnt<0>();

    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wm * MT + i) * 16 + r;
        mrow[i] = m < M ? m : -1;
    }
    if constexpr (TAIL) {
        static_assert(WN == 1, "the tail needs a wave that owns every channel of its pixels");
        tail_1x1<MT, NT>(a, acc, mrow, lane);
    } else {
        epilogue_dispatch<T, MT, NT, true, false>(a, acc, mrow, n0 + wn * NT * 16, q);
    }
}

// ---- the K walk of a layer shape as a table (v2's walks, kernels_conv.hip: k_order 0 = tap outer / chunk inner, 2 = tap column / chunk / tap row, 1 = chunk outer / taps in
// order with the second source's chunk e behind tap (0, 0) of chunk e + 1, 3 = chunk outer / taps plane by plane).  One per distinct
// (KH, W, x_cs, Cin, k_order, Cin2), built at first use, kept for the life of the process (a few KB each).
struct KTab { const uint4* dev; int nsteps; };

static KTab ktab_for(const ConvArgs& a) {
    static std::mutex mu;
    static std::map<std::array<int, 6>, KTab> cache;
    const int cin2 = a.x2 ? a.Cin2 : 0;
    const std::array<int, 6> key{a.KH, a.W, a.x_cs, a.Cin, a.k_order, cin2};
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int ntap = a.KH * a.KW, csteps = a.Cin / 32, csteps2 = cin2 / 32;
    const int nsteps = ntap * csteps + csteps2;
    std::vector<uint4> t;
    int ti = 0, cc = 0;
    bool xs = false;
    for (int k = 0; k < nsteps; ++k) {
        if (a.k_order == 2) {                   // (kw, chunk, kh): the weights-resident kernels' order (3x3 only, conv_wide_ok)
            const int kw2 = k / (3 * csteps), c2 = (k / 3) % csteps, kh2 = k % 3, tap2 = kh2 * 3 + kw2;
            t.push_back(uint4{(unsigned)(((kh2 * a.W + kw2) * a.x_cs + c2 * 32) * 2), (unsigned)((tap2 * a.Cin + c2 * 32) * 2), 1u << tap2, 0u});
            continue;
        }
        const int tap = a.k_order == 3 ? (int)((0x453718620ull >> (4 * ti)) & 15) : ti;
        const int kh = ntap == 1 ? 0 : tap / 3, kw = tap - 3 * kh;
        uint4 e;
        if (xs) e = uint4{(unsigned)((cc - 1) * 64), (unsigned)((ntap * a.Cin + (cc - 1) * 32) * 2), 1u << 16, 1u};
        else e = uint4{(unsigned)(((kh * a.W + kw) * a.x_cs + cc * 32) * 2), (unsigned)((tap * a.Cin + cc * 32) * 2), 1u << tap, 0u};
        t.push_back(e);
        if (a.k_order == 0) { if (++cc == csteps) { cc = 0; ++ti; } }
        else if (!xs && ti == 0 && cc >= 1 && cc <= csteps2 && cc < csteps) xs = true;      // the second source's chunk cc - 1 comes next
        else { xs = false; if (++ti == ntap) { ti = 0; ++cc; } }
    }
    for (int k = 0; k < 32; ++k) t.push_back(uint4{0u, 0u, 0u, 0u});      // the groups issued past the end: zero page x the first step's weights
    uint4* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, t.size() * sizeof(uint4)));
    HIP_CHECK(hipMemcpy(d, t.data(), t.size() * sizeof(uint4), hipMemcpyHostToDevice));
    const KTab r{d, nsteps};
    cache.emplace(key, r);
    return r;
}

// The layers this kernel takes; `blocks`: the grid v2 would launch for the same tile.  AICAM_WIDE_BLOCKS: the largest grid (0: off)
static bool conv_wide_ok(const ConvArgs& a, long blocks, bool tail) {
    static const int max_blocks = [] { const char* e = getenv("AICAM_WIDE_BLOCKS"); return e ? atoi(e) : 256; }();
    if (blocks > max_blocks) return false;
    if (a.xs || a.n_dev || (a.w_tail != nullptr) != tail) return false;
    if (a.k_order < 0 || a.k_order > 3) return false;
    if ((a.k_order == 2) != (a.bias_init != nullptr) || (a.k_order == 2 && (a.KH != 3 || a.x2 || tail))) return false;      // order 2 comes with the bias in front
    if (a.x2 && (a.k_order != 1 || tail || a.Cin2 <= 0 || a.Cin2 % 32 || a.Cin2 / 32 >= a.Cin / 32)) return false;
    if (a.KH != a.KW || (a.KH != 1 && a.KH != 3) || a.pad != a.KH / 2 || a.Cin % 32 || a.Kp != a.KH * a.KW * a.Cin + (a.x2 ? a.Cin2 : 0)) return false;
    if (a.KH == 3 && a.tap_rows != 0x49u) return false;
    if ((long)(2 * a.W + 2) * a.x_cs * 2 + a.Cin * 2 >= (1l << 31) || (long)a.Kp * 2 >= (1l << 31)) return false;      // the table's 32-bit byte offsets
    if (a.Kp / 32 < 8) return false;                             // a K loop of a few steps has nothing to group
    return true;
}

template <int MT, int NT, int WM, int WN, int G, int NG, bool TAIL, bool X2>
static void launch_wide(const ConvArgs& a, dim3 grid, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + 63) / 64 * 64;
    constexpr size_t lds = (size_t)NG * G * (BM + BNP) * 64;
    auto kfn = conv_wide_kernel<MT, NT, WM, WN, G, NG, TAIL, X2>;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const KTab t = ktab_for(a);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a, t.dev, t.nsteps);
    KCHECK();
}

template <int MT, int NT, int WM, int WN, bool TAIL>
static bool try_wide(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + 63) / 64 * 64;
    constexpr int STAGE = (BM + BNP) * 64;
    dim3 grid(ceil_div(a.M, BM), ceil_div(a.Cout, BN));
    if (!conv_wide_ok(a, (long)grid.x * grid.y, TAIL)) return false;
    // Ring shape (tools/ab_wide.sh, ReID layer4 at 28 crops, 60.7 us on v2): G = 4 with 3 or 5 groups in the ring 32.3 / 32.8 us, G = 2 with 10
    // groups 34.3 us -- the depth does not matter.  At most 96 .. 120 KB of LDS, so that a block of another stream's kernel still fits beside it.
    constexpr int G = STAGE <= 8 * 1024 ? 4 : 2;
    if constexpr (!TAIL && (BN == 64 || BN == 128) && WN == 2) {          // (the tiles the ReID trunk's second-source layers take at these sizes)
        if (a.x2) { launch_wide<MT, NT, WM, WN, G, 3, false, true>(a, grid, s); return true; }
    }
    if (a.x2) return false;
    launch_wide<MT, NT, WM, WN, G, 3, TAIL, false>(a, grid, s);
    return true;
}

template <int MT, int NT, int WM, int WN> bool conv_try_wide(const ConvArgs& a, hipStream_t s) { return try_wide<MT, NT, WM, WN, false>(a, s); }
template <int MT, int NT> bool conv_try_wide_tail(const ConvArgs& a, hipStream_t s) { return try_wide<MT, NT, 4, 1, true>(a, s); }

// the 4-wave tiles launch_variant() hands out to fp16 layers (kernels_conv.hip)
template bool conv_try_wide<4, 4, 2, 2>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<2, 2, 2, 2>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<2, 5, 4, 1>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<4, 4, 4, 1>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<2, 4, 4, 1>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<2, 3, 4, 1>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<4, 2, 4, 1>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<4, 1, 4, 1>(const ConvArgs&, hipStream_t);
template bool conv_try_wide<2, 9, 4, 1>(const ConvArgs&, hipStream_t);      // the merged first convs of a detect level (144 channels)
template bool conv_try_wide_tail<2, 4>(const ConvArgs&, hipStream_t);       // 128 px x 64 ch + 1x1 tail
template bool conv_try_wide_tail<2, 5>(const ConvArgs&, hipStream_t);       // 128 px x 80 ch + 1x1 tail

}  // namespace aic
