// kernels_conv_pp.hip -- the one-block-per-CU "ping-pong" members of the conv class (v4 im2col, v5 patch): 8 waves,
// every K-step split into a load and a compute segment, waves 4..7 one barrier behind waves 0..3.
#include "conv_common.hpp"

namespace aic {

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
__device__ unsigned long long g_pp_times[8 * 4096];     // [0, 4*4096): 100 MHz wall clock; [4*4096, 8*4096): shader clock (s_memtime)
__device__ int g_pp_times_on;
#define PP_STAMP(k) do { if (g_pp_times_on && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) { g_pp_times[4 * blockIdx.x + (k)] = wall_clock64(); g_pp_times[4 * 4096 + 4 * blockIdx.x + (k)] = clock64(); } } while (0)

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
    const int M = a.n_dev ? min(a.M, min(a.n_dev[0], a.M / (a.Ho * a.Wo)) * (a.Ho * a.Wo)) : a.M;     // device-side item count: the grid was sized for a bound
    int tbx, tby;
    if (!xcd_tile_xy_live(a.xcd_map, (M + BM - 1) / BM, tbx, tby)) return;
    const int m0 = tbx * BM;
    const int n0 = tby * BN;
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
        if (m < M) {
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
    int tap = 0, kh = 0, kw = 0, cc = 0, ti3 = 0;
    const int kord = a.k_order;                 // != 0: K-steps in another order than memory's, source pointers rebuilt every step
    const bool cmaj = kord != 0;
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
    auto issue = [&](int stage) {
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
            } else {                            // (kw, cc, kh)
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
                // ---- LOAD segment: fragment reads first, LDS-DMA issue behind them (see conv3x3_pp_patch_kernel)
                const char* base = smem + u * STAGE;
                frag_t xf[MT], wf[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(base + woff[j]);
#pragma unroll
                for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(base + xoff[i]);
#ifndef AICAM_PP_READS_LAST
                __builtin_amdgcn_sched_barrier(0);
#endif
                issue((u + NSTAGE - 2) % NSTAGE);
                wait_vmcnt<(NSTAGE - 3) * LPS>();      // step+1 has landed (this wave's part)
                __builtin_amdgcn_s_barrier();
                // ---- COMPUTE segment
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mma_tiles<T, MT, NT>(acc, wf, xf);
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
        mrow[i] = m < M ? m : -1;
    }
    epilogue_dispatch<T, MT, NT, true, true>(a, acc, mrow, n0 + wn * NT * 16, q);
    if (g_pp_times_on) { wait_vmcnt<0>(); PP_STAMP(3); }
}

static void pp_times_report(hipStream_t s, int nblk) {
    static std::vector<unsigned long long> h(8 * 4096);
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pp_times), sizeof(unsigned long long) * 8 * 4096));
    nblk = std::min(nblk, 4096);
    double cyc = 0, us = 0;                    // shader cycles / wall time of the K loops: the clock the chip holds inside them
    for (int b = 0; b < nblk; ++b) {
        cyc += (double)(h[4 * 4096 + 4 * b + 2] - h[4 * 4096 + 4 * b + 1]);
        us += (double)(h[4 * b + 2] - h[4 * b + 1]) * 0.01;
    }
    fprintf(stderr, "[pp_times] k-loop: %.0f shader cycles per block, %.3f GHz\n", cyc / nblk, cyc / us * 1e-3);
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
// Maps narrower than 16 pixels: a 16-pixel MFMA tile used to be 16 / TW consecutive ROWS of one image, whose 16-byte chunks
// sit PW * 16 bytes apart in a plane -- bank conflicts on 40 % of the LDS cycles of ReID layers 3/4 (SQ_LDS_BANK_CONFLICT).
// Now it is ONE row of G = 16 / TW consecutive IMAGES of the tile, and the image pitch is padded so that the G row pieces fall
// into disjoint bank ranges (pitch * 16 B = TW * 16 B mod 256).
// (ppp_ipix_pad, lane_here: conv_common.hpp -- shared with the software-pipelined form, kernels_conv_sp.hip)
template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE, bool X2 = false>
__global__ __launch_bounds__(512) void conv3x3_pp_patch_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ny, int run, int pf) {
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int BKE = 4 * CH;
    constexpr int RP = 128;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int BNP = (BN + RP - 1) / RP * RP;
    constexpr int B_PER = BNP / RP;
    constexpr int LPS = B_PER + 1;
    constexpr int TPIX = TH * TW, NI = BM / TPIX;
    constexpr int PW = TW + 2, PH = TH + 2, IPIX = PW * PH, IPIXP = ppp_ipix_pad(TH, TW), NPIX = NI * IPIXP;
    constexpr int G = TW >= 16 ? 1 : 16 / TW;                          // images per 16-pixel MFMA tile
    constexpr int NPASS = (NPIX + 127) / 128, NPIXP = NPASS * 128;
    constexpr int PLANE = NPIXP * 16, PBUF = 4 * PLANE, DUMMY = 8192, WSTAGE = BNP * 64;
    constexpr int RING = 2 * PBUF + DUMMY;
    constexpr int EOFF = RING + NSTAGE * WSTAGE, NE = BM / 128;        // second source: buffer offset, LDS-DMA passes per chunk
    static_assert(WM * WN == 8 && BM % TPIX == 0 && NPASS <= 11 - NSTAGE && TW % 4 == 0 && NSTAGE >= 4, "geometry");
    static_assert(!X2 || (NSTAGE == 4 && NPASS <= 5 && (NE == 2 || NE == 4) && MT % 2 == 0), "second source: slots at taps 6..8, counted waits of the 4-stage ring");
    // LEAN (4-stage ring, no second source): a LOAD segment whose tap has no patch pass due issues its weight loads only.  The form before
    // kept every segment at B_PER + 1 loads with a zero-page load into a dummy slot, so that ONE counted wait fitted all taps; an LDS-DMA
    // costs 60-185 cycles to issue, 9 - NPASS of a chunk's nine were stand-ins, and a wave's LOAD and COMPUTE segments are serial.  The
    // wait of a segment still lets exactly its OWN loads stay in flight -- B_PER or B_PER + 1, a compile-time property of the tap.
    // Measured: nothing (layer2 / 3 / 4 shapes against the build before both changes, same box each: +2.4 / +1.8 / +1.8 % with the cheaper
    // integer work of issue_patch AND this, +2.5 / +1.6 / +1.7 % with the integer work alone): a zero-page LDS-DMA is cheap to issue after
    // all.  Kept as a build switch (-DAICAM_PPP_LEAN), off.
#if defined(AICAM_PPP_LEAN) && !defined(AICAM_PPP_PF_BUILD)
    constexpr bool LEAN = !X2 && NSTAGE == 4;
#else
    constexpr bool LEAN = false;
#endif

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool late = wv >= 4;
    PP_STAMP(0);
    const int n_img = a.n_dev ? min(a.M / (a.Ho * a.Wo), a.n_dev[0]) : a.M / (a.Ho * a.Wo);     // device-side item count: the grid was sized for a bound
    // A block takes a RUN of `run` consecutive tiles (channel tile fastest, then the pixel tiles of an image, then images) and
    // carries the pipeline across them: the weights of the next tile's first steps and its first patch chunk are in flight while the
    // last chunk of this one is computed, so only the first tile of a run pays the 3-4.5 us prologue (and the block launch) that every
    // tile used to pay -- a sixth of a layer2 tile's time (AICAM_PP_TIMES).  Runs, not tiles, are dealt to the XCDs in contiguous
    // stretches; the live runs (device-side count) are re-dealt over the first blocks of the grid.
    const int ntiles = ((n_img + NI - 1) / NI) * tiles_x * tiles_y * ny;
    const int nruns = (ntiles + run - 1) / run;
    if ((int)blockIdx.x >= nruns) return;
    int t_cur = xcd_tile((int)blockIdx.x, nruns, a.xcd_map) * run;
    const int t_end = min(t_cur + run, ntiles);
    int img0, oy0, ox0, n0;                     // the tile being computed (block-uniform)
    auto coords = [&](int tt, int& im, int& oy, int& ox, int& nn) {
        const int nt = tt % ny; int m = tt / ny;
        const int tx = m % tiles_x; m /= tiles_x;
        const int ty = m % tiles_y;
        im = (m / tiles_y) * NI, oy = ty * TH, ox = tx * TW, nn = nt * BN;
    };
    coords(t_cur, img0, oy0, ox0, n0);

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    // ---- patch passes: pass i, wave w -> plane w&3, pixels i*128 + (w>>2)*64 + lane
    const int plane = wv & 3;
    // The source element of pass i is worked out when the pass is issued (some twenty VALU instructions in a LOAD segment that has
    // room for them) instead of living in NPASS registers: with the tile loop around it the kernel has no register to spare, and every
    // spilled value reloaded after an epilogue is an s_waitcnt vmcnt(0) -- 8 us per tile on the 512 x 128 tile.
    char* const pdst = smem + plane * PLANE + (wv >> 2) * 1024;     // + buffer*PBUF + pass*2048 (+ lane*16 by the DMA)
    // (integer work of a pass, per thread: the two divisions by compile-time constants as ONE 24-bit multiply + shift each -- exact for
    //  p < 1024 and divisors >= 4, checked below -- and the three multiplications of the address as 24-bit ones: a 32-bit v_mul_lo /
    //  v_mul_hi / v_mad_u64 is a quarter-rate instruction, there were eleven of them per pass, and a LOAD segment shares its SIMD's VALU
    //  port with the other half's MFMAs.  launch_pp_patch admits only tensors whose pixel count and strides fit 24 bits.)
    static_assert(NPIXP <= 1024 && PW >= 4 && IPIXP >= 4, "udiv24 range");
    auto udiv24 = [](int x, auto dc) { constexpr int D = decltype(dc)::value; constexpr unsigned M = (1u << 24) / D + 1u; return (int)(__umul24((unsigned)x, M) >> 24); };
    auto issue_patch = [&](int i, int buf, int chunk_off, int im0, int oy, int ox) {        // i: compile-time pass index; tile (im0, oy, ox)
        const int p = i * 128 + (wv >> 2) * 64 + lane_here();
        int il = 0, rem = p;
        if constexpr (NI > 1) { il = udiv24(p, std::integral_constant<int, IPIXP>{}); rem = p - __mul24(il, IPIXP); }
        const int py = udiv24(rem, std::integral_constant<int, PW>{}), px = rem - __mul24(py, PW);
        const int img = im0 + il, iy = oy + py - 1, ix = ox + px - 1;
        const bool ok = p < NPIX && rem < IPIX && img < n_img && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const T* src = ok ? xg + (__mul24(__mul24(__mul24(img, a.H) + iy, a.W) + ix, a.x_cs) + a.x_coff + plane * CH + chunk_off) : zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(pdst + buf * PBUF + i * 2048), 16, 0, 0);
    };
    auto issue_dummy = [&] {
        const T* z = zero;
        asm volatile("" : "+s"(z));                 // the address is copied from its scalar registers here: a vector copy carried through the loop was spilled
        const T* src = z;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + 2 * PBUF + wv * 1024), 16, 0, 0);
    };
    // A residual tile is 128 KB that the epilogue pulls from HBM with nothing left to hide it behind.  Optional (pf != 0, see launch_pp_patch:
    // measured, no net gain, off): the spare LDS-DMA slots of the tile's LAST chunk (bit `tap` of pf) carry it towards the L2 a few K-steps
    // ahead: pass k, thread t touches 128-byte line k * 512 + t of the tile's rows, the 16 bytes land in the dummy slot and are never read.
    // Same instruction count per segment: the counted waits do not change.
    constexpr int LPP = BN * (int)sizeof(T) / 128 > 0 ? BN * (int)sizeof(T) / 128 : 1, NRP = BM * LPP / 512;
    auto issue_res = [&](int k) {
        const int L = k * 512 + wv * 64 + lane_here();
        const int m = L / LPP, part = L - m * LPP;
        int il, ly, lx;
        if constexpr (G == 1) {
            il = m / TPIX;
            const int rem = m - il * TPIX;
            ly = rem / TW, lx = rem - ly * TW;
        } else {
            const int tl = m >> 4, rr = m & 15;
            il = (tl / TH) * G + rr / TW, ly = tl % TH, lx = rr % TW;
        }
        const int img = img0 + il;
        const T* src = (k < NRP && m < BM && img < n_img) ? reinterpret_cast<const T*>(a.res) + (((img * a.Ho + oy0 + ly) * a.Wo + ox0 + lx) * a.r_cs + a.r_coff + n0 + part * (128 / (int)sizeof(T)))
                                                          : zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + 2 * PBUF + wv * 1024), 16, 0, 0);
    };

    // ---- weight stream: row r0 + RP*j of the tile's channel tile, K offset of step (chunk c, tap) = tap*Cin + c*BKE.  The stream runs
    // NSTAGE - 2 steps ahead of the compute and simply continues with the next tile of the run (n0_w: the channel tile it fetches for)
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    int wofs[B_PER];                          // element offset of this thread's weight chunk at K = 0, or -1 (zero page)
    auto set_wofs = [&](int nn) {
#pragma unroll
        for (int j = 0; j < B_PER; ++j) wofs[j] = (r0 + RP * j < BN) ? (nn + r0 + RP * j) * a.Kp + kc * CH : -1;
    };
    set_wofs(n0);
    const int nchunks = a.Cin / BKE;
    const int ns2 = X2 ? a.Cin2 / BKE : 0;           // chunks of the second source (< nchunks)
    const int nsteps = 9 * nchunks + ns2;
    char* const wdst = smem + RING + (16 * wv) * 64;
    int is_c = 0, is_tap = 0, is_k = 0, is_st = 0;   // the step whose weights are fetched next (and its ring stage)
    bool is_x = false;                               // ... is the second source's chunk is_c - 1
    int n0_next = n0;                               // channel tile of the run's next tile (== n0 when there is none)
    bool has_next = false;
    auto issue_w = [&] {
        const int koff = (X2 && is_x) ? 9 * a.Cin + (is_c - 1) * BKE : is_tap * a.Cin + is_c * BKE;
        char* sbase = wdst + is_st * WSTAGE;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            // nothing to fetch (a row past the tile, a step past the end): a weight row past Cout -- rows Cout .. cout_pad - 1 of the packed
            // weights are zeros -- so that the select is one 32-bit offset and no vector copy of the zero page's address lives in the loop
            const T* src = wg + ((wofs[j] >= 0 && is_k < nsteps) ? wofs[j] + koff : a.Cout * a.Kp);
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + j * (RP * 64)), 16, 0, 0);
        }
        ++is_k;
        if (++is_st == NSTAGE) is_st = 0;
        if (X2 && !is_x && is_tap == 0 && is_c >= 1 && is_c <= ns2) is_x = true;
        else {
            is_x = false;
            if (++is_tap == 9) { is_tap = 0; ++is_c; }
        }
        if (is_k == nsteps && has_next) {            // the stream moves on to the next tile of the run
            is_k = 0, is_c = 0, is_tap = 0;
            if (ny > 1) set_wofs(n0_next);
        }
    };

    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;
    // LDS byte address of (this lane's pixel of tile 0, tap (0,0)) in buffer 0; tile i sits a compile-time distance away
    // because a wave's MT*16 pixels either tile whole images or lie inside one (static_assert below)
    static_assert((MT * 16) % TPIX == 0 || TPIX % (MT * 16) == 0, "a wave's pixels must not straddle images irregularly");
    static_assert(TW % 16 == 0 || 16 % TW == 0, "a 16-pixel MFMA tile is whole rows or a piece of one row");
    static_assert(G == 1 || (NI % G == 0 && (MT % TH == 0 || TH % MT == 0)), "image groups");
    // pixel (patch units) of 16-pixel tile number t of the block, and of a wave's tile i relative to its tile 0
    auto tile_pix = [](int t) constexpr { return G == 1 ? (t * 16 / TPIX) * IPIXP + ((t * 16 % TPIX) / TW) * PW + (t * 16 % TPIX) % TW
                                                        : (t / TH) * G * IPIXP + (t % TH) * PW; };
    auto patch_pix = [tile_pix](int m) constexpr { return G == 1 ? tile_pix(m / 16) : (TH % MT == 0 ? (m / 16) * PW : tile_pix(m / 16)); };
    int xa0;
    if constexpr (G == 1) {
        const int ml = wm * MT * 16 + r;
        const int il = ml / TPIX, rem = ml - il * TPIX;
        const int ly = rem / TW, lx = rem - ly * TW;
        xa0 = q * PLANE + (il * IPIXP + ly * PW + lx) * 16;
    } else {
        const int t0 = wm * MT;
        xa0 = q * PLANE + ((t0 / TH) * G * IPIXP + (t0 % TH) * PW + (r / TW) * IPIXP + r % TW) * 16;
    }
    // weight fragment addresses: tiles j and j+2 are 32 rows (2048 B) apart, j and j+1 differ in the swizzle term
    int woff2[2];
#pragma unroll
    for (int j = 0; j < 2 && j < NT; ++j) woff2[j] = RING + lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);
    static_assert(NT % 2 == 0, "tile pairs");

    floatx4 acc[MT][NT];
    typedef typename Frag<T>::type frag_t;

    // ---- prologue (first tile of the run only): patch chunk 0, weights of steps 0 and 1
#pragma unroll
    for (int i = 0; i < NPASS; ++i) issue_patch(i, 0, 0, img0, oy0, ox0);
    has_next = !X2 && t_cur + 1 < t_end;        // (second-source kernels: one tile per block -- they have no register for the run's state)
    if (has_next) { int i0, y0, x0; coords(t_cur + 1, i0, y0, x0, n0_next); }
#pragma unroll
    for (int st = 0; st < NSTAGE - 2; ++st) {
        issue_w();
        if (st && !LEAN) issue_dummy();        // (not LEAN) every set in flight has LPS loads: the counted waits below rely on it
    }
    if constexpr (LEAN) wait_vmcnt<B_PER>();   // patch chunk 0 and the weights of step 0 have landed (this wave's part): only step 1's weights may be in flight
    else wait_vmcnt<(NSTAGE - 3) * LPS>();
    __builtin_amdgcn_s_barrier();
    PP_STAMP(1);

    int rd_st = 0;                             // ring stage of the step being computed
    int img_n = img0, oy_n = oy0, ox_n = ox0;   // the run's next tile
    // second source: pass p of chunk e -> rows p * 128 + t / 4 of the tile (rows in the order of mrow below), K-chunk kc of the row's 64 bytes
    auto issue_e = [&](int p, int e, bool on) {
        const int ln = lane_here(), rl = 16 * wv + (ln >> 2), kcl = (ln & 3) ^ lds_swz(rl);      // r0 and kc of this thread, worked out here
        const int m = p * 128 + rl;
        int il, ly, lx;
        if constexpr (G == 1) {
            il = m / TPIX;
            const int rem = m - il * TPIX;
            ly = rem / TW, lx = rem - ly * TW;
        } else {
            const int tl = m >> 4, rr = m & 15;
            il = (tl / TH) * G + rr / TW, ly = tl % TH, lx = rr % TW;
        }
        const int img = img0 + il;
        // (24-bit multiplies, as in issue_patch: row of the second source < 2^23, element offset < 2^31 -- launch_pp_patch checks both;
        //  the 64-bit form was three quarter-rate multiplies per thread and pass)
        const int row2 = __mul24(img, a.H2) + __mul24(oy0 + ly, a.s2);
        const T* src = (on && img < n_img) ? reinterpret_cast<const T*>(a.x2) + (__mul24(row2, a.W2 * a.x2_cs) + __mul24(__mul24(ox0 + lx, a.s2), a.x2_cs) +
                                                                                 a.x2_coff + e * BKE + kcl * CH)
                                           : zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + EOFF + p * 8192 + (16 * wv) * 64), 16, 0, 0);
    };
    auto xstep = [&](int) {                    // the extra step: weights from the ring as ever, pixels from the second source's buffer
        const int so = rd_st * WSTAGE;
        if (++rd_st == NSTAGE) rd_st = 0;
        frag_t xf[MT], wf[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(smem + woff2[j & 1] + so + (j >> 1) * 2048);
        const int ln = lane_here(), rr = ln & 15, qq = ln >> 4;     // (the two addresses are worked out here, not carried -- and spilled -- through the loop)
        const int eb = EOFF + (wm * MT * 16 + rr) * 64, s0 = lds_swz(rr), e0 = eb + 16 * (qq ^ s0), e1 = eb + 16 * (qq ^ s0 ^ 2);   // rows 16 apart: swizzle bit 1 flips
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(smem + ((i & 1) ? e1 : e0) + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        issue_w();
        issue_dummy();
        wait_vmcnt<(NSTAGE - 3) * LPS>();
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma_tiles<T, MT, NT>(acc, wf, xf);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    auto chunk = [&](int c, auto bufc) {       // bufc: compile-time parity of the patch buffer read in this chunk
        constexpr int BUF = decltype(bufc)::value;
        // the patch that streams in during this chunk: the next channel chunk of this tile, or -- in the last chunk -- chunk 0 of the
        // run's next tile (nchunks is even: that is buffer 0, where every tile starts)
        const bool inner = c + 1 < nchunks;
        const bool more = inner || has_next;
        const int noff = inner ? (c + 1) * BKE : 0;
        const int p_im = inner ? img0 : img_n, p_oy = inner ? oy0 : oy_n, p_ox = inner ? ox0 : ox_n;
        static_for<9>([&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
            // ---- LOAD segment: the fragment reads, then the LDS-DMA issue (an LDS-DMA costs 60-185 cycles to issue)
            constexpr bool EP = X2 && tap >= 6 && tap >= 9 - NE + (NE == 4 ? 1 : 0);      // this tap's spare slot carries a pass of the second source
            const int so = rd_st * WSTAGE;
            if (++rd_st == NSTAGE) rd_st = 0;
            const int tapoff = BUF * PBUF + ((tap / 3) * PW + tap % 3) * 16;
            frag_t xf[MT], wf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag_t*>(smem + woff2[j & 1] + so + (j >> 1) * 2048);
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const frag_t*>(smem + xa0 + tapoff + patch_pix(16 * i) * 16);
            __builtin_amdgcn_sched_barrier(0);
            issue_w();
            constexpr bool SLOT = tap >= 1 && tap <= NPASS;          // this tap's segment carries a patch pass
            if (SLOT && more) issue_patch(SLOT ? tap - 1 : 0, BUF ^ 1, noff, p_im, p_oy, p_ox);
            else if constexpr (!EP && (SLOT || !LEAN)) {              // LEAN: a tap without a pass issues NO stand-in (its wait is one lower)
#ifdef AICAM_PPP_PF_BUILD                               // (the residual prefetch experiment: measured, no net gain; not in the default build's K loop)
                if (!X2 && !inner && ((pf >> tap) & 1)) issue_res(__builtin_popcount(pf & ((1 << tap) - 1)));
                else
#endif
                issue_dummy();
            }
            if constexpr (EP) {
                if constexpr (NE == 4) {
                    if (tap == 6) { issue_e(0, c, c < ns2); issue_e(1, c, c < ns2); } else issue_e(tap - 5, c, c < ns2);
                } else issue_e(tap - 7, c, c < ns2);
            }
            if constexpr (LEAN) wait_vmcnt<B_PER + (SLOT ? 1 : 0)>();         // everything older than THIS segment's own loads has landed
            else if constexpr (X2 && NE == 4 && tap == 6) wait_vmcnt<(NSTAGE - 3) * LPS + 1>();
            else wait_vmcnt<(NSTAGE - 3) * LPS>();
            __builtin_amdgcn_s_barrier();
            // ---- COMPUTE segment
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma_tiles<T, MT, NT>(acc, wf, xf);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if constexpr (X2 && tap == 0) {
                if (c >= 1 && c <= ns2) xstep(0);
            }
        });
    };
    for (;;) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (has_next) coords(t_cur + 1, img_n, oy_n, ox_n, n0_next);
        if (late) __builtin_amdgcn_s_barrier();    // waves 4..7 run one segment behind
        for (int c = 0; c < nchunks; c += 2) {     // Cin / BKE is even for every layer that reaches this kernel
            chunk(c, std::integral_constant<int, 0>{});
            chunk(c + 1, std::integral_constant<int, 1>{});
        }
        if (!late) __builtin_amdgcn_s_barrier();   // every wave has executed the same number of barriers: both halves store together
        if (!has_next) wait_vmcnt<0>();
        if (!has_next) PP_STAMP(2);

        int mrow[MT];
        const int ln = lane_here(), rr = ln & 15, qe = ln >> 4;    // (hoisted out of the tile loop, the lane parts of the eight rows are 24 registers the
                                                                    //  kernel does not have -- they were spilled and reloaded one wait at a time)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int il, ly, lx;
            if constexpr (G == 1) {
                const int ml = (wm * MT + i) * 16 + rr;
                il = ml / TPIX;
                const int rem = ml - il * TPIX;
                ly = rem / TW, lx = rem - ly * TW;
            } else {
                const int tl = wm * MT + i;
                il = (tl / TH) * G + rr / TW, ly = tl % TH, lx = rr % TW;
            }
            const int img = img0 + il;
            mrow[i] = img < n_img ? (img * a.Ho + oy0 + ly) * a.Wo + ox0 + lx : -1;
        }
        epilogue_dispatch<T, MT, NT, true, true>(a, acc, mrow, n0 + wn * NT * 16, qe);
        if (!has_next) break;
        ++t_cur;
        img0 = img_n, oy0 = oy_n, ox0 = ox_n, n0 = n0_next;
        has_next = !X2 && t_cur + 1 < t_end;
    }
    if (g_pp_times_on) { wait_vmcnt<0>(); PP_STAMP(3); }
}

template <typename T, int MT, int NT, int WM, int WN, int TH, int TW, int NSTAGE, bool X2 = false>
static bool launch_pp_patch(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, BNP = (BN + 127) / 128 * 128;
    constexpr int NI = BM / (TH * TW), NPIX = NI * ppp_ipix_pad(TH, TW), NPASS = (NPIX + 127) / 128;
    constexpr size_t lds = (size_t)2 * 4 * NPASS * 128 * 16 + 8192 + (size_t)NSTAGE * BNP * 64 + (X2 ? BM * 64 : 0);
    static_assert(lds <= 160 * 1024, "does not fit the LDS");
    if (a.H % TH || a.W % TW || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.M >= (1 << 23) || a.x_cs >= (1 << 23) || (long)a.M * a.x_cs >= (1l << 31)) return false;      // issue_patch's 24-bit address arithmetic
    if (X2 && ((long)(a.M / (a.Ho * a.Wo)) * a.H2 >= (1 << 23) || (long)a.W2 * a.x2_cs >= (1 << 23) ||
               (long)(a.M / (a.Ho * a.Wo)) * a.H2 * a.W2 * a.x2_cs >= (1l << 31))) return false;                // issue_e's
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    const int n_img = a.M / (a.Ho * a.Wo);
    auto kfn = conv3x3_pp_patch_kernel<T, MT, NT, WM, WN, TH, TW, NSTAGE, X2>;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    // tiles per block: at most AICAM_PPP_RUN (default 6).  Same box, 15 360 crops: layer2 conv1 1 993 (one tile per block, the kernel before) ->
    // 1 975 / 1 945 / 1 912 us at runs of 1 / 4 / 6; with a residual 2 257 -> 2 256 (its 9 us epilogue is what is left); layer3 / 4 -2 % / 0
    static const int run_max = [] { const char* e = getenv("AICAM_PPP_RUN"); return e ? std::max(1, atoi(e)) : 6; }();
    const int ny = ceil_div(a.Cout, BN);
    const long ntiles = (long)ceil_div(n_img, NI) * tiles_x * tiles_y * ny;
    int run = 1;                                  // (second source: one tile per block)  the longest run that does not add a round of tiles (256 CUs, one block each) and leaves >= 4 rounds of blocks
    {
        long best = -1;
        for (int r = 1; r <= (X2 ? 1 : run_max); ++r) {
            const long blocks = (ntiles + r - 1) / r, rounds = (blocks + 255) / 256;
            if (r > 1 && rounds < 4) break;
            const long cost = rounds * r;                 // tile times until the last block ends
            if (best < 0 || cost <= best) best = cost, run = r;
        }
    }
    static const bool times = getenv("AICAM_PP_TIMES") != nullptr;
    if (times) {
        const int on = 1;
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_pp_times_on), &on, sizeof(int)));
    }
    // residual prefetch: taps of the last chunk whose spare LDS-DMA slot carries a pass of the residual tile (AICAM_PPP_PF=0xC0: taps 6 and 7).
    // OFF by default: measured on the 512 x 128 tile (AICAM_PP_TIMES, one tile per block) it takes 2 us off the epilogue (9.2 -> 7.2 us) and puts
    // 2.5 us onto the K loop (27.6 -> 30.3 us) -- the pass is a load from HBM in the in-order vmcnt queue, and the next segment's counted wait
    // stands behind it where it used to stand behind a zero-page hit; layer2 / 3 / 4 conv2 alone: 2 432 / 2 021 / 1 894 us off, 2 452 / 2 019 / 1 889 on
    static const int pf_env = [] { const char* e = getenv("AICAM_PPP_PF"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
    constexpr int NPASS_ = NPASS;
    const int pf = (!X2 && a.res_mode != 0 && a.res && sizeof(T) == 2 && BN * sizeof(T) % 128 == 0) ? (pf_env & 0x1ff & ~(((1 << NPASS_) - 1) << 1)) : 0;
    const int nblk = (int)ceil_div(ntiles, (long)run);
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(512), lds, s, a, tiles_x, tiles_y, ny, run, pf);
    if (times) { KCHECK(); pp_times_report(s, nblk); }
    KCHECK();
    return true;
}

// 3x3/s1/p1 layers whose map tiles exactly: pick the tile by map shape and Cout (ReID layer1..4 shapes and their multiples).
// bit 0: Cout 64 (slower than the 4-wave patch kernel: 16 MFMAs per segment), 1: Cout 128, 2: Cout % 256, 3: deeper ring (no gain)
static int ppp_mode() {
    static const int mode = [] { const char* e = getenv("AICAM_PPP"); return e ? atoi(e) : 6; }();
    return mode;
}
// The layer SHAPES this kernel takes (a property of the graph, not of the batch): 1 = Cout 64 tile, 2 = Cout 128, 3 = Cout % 256 on
// 16 x 8 tiles, 4 = on 8 x 4 tiles; 0 = not one of them.  Such a layer is walked chunk-major by EVERY conv kernel (ConvArgs::k_order = 1).
template <typename T>
static int pp_patch_shape(const ConvArgs& a) {
    constexpr int BKE = 64 / (int)sizeof(T);
    const int mode = ppp_mode();
    if (!mode || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % (2 * BKE) || a.Ho != a.H || a.Wo != a.W) return 0;
    const int c = a.Cout;
    if (c == 64 && (mode & 1)) return (a.H % 16 == 0 && a.W % 32 == 0) ? 1 : 0;
    if (c == 128 && (mode & 2)) return (a.H % 32 == 0 && a.W % 16 == 0) ? 2 : 0;
    if (c % 256 == 0 && (mode & 4)) return (a.H % 16 == 0 && a.W % 8 == 0) ? 3 : ((a.H % 8 == 0 && a.W % 4 == 0) ? 4 : 0);
    return 0;
}

template <typename T>
static bool try_pp_patch(const ConvArgs& a, hipStream_t s) {
    static const int pp_min = [] { const char* e = getenv("AICAM_PP_MIN"); return e ? atoi(e) : 200; }();
    const int shape = pp_patch_shape<T>(a);
    if (!shape) return false;
    if ((long)a.M * a.x_cs >= (1l << 31)) return false;                       // 32-bit element offsets inside the kernel
    const int c = a.Cout;
    const bool deep = ppp_mode() & 8;
    if (a.x2) {                                     // a second source (conv_x2_supported: shapes 2, 3 and 4)
        if (shape == 2 && a.M / 512 >= pp_min) return launch_pp_patch<T, 8, 4, 4, 2, 32, 16, 4, true>(a, s);
        if (shape == 3 && (long)(a.M / 256) * (c / 256) >= pp_min) return launch_pp_patch<T, 8, 4, 2, 4, 16, 8, 4, true>(a, s);
        if (shape == 4 && (long)(a.M / 256) * (c / 256) >= pp_min) return launch_pp_patch<T, 8, 4, 2, 4, 8, 4, 4, true>(a, s);
        return false;
    }
    if constexpr (sizeof(T) == 2) {                 // fp16: the software-pipelined form of the same tiles (kernels_conv_sp.hip; same K order, same bits)
        if ((shape == 2 && a.M / 512 >= pp_min) || (shape >= 3 && (long)(a.M / 256) * (c / 256) >= pp_min))
            if (conv_try_sp_patch(a, shape, s)) return true;
    }
    if (shape == 1 && a.M / 512 >= pp_min) return deep ? launch_pp_patch<T, 4, 4, 8, 1, 16, 32, 6>(a, s) : launch_pp_patch<T, 4, 4, 8, 1, 16, 32, 4>(a, s);
    if (shape == 2 && a.M / 512 >= pp_min) return deep ? launch_pp_patch<T, 8, 4, 4, 2, 32, 16, 6>(a, s) : launch_pp_patch<T, 8, 4, 4, 2, 32, 16, 4>(a, s);
    if (shape >= 3 && (long)(a.M / 256) * (c / 256) >= pp_min) {
        if (shape == 3) return deep ? launch_pp_patch<T, 8, 4, 2, 4, 16, 8, 6>(a, s) : launch_pp_patch<T, 8, 4, 2, 4, 16, 8, 4>(a, s);
        return deep ? launch_pp_patch<T, 8, 4, 2, 4, 8, 4, 5>(a, s) : launch_pp_patch<T, 8, 4, 2, 4, 8, 4, 4>(a, s);
    }
    return false;
}


bool conv_try_pp_patch(int dtype, const ConvArgs& a, hipStream_t s) {
    return dtype == AIC_F16 ? try_pp_patch<half_t>(a, s) : false;       // (fp32 engines: the LDS-DMA implicit GEMM only, kernels_conv.hip)
}

int conv_pp_patch_shape(int dtype, const ConvArgs& a) {
    return dtype == AIC_F16 ? pp_patch_shape<half_t>(a) : pp_patch_shape<float>(a);
}

// Ping-pong kernels (one block per CU) where the K loop is long enough to amortise the tile's prologue/epilogue:
// measured on MI355X (tools/conv_bench.py, profiles/): +17..19% on ReID layer3/4, +14% on layer2, a loss at K < 512.
template <typename T>
static bool try_pp(const ConvArgs& a, hipStream_t s) {
    static const bool pp = getenv("AICAM_NO_PP") == nullptr;
    static const int pp_min = [] { const char* e = getenv("AICAM_PP_MIN"); return e ? atoi(e) : 200; }();
    // 18 K-steps: ReID layer2.0.conv1 (3x3 / 2, 64 -> 128, K = 576) takes the 512 x 128 ping-pong tile: 1 058 -> 948 us per 7 680 crops
    // against the 8-wave 256 x 128 LDS-DMA tile (tools/conv_bench.py 64 32 64 128 3 7680 1 0 2); below that the short loop loses
    static const int pp128_k = [] { const char* e = getenv("AICAM_PP128_K"); return e ? atoi(e) : 18; }();
    constexpr int BKE_ = 64 / (int)sizeof(T);
    const int c = a.Cout;
    if (!pp || a.Cin % BKE_ != 0 || !(a.Kp >= 16 * BKE_ || pp_min == 0)) return false;
    if constexpr (sizeof(T) == 2) {                 // stride-2 3x3 layers of the patch kernels' maps: the space-to-depth patch form (kernels_conv_sp.hip)
        if (a.k_order == 3 && ((c % 256 == 0 && (long)ceil_div(a.M, 256) * (c / 256) >= pp_min) || (c == 128 && ceil_div(a.M, 512) >= pp_min)))
            if (conv_try_s2_patch(a, s)) return true;
    }
    if (c % 256 == 0 && (long)ceil_div(a.M, 256) * (c / 256) >= pp_min) {                                  // 256 px x 256 ch
        launch_pp<T, 8, 4, 2, 4, 4>(a, s);
        return true;
    }
    if (c == 128 && (a.Kp >= pp128_k * BKE_ || pp_min == 0) && ceil_div(a.M, 512) >= pp_min) {              // 512 px x 128 ch
        launch_pp<T, 8, 4, 4, 2, 4>(a, s);
        return true;
    }
    return false;
}

bool conv_try_pp(int dtype, const ConvArgs& a, hipStream_t s) {
    return dtype == AIC_F16 ? try_pp<half_t>(a, s) : false;
}

}  // namespace aic
