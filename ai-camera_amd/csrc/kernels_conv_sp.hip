// kernels_conv_sp.hip -- v6: the 3x3 / stride-1 patch kernel SOFTWARE-PIPELINED and stripped to the instructions a K-step needs
// (fp16, whole-image tiles: ReID layer2 / 3 / 4; round 5).
//
// v5 (conv3x3_pp_patch_kernel, kernels_conv_pp.hip) runs two waves per SIMD half a K-step apart: every wave executes a LOAD segment
// (twelve ds_read_b128 of the step's fragments, two or three LDS-DMA issues with their address arithmetic, the ring bookkeeping, the
// counted wait) and then a COMPUTE segment (32 MFMAs), each closed by a block barrier.  Stamped in round 4: LOAD ~822 cycles against
// COMPUTE 512, so a K-step lasts 2 x 822 cycles and the matrix pipe is busy 62 % of them.  Round 4 blamed the LDS-DMA issue; what the
// segment really is, is ~200 INSTRUCTIONS, and a wave issues at most one instruction per four cycles (each SIMD gets an issue turn every
// fourth cycle, one instruction per wave per turn).  tools/ubench/wave_tile.hip strips a K-step to its essential mix -- 32 MFMAs, 12
// fragment reads, 2-3 LDS-DMA, one wait, one barrier -- and measures, in shader clocks per K-step where the matrix pipe needs 1 024:
//     two waves per SIMD, serial LOAD then COMPUTE, all waves in step (v5 without its offset)     1 507   0.68 busy
//     ONE wave per SIMD with a 128 x 128 tile (256 accumulators), reads under its own MFMAs       1 726 - 1 901   0.54 - 0.59
//     two waves per SIMD, 128 x 64 tiles, the NEXT step's fragments read UNDER this step's MFMAs  1 176 - 1 203   0.85 - 0.87
// This kernel is the third form with a K-step of ~75 instructions besides its MFMAs:
//   * same tiles, LDS layout, K order and MFMA sequence per accumulator as v5 (bit-identical outputs, tests/test_gpu_nets.py);
//   * a wave holds the weight fragments twice; while the MFMAs of step k issue, the fragments of step k + 1 are read -- a pixel
//     fragment into the registers its group of four MFMAs has just read, the weight fragments into the second set;
//   * EVERYTHING about a step that can be a constant is one: the K loop is unrolled over 4 chunks x 9 taps (ring stage, fragment set,
//     patch buffer, tap offset, patch-pass number are immediates), the MFMAs are inline asm with the accumulator tied in place (as
//     pure values the allocator moved the accumulators from body to body and fragmented the register file: 17 - 69 spills), the
//     LDS-DMA sources are 64-bit per-lane pointers built once per block plus ONE scalar offset (one VALU instruction per weight
//     load, five per patch pass -- v5 works out (image, y, x) of every lane for every pass: ~25);
//   * first measured form of this file (rolled loop, run-time tap / stage / pass, ~200 instructions per step): 1 028 / 1 127 / 1 138 TFLOP/s on
//     the layer2 / 3 / 4 shapes against v5's 1 126 / 1 270 / 1 319 -- slower, and slower still with v5's half-step offset (981 / 1 070 /
//     1 086): the instruction count, not the arrangement, was what the step waited for.
//
// Ring discipline (4 weight stages, two patch buffers), body k = the K-step (chunk c, tap t), k = 9 c + t:
//   body k   issues the LDS-DMA of step k + 3 into stage (k + 3) & 3 -- last read in body k - 2 -- plus one patch pass / stand-in,
//            reads the fragments of step k + 1 (landed: see the wait of body k - 1), issues the MFMAs of step k,
//            waits until only its OWN loads are in flight (step k + 2 and every patch pass issued before this body have landed) and
//            for its fragment reads, s_barrier.
//   patch    passes of chunk c + 1 are issued in the bodies of taps 0 .. NPASS - 1 of chunk c (NPASS <= 7) into the other buffer,
//            whose last reader was body (c - 1, tap 7); the last pass has landed at the end of body tap NPASS <= 7, the first read of
//            the buffer is in body tap 8.
//   tiles    a block walks a run of tiles as in v5; a tile's LAST body reads whatever the next tile's buffers hold by then and the
//            next tile's first fragments are read again behind the epilogue (the fragments would have to live across it).
#include "conv_common.hpp"

namespace aic {

template <int MT, int NT, int WM, int WN, int TH, int TW, bool SKEW>
__global__ __launch_bounds__(512) void conv3x3_sp_patch_kernel(const ConvArgs a, int ny, int run) {
    typedef half_t T;
    constexpr int NSTAGE = 4;
    constexpr int CH = 8, BKE = 32, RP = 128;
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    constexpr int B_PER = BN / RP;
    constexpr int LPS = B_PER + 1;
    constexpr int TPIX = TH * TW, NI = BM / TPIX;
    constexpr int PW = TW + 2, PH = TH + 2, IPIX = PW * PH, IPIXP = ppp_ipix_pad(TH, TW), NPIX = NI * IPIXP;
    constexpr int G = TW >= 16 ? 1 : 16 / TW;                          // images per 16-pixel MFMA tile
    constexpr int NPASS = (NPIX + 127) / 128, NPIXP = NPASS * 128;
    constexpr int PLANE = NPIXP * 16, PBUF = 4 * PLANE, WSTAGE = BN * 64;
    // LDS: the weight ring FIRST (ring stage + tile offsets of the weight fragment reads fit ds_read's 16-bit immediate), then the two patch
    // buffers, then the stand-in slot
    constexpr int PATCH0 = NSTAGE * WSTAGE, DUM0 = PATCH0 + 2 * PBUF;
    static_assert(WM * WN == 8 && MT == 8 && NT == 4 && BN % RP == 0 && BM % TPIX == 0 && NPASS <= 7 && TW % 4 == 0, "geometry");
    static_assert((NSTAGE - 1) * WSTAGE + 2048 < 65536 && PBUF + (2 * PW + 2 + IPIXP * NI) * 16 < 65536, "ds_read immediates");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool late = SKEW && wv >= 4;
    const int HW = a.H * a.W;
    const int n_img = a.n_dev ? min(a.M / HW, a.n_dev[0]) : a.M / HW;     // device-side item count: the grid was sized for a bound
    // tiles: (image group, channel tile), channel tile fastest; a block takes a run of `run` consecutive tiles
    const int ntiles = ((n_img + NI - 1) / NI) * ny;
    const int nruns = (ntiles + run - 1) / run;
    if ((int)blockIdx.x >= nruns) return;
    int t_cur = xcd_tile((int)blockIdx.x, nruns, a.xcd_map) * run;
    const int t_end = min(t_cur + run, ntiles);
    int img0 = (t_cur / ny) * NI, n0 = (t_cur % ny) * BN;          // the tile being computed (block-uniform)

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    // ---- per-lane LDS-DMA sources, built once.  Patch pass i, wave w -> plane w & 3, patch pixels i * 128 + (w >> 2) * 64 + lane: the pixel's
    // address in image 0 of the tensor, channel chunk 0 (or the zero page for halo pixels outside the image and padding slots), and its image
    // within the tile (or a huge number: "never live").  A pass then is: live lanes add ONE scalar offset (tile + chunk), the others add 0.
    const int plane = wv & 3;
    const T* pptr[NPASS];
    int pil[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int p = i * 128 + (wv >> 2) * 64 + lane;
        const int il = p / IPIXP, rem = p - il * IPIXP;
        const int py = rem / PW, px = rem - py * PW;
        const int iy = py - 1, ix = px - 1;
        const bool ok = p < NPIX && rem < IPIX && (unsigned)iy < (unsigned)TH && (unsigned)ix < (unsigned)TW;
        pptr[i] = ok ? xg + ((size_t)((il * TH + iy) * TW + ix) * a.x_cs + a.x_coff + plane * CH) : zero;
        pil[i] = ok ? il : 0x40000000;
    }
    char* const pdst = smem + PATCH0 + plane * PLANE + (wv >> 2) * 1024;
    const unsigned img_bytes = (unsigned)HW * a.x_cs * 2u;            // one image of the input, bytes (launch_sp_patch: the tensor fits 2^32 bytes)
    // weights: row r0 + RP * j of channel tile 0 at K = 0; a load adds the scalar offset of (channel tile, step)
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    const T* wptr[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) wptr[j] = wg + (size_t)(r0 + RP * j) * a.Kp + kc * CH;
    char* const wdst = smem + (16 * wv) * 64;
    const int nchunks = a.Cin / BKE;
    const unsigned tile_wbytes = (unsigned)BN * a.Kp * 2u;            // weights of one channel tile, bytes
    const unsigned tap_bytes = (unsigned)a.Cin * 2u;

    // the weight stream: (channel tile, chunk, tap) of the step fetched next as ONE scalar byte offset, three steps ahead of the MFMAs
    unsigned long long w_off = (unsigned long long)(n0 / BN) * tile_wbytes;
    int w_c = 0;                                   // its chunk
    int n0_next = n0, img_next = img0;
    bool has_next = t_cur + 1 < t_end;
    if (has_next) { img_next = ((t_cur + 1) / ny) * NI; n0_next = ((t_cur + 1) % ny) * BN; }
    auto issue_w = [&](int st, int wtap) {         // st: ring stage; wtap: the fetched step's tap -- constants after inlining
        char* sbase = wdst + st * WSTAGE;
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const T* src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(wptr[j]) + w_off);
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + j * (RP * 64)), 16, 0, 0);
        }
        if (wtap < 8) w_off += tap_bytes;          // next tap of the chunk
        else {                                     // next chunk: back eight taps, on one K-step of channels
            w_off += 64u - 8u * (unsigned long long)tap_bytes;
            if (++w_c == nchunks) {                // the stream moves on: the next tile of the run, or -- nothing to fetch for -- this tile again
                w_c = 0;                           // (a harmless re-fetch of at most three steps: always inside the weights)
                w_off = (unsigned long long)((has_next ? n0_next : n0) / BN) * tile_wbytes;
            }
        }
    };
    // this chunk's patch passes fetch the chunk that FOLLOWS (the next channel chunk of this tile, or chunk 0 of the run's next tile)
    unsigned long long p_off = 0;                  // scalar byte offset the live lanes add
    int p_lim = 0;                                 // images of the tile that exist (0: no pass is live)
    auto set_patch = [&](int c) {
        const bool inner = c + 1 < nchunks;
        const int im = inner ? img0 : img_next;
        p_off = (unsigned long long)im * img_bytes + (inner ? (unsigned)(c + 1) * 64u : 0u);
        p_lim = (inner || has_next) ? n_img - im : 0;
    };
    auto issue_patch = [&](auto ic, int buf) {
        constexpr int i = decltype(ic)::value;
        const bool live = pil[i] < p_lim;
        const unsigned lo = live ? (unsigned)p_off : 0u, hi = live ? (unsigned)(p_off >> 32) : 0u;
        const T* src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(pptr[i]) + (((unsigned long long)hi << 32) | lo));
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(pdst + buf * PBUF + i * 2048), 16, 0, 0);
    };
    auto issue_dummy = [&] {
        const T* src = zero;
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + DUM0 + wv * 1024), 16, 0, 0);
    };

    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;
    static_assert((MT * 16) % TPIX == 0 || TPIX % (MT * 16) == 0, "a wave's pixels must not straddle images irregularly");
    static_assert(TW % 16 == 0 || 16 % TW == 0, "a 16-pixel MFMA tile is whole rows or a piece of one row");
    static_assert(G == 1 || (NI % G == 0 && (MT % TH == 0 || TH % MT == 0)), "image groups");
    auto tile_pix = [](int t) constexpr { return G == 1 ? (t * 16 / TPIX) * IPIXP + ((t * 16 % TPIX) / TW) * PW + (t * 16 % TPIX) % TW
                                                        : (t / TH) * G * IPIXP + (t % TH) * PW; };
    auto patch_pix = [tile_pix](int m) constexpr { return G == 1 ? tile_pix(m / 16) : (TH % MT == 0 ? (m / 16) * PW : tile_pix(m / 16)); };
    int xa0;                                    // LDS byte address of (this lane's pixel of tile 0, tap (0, 0)) in patch buffer 0
    if constexpr (G == 1) {
        const int ml = wm * MT * 16 + r;
        const int il = ml / TPIX, rem = ml - il * TPIX;
        const int ly = rem / TW, lx = rem - ly * TW;
        xa0 = PATCH0 + q * PLANE + (il * IPIXP + ly * PW + lx) * 16;
    } else {
        const int t0 = wm * MT;
        xa0 = PATCH0 + q * PLANE + ((t0 / TH) * G * IPIXP + (t0 % TH) * PW + (r / TW) * IPIXP + r % TW) * 16;
    }
    int woff2[2];                               // weight fragment addresses: tiles j and j + 2 are 32 rows (2048 B) apart, j and j + 1 differ in the swizzle term
#pragma unroll
    for (int j = 0; j < 2; ++j) woff2[j] = lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);

    floatx4 acc[MT][NT];
    half8 xf[MT], wf[2][NT];
    auto read_w = [&](auto setc, int st) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[S][j] = *reinterpret_cast<const half8*>(smem + woff2[j & 1] + st * WSTAGE + (j >> 1) * 2048);
    };
    auto read_x = [&](int i, int buf, int tap) {       // all three are constants after inlining: one ds_read_b128 with an immediate
        xf[i] = *reinterpret_cast<const half8*>(smem + xa0 + buf * PBUF + ((tap / 3) * PW + tap % 3) * 16 + patch_pix(16 * i) * 16);
    };

    // ---- prologue (first tile of the run): patch chunk 0, weights of steps 0 .. 2
    p_off = (unsigned long long)img0 * img_bytes, p_lim = n_img - img0;
    static_for<NPASS>([&](auto ic) { issue_patch(ic, 0); });
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) {
        issue_w(st, st);
        if (st) issue_dummy();                  // every set in flight has LPS loads: the counted waits rely on it
    }
    wait_vmcnt<LPS>();                          // patch chunk 0 and the weights of steps 0 and 1 have landed (this wave's part)
    __builtin_amdgcn_s_barrier();

    // One K-step; C4 = chunk & 3 and the tap are compile-time.  Issue order (a scheduling fence behind every group):
    //   group i = the four MFMAs of pixel tile i, then that tile's fragment of step k + 1 into the registers they have just read;
    //   group 0 also reads the four weight fragments of step k + 1 (second set), group 1 issues the LDS-DMA of the weights of step k + 3,
    //   group 2 this body's patch pass (or its stand-in).
    auto body = [&](int c, auto c4c, auto tapc) {
        constexpr int C4 = decltype(c4c)::value, tap = decltype(tapc)::value;
        constexpr int BUF = C4 & 1, ST = (C4 + tap) & 3, CUR = ST & 1, NXT = CUR ^ 1;      // 9 c + tap = c + tap (mod 4); a tile starts at a multiple of 4
        constexpr int nbuf = tap < 8 ? BUF : BUF ^ 1, ntap = tap < 8 ? tap + 1 : 0;
        if constexpr (tap == 0) set_patch(c);
        static_for<MT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int j = 0; j < NT; ++j)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(wf[CUR][j]), "v"(xf[i]));
            read_x(i, nbuf, ntap);
            if constexpr (i == 0) read_w(std::integral_constant<int, NXT>{}, (ST + 1) & 3);
            if constexpr (i == 1) issue_w((ST + 3) & 3, (tap + 3) % 9);
            if constexpr (i == 2) {
                if constexpr (tap < NPASS) issue_patch(tapc, BUF ^ 1);
                else issue_dummy();
            }
            if constexpr (SKEW && i == MT / 2 - 1) {          // mid-body barrier: waves 4 .. 7 run half a body behind (see the tile loop)
                __builtin_amdgcn_s_waitcnt(0x0F70 | LPS);     // vmcnt(LPS): everything older than this body's own loads has landed
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // the waits as BUILTINS, not inline asm: the compiler's own wait-count pass must see them, or it protects the first MFMA of the next
        // body -- whose operands these very waits have covered -- with an lgkmcnt(0) of its own BEHIND the next body's reads.
        // s_waitcnt immediate (gfx9+): vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
        __builtin_amdgcn_s_waitcnt(0x0070 | LPS);             // vmcnt(LPS), expcnt(7), lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };
    for (;;) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        // the tile's first fragments, un-overlapped: step 0 = ring stage 0, patch buffer 0, tap 0 (landed and behind a barrier: prologue / last body)
        read_w(std::integral_constant<int, 0>{}, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) read_x(i, 0, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0)
        // SKEW: waves 4 .. 7 run half a body behind waves 0 .. 3 (one extra barrier here, one for the early half behind the loop).  Every wave
        // waits for its older loads in front of BOTH barriers of a body, so a late wave's part of step k + 1 has landed before barrier 2k - 1
        // (its mid-body k - 1), which an early wave passes before it reads step k + 1 in body k
        if (late) __builtin_amdgcn_s_barrier();
        for (int c = 0; c < nchunks; c += 4) {     // Cin / 32 is a multiple of 4 for every layer that reaches this kernel
            static_for<4>([&](auto c4c) {
                static_for<9>([&](auto tapc) { body(c + decltype(c4c)::value, c4c, tapc); });
            });
        }
        if (SKEW && !late) __builtin_amdgcn_s_barrier();      // every wave has executed the same number of barriers: both halves store together
        if (!has_next) wait_vmcnt<0>();
        // the MFMAs are inline asm: the compiler does not know that the accumulators come out of the matrix pipe (up to 18 wait states
        // before a VALU may read them); the barrier and the row arithmetic below are far more than that, two s_nop make it independent of them
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

        int mrow[MT];
        const int ln = lane_here(), rr = ln & 15, qe = ln >> 4;
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
            mrow[i] = img < n_img ? (img * TH + ly) * TW + lx : -1;
        }
        epilogue_dispatch<T, MT, NT, true, true>(a, acc, mrow, n0 + wn * NT * 16, qe);
        if (!has_next) break;
        ++t_cur;
        img0 = img_next, n0 = n0_next;
        has_next = t_cur + 1 < t_end;
        if (has_next) { img_next = ((t_cur + 1) / ny) * NI; n0_next = ((t_cur + 1) % ny) * BN; }
    }
}

template <int MT, int NT, int WM, int WN, int TH, int TW>
static bool launch_sp_patch(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int NI = BM / (TH * TW), NPIX = NI * ppp_ipix_pad(TH, TW), NPASS = (NPIX + 127) / 128;
    constexpr size_t lds = (size_t)2 * 4 * NPASS * 128 * 16 + 8192 + (size_t)4 * BN * 64;
    static_assert(lds <= 160 * 1024, "does not fit the LDS");
    if (a.H != TH || a.W != TW || a.Ho != a.H || a.Wo != a.W || a.Cout % BN) return false;            // whole-image tiles, whole channel tiles
    if ((long)a.M * a.x_cs * 2 >= (1l << 32) || (long)a.Cout * a.Kp * 2 >= (1l << 32)) return false;   // 32-bit byte strides inside the kernel
    const int n_img = a.M / (a.Ho * a.Wo);
    static const bool skew = getenv("AICAM_SP_SKEW") != nullptr;
    auto kfn = skew ? conv3x3_sp_patch_kernel<MT, NT, WM, WN, TH, TW, true> : conv3x3_sp_patch_kernel<MT, NT, WM, WN, TH, TW, false>;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    static const int run_max = [] { const char* e = getenv("AICAM_PPP_RUN"); return e ? std::max(1, atoi(e)) : 6; }();
    const int ny = a.Cout / BN;
    const long ntiles = (long)ceil_div(n_img, NI) * ny;
    int run = 1;                                  // the longest run that does not add a round of tiles (256 CUs, one block each) and leaves >= 4 rounds of blocks
    {
        long best = -1;
        for (int r = 1; r <= run_max; ++r) {
            const long blocks = (ntiles + r - 1) / r, rounds = (blocks + 255) / 256;
            if (r > 1 && rounds < 4) break;
            const long cost = rounds * r;
            if (best < 0 || cost <= best) best = cost, run = r;
        }
    }
    const int nblk = (int)ceil_div(ntiles, (long)run);
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(512), lds, s, a, ny, run);
    KCHECK();
    return true;
}

// (The same schedule with the pixel operand as an im2col tile -- for the STRIDE-2 3x3 convs of ReID layerN.0.conv1 -- was built too
// (conv3x3_sp_igemm_kernel: per-lane row pointers + one scalar offset per step, a 9-bit tap mask per row, runs of tiles, ~70 instructions
// per K-step besides the MFMAs; bit-identical to v4, 12 ReID tests green) and measured against v4 on 15 360 crops: 1 906 against ~1 900 us
// on layer2.0 (64 -> 128), 951 / 1 079 against 963 / 1 159 TFLOP/s on layer3.0 / 4.0.  No gain: an im2col K-step of a stride-2 conv pulls
// 512 half cache lines (64 bytes, 256 bytes apart) through the L2 -- 9 GB per launch on layer2.0, 4.8 TB/s of LDS-DMA at the time it
// takes, on top of 3.2 TB/s of HBM -- and that, not the instruction count, is what those three layers wait for.  Removed; what they
// need is the patch form on the space-to-depth view of their input, which reads every input byte once.)

// ------------------------------------------------------------------------------------------------
// v6 for the STRIDE-2 3x3 convs (ReID layerN.0.conv1: 64 -> 128 on 64 x 32, 128 -> 256 on 32 x 16, 256 -> 512 on 16 x 8 maps), which the
// im2col kernels run L2-bound (see the note above): the PATCH form on the SPACE-TO-DEPTH view of the input, every input byte fetched once.
// Measured against v4 on 15 360 crops, each layer alone (tools/ab_s2d.sh): layer2.0 1 748 against 1 918 us (663 TFLOP/s; its HBM floor -- 4 GB in,
// 2 GB out -- is ~1.0 ms: a tile is only 18 K-steps and its 128 KB epilogue is not overlapped), layer3.0 1 169 against 1 215 us, layer4.0 slower.
//
// Row 2i + di - 1 of the input is row a = i - 1 (di = 0) or a = i (di = 1, 2) of the parity plane pi = 1, 0, 1: with z[a][b][(pi, pj)][c] =
// x[2a + pi][2b + pj][c] the conv is a 2 x 2 / stride-1 conv over the four parity planes -- tap shifts (kh, kw) in {0, 1}^2 with a halo of
// one row / column on the top / left only -- in which plane (pi, pj) has 1 (pi = 0) or 2 (pi = 1) row taps and as many column taps:
//   plane (1,1): taps (0,0) (0,1) (1,0) (1,1) = original (di, dj) (0,0) (0,2) (2,0) (2,2);   plane (1,0): (0,1) (1,1) = (0,1) (2,1);
//   plane (0,1): (1,0) (1,1) = (1,0) (1,2);                                                 plane (0,0): (1,1) = (1,1).
// The view needs no copy: patch pixel (a, b) of plane (pi, pj), channel chunk h is x[2a + pi][2b + pj][32 h ..] -- a per-lane pointer built
// once (plane (0,0), chunk 0) plus ONE scalar offset per pass; the weights stay in their [Cout][tap][Cin] layout, a K-step fetches
// K offset tap * Cin + 32 h.  K order (ConvArgs::k_order 3, walked by every kernel for these layers): for each channel chunk h the nine
// taps in the order 0 2 6 8 | 1 7 | 3 5 | 4 (plane by plane).
// A chunk (plane, h) lives in one of THREE patch buffers and lasts 4 / 2 / 2 / 1 K-steps, so its LDS-DMA passes are issued up to two chunks
// ahead, by a fixed schedule over the nine bodies of an h-period (PASSES below: which chunk, how many passes; every body has at least
// one, none more than three); a body's counted wait lets exactly its own loads stay in flight.  Ring stage and patch buffer are run-time
// (scalar) state here -- the period is 9 bodies, not a multiple of 4 or 3 --: two v_add per body for the weight fragments' addresses, one
// for the patch's.  Everything else as conv3x3_sp_patch_kernel.
template <int NPASS> struct S2dSched;          // passes issued in body b for target t: 0 = c1 (plane 1,0), 1 = c2 (0,1), 2 = c3 (0,0), 3 = c0 of the NEXT period
template <> struct S2dSched<5> { static constexpr int n[9][4] = {{1,1,0,0},{1,1,0,0},{0,2,0,0},{0,0,2,0},{0,0,3,0},{0,0,0,3},{0,0,0,2},{2,0,0,0},{1,1,0,0}}; };
template <> struct S2dSched<4> { static constexpr int n[9][4] = {{1,1,0,0},{1,1,0,0},{0,1,0,0},{0,1,1,0},{0,0,2,0},{0,0,1,2},{0,0,0,2},{1,0,0,0},{1,0,0,0}}; };
template <> struct S2dSched<3> { static constexpr int n[9][4] = {{1,0,0,0},{0,1,0,0},{0,1,0,0},{0,1,1,0},{0,0,2,0},{0,0,0,2},{0,0,0,1},{1,0,0,0},{1,0,0,0}}; };
// (bodies 7 and 8 fetch c1 / c2 of the NEXT period; the counts per target sum to NPASS over the cyclic window that ends two bodies before
//  the chunk's first fragment read: c1 [7, 8, 0, 1], c2 [8, 0 .. 3], c3 [3 .. 5], next c0 [5, 6])

template <int WM, int WN, int TH, int TW>
__global__ __launch_bounds__(512) void conv3x3s2_sp_patch_kernel(const ConvArgs a, int ny, int run) {
    typedef half_t T;
    constexpr int MT = 8, NT = 4, NSTAGE = 4;
    constexpr int CH = 8, BKE = 32, RP = 128;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, B_PER = BN / RP;
    constexpr int TPIX = TH * TW, NI = BM / TPIX;
    constexpr int PW = TW + 2, PH = TH + 2, IPIX = PW * PH, IPIXP = ppp_ipix_pad(TH, TW), NPIX = NI * IPIXP;
    constexpr int G = TW >= 16 ? 1 : 16 / TW;
    constexpr int NPASS = (NPIX + 127) / 128, NPIXP = NPASS * 128;
    constexpr int PLANE = NPIXP * 16, PBUF = 4 * PLANE, WSTAGE = BN * 64;
    constexpr int PATCH0 = NSTAGE * WSTAGE;                          // LDS: weight ring, then THREE patch buffers
    typedef S2dSched<NPASS> SCH;
    static_assert(WM * WN == 8 && BN % RP == 0 && BM % TPIX == 0 && TW % 4 == 0, "geometry");
    static_assert((PW + 1 + IPIXP * NI) * 16 < 65536, "ds_read immediates");
    constexpr int H = 2 * TH, W = 2 * TW;                            // the input map

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n_img = a.n_dev ? min(a.M / TPIX, a.n_dev[0]) : a.M / TPIX;
    const int ntiles = ((n_img + NI - 1) / NI) * ny;
    const int nruns = (ntiles + run - 1) / run;
    if ((int)blockIdx.x >= nruns) return;
    int t_cur = xcd_tile((int)blockIdx.x, nruns, a.xcd_map) * run;
    const int t_end = min(t_cur + run, ntiles);
    int img0 = (t_cur / ny) * NI, n0 = (t_cur % ny) * BN;

    const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
    const T* zero = reinterpret_cast<const T*>(a.zero);

    // per-lane LDS-DMA sources: patch pixel (py, px) = view pixel (a, b) = (py - 1, px - 1) of plane (0, 0), channel chunk 0, image 0 of the
    // tensor as a 32-bit BYTE offset from the input (launch_s2_patch: the tensor fits 2^32 bytes), or ~0 for the halo and the padding slots
    // (they fetch the zero page).  One register per pass: with 64-bit pointers and the pixel's image kept as well the kernel spilled.
    const int plane = wv & 3;
    unsigned poff[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int p = i * 128 + (wv >> 2) * 64 + lane;
        const int il = p / IPIXP, rem = p - il * IPIXP;
        const int py = rem / PW, px = rem - py * PW;
        const int va = py - 1, vb = px - 1;
        const bool ok = p < NPIX && rem < IPIX && (unsigned)va < (unsigned)TH && (unsigned)vb < (unsigned)TW;
        poff[i] = ok ? (unsigned)((((il * H + 2 * va) * W + 2 * vb) * a.x_cs + a.x_coff + plane * CH) * 2) : 0xffffffffu;
    }
    char* const pdst = smem + PATCH0 + plane * PLANE + (wv >> 2) * 1024;
    const unsigned img_bytes = (unsigned)(H * W) * a.x_cs * 2u;
    const int slot = t & 3, r0 = t >> 2;
    const int kc = slot ^ lds_swz(r0);
    const T* wptr[B_PER];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) wptr[j] = wg + (size_t)(r0 + RP * j) * a.Kp + kc * CH;
    char* const wdst = smem + (16 * wv) * 64;
    const int nh = a.Cin / BKE;                                      // channel chunks = h-periods of a tile (even)
    const unsigned tile_wbytes = (unsigned)BN * a.Kp * 2u, tap_bytes = (unsigned)a.Cin * 2u;

    int n0_next = n0, img_next = img0;
    bool has_next = t_cur + 1 < t_end;
    if (has_next) { img_next = ((t_cur + 1) / ny) * NI; n0_next = ((t_cur + 1) % ny) * BN; }

    // ---- compile-time tables of an h-period: body b computes plane PL[b] at tap (KH[b], KW[b]); its weights are the original tap TAP[b]
    constexpr int PL_[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};              // chunk within the period: c0 = plane (1,1), c1 = (1,0), c2 = (0,1), c3 = (0,0)
    constexpr int KH_[9] = {0, 0, 1, 1, 0, 1, 1, 1, 1}, KW_[9] = {0, 1, 0, 1, 1, 1, 0, 1, 1};
    constexpr int TAP_[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
    constexpr int CPI_[4] = {1, 1, 0, 0}, CPJ_[4] = {1, 0, 1, 0};      // parity plane of chunk c0 .. c3

    // weights of K-step (period h, body b) of channel tile nn -> ring stage st
    auto issue_w = [&](int st, int nn, int hh, int tapi) {
        unsigned long long w_off = (unsigned long long)(nn / BN) * tile_wbytes + (unsigned long long)tapi * tap_bytes + (unsigned)hh * 64u;
        asm volatile("" : "+s"(w_off));            // opaque: or the nine wptr + tap * tap_bytes sums become loop invariants that do not fit the register file
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const T* src = reinterpret_cast<const T*>(reinterpret_cast<const char*>(wptr[j]) + w_off);
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wdst + st * WSTAGE + j * (RP * 64)), 16, 0, 0);
        }
    };
    // pass i of chunk (plane (pi, pj), period hh) of the tile at image im into patch buffer `buf`; lim: images of that tile that exist (0: none)
    auto udiv24 = [](int x, auto dc) { constexpr int D = decltype(dc)::value; constexpr unsigned Mg = (1u << 24) / D + 1u; return (int)(__umul24((unsigned)x, Mg) >> 24); };
    auto issue_patch = [&](auto ic, int buf, int im, int lim, int pi, int pj, int hh) {
        constexpr int i = decltype(ic)::value;
        unsigned long long base = (unsigned long long)reinterpret_cast<uintptr_t>(xg) + (unsigned long long)im * img_bytes +
                                  (unsigned long long)(((pi * W + pj) * a.x_cs + hh * BKE) * 2);                // scalar
        asm volatile("" : "+s"(base));             // (opaque, as in issue_w)
        int il = 0;                                            // the pixel's image within the tile, worked out here (not kept)
        if constexpr (NI > 1) il = udiv24(i * 128 + (wv >> 2) * 64 + lane_here(), std::integral_constant<int, IPIXP>{});
        const bool live = poff[i] != 0xffffffffu && il < lim;
        const unsigned long long addr = base + poff[i];
        const unsigned long long zaddr = (unsigned long long)reinterpret_cast<uintptr_t>(zero);
        const T* src = reinterpret_cast<const T*>(live ? addr : zaddr);
        asm volatile("" : "+v"(src));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(pdst + buf * PBUF + i * 2048), 16, 0, 0);
    };

    const int wm = wv / WN, wn = wv % WN;
    const int q = lane >> 4, r = lane & 15;
    static_assert((MT * 16) % TPIX == 0 || TPIX % (MT * 16) == 0, "a wave's pixels must not straddle images irregularly");
    static_assert(TW % 16 == 0 || 16 % TW == 0, "a 16-pixel MFMA tile is whole rows or a piece of one row");
    static_assert(G == 1 || (NI % G == 0 && (MT % TH == 0 || TH % MT == 0)), "image groups");
    auto tile_pix = [](int t) constexpr { return G == 1 ? (t * 16 / TPIX) * IPIXP + ((t * 16 % TPIX) / TW) * PW + (t * 16 % TPIX) % TW
                                                        : (t / TH) * G * IPIXP + (t % TH) * PW; };
    auto patch_pix = [tile_pix](int m) constexpr { return G == 1 ? tile_pix(m / 16) : (TH % MT == 0 ? (m / 16) * PW : tile_pix(m / 16)); };
    int xa0;
    if constexpr (G == 1) {
        const int ml = wm * MT * 16 + r;
        const int il = ml / TPIX, rem = ml - il * TPIX;
        const int ly = rem / TW, lx = rem - ly * TW;
        xa0 = PATCH0 + q * PLANE + (il * IPIXP + ly * PW + lx) * 16;
    } else {
        const int t0 = wm * MT;
        xa0 = PATCH0 + q * PLANE + ((t0 / TH) * G * IPIXP + (t0 % TH) * PW + (r / TW) * IPIXP + r % TW) * 16;
    }
    int woff2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) woff2[j] = lds_off(wn * NT * 16 + perm_row<NT>(j, r), q);

    floatx4 acc[MT][NT];
    half8 xf[MT], wf[2][NT];
    auto read_w = [&](auto setc, int st) {                           // st: run-time ring stage
        constexpr int S = decltype(setc)::value;
        const int w0 = woff2[0] + st * WSTAGE, w1 = woff2[1] + st * WSTAGE;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[S][j] = *reinterpret_cast<const half8*>(smem + ((j & 1) ? w1 : w0) + (j >> 1) * 2048);
    };
    auto read_x = [&](int i, int xaddr, int kh, int kw) {            // xaddr = xa0 + buffer offset (run time); (kh, kw): constants
        xf[i] = *reinterpret_cast<const half8*>(smem + xaddr + (kh * PW + kw) * 16 + patch_pix(16 * i) * 16);
    };

    // ---- prologue (first tile of the run): chunk c0 of period 0 whole, and of c1 / c2 the passes the schedule issues in bodies 7 / 8 of the
    // period BEFORE; the weights of steps 0 .. 2.  Patch buffers: c0 -> 0, c1 -> 1, c2 -> 2.
    int st = 0;                                 // ring stage of the step whose MFMAs are issued
    int bufc = 0;                               // patch buffer of the chunk whose MFMAs are issued
    {
        const int lim = n_img - img0;
        static_for<NPASS>([&](auto ic) { issue_patch(ic, 0, img0, lim, 1, 1, 0); });
        static_for<NPASS>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < SCH::n[7][0] + SCH::n[8][0]) issue_patch(ic, 1, img0, lim, 1, 0, 0);
            if constexpr (i < SCH::n[8][1]) issue_patch(ic, 2, img0, lim, 0, 1, 0);
        });
        issue_w(0, n0, 0, TAP_[0]);
        issue_w(1, n0, 0, TAP_[1]);
        wait_vmcnt<0>();
        issue_w(2, n0, 0, TAP_[2]);             // (in flight across the first barrier like a body's own loads)
        __builtin_amdgcn_s_barrier();
    }

    // One K-step: body b (compile time) of period h (run time); PAR = h & 1 decides the fragment set
    auto body = [&](int h, auto parc, auto bc) {
        constexpr int PAR = decltype(parc)::value, b = decltype(bc)::value;
        constexpr int CUR = (PAR + b) & 1, NXT = CUR ^ 1;            // 9 h + b = h + b (mod 2); a tile has an even number of periods
        constexpr int nb = (b + 1) % 9;                              // the next step: its plane, tap -- and whether it opens a new chunk
        constexpr bool new_chunk = PL_[nb] != PL_[b];
        const int buf_next = new_chunk ? (bufc == 2 ? 0 : bufc + 1) : bufc;
        const int xaddr = xa0 + buf_next * PBUF;
        const int st_n = (st + 1) & 3, st_w = (st + 3) & 3;
        // the weight stream is three steps ahead: body (b + 3) % 9 of this period or the next; past the tile's last period: the next tile
        constexpr int wb = (b + 3) % 9;
        int w_h = h + (b + 3 >= 9 ? 1 : 0), w_n = n0;
        if (w_h == nh) { w_h = 0; w_n = has_next ? n0_next : n0; }    // (nothing to fetch for: this tile again -- always inside the weights)
        // patch targets of this body: chunk c1 / c2 / c3 of this period (bodies 7, 8: of the next) and c0 of the next period
        constexpr int n1 = SCH::n[b][0], n2 = SCH::n[b][1], n3 = SCH::n[b][2], n4 = SCH::n[b][3];
        constexpr int LPS_B = B_PER + n1 + n2 + n3 + n4;
        // passes already issued for a target before this body (cyclic over the target's window)
        auto before = [](int tgt, int bb) constexpr {
            constexpr int start[4] = {7, 8, 3, 5};
            int cnt = 0;
            for (int k = start[tgt]; k != bb; k = (k + 1) % 9) cnt += SCH::n[k][tgt];
            return cnt;
        };
        constexpr bool nextp1 = b >= 7, nextp2 = b >= 8;             // c1 / c2 fetched for the NEXT period
        int t_h[4] = {h + (nextp1 ? 1 : 0), h + (nextp2 ? 1 : 0), h, h + 1}, t_im[4], t_lim[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool over = t_h[k] >= nh;                           // the next tile's first period
            t_im[k] = over ? img_next : img0;
            t_lim[k] = over ? (has_next ? n_img - img_next : 0) : n_img - img0;
            t_h[k] = over ? 0 : t_h[k];
        }
        // buffer of a target relative to the chunk being computed: (chunks ahead) mod 3
        constexpr int ct = PL_[b];
        constexpr int ahead[4] = {((nextp1 ? 4 : 0) + 1 - ct + 3) % 3, ((nextp2 ? 4 : 0) + 2 - ct + 3) % 3, (3 - ct) % 3, (4 - ct) % 3};
        auto tbuf = [&](int k) { const int v = bufc + ahead[k]; return v >= 3 ? v - 3 : v; };
        static_for<MT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int j = 0; j < NT; ++j)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(wf[CUR][j]), "v"(xf[i]));
            read_x(i, xaddr, KH_[nb], KW_[nb]);
            if constexpr (i == 0) read_w(std::integral_constant<int, NXT>{}, st_n);
            if constexpr (i == 1) issue_w(st_w, w_n, w_h, TAP_[wb]);
            if constexpr (i == 2) {
                static_for<NPASS>([&](auto pc) {
                    constexpr int pi_ = decltype(pc)::value;
                    if constexpr (pi_ >= before(0, b) && pi_ < before(0, b) + n1) issue_patch(pc, tbuf(0), t_im[0], t_lim[0], CPI_[1], CPJ_[1], t_h[0]);
                    if constexpr (pi_ >= before(1, b) && pi_ < before(1, b) + n2) issue_patch(pc, tbuf(1), t_im[1], t_lim[1], CPI_[2], CPJ_[2], t_h[1]);
                    if constexpr (pi_ >= before(2, b) && pi_ < before(2, b) + n3) issue_patch(pc, tbuf(2), t_im[2], t_lim[2], CPI_[3], CPJ_[3], t_h[2]);
                    if constexpr (pi_ >= before(3, b) && pi_ < before(3, b) + n4) issue_patch(pc, tbuf(3), t_im[3], t_lim[3], CPI_[0], CPJ_[0], t_h[3]);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        st = st_n;
        bufc = buf_next;
        __builtin_amdgcn_s_waitcnt(0x0070 | LPS_B);            // vmcnt(this body's loads), expcnt(7), lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };
    for (;;) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        // the tile's first fragments, un-overlapped: ring stage st, patch buffer bufc, plane (1,1), tap (0, 0)
        read_w(std::integral_constant<int, 0>{}, st);
        {
            const int xaddr = xa0 + bufc * PBUF;
#pragma unroll
            for (int i = 0; i < MT; ++i) read_x(i, xaddr, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0)
        for (int h = 0; h < nh; h += 2) {
            static_for<9>([&](auto bc) { body(h, std::integral_constant<int, 0>{}, bc); });
            static_for<9>([&](auto bc) { body(h + 1, std::integral_constant<int, 1>{}, bc); });
        }
        if (!has_next) wait_vmcnt<0>();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // (inline-asm MFMAs: see conv3x3_sp_patch_kernel)

        int mrow[MT];
        const int ln = lane_here(), rr = ln & 15, qe = ln >> 4;
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
            mrow[i] = img < n_img ? (img * TH + ly) * TW + lx : -1;
        }
        epilogue_dispatch<T, MT, NT, true, true>(a, acc, mrow, n0 + wn * NT * 16, qe);
        if (!has_next) break;
        ++t_cur;
        img0 = img_next, n0 = n0_next;
        has_next = t_cur + 1 < t_end;
        if (has_next) { img_next = ((t_cur + 1) / ny) * NI; n0_next = ((t_cur + 1) % ny) * BN; }
    }
}

template <int WM, int WN, int TH, int TW>
static bool launch_s2_patch(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * 128, BN = WN * 64;
    constexpr int NI = BM / (TH * TW), NPIX = NI * ppp_ipix_pad(TH, TW), NPASS = (NPIX + 127) / 128;
    constexpr size_t lds = (size_t)3 * 4 * NPASS * 128 * 16 + (size_t)4 * BN * 64;
    static_assert(lds <= 160 * 1024, "does not fit the LDS");
    if (a.Ho != TH || a.Wo != TW || a.H != 2 * TH || a.W != 2 * TW || a.Cout % BN || a.Cin % 64) return false;
    if ((long)a.M * 4 * a.x_cs * 2 >= (1l << 32) || (long)a.Cout * a.Kp * 2 >= (1l << 32)) return false;   // 32-bit byte strides inside the kernel
    const int n_img = a.M / (a.Ho * a.Wo);
    auto kfn = conv3x3s2_sp_patch_kernel<WM, WN, TH, TW>;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    static const int run_max = [] { const char* e = getenv("AICAM_PPP_RUN"); return e ? std::max(1, atoi(e)) : 6; }();
    const int ny = a.Cout / BN;
    const long ntiles = (long)ceil_div(n_img, NI) * ny;
    int run = 1;
    {
        long best = -1;
        for (int r = 1; r <= run_max; ++r) {
            const long blocks = (ntiles + r - 1) / r, rounds = (blocks + 255) / 256;
            if (r > 1 && rounds < 4) break;
            const long cost = rounds * r;
            if (best < 0 || cost <= best) best = cost, run = r;
        }
    }
    const int nblk = (int)ceil_div(ntiles, (long)run);
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(512), lds, s, a, ny, run);
    KCHECK();
    return true;
}

// The stride-2 shapes (a property of the layer, not of the batch: such a layer is walked in k_order 3 by EVERY kernel): 3x3 / 2 / 1 convs whose
// OUTPUT is one of the patch kernels' maps -- 1: Cout 128 on 32 x 16, 2: Cout % 256 on 16 x 8, 3: on 8 x 4 -- with Cin a multiple of 64.
int conv_s2_patch_shape(const ConvArgs& a) {
    static const bool on = getenv("AICAM_NO_S2D") == nullptr && getenv("AICAM_NO_SP") == nullptr;
    if (!on || a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.Cin % 64 || a.Kp != 9 * a.Cin || a.x2 || a.xs || a.w_tail) return 0;
    if (a.H != 2 * a.Ho || a.W != 2 * a.Wo) return 0;
    if (a.Cout == 128 && a.Ho == 32 && a.Wo == 16) return 1;
    if (a.Cout % 256 == 0 && a.Ho == 16 && a.Wo == 8) return 2;
    // (8 x 4 output maps, ReID layer4.0.conv1: built and measured -- 1 135 against v4's 1 008 us per 15 360 crops; the map is small enough for the
    //  im2col gather to stay in the L2.  Not taken: AICAM_S2D_ALL=1 takes it.)
    static const bool all = getenv("AICAM_S2D_ALL") != nullptr;
    if (all && a.Cout % 256 == 0 && a.Ho == 8 && a.Wo == 4) return 3;
    return 0;
}
bool conv_try_s2_patch(const ConvArgs& a, hipStream_t s) {           // fp16, a batch large enough for one-block-per-CU tiles (the caller's check)
    const int shape = conv_s2_patch_shape(a);
    if (shape == 1) return launch_s2_patch<4, 2, 32, 16>(a, s);
    if (shape == 2) return launch_s2_patch<2, 4, 16, 8>(a, s);
    if (shape == 3) return launch_s2_patch<2, 4, 8, 4>(a, s);
    return false;
}

// shape: conv_pp_patch_shape()'s (2 = Cout 128 on 32 x 16 maps, 3 / 4 = Cout % 256 on 16 x 8 / 8 x 4 maps); the caller has checked that the
// batch is large enough for one-block-per-CU tiles.  AICAM_NO_SP=1: v5 everywhere (A/B).
bool conv_try_sp_patch(const ConvArgs& a, int shape, hipStream_t s) {
    static const bool on = getenv("AICAM_NO_SP") == nullptr;
    if (!on || a.x2 || a.Cin % 128) return false;
    if (shape == 2) return launch_sp_patch<8, 4, 4, 2, 32, 16>(a, s);
    if (shape == 3) return launch_sp_patch<8, 4, 2, 4, 16, 8>(a, s);
    if (shape == 4) return launch_sp_patch<8, 4, 2, 4, 8, 4>(a, s);
    return false;
}

}  // namespace aic
