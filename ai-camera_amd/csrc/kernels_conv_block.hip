// kernels_conv_block.hip -- one kernel for a whole ResNet BasicBlock of 64 channels (ReID layer1: conv3x3+ReLU, conv3x3,
// + input, ReLU).  Run as two convs the block moves its 64-channel intermediate through HBM (1 GB out + 1 GB in per
// 128 frames) and each conv is bound by its output writes; fused, the intermediate lives in LDS.
#include "conv_common.hpp"

namespace aic {

// One 8-wave block per CU streams whole images top to bottom, four rows per step:
//  * waves 0-3 (conv1) turn input rows into intermediate ("mid") rows in an LDS ring, waves 4-7 (conv2) turn mid rows
//    into output rows two steps behind; wave = (column half, output-channel half), 4 rows x 16 pixels x 32 channels and
//    144 MFMAs per step for every wave, the weights of ITS conv resident in 144 VGPRs (as conv3x3_c64_resident_kernel,
//    whose K loop -- input-row fragment reuse, reads pipelined by one group, hand-placed lgkmcnt -- is used unchanged);
//  * a stream has no halo to recompute: images follow each other separated by one all-zero step (the padding rows of
//    both convs), so the only overhead is that step (1/17 for 64-row images);
//  * input rows arrive by LDS-DMA in aligned groups of four rows, three steps ahead of their first use, into a ring of
//    five groups; the mid ring holds four groups (written at step s, read at steps s+1 .. s+3); rows are pixel-major
//    (34 pixels x 128 bytes, chunks XOR-swizzled by the column); the pad columns of the mid ring are zeroed once;
//  * two barriers per step, the conv2 waves half a step behind the conv1 waves (round 5; one barrier per step with both in step
//    before: 1 070 -> 1 125 TFLOP/s on the block alone); everything a half-step reads was written at least one barrier earlier,
//    everything it writes was last read at least one barrier earlier (ring arithmetic and vmcnt accounting at the loop).
struct BlockArgs {
    const void* x; void* y;                 // block input (also the residual) and output, NHWC fp16, W == 32
    const void* w1; const float* b1;        // [64][576] fp16 each, K = (kh, kw, cin)
    const void* w2; const float* b2;
    const void* zero;
    int x_cs, x_coff, y_cs, y_coff, H, n_img, ipb;
    const int* n_dev;                       // optional device-side image count (n_img is then the bound the grid was sized for)
};

__global__ __launch_bounds__(512) void conv3x3_c64_block_kernel(const BlockArgs a) {
    constexpr int PW = 34, ROWB = PW * 128, IN_ROWS = 20, MID_ROWS = 16;
    constexpr int MID_OFF = IN_ROWS * ROWB, DUMMY_OFF = MID_OFF + MID_ROWS * ROWB, BIAS_OFF = DUMMY_OFF + 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), r = lane & 15, q = lane >> 4;
    const int grp = wv >> 2, cx = wv & 1, ch = (wv >> 1) & 1;
    const half_t* __restrict__ xg = reinterpret_cast<const half_t*>(a.x);
    const half_t* zero = reinterpret_cast<const half_t*>(a.zero);
    const half_t* __restrict__ wg = reinterpret_cast<const half_t*>(grp ? a.w2 : a.w1);
    const float* bg = grp ? a.b2 : a.b1;

    // ---- this wave's conv, its 32 output channels: A fragments for K-step s2 = (tap, channel half), k = 32 s2 + 8 q
    half8 wreg[18][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const half_t* wr = wg + (size_t)(32 * ch + perm_row<2>(j, r)) * 576 + 8 * q;
#pragma unroll
        for (int s2 = 0; s2 < 18; ++s2) wreg[s2][j] = *reinterpret_cast<const half8*>(wr + 32 * s2);
    }
#pragma unroll
    for (int s2 = 0; s2 < 18; ++s2) asm volatile("" :: "v"(wreg[s2][0]), "v"(wreg[s2][1]));    // landed before the loop
    float* bias_l = reinterpret_cast<float*>(smem + BIAS_OFF) + ((grp * 2 + ch) * 4 + q) * 8;
    if (cx == 0 && r == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias_l[4 * j + e] = bg[32 * ch + perm_ch<2>(j, q, e)];
    }
    for (int o = t * 16; o < MID_ROWS * ROWB; o += 512 * 16) *reinterpret_cast<uint4*>(smem + MID_OFF + o) = make_uint4(0u, 0u, 0u, 0u);

    const int gpi = a.H / 4 + 1;                        // row groups per image: H/4 real ones and the separator
    // device-side image count (the grid was sized for a bound): the live images are re-dealt over ALL blocks of the grid
    const int n_live = a.n_dev ? min(a.n_img, a.n_dev[0]) : a.n_img;
    const int ipb = a.n_dev ? (n_live + (int)gridDim.x - 1) / (int)gridDim.x : a.ipb;
    const int img0 = blockIdx.x * ipb;
    const int n_loc = max(0, min(ipb, n_live - img0));
    const int S_tot = gpi * n_loc;

    // ---- LDS-DMA of row group g (rows 4g .. 4g+3 of this block's stream) into ring slot g mod 5: 17 wave-instructions of
    // 64 x 16 bytes cover the 4 x 34 x 8 chunks exactly; every wave issues three (the surplus ones hit a dummy page) so
    // that the counted waits see the same number of operations on every wave
    int doff[3];                                // element offset of this lane's chunk inside a row group, or -1 (pad column / surplus instruction)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int sl = (p * 8 + wv) * 64 + lane;
        const int rr = sl / 272, rem = sl - rr * 272;
        const int px = rem >> 3, c = (rem & 7) ^ (px & 7);              // slot of a pixel holds channel chunk slot ^ (column & 7)
        doff[p] = (p * 8 + wv < 17 && (unsigned)(px - 1) < 32u) ? (rr * 32 + (px - 1)) * a.x_cs + a.x_coff + c * 8 : -1;
    }
    auto issue = [&](int g) {
        int gs = g % 5; if (gs < 0) gs += 5;
        const int jl = g >= 0 ? g / gpi : 0, gy = g - jl * gpi;
        const bool real = g >= 0 && jl < n_loc && gy < gpi - 1;
        const half_t* gbase = xg + ((size_t)(img0 + jl) * a.H + gy * 4) * 32 * a.x_cs;     // (block-uniform)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int j = p * 8 + wv;
            const half_t* src = (real && doff[p] >= 0) ? gbase + doff[p] : zero;
            asm volatile("" : "+v"(src));
            char* dst = j < 17 ? smem + gs * 4 * ROWB + j * 1024 : smem + DUMMY_OFF;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        }
    };

    int xk[3];                                  // lane part of a fragment address: column cx 16 + r + kw, channel chunk q
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = cx * 16 + r + kw;
        xk[kw] = (px * 8 + (q ^ (px & 7))) * 16;
    }
    const int px1 = cx * 16 + r + 1;            // this lane's own pixel in a ring row (column 0 is the left pad)
    const int mid_w = (px1 * 8 + ((4 * ch + q) ^ (px1 & 7))) * 16;

    half_t* yg = reinterpret_cast<half_t*>(a.y);
    const unsigned yl0 = (unsigned)((((cx * 16 + r) * a.y_cs) + a.y_coff + 32 * ch + 8 * q) * 2);
    const unsigned rl0 = (unsigned)((((cx * 16 + r) * a.x_cs) + a.x_coff + 32 * ch + 8 * q) * 2);
    const unsigned ystep = (unsigned)(32 * a.y_cs * 2), rstep = (unsigned)(32 * a.x_cs * 2);

    floatx4 acc[4][2];
    half8 rv[4];
    half8 xf[6];
    int rowb[6];                                // LDS byte offset of the six ring rows this step reads (block-uniform)
    auto xread = [&xf, &xk, &rowb](auto g, auto ir) {
        constexpr int gg = decltype(g)::value, ii = decltype(ir)::value, kw = gg >> 1;
        const int ad = ((gg & 1) ? (xk[kw] ^ 64) : xk[kw]) + rowb[ii];      // channel half 1: chunk q + 4 = slot ^ 4
        asm volatile("ds_read_b128 %0, %1" : "=v"(xf[ii]) : "v"(ad));
    };
    auto xwait = [&xf](auto n, auto ir) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[decltype(ir)::value]) : "n"(decltype(n)::value));
    };
    auto kfirst = [&]() {                       // the first group's six row fragments requested, the accumulators start from the bias
        static_for<6>([&](auto ir) { xread(std::integral_constant<int, 0>{}, ir); });
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = *reinterpret_cast<const floatx4*>(bias_l + 4 * j);
    };
    auto kpart = [&](auto part) {               // (tap column, channel half) groups 3 part .. 3 part + 2, six row fragments each: 72 MFMAs; see conv3x3_c64_resident_kernel
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
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // ---- the pieces of a step
    auto rows1 = [&](int s) {                                           // conv1: ring rows of input stream rows 4s - 1 .. 4s + 4
#pragma unroll
        for (int ir = 0; ir < 6; ++ir) {
            const int v = 4 * s - 1 + ir;                               // stream row; group v >> 2 (arithmetic: -1 -> group -1), ring slot mod 5
            int gsl = (v >> 2) % 5; if (gsl < 0) gsl += 5;
            rowb[ir] = (gsl * 4 + (v & 3)) * ROWB;
        }
    };
    auto store_mid = [&](int s) {                                       // conv1: bias + ReLU, fp16, into the mid ring
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act_fast<2>(acc[i][e >> 2][e & 3]);
            const half8 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
            const int ad = MID_OFF + ((4 * s + i) & 15) * ROWB + mid_w;
            asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(o) : "memory");
        }
    };
    auto zero_mid = [&](int s) {                                        // separator rows of the mid stream: zeros
        const half8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ad = MID_OFF + ((4 * s + i) & 15) * ROWB + mid_w;
            asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(o) : "memory");
        }
    };
    auto load_res = [&](size_t pix0) {                                  // conv2: the block input of its four output rows (the residual)
        const char* rb = reinterpret_cast<const char*>(xg) + pix0 * a.x_cs * 2;
        unsigned rl = rl0;
        asm volatile("" : "+v"(rl));            // form the four row offsets here, not as loop-invariant VGPRs
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rv[i]) : "v"(rl + i * rstep), "s"(rb) : "memory");
    };
    auto rows2 = [&](int tt) {
#pragma unroll
        for (int ir = 0; ir < 6; ++ir) rowb[ir] = MID_OFF + ((4 * tt - 1 + ir) & 15) * ROWB;
    };
    auto store_out = [&](size_t pix0) {                                 // conv2: + input, ReLU, fp16, out
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]) :: "memory");   // residual landed; this step's row group may stay in flight
        char* yb = reinterpret_cast<char*>(yg) + pix0 * a.y_cs * 2;
        unsigned yl = yl0;
        asm volatile("" : "+v"(yl));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act_fast<2>(acc[i][e >> 2][e & 3] + (float)rv[i][e]);
            const half8 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
            *reinterpret_cast<half8*>(yb + (size_t)(yl + i * ystep)) = o;
        }
    };

    issue(-1); issue(0); issue(1); issue(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // bias and the zeroed mid ring are in LDS before the first barrier
    // Two barriers per step, the conv2 waves HALF A STEP behind the conv1 waves (they pass one extra barrier first, the conv1 waves
    // one extra at the end).  A SIMD holds one wave of each conv: in step with each other both reach their epilogue, the barrier,
    // the LDS-DMA issue and the first six fragment reads together and the matrix pipe idles; half a step apart those sit under the
    // partner's 72-MFMA half (the weights-resident kernel's schedule, kernels_conv_direct.hip).  Barrier 2s: conv1 starts step s;
    // barrier 2s + 1: conv2 starts its step s (mid rows of step s - 2).  Ring hazards: the mid rows conv1 writes at the end of
    // half-step 2s + 1 replace rows conv2 last read in its step s - 1 (half-steps 2s - 1, 2s), and are first read in conv2's step
    // s + 1 (from 2s + 3); an input row group issued in step s (by conv1 behind barrier 2s, by conv2 behind 2s + 1) replaces the group
    // conv1 last read in step s - 1 and must have landed before barrier 2s + 4 -- conv2's mid-step barrier of its step s + 1.
    // vmcnt, oldest first, at that wait of conv2's step s: group s + 2 | stores of step s - 1 | residual of step s | group s + 3 --
    // 3, 7 or 11 may stay in flight according to which of the two steps are live ones (a count that assumed absent operations would not wait).
    if (grp == 0) {
        for (int s = 0; s < S_tot + 2; ++s) {
            wait_vmcnt<3>();                                            // group s + 1 | group s + 2
            __builtin_amdgcn_s_barrier();
            issue(s + 3);
            const bool live = s < S_tot, act1 = live && s % gpi < gpi - 1;
            if (act1) {
                rows1(s);
                kfirst();
                kpart(P0{});
            }
            __builtin_amdgcn_s_barrier();
            if (live) {
                if (act1) {
                    kpart(P1{});
                    store_mid(s);
                } else zero_mid(s);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the rows are in LDS before the next barrier
            }
        }
        __builtin_amdgcn_s_barrier();
    } else {
        wait_vmcnt<3>();
        __builtin_amdgcn_s_barrier();
        bool prev = false;
        for (int s = 0; s < S_tot + 2; ++s) {
            __builtin_amdgcn_s_barrier();
            const int tt = s - 2;
            const bool act2 = tt >= 0 && tt < S_tot && (tt % gpi) < gpi - 1;
            const int jl = act2 ? tt / gpi : 0, gy = act2 ? tt % gpi : 0;
            const size_t pix0 = ((size_t)(img0 + jl) * a.H + gy * 4) * 32;
            if (act2) load_res(pix0);
            issue(s + 3);
            if (act2) {
                rows2(tt);
                kfirst();
                kpart(P0{});
            }
            if (act2 && prev) wait_vmcnt<11>();
            else if (act2 || prev) wait_vmcnt<7>();
            else wait_vmcnt<3>();
            __builtin_amdgcn_s_barrier();
            if (act2) {
                kpart(P1{});
                store_out(pix0);
            }
            prev = act2;
        }
    }
    wait_vmcnt<0>();
}

bool conv_try_c64_block(const ConvArgs& c1, const ConvArgs& c2, hipStream_t s) {
    static const bool on = [] { const char* e = getenv("AICAM_C64_BLOCK"); return !e || atoi(e) != 0; }();
    auto conv64 = [](const ConvArgs& a) {
        return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 64 && a.Cout == 64 && !a.out_f32 && a.Kp == 576 && a.act == 2 &&
               a.Ho == a.H && a.Wo == a.W && (a.x_cs | a.x_coff | a.y_cs | a.y_coff) % 8 == 0;
    };
    if (!on || !conv64(c1) || !conv64(c2) || c1.res_mode != 0 || c2.res_mode != 1) return false;
    if (c1.W != 32 || c1.H % 4 || c1.H != c2.H || c2.W != 32 || c1.M != c2.M || c1.M < 1500000) return false;
    if (c2.x != c1.y || c2.x_cs != c1.y_cs || c2.x_coff != c1.y_coff) return false;            // conv2 reads what conv1 writes
    if (c2.res != c1.x || c2.r_cs != c1.x_cs || c2.r_coff != c1.x_coff) return false;        // and adds the block input
    if ((long)c1.M * c1.x_cs >= (1l << 31) || (long)c2.M * c2.y_cs >= (1l << 31)) return false;
    BlockArgs a{};
    a.x = c1.x, a.y = c2.y, a.w1 = c1.w, a.b1 = c1.bias, a.w2 = c2.w, a.b2 = c2.bias, a.zero = c1.zero;
    a.x_cs = c1.x_cs, a.x_coff = c1.x_coff, a.y_cs = c2.y_cs, a.y_coff = c2.y_coff, a.H = c1.H;
    a.n_img = c1.M / (c1.H * c1.W);
    a.n_dev = c1.n_dev;
    // images per block: n_img / 256 = one persistent block per CU, or AICAM_BLK_IPB for shorter-lived blocks (each pays 3 steps
    // of pipeline fill and a weight reload; in exchange the CU takes other streams' waiting blocks in between)
    // Default 16 images per block (960 blocks for a 15 360-crop group): same box, interleaved, round 3: 9 490 / 9 482 frames/s with one
    // persistent block per CU (0), 9 553 / 9 624 with 16 -- the NMS blocks of the side stream hold whole CUs for ~2 ms at the start of
    // the ReID trunk, and a persistent block that starts late finishes late; shorter-lived blocks just flow around them.
    // Round 5: 960 blocks are 3.75 rounds of 256 CUs -- the last round runs three quarters full.  12 .. 16 images per block, whichever
    // wastes least of the last round (15 360 crops: 12 -> 1 280 blocks = 5 rounds; the layer alone 2 065 -> 2 000 us per conv, same box).
    static const int ipb_env = [] { const char* e = getenv("AICAM_BLK_IPB"); return e ? atoi(e) : -1; }();
    // (the CUs the blocks are dealt over: the budget leaves ONE out for the tracker's epoch block while it runs, which does not make 1 280
    //  blocks six rounds -- counted against the whole chip)
    const int cus = (conv_cu_budget() + 7) / 8 * 8;
    if (ipb_env > 0) a.ipb = ipb_env;
    else if (ipb_env == 0) a.ipb = (a.n_img + std::min(cus, a.n_img) - 1) / std::min(cus, a.n_img);
    else {
        long best = -1;
        for (int ipb = 12; ipb <= 16; ++ipb) {
            const long blocks = (a.n_img + ipb - 1) / ipb, rounds = (blocks + cus - 1) / cus;
            const long waste = (rounds * cus - blocks) * 1000 / (rounds * cus);           // idle share of the CUs' block slots, per mille
            if (best < 0 || waste < best) best = waste, a.ipb = ipb;
        }
    }
    constexpr size_t lds = (size_t)(20 + 16) * 34 * 128 + 1024 + 512;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(conv3x3_c64_block_kernel, dim3((a.n_img + a.ipb - 1) / a.ipb), dim3(512), lds, s, a);
    KCHECK();
    return true;
}

}  // namespace aic
