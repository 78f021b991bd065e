// kernels_pre.hip -- letterbox+normalise (K1) and crop+resize+normalise (K5), HBM-bound u8 work.
//
// Replaces letterbox / preprocess_yolo_input (src/utils/image_processing.py:7-70,73-102) and
// _extract_image_crops + preprocess_reid_input (src/tracker/deepsort_tracker.py:143-159,
// image_processing.py:105-138).  The u8 resize restates cv2.resize INTER_LINEAR for 8-bit images
// (opencv-python 4.11: half-pixel centres, 11-bit fixed-point taps, the 2x2 area fast path when both
// scales are exactly 2) bit for bit the way oracle/image_oracle.py states it -- SURVEY.md §7.1 D5.
// Compiled with -ffp-contract=off: the coordinate arithmetic must round like NumPy's.
//
// One thread per output pixel; every thread reads its <= 4 source pixels straight from the u8
// frame in HBM (rows are contiguous, neighbouring threads read neighbouring bytes) and writes one
// 16-byte NHWC8 pixel (engine input) or three fp32 planes (API parity output).
#include "kernels.hpp"
#include "pre_math.hpp"

#include <cmath>

namespace aic {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// ---- host: geometry of image_processing.py:33-67 (auto=False, scaleup=False as called at :92)
static long py_round(double v) {   // Python round(): half to even
    return (long)std::nearbyint(v);
}

LetterboxGeom letterbox_geometry(int h, int w, int out_h, int out_w) {
    LetterboxGeom g{};
    double rh = (double)out_h / h, rw = (double)out_w / w;
    rh = rh < 1.0 ? rh : 1.0;
    rw = rw < 1.0 ? rw : 1.0;
    const double r = rh < rw ? rh : rw;
    g.src_h = h, g.src_w = w, g.out_h = out_h, g.out_w = out_w;
    g.unpad_h = (int)py_round(h * r);
    g.unpad_w = (int)py_round(w * r);
    const double dw = (out_w - g.unpad_w) / 2.0, dh = (out_h - g.unpad_h) / 2.0;
    g.top = (int)py_round(dh - 0.1);
    g.left = (int)py_round(dw - 0.1);
    // bottom/right = round(d + 0.1); the output canvas is cropped/padded to out_h x out_w
    g.ratio = (float)r;
    g.pad_w = (float)dw;
    g.pad_h = (float)dh;
    return g;
}

// Resample one output pixel (3 channels, BGR order as stored) from a u8 region.
// region origin (x0,y0), size (sw,sh) inside a frame with row pitch `pitch` bytes.
__device__ __forceinline__ void sample_px(const uint8_t* __restrict__ img, int pitch, int x0, int y0, int sw, int sh,
                                          int dx, int dy, int dw, int dh, bool area2, double scale_x, double scale_y,
                                          int out[3]) {
    if (area2) {
        const uint8_t* p0 = img + (size_t)(y0 + 2 * dy) * pitch + (size_t)(x0 + 2 * dx) * 3;
        const uint8_t* p1 = p0 + pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = ((int)p0[c] + (int)p0[3 + c] + (int)p1[c] + (int)p1[3 + c] + 2) >> 2;
        return;
    }
    const Taps tx = taps_x(dx, scale_x, sw);
    const Taps ty = taps_y(dy, scale_y, sh);
    const uint8_t* r0 = img + (size_t)(y0 + ty.i0) * pitch + (size_t)x0 * 3;
    const uint8_t* r1 = img + (size_t)(y0 + ty.i1) * pitch + (size_t)x0 * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[tx.i0 * 3 + c] * tx.w0 + (int)r0[tx.i1 * 3 + c] * tx.w1;
        const int h1 = (int)r1[tx.i0 * 3 + c] * tx.w0 + (int)r1[tx.i1 * 3 + c] * tx.w1;
        out[c] = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
    }
}


template <typename T>
__device__ __forceinline__ void store_nhwc8(T* dst, float r, float g, float b) {
    T o[8] = {(T)r, (T)g, (T)b, (T)0.f, (T)0.f, (T)0.f, (T)0.f, (T)0.f};
    if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<half8*>(dst) = *reinterpret_cast<half8*>(o);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = o[e];
    }
}

// mode 0: fp32 NCHW planes; mode 1: NHWC8 of T
template <typename T>
__global__ void letterbox_kernel(const uint8_t* __restrict__ frames, int n, LetterboxGeom g, int mode, void* out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)g.out_h * g.out_w;
    if (idx >= per * n) return;
    const int img = (int)(idx / per);
    const int p = (int)(idx - (long)img * per);
    const int oy = p / g.out_w, ox = p - oy * g.out_w;
    const int y = oy - g.top, x = ox - g.left;
    int px[3] = {114, 114, 114};
    if (y >= 0 && y < g.unpad_h && x >= 0 && x < g.unpad_w) {
        const uint8_t* f = frames + (size_t)img * g.src_h * g.src_w * 3;
        const double sx = 1.0 / ((double)g.unpad_w / (double)g.src_w);
        const double sy = 1.0 / ((double)g.unpad_h / (double)g.src_h);
        sample_px(f, g.src_w * 3, 0, 0, g.src_w, g.src_h, x, y, g.unpad_w, g.unpad_h,
                  is_area2(g.src_w, g.src_h, g.unpad_w, g.unpad_h), sx, sy, px);
    }
    // BGR -> RGB, /255 in fp32 (image_processing.py:93-99)
    const float r = (float)px[2] / 255.0f, gg = (float)px[1] / 255.0f, b = (float)px[0] / 255.0f;
    if (mode == 0) {
        float* o = reinterpret_cast<float*>(out) + (size_t)img * 3 * per + p;
        o[0] = r; o[per] = gg; o[2 * per] = b;
    } else {
        store_nhwc8<T>(reinterpret_cast<T*>(out) + (size_t)idx * 8, r, gg, b);
    }
}

// letterbox() as an IMAGE (image_processing.py:7-70, every mode: the caller supplies the geometry the mode produces): the resized frame
// (g.unpad_h x g.unpad_w at (g.top, g.left)) inside a g.out_h x g.out_w canvas of `color`, u8 BGR HWC.
__global__ void letterbox_u8_kernel(const uint8_t* __restrict__ frame, LetterboxGeom g, int cb, int cg, int cr, uint8_t* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)g.out_h * g.out_w) return;
    const int oy = (int)(idx / g.out_w), ox = (int)(idx - (long)oy * g.out_w);
    const int y = oy - g.top, x = ox - g.left;
    int px[3] = {cb, cg, cr};
    if (y >= 0 && y < g.unpad_h && x >= 0 && x < g.unpad_w) {
        const double sx = 1.0 / ((double)g.unpad_w / (double)g.src_w);
        const double sy = 1.0 / ((double)g.unpad_h / (double)g.src_h);
        sample_px(frame, g.src_w * 3, 0, 0, g.src_w, g.src_h, x, y, g.unpad_w, g.unpad_h, is_area2(g.src_w, g.src_h, g.unpad_w, g.unpad_h), sx, sy, px);
    }
    out[idx * 3 + 0] = (uint8_t)px[0]; out[idx * 3 + 1] = (uint8_t)px[1]; out[idx * 3 + 2] = (uint8_t)px[2];
}

// ------------------------------------------------------------------------------------------------
// Fused letterbox + YOLOv8 stem (conv 3x3 / stride 2, 3 -> 16, SiLU), fp16.  Unfused, the letterbox writes a
// 6.5 MB NHWC8 canvas per frame that the stem reads straight back (the two are 6 % of the per-frame GPU time, both
// HBM-bound); here a block owns 8 x 32 output pixels: the 17 x 65 letterboxed input pixels it needs are resampled
// from the u8 frame with the very arithmetic of letterbox_kernel (sample_px, /255 in fp32, fp16 rounding) into LDS
// as RGB0, even and odd columns apart (stride 2: 16 consecutive outputs read 16 consecutive entries), and the
// convolution runs on the matrix cores exactly like the ReID stem: K = (tap, RGB0), taps 0..7 in one
// v_mfma_f32_16x16x32_f16, tap 8 in a second one, bias as the accumulator's initial value.
typedef float floatx4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void yolo_stem_fused_kernel(const uint8_t* __restrict__ frames, LetterboxGeom g,
                                                              const half_t* __restrict__ w, const float* __restrict__ bias, int Kp,
                                                              half_t* __restrict__ y, int y_cs, int y_coff, int Ho, int Wo, int tiles_x,
                                                              int tiles_y, unsigned frames_limit) {
    constexpr int TH = 8, TW = 32, PR = 2 * TH + 1, PC = 2 * TW + 1, PCP = (PC + 1) / 2;
    __shared__ uint2 patch[PR * 2 * PCP];      // entry (row, column parity, column / 2)

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 15, q = lane >> 4;
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int img = bx / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int Y0 = 2 * oy0 - 1, X0 = 2 * ox0 - 1;
    const uint8_t* f = frames + (size_t)img * g.src_h * g.src_w * 3;
    const bool area2 = is_area2(g.src_w, g.src_h, g.unpad_w, g.unpad_h);
    // 32-bit byte offsets from the frames pointer rounded down to 4 bytes (fast6: the launcher checked that every frame of the call fits them)
    const uintptr_t fb = reinterpret_cast<uintptr_t>(frames);
    const uint8_t* fa = reinterpret_cast<const uint8_t*>(fb & ~(uintptr_t)3);
    const int pitch = g.src_w * 3;
    const unsigned img_off = (unsigned)(fb & 3) + (unsigned)img * (unsigned)g.src_h * (unsigned)pitch;
    const bool fast6 = frames_limit != 0u;
    const double sx = 1.0 / ((double)g.unpad_w / (double)g.src_w);
    const double sy = 1.0 / ((double)g.unpad_h / (double)g.src_h);

    for (int idx = t; idx < PR * PC; idx += 256) {
        const int pr = idx / PC, pc = idx - pr * PC;
        const int Y = Y0 + pr, X = X0 + pc;
        uint2 v = make_uint2(0u, 0u);                                  // the convolution's zero padding
        if ((unsigned)Y < (unsigned)g.out_h && (unsigned)X < (unsigned)g.out_w) {
            int px[3] = {114, 114, 114};
            const int yy = Y - g.top, xx = X - g.left;
            if (yy >= 0 && yy < g.unpad_h && xx >= 0 && xx < g.unpad_w) {
                // the 2 x 2 area path (1280 x 720 -> 640 x 360): a pixel is two rows of six consecutive bytes -- one aligned 12-byte load per
                // row and a byte alignment instead of twelve byte loads (sample_px's sums on the same bytes: same integers); the byte form
                // where the 12 bytes would reach past the frames handed in
                const unsigned o0 = img_off + (unsigned)(2 * yy) * (unsigned)pitch + (unsigned)(2 * xx) * 3u;
                if (area2 && fast6 && o0 + (unsigned)pitch + 12u <= frames_limit) {
                    int b[2][6];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const unsigned o = o0 + (rr ? (unsigned)pitch : 0u);
                        const uint3 w3 = *reinterpret_cast<const uint3*>(fa + (o & ~3u));
                        const unsigned shb = o & 3u;
                        const unsigned q0 = __builtin_amdgcn_alignbyte(w3.y, w3.x, shb), q1 = __builtin_amdgcn_alignbyte(w3.z, w3.y, shb);
                        b[rr][0] = q0 & 255u, b[rr][1] = (q0 >> 8) & 255u, b[rr][2] = (q0 >> 16) & 255u;
                        b[rr][3] = q0 >> 24, b[rr][4] = q1 & 255u, b[rr][5] = (q1 >> 8) & 255u;
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) px[c] = (b[0][c] + b[0][3 + c] + b[1][c] + b[1][3 + c] + 2) >> 2;
                } else {
                    sample_px(f, g.src_w * 3, 0, 0, g.src_w, g.src_h, xx, yy, g.unpad_w, g.unpad_h, area2, sx, sy, px);
                }
            }
            const half4_t h = {(half_t)((float)px[2] / 255.0f), (half_t)((float)px[1] / 255.0f), (half_t)((float)px[0] / 255.0f), (half_t)0.f};
            v = __builtin_bit_cast(uint2, h);
        }
        patch[(pr * 2 + (pc & 1)) * PCP + (pc >> 1)] = v;
    }

    half8 wa, wb;
    floatx4_t bi;
    {
        const half_t* wr = w + (size_t)r * Kp;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 2 * q + (j >> 2), ci = j & 3;
            wa[j] = ci < 3 ? wr[tap * 8 + ci] : (half_t)0.f;
            wb[j] = (q == 0 && j < 3) ? wr[8 * 8 + j] : (half_t)0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[e] = bias[4 * q + e];
    }
    auto tap_off = [](int tap) { const int kh = tap / 3, kw = tap - 3 * kh; return (kh * 2 + (kw & 1)) * PCP + (kw >> 1); };
    const int off0 = tap_off(2 * q), off1 = tap_off(2 * q + 1), off2 = tap_off(8);
    __syncthreads();

#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int oyl = 2 * wv + (tile >> 1), oxl = (tile & 1) * 16 + r;
        const int base = oyl * 4 * PCP + oxl;
        const uint2 x0 = patch[base + off0], x1 = patch[base + off1], x2 = patch[base + off2];
        const uint4 xa4 = make_uint4(x0.x, x0.y, x1.x, x1.y), xb4 = make_uint4(x2.x, x2.y, 0u, 0u);
        floatx4_t acc = bi;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, __builtin_bit_cast(half8, xa4), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, __builtin_bit_cast(half8, xb4), acc, 0, 0, 0);
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // SiLU = v * sigmoid(v), same form as the conv epilogue (v_exp_f32 + v_rcp_f32)
            const float v = acc[e];
            o[e] = (half_t)(v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)));
        }
        const size_t pix = ((size_t)img * Ho + oy0 + oyl) * Wo + ox0 + oxl;
        *reinterpret_cast<half4_t*>(y + pix * y_cs + y_coff + 4 * q) = o;
    }
}

bool launch_yolo_stem_fused(const uint8_t* frames, int n, const LetterboxGeom& g, const void* w, const float* bias, int Kp, void* y,
                            int y_cs, int y_coff, int Ho, int Wo, hipStream_t s) {
    if (n <= 0) return true;
    if (Ho % 8 || Wo % 32 || Ho * 2 != g.out_h || Wo * 2 != g.out_w || (y_cs | y_coff) % 4) return false;
    const int tiles_x = Wo / 32, tiles_y = Ho / 8;
    // bytes from the 4-byte-aligned base to the end of the call's last frame, or 0 when they do not fit 32 bits (the kernel then reads bytes)
    const unsigned long long total = (unsigned long long)(reinterpret_cast<uintptr_t>(frames) & 3) + (unsigned long long)n * g.src_h * g.src_w * 3;
    const unsigned frames_limit = total < (1ull << 32) - 64 ? (unsigned)total : 0u;
    hipLaunchKernelGGL(yolo_stem_fused_kernel, dim3(n * tiles_x * tiles_y), dim3(256), 0, s, frames, g, reinterpret_cast<const half_t*>(w),
                       bias, Kp, reinterpret_cast<half_t*>(y), y_cs, y_coff, Ho, Wo, tiles_x, tiles_y, frames_limit);
    KCHECK();
    return true;
}

#ifndef CROP_ROWS
#define CROP_ROWS 16
#endif
// One block = CROP_ROWS output rows of one crop (32 / 64 / 128 rows per block: same 1.75 ms per 15 360 crops; staging the source
// rows in LDS: slower, 2.16 ms -- the kernel is bound by its 12 byte-granular taps per pixel either way). The tap tables (fp64 coordinate math of the cv2 spec) are
// computed once per block into LDS -- 64 + 16 entries instead of once per output pixel.
template <typename T>
__global__ __launch_bounds__(256) void crop_resize_kernel(const uint8_t* __restrict__ frames, int fh, int fw, const float* __restrict__ boxes,
                                                          const int* __restrict__ frame_of, int n, const int* __restrict__ n_dev, int oh, int ow,
                                                          int mode, void* out, int* __restrict__ valid, int wide) {
    constexpr int ROWS = CROP_ROWS, MAXW = 256;
    __shared__ Taps xt[MAXW];
    __shared__ Taps yt[ROWS];
    __shared__ float lut[3][256];               // (v/255 - mean)/std for the 256 pixel values: the same fp32 divisions, once per block
    const int crop = blockIdx.y;
    const int row0 = blockIdx.x * ROWS;
    const int live = n_dev ? min(*n_dev, n) : n;
    const int t = threadIdx.x;
    // deepsort_tracker.py:148-153: int() truncation towards zero, then clamp (block-uniform)
    int x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    if (crop < live) {
        const float* b = boxes + (size_t)crop * 4;
        const float lim = 1.0e9f;
        x1 = (int)fminf(fmaxf(b[0], -lim), lim); y1 = (int)fminf(fmaxf(b[1], -lim), lim);
        x2 = (int)fminf(fmaxf(b[2], -lim), lim); y2 = (int)fminf(fmaxf(b[3], -lim), lim);
        x1 = max(0, x1); y1 = max(0, y1); x2 = min(fw, x2); y2 = min(fh, y2);
    }
    const bool ok = crop < live && x1 < x2 && y1 < y2;
    if (t == 0 && blockIdx.x == 0 && valid) valid[crop] = ok ? 1 : 0;
    const int sw = x2 - x1, sh = y2 - y1;
    const bool area2 = ok && is_area2(sw, sh, ow, oh);
    if (ok && !area2) {
        if (t < ow) xt[t] = taps_x(t, 1.0 / ((double)ow / (double)sw), sw);
        const int ty_i = (t + ROWS) % 256;          // the y taps on the threads that have no x tap to compute, where there are such
        if (ty_i < ROWS && row0 + ty_i < oh) yt[ty_i] = taps_y(row0 + ty_i, 1.0 / ((double)oh / (double)sh), sh);
    }
    {
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
        for (int c = 0; c < 3; ++c) lut[c][t] = ((float)t / 255.0f - mean[c]) / stdv[c];      // image_processing.py:126-131 (fp32)
    }
    __syncthreads();
    const int per = oh * ow;
    const int fi = (ok && frame_of) ? frame_of[crop] : 0;
    const uint8_t* f = frames + (size_t)fi * fh * fw * 3;
    const int pitch = fw * 3;
    for (int idx = t; idx < ROWS * ow; idx += 256) {
        const int ry = idx / ow, ox = idx - ry * ow, oy = row0 + ry;
        if (oy >= oh) break;
        float v[3] = {0.f, 0.f, 0.f};
        if (ok) {
            int px[3];
            if (area2) {
                const uint8_t* p0 = f + (size_t)(y1 + 2 * oy) * pitch + (size_t)(x1 + 2 * ox) * 3;
                const uint8_t* p1 = p0 + pitch;
#pragma unroll
                for (int c = 0; c < 3; ++c) px[c] = ((int)p0[c] + (int)p0[3 + c] + (int)p1[c] + (int)p1[3 + c] + 2) >> 2;
            } else if (wide) {
                // both taps of a row are 6 consecutive bytes (3 when the right tap is clamped onto the left one): one aligned
                // 12-byte load per row and v_alignbyte instead of six byte loads -- the kernel is bound by its tap loads
                const Taps tx = xt[ox], ty = yt[ry];
                const bool two = tx.i1 != tx.i0;
                int b0[2][3], b1[2][3];                                 // [row][channel] of the left / right tap
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const uintptr_t A = reinterpret_cast<uintptr_t>(f + (size_t)(y1 + (rr ? ty.i1 : ty.i0)) * pitch + (size_t)(x1 + tx.i0) * 3);
                    const uint3 w = *reinterpret_cast<const uint3*>(A & ~(uintptr_t)3);
                    const unsigned sh = (unsigned)(A & 3);
                    const unsigned q0 = __builtin_amdgcn_alignbyte(w.y, w.x, sh), q1 = __builtin_amdgcn_alignbyte(w.z, w.y, sh);
                    b0[rr][0] = q0 & 255u, b0[rr][1] = (q0 >> 8) & 255u, b0[rr][2] = (q0 >> 16) & 255u;
                    b1[rr][0] = two ? (q0 >> 24) : b0[rr][0], b1[rr][1] = two ? (q1 & 255u) : b0[rr][1], b1[rr][2] = two ? ((q1 >> 8) & 255u) : b0[rr][2];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0 = b0[0][c] * tx.w0 + b1[0][c] * tx.w1;
                    const int h1 = b0[1][c] * tx.w0 + b1[1][c] * tx.w1;
                    px[c] = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
                }
            } else {
                const Taps tx = xt[ox], ty = yt[ry];
                const uint8_t* r0 = f + (size_t)(y1 + ty.i0) * pitch + (size_t)x1 * 3;
                const uint8_t* r1 = f + (size_t)(y1 + ty.i1) * pitch + (size_t)x1 * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0 = (int)r0[tx.i0 * 3 + c] * tx.w0 + (int)r0[tx.i1 * 3 + c] * tx.w1;
                    const int h1 = (int)r1[tx.i0 * 3 + c] * tx.w0 + (int)r1[tx.i1 * 3 + c] * tx.w1;
                    px[c] = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
                }
            }
            // image_processing.py:126-131: BGR->RGB, (x/255 - mean)/std in fp32
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = lut[c][px[2 - c]];
        }
        const int p = oy * ow + ox;
        if (mode == 0) {
            float* o = reinterpret_cast<float*>(out) + (size_t)crop * 3 * per + p;
            o[0] = v[0]; o[per] = v[1]; o[2 * per] = v[2];
        } else if (mode == 2) {   // NHWC4 (RGB0, 8 bytes per pixel): what the fused fp16 ReID stem consumes, half the traffic of NHWC8
            if constexpr (sizeof(T) == 2) {
                const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)0.f};
                *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(out) + ((size_t)crop * per + p) * 4) = h;
            }
        } else {
            store_nhwc8<T>(reinterpret_cast<T*>(out) + ((size_t)crop * per + p) * 8, v[0], v[1], v[2]);
        }
    }
}

void launch_letterbox(const uint8_t* frames, int n, const LetterboxGeom& g, int mode, int dtype, void* out, hipStream_t s) {
    const long tot = (long)n * g.out_h * g.out_w;
    if (tot <= 0) return;
    if (dtype == AIC_F16) hipLaunchKernelGGL(letterbox_kernel<half_t>, dim3(ceil_div(tot, 256)), dim3(256), 0, s, frames, n, g, mode, out);
    else hipLaunchKernelGGL(letterbox_kernel<float>, dim3(ceil_div(tot, 256)), dim3(256), 0, s, frames, n, g, mode, out);
    KCHECK();
}

void launch_letterbox_u8(const uint8_t* frame, const LetterboxGeom& g, const int color_bgr[3], uint8_t* out, hipStream_t s) {
    const long tot = (long)g.out_h * g.out_w;
    if (tot <= 0) return;
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, s, frame, g, color_bgr[0], color_bgr[1], color_bgr[2], out);
    KCHECK();
}

void launch_crop_resize(const uint8_t* frames, int h, int w, const float* boxes, const int* frame_of, int n,
                        const int* n_dev, int out_h, int out_w, int mode, int dtype, void* out, int* valid, hipStream_t s, bool slack) {
    if (n <= 0) return;
    static const bool no_wide = getenv("AICAM_CROP_BYTES") != nullptr;
    const int wide = slack && !no_wide;
    AIC_REQUIRE(out_w <= 240, AIC_ERR_CAPACITY, "crop width above 240 is not supported");
    dim3 grid(ceil_div(out_h, CROP_ROWS), n);
    if (dtype == AIC_F16)
        hipLaunchKernelGGL(crop_resize_kernel<half_t>, grid, dim3(256), 0, s, frames, h, w, boxes, frame_of, n, n_dev, out_h, out_w, mode, out, valid, wide);
    else
        hipLaunchKernelGGL(crop_resize_kernel<float>, grid, dim3(256), 0, s, frames, h, w, boxes, frame_of, n, n_dev, out_h, out_w, mode, out, valid, wide);
    KCHECK();
}

}  // namespace aic
