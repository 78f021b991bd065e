// pre_math.hpp -- the u8 bilinear-resize tap arithmetic shared by kernels_pre.hip (letterbox, crop + resize) and the fused
// crop + ReID stem of kernels_conv_direct.hip: cv2.resize INTER_LINEAR for 8-bit images as oracle/image_oracle.py states it
// (opencv-python 4.11: half-pixel centres, 11-bit fixed-point taps, the 2x2 area fast path when both scales are exactly 2).
// -ffp-contract=off: the coordinate arithmetic must round like NumPy's.
#pragma once

namespace aic {

struct Taps { int i0, i1, w0, w1; };

// cv2 horizontal taps: clamp with the weight forced onto the surviving tap
__device__ __forceinline__ Taps taps_x(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    Taps t;
    t.i0 = s;
    t.i1 = min(s + 1, n - 1);
    t.w0 = __float2int_rn((1.f - f) * 2048.f);
    t.w1 = __float2int_rn(f * 2048.f);
    return t;
}
// cv2 vertical taps: rows clipped, weights kept
__device__ __forceinline__ Taps taps_y(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f = f - (float)s;
    Taps t;
    t.i0 = min(max(s, 0), n - 1);
    t.i1 = min(max(s + 1, 0), n - 1);
    t.w0 = __float2int_rn((1.f - f) * 2048.f);
    t.w1 = __float2int_rn(f * 2048.f);
    return t;
}

__device__ __forceinline__ bool is_area2(int sw, int sh, int dw, int dh) { return sw == 2 * dw && sh == 2 * dh; }

}  // namespace aic
