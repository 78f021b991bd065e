// trk_math.hpp -- device arithmetic of the Kalman filter shared by the tracker kernels (kernels_trk.hip: one launch per
// frame; kernels_trk_dev.hip: association on the device, k frames per launch).  fp32, -ffp-contract=off.
//   src/tracker/core/kalman_filter.py:55-83 (initiate), :85-120 (predict), :122-151 (project), :153-204 (update), :206-249 (gating)
// Noise terms follow NumPy 2.x scalar promotion: std = fp32(weight) * h in fp32, squared in fp64, rounded once.
#pragma once
#include "kernels.hpp"

namespace aic {

__device__ __forceinline__ float sq64(float s) { return (float)((double)s * (double)s); }

#define W_POS 0.05f       /* fp32(1/20)   kalman_filter.py:52  */
#define W_VEL 0.00625f    /* fp32(1/160)  kalman_filter.py:53  */

__device__ __forceinline__ float q_diag(int i, float h) {   // process noise, kalman_filter.py:99-112
    if (i == 2) return (float)(1e-2 * 1e-2);
    if (i == 6) return (float)(1e-5 * 1e-5);
    return sq64((i < 4 ? W_POS : W_VEL) * h);
}
__device__ __forceinline__ float r_diag(int i, float h) {   // measurement noise, kalman_filter.py:136-143
    if (i == 2) return (float)(1e-1 * 1e-1);
    return sq64(W_POS * h);
}

// S = H P H^T + R (4x4, symmetric) and its lower Cholesky factor. Returns false if not PD.
__device__ __forceinline__ void innovation_cov(const float* P, float h, float S[4][4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) S[a][b] = P[a * 8 + b] + (a == b ? r_diag(a, h) : 0.f);
}
template <int N>
__device__ __forceinline__ bool cholesky(const float S[4][4], float L[4][4]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float d = S[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = d - L[j][k] * L[j][k];
        if (!(d > 0.f)) ok = false;
        const float ljj = sqrtf(d);
        L[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            float s = S[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = s - L[i][k] * L[j][k];
            L[i][j] = s / ljj;
        }
    }
    return ok;
}
template <int N>
__device__ __forceinline__ void fwd_solve(const float L[4][4], const float b[4], float y[4]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = s - L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
}
__device__ __forceinline__ void bwd_solve(const float L[4][4], const float y[4], float x[4]) {   // L^T x = y
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        float s = y[i];
#pragma unroll
        for (int k = i + 1; k < 4; ++k) s = s - L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
}


// One wavefront = one track, lane (i, j) = element P[i][j] of the 8x8 covariance (state in global or LDS memory).
// Kalman predict in place (kalman_filter.py:85-120): bit-exact with the reference (F P F^T has at most two non-zero terms per
// element and multi_dot evaluates F (P F^T)).  All loads of a lane precede its stores; one wave = lock-step.
__device__ __forceinline__ void kf_predict_wave(float* P, float* m, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const float h = m[3];
    float t1 = P[i * 8 + j];
    if (j < 4) t1 = t1 + P[i * 8 + j + 4];
    float t2 = t1;
    if (i < 4) {
        float u = P[(i + 4) * 8 + j];
        if (j < 4) u = u + P[(i + 4) * 8 + j + 4];
        t2 = t1 + u;
    }
    if (i == j) t2 = t2 + q_diag(i, h);
    float mi = 0.f;
    if (j == 0) { mi = m[i]; if (i < 4) mi = mi + m[i + 4]; }
    __builtin_amdgcn_s_waitcnt(0);          // every lane has its operands before anyone overwrites them
    P[i * 8 + j] = t2;
    if (j == 0) m[i] = mi;
}

// Kalman update in place with measurement zz (xyah) (kalman_filter.py:153-204); returns the updated mean element i of lane (i, *).
__device__ __forceinline__ float kf_update_wave(float* P, float* m, const float* zz, int lane) {
    const int i = lane >> 3, j = lane & 7;
    float S[4][4], L[4][4];
    innovation_cov(P, m[3], S);
    cholesky<4>(S, L);
    float bi[4], bj[4], y[4], Ki[4], Kj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { bi[a] = P[i * 8 + a]; bj[a] = P[j * 8 + a]; }
    fwd_solve<4>(L, bi, y); bwd_solve(L, y, Ki);
    fwd_solve<4>(L, bj, y); bwd_solve(L, y, Kj);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float u = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) u = u + S[a][c] * Kj[c];
        acc = acc + Ki[a] * u;
    }
    const float pij = P[i * 8 + j] - acc;
    float mi = m[i];
    {
        float dot = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) dot = dot + Ki[a] * (zz[a] - m[a]);
        mi = mi + dot;
    }
    __builtin_amdgcn_s_waitcnt(0);
    P[i * 8 + j] = pij;
    if (j == 0) m[i] = mi;
    return mi;
}

// KalmanFilter.initiate (kalman_filter.py:55-83) for measurement zz (xyah).
__device__ __forceinline__ void kf_initiate_wave(float* P, float* m, const float* zz, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const float h = zz[3];
    float v = 0.f;
    if (i == j) {   // kalman_filter.py:72-82
        if (i == 2) v = (float)(1e-2 * 1e-2);
        else if (i == 6) v = (float)(1e-5 * 1e-5);
        else v = sq64((i < 4 ? 0.1f : 0.0625f) * h);
    }
    P[lane] = v;
    if (j == 0) m[i] = i < 4 ? zz[i] : 0.f;
}

}  // namespace aic
