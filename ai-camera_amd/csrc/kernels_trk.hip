// kernels_trk.hip -- batched Kalman filter (K7, K8, K12), IoU cost (K10), cosine-min over the
// per-track feature galleries (K9).  fp32 throughout, -ffp-contract=off.
//
// Reference arithmetic (all fp32, per track / per pair in Python loops):
//   KalmanFilter.initiate/predict/project/update/gating_distance
//       src/tracker/core/kalman_filter.py:55-83, 85-120, 122-151, 153-204, 206-249
//   iou / iou_cost                      src/tracker/core/matching.py:13-106
//   cosine_distance / appearance_cost   src/tracker/core/matching.py:109-217
//
// Noise terms follow NumPy 2.x scalar promotion exactly: std = fp32(weight) * h in
// fp32, squared in fp64, rounded to fp32 once.  predict is bit-exact with the reference
// (F P F^T has at most two non-zero terms per element and multi_dot evaluates F (P F^T));
// project/update/gating go through a 4x4 Cholesky like LAPACK's potrf + trtrs / potrs and agree
// to fp32 rounding.
//
// Layout: state SoA in HBM, mean[slot][8], cov[slot][64]; one wavefront (64 lanes = the 8x8
// covariance) per track for predict/update, one thread per (track, detection) pair for gating/IoU.
#include "kernels.hpp"
#include "trk_math.hpp"

namespace aic {

__global__ void kf_initiate_kernel(const float* __restrict__ z, int n, float* mean, float* cov, const int* slots,
                                   const int* zidx) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
    const int slot = slots ? slots[k] : k;
    const float* zz = z + (size_t)(zidx ? zidx[k] : k) * 4;
    const float h = zz[3];
    float v = 0.f;
    if (i == j) {   // kalman_filter.py:72-82
        if (i == 2) v = (float)(1e-2 * 1e-2);
        else if (i == 6) v = (float)(1e-5 * 1e-5);
        else v = sq64((i < 4 ? 0.1f : 0.0625f) * h);
    }
    cov[(size_t)slot * 64 + lane] = v;
    if (j == 0) mean[(size_t)slot * 8 + i] = i < 4 ? zz[i] : 0.f;
}

// dt: KalmanFilter(dt) (kalman_filter.py:34-44), fp32 like the reference's motion matrix; x * 1.0f == x, so dt = 1 keeps its bits
__global__ void kf_predict_kernel(float* mean, float* cov, const int* slots, int n, float dt) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
    const int slot = slots ? slots[k] : k;
    float* P = cov + (size_t)slot * 64;
    float* m = mean + (size_t)slot * 8;
    const float h = m[3];
    // T1 = P F^T ; T2 = F T1   (np.linalg.multi_dot picks F (P F^T) for equal cost)
    float t1 = P[i * 8 + j];
    if (j < 4) t1 = t1 + P[i * 8 + j + 4] * dt;
    float t2 = t1;
    if (i < 4) {
        float u = P[(i + 4) * 8 + j];
        if (j < 4) u = u + P[(i + 4) * 8 + j + 4] * dt;
        t2 = t1 + u * dt;
    }
    if (i == j) t2 = t2 + q_diag(i, h);
    float mi = 0.f;
    if (j == 0) { mi = m[i]; if (i < 4) mi = mi + m[i + 4] * dt; }
    P[i * 8 + j] = t2;          // same wavefront: every load above has retired before these stores
    if (j == 0) m[i] = mi;
}

__global__ void kf_project_kernel(const float* __restrict__ mean, const float* __restrict__ cov, int n, float* pmean, float* pcov) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* P = cov + (size_t)k * 64;
    const float* m = mean + (size_t)k * 8;
    float S[4][4];
    innovation_cov(P, m[3], S);
    for (int a = 0; a < 4; ++a) {
        pmean[(size_t)k * 4 + a] = m[a];
        for (int b = 0; b < 4; ++b) pcov[(size_t)k * 16 + a * 4 + b] = S[a][b];
    }
}

// grid.x = tracks, threads over measurements.
__global__ void kf_gating_kernel(const float* __restrict__ mean, const float* __restrict__ cov, const int* __restrict__ slots,
                                 int n, const float* __restrict__ zs, int m, int shared_z, int only_position, float* d2) {
    const int t = blockIdx.x;
    if (t >= n) return;
    const int slot = slots ? slots[t] : t;
    const float* P = cov + (size_t)slot * 64;
    const float* mu = mean + (size_t)slot * 8;
    float S[4][4], L[4][4];
    innovation_cov(P, mu[3], S);
    const bool ok = only_position ? cholesky<2>(S, L) : cholesky<4>(S, L);
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        const float* z = zs + (size_t)(shared_z ? j : (size_t)t * m + j) * 4;
        float d[4], y[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) d[a] = z[a] - mu[a];
        float acc;
        if (only_position) {
            fwd_solve<2>(L, d, y);
            acc = y[0] * y[0] + y[1] * y[1];
        } else {
            fwd_solve<4>(L, d, y);
            acc = y[0] * y[0];
            acc = acc + y[1] * y[1];
            acc = acc + y[2] * y[2];
            acc = acc + y[3] * y[3];
        }
        d2[(size_t)t * m + j] = ok ? acc : __builtin_inff();   // kalman_filter.py:241-247
    }
}

// One wavefront per (track, measurement) pair: lane (i,j) owns P[i][j].
__global__ void kf_update_kernel(float* mean, float* cov, const int* slots, const float* __restrict__ z,
                                 const int* __restrict__ zidx, int n, float* out_tlwh) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
    const int slot = slots ? slots[k] : k;
    float* P = cov + (size_t)slot * 64;
    float* m = mean + (size_t)slot * 8;
    const float* zz = z + (size_t)(zidx ? zidx[k] : k) * 4;
    float S[4][4], L[4][4];
    innovation_cov(P, m[3], S);
    cholesky<4>(S, L);
    // rows i and j of K = (S^-1 (P H^T)^T)^T, kalman_filter.py:185-190
    float bi[4], bj[4], y[4], Ki[4], Kj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { bi[a] = P[i * 8 + a]; bj[a] = P[j * 8 + a]; }
    fwd_solve<4>(L, bi, y); bwd_solve(L, y, Ki);
    fwd_solve<4>(L, bj, y); bwd_solve(L, y, Kj);
    // P - K (S K^T), kalman_filter.py:201-202
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float u = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) u = u + S[a][b] * Kj[b];
        acc = acc + Ki[a] * u;
    }
    const float pij = P[i * 8 + j] - acc;
    // mean + K (z - H mean), kalman_filter.py:193-196
    float mi = m[i];
    {
        float dot = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) dot = dot + Ki[a] * (zz[a] - m[a]);
        mi = mi + dot;
    }
    P[i * 8 + j] = pij;
    if (j == 0) m[i] = mi;
    if (out_tlwh) {   // Track.to_tlwh of the updated state, track.py:133-151
        const float cx = __shfl(mi, 0), cy = __shfl(mi, 8), ar = __shfl(mi, 16), hh = __shfl(mi, 24);
        if (lane == 0) {
            float w = 0.f, h2 = hh;
            if (hh > 0.f) w = ar * hh; else h2 = fmaxf(0.f, hh);
            float* o = out_tlwh + (size_t)k * 4;
            o[0] = cx - w / 2.0f; o[1] = cy - h2 / 2.0f; o[2] = w; o[3] = h2;
        }
    }
}

// 1 - IoU in tlwh (matching.py:13-106). Track boxes either given or derived from the state means.
__global__ void iou_cost_kernel(const float* __restrict__ trk_tlwh, const float* __restrict__ mean, const int* __restrict__ slots,
                                int t, const float* __restrict__ det, int n, float* cost) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= t * n) return;
    const int ti = idx / n, di = idx - ti * n;
    float bx, by, bw, bh;
    if (trk_tlwh) {
        bx = trk_tlwh[ti * 4], by = trk_tlwh[ti * 4 + 1], bw = trk_tlwh[ti * 4 + 2], bh = trk_tlwh[ti * 4 + 3];
    } else {
        const float* m = mean + (size_t)(slots ? slots[ti] : ti) * 8;
        float w = 0.f, h = m[3];
        if (h > 0.f) w = m[2] * h; else h = fmaxf(0.f, h);
        bx = m[0] - w / 2.0f; by = m[1] - h / 2.0f; bw = w; bh = h;
    }
    const float* c = det + (size_t)di * 4;
    const float brx = bx + bw, bry = by + bh, crx = c[0] + c[2], cry = c[1] + c[3];
    const float iw = fmaxf(0.f, fminf(brx, crx) - fmaxf(bx, c[0]));
    const float ih = fmaxf(0.f, fminf(bry, cry) - fmaxf(by, c[1]));
    const float inter = iw * ih;
    const float uni = bw * bh + c[2] * c[3] - inter;
    cost[idx] = 1.0f - inter / fmaxf(uni, 1e-7f);
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// rows / max(||row||, 1e-7)  (matching.py:126-130); one wavefront per row.
__global__ void normalize_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int dim, const int* __restrict__ n_dev) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (n_dev ? min(n, n_dev[0]) : n)) return;
    const int lane = threadIdx.x & 63;
    const float* s = src + (size_t)row * dim;
    float ss = 0.f;
    for (int c = lane; c < dim; c += 64) ss = ss + s[c] * s[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss = ss + __shfl_xor(ss, o);
    const float nrm = fmaxf(sqrtf(ss), 1e-7f);
    for (int c = lane; c < dim; c += 64) dst[(size_t)row * dim + c] = s[c] / nrm;
}

// cost[t][n] = min over gallery rows g of max(0, 1 - <gal_n[g], det_n[n]>)   (matching.py:136-141,207)
// grid (tracks, gallery chunks of 16 rows); each wave normalises 4 gallery rows in registers, then
// streams the (L2-resident) normalised detection features past them. atomicMin on the fp32 bit
// pattern is order-preserving because every value is >= +0.
template <int NPER>
__global__ __launch_bounds__(256) void cosine_min_kernel(const float* __restrict__ gal, const int* __restrict__ slots,
                                                         const int* __restrict__ glen, int gmax, int dim,
                                                         const float* __restrict__ det_n, const unsigned char* __restrict__ has_feat,
                                                         int n, float* cost) {
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g0 = blockIdx.y * 16 + wv * 4;
    const int len = glen[t];
    if (g0 >= len) return;
    const float* base = gal + ((size_t)slots[t] * gmax) * dim;
    float a[4][NPER];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool live = g0 + r < len;
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < NPER; ++k) {
            const int c = k * 64 + lane;
            const float v = (live && c < dim) ? base[(size_t)(g0 + r) * dim + c] : 0.f;
            a[r][k] = v;
            ss = ss + v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss = ss + __shfl_xor(ss, o);
        const float nrm = fmaxf(sqrtf(ss), 1e-7f);
#pragma unroll
        for (int k = 0; k < NPER; ++k) a[r][k] = a[r][k] / nrm;
    }
    unsigned int* out = reinterpret_cast<unsigned int*>(cost) + (size_t)t * n;
    for (int d = 0; d < n; ++d) {
        if (has_feat && !has_feat[d]) continue;
        float b[NPER];
#pragma unroll
        for (int k = 0; k < NPER; ++k) {
            const int c = k * 64 + lane;
            b[k] = c < dim ? det_n[(size_t)d * dim + c] : 0.f;
        }
        float dot[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NPER; ++k) s = s + a[r][k] * b[k];
            dot[r] = s;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dot[r] = dot[r] + __shfl_xor(dot[r], o);
        }
        if (lane == 0) {
            float best = 3.0e38f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (g0 + r < len) {
                    float dist = 1.0f - dot[r];
                    dist = dist > 0.f ? dist : 0.f;
                    best = fminf(best, dist);
                }
            }
            atomicMin(out + d, __float_as_uint(best));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused per-frame kernels of the tracker (one stream, three launches per frame).
//
// trk_assoc_kernel: block t = one live track. Wave 0 runs the Kalman predict in place
// (kalman_filter.py:85-120), then all threads walk the detections: squared Mahalanobis distance
// (kalman_filter.py:206-249), 1-IoU (matching.py:13-106) and the INFTY_COST initialisation of the
// appearance row (matching.py:173).
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(128) void trk_assoc_kernel(float* mean, float* cov, const int* __restrict__ slots, int do_predict,
                                                        const float* __restrict__ det_tlwh, const float* __restrict__ det_xyah,
                                                        int n, float* app, float* d2, float* iouc) {
    const int t = blockIdx.x;
    const int slot = slots[t];
    float* P = cov + (size_t)slot * 64;
    float* m = mean + (size_t)slot * 8;
    if (do_predict && threadIdx.x < 64) {
        const int lane = threadIdx.x, i = lane >> 3, j = lane & 7;
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
        P[i * 8 + j] = t2;
        if (j == 0) m[i] = mi;
    }
    __syncthreads();   // the block's own global writes are visible to its other waves after the barrier
    float S[4][4], L[4][4];
    innovation_cov(P, m[3], S);
    const bool ok = cholesky<4>(S, L);
    float bw = 0.f, bh = m[3];
    if (bh > 0.f) bw = m[2] * bh; else bh = fmaxf(0.f, bh);
    const float bx = m[0] - bw / 2.0f, by = m[1] - bh / 2.0f;
    const float brx = bx + bw, bry = by + bh;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const float* z = det_xyah + (size_t)j * 4;
        float d[4], y[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) d[a] = z[a] - m[a];
        fwd_solve<4>(L, d, y);
        float acc = y[0] * y[0];
        acc = acc + y[1] * y[1];
        acc = acc + y[2] * y[2];
        acc = acc + y[3] * y[3];
        const size_t o = (size_t)t * n + j;
        d2[o] = ok ? acc : __builtin_inff();
        const float* c = det_tlwh + (size_t)j * 4;
        const float crx = c[0] + c[2], cry = c[1] + c[3];
        const float iw = fmaxf(0.f, fminf(brx, crx) - fmaxf(bx, c[0]));
        const float ih = fmaxf(0.f, fminf(bry, cry) - fmaxf(by, c[1]));
        const float inter = iw * ih;
        const float uni = bw * bh + c[2] * c[3] - inter;
        iouc[o] = 1.0f - inter / fmaxf(uni, 1e-7f);
        app[o] = 1e5f;
    }
}

// cosine_min on the matrix cores: cost[t][n] = min_g max(0, 1 - <gal_n[t][g], det_n[n]>) with both
// operands ALREADY normalised (gallery rows at append time, detections once per launch group).
// v_mfma_f32_16x16x4_f32: exact fp32 fmaf chain. One wave = 16 gallery rows x up to 32 detections;
// lane (r, q) streams 16 bytes of its row per 16-deep K slice and MFMA i consumes element i of both
// operands (k = 16*kb + 4*q + i on both sides, so the contraction index matches).
__global__ __launch_bounds__(256) void cosine_min_mfma_kernel(const float* __restrict__ gal_n, const int* __restrict__ slots,
                                                              const int* __restrict__ glen, int gmax, int dim,
                                                              const float* __restrict__ det_n, const unsigned char* __restrict__ has_feat,
                                                              int n, float* cost) {
    const int t = blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int g0 = (blockIdx.y * 4 + wv) * 16;
    const int len = glen[t];
    if (g0 >= len) return;
    // rows / detections past the end are CLAMPED to a valid row instead of zero-filled: their products
    // are never read (the min below and the atomics are guarded), and the loads stay branch-free
    const float* grow = gal_n + ((size_t)slots[t] * gmax + min(g0 + r, len - 1)) * dim;
    unsigned int* out = reinterpret_cast<unsigned int*>(cost) + (size_t)t * n;
    for (int d0 = 0; d0 < n; d0 += 32) {
        const int da = d0 + r, db = d0 + 16 + r;
        const float* pa = det_n + (size_t)min(da, n - 1) * dim;
        const float* pb = det_n + (size_t)min(db, n - 1) * dim;
        const bool oka = da < n, okb = db < n;
        floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int k0 = 0;
        for (; k0 + 64 <= dim; k0 += 64) {          // 12 independent 16-byte loads in flight per lane
            floatx4 a[4], b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 16 * u + 4 * q;
                a[u] = *reinterpret_cast<const floatx4*>(grow + k);
                b0[u] = *reinterpret_cast<const floatx4*>(pa + k);
                b1[u] = *reinterpret_cast<const floatx4*>(pb + k);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b0[u][e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b1[u][e], acc1, 0, 0, 0);
                }
        }
        for (; k0 < dim; k0 += 16) {                 // tail: any dimension, element-guarded
            const int k = k0 + 4 * q;
            floatx4 a = {0.f, 0.f, 0.f, 0.f}, b0 = a, b1 = a;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k + e < dim) { a[e] = grow[k + e]; b0[e] = pa[k + e]; b1[e] = pb[k + e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc1, 0, 0, 0);
            }
        }
        // D[row = gallery 4q+e][col = detection r]: min over the valid gallery rows of this tile
        float m0 = 3.0e38f, m1 = 3.0e38f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (g0 + 4 * q + e < len) {
                float x0 = 1.0f - acc0[e], x1 = 1.0f - acc1[e];
                x0 = x0 > 0.f ? x0 : 0.f;
                x1 = x1 > 0.f ? x1 : 0.f;
                m0 = fminf(m0, x0);
                m1 = fminf(m1, x1);
            }
        }
        m0 = fminf(m0, __shfl_xor(m0, 16)); m0 = fminf(m0, __shfl_xor(m0, 32));
        m1 = fminf(m1, __shfl_xor(m1, 16)); m1 = fminf(m1, __shfl_xor(m1, 32));
        if (q == 0) {
            if (oka && (!has_feat || has_feat[da])) atomicMin(out + da, __float_as_uint(m0));
            if (okb && (!has_feat || has_feat[db])) atomicMin(out + db, __float_as_uint(m1));
        }
    }
}

struct TrkCommit {   // per-track commit of the previous frame, folded into the next frame's association launch (kind == nullptr: none)
    const int* kind;      // 0 none, 1 Kalman update, 2 initiate
    const int* det;       // detection row (previous frame) for kind 1 / 2
    const int* kout;      // row of out_tlwh for kind 1
    const int* appos;     // gallery ring position to write, or -1
    const int* apdet;     // detection row whose feature is appended
    const float* xyah;    // previous frame's detections
    const float* feat; const float* feat_n;
    float* out_tlwh; float* gal_raw; float* gal_w;
};

// trk_assoc_all_kernel: ONE launch per frame for everything the host association needs about track t (one 1024-thread
// block per track): the lazy Kalman predict + squared Mahalanobis + IoU rows of trk_assoc_kernel (waves 0..1) and the
// appearance row of cosine_min_mfma_kernel -- wave pair w>>1 takes the 16-row gallery slices, each wave half of K, and the per-slice minima
// meet in LDS instead of atomicMin, so the three rows are plain stores and may go STRAIGHT to pinned host memory: no
// init pass, no second launch, no device-to-host blit on the per-frame chain.  Same products; the dot product is summed as
// two K halves (fp32 rounding differs from the one-pass kernel in the last bit).
__global__ __launch_bounds__(1024) void trk_assoc_all_kernel(float* mean, float* cov, const int* __restrict__ slots,
                                                             const int* __restrict__ glen, int do_predict,
                                                             const float* __restrict__ det_tlwh, const float* __restrict__ det_xyah,
                                                             const float* gal_n, int gmax, int dim,
                                                             const float* __restrict__ det_n, const unsigned char* __restrict__ has_feat,
                                                             int n, float* app, float* d2, float* iouc, TrkCommit cm) {
    // 16 waves: wave w = (gallery slice lane w>>1, K half w&1).  The two K halves of a slice meet in LDS (the chain is
    // latency-bound: half the dependent load batches per wave), then the per-slice minima meet in red[].
    extern __shared__ float red[];                     // [8][n] per-slice minima, then [8][64][8] partial sums
    float* part = red + 8 * n;
    const int t = blockIdx.x;
    const int slot = slots[t];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const bool two = blockDim.x == 1024;                // 16 waves: K split in halves; 8 waves: one wave per slice
    const int sl = two ? wv >> 1 : wv, ks = two ? wv & 1 : 0;
    float* P = cov + (size_t)slot * 64;
    float* m = mean + (size_t)slot * 8;
    // ---- commit of the PREVIOUS frame for this track (pipelined form: one launch per frame): Kalman update or initiate on
    // wave 0, the gallery row on the other waves; then the block continues with the next frame's rows
    if (cm.kind != nullptr) {
        const int kind = cm.kind[t];
        if (threadIdx.x < 64) {
            const int i = lane >> 3, j = lane & 7;
            if (kind == 1) {
                const float* zz = cm.xyah + (size_t)cm.det[t] * 4;
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
                __builtin_amdgcn_s_waitcnt(0);      // every lane has read P / m before anyone overwrites them (one wave: lock-step, but loads are async)
                P[i * 8 + j] = pij;
                if (j == 0) m[i] = mi;
                const float cx = __shfl(mi, 0), cy = __shfl(mi, 8), ar = __shfl(mi, 16), hh = __shfl(mi, 24);
                if (lane == 0) {
                    float w = 0.f, h2 = hh;
                    if (hh > 0.f) w = ar * hh; else h2 = fmaxf(0.f, hh);
                    float* o = cm.out_tlwh + (size_t)cm.kout[t] * 4;
                    o[0] = cx - w / 2.0f; o[1] = cy - h2 / 2.0f; o[2] = w; o[3] = h2;
                }
            } else if (kind == 2) {
                const float* zz = cm.xyah + (size_t)cm.det[t] * 4;
                const float h = zz[3];
                float v = 0.f;
                if (i == j) {
                    if (i == 2) v = (float)(1e-2 * 1e-2);
                    else if (i == 6) v = (float)(1e-5 * 1e-5);
                    else v = sq64((i < 4 ? 0.1f : 0.0625f) * h);
                }
                cov[(size_t)slot * 64 + lane] = v;
                if (j == 0) mean[(size_t)slot * 8 + i] = i < 4 ? zz[i] : 0.f;
            }
        } else if (cm.appos[t] >= 0) {
            const size_t dst = ((size_t)slot * gmax + cm.appos[t]) * dim;
            const size_t src = (size_t)cm.apdet[t] * dim;
            for (int c = threadIdx.x - 64; c < dim; c += blockDim.x - 64) {
                cm.gal_raw[dst + c] = cm.feat[src + c];
                cm.gal_w[dst + c] = cm.feat_n[src + c];
            }
        }
        __syncthreads();                            // the block's own global writes are visible to all its waves from here on
        if (n <= 0) return;                         // (block-uniform) commit only: no next frame to prepare
    }
    if (do_predict && threadIdx.x < 64) {
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
        P[i * 8 + j] = t2;
        if (j == 0) m[i] = mi;
    }
    // ---- appearance
    const int len = (gal_n != nullptr && dim > 0) ? glen[t] : 0;
    if (ks == 0)
        for (int j = lane; j < n; j += 64) red[sl * n + j] = 3.0e38f;
    const int kmid = two ? (dim / 2) / 64 * 64 : dim;   // K split on a 64-element boundary (the tail stays in the upper half)
    const int k_lo = ks ? kmid : 0, k_hi = ks ? dim : kmid;
    for (int gb = 0; gb < len; gb += 128) {             // block-uniform trip counts: the barriers below are safe
        const int g0 = gb + sl * 16;
        const bool live = g0 < len;
        const float* grow = gal_n + ((size_t)slot * gmax + min(g0 + r, max(len - 1, 0))) * dim;
        for (int d0 = 0; d0 < n; d0 += 32) {
            const int da = d0 + r, db = d0 + 16 + r;
            floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                const float* pa = det_n + (size_t)min(da, n - 1) * dim;
                const float* pb = det_n + (size_t)min(db, n - 1) * dim;
                int k0 = k_lo;
                for (; k0 + 64 <= k_hi; k0 += 64) {
                    floatx4 a[4], b0[4], b1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 + 16 * u + 4 * q;
                        a[u] = *reinterpret_cast<const floatx4*>(grow + k);
                        b0[u] = *reinterpret_cast<const floatx4*>(pa + k);
                        b1[u] = *reinterpret_cast<const floatx4*>(pb + k);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b0[u][e], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b1[u][e], acc1, 0, 0, 0);
                        }
                }
                for (; k0 < k_hi; k0 += 16) {
                    const int k = k0 + 4 * q;
                    floatx4 a = {0.f, 0.f, 0.f, 0.f}, b0 = a, b1 = a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (k + e < k_hi) { a[e] = grow[k + e]; b0[e] = pa[k + e]; b1[e] = pb[k + e]; }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc1, 0, 0, 0);
                    }
                }
            }
            float* ps = part + ((size_t)sl * 64 + lane) * 8;
            if (two) {
                if (ks == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ps[e] = acc0[e]; ps[4 + e] = acc1[e]; }
                }
                __syncthreads();
                if (ks == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc0[e] += ps[e]; acc1[e] += ps[4 + e]; }
                }
            }
            if (ks == 0 && live) {
                float m0 = 3.0e38f, m1 = 3.0e38f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (g0 + 4 * q + e < len) {
                        float x0 = 1.0f - acc0[e], x1 = 1.0f - acc1[e];
                        x0 = x0 > 0.f ? x0 : 0.f;
                        x1 = x1 > 0.f ? x1 : 0.f;
                        m0 = fminf(m0, x0);
                        m1 = fminf(m1, x1);
                    }
                }
                m0 = fminf(m0, __shfl_xor(m0, 16)); m0 = fminf(m0, __shfl_xor(m0, 32));
                m1 = fminf(m1, __shfl_xor(m1, 16)); m1 = fminf(m1, __shfl_xor(m1, 32));
                if (q == 0) {                       // this wave owns red[sl][*]
                    if (da < n) red[sl * n + da] = fminf(red[sl * n + da], m0);
                    if (db < n) red[sl * n + db] = fminf(red[sl * n + db], m1);
                }
            }
            if (two) __syncthreads();               // part[] is reused by the next tile
        }
    }
    __syncthreads();   // predict's global writes and every wave's minima are visible to the whole block
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        float v = 1e5f;                              // INFTY_COST (matching.py:148,175): empty gallery / featureless detection
        if (len > 0 && (!has_feat || has_feat[j])) {
            float mn = red[j];
#pragma unroll
            for (int w = 1; w < 8; ++w) mn = fminf(mn, red[w * n + j]);
            v = mn;
        }
        app[(size_t)t * n + j] = v;
    }
    // ---- gating distance + IoU rows (trk_assoc_kernel, threads 0..127)
    if (threadIdx.x < 128) {
        float S[4][4], L[4][4];
        innovation_cov(P, m[3], S);
        const bool ok = cholesky<4>(S, L);
        float bw = 0.f, bh = m[3];
        if (bh > 0.f) bw = m[2] * bh; else bh = fmaxf(0.f, bh);
        const float bx = m[0] - bw / 2.0f, by = m[1] - bh / 2.0f;
        const float brx = bx + bw, bry = by + bh;
        for (int j = threadIdx.x; j < n; j += 128) {
            const float* z = det_xyah + (size_t)j * 4;
            float d[4], y[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) d[a] = z[a] - m[a];
            fwd_solve<4>(L, d, y);
            float acc = y[0] * y[0];
            acc = acc + y[1] * y[1];
            acc = acc + y[2] * y[2];
            acc = acc + y[3] * y[3];
            const size_t o = (size_t)t * n + j;
            d2[o] = ok ? acc : __builtin_inff();
            const float* c = det_tlwh + (size_t)j * 4;
            const float crx = c[0] + c[2], cry = c[1] + c[3];
            const float iw = fmaxf(0.f, fminf(brx, crx) - fmaxf(bx, c[0]));
            const float ih = fmaxf(0.f, fminf(bry, cry) - fmaxf(by, c[1]));
            const float inter = iw * ih;
            const float uni = bw * bh + c[2] * c[3] - inter;
            iouc[o] = 1.0f - inter / fmaxf(uni, 1e-7f);
        }
    }
}

// trk_commit_kernel: one wavefront-sized block per work item, three roles by block index:
//   [0, M)        Kalman update of a matched track (kalman_filter.py:153-204) + its new tlwh
//   [M, M+U)      initiate a new track (kalman_filter.py:55-83)
//   [M+U, M+U+A)  append a feature to a gallery ring (track.py:70-74): raw row + normalised row
__global__ __launch_bounds__(64) void trk_commit_kernel(float* mean, float* cov, const int* __restrict__ lists, int M, int U, int A,
                                                        const float* __restrict__ xyah, float* out_tlwh, float* gal_raw, float* gal_n,
                                                        int gmax, int dim, const float* __restrict__ feat, const float* __restrict__ feat_n) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int* upd_slot = lists;
    const int* upd_det = lists + M;
    const int* ini_slot = lists + 2 * M;
    const int* ini_det = lists + 2 * M + U;
    const int* ap_slot = lists + 2 * M + 2 * U;
    const int* ap_pos = ap_slot + A;
    const int* ap_det = ap_pos + A;
    if (b < M) {
        const int i = lane >> 3, j = lane & 7;
        float* P = cov + (size_t)upd_slot[b] * 64;
        float* m = mean + (size_t)upd_slot[b] * 8;
        const float* zz = xyah + (size_t)upd_det[b] * 4;
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
        P[i * 8 + j] = pij;
        if (j == 0) m[i] = mi;
        const float cx = __shfl(mi, 0), cy = __shfl(mi, 8), ar = __shfl(mi, 16), hh = __shfl(mi, 24);
        if (lane == 0) {
            float w = 0.f, h2 = hh;
            if (hh > 0.f) w = ar * hh; else h2 = fmaxf(0.f, hh);
            float* o = out_tlwh + (size_t)b * 4;
            o[0] = cx - w / 2.0f; o[1] = cy - h2 / 2.0f; o[2] = w; o[3] = h2;
        }
    } else if (b < M + U) {
        const int k = b - M, i = lane >> 3, j = lane & 7;
        const int slot = ini_slot[k];
        const float* zz = xyah + (size_t)ini_det[k] * 4;
        const float h = zz[3];
        float v = 0.f;
        if (i == j) {
            if (i == 2) v = (float)(1e-2 * 1e-2);
            else if (i == 6) v = (float)(1e-5 * 1e-5);
            else v = sq64((i < 4 ? 0.1f : 0.0625f) * h);
        }
        cov[(size_t)slot * 64 + lane] = v;
        if (j == 0) mean[(size_t)slot * 8 + i] = i < 4 ? zz[i] : 0.f;
    } else {
        const int k = b - M - U;
        const size_t dst = ((size_t)ap_slot[k] * gmax + ap_pos[k]) * dim;
        const size_t src = (size_t)ap_det[k] * dim;
        for (int c = lane; c < dim; c += 64) {
            gal_raw[dst + c] = feat[src + c];
            gal_n[dst + c] = feat_n[src + c];
        }
    }
}

// gallery[slot][pos] <- feat[det]   (track.py:70-74; FIFO realised as a ring, the host keeps heads)
__global__ void gallery_append_kernel(float* gal, int gmax, int dim, const int* __restrict__ slot, const int* __restrict__ pos,
                                      const int* __restrict__ det, const float* __restrict__ feat, int count) {
    const int k = blockIdx.x;
    if (k >= count) return;
    float* dst = gal + ((size_t)slot[k] * gmax + pos[k]) * dim;
    const float* src = feat + (size_t)det[k] * dim;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) dst[c] = src[c];
}

// ------------------------------------------------------------------------------------------------
void launch_kf_initiate(const float* z, int n, float* mean, float* cov, const int* slots, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(kf_initiate_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, z, n, mean, cov, slots, (const int*)nullptr);
    KCHECK();
}
void launch_kf_initiate_idx(const float* z, const int* zidx, int n, float* mean, float* cov, const int* slots, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(kf_initiate_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, z, n, mean, cov, slots, zidx);
    KCHECK();
}
void launch_kf_predict(float* mean, float* cov, const int* slots, int n, hipStream_t s, float dt) {
    if (n <= 0) return;
    hipLaunchKernelGGL(kf_predict_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, mean, cov, slots, n, dt);
    KCHECK();
}
void launch_kf_project(const float* mean, const float* cov, int n, float* pmean, float* pcov, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(kf_project_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, s, mean, cov, n, pmean, pcov);
    KCHECK();
}
void launch_kf_update(float* mean, float* cov, const int* slots, const float* z, const int* zidx, int n, float* out_tlwh, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(kf_update_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, mean, cov, slots, z, zidx, n, out_tlwh);
    KCHECK();
}
void launch_kf_gating(const float* mean, const float* cov, const int* slots, int n, const float* zs, int m, int shared_z,
                      int only_position, float* d2, hipStream_t s) {
    if (n <= 0 || m <= 0) return;
    hipLaunchKernelGGL(kf_gating_kernel, dim3(n), dim3(64), 0, s, mean, cov, slots, n, zs, m, shared_z, only_position, d2);
    KCHECK();
}
void launch_iou_cost(const float* trk_tlwh, const float* mean, const int* slots, int t, const float* det_tlwh, int n, float* cost, hipStream_t s) {
    if (t <= 0 || n <= 0) return;
    hipLaunchKernelGGL(iou_cost_kernel, dim3(ceil_div((long)t * n, 256)), dim3(256), 0, s, trk_tlwh, mean, slots, t, det_tlwh, n, cost);
    KCHECK();
}
void launch_fill(float* p, float v, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(fill_kernel, dim3(ceil_div((long)n, 256)), dim3(256), 0, s, p, v, n);
    KCHECK();
}
void launch_normalize_rows(const float* src, float* dst, int n, int dim, hipStream_t s, const int* n_dev) {
    if (n <= 0) return;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, src, dst, n, dim, n_dev);
    KCHECK();
}
void launch_cosine_min(const float* gal, const int* slots, const int* glen, int t, int gmax, int dim, const float* det_n,
                       const unsigned char* has_feat, int n, float* cost, hipStream_t s) {
    if (t <= 0 || n <= 0 || gmax <= 0) return;
    dim3 grid(t, ceil_div(gmax, 16));
    AIC_REQUIRE(dim <= 1024, AIC_ERR_CAPACITY, "feature dimension above 1024 is not supported");
#define COS_LAUNCH(NP) hipLaunchKernelGGL(cosine_min_kernel<NP>, grid, dim3(256), 0, s, gal, slots, glen, gmax, dim, det_n, has_feat, n, cost)
    if (dim <= 64) COS_LAUNCH(1);
    else if (dim <= 128) COS_LAUNCH(2);
    else if (dim <= 256) COS_LAUNCH(4);
    else if (dim <= 512) COS_LAUNCH(8);
    else COS_LAUNCH(16);
#undef COS_LAUNCH
    KCHECK();
}
void launch_gallery_append(float* gal, int gmax, int dim, const int* slot, const int* pos, const int* det, const float* feat,
                           int count, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(gallery_append_kernel, dim3(count), dim3(128), 0, s, gal, gmax, dim, slot, pos, det, feat, count);
    KCHECK();
}

void launch_trk_assoc(float* mean, float* cov, const int* slots, int t, int do_predict, const float* det_tlwh,
                      const float* det_xyah, int n, float* app, float* d2, float* iouc, hipStream_t s) {
    if (t <= 0) return;
    hipLaunchKernelGGL(trk_assoc_kernel, dim3(t), dim3(128), 0, s, mean, cov, slots, do_predict, det_tlwh, det_xyah, n, app, d2, iouc);
    KCHECK();
}
void launch_cosine_min_mfma(const float* gal_n, const int* slots, const int* glen, int t, int gmax, int dim, const float* det_n,
                            const unsigned char* has_feat, int n, float* cost, hipStream_t s) {
    if (t <= 0 || n <= 0 || gmax <= 0) return;
    hipLaunchKernelGGL(cosine_min_mfma_kernel, dim3(t, ceil_div(gmax, 64)), dim3(256), 0, s, gal_n, slots, glen, gmax, dim, det_n, has_feat, n, cost);
    KCHECK();
}
void launch_trk_assoc_all(float* mean, float* cov, const int* slots, const int* glen, int t, int do_predict, const float* det_tlwh,
                          const float* det_xyah, const float* gal_n, int gmax, int dim, const float* det_n,
                          const unsigned char* has_feat, int n, float* app, float* d2, float* iouc, hipStream_t s) {
    launch_trk_step(mean, cov, slots, glen, t, do_predict, det_tlwh, det_xyah, gal_n, gmax, dim, det_n, has_feat, n, app, d2, iouc,
                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s);
}
void launch_trk_step(float* mean, float* cov, const int* slots, const int* glen, int t, int do_predict, const float* det_tlwh,
                     const float* det_xyah, const float* gal_n, int gmax, int dim, const float* det_n,
                     const unsigned char* has_feat, int n, float* app, float* d2, float* iouc,
                     const int* c_kind, const int* c_det, const int* c_kout, const int* c_appos, const int* c_apdet,
                     const float* c_xyah, const float* c_feat, const float* c_feat_n, float* c_out_tlwh, float* c_gal_raw, float* c_gal_w,
                     hipStream_t s) {
    if (t <= 0 || (n <= 0 && c_kind == nullptr)) return;
    const TrkCommit cm{c_kind, c_det, c_kout, c_appos, c_apdet, c_xyah, c_feat, c_feat_n, c_out_tlwh, c_gal_raw, c_gal_w};
    static const int ks = [] { const char* e = getenv("AICAM_TRK_KS"); return e ? atoi(e) : 1; }();   // 2: K split over wave pairs (1024-thread blocks wait longer for a CU: 88 vs 83 us chain)
    hipLaunchKernelGGL(trk_assoc_all_kernel, dim3(t), dim3(ks == 2 ? 1024 : 512), ((size_t)8 * std::max(n, 0) + 8 * 64 * 8) * sizeof(float), s, mean, cov, slots, glen, do_predict,
                       det_tlwh, det_xyah, gal_n, gmax, dim, det_n, has_feat, n, app, d2, iouc, cm);
    KCHECK();
}
void launch_trk_commit(float* mean, float* cov, const int* lists, int M, int U, int A, const float* xyah, float* out_tlwh,
                       float* gal_raw, float* gal_n, int gmax, int dim, const float* feat, const float* feat_n, hipStream_t s) {
    if (M + U + A <= 0) return;
    hipLaunchKernelGGL(trk_commit_kernel, dim3(M + U + A), dim3(64), 0, s, mean, cov, lists, M, U, A, xyah, out_tlwh, gal_raw, gal_n, gmax, dim, feat, feat_n);
    KCHECK();
}

}  // namespace aic
