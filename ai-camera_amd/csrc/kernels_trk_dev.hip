// kernels_trk_dev.hip -- DeepSORT association on the device, k frames per launch.
//
// Reference, per frame (Python, on the host):
//   TrackerCore.predict / update / _match / _initiate_track     src/tracker/core/tracker_core.py:44-81, 83-177, 180-194
//   matching_cascade, min_cost_matching, Mahalanobis gate       src/tracker/core/linear_assignment.py:19-88, 91-157, 160-212
//   scipy.optimize.linear_sum_assignment (SciPy 1.15.3)          called at linear_assignment.py:62
//   Track.update / mark_missed / _add_feature                   src/tracker/core/track.py:70-74, 82-119
//   output formatting                                           src/tracker/deepsort_tracker.py:126-141
// The arithmetic (Kalman, gating, IoU, cosine) is that of kernels_trk.hip (trk_math.hpp; same MFMA contraction order for
// the cosine products, so the cost values are bit-identical to the one-launch-per-frame path); the integer logic restates
// csrc/assoc_host.cpp + csrc/lsap.cpp + Tracker::update of tracker.cpp for one workgroup.  Structure: trk_dev.hpp.
#include "kernels.hpp"
#include "trk_dev.hpp"
#include "trk_math.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace aic {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BT = TRK_DEV_TMAX;        // threads of the epoch kernel: thread t <-> track t, thread j <-> detection j
constexpr int NW = BT / 64;
constexpr float kInfty = 1e5f;                               // linear_assignment.py:9
constexpr float kChi2_4 = (float)9.487729036781154;          // kalman_filter.py:16, compared in fp32
constexpr float kBig = 3.0e38f;

// ------------------------------------------------------------------------------------------------ prep kernel
// One wave = one 16-row x 32-detection tile of cosine distances, K walked exactly like cosine_min_mfma_kernel /
// trk_assoc_all_kernel (lane (r, q): 16 bytes of its row per 16-deep slice, MFMA e consumes element e of both operands).
__device__ __forceinline__ void cos_tile(const float* __restrict__ grow, const float* __restrict__ pa, const float* __restrict__ pb,
                                         int dim, int q, floatx4& acc0, floatx4& acc1) {
    acc0 = floatx4{0.f, 0.f, 0.f, 0.f};
    acc1 = acc0;
    int k0 = 0;
    for (; k0 + 64 <= dim; k0 += 64) {
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
}

__device__ __forceinline__ float cos_dist(float dot) {       // matching.py:136-141
    const float x = 1.0f - dot;
    return x > 0.f ? x : 0.f;
}

__device__ __forceinline__ void wave_lds_sync() {            // LDS traffic of ONE wave: program order + a compiler fence
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace

// grid: any (wave-task loop); block 256 = 4 waves, each with its own 16 x 36 LDS tile (row pitch 36 floats: the four row groups
// a lane quartet writes land in disjoint bank ranges).
// SM task = (track, 32 detections): K is the OUTER loop and up to SMG = 7 gallery row groups (112 rows) are accumulated side by
// side, so the detection features cross L2 -> registers once per task instead of once per row group (36 loads per 224 MFMAs
// instead of 84); the accumulation order of every single dot product is unchanged (k ascending), so the values are
// bit-identical to cos_tile's.  The suffix minima are then taken group by group, newest rows first.
constexpr int SMG = 7;

__global__ __launch_bounds__(256) void trk_epoch_prep_kernel(const DevTrkHdr* __restrict__ hdr, const DevTrack* __restrict__ trk,
                                                             const float* __restrict__ gal_n, int gmax, int dim,
                                                             const float* __restrict__ featn, int dn, int dn_pad, int k,
                                                             float* __restrict__ sm, float* __restrict__ gram) {
    __shared__ float tiles[4][16][36];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    float (*tile)[36] = tiles[wv];
    const int T0 = hdr->n_tracks;
    const int nchunk = dn_pad / 32;
    const int n_sm = T0 * nchunk, n_all = n_sm + (dn_pad / 16) * nchunk;
    for (int task = blockIdx.x * 4 + wv; task < n_all; task += gridDim.x * 4) {
        const bool is_sm = task < n_sm;
        const int tk = is_sm ? task : task - n_sm;
        const int rowi = tk / nchunk, c = tk - rowi * nchunk;     // track (SM) or 16-row detection group (GRAM)
        const int d_base = c * 32;
        const float* pa = featn + (size_t)min(d_base + r, dn - 1) * dim;
        const float* pb = featn + (size_t)min(d_base + 16 + r, dn - 1) * dim;
        if (!is_sm) {
            const int a0 = rowi * 16;
            if (a0 >= d_base + 32) continue;                      // rows all later than the columns: never read (an appended row only meets LATER frames)
            floatx4 acc0, acc1;
            cos_tile(featn + (size_t)min(a0 + r, dn - 1) * dim, pa, pb, dim, q, acc0, acc1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* o = gram + (size_t)(a0 + 4 * q + e) * dn_pad + d_base;
                o[r] = cos_dist(acc0[e]);
                o[16 + r] = cos_dist(acc1[e]);
            }
            continue;
        }
        const DevTrack tr = trk[rowi];
        const int glen0 = tr.glen;
        float* smt = sm + (size_t)rowi * (TRK_KMAX + 1) * dn_pad + d_base;
        if (lane < 32)
            for (int e = glen0; e <= k; ++e) smt[(size_t)e * dn_pad + lane] = kBig;       // empty suffix
        float R = kBig;
        const int G = (glen0 + 15) / 16;
        for (int gb = ((G - 1) / SMG) * SMG; gb >= 0; gb -= SMG) {   // batches of SMG row groups, newest batch first
            const int ng = min(SMG, G - gb);
            const float* grow[SMG];
#pragma unroll
            for (int g = 0; g < SMG; ++g) {
                const int j = min(16 * (gb + min(g, ng - 1)) + r, glen0 - 1);      // groups past the end alias the last one (never read back)
                int pos = tr.ghead + j;
                if (pos >= gmax) pos -= gmax;
                grow[g] = gal_n + ((size_t)tr.slot * gmax + pos) * dim;
            }
            floatx4 acc[SMG][2];
#pragma unroll
            for (int g = 0; g < SMG; ++g) acc[g][0] = acc[g][1] = floatx4{0.f, 0.f, 0.f, 0.f};
            int k0 = 0;
            for (; k0 + 64 <= dim; k0 += 64) {
                floatx4 b0[4], b1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + 16 * u + 4 * q;
                    b0[u] = *reinterpret_cast<const floatx4*>(pa + kk);
                    b1[u] = *reinterpret_cast<const floatx4*>(pb + kk);
                }
#pragma unroll
                for (int g = 0; g < SMG; ++g) {
                    if (g < ng) {
                        floatx4 a[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const floatx4*>(grow[g] + k0 + 16 * u + 4 * q);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b0[u][e], acc[g][0], 0, 0, 0);
                                acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b1[u][e], acc[g][1], 0, 0, 0);
                            }
                    }
                }
            }
            for (; k0 < dim; k0 += 16) {                          // tail: any dimension, element-guarded
                const int kk = k0 + 4 * q;
                floatx4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kk + e < dim) { b0[e] = pa[kk + e]; b1[e] = pb[kk + e]; }
#pragma unroll
                for (int g = 0; g < SMG; ++g) {
                    if (g < ng) {
                        floatx4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (kk + e < dim) a[e] = grow[g][kk + e];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc[g][0], 0, 0, 0);
                            acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc[g][1], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int g = SMG - 1; g >= 0; --g) {                  // suffix minima, newest rows first
                if (g < ng) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        tile[4 * q + e][r] = cos_dist(acc[g][0][e]);
                        tile[4 * q + e][16 + r] = cos_dist(acc[g][1][e]);
                    }
                    wave_lds_sync();
                    if (lane < 32) {
                        for (int jj = 15; jj >= 0; --jj) {
                            const int row = 16 * (gb + g) + jj;
                            if (row < glen0) {
                                R = fminf(R, tile[jj][lane]);
                                if (row <= k) smt[(size_t)row * dn_pad + lane] = R;     // min over FIFO rows >= row
                            }
                        }
                    }
                    wave_lds_sync();
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ epoch kernel
namespace {

struct Lds {                        // carved out of the dynamic LDS block by lds_carve()
    // track table (SoA), list order
    int *id, *state, *hits, *age, *tsu, *cls, *slot, *glen, *ghead, *sm, *napp, *glen0;
    float* conf;
    unsigned short* newrow;         // [TMAX][ks] epoch rows appended to the track in this epoch
    int ks;                         // ... ks = frames of THIS epoch (<= TRK_KMAX): 16 KB instead of 32 at the default epoch length
    int *mdet;                      // [TMAX] matched detection of the frame or -1
    int *feas;                      // [TMAX] 1: the track has a detection inside both gates in this frame (its row of the cascade's sub-matrix is not all clamp)
    float* tbox;                    // [TMAX][4] tlwh of the updated state (outputs)
    int* free_slots;                // [cap]
    // detections of the frame
    float *tlwh, *xyah, *dconf;
    int *dcls, *dhas, *mtrk, *und, *cols;
    int* rows;                      // [TMAX] row list of the current assignment problem
    // LSAP
    double *u, *v, *dist;
    int *pred, *rowof, *colof, *todo, *pos, *asg;
    // scalars / scratch
    int* wcnt;                      // [NW + 8]
    float* kf;                      // [KF_SLOTS][72] Kalman state of the low slots for the whole epoch: cov[64] | mean[8]
    float* arena;                   // cost matrices of the frame (when they fit) + assignment sub-matrix
    int arena_floats;
};

constexpr int KF_SLOTS = 96;        // slots are handed out lowest first, so live tracks sit here unless > 96 are alive

__device__ __forceinline__ Lds lds_carve(char* base, int cap, int nmax, int total_bytes, int ks = TRK_KMAX) {
    Lds L;
    char* p = base;
    auto take = [&](size_t bytes) { char* q = p; p += (bytes + 15) & ~(size_t)15; return q; };
    const size_t tc = (size_t)cap, mx = (size_t)max(cap, nmax);     // table rows; side of the largest assignment problem
    L.u = (double*)take(8 * mx); L.v = (double*)take(8 * mx); L.dist = (double*)take(8 * mx);
    int** ti[] = {&L.id, &L.state, &L.hits, &L.age, &L.tsu, &L.cls, &L.slot, &L.glen, &L.ghead, &L.sm, &L.napp, &L.glen0, &L.mdet, &L.rows, &L.feas};
    for (auto a : ti) *a = (int*)take(4 * tc);
    int** li[] = {&L.pred, &L.rowof, &L.colof, &L.todo, &L.pos, &L.asg};
    for (auto a : li) *a = (int*)take(4 * mx);
    L.conf = (float*)take(4 * tc);
    L.tbox = (float*)take(16 * tc);
    L.ks = ks;
    L.newrow = (unsigned short*)take(2 * tc * (size_t)ks);
    L.free_slots = (int*)take(4 * tc);
    L.kf = (float*)take(4 * 72 * (size_t)KF_SLOTS);
    L.tlwh = (float*)take(16 * (size_t)nmax); L.xyah = (float*)take(16 * (size_t)nmax); L.dconf = (float*)take(4 * (size_t)nmax);
    int** di[] = {&L.dcls, &L.dhas, &L.mtrk, &L.und, &L.cols};
    for (auto a : di) *a = (int*)take(4 * (size_t)nmax);
    L.wcnt = (int*)take(4 * (NW + 8));
    L.arena = (float*)p;
    L.arena_floats = (int)((total_bytes - (p - base)) / 4);
    return L;
}

// ordered (stable) compaction over the block: flagged threads write `value` at list[rank]; returns the count. Two barriers.
__device__ __forceinline__ int block_compact(bool flag, int value, int* list, int* wcnt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[w] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int cnt = wcnt[i];
        if (i < w) off += cnt;
        tot += cnt;
    }
    if (flag) list[off + before] = value;
    __syncthreads();
    return tot;
}

__device__ __forceinline__ int block_min_int(int v, int* wcnt) {     // two barriers
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    if (lane == 0) wcnt[w] = v;
    __syncthreads();
    int m = wcnt[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = min(m, wcnt[i]);
    __syncthreads();
    return m;
}

// ---- wave reductions on DPP row shifts (full wave active) -------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ unsigned dpp_keep(unsigned old, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned wave_umin32(unsigned v) {
    v = min(v, dpp_keep<0x111>(0xffffffffu, v));   // row_shr:1  (lanes without a source keep the identity)
    v = min(v, dpp_keep<0x112>(0xffffffffu, v));   // row_shr:2
    v = min(v, dpp_keep<0x114>(0xffffffffu, v));   // row_shr:4
    v = min(v, dpp_keep<0x118>(0xffffffffu, v));   // row_shr:8  -> lane 15 of every row of 16 holds the row minimum
    const unsigned a = __builtin_amdgcn_readlane((int)v, 15), b = __builtin_amdgcn_readlane((int)v, 31);
    const unsigned c = __builtin_amdgcn_readlane((int)v, 47), d = __builtin_amdgcn_readlane((int)v, 63);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ unsigned wave_umax32(unsigned v) { return ~wave_umin32(~v); }
// order-preserving 64-bit key of a double (no NaN here): smaller double <=> smaller unsigned key
__device__ __forceinline__ unsigned long long f64_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ double wave_min_f64(double x) {
    const unsigned long long k = f64_key(x);
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned mh = wave_umin32(hi);
    const unsigned long long top = __ballot(hi == mh);
    if (__popcll(top) == 1) {                                   // one lane holds the smallest high word: it IS the minimum
        const int l = __ffsll((long long)top) - 1;
        const unsigned ml1 = (unsigned)__builtin_amdgcn_readlane((int)lo, l);
        return key_f64(((unsigned long long)mh << 32) | ml1);
    }
    const unsigned ml = wave_umin32(hi == mh ? lo : 0xffffffffu);
    return key_f64(((unsigned long long)mh << 32) | ml);
}

// Rectangular linear sum assignment of SciPy 1.15.3 (Crouse's shortest augmenting path), restating csrc/lsap.cpp for ONE
// wavefront: the scan over the unscanned columns is spread over the lanes, the three details that decide WHICH optimum comes
// back are kept exactly --
//   (1) unscanned columns live in a list initialised in DESCENDING column order, a scanned column is replaced by the list's
//       last entry (pos[] is the inverse of todo[]);
//   (2) among equal reduced path costs the LAST visited unassigned column wins, otherwise the FIRST visited column
//       (= max list position over the unassigned minima if there is one, else min list position over the minima);
//   (3) dual update and back-tracking along pred[] in the reference's order and fp64 operation order.
// cm: nr x nc fp32 (LDS or global), solved transposed when nr > nc (lsap.cpp:113-123).  Result asg[orig row] = orig column or -1.
// Returns false when no finite completion exists (cannot happen for the clamped matrices of min_cost_matching).
__device__ __noinline__ bool lsap_wave(const float* cm, int nr, int nc, const Lds& L, int lane) {
    const bool tall = nr > nc;
    const int R = tall ? nc : nr, C = tall ? nr : nc;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int j = lane; j < C; j += 64) { L.v[j] = 0.0; L.rowof[j] = -1; }
    for (int i = lane; i < R; i += 64) { L.u[i] = 0.0; L.colof[i] = -1; }
    wave_lds_sync();
    for (int root = 0; root < R; ++root) {
        for (int j = lane; j < C; j += 64) { L.dist[j] = inf; L.todo[j] = C - 1 - j; L.pos[j] = C - 1 - j; }
        wave_lds_sync();
        double base = 0.0;
        int live = C, i = root, sink = -1;
        while (sink < 0) {
            const double ui = L.u[i];
            double lmin = inf;
            for (int j = lane; j < C; j += 64) {
                if (L.pos[j] >= 0) {
                    const float cij = tall ? cm[(size_t)j * nc + i] : cm[(size_t)i * nc + j];
                    const double red = ((base + (double)cij) - ui) - L.v[j];
                    double dj = L.dist[j];
                    if (red < dj) { dj = red; L.dist[j] = red; L.pred[j] = i; }
                    lmin = dj < lmin ? dj : lmin;
                }
            }
            const double m = wave_min_f64(lmin);
            if (!(m < inf)) return false;
            unsigned pa = 0u, pb = 0xffffffffu;               // pa: 1 + max position of an unassigned minimum; pb: min position of a minimum
            for (int j = lane; j < C; j += 64) {
                const int pj = L.pos[j];
                if (pj >= 0 && L.dist[j] == m) {
                    pb = min(pb, (unsigned)pj);
                    if (L.rowof[j] < 0) pa = max(pa, (unsigned)pj + 1u);
                }
            }
            pa = wave_umax32(pa);
            const int pick = pa ? (int)pa - 1 : (int)wave_umin32(pb);
            base = m;
            const int j = L.todo[pick];
            const int rj = L.rowof[j];
            if (rj < 0) sink = j; else i = rj;
            if (lane == 0) {
                const int last = L.todo[live - 1];
                L.todo[pick] = last;
                L.pos[last] = pick;
                L.pos[j] = -1;                                  // scanned
            }
            --live;
            wave_lds_sync();
        }
        // dual update (lsap.cpp:83-87), column side: every scanned assigned column's partner row is a seen row
        for (int j = lane; j < C; j += 64) {
            if (L.pos[j] < 0) {
                const double dlt = base - L.dist[j];
                const int i2 = L.rowof[j];
                if (i2 >= 0) L.u[i2] = L.u[i2] + dlt;
                L.v[j] = L.v[j] - dlt;
            }
        }
        wave_lds_sync();
        if (lane == 0) {
            L.u[root] = L.u[root] + base;
            int j = sink;
            for (;;) {                                          // flip the path back to the root
                const int i2 = L.pred[j];
                L.rowof[j] = i2;
                const int t = L.colof[i2];
                L.colof[i2] = j;
                j = t;
                if (i2 == root) break;
            }
        }
        wave_lds_sync();
    }
    if (!tall) {
        for (int r2 = lane; r2 < nr; r2 += 64) L.asg[r2] = L.colof[r2];
    } else {
        for (int r2 = lane; r2 < nr; r2 += 64) L.asg[r2] = L.rowof[r2];   // solver columns = original rows
    }
    wave_lds_sync();
    return true;
}

__device__ __forceinline__ double readlane_f64(double x, int l) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// The same algorithm with max(nr, nc) <= 64: lane j IS column j (dist, v, pred, row_of_col, list position in registers), lane i
// IS row i (u, col_of_row); the list of unscanned columns is only its inverse pos[] (the lane whose pos == p sits at list
// position p).  LDS is touched for the cost entries alone.
__device__ __noinline__ bool lsap_wave64(const float* cm, int nr, int nc, const Lds& L, int lane, int ld = 0) {
    if (ld == 0) ld = nc;                                         // row pitch of cm (the wave cascade pads it to an odd number of words)
    const bool tall = nr > nc;
    const int R = tall ? nc : nr, C = tall ? nr : nc;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double v = 0.0, u = 0.0, dist = inf;
    int rowof = -1, colof = -1, pred = -1;
    for (int root = 0; root < R; ++root) {
        dist = inf;
        int pos = lane < C ? C - 1 - lane : -1;
        bool seen = false;
        double base = 0.0;
        int live = C, i = root, sink = -1;
        while (sink < 0) {
            const double ui = readlane_f64(u, i);
            if (pos >= 0) {
                const float cij = tall ? cm[lane * ld + i] : cm[i * ld + lane];
                const double red = ((base + (double)cij) - ui) - v;
                if (red < dist) { dist = red; pred = i; }
            }
            const double m = wave_min_f64(pos >= 0 ? dist : inf);
            if (!(m < inf)) return false;
            const bool cand = pos >= 0 && dist == m;
            const bool cand_u = cand && rowof < 0;
            const unsigned long long bu = __ballot(cand_u);
            int jp;
            if (bu) {                                              // the LAST visited unassigned minimum
                if (__popcll(bu) == 1) jp = __ffsll((long long)bu) - 1;
                else {
                    const unsigned pp = wave_umax32(cand_u ? (unsigned)pos + 1u : 0u) - 1u;
                    jp = __ffsll((long long)__ballot(cand_u && (unsigned)pos == pp)) - 1;
                }
            } else {                                               // the FIRST visited minimum
                const unsigned long long bc = __ballot(cand);
                if (__popcll(bc) == 1) jp = __ffsll((long long)bc) - 1;
                else {
                    const unsigned pp = wave_umin32(cand ? (unsigned)pos : 0xffffffffu);
                    jp = __ffsll((long long)__ballot(cand && (unsigned)pos == pp)) - 1;
                }
            }
            jp = __builtin_amdgcn_readfirstlane(jp);
            base = m;
            const int pick = __builtin_amdgcn_readlane(pos, jp);
            const int rj = __builtin_amdgcn_readlane(rowof, jp);
            if (rj < 0) sink = jp; else i = rj;
            const int jl = __ffsll((long long)__ballot(pos == live - 1)) - 1;   // the list's last entry moves into the freed place
            if (lane == jl) pos = pick;
            if (lane == jp) { pos = -1; seen = true; }
            --live;
        }
        // dual update (lsap.cpp:83-87): a seen row is the partner of a scanned assigned column
        const double dlt = base - dist;
        {
            const int src = colof >= 0 ? colof : 0;
            const double dl = __shfl(dlt, src);
            const int sn = __shfl((int)seen, src);
            if (lane < R && colof >= 0 && sn) u = u + dl;
        }
        if (seen) v = v - dlt;
        if (lane == root) u = u + base;
        int j = sink;
        for (;;) {                                                 // flip the path back to the root
            const int i2 = __builtin_amdgcn_readlane(pred, j);
            if (lane == j) rowof = i2;
            const int t = __builtin_amdgcn_readlane(colof, i2);
            if (lane == i2) colof = j;
            j = t;
            if (i2 == root) break;
        }
    }
    if (lane < nr) L.asg[lane] = tall ? rowof : colof;
    wave_lds_sync();
    return true;
}

// Register-resident form for up to 64 * CPL columns: lane l holds columns l, l + 64, ... (and rows likewise).  Same algorithm and
// tie rules as lsap_wave64 (CPL = 1 compiles to it); every register array is indexed by unrolled constants only.
template <int CPL>
__device__ __noinline__ bool lsap_wave_reg(const float* cm, int nr, int nc, const Lds& L, int lane) {
    const bool tall = nr > nc;
    const int R = tall ? nc : nr, C = tall ? nr : nc;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double v[CPL], u[CPL], dist[CPL];
    int rowof[CPL], colof[CPL], pred[CPL], pos[CPL];
    bool seen[CPL];
#pragma unroll
    for (int s = 0; s < CPL; ++s) { v[s] = 0.0; u[s] = 0.0; dist[s] = inf; rowof[s] = -1; colof[s] = -1; pred[s] = -1; }
    auto pick_i = [&](const int (&a)[CPL], int idx) {            // a[idx >> 6] on lane idx & 63, idx uniform
        int x = a[0];
#pragma unroll
        for (int s = 1; s < CPL; ++s) if ((idx >> 6) == s) x = a[s];
        return __builtin_amdgcn_readlane(x, idx & 63);
    };
    for (int root = 0; root < R; ++root) {
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            const int j = lane + 64 * s;
            dist[s] = inf;
            pos[s] = j < C ? C - 1 - j : -1;
            seen[s] = false;
        }
        double base = 0.0;
        int live = C, i = root, sink = -1;
        while (sink < 0) {
            double us = u[0];
#pragma unroll
            for (int s = 1; s < CPL; ++s) if ((i >> 6) == s) us = u[s];
            const double ui = readlane_f64(us, i & 63);
            double lmin = inf;
#pragma unroll
            for (int s = 0; s < CPL; ++s) {
                if (pos[s] >= 0) {
                    const int j = lane + 64 * s;
                    const float cij = tall ? cm[(size_t)j * nc + i] : cm[(size_t)i * nc + j];
                    const double red = ((base + (double)cij) - ui) - v[s];
                    if (red < dist[s]) { dist[s] = red; pred[s] = i; }
                    lmin = dist[s] < lmin ? dist[s] : lmin;
                }
            }
            const double m = wave_min_f64(lmin);
            if (!(m < inf)) return false;
            unsigned pa = 0u, pb = 0xffffffffu;
            int ncand = 0;
#pragma unroll
            for (int s = 0; s < CPL; ++s) {
                const bool cand = pos[s] >= 0 && dist[s] == m;
                ncand += __popcll(__ballot(cand));
                if (cand) {
                    pb = min(pb, (unsigned)pos[s]);
                    if (rowof[s] < 0) pa = max(pa, (unsigned)pos[s] + 1u);
                }
            }
            int pick;
            if (ncand == 1) pick = (int)wave_umin32(pb);            // (one lane holds it; a 32-bit reduce is cheaper than locating it twice)
            else {
                const unsigned pam = wave_umax32(pa);
                pick = pam ? (int)pam - 1 : (int)wave_umin32(pb);
            }
            base = m;
            int jp = 0, jl = 0;
#pragma unroll
            for (int s = 0; s < CPL; ++s) {
                const unsigned long long b1 = __ballot(pos[s] == pick);
                if (b1) jp = 64 * s + __ffsll((long long)b1) - 1;
                const unsigned long long b2 = __ballot(pos[s] == live - 1);
                if (b2) jl = 64 * s + __ffsll((long long)b2) - 1;
            }
            const int rj = pick_i(rowof, jp);
            if (rj < 0) sink = jp; else i = rj;
#pragma unroll
            for (int s = 0; s < CPL; ++s) {
                const int j = lane + 64 * s;
                if (j == jl) pos[s] = pick;                        // the list's last entry moves into the freed place
                if (j == jp) { pos[s] = -1; seen[s] = true; }
            }
            --live;
        }
        // dual update (lsap.cpp:83-87)
        double dlt[CPL];
#pragma unroll
        for (int s = 0; s < CPL; ++s) dlt[s] = base - dist[s];
#pragma unroll
        for (int s = 0; s < CPL; ++s) {                           // row (lane, s): partner column colof[s] = lane' + 64 * s2
            const int cj = colof[s] >= 0 ? colof[s] : 0;
            double dl = 0.0;
            int sn = 0;
#pragma unroll
            for (int s2 = 0; s2 < CPL; ++s2) {
                const double d2 = __shfl(dlt[s2], cj & 63);
                const int n2 = __shfl((int)seen[s2], cj & 63);
                if ((cj >> 6) == s2) { dl = d2; sn = n2; }
            }
            if (lane + 64 * s < R && colof[s] >= 0 && sn) u[s] = u[s] + dl;
        }
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            if (seen[s]) v[s] = v[s] - dlt[s];
            if (lane + 64 * s == root) u[s] = u[s] + base;
        }
        int j = sink;
        for (;;) {                                                 // flip the path back to the root
            const int i2 = pick_i(pred, j);
            const int t = pick_i(colof, i2);
#pragma unroll
            for (int s = 0; s < CPL; ++s) {
                if (lane + 64 * s == j) rowof[s] = i2;
                if (lane + 64 * s == i2) colof[s] = j;
            }
            j = t;
            if (i2 == root) break;
        }
    }
#pragma unroll
    for (int s = 0; s < CPL; ++s)
        if (lane + 64 * s < nr) L.asg[lane + 64 * s] = tall ? rowof[s] : colof[s];
    wave_lds_sync();
    return true;
}

}  // namespace

struct EpochArgs {
    DevTrkHdr* hdr; DevTrack* trk; int* free_slots;
    float* mean; float* cov; float* gal_raw; float* gal_n;
    TrkDevParams prm;
    EpochDets dets;
    int f0, k, d_begin, dn_pad, nmax;     // frames [f0, f0 + k) of the group; first detection row of the epoch; padded row count
    int has_sm;                            // the prep kernel ran (features present)
    EpochScratch scr;
    EpochOut out;
    int lds_bytes;
    int commit_ext;                        // 1: the appended gallery rows are copied by gallery_commit_kernel after this launch (the list + its count stay in scr.appends)
    long long* prof;                       // AICAM_TRK_PHASES: shader-clock cycles per phase, accumulated by thread 0 (NULL: off)
};

namespace {

// min_cost_matching (linear_assignment.py:19-88) of rows[0..nr) x cols[0..nc) on the full matrices of the frame. Block-wide.
//   stage 1: sub[r][c] = maha > chi2 ? INFTY : app (linear_assignment.py:187-210), threshold max_cos
//   stage 2: sub[r][c] = iou, threshold max_iou
// Matched pairs are entered into mdet / mtrk. *err != 0 on an LSAP failure.
// Two forms.  FULL: the three [T, n] matrices (app, maha, iou).  LEAN (gated != NULL): ONE [T, n] matrix, the appearance cost behind the
// Mahalanobis gate (maha > chi2 ? INFTY : app -- what stage 1 reads, linear_assignment.py:187-210), and the IoU cost of a pair worked out
// where stage 2 asks for it (iou_pair: the expressions of the cost phase, so the same bits): a third of the LDS, which is what decides
// whether a frame's costs stay on the CU at all (T * n <= ~12 k instead of ~3 k entries beside the 512-track table).
struct FrameCosts { const float* app; const float* maha; const float* iou; float* sub_lds; int sub_floats; const float* gated; const float* mean_hbm; };

// 1 - IoU of track t's predicted box and detection j (matching.py:13-106 with track.py:133-151), as the cost phase computes it
__device__ __forceinline__ float iou_pair(const Lds& L, const float* mean_hbm, int t, int j);

// The assignment read off the sub-matrix when it is the ONLY optimum (block-wide; true = matches entered, nothing left to do).
// Take the short side's lines (rows when nr <= nc, else columns): R lines of C entries, every line must be assigned, to distinct
// entries.  A line whose minimum is above the threshold holds the clamp value everywhere (linear_assignment.py:58): whatever it
// gets is rejected at :76 and any free entry costs it the same.  If every OTHER line has a strict minimum (below its runner-up)
// and those minima sit in distinct places, then sum-of-line-minima is attained, and only by assignments that put each such line
// on its minimum: SciPy's answer restricted to the accepted pairs is exactly this, its tie rules never come into play.  (The
// solver's arithmetic is exact here: entries are fp32 values below 1 on a 2^-27 grid, sums of <= 512 of them fit a double.)
__shared__ long long s_tsub;            // AICAM_TRK_PHASES: sub-phase stamps of the cascade (thread 0)
#define SUBPH(prof, i) do { if ((prof) && threadIdx.x == 0) { const long long _t = clock64(); atomicAdd((unsigned long long*)&(prof)[8 + (i)], (unsigned long long)(_t - s_tsub)); s_tsub = _t; } } while (0)
// One ordered scan per line instead of ~R augmenting paths on one wavefront; anything else falls through to the LSAP.
// by_cols (square problems only): the COLUMNS are the lines.  Both sums of line minima bound a square problem's optimum from below, and
// rows whose nearest entries coincide often have columns whose nearest rows do not (two tracks that look like one detection, while the
// second detection looks like only one of them): a second chance before the one-wavefront LSAP (17 % of the bench clip's frames took it,
// at 56 k shader cycles each).
__device__ bool unique_optimum(const Lds& L, const float* sub, const int* cols, int nr, int nc, float maxd, bool by_cols = false) {
    const int tid = threadIdx.x;
    const bool tall = nr > nc || (by_cols && nr == nc);
    const int R = tall ? nc : nr, C = tall ? nr : nc;
    int* cnt = L.pred;                                          // scratch of the LSAP, free here
    for (int c = tid; c < C; c += BT) cnt[c] = 0;
    float m1 = __builtin_inff(), m2 = __builtin_inff();
    int arg = -1;
    if (C <= 64) {
        // a wave per line, a lane per entry: minimum by a DPP reduction on order-preserving keys of the fp32 entries (sign bit flipped,
        // negative values complemented: smaller value <=> smaller key), strict <=> exactly one lane holds it.  The
        // thread-per-line scan below is a dependent chain of C compare-selects on one wavefront: 3 k of a frame's 56 k shader cycles at
        // 30 x 30, twice when the row-wise check fails.
        const int lane = tid & 63, wv = tid >> 6;
        int* l_m = L.todo;                                      // line minimum (bits) and its place, or -1 when it is not strict
        int* l_a = L.pos;
        const int step = tall ? nc : 1;
        for (int l = wv; l < R; l += NW) {
            const float* p = tall ? sub + l : sub + (size_t)l * nc;
            const unsigned raw = lane < C ? __float_as_uint(p[(size_t)lane * step]) : 0x7f800000u;
            const unsigned xb = raw ^ ((raw >> 31) ? 0xffffffffu : 0x80000000u);
            const unsigned mk = wave_umin32(xb);
            const unsigned long long bal = __ballot(xb == mk);
            if (lane == 0) {
                const unsigned mb = mk ^ ((mk >> 31) ? 0x80000000u : 0xffffffffu);
                l_m[l] = (int)mb; l_a[l] = __popcll(bal) == 1 ? (int)__ffsll((long long)bal) - 1 : -1;
            }
        }
        __syncthreads();
        if (tid < R) {
            m1 = __uint_as_float((unsigned)l_m[tid]);
            arg = l_a[tid];
            m2 = arg >= 0 ? __builtin_inff() : m1;               // only `m1 < m2` is asked below
            if (arg < 0) arg = 0;
        }
    } else
    if (tid < R) {
        const float* p = tall ? sub + tid : sub + (size_t)tid * nc;
        const int step = tall ? nc : 1;
        for (int c0 = 0; c0 < C; c0 += 8) {                     // eight independent LDS reads in flight, then the ordered compare chain
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = p[(size_t)min(c0 + u, C - 1) * step];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float xv = c0 + u < C ? x[u] : __builtin_inff();
                if (xv < m1) { m2 = m1; m1 = xv; arg = c0 + u; }
                else if (xv < m2) m2 = xv;
            }
        }
    }
    __syncthreads();
    const bool live = tid < R && m1 <= maxd;
    if (live) atomicAdd(&cnt[arg], 1);
    __syncthreads();
    const int ok = !live || (m1 < m2 && cnt[arg] == 1);
    if (!__syncthreads_and(ok)) return false;
    if (live) {
        const int r = tall ? arg : tid, c = tall ? tid : arg;
        L.mdet[L.rows[r]] = cols[c];
        L.mtrk[cols[c]] = L.rows[r];
    }
    __syncthreads();
    return true;
}

__device__ void match_block(const Lds& L, const EpochArgs& a, const FrameCosts& fc, const int* cols, int nr, int nc, int n, bool stage2, int* err) {
    const float* app = fc.app;
    const float* maha = fc.maha;
    const float* iou = fc.iou;
    const float maxd = stage2 ? a.prm.max_iou : a.prm.max_cos, clamp = stage2 ? a.prm.clamp_iou : a.prm.clamp_cos;
    float* sub = nr * nc <= fc.sub_floats ? fc.sub_lds : a.scr.sub;
    for (int e = threadIdx.x; e < nr * nc; e += BT) {
        const int r = e / nc, c = e - r * nc;
        const size_t kk = (size_t)L.rows[r] * n + cols[c];
        float x;
        if (stage2) x = iou ? iou[kk] : iou_pair(L, fc.mean_hbm, L.rows[r], cols[c]);
        else x = fc.gated ? fc.gated[kk] : (maha[kk] > kChi2_4 ? kInfty : app[kk]);
        if (x > maxd) x = clamp;                                  // linear_assignment.py:58
        sub[e] = x;
    }
    __threadfence_block();
    __syncthreads();
    SUBPH(a.prof, 2);
    if (!a.prm.no_fast && (unique_optimum(L, sub, cols, nr, nc, maxd) || (nr == nc && unique_optimum(L, sub, cols, nr, nc, maxd, true)))) {
        if (threadIdx.x == 0) L.wcnt[NW + 1] += 1;
        SUBPH(a.prof, 3);
        return;
    }
    if (threadIdx.x == 0) L.wcnt[NW + 2] += 1;
    if (threadIdx.x < 64) {
        const int side = max(nr, nc);
        const bool ok = side <= 64 ? lsap_wave64(sub, nr, nc, L, threadIdx.x)
                      : side <= 128 ? lsap_wave_reg<2>(sub, nr, nc, L, threadIdx.x)
                      : side <= 256 ? lsap_wave_reg<4>(sub, nr, nc, L, threadIdx.x)
                                    : lsap_wave(sub, nr, nc, L, threadIdx.x);
        if (!ok && threadIdx.x == 0) *err = 2;
    }
    __syncthreads();
    if (*err) return;
    for (int r = threadIdx.x; r < nr; r += BT) {
        const int c = L.asg[r];
        if (c >= 0 && sub[(size_t)r * nc + c] <= maxd) {          // linear_assignment.py:76
            L.mdet[L.rows[r]] = cols[c];
            L.mtrk[cols[c]] = L.rows[r];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float iou_pair(const Lds& L, const float* mean_hbm, int t, int j) {
    const int slot = L.slot[t];
    const float* m = slot < KF_SLOTS ? L.kf + slot * 72 + 64 : mean_hbm + (size_t)slot * 8;
    const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
    float bw = 0.f, bh = m3;
    if (bh > 0.f) bw = m2 * bh; else bh = fmaxf(0.f, bh);
    const float bx = m0 - bw / 2.0f, by = m1 - bh / 2.0f;
    const float brx = bx + bw, bry = by + bh;
    const float* c = L.tlwh + j * 4;
    const float crx = c[0] + c[2], cry = c[1] + c[3];
    const float iw = fmaxf(0.f, fminf(brx, crx) - fmaxf(bx, c[0]));
    const float ih = fmaxf(0.f, fminf(bry, cry) - fmaxf(by, c[1]));
    const float inter = iw * ih;
    const float uni = bw * bh + c[2] * c[3] - inter;
    return 1.0f - inter / fmaxf(uni, 1e-7f);
}

typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) const int lds_cint;

// unique_optimum for ONE wavefront, straight off the frame's gated matrix G (LDS, [T, n]): rows = L.rows[0 .. nr), columns =
// L.und[0 .. nc), both <= 64.  K = 64 / lines lanes (a power of two) share a line: each scans every K-th entry -- smallest entry, its place,
// the runner-up -- eight entries in flight at a time (one wave: nothing else hides an LDS round trip), and the K partial results meet by
// xor-shuffles.  "The line's minimum is strict, and it is here" is what the block form's ordered scan answers too.
__device__ bool unique_wave(const Lds& L, lds_cfloat* G, int n, int nr, int nc, float maxd, float clamp, bool by_cols) {
    const int lane = threadIdx.x;
    const bool tall = nr > nc || (by_cols && nr == nc);          // tall: the columns are the lines
    const int R = tall ? nc : nr, C = tall ? nr : nc;
    int lgk = 0;
    while ((R << (lgk + 1)) <= 64) ++lgk;
    const int K = 1 << lgk, line = lane >> lgk, seg = lane & (K - 1);
    lds_cint* und = (lds_cint*)L.und;
    lds_cint* rows = (lds_cint*)L.rows;
    int* cnt = L.pred;
    if (lane < C) cnt[lane] = 0;
    const int ln = min(line, R - 1);
    const int fixed = tall ? und[ln] : rows[ln] * n;              // the line's column / its row offset
    float m1 = __builtin_inff(), m2 = __builtin_inff();
    int arg = 0;
    for (int c0 = seg; c0 < C; c0 += 8 * K) {
        int idx[8];
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) idx[u] = tall ? rows[min(c0 + u * K, C - 1)] : und[min(c0 + u * K, C - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = tall ? G[idx[u] * n + fixed] : G[fixed + idx[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float xv = x[u];
            if (xv > maxd) xv = clamp;                            // linear_assignment.py:58
            if (c0 + u * K >= C) xv = __builtin_inff();
            if (xv < m1) { m2 = m1; m1 = xv; arg = c0 + u * K; }
            else if (xv < m2) m2 = xv;
        }
    }
    for (int o = 1; o < K; o <<= 1) {                             // the K lanes of a line are neighbours: both sides end with the same triple
        const float p1 = __shfl_xor(m1, o), p2 = __shfl_xor(m2, o);
        const int pa = __shfl_xor(arg, o);
        if (p1 < m1) { m2 = fminf(m1, p2); m1 = p1; arg = pa; }
        else m2 = fminf(m2, p1);                                  // (p1 == m1: the minimum is not strict)
    }
    wave_lds_sync();
    const bool live = seg == 0 && line < R && m1 <= maxd;
    if (live) atomicAdd(&cnt[arg], 1);
    wave_lds_sync();
    const bool ok = !live || (m1 < m2 && cnt[arg] == 1);
    if (__ballot(!ok)) return false;
    if (live) {
        const int t = tall ? L.rows[arg] : L.rows[line];
        const int dj = tall ? fixed : L.und[arg];
        L.mdet[t] = dj;
        L.mtrk[dj] = t;
    }
    wave_lds_sync();
    return true;
}

// The cascade's levels on ONE wavefront (wave 0 of the block; the others wait at the caller's barrier), for frames with at most 64
// detections whose gated matrix is in LDS: level search, row list, unique-optimum checks straight off the matrix, (rarely) sub-matrix +
// the one-wavefront LSAP, and the list of unmatched detections -- all without a block barrier.  A level costs the block form eleven of
// them (~13 k shader cycles whatever its size), and a scene that keeps a hundred stale confirmed tracks alive walks twenty levels per
// frame.  Same lists in the same order, same entries, same predicate, same LSAP as the block form.  Stops at a level with more than 64
// tracks (or whose sub-matrix does not fit): the caller's block loop goes on from `cur`.  done = 1: the cascade is through.
__device__ void cascade_wave(const Lds& L, const EpochArgs& a, const FrameCosts& fc, int T, int n, int& cur, int& nund, int& done, int* err) {
    const int lane = threadIdx.x;
    const float maxd = a.prm.max_cos, clamp = a.prm.clamp_cos;
    lds_cfloat* G = (lds_cfloat*)fc.gated;
    int ud = lane < nund ? L.und[lane] : 0;
    done = 0;
    for (;;) {
        if (nund == 0) { done = 1; break; }
        unsigned lvk = 0xffffffffu;
        for (int t = lane; t < T; t += 64)
            if (L.state[t] == 2 && L.tsu[t] > cur && L.tsu[t] <= a.prm.max_age && L.feas[t]) lvk = min(lvk, (unsigned)L.tsu[t]);
        const unsigned lv = wave_umin32(lvk);
        SUBPH(a.prof, 0);
        if (lv == 0xffffffffu) { done = 1; break; }
        int nr = 0, myrow = 0;
        for (int t0 = 0; t0 < T; t0 += 64) {                       // the level's tracks in list order (block_compact's order)
            const int t = t0 + lane;
            const bool fl = t < T && L.state[t] == 2 && L.tsu[t] == (int)lv;
            const unsigned long long bal = __ballot(fl);
            if (fl) {
                const int p = nr + __popcll(bal & ((1ull << lane) - 1ull));
                if (p < 64) L.rows[p] = t;
            }
            nr += __popcll(bal);
        }
        const int nc = nund, ld = nc | 1;
        if (nr > 64 || nr * ld > fc.sub_floats) break;
        cur = (int)lv;
        wave_lds_sync();
        myrow = L.rows[min(lane, nr - 1)];
        SUBPH(a.prof, 1);
        const bool solved = !a.prm.no_fast && (unique_wave(L, G, n, nr, nc, maxd, clamp, false) || (nr == nc && unique_wave(L, G, n, nr, nc, maxd, clamp, true)));
        if (lane == 0) L.wcnt[NW + (solved ? 1 : 2)] += 1;
        SUBPH(a.prof, 3);
        if (!solved) {
            lds_float* sub = (lds_float*)fc.sub_lds;
            for (int r0 = 0; r0 < nr; r0 += 8) {                    // lane = column; eight rows' entries in flight at a time
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = G[__builtin_amdgcn_readlane(myrow, min(r0 + u, nr - 1)) * n + ud];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float xv = x[u];
                    if (xv > maxd) xv = clamp;                     // linear_assignment.py:58
                    if (r0 + u < nr && lane < nc) sub[(r0 + u) * ld + lane] = xv;
                }
            }
            wave_lds_sync();
            SUBPH(a.prof, 2);
            const bool ok = lsap_wave64(fc.sub_lds, nr, nc, L, lane, ld);
            if (!ok) { if (lane == 0) *err = 2; break; }
            if (lane < nr) {
                const int c = L.asg[lane];
                if (c >= 0 && fc.sub_lds[lane * ld + c] <= maxd) {   // linear_assignment.py:76
                    const int dj = L.und[c];
                    L.mdet[myrow] = dj;
                    L.mtrk[dj] = myrow;
                }
            }
            wave_lds_sync();
        }
        const bool keep = lane < nund && L.mtrk[ud] < 0;           // the detections still unmatched, order kept
        const unsigned long long bk = __ballot(keep);
        if (keep) L.und[__popcll(bk & ((1ull << lane) - 1ull))] = ud;
        nund = __popcll(bk);
        wave_lds_sync();
        ud = lane < nund ? L.und[lane] : 0;
        SUBPH(a.prof, 4);
    }
}

// track.py:70-74 as a ring: returns the ring position the new row goes to and advances (glen, ghead)
__device__ __forceinline__ int ring_push(int& glen, int& ghead, int gmax) {
    int pos;
    if (glen < gmax) {
        pos = ghead + glen;
        if (pos >= gmax) pos -= gmax;
        glen += 1;
    } else {
        pos = ghead;
        ghead = ghead + 1 == gmax ? 0 : ghead + 1;
    }
    return pos;
}

}  // namespace

#define PHASE(i) do { if (a.prof && threadIdx.x == 0) { const long long _t = clock64(); atomicAdd((unsigned long long*)&a.prof[i], (unsigned long long)(_t - t_ph)); t_ph = _t; } } while (0)

__global__ __launch_bounds__(TRK_DEV_TMAX) void trk_epoch_kernel(EpochArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_err;
    long long t_ph = a.prof ? clock64() : 0;
    const Lds L = lds_carve(smem, a.prm.cap, a.nmax, a.lds_bytes, a.k);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int gmax = a.prm.gmax, dim = a.prm.dim, KS = L.ks;
    // Kalman state: the low slots live in LDS for the whole epoch, the rest stays in HBM
    auto kP = [&](int slot) -> float* { return slot < KF_SLOTS ? L.kf + slot * 72 : a.cov + (size_t)slot * 64; };
    auto kM = [&](int slot) -> float* { return slot < KF_SLOTS ? L.kf + slot * 72 + 64 : a.mean + (size_t)slot * 8; };
    {
        const int ns = min(KF_SLOTS, a.prm.cap);
        for (int e = tid; e < ns * 64; e += BT) L.kf[(e >> 6) * 72 + (e & 63)] = a.cov[e];
        for (int e = tid; e < ns * 8; e += BT) L.kf[(e >> 3) * 72 + 64 + (e & 7)] = a.mean[e];
    }

    // ---- load the track table (uniform copies of the scalars in registers)
    int T = a.hdr->n_tracks, next_id = a.hdr->next_id, nfree = a.hdr->n_free;
    if (tid == 0) { s_err = 0; L.wcnt[NW + 1] = 0; L.wcnt[NW + 2] = 0; if (a.out.dbg_match && !a.out.dbg_stride) a.out.dbg_match[0] = 0; }
    if (tid < T) {
        const DevTrack t = a.trk[tid];
        L.id[tid] = t.id, L.state[tid] = t.state, L.hits[tid] = t.hits, L.age[tid] = t.age, L.tsu[tid] = t.tsu, L.cls[tid] = t.cls;
        L.conf[tid] = t.conf, L.slot[tid] = t.slot, L.glen[tid] = t.glen, L.ghead[tid] = t.ghead, L.glen0[tid] = t.glen;
        L.sm[tid] = a.has_sm ? tid : -1;                          // row of the epoch's suffix-minimum table
        L.napp[tid] = 0;
    }
    for (int i = tid; i < nfree; i += BT) L.free_slots[i] = a.free_slots[i];
    __syncthreads();
    PHASE(0);

    // the detections of a frame are requested one frame ahead (four global loads per detection, a round trip to L2 / HBM under the conv
    // kernels' traffic: they used to open every frame) and wait in registers
    int pn = a.dets.frame_n[a.f0], pd0 = a.dets.frame_d0[a.f0];
    floatx4 pb = {0.f, 0.f, 0.f, 0.f};
    float pconf = 0.f;
    int pcls = 0, phas = 0;
    auto fetch_dets = [&](int n_, int d0_) {
        if (tid < n_ && n_ <= TRK_DEV_NMAX) {
            pb = *reinterpret_cast<const floatx4*>(a.dets.tlwh + (size_t)(d0_ + tid) * 4);
            pconf = a.dets.conf[d0_ + tid];
            pcls = a.dets.cls[d0_ + tid];
            phas = (a.dets.feat_n != nullptr && (a.dets.valid == nullptr || a.dets.valid[d0_ + tid] != 0)) ? 1 : 0;
        }
    };
    fetch_dets(pn, pd0);
    int fi = 0, err_frame = -1;
    for (; fi < a.k; ++fi) {
        const int f = a.f0 + fi;
        const int n = pn, d0 = pd0;
        const int erow0 = d0 - a.d_begin;                        // first epoch row of this frame
        if (n > a.nmax || n > TRK_DEV_NMAX || erow0 + n > a.dn_pad) {     // (block-uniform: every thread read the same counts -- no barrier needed,
            if (tid == 0) s_err = 3;                                      //  and the one that stood here waited for the previous frame's output stores)
            err_frame = f;
            break;
        }
        // ---- detections of the frame -> LDS (detection.py:36-47 for xyah); Kalman predict of every track (tracker_core.py:44-49)
        if (tid < n) {
            const floatx4 b = pb;
            L.tlwh[tid * 4] = b[0], L.tlwh[tid * 4 + 1] = b[1], L.tlwh[tid * 4 + 2] = b[2], L.tlwh[tid * 4 + 3] = b[3];
            L.xyah[tid * 4] = b[0] + b[2] / 2.0f;
            L.xyah[tid * 4 + 1] = b[1] + b[3] / 2.0f;
            L.xyah[tid * 4 + 2] = b[3] > 0.f ? b[2] / b[3] : 0.f;
            L.xyah[tid * 4 + 3] = b[3];
            L.dconf[tid] = pconf;
            L.dcls[tid] = pcls;
            L.dhas[tid] = phas;
            L.mtrk[tid] = -1;
            L.und[tid] = tid;
        }
        if (fi + 1 < a.k) {                                       // the next frame's detections: in flight until the next iteration
            pn = a.dets.frame_n[f + 1], pd0 = a.dets.frame_d0[f + 1];
            fetch_dets(pn, pd0);
        }
        if (tid == 0 && a.out.dbg_match && a.out.dbg_stride) a.out.dbg_match[(size_t)f * a.out.dbg_stride] = 0;
        if (tid < T) { L.age[tid] += 1; L.tsu[tid] += 1; L.mdet[tid] = -1; }
        for (int t = wv; t < T; t += NW) {
            const int slot = L.slot[t];
            kf_predict_wave(kP(slot), kM(slot), lane);
        }
        __threadfence_block();
        __syncthreads();
        PHASE(1);

        // ---- cost rows of every track: squared Mahalanobis (kalman_filter.py:206-249), 1 - IoU (matching.py:13-106),
        //      min-over-gallery cosine distance (matching.py:144-217) out of the epoch's SM / GRAM tables.
        //      The three [T, n] matrices sit in LDS when they fit beside the assignment sub-matrix, else in HBM scratch.
        const int tn = T * n;
        // LEAN form (FrameCosts): one gated [T, n] matrix in LDS, 24 floats of gate data per track beside it while the pairs run, the
        // assignment sub-matrix in their place afterwards.  Otherwise (debug dumps of the three matrices, or T * n beyond the CU's LDS):
        // the three full matrices in HBM scratch.
        const bool cost_lds = !a.out.dbg_tn && T > 0 && n > 0 && tn + 24 * T <= L.arena_floats;
        float* c_app = cost_lds ? L.arena : a.scr.cost;            // lean: the gated matrix (appearance first, the gate applied in place)
        float* c_maha = cost_lds ? nullptr : a.scr.cost + (size_t)TRK_DEV_TMAX * TRK_DEV_NMAX;
        float* c_iou = cost_lds ? nullptr : a.scr.cost + 2 * (size_t)TRK_DEV_TMAX * TRK_DEV_NMAX;
        const FrameCosts fc{cost_lds ? nullptr : c_app, c_maha, c_iou, cost_lds ? L.arena + tn : L.arena, cost_lds ? L.arena_floats - tn : L.arena_floats,
                            cost_lds ? c_app : nullptr, a.mean};
        // Gate data of a track (innovation covariance, its Cholesky factor, the predicted box) are the same for all its pairs: ONE THREAD per
        // track works them out -- all tracks at once, 24 floats each in the still unused sub-matrix area -- and the pairs then run one per
        // thread with their appearance loads in flight.  (Before: one WAVE per track, every lane repeating the track's Cholesky: four
        // rounds of ~800 instructions at 30 tracks, 16 k of this phase's 20 k shader cycles.)  Same functions, same operations per value.
        const bool gate_lds = cost_lds;
        // feas[t]: some detection of the frame lies inside both of track t's gates.  A cascade level none of whose tracks has one is a no-op
        // (every row of its sub-matrix is the clamp value: nothing is accepted at linear_assignment.py:76, the unmatched detections stay as
        // they are) and is skipped below without its six block barriers; scenes that keep many stale confirmed tracks alive (max_age 70)
        // walk 20+ such levels per frame.  Set by the pair loop of the gate_lds form; the other form keeps every level (feas = 1).
        if (tid < T) L.feas[tid] = gate_lds ? 0 : 1;
        if (gate_lds) {
            float* gd = fc.sub_lds;
            if (tid < T) {
                const int slot = L.slot[tid];
                const float* P = kP(slot);
                const float* m = kM(slot);
                float S[4][4], Lc[4][4];
                innovation_cov(P, m[3], S);
                const bool ok = cholesky<4>(S, Lc);
                float* g = gd + tid * 24;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) g[4 * i + jj] = Lc[i][jj];
                const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
                float bw = 0.f, bh = m3;
                if (bh > 0.f) bw = m2 * bh; else bh = fmaxf(0.f, bh);
                g[16] = m0, g[17] = m1, g[18] = m2, g[19] = m3;
                g[20] = bw, g[21] = bh, g[22] = ok ? 1.f : 0.f;
            }
        }
        if (T > 0 && n > 0) {
            for (int e = tid; e < tn; e += BT) {                   // appearance: one (track, detection) pair per thread, all loads of a pair in flight together
                const int t = e / n, j = e - t * n;
                float v = kInfty;                                  // empty gallery / featureless detection (matching.py:148,175)
                if (L.glen[t] > 0 && L.dhas[j]) {
                    // gallery of the track inside the epoch: rows older than the epoch that survived `ev` evictions (suffix minimum
                    // SM[ev]) + the rows the epoch appended (never evicted inside it: k <= gmax)
                    const int napp = L.napp[t], smr = L.sm[t], g0 = L.glen0[t];
                    const int ev = max(0, g0 + napp - gmax);
                    const int erow = erow0 + j;
                    float x[TRK_KMAX + 1];
                    x[TRK_KMAX] = (smr >= 0 && ev < g0) ? a.scr.sm[((size_t)smr * (TRK_KMAX + 1) + ev) * a.dn_pad + erow] : kBig;
                    const unsigned short* nr_ = L.newrow + t * KS;
#pragma unroll
                    for (int q = 0; q < TRK_KMAX; ++q) x[q] = q < napp ? a.scr.gram[(size_t)nr_[q] * a.dn_pad + erow] : kBig;
                    float mn = x[TRK_KMAX];
#pragma unroll
                    for (int q = 0; q < TRK_KMAX; ++q) mn = fminf(mn, x[q]);
                    v = mn;
                }
                c_app[e] = v;
            }
            if (gate_lds) __syncthreads();                         // the gate data of every track are in LDS (their threads came here through the loop above)
            if (gate_lds)
            for (int e = tid; e < tn; e += BT) {                   // gating distance + IoU of the pair
                const int t = e / n, j = e - t * n;
                {
                    const float* g = fc.sub_lds + t * 24;
                    float Lc[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) Lc[i][jj] = g[4 * i + jj];
                    const float m0 = g[16], m1 = g[17], m2 = g[18], m3 = g[19], bw = g[20], bh = g[21];
                    const bool ok = g[22] != 0.f;
                    const float bx = m0 - bw / 2.0f, by = m1 - bh / 2.0f;
                    const float brx = bx + bw, bry = by + bh;
                    const float* z = L.xyah + j * 4;
                    float d[4], y[4];
                    d[0] = z[0] - m0, d[1] = z[1] - m1, d[2] = z[2] - m2, d[3] = z[3] - m3;
                    fwd_solve<4>(Lc, d, y);
                    float acc = y[0] * y[0];
                    acc = acc + y[1] * y[1];
                    acc = acc + y[2] * y[2];
                    acc = acc + y[3] * y[3];
                    const float mh = ok ? acc : __builtin_inff();
                    const float gx = mh > kChi2_4 ? kInfty : c_app[e];     // (this thread wrote c_app[e] above) linear_assignment.py:187-210
                    c_app[e] = gx;
                    if (gx <= a.prm.max_cos) L.feas[t] = 1;
                    (void)bx, (void)by, (void)brx, (void)bry;              // the IoU of a pair is worked out where stage 2 asks for it (iou_pair)
                }
            }
            if (!gate_lds)
            for (int t = wv; t < T; t += NW) {                     // gating distance + IoU: one wave per track, lanes over the detections
                const int slot = L.slot[t];
                const float* P = kP(slot);
                const float* m = kM(slot);
                float S[4][4], Lc[4][4];
                innovation_cov(P, m[3], S);
                const bool ok = cholesky<4>(S, Lc);
                const float m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
                float bw = 0.f, bh = m3;
                if (bh > 0.f) bw = m2 * bh; else bh = fmaxf(0.f, bh);
                const float bx = m0 - bw / 2.0f, by = m1 - bh / 2.0f;
                const float brx = bx + bw, bry = by + bh;
                for (int j = lane; j < n; j += 64) {
                    const float* z = L.xyah + j * 4;
                    float d[4], y[4];
                    d[0] = z[0] - m0, d[1] = z[1] - m1, d[2] = z[2] - m2, d[3] = z[3] - m3;
                    fwd_solve<4>(Lc, d, y);
                    float acc = y[0] * y[0];
                    acc = acc + y[1] * y[1];
                    acc = acc + y[2] * y[2];
                    acc = acc + y[3] * y[3];
                    const size_t o = (size_t)t * n + j;
                    c_maha[o] = ok ? acc : __builtin_inff();
                    const float* c = L.tlwh + j * 4;
                    const float crx = c[0] + c[2], cry = c[1] + c[3];
                    const float iw = fmaxf(0.f, fminf(brx, crx) - fmaxf(bx, c[0]));
                    const float ih = fmaxf(0.f, fminf(bry, cry) - fmaxf(by, c[1]));
                    const float inter = iw * ih;
                    const float uni = bw * bh + c[2] * c[3] - inter;
                    c_iou[o] = 1.0f - inter / fmaxf(uni, 1e-7f);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        PHASE(2);

        // ---- matching cascade over time_since_update = 1 .. max_age (linear_assignment.py:91-157)
        if (a.prof && tid == 0) s_tsub = clock64();
        int nund = n;
        if (T > 0 && n > 0) {
            int cur = 0;
            bool through = false;
            if (n <= 64 && fc.gated && !a.prm.no_wave) {           // the levels on one wavefront while they are small (cascade_wave)
                if (wv == 0) {
                    int done = 0;
                    cascade_wave(L, a, fc, T, n, cur, nund, done, &s_err);
                    if (lane == 0) { L.wcnt[NW + 3] = cur; L.wcnt[NW + 4] = nund; L.wcnt[NW + 5] = done; }
                }
                __threadfence_block();
                __syncthreads();
                cur = L.wcnt[NW + 3], nund = L.wcnt[NW + 4];
                through = L.wcnt[NW + 5] != 0 || s_err != 0;
            }
            if (!through)
            for (;;) {
                if (nund == 0) break;
                const bool conf_t = tid < T && L.state[tid] == 2;
                const int tv = (conf_t && L.tsu[tid] > cur && L.tsu[tid] <= a.prm.max_age && L.feas[tid]) ? L.tsu[tid] : 0x7fffffff;
                const int lv = block_min_int(tv, L.wcnt);             // the next level that can accept anything
                SUBPH(a.prof, 0);
                if (lv == 0x7fffffff) break;
                cur = lv;
                const int nr = block_compact(conf_t && L.tsu[tid] == lv, tid, L.rows, L.wcnt);
                SUBPH(a.prof, 1);
                match_block(L, a, fc, L.und, nr, nund, n, false, &s_err);
                if (s_err) break;
                const int dj = tid < nund ? L.und[tid] : -1;
                nund = block_compact(dj >= 0 && L.mtrk[dj] < 0, dj, L.und, L.wcnt);
                SUBPH(a.prof, 4);
            }
            PHASE(3);
            // ---- IoU stage: tentative tracks, then confirmed tracks that missed exactly this frame (tracker_core.py:138-166)
            if (!s_err) {
                const int n1 = block_compact(tid < T && L.state[tid] == 1, tid, L.rows, L.wcnt);
                const int n2 = block_compact(tid < T && L.state[tid] == 2 && L.mdet[tid] < 0 && L.tsu[tid] == 1, tid, L.rows + n1, L.wcnt);
                if (n1 + n2 > 0 && nund > 0) match_block(L, a, fc, L.und, n1 + n2, nund, n, true, &s_err);
            }
        }
        __syncthreads();
        if (s_err) { err_frame = f; break; }
        PHASE(4);

        // ---- lifecycle (tracker_core.py:63-81): unmatched detections in ascending order become new tracks
        const int U = block_compact(tid < n && L.mtrk[tid] < 0, tid, L.cols, L.wcnt);
        if (nfree < U || T + U > BT) { if (tid == 0) s_err = 1; }
        __syncthreads();
        if (s_err) { err_frame = f; break; }
        if (tid < T) {
            const int det = L.mdet[tid];
            if (det >= 0) {                                        // Track.update, track.py:82-104
                if (L.dhas[det]) {
                    int gl = L.glen[tid], gh = L.ghead[tid];
                    const int pos = ring_push(gl, gh, gmax);
                    L.glen[tid] = gl, L.ghead[tid] = gh;
                    const int na = L.napp[tid];
                    (void)pos;                                     // the ring position is re-derived at the end of the epoch (see there)
                    L.newrow[tid * KS + na] = (unsigned short)(erow0 + det);
                    L.napp[tid] = na + 1;
                }
                L.hits[tid] += 1;
                L.tsu[tid] = 0;
                L.conf[tid] = L.dconf[det];
                L.cls[tid] = L.dcls[det];
                if (L.state[tid] == 1 && L.hits[tid] >= a.prm.n_init) L.state[tid] = 2;
                if (a.out.dbg_match && (a.out.dbg_stride || fi == a.k - 1)) {
                    int* dm = a.out.dbg_match + (size_t)f * a.out.dbg_stride;
                    const int mi = atomicAdd(&dm[0], 1);
                    dm[1 + 2 * mi] = L.id[tid], dm[2 + 2 * mi] = det;
                }
            } else {                                               // Track.mark_missed, track.py:106-119
                if (L.state[tid] == 1) L.state[tid] = 3;
                else if (L.state[tid] == 2 && L.tsu[tid] > a.prm.max_age) L.state[tid] = 3;
            }
        }
        // Kalman update of the matched tracks + their new tlwh (track.py:133-151).  kf_update_wave gives a track a whole wave whose 64
        // lanes each repeat the innovation covariance, its Cholesky factor and two triangular solves (~600 instructions, four rounds at 30
        // tracks: 19.8 k of a frame's 68 k shader cycles).  Here the gain row K[i], S K[i] and the new mean element are worked out once per
        // (track, row) -- all tracks' rows at once, one per thread -- and the 64 covariance elements of a track then take four
        // multiply-adds each.  Same functions, same operations and order per value as kf_update_wave.
        if (72 * T <= L.arena_floats) {
            float* ks = L.arena;                                   // [T][8][9]: K[i][0..3], (S K[i])[0..3], new mean[i]  (the cost matrices are dead)
            for (int idx = tid; idx < 8 * T; idx += BT) {
                const int t = idx >> 3, i = idx & 7;
                const int det = L.mdet[t];
                if (det < 0) continue;
                const int slot = L.slot[t];
                const float* P = kP(slot);
                const float* m = kM(slot);
                const float* zz = L.xyah + det * 4;
                float S[4][4], Lc[4][4];
                innovation_cov(P, m[3], S);
                cholesky<4>(S, Lc);
                float bi[4], y[4], Ki[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) bi[q] = P[i * 8 + q];
                fwd_solve<4>(Lc, bi, y); bwd_solve(Lc, y, Ki);
                float* o = ks + idx * 9;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float u = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) u = u + S[q][c] * Ki[c];
                    o[q] = Ki[q], o[4 + q] = u;
                }
                float mi = m[i];
                {
                    float dot = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dot = dot + Ki[q] * (zz[q] - m[q]);
                    mi = mi + dot;
                }
                o[8] = mi;
            }
            __threadfence_block();
            __syncthreads();
            for (int idx = tid; idx < 64 * T; idx += BT) {
                const int t = idx >> 6, i = (idx >> 3) & 7, j = idx & 7;
                if (L.mdet[t] < 0) continue;
                const int slot = L.slot[t];
                float* P = kP(slot);
                const float* ki = ks + (t * 8 + i) * 9;
                const float* uj = ks + (t * 8 + j) * 9 + 4;
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = acc + ki[q] * uj[q];
                P[i * 8 + j] = P[i * 8 + j] - acc;
                if (j == 0) kM(slot)[i] = ki[8];
            }
            __threadfence_block();
            __syncthreads();
            if (tid < T && L.mdet[tid] >= 0) {
                const float* m = kM(L.slot[tid]);
                const float cx = m[0], cy = m[1], ar = m[2], hh = m[3];
                float w = 0.f, h2 = hh;
                if (hh > 0.f) w = ar * hh; else h2 = fmaxf(0.f, hh);
                float* o = L.tbox + tid * 4;
                o[0] = cx - w / 2.0f; o[1] = cy - h2 / 2.0f; o[2] = w; o[3] = h2;
            }
        } else
        for (int t = wv; t < T; t += NW) {
            const int det = L.mdet[t];
            if (det < 0) continue;
            const int slot = L.slot[t];
            const float mi = kf_update_wave(kP(slot), kM(slot), L.xyah + det * 4, lane);
            const float cx = __shfl(mi, 0), cy = __shfl(mi, 8), ar = __shfl(mi, 16), hh = __shfl(mi, 24);
            if (lane == 0) {
                float w = 0.f, h2 = hh;
                if (hh > 0.f) w = ar * hh; else h2 = fmaxf(0.f, hh);
                float* o = L.tbox + t * 4;
                o[0] = cx - w / 2.0f; o[1] = cy - h2 / 2.0f; o[2] = w; o[3] = h2;
            }
        }
        for (int r = wv; r < U; r += NW) {                         // _initiate_track, tracker_core.py:180-194
            const int det = L.cols[r];
            const int slot = L.free_slots[nfree - 1 - r];
            kf_initiate_wave(kP(slot), kM(slot), L.xyah + det * 4, lane);
            if (lane == 0) {
                const int ti = T + r;
                L.id[ti] = next_id + r, L.state[ti] = 1, L.hits[ti] = 1, L.age[ti] = 1, L.tsu[ti] = 0;
                L.cls[ti] = L.dcls[det], L.conf[ti] = L.dconf[det], L.slot[ti] = slot;
                L.sm[ti] = -1, L.glen0[ti] = 0, L.mdet[ti] = -1;
                int gl = 0, gh = 0, na = 0;
                if (L.dhas[det]) {
                    (void)ring_push(gl, gh, gmax);
                    L.newrow[ti * KS] = (unsigned short)(erow0 + det);
                    na = 1;
                }
                L.glen[ti] = gl, L.ghead[ti] = gh, L.napp[ti] = na;
            }
        }
        nfree -= U;
        next_id += U;
        if (a.out.dbg_tn && fi == a.k - 1 && tid == 0) { a.out.dbg_tn[0] = T; a.out.dbg_tn[1] = n; }
        __threadfence_block();
        __syncthreads();
        PHASE(5);

        // ---- outputs: confirmed tracks updated in this frame, list order (deepsort_tracker.py:126-141)
        {
            const int cnt = block_compact(tid < T && L.state[tid] == 2 && L.mdet[tid] >= 0, tid, L.rows, L.wcnt);
            if (tid == 0) a.out.n_tracks[f] = cnt;
            if (tid < cnt && tid < a.out.max_rows) {
                const int t = L.rows[tid];
                const float* b = L.tbox + t * 4;
                const float x1 = b[0], y1 = b[1];
                const float w = b[2] > 0.f ? b[2] : 0.f, h = b[3] > 0.f ? b[3] : 0.f;
                int* r = a.out.rows + ((size_t)f * a.out.max_rows + tid) * 6;
                r[0] = (int)rintf(x1), r[1] = (int)rintf(y1), r[2] = (int)rintf(x1 + w), r[3] = (int)rintf(y1 + h);   // round half to even
                r[4] = L.id[t], r[5] = L.cls[t];
                a.out.conf[(size_t)f * a.out.max_rows + tid] = L.conf[t];
            }
        }
        // ---- prune deleted tracks (tracker_core.py:75): their slots go back in list order, the table closes up
        {
            const int Tn = T + U;
            const bool dead = tid < Tn && L.state[tid] == 3;
            const int nd = block_compact(dead, tid < Tn ? L.slot[tid] : 0, L.free_slots + nfree, L.wcnt);
            nfree += nd;
            const int Tk = block_compact(tid < Tn && !dead, tid, L.rows, L.wcnt);
            int e_id = 0, e_state = 0, e_hits = 0, e_age = 0, e_tsu = 0, e_cls = 0, e_slot = 0, e_glen = 0, e_ghead = 0, e_sm = 0, e_napp = 0, e_g0 = 0;
            float e_conf = 0.f;
            unsigned short e_new[TRK_KMAX];
            if (tid < Tk) {
                const int s = L.rows[tid];
                e_id = L.id[s], e_state = L.state[s], e_hits = L.hits[s], e_age = L.age[s], e_tsu = L.tsu[s], e_cls = L.cls[s];
                e_slot = L.slot[s], e_glen = L.glen[s], e_ghead = L.ghead[s], e_sm = L.sm[s], e_napp = L.napp[s], e_g0 = L.glen0[s];
                e_conf = L.conf[s];
#pragma unroll
                for (int q = 0; q < TRK_KMAX; ++q) e_new[q] = q < KS ? L.newrow[s * KS + q] : (unsigned short)0;
            }
            __syncthreads();
            if (tid < Tk) {
                L.id[tid] = e_id, L.state[tid] = e_state, L.hits[tid] = e_hits, L.age[tid] = e_age, L.tsu[tid] = e_tsu, L.cls[tid] = e_cls;
                L.slot[tid] = e_slot, L.glen[tid] = e_glen, L.ghead[tid] = e_ghead, L.sm[tid] = e_sm, L.napp[tid] = e_napp, L.glen0[tid] = e_g0;
                L.conf[tid] = e_conf;
#pragma unroll
                for (int q = 0; q < TRK_KMAX; ++q) if (q < KS) L.newrow[tid * KS + q] = e_new[q];
            }
            T = Tk;
            __syncthreads();
        }
        PHASE(6);
    }

    // ---- gallery rows appended in this epoch: raw row (export) + unit row (cost kernels), one ring position each.
    //      The list is built HERE from the tracks that are alive at the end of the epoch, not queued frame by frame: slots are
    //      recycled LIFO, so a tentative track born in frame f, deleted in f + 1 and its slot reused in f + 2 of the same epoch
    //      left two queue entries for one (slot, position) -- the dead track's row and the new owner's -- written by different
    //      waves in this loop: whichever landed last became the new track's first gallery row (K >= 3 frames per epoch and spurious
    //      tentative tracks needed; found on the texture scene, where two identical runs disagreed).  A live track owns its slot
    //      alone, its `napp` appended rows are its newest (never evicted inside the epoch: k <= gmax), and the newest row of a
    //      ring is at ghead + glen - 1: entry q of track t goes to ghead + glen - napp + q.  Dead tracks' rows are not written.
    //      16-byte copies, four rows per wave in flight (dim % 4 == 0 on this path)
    __syncthreads();
    {
        const int myn = tid < T ? L.napp[tid] : 0;
        int incl = myn;                                          // inclusive scan over the wave, wave totals through wcnt
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) L.wcnt[wv] = incl;
        __syncthreads();
        int off = incl - myn, na = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = L.wcnt[i];
            if (i < wv) off += c;
            na += c;
        }
        if (myn > 0) {
            const int slot = L.slot[tid];
            int pos = L.ghead[tid] + L.glen[tid] - myn;           // >= 0: glen counts the appended rows
            if (pos >= gmax) pos -= gmax;
            for (int q = 0; q < myn; ++q) {
                int* e = a.scr.appends + (size_t)(off + q) * 3;
                e[0] = slot, e[1] = pos, e[2] = L.newrow[tid * KS + q];
                if (++pos == gmax) pos = 0;
            }
        }
        __threadfence_block();
        __syncthreads();
        const floatx4* fr = reinterpret_cast<const floatx4*>(a.dets.feat + (size_t)a.d_begin * dim);
        const floatx4* fn = reinterpret_cast<const floatx4*>(a.dets.feat_n + (size_t)a.d_begin * dim);
        floatx4* graw = reinterpret_cast<floatx4*>(a.gal_raw);
        floatx4* gn = reinterpret_cast<floatx4*>(a.gal_n);
        const int d4 = dim >> 2;
        // The copies themselves (480 rows x 4 KB per epoch at 30 detections per frame) are ONE CU's memory bandwidth when this block makes
        // them -- 125 us of the kernel's 657, whatever the loop's shape (four rows per wave in flight or sixteen loads per thread: the
        // same) -- so by default a many-block kernel makes them right behind this launch (gallery_commit_kernel) and this block only leaves
        // the list and its length.
        if (a.commit_ext) {
            if (tid == 0) a.scr.appends[3 * TRK_DEV_DNMAX] = na;
        } else {
            for (int i0 = wv * 4; i0 < na; i0 += NW * 4) {
                for (int c = lane; c < d4; c += 64) {
                    floatx4 xr[4], xn[4];
                    size_t dst[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = min(i0 + u, na - 1);
                        const int slot = a.scr.appends[i * 3], pos = a.scr.appends[i * 3 + 1], er = a.scr.appends[i * 3 + 2];
                        dst[u] = ((size_t)slot * gmax + pos) * d4 + c;
                        xr[u] = fr[(size_t)er * d4 + c];
                        xn[u] = fn[(size_t)er * d4 + c];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (i0 + u < na) { graw[dst[u]] = xr[u]; gn[dst[u]] = xn[u]; }
                    }
                }
            }
        }
        const int ns = min(KF_SLOTS, a.prm.cap);                  // Kalman state of the cached slots back to HBM
        for (int e = tid; e < ns * 64; e += BT) a.cov[e] = L.kf[(e >> 6) * 72 + (e & 63)];
        for (int e = tid; e < ns * 8; e += BT) a.mean[e] = L.kf[(e >> 3) * 72 + 64 + (e & 7)];
    }
    // ---- write the table back
    if (tid < T) {
        DevTrack t;
        t.id = L.id[tid], t.state = L.state[tid], t.hits = L.hits[tid], t.age = L.age[tid], t.tsu = L.tsu[tid], t.cls = L.cls[tid];
        t.conf = L.conf[tid], t.slot = L.slot[tid], t.glen = L.glen[tid], t.ghead = L.ghead[tid], t.pad[0] = 0, t.pad[1] = 0;
        a.trk[tid] = t;
    }
    for (int i = tid; i < nfree; i += BT) a.free_slots[i] = L.free_slots[i];
    PHASE(7);
    if (tid == 0) {
        a.hdr->n_tracks = T, a.hdr->next_id = next_id, a.hdr->n_free = nfree;
        a.hdr->err = s_err, a.hdr->err_frame = err_frame, a.hdr->frames_done = fi;
        a.hdr->n_fast += L.wcnt[NW + 1], a.hdr->n_lsap += L.wcnt[NW + 2];
        if (a.prof) { atomicAdd((unsigned long long*)&a.prof[13], (unsigned long long)L.wcnt[NW + 1]); atomicAdd((unsigned long long*)&a.prof[14], (unsigned long long)L.wcnt[NW + 2]); }
    }
}

// ------------------------------------------------------------------------------------------------ test entry: cascade on given matrices
// aic_match_cascade_device: the device cascade + LSAP on caller-provided cost matrices of ONE frame (parity tests against
// csrc/assoc_host.cpp, SciPy and tests/golden/assign.npz).  One block; matrices in global memory at scr.cost.
__global__ __launch_bounds__(TRK_DEV_TMAX) void trk_cascade_test_kernel(EpochArgs a, int T, int n, const int* state, const int* tsu,
                                                                        int* out_mdet, int* out_err, int lds_bytes, int stage1_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_err;
    const Lds L = lds_carve(smem, a.prm.cap, a.nmax, lds_bytes);
    const int tid = threadIdx.x;
    if (tid == 0) { s_err = 0; L.wcnt[NW + 1] = 0; L.wcnt[NW + 2] = 0; }
    const size_t stride = (size_t)TRK_DEV_TMAX * TRK_DEV_NMAX;
    // the gated matrix of the epoch kernel's lean form (FrameCosts), built here from the caller's matrices when it fits the LDS, so that
    // this entry walks the same code as the pipeline: cascade_wave for n <= 64, match_block's gated branch otherwise
    const int tn = T * n;
    const bool lean = tn > 0 && tn + 64 * 65 <= L.arena_floats;
    if (lean)
        for (int e = tid; e < tn; e += BT) L.arena[e] = a.scr.cost[stride + e] > kChi2_4 ? kInfty : a.scr.cost[e];
    const FrameCosts fc{a.scr.cost, a.scr.cost + stride, a.scr.cost + 2 * stride, lean ? L.arena + tn : L.arena, lean ? L.arena_floats - tn : L.arena_floats,
                        lean ? L.arena : nullptr, nullptr};
    if (tid < T) { L.state[tid] = state[tid]; L.tsu[tid] = tsu[tid]; L.mdet[tid] = -1; L.feas[tid] = 1; }
    if (tid < n) { L.mtrk[tid] = -1; L.und[tid] = tid; }
    __threadfence_block();
    __syncthreads();
    int nund = n;
    if (T > 0 && n > 0) {
        int cur = 0;
        bool through = false;
        if (n <= 64 && fc.gated && !a.prm.no_wave) {               // as in the epoch kernel: the small levels on one wavefront
            if (tid < 64) {
                int done = 0;
                cascade_wave(L, a, fc, T, n, cur, nund, done, &s_err);
                if (tid == 0) { L.wcnt[NW + 3] = cur; L.wcnt[NW + 4] = nund; L.wcnt[NW + 5] = done; }
            }
            __threadfence_block();
            __syncthreads();
            cur = L.wcnt[NW + 3], nund = L.wcnt[NW + 4];
            through = L.wcnt[NW + 5] != 0 || s_err != 0;
        }
        if (!through)
        for (;;) {
            if (nund == 0) break;
            const bool conf_t = tid < T && L.state[tid] == 2;
            const int tv = (conf_t && L.tsu[tid] > cur && L.tsu[tid] <= a.prm.max_age) ? L.tsu[tid] : 0x7fffffff;
            const int lv = block_min_int(tv, L.wcnt);
            if (lv == 0x7fffffff) break;
            cur = lv;
            const int nr = block_compact(conf_t && L.tsu[tid] == lv, tid, L.rows, L.wcnt);
            match_block(L, a, fc, L.und, nr, nund, n, false, &s_err);
            if (s_err) break;
            const int dj = tid < nund ? L.und[tid] : -1;
            nund = block_compact(dj >= 0 && L.mtrk[dj] < 0, dj, L.und, L.wcnt);
        }
        if (!s_err && !stage1_only) {
            const int n1 = block_compact(tid < T && L.state[tid] == 1, tid, L.rows, L.wcnt);
            const int n2 = block_compact(tid < T && L.state[tid] == 2 && L.mdet[tid] < 0 && L.tsu[tid] == 1, tid, L.rows + n1, L.wcnt);
            if (n1 + n2 > 0 && nund > 0) match_block(L, a, fc, L.und, n1 + n2, nund, n, true, &s_err);
        }
    }
    __syncthreads();
    if (tid < T) out_mdet[tid] = L.mdet[tid];
    if (tid == 0) { out_err[0] = s_err; out_err[1] = L.wcnt[NW + 1]; out_err[2] = L.wcnt[NW + 2]; }
}

// ------------------------------------------------------------------------------------------------ cross-camera gallery shard
// configs[4]: fp32 [t_max, 2 + dim] = (valid, track id, unit embedding of the newest gallery row) of the first
// t_max confirmed tracks in list order, straight from the HBM-resident table. One block; rows are copied 16 bytes per lane.
__global__ __launch_bounds__(TRK_DEV_TMAX) void gallery_shard_kernel(const DevTrkHdr* __restrict__ hdr, const DevTrack* __restrict__ trk,
                                                                    const float* __restrict__ gal_n, int gmax, int dim, float* __restrict__ out, int t_max) {
    __shared__ int wcnt[NW + 8];
    __shared__ int sel[TRK_DEV_TMAX];
    const int tid = threadIdx.x, T = hdr->n_tracks;
    const bool ok = tid < T && trk[tid].state == 2 && trk[tid].glen > 0;
    const int cnt = min(block_compact(ok, tid, sel, wcnt), t_max);
    const int w = 2 + dim;
    for (int r = tid >> 6; r < t_max; r += NW) {
        float* o = out + (size_t)r * w;
        if (r < cnt) {
            const DevTrack t = trk[sel[r]];
            int pos = t.ghead + t.glen - 1;
            if (pos >= gmax) pos -= gmax;
            const float* g = gal_n + ((size_t)t.slot * gmax + pos) * dim;
            for (int c = tid & 63; c < dim; c += 64) o[2 + c] = g[c];
            if ((tid & 63) == 0) { o[0] = 1.0f; o[1] = (float)t.id; }
        } else if ((tid & 63) == 0) {
            o[0] = 0.0f; o[1] = 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ cross-camera nearest neighbours
// configs[4] annotation pass ("an extra cosine_min_gallery pass") on the all-gathered shards, fp32
// [world, t_max, 2 + dim] = (valid, track id, unit embedding).  One block per row i of ANY rank (every rank computes the whole
// table, so the global-id policy of global_id.cpp needs no second exchange): distance max(0, 1 - <e_i, e_j>) (matching.py:136-141)
// to every valid row j of another rank, the products summed k-ascending in fp32 with separate multiply and add (-ffp-contract=off):
// d(i, j) and d(j, i) are the same bits on every rank.  Output: the nearest such row (ties: the lowest row), or -1.
__global__ __launch_bounds__(256) void gallery_nearest_kernel(const float* __restrict__ g, int world, int t_max, int dim,
                                                              int* __restrict__ ids, int* __restrict__ near_row, float* __restrict__ near_dist) {
    extern __shared__ float s_e[];                       // [dim] row i
    __shared__ unsigned long long s_best[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = 2 + dim, n = world * t_max, ri = i / t_max;
    const float* gi = g + (size_t)i * w;
    const bool valid_i = gi[0] > 0.5f;
    if (tid == 0) ids[i] = valid_i ? (int)gi[1] : -1;
    if (!valid_i) {                                       // block-uniform
        if (tid == 0) { near_row[i] = -1; near_dist[i] = kInfty; }
        return;
    }
    for (int k = tid; k < dim; k += 256) s_e[k] = gi[2 + k];
    __syncthreads();
    unsigned long long best = ~0ull;                      // (distance bits << 32 | row): distances are >= +0, so the bit pattern orders like the value
    for (int j = tid; j < n; j += 256) {
        if (j / t_max == ri) continue;
        const float* gj = g + (size_t)j * w;
        if (!(gj[0] > 0.5f)) continue;
        float dot = 0.f;
        int k = 0;
        for (; k + 2 <= dim; k += 2) {                    // rows are 8-byte aligned (w even, payload at +2 floats)
            const float2 b = *reinterpret_cast<const float2*>(gj + 2 + k);
            dot = dot + s_e[k] * b.x;
            dot = dot + s_e[k + 1] * b.y;
        }
        for (; k < dim; ++k) dot = dot + s_e[k] * gj[2 + k];
        const float d = cos_dist(dot);
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if (lane == 0) s_best[wv] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = s_best[0];
        for (int q = 1; q < 4; ++q) b = s_best[q] < b ? s_best[q] : b;
        near_row[i] = b == ~0ull ? -1 : (int)(unsigned)b;
        near_dist[i] = b == ~0ull ? kInfty : __uint_as_float((unsigned)(b >> 32));
    }
}

void launch_gallery_nearest(const float* gathered, int world, int t_max, int dim, int* ids, int* near_row, float* near_dist, hipStream_t s) {
    if (world * t_max <= 0) return;
    hipLaunchKernelGGL(gallery_nearest_kernel, dim3(world * t_max), dim3(256), (size_t)dim * 4, s, gathered, world, t_max, dim, ids, near_row, near_dist);
    KCHECK();
}

void launch_gallery_shard(const DevTrkHdr* hdr, const DevTrack* trk, const float* gal_n, int gmax, int dim, float* out, int t_max, hipStream_t s) {
    hipLaunchKernelGGL(gallery_shard_kernel, dim3(1), dim3(TRK_DEV_TMAX), 0, s, hdr, trk, gal_n, gmax, dim, out, t_max);
    KCHECK();
}

// ------------------------------------------------------------------------------------------------ gallery commit
// The rows an epoch appended to the galleries (list left by trk_epoch_kernel: slot, ring position, epoch row; length behind the list),
// raw + unit, one row per block and pass.
__global__ __launch_bounds__(256) void gallery_commit_kernel(const int* __restrict__ appends, const float* __restrict__ feat, const float* __restrict__ feat_n,
                                                             int d_begin, int dim, int gmax, float* __restrict__ gal_raw, float* __restrict__ gal_n) {
    const int na = min(appends[3 * TRK_DEV_DNMAX], TRK_DEV_DNMAX), d4 = dim >> 2;
    const floatx4* fr = reinterpret_cast<const floatx4*>(feat + (size_t)d_begin * dim);
    const floatx4* fn = reinterpret_cast<const floatx4*>(feat_n + (size_t)d_begin * dim);
    floatx4* graw = reinterpret_cast<floatx4*>(gal_raw);
    floatx4* gn = reinterpret_cast<floatx4*>(gal_n);
    for (int i = blockIdx.x; i < na; i += gridDim.x) {
        const int slot = appends[i * 3], pos = appends[i * 3 + 1], er = appends[i * 3 + 2];
        const size_t dst = ((size_t)slot * gmax + pos) * d4, src = (size_t)er * d4;
        for (int c = threadIdx.x; c < 2 * d4; c += 256) {
            if (c < d4) graw[dst + c] = fr[src + c];
            else gn[dst + c - d4] = fn[src + c - d4];
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static int epoch_lds_bytes() { return 159 * 1024; }

void launch_trk_epoch_prep(const DevTrkHdr* hdr, const DevTrack* trk, const float* gal_n, int gmax, int dim, int cap, const float* featn,
                           int dn, int dn_pad, int k, float* sm, float* gram, hipStream_t s) {
    if (dn <= 0) return;
    const long tasks = ((long)cap + dn_pad / 16) * (dn_pad / 32);
    // Blocks: enough for the tasks the epoch is LIKELY to have (about as many live tracks as detections per frame), nine wave-tasks
    // each, not one block per four tasks of the worst case (every slot of the table alive).  Every block of this launch has to find
    // a CU behind the one-block-per-CU conv kernels before the epoch kernel may start; with 512 blocks (225 of them with work at 30
    // tracks) the tracker chain was what the pipeline waited for in every run of the bench (9 831 frames/s, six runs), with 96 in none
    // (9 969, five runs), with 32 the launch itself gets too long (9 708).  AICAM_TRK_PREP_GRID=n: exactly n blocks.
    static const int grid_env = [] { const char* e = getenv("AICAM_TRK_PREP_GRID"); return e ? std::max(1, atoi(e)) : 0; }();
    const long likely = ((long)dn_pad / std::max(k, 1) + dn_pad / 16) * (dn_pad / 32);
    const int grid = grid_env ? grid_env : (int)std::min<long>(std::min<long>(512, (tasks + 3) / 4), std::max<long>(64, likely / 9));
    hipLaunchKernelGGL(trk_epoch_prep_kernel, dim3(grid), dim3(256), 0, s, hdr, trk, gal_n, gmax, dim, featn, dn, dn_pad, k, sm, gram);
    KCHECK();
}

static void epoch_attr() {
    static bool done = false;
    if (!done) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trk_epoch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, epoch_lds_bytes()));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trk_cascade_test_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, epoch_lds_bytes()));
        done = true;
    }
}

// AICAM_TRK_PHASES=1: per-phase shader-clock totals of the epoch kernel, printed at process exit
namespace {
struct PhaseProf {
    long long* d = nullptr;
    long launches = 0, frames = 0;
    PhaseProf() {
        if (getenv("AICAM_TRK_PHASES")) { (void)hipMalloc((void**)&d, 16 * 8); (void)hipMemset(d, 0, 16 * 8); }
    }
    ~PhaseProf() {
        if (!d) return;
        long long h[16];
        (void)hipDeviceSynchronize();
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || !frames) return;
        static const char* nm[8] = {"load table", "dets+predict", "cost rows", "cascade", "IoU stage", "lifecycle+KF update", "outputs+prune", "epoch tail"};
        fprintf(stderr, "[trk_phases] %ld epoch launches, %ld frames; shader cycles per frame:", launches, frames);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f |", nm[i], (double)h[i] / frames);
        fprintf(stderr, "\n[trk_phases] inside the cascade: level search %.0f | row list %.0f | sub-matrix %.0f | unique-optimum check %.0f | unmatched list (+ the LSAP when the check fails) %.0f\n",
                (double)h[8] / frames, (double)h[9] / frames, (double)h[10] / frames, (double)h[11] / frames, (double)h[12] / frames);
        fprintf(stderr, "[trk_phases] assignment problems: %lld read off as the unique optimum, %lld through the LSAP\n", h[13], h[14]);
    }
};
PhaseProf g_phase;
}  // namespace

void launch_trk_epoch(DevTrkHdr* hdr, DevTrack* trk, int* free_slots, float* mean, float* cov, float* gal_raw, float* gal_n,
                      const TrkDevParams& prm, const EpochDets& dets, int f0, int k, int d_begin, int dn_pad, int nmax, int has_sm,
                      const EpochScratch& scr, const EpochOut& out, hipStream_t s) {
    AIC_REQUIRE(prm.cap <= TRK_DEV_TMAX && nmax <= TRK_DEV_NMAX && k >= 1 && k <= TRK_KMAX, AIC_ERR_CAPACITY, "device association: shape beyond the epoch kernel");
    epoch_attr();
    EpochArgs a;
    a.hdr = hdr, a.trk = trk, a.free_slots = free_slots, a.mean = mean, a.cov = cov, a.gal_raw = gal_raw, a.gal_n = gal_n;
    a.prm = prm, a.dets = dets, a.f0 = f0, a.k = k, a.d_begin = d_begin, a.dn_pad = dn_pad, a.nmax = std::max(nmax, 1), a.has_sm = has_sm;
    a.scr = scr, a.out = out, a.lds_bytes = epoch_lds_bytes();
    static const bool commit_inline = getenv("AICAM_TRK_COMMIT_INLINE") != nullptr;       // A/B: the epoch kernel copies its appended rows itself
    a.commit_ext = (!commit_inline && dets.feat != nullptr && dets.feat_n != nullptr && scr.appends != nullptr && prm.dim % 4 == 0) ? 1 : 0;
    a.prof = g_phase.d;
    g_phase.launches += 1, g_phase.frames += k;
    hipLaunchKernelGGL(trk_epoch_kernel, dim3(1), dim3(TRK_DEV_TMAX), (size_t)epoch_lds_bytes(), s, a);
    KCHECK();
    if (a.commit_ext) {
        static const int cgrid = [] { const char* e = getenv("AICAM_TRK_COMMIT_GRID"); return e ? std::max(1, atoi(e)) : 256; }();
        hipLaunchKernelGGL(gallery_commit_kernel, dim3(cgrid), dim3(256), 0, s, scr.appends, dets.feat, dets.feat_n, d_begin, prm.dim, prm.gmax, gal_raw, gal_n);
        KCHECK();
    }
}

void launch_trk_cascade_test(const TrkDevParams& prm, const EpochScratch& scr, int T, int n, const int* state, const int* tsu,
                             int* out_mdet, int* out_err, int stage1_only, hipStream_t s) {
    AIC_REQUIRE(T <= TRK_DEV_TMAX && n <= TRK_DEV_NMAX, AIC_ERR_CAPACITY, "device cascade: at most 512 tracks x 512 detections");
    epoch_attr();
    EpochArgs a{};
    a.prm = prm, a.scr = scr, a.nmax = std::max(n, 1), a.lds_bytes = epoch_lds_bytes();
    hipLaunchKernelGGL(trk_cascade_test_kernel, dim3(1), dim3(TRK_DEV_TMAX), (size_t)epoch_lds_bytes(), s, a, T, n, state, tsu, out_mdet, out_err,
                       epoch_lds_bytes(), stage1_only);
    KCHECK();
}

}  // namespace aic
