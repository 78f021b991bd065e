// kernels_det.hip -- YOLOv8 head decode (K3) + candidate sort + greedy NMS + un-letterbox (K4).
//
// In the reference all of this is inside the TensorRT engine's NMS plugin; only its four output
// tensors are visible (src/detector/yolo_detector.py:44-54,108-112), followed on the host by the
// score filter (:131-135) and scale_bboxes (src/utils/image_processing.py:141-183).  Semantics are
// build decision D4, stated once in oracle/nets_oracle.py::nms:
//   candidate  <=> max-class logit >= logit(conf)        (fp32 compare on the raw logit)
//   order       = (logit descending, anchor index ascending)
//   suppression = same label and IoU > iou_thresh, greedy, at most max_det survivors
// Compiled with -ffp-contract=off so the IoU arithmetic rounds exactly like the oracle's NumPy.
#include "kernels.hpp"

namespace aic {

// ---- decode: one thread per (image, anchor). DFL softmax expectation per side (wave-free: the 16
// bins of a side sit in one thread's registers), arg-max class on raw logits.
__global__ void decode_kernel(const DetArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.batch * a.n_anchors) return;
    const int img = (int)(idx / a.n_anchors);
    const int an = (int)(idx - (long)img * a.n_anchors);
    int l = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.n_levels && an >= a.lvl[k].a0) l = k;
    const HeadLevel L = a.lvl[l];
    const int cell = an - L.a0;
    const int gy = cell / L.w, gx = cell - gy * L.w;
    const size_t pix = (size_t)img * L.h * L.w + cell;
    const float* d = L.box + pix * (4 * a.reg_max);
    float dist[4];
    if (L.box_decoded) {
        if (L.cls_reduced) return;              // both halves of this level came from the detect branches' own tails (ConvArgs::t_box / t_max)
    } else if (a.reg_max == 16) {   // the common case: 16-byte loads, bins in registers
        for (int sd = 0; sd < 4; ++sd) {
            const float4* v4 = reinterpret_cast<const float4*>(d + sd * 16);
            float v[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 x = v4[k]; v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w; }
            float mx = v[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
            float sum = 0.f, ex = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float e = a.fast_exp ? __builtin_amdgcn_exp2f((v[k] - mx) * 1.4426950408889634f) : expf(v[k] - mx);
                sum += e;
                ex += e * (float)k;
            }
            dist[sd] = ex / sum;
        }
    } else {
        for (int sd = 0; sd < 4; ++sd) {
            const float* v = d + sd * a.reg_max;
            float mx = v[0];
            for (int k = 1; k < a.reg_max; ++k) mx = fmaxf(mx, v[k]);
            float sum = 0.f, ex = 0.f;
            for (int k = 0; k < a.reg_max; ++k) {
                const float e = a.fast_exp ? __builtin_amdgcn_exp2f((v[k] - mx) * 1.4426950408889634f) : expf(v[k] - mx);
                sum += e;
                ex += e * (float)k;
            }
            dist[sd] = ex / sum;
        }
    }
    if (!L.box_decoded) {
        const float cx = (float)gx + 0.5f, cy = (float)gy + 0.5f, st = (float)L.stride;
        float* b = a.boxes + idx * 4;
        b[0] = (cx - dist[0]) * st;
        b[1] = (cy - dist[1]) * st;
        b[2] = (cx + dist[2]) * st;
        b[3] = (cy + dist[3]) * st;
    }
    if (L.cls_reduced) return;                  // max logit + label of this level came from the class branch's own tail (ConvArgs::t_max)
    const float* c = L.cls + pix * a.nc;
    float best = c[0];
    int arg = 0;
    if ((a.nc & 3) == 0) {
        const float4* c4 = reinterpret_cast<const float4*>(c);
        for (int k = 0; k < a.nc / 4; ++k) {
            const float4 x = c4[k];
            if (x.x > best) { best = x.x; arg = 4 * k; }       // first maximum wins, as np.argmax
            if (x.y > best) { best = x.y; arg = 4 * k + 1; }
            if (x.z > best) { best = x.z; arg = 4 * k + 2; }
            if (x.w > best) { best = x.w; arg = 4 * k + 3; }
        }
    } else {
        for (int k = 1; k < a.nc; ++k) {
            const float v = c[k];
            if (v > best) { best = v; arg = k; }
        }
    }
    a.max_logit[idx] = best;
    a.labels[idx] = arg;
}

__device__ __forceinline__ unsigned int ordered_bits(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone map float -> uint
}

__device__ __forceinline__ float iou_xyxy(const float4 p, const float4 q) {
    const float iw = fmaxf(0.f, fminf(p.z, q.z) - fmaxf(p.x, q.x));
    const float ih = fmaxf(0.f, fminf(p.w, q.w) - fmaxf(p.y, q.y));
    const float inter = iw * ih;
    const float area_p = (p.z - p.x) * (p.w - p.y);
    const float area_q = (q.z - q.x) * (q.w - q.y);
    return inter / fmaxf(area_p + area_q - inter, 1e-9f);
}

// ---- one block per image: select candidates, bitonic-sort their keys in LDS, greedy NMS against
// the kept list (<= max_det boxes in LDS), write the four output tensors + un-letterboxed boxes.
// Dynamic LDS: keys[P] (u64) | kept boxes[max_det] (float4) | kept labels[max_det] | flags[NT]
constexpr int NMS_THREADS = 1024;

__global__ __launch_bounds__(NMS_THREADS) void select_sort_nms_kernel(const DetArgs a, int P, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    float4* kbox = reinterpret_cast<float4*>(smem + (size_t)P * 8);
    int* klab = reinterpret_cast<int*>(kbox + a.max_det);
    int* flags = klab + a.max_det;
    // counters live at the end of the dynamic region: a static __shared__ in front of it would shift
    // the 16-byte alignment of the dynamic base (cdna_hip_programming.md Guideline 17)
    int& s_count = flags[NMS_THREADS];

    const int img = blockIdx.x, t = threadIdx.x;
    const float* ml = a.max_logit + (size_t)img * a.n_anchors;
    if (t == 0) s_count = 0;
    __syncthreads();
    // 1. select (unordered append; the sort fixes the order): one LDS atomic per wave and round, not one per candidate
    for (int i0 = 0; i0 < a.n_anchors; i0 += NMS_THREADS) {
        const int i = i0 + t;
        const float v = i < a.n_anchors ? ml[i] : 0.f;
        const bool pass = i < a.n_anchors && v >= a.logit_thr;
        const unsigned long long bal = __ballot(pass);
        int base = 0;
        if ((t & 63) == 0 && bal) base = atomicAdd(&s_count, __popcll(bal));
        base = __shfl(base, 0);
        if (pass) keys[base + __popcll(bal & ((1ull << (t & 63)) - 1ull))] =
            ((unsigned long long)ordered_bits(v) << 32) | (unsigned int)(0xFFFFFFFFu - (unsigned int)i);
    }
    __syncthreads();
    if (dbg == 1) return;                      // (AICAM_NMS_DBG, timing only: 1 = leave after the selection, 2 = after the sort)
    const int n = s_count;
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    for (int i = n + t; i < p2; i += NMS_THREADS) keys[i] = 0ull;   // sorts to the end
    __syncthreads();
    // 2. bitonic sort, descending
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < p2; i += NMS_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    if (t == 0) a.n_cand[img] = n;
    if (dbg == 2) return;
    // 3. greedy NMS in chunks of NMS_THREADS candidates (thread = candidate, sorted order = chunk, wave, lane).  A chunk
    //    first drops what the boxes kept so far suppress; then its 16 wave tiles take turns: everyone still waiting tests
    //    against the boxes the previous tile kept, and the tile's own 64 candidates are walked greedily with ballot /
    //    readlane only.  One block barrier per tile instead of two per kept box (16-wave barriers and the instruction
    //    issue of 1024 threads made each kept box cost ~3 500 cycles: half of this kernel with ~270 boxes kept).
    const float4* boxes = reinterpret_cast<const float4*>(a.boxes) + (size_t)img * a.n_anchors;
    const int* labels = a.labels + (size_t)img * a.n_anchors;
    int* s_nk = flags;                          // two alternating slots: kept count after wave tile w lives in s_nk[w & 1]
    const int wv = t >> 6, lane = t & 63;
    int nk = 0;                                 // boxes kept so far (block-uniform register copy)
    for (int c0 = 0; c0 < n && nk < a.max_det; c0 += NMS_THREADS) {
        const int ci = c0 + t;
        const bool have = ci < n;
        int anchor = 0, lab = -1;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have) {
            anchor = (int)(0xFFFFFFFFu - (unsigned int)(keys[ci] & 0xFFFFFFFFull));
            bx = boxes[anchor];
            lab = labels[anchor];
        }
        bool alive = have;
        int seen = 0;                           // this candidate has been tested against kbox[0 .. seen)
        const int wtiles = min(NMS_THREADS / 64, (n - c0 + 63) / 64);   // wave tiles that hold candidates (a trained detector: ~30 candidates, one tile;
                                                                        // the empty tiles were fifteen block barriers per frame)
        for (int w = 0; w < wtiles && nk < a.max_det; ++w) {
            if (wv >= w) {                      // (wave-uniform) earlier tiles of this chunk are settled
                for (int k = seen; k < nk && alive; ++k)
                    if (klab[k] == lab && iou_xyxy(kbox[k], bx) > a.iou_thr) alive = false;
                seen = nk;
            }
            if (wv == w) {                      // this wave's 64 candidates: greedy walk with wave operations only
                unsigned long long m = __ballot(alive);
                int cnt = nk;
                while (m && cnt < a.max_det) {
                    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                    float4 kb;
                    kb.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.x), l));
                    kb.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.y), l));
                    kb.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.z), l));
                    kb.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.w), l));
                    const int kl = __builtin_amdgcn_readlane(lab, l);
                    if (lane == l) {
                        kbox[cnt] = bx;
                        klab[cnt] = lab;
                        const size_t o = (size_t)img * a.max_det + cnt;
                        reinterpret_cast<float4*>(a.out_boxes)[o] = bx;
                        a.out_scores[o] = 1.0f / (1.0f + expf(-ml[anchor]));
                        a.out_labels[o] = lab;
                        if (a.out_boxes_orig) {
                            // image_processing.py:161-181: remove padding, divide by ratio, clip
                            float4 ob;
                            ob.x = fminf(fmaxf((bx.x - a.pad_w) / a.ratio, 0.f), (float)a.orig_w);
                            ob.y = fminf(fmaxf((bx.y - a.pad_h) / a.ratio, 0.f), (float)a.orig_h);
                            ob.z = fminf(fmaxf((bx.z - a.pad_w) / a.ratio, 0.f), (float)a.orig_w);
                            ob.w = fminf(fmaxf((bx.w - a.pad_h) / a.ratio, 0.f), (float)a.orig_h);
                            reinterpret_cast<float4*>(a.out_boxes_orig)[o] = ob;
                        }
                        alive = false;
                    }
                    if (alive && lab == kl && iou_xyxy(kb, bx) > a.iou_thr) alive = false;
                    m = __ballot(alive);
                    ++cnt;
                }
                if (lane == 0) s_nk[w & 1] = cnt;
            }
            __syncthreads();                    // the tile's kept boxes and count are visible
            nk = s_nk[w & 1];
        }
        __syncthreads();                        // (slot reuse across chunks)
    }
    if (t == 0) a.num_dets[img] = min(nk, a.max_det);
}

// ---- the tracker's detection filter on the device (deepsort_tracker.py:88-101): keep detections with conf >= min_confidence
// whose class is tracked, IN ORDER, and lay the survivors of a launch group out as the arrays crop + ReID + the association
// epochs consume -- all in HBM, nothing crosses to the host between NMS and ReID.
//   pass 1 (one wave per frame): rank of every kept detection inside its frame (ballot prefix), count per frame;
//   pass 2 (one wave per frame): first row of the frame = sum of the counts before it (a wave reduction over <= 512 values),
//          then one thread per kept detection writes xyxy (the crop box, :148), tlwh (:185-186), conf, class, frame index.
__global__ __launch_bounds__(64) void det_filter_rank_kernel(const DetFilterArgs a) {
    const int f = blockIdx.x, lane = threadIdx.x;
    const int nd = min(a.num_dets[f], a.max_det);
    int cnt = 0;
    for (int i0 = 0; i0 < nd; i0 += 64) {
        const int i = i0 + lane;
        bool keep = false;
        if (i < nd) {
            const float sc = a.scores[(size_t)f * a.max_det + i];
            const int c = a.labels[(size_t)f * a.max_det + i];
            keep = sc >= a.min_conf && c >= 0 && c < 128 && ((a.mask[c >> 6] >> (c & 63)) & 1ull);
        }
        const unsigned long long bal = __ballot(keep);
        if (i < nd) a.rank[(size_t)f * a.max_det + i] = keep ? cnt + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
        cnt += __popcll(bal);
    }
    if (lane == 0) a.frame_n[f] = cnt;
}

__global__ __launch_bounds__(64) void det_filter_scatter_kernel(const DetFilterArgs a) {
    const int f = blockIdx.x, lane = threadIdx.x;
    int before = 0;
    for (int g = lane; g < f; g += 64) before += a.frame_n[g];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    const int mine = a.frame_n[f];
    if (lane == 0) {
        a.frame_d0[f] = before;
        if (f == a.batch - 1) { a.total[0] = min(before + mine, a.cap); a.total[1] = before + mine; }   // [0]: rows present, [1]: rows the filter passed
    }
    const int nd = min(a.num_dets[f], a.max_det);
    for (int i = lane; i < nd; i += 64) {
        const int rk = a.rank[(size_t)f * a.max_det + i];
        const int row = before + rk;
        if (rk < 0 || row >= a.cap) continue;
        const float4 b = *reinterpret_cast<const float4*>(a.boxes + ((size_t)f * a.max_det + i) * 4);
        *reinterpret_cast<float4*>(a.xyxy + (size_t)row * 4) = b;
        *reinterpret_cast<float4*>(a.tlwh + (size_t)row * 4) = make_float4(b.x, b.y, b.z - b.x, b.w - b.y);
        a.conf[row] = a.scores[(size_t)f * a.max_det + i];
        a.cls[row] = a.labels[(size_t)f * a.max_det + i];
        a.frame_of[row] = f;
    }
}

void launch_det_filter(const DetFilterArgs& a, hipStream_t s) {
    if (a.batch <= 0) return;
    hipLaunchKernelGGL(det_filter_rank_kernel, dim3(a.batch), dim3(64), 0, s, a);
    KCHECK();
    hipLaunchKernelGGL(det_filter_scatter_kernel, dim3(a.batch), dim3(64), 0, s, a);
    KCHECK();
}

void launch_decode(const DetArgs& a, hipStream_t s) {
    const long tot = (long)a.batch * a.n_anchors;
    if (tot <= 0) return;
    bool left = a.n_levels <= 0;                // every level already decoded by the detect branches' own tails (ConvArgs::t_max / t_box)?
    for (int l = 0; l < a.n_levels; ++l) left = left || !(a.lvl[l].cls_reduced && a.lvl[l].box_decoded);
    if (!left) return;
    hipLaunchKernelGGL(decode_kernel, dim3(ceil_div(tot, 128)), dim3(128), 0, s, a);
    KCHECK();
}

void launch_select_sort_nms(const DetArgs& a, hipStream_t s) {
    if (a.batch <= 0) return;
    int P = 1;
    while (P < a.n_anchors) P <<= 1;
    const size_t lds = (size_t)P * 8 + (size_t)a.max_det * (16 + 4) + NMS_THREADS * 4 + 16;
    AIC_REQUIRE(lds <= 160 * 1024 - 64, AIC_ERR_CAPACITY, "too many anchors / max_det for the NMS LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(select_sort_nms_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        attr_set = true;
    }
    static const int dbg = [] { const char* e = getenv("AICAM_NMS_DBG"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL(select_sort_nms_kernel, dim3(a.batch), dim3(NMS_THREADS), lds, s, a, P, dbg);
    KCHECK();
}

}  // namespace aic
