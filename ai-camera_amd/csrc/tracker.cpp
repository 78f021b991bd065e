// tracker.cpp -- TrackerCore / Track lifecycle on the host, numerics on the GPU.
//
// Mirrors src/tracker/core/tracker_core.py:11-198, track.py:16-171, linear_assignment.py:19-212
// and the output formatting of src/tracker/deepsort_tracker.py:126-141.  What runs where:
//   GPU  (kernels_trk.hip): Kalman predict / update / initiate, squared Mahalanobis gating
//        distances, 1-IoU, min-over-gallery cosine distance -- the FULL [T,N] matrices once per
//        frame (the reference recomputes sub-blocks per cascade level with identical values);
//   host (this file, lsap.cpp): hit/age counters, track states, the age-ordered matching cascade,
//        thresholding and the linear assignment -- integer decisions that must match the
//        reference exactly.
// Per frame: one packed H2D, 5 small kernels, one D2H of the three cost matrices, host matching,
// one packed H2D, 3 kernels, one D2H of the updated boxes.
#include "tracker.hpp"

#include <chrono>
#include <cstdio>

#include <algorithm>
#include <cmath>

namespace aic {

static const float kInfty = 1e5f;                              // linear_assignment.py:9

Tracker::Tracker(Device& d, const aic_tracker_params& p) : dev(&d), prm(p) {
    cap = p.max_tracks > 0 ? p.max_tracks : 512;
    unlimited = p.nn_budget <= 0;
    gmax = unlimited ? 256 : p.nn_budget;
    next_id = p.first_track_id > 0 ? p.first_track_id : 1;
    d.use();
    d_mean.alloc((size_t)cap * 8);
    d_cov.alloc((size_t)cap * 64);
    for (int s = cap - 1; s >= 0; --s) free_slots.push_back(s);
    if (p.feature_dim > 0) ensure_dim(p.feature_dim);
}

void Tracker::ensure_dim(int d) {
    if (dim == d) return;
    AIC_REQUIRE(dim == 0, AIC_ERR_INVALID, "feature dimension changed between updates");
    AIC_REQUIRE(d > 0 && d <= 1024, AIC_ERR_INVALID, "feature dimension must be in 1..1024");
    dim = d;
    d_gal_raw.alloc((size_t)cap * gmax * dim);
    d_gal_n.alloc((size_t)cap * gmax * dim);
}

// nn_budget=None (track.py:70-74 never pops): the rings never wrap, so growing = re-striding rows [slot][0..glen) in place
// order. Doubles the per-track capacity; called BEFORE update() mutates anything, so a failure leaves the tracker intact.
void Tracker::grow_galleries() {
    const int kMaxRows = 16384;
    AIC_REQUIRE(gmax < kMaxRows, AIC_ERR_CAPACITY, "unlimited gallery (nn_budget=None) reached 16384 rows per track: set an nn_budget");
    const int g2 = gmax * 2;
    dev->use();
    hipStream_t s = dev->s_trk;
    DevBuf<float> raw2((size_t)cap * g2 * dim), n2((size_t)cap * g2 * dim);
    const size_t spitch = (size_t)gmax * dim * 4, dpitch = (size_t)g2 * dim * 4;
    HIP_CHECK(hipMemcpy2DAsync(raw2.p, dpitch, d_gal_raw.p, spitch, spitch, cap, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpy2DAsync(n2.p, dpitch, d_gal_n.p, spitch, spitch, cap, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    d_gal_raw = std::move(raw2);
    d_gal_n = std::move(n2);
    gmax = g2;
}

void Tracker::predict() {   // tracker_core.py:44-49 -> track.py:76-80
    if (use_device()) { dev_predicts += 1; return; }     // device path: the epoch kernel predicts at the start of the frame
    to_host();
    if (pre_predicted) {                       // the device side of this predict already ran inside the previous frame's launch
        pre_predicted = false;
        for (auto& t : tracks) { t.age += 1; t.tsu += 1; }
        return;
    }
    if (pending_predict) flush_predict();      // two predicts in a row: run the first one now
    for (auto& t : tracks) { t.age += 1; t.tsu += 1; }
    pending_predict = !tracks.empty();         // the kernel runs fused into the next association launch
}

void Tracker::flush_predict() {
    if (!pending_predict) return;
    pending_predict = false;
    dev->use();
    const int T = (int)tracks.size();
    if (!T) return;
    hipStream_t s = dev->s_trk;
    HIP_CHECK(hipStreamSynchronize(s));        // h_slots may still be the source of an earlier copy
    h_slots.ensure((size_t)T);
    d_slots.ensure((size_t)T);
    for (int i = 0; i < T; ++i) h_slots.p[i] = tracks[i].slot;
    HIP_CHECK(hipMemcpyAsync(d_slots.p, h_slots.p, (size_t)T * 4, hipMemcpyHostToDevice, s));
    Prof pr(*dev, PROF_TRK, s, 0, (double)T * 72 * 4 * 2);
    launch_kf_predict(d_mean.p, d_cov.p, d_slots.p, T, s);
}

// linear_assignment.py:91-157 + tracker_core.py:83-177 on precomputed full matrices: assoc_host.cpp (HIP-free, built under
// ASan/UBSan by tools/asan_host.sh).
void Tracker::match(int T, int N, const float* app, const float* maha, const float* iou,
                    std::vector<std::pair<int, int>>& matches, std::vector<int>& unmatched_t,
                    std::vector<int>& unmatched_d) {
    std::vector<int> state(T), tsu(T);
    for (int i = 0; i < T; ++i) state[i] = tracks[i].state, tsu[i] = tracks[i].tsu;
    cascade_match(T, N, state.data(), tsu.data(), app, maha, iou, prm.max_cosine_distance, prm.max_iou_distance, prm.max_age,
                  matches, unmatched_t, unmatched_d);
}

static inline int round_half_even(float v) { return (int)std::nearbyintf(v); }

void Tracker::resolve_pending(bool need_sync) {
    if (!pend.active) return;
    if (need_sync) HIP_CHECK(hipStreamSynchronize(dev->s_trk));
    resolved.clear();
    const float* tl = h_tlwh2[pend.buf].p;
    for (const OutMeta& m : pend.meta) {
        const float* b = tl + (size_t)m.k * 4;
        const float x1 = b[0], y1 = b[1];
        const float w = b[2] > 0.f ? b[2] : 0.f, h = b[3] > 0.f ? b[3] : 0.f;
        TrackOut o;
        o.x1 = round_half_even(x1), o.y1 = round_half_even(y1);
        o.x2 = round_half_even(x1 + w), o.y2 = round_half_even(y1 + h);
        o.id = m.id, o.cls = m.cls, o.conf = m.conf;
        resolved.push_back(o);
    }
    pend.active = false;
}

namespace {
struct TrkTimes {
    double prep = 0, gpu = 0, host = 0, commit = 0, copy = 0, cascade = 0; long n = 0;
    bool on = getenv("AICAM_TRK_TIMES") != nullptr;
    ~TrkTimes() {
        if (on && n) fprintf(stderr, "[trk_times] frames %ld: prep+launch %.1f us, wait for the association rows %.1f, host match+lifecycle %.1f (of which: previous outputs + copy of the rows %.1f, cascade %.1f), commit issue %.1f\n",
                             n, 1e6 * prep / n, 1e6 * gpu / n, 1e6 * host / n, 1e6 * copy / n, 1e6 * cascade / n, 1e6 * commit / n);
    }
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} g_trk_times;
}  // namespace

void Tracker::update(const float* det_tlwh, const float* conf, const int32_t* cls, const float* feat, int feat_mem,
                     const uint8_t* has_feat, int n, int dim_in, const float* feat_n, const NextDets* nx) {
    dev->use();
    if (use_device() && nx == nullptr) {
        if (feat != nullptr && n > 0) ensure_dim(dim_in);
        if (dev_predicts == 1 && dev_capable()) {          // the reference's protocol: one predict(), one update()
            dev_predicts = 0;
            update_device(det_tlwh, conf, cls, feat, feat_mem, has_feat, n, dim_in);
            return;
        }
        const int replay = dev_predicts;                    // unusual call sequence: replay it on the host path
        dev_predicts = 0;
        const bool keep = dev_assoc;
        dev_assoc = false;
        try {
            for (int i = 0; i < replay; ++i) predict();
            update(det_tlwh, conf, cls, feat, feat_mem, has_feat, n, dim_in, feat_n, nx);
        } catch (...) { dev_assoc = keep; throw; }
        dev_assoc = keep;
        return;
    }
    to_host();
    const double tt0 = g_trk_times.on ? TrkTimes::now() : 0.0;
    double tt1 = tt0, tt2 = tt0, tt3 = tt0;
    hipStream_t s = dev->s_trk;
    const int T = (int)tracks.size();
    const bool any_feat = feat != nullptr && n > 0;
    if (any_feat) ensure_dim(dim_in);
    AIC_REQUIRE(n >= 0, AIC_ERR_INVALID, "negative detection count");
    AIC_REQUIRE(n <= 1536, AIC_ERR_CAPACITY, "more than 1536 detections in one frame (association kernel LDS rows)");
    if (unlimited && dim > 0) {                // every capacity check happens before the first mutation of host or device state
        int longest = 0;
        for (const auto& t : tracks) longest = std::max(longest, t.glen);
        if (longest + 1 > gmax) grow_galleries();
    }
    std::vector<uint8_t> hf(n, any_feat ? 1 : 0);
    if (any_feat && has_feat) std::copy(has_feat, has_feat + n, hf.begin());

    // detection.py:36-47 on the host, fp32
    std::vector<float> xyah((size_t)n * 4);
    for (int j = 0; j < n; ++j) {
        const float x = det_tlwh[j * 4], y = det_tlwh[j * 4 + 1], w = det_tlwh[j * 4 + 2], h = det_tlwh[j * 4 + 3];
        xyah[j * 4] = x + w / 2.0f;
        xyah[j * 4 + 1] = y + h / 2.0f;
        xyah[j * 4 + 2] = h > 0.f ? w / h : 0.f;
        xyah[j * 4 + 3] = h;
    }

    // ---- device features
    const float* d_featp = nullptr;
    if (any_feat) {
        if (feat_mem == AIC_DEVICE) d_featp = feat;
        else {
            d_feat.ensure((size_t)n * dim);
            HIP_CHECK(hipMemcpyAsync(d_feat.p, feat, (size_t)n * dim * 4, hipMemcpyHostToDevice, s));
            d_featp = d_feat.p;
        }
    }
    const float* d_featn = feat_n;
    if (any_feat && !d_featn) {                 // rows / max(||row||, 1e-7)  (matching.py:126-130), once per frame
        d_detn.ensure((size_t)n * dim);
        launch_normalize_rows(d_featp, d_detn.p, n, dim, s);
        d_featn = d_detn.p;
    }

    // ---- packed per-frame parameters: slots[T] glen[T] | tlwh[n*4] xyah[n*4] | has[n]
    const size_t off_slots = 0, off_glen = (size_t)T * 4, off_tlwh = (size_t)T * 8;
    const size_t off_xyah = off_tlwh + (size_t)n * 16, off_has = off_xyah + (size_t)n * 16;
    const size_t stage_bytes = ((off_has + n + 15) / 16) * 16;
    if (pending_predict && !(T > 0 && n > 0)) flush_predict();   // nothing to fuse it into
    last_t = T, last_n = n;
    last_app.assign((size_t)T * n, kInfty);
    last_maha.assign((size_t)T * n, 0.f);
    last_iou.assign((size_t)T * n, kInfty);
    for (auto& tr : tracks) { tr.c_kind = 0; tr.c_appos = -1; }
    const bool have_rows = pre_rows;           // requested by the previous frame's launch (pipelined form)
    if (have_rows) {
        AIC_REQUIRE(pre_T == T && pre_n == n && T > 0 && n > 0, AIC_ERR_RUNTIME, "pipelined tracker: the next frame differs from the one announced");
        pre_rows = false;
    }
    if (T > 0 && n > 0) {
        h_stage.ensure(stage_bytes + 16);
        d_stage.ensure(stage_bytes + 16);
        int* hs = reinterpret_cast<int*>(h_stage.p);
        for (int i = 0; i < T; ++i) { hs[i] = tracks[i].slot; hs[T + i] = tracks[i].glen; }
        std::memcpy(h_stage.p + off_tlwh, det_tlwh, (size_t)n * 16);
        std::memcpy(h_stage.p + off_xyah, xyah.data(), (size_t)n * 16);
        std::memcpy(h_stage.p + off_has, hf.data(), n);
        // The per-frame parameter block (a few hundred bytes) is read by the kernels straight from pinned host memory:
        // one PCIe read per wave instead of a blit launch on the critical chain (AICAM_TRK_COPY=1 restores the copy).
        static const bool zero_copy = getenv("AICAM_TRK_COPY") == nullptr;
        const char* pbase = h_stage.p;
        if (!zero_copy) {
            HIP_CHECK(hipMemcpyAsync(d_stage.p, h_stage.p, stage_bytes, hipMemcpyHostToDevice, s));
            pbase = d_stage.p;
        }
        const int* d_slots = reinterpret_cast<const int*>(pbase + off_slots);
        const int* d_glen = reinterpret_cast<const int*>(pbase + off_glen);
        const float* d_tl = reinterpret_cast<const float*>(pbase + off_tlwh);
        const float* d_xy = reinterpret_cast<const float*>(pbase + off_xyah);
        const unsigned char* d_has = reinterpret_cast<const unsigned char*>(pbase + off_has);
        const size_t tn = (size_t)T * n;
        d_cost.ensure(3 * tn);
        h_cost.ensure(3 * tn);
        static const bool fused = getenv("AICAM_TRK_SPLIT") == nullptr;   // one launch per frame; AICAM_TRK_SPLIT=1: the two-kernel form
        {
            Prof pr(*dev, PROF_TRK, s, any_feat ? 2.0 * T * gmax * (double)n * dim : 0.0,
                    any_feat ? ((double)T * gmax + n) * dim * 4 : 0.0);
            if (have_rows) {
                // nothing to launch
            } else if (fused) {
                // gating + IoU + appearance rows of every track in ONE launch, written straight into pinned host memory
                float* out = zero_copy ? h_cost.p : d_cost.p;
                launch_trk_assoc_all(d_mean.p, d_cov.p, d_slots, d_glen, T, pending_predict ? 1 : 0, d_tl, d_xy,
                                     (any_feat && dim > 0) ? d_gal_n.p : nullptr, gmax, dim, d_featn, d_has, n, out, out + tn, out + 2 * tn, s);
                pending_predict = false;
            } else {
                launch_trk_assoc(d_mean.p, d_cov.p, d_slots, T, pending_predict ? 1 : 0, d_tl, d_xy, n, d_cost.p, d_cost.p + tn,
                                 d_cost.p + 2 * tn, s);
                pending_predict = false;
                if (any_feat && dim > 0)
                    launch_cosine_min_mfma(d_gal_n.p, d_slots, d_glen, T, gmax, dim, d_featn, d_has, n, d_cost.p, s);
            }
        }
        if (!(fused && zero_copy) && !have_rows) HIP_CHECK(hipMemcpyAsync(h_cost.p, d_cost.p, 3 * tn * 4, hipMemcpyDeviceToHost, s));
        if (g_trk_times.on) tt1 = TrkTimes::now();
        HIP_CHECK(hipStreamSynchronize(s));
        if (g_trk_times.on) tt2 = TrkTimes::now();
        resolve_pending(false);                // the previous frame's box read-back was queued ahead of this sync
        std::copy(h_cost.p, h_cost.p + tn, last_app.begin());
        std::copy(h_cost.p + tn, h_cost.p + 2 * tn, last_maha.begin());
        std::copy(h_cost.p + 2 * tn, h_cost.p + 3 * tn, last_iou.begin());
    }

    resolve_pending(true);                     // (no-op when the sync above already resolved it)
    const double tt2a = g_trk_times.on ? TrkTimes::now() : 0.0;

    // ---- host: association
    std::vector<std::pair<int, int>> matches;
    std::vector<int> un_t, un_d;
    match(T, n, last_app.data(), last_maha.data(), last_iou.data(), matches, un_t, un_d);
    if (g_trk_times.on) { const double tt2b = TrkTimes::now(); g_trk_times.copy += tt2a - tt2, g_trk_times.cascade += tt2b - tt2a; }
    last_matches.clear();
    for (auto& m : matches) last_matches.emplace_back(tracks[m.first].id, m.second);

    // ---- host: lifecycle + commit lists
    const int M = (int)matches.size(), U = (int)un_d.size();
    AIC_REQUIRE((int)free_slots.size() >= U, AIC_ERR_CAPACITY, "track capacity exhausted (raise max_tracks)");
    std::vector<int> upd_slot(M), upd_det(M), ini_slot(U), ini_det(U);
    std::vector<int> ap_slot, ap_pos, ap_det;
    auto push_feature = [&](TrackRec& t, int det) {   // track.py:70-74 as a ring
        if (!hf[det]) return;
        int pos;
        if (t.glen < gmax) {
            pos = (t.ghead + t.glen) % gmax;
            t.glen += 1;
        } else {
            AIC_REQUIRE(!unlimited, AIC_ERR_RUNTIME, "unlimited gallery ring wrapped (grow_galleries runs before the lifecycle)");
            pos = t.ghead;
            t.ghead = (t.ghead + 1) % gmax;
        }
        ap_slot.push_back(t.slot), ap_pos.push_back(pos), ap_det.push_back(det);
        t.c_appos = pos, t.c_apdet = det;
    };
    std::vector<int> match_index_of_track(T, -1);
    for (int k = 0; k < M; ++k) {   // track.py:82-104
        TrackRec& t = tracks[matches[k].first];
        const int det = matches[k].second;
        upd_slot[k] = t.slot, upd_det[k] = det;
        t.c_kind = 1, t.c_det = det, t.c_kout = k;
        match_index_of_track[matches[k].first] = k;
        push_feature(t, det);
        t.hits += 1;
        t.tsu = 0;
        t.conf = conf[det];
        t.cls = cls[det];
        if (t.state == TRK_TENTATIVE && t.hits >= prm.n_init) t.state = TRK_CONFIRMED;
        else if (t.state == TRK_DELETED) t.state = TRK_CONFIRMED;
    }
    for (int i : un_t) {            // track.py:106-119
        TrackRec& t = tracks[i];
        if (t.state == TRK_TENTATIVE) t.state = TRK_DELETED;
        else if (t.state == TRK_CONFIRMED && t.tsu > prm.max_age) t.state = TRK_DELETED;
    }
    for (int k = 0; k < U; ++k) {   // tracker_core.py:180-194, track.py:23-67
        TrackRec t{};
        t.id = next_id++;
        t.state = TRK_TENTATIVE;
        t.hits = 1, t.age = 1, t.tsu = 0;
        t.cls = cls[un_d[k]];
        t.conf = conf[un_d[k]];
        t.slot = free_slots.back();
        free_slots.pop_back();
        t.glen = 0, t.ghead = 0;
        ini_slot[k] = t.slot, ini_det[k] = un_d[k];
        t.c_kind = 2, t.c_det = un_d[k];
        push_feature(t, un_d[k]);
        tracks.push_back(t);
    }

    if (g_trk_times.on) tt3 = TrkTimes::now();
    // ---- device: commit
    const int A = (int)ap_slot.size();
    static const bool step_ok = getenv("AICAM_TRK_NOSTEP") == nullptr;
    const int tl_buf = out_parity;             // the pinned box buffer this frame's outputs will be resolved from
    const bool stepped = nx != nullptr && defer_outputs && step_ok && getenv("AICAM_TRK_SPLIT") == nullptr && getenv("AICAM_TRK_COPY") == nullptr;
    if (stepped) {                              // pipelined form: the commit rides in front of the next frame's association (below, after the prune)
        d_tlwh.ensure((size_t)std::max(M, 1) * 4);
        h_tlwh2[out_parity].ensure((size_t)std::max(M, 1) * 4);
    } else if (M + U + A > 0) {
        const size_t words = (size_t)2 * M + 2 * U + 3 * A;
        const size_t xy_off = ((words * 4 + 15) / 16) * 16;
        const size_t bytes = xy_off + (size_t)n * 16;
        PinBuf<char>& hst = defer_outputs ? h_stage2 : h_stage;
        DevBuf<char>& dst = defer_outputs ? d_stage2 : d_stage;
        hst.ensure(bytes);
        dst.ensure(bytes);
        int* hs = reinterpret_cast<int*>(hst.p);
        int* p = hs;
        std::copy(upd_slot.begin(), upd_slot.end(), p); p += M;
        std::copy(upd_det.begin(), upd_det.end(), p); p += M;
        std::copy(ini_slot.begin(), ini_slot.end(), p); p += U;
        std::copy(ini_det.begin(), ini_det.end(), p); p += U;
        std::copy(ap_slot.begin(), ap_slot.end(), p); p += A;
        std::copy(ap_pos.begin(), ap_pos.end(), p); p += A;
        std::copy(ap_det.begin(), ap_det.end(), p); p += A;
        std::memcpy(hst.p + xy_off, xyah.data(), (size_t)n * 16);
        static const bool zero_copy2 = getenv("AICAM_TRK_COPY") == nullptr;
        const char* pbase2 = hst.p;
        if (!zero_copy2) {
            HIP_CHECK(hipMemcpyAsync(dst.p, hst.p, bytes, hipMemcpyHostToDevice, s));
            pbase2 = dst.p;
        }
        const int* d = reinterpret_cast<const int*>(pbase2);
        const float* d_xy = reinterpret_cast<const float*>(pbase2 + xy_off);
        d_tlwh.ensure((size_t)std::max(M, 1) * 4);
        PinBuf<float>& htl = defer_outputs ? h_tlwh2[out_parity] : h_tlwh;
        htl.ensure((size_t)std::max(M, 1) * 4);
        {
            Prof pr(*dev, PROF_TRK, s, 0, (double)(M + U) * 72 * 4 * 2 + (double)A * dim * 8);
            launch_trk_commit(d_mean.p, d_cov.p, d, M, U, A, d_xy, zero_copy2 ? htl.p : d_tlwh.p, d_gal_raw.p, d_gal_n.p, gmax, dim,
                              d_featp, d_featn, s);   // the boxes go straight to pinned host memory
        }
        if (M && !zero_copy2) HIP_CHECK(hipMemcpyAsync(htl.p, d_tlwh.p, (size_t)M * 16, hipMemcpyDeviceToHost, s));
        if (!defer_outputs) HIP_CHECK(hipStreamSynchronize(s));
    }

    // ---- outputs (deepsort_tracker.py:126-141), before pruning: indices still refer to `tracks`
    std::vector<OutMeta> meta;
    for (int i = 0; i < T; ++i) {
        const TrackRec& t = tracks[i];
        if (t.state != TRK_CONFIRMED || t.tsu != 0) continue;
        const int k = match_index_of_track[i];
        if (k < 0) continue;
        meta.push_back(OutMeta{k, t.id, t.cls, t.conf});
    }
    auto build = [](const std::vector<OutMeta>& mt, const float* tl, std::vector<TrackOut>& out) {
        out.clear();
        for (const OutMeta& m : mt) {
            const float* b = tl + (size_t)m.k * 4;
            const float x1 = b[0], y1 = b[1];
            const float w = b[2] > 0.f ? b[2] : 0.f, h = b[3] > 0.f ? b[3] : 0.f;
            TrackOut o;
            o.x1 = round_half_even(x1), o.y1 = round_half_even(y1);
            o.x2 = round_half_even(x1 + w), o.y2 = round_half_even(y1 + h);
            o.id = m.id, o.cls = m.cls, o.conf = m.conf;
            out.push_back(o);
        }
    };
    if (defer_outputs) {
        pend.active = true, pend.buf = out_parity, pend.meta.swap(meta);
        out_parity ^= 1;
        outputs.clear();
    } else {
        build(meta, h_tlwh.p, outputs);
    }
    if (g_trk_times.on) {
        const double tt4 = TrkTimes::now();
        g_trk_times.prep += tt1 - tt0, g_trk_times.gpu += tt2 - tt1, g_trk_times.host += tt3 - tt2, g_trk_times.commit += tt4 - tt3, g_trk_times.n += 1;
    }
    // ---- prune (tracker_core.py:75)
    std::vector<TrackRec> alive;
    alive.reserve(tracks.size());
    for (auto& t : tracks) {
        if (t.state == TRK_DELETED) free_slots.push_back(t.slot);
        else alive.push_back(t);
    }
    tracks.swap(alive);

    if (stepped) {
        // ---- ONE launch: per surviving track, this frame's commit (update / initiate / gallery row), then -- if the next
        // frame has detections -- its lazy predict and association rows, straight into pinned host memory
        const int T2 = (int)tracks.size(), n1 = nx->n;
        if (T2 > 0) {
            const bool feat1 = nx->feat_n != nullptr && n1 > 0 && dim > 0;
            const size_t off_f = (size_t)7 * T2 * 4;
            const size_t off_xy0 = ((off_f + 15) / 16) * 16, off_tl1 = off_xy0 + (size_t)n * 16, off_xy1 = off_tl1 + (size_t)n1 * 16;
            const size_t off_has = off_xy1 + (size_t)n1 * 16, bytes = ((off_has + n1 + 15) / 16) * 16;
            h_step.ensure(bytes + 16);
            int* hi = reinterpret_cast<int*>(h_step.p);
            for (int i = 0; i < T2; ++i) {
                const TrackRec& tr = tracks[i];
                hi[i] = tr.slot, hi[T2 + i] = tr.glen, hi[2 * T2 + i] = tr.c_kind, hi[3 * T2 + i] = tr.c_det, hi[4 * T2 + i] = tr.c_kout;
                hi[5 * T2 + i] = tr.c_appos, hi[6 * T2 + i] = tr.c_apdet;
            }
            std::memcpy(h_step.p + off_xy0, xyah.data(), (size_t)n * 16);
            float* tl1 = reinterpret_cast<float*>(h_step.p + off_tl1);
            float* xy1 = reinterpret_cast<float*>(h_step.p + off_xy1);
            for (int j = 0; j < n1; ++j) {   // detection.py:36-47
                const float x = nx->tlwh[j * 4], y = nx->tlwh[j * 4 + 1], w = nx->tlwh[j * 4 + 2], h = nx->tlwh[j * 4 + 3];
                tl1[j * 4] = x, tl1[j * 4 + 1] = y, tl1[j * 4 + 2] = w, tl1[j * 4 + 3] = h;
                xy1[j * 4] = x + w / 2.0f, xy1[j * 4 + 1] = y + h / 2.0f, xy1[j * 4 + 2] = h > 0.f ? w / h : 0.f, xy1[j * 4 + 3] = h;
            }
            for (int j = 0; j < n1; ++j) h_step.p[off_has + j] = (feat1 && (!nx->has || nx->has[j])) ? 1 : 0;
            const size_t tn1 = (size_t)T2 * n1;
            h_cost.ensure(3 * std::max<size_t>(tn1, 1));
            Prof pr(*dev, PROF_TRK, s, feat1 ? 2.0 * T2 * gmax * (double)n1 * dim : 0.0, feat1 ? ((double)T2 * gmax + n1) * dim * 4 : 0.0);
            launch_trk_step(d_mean.p, d_cov.p, hi, hi + T2, T2, 1, tl1, xy1, feat1 ? d_gal_n.p : nullptr, gmax, dim, nx->feat_n,
                            reinterpret_cast<const unsigned char*>(h_step.p + off_has), n1, h_cost.p, h_cost.p + tn1, h_cost.p + 2 * tn1,
                            hi + 2 * T2, hi + 3 * T2, hi + 4 * T2, hi + 5 * T2, hi + 6 * T2,
                            reinterpret_cast<const float*>(h_step.p + off_xy0), d_featp, d_featn, h_tlwh2[tl_buf].p, d_gal_raw.p, d_gal_n.p, s);
            if (n1 > 0) pre_rows = true, pre_predicted = true, pre_T = T2, pre_n = n1;
        }
    }
}


// ------------------------------------------------------------------------------------------------ association on the device
void Tracker::to_device() {
    if (on_device) return;
    AIC_REQUIRE(dev_capable(), AIC_ERR_INVALID, "device association needs nn_budget > 0, max_tracks <= 512 and a feature dimension divisible by 4");
    AIC_REQUIRE(!pre_rows && !pend.active, AIC_ERR_RUNTIME, "device association: a pipelined host step is still open");
    dev->use();
    flush_predict();
    hipStream_t s = dev->s_trk;
    d_tbl.ensure(tbl_bytes());
    h_tbl.ensure(tbl_bytes());
    HIP_CHECK(hipStreamSynchronize(s));                    // h_tbl may still be the target of the previous header copy
    DevTrkHdr* hh = reinterpret_cast<DevTrkHdr*>(h_tbl.p);
    DevTrack* ht = reinterpret_cast<DevTrack*>(h_tbl.p + sizeof(DevTrkHdr));
    int* hf = reinterpret_cast<int*>(h_tbl.p + sizeof(DevTrkHdr) + sizeof(DevTrack) * (size_t)cap);
    std::memset(hh, 0, sizeof(DevTrkHdr));
    hh->n_tracks = (int)tracks.size(), hh->next_id = next_id, hh->n_free = (int)free_slots.size();
    for (size_t i = 0; i < tracks.size(); ++i) {
        const TrackRec& r = tracks[i];
        DevTrack& t = ht[i];
        t.id = r.id, t.state = r.state, t.hits = r.hits, t.age = r.age, t.tsu = r.tsu, t.cls = r.cls, t.conf = r.conf;
        t.slot = r.slot, t.glen = r.glen, t.ghead = r.ghead, t.pad[0] = t.pad[1] = 0;
    }
    std::copy(free_slots.begin(), free_slots.end(), hf);
    HIP_CHECK(hipMemcpyAsync(d_tbl.p, h_tbl.p, tbl_bytes(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    on_device = true;
}

void Tracker::to_host() {
    if (!on_device) return;
    dev->use();
    hipStream_t s = dev->s_trk;
    HIP_CHECK(hipMemcpyAsync(h_tbl.p, d_tbl.p, tbl_bytes(), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    const DevTrkHdr* hh = reinterpret_cast<const DevTrkHdr*>(h_tbl.p);
    const DevTrack* ht = reinterpret_cast<const DevTrack*>(h_tbl.p + sizeof(DevTrkHdr));
    const int* hf = reinterpret_cast<const int*>(h_tbl.p + sizeof(DevTrkHdr) + sizeof(DevTrack) * (size_t)cap);
    tracks.clear();
    for (int i = 0; i < hh->n_tracks; ++i) {
        TrackRec r{};
        r.id = ht[i].id, r.state = ht[i].state, r.hits = ht[i].hits, r.age = ht[i].age, r.tsu = ht[i].tsu, r.cls = ht[i].cls;
        r.conf = ht[i].conf, r.slot = ht[i].slot, r.glen = ht[i].glen, r.ghead = ht[i].ghead;
        tracks.push_back(r);
    }
    free_slots.assign(hf, hf + hh->n_free);
    next_id = hh->next_id;
    acc_fast += hh->n_fast, acc_lsap += hh->n_lsap;       // the header's counters restart when the table next goes up
    on_device = false;
    for (int i = 0; i < dev_predicts; ++i)                // predicts announced but not yet consumed by an epoch
        for (auto& t : tracks) { t.age += 1; t.tsu += 1; }
    if (dev_predicts > 0 && !tracks.empty()) {
        for (int i = 0; i + 1 < dev_predicts; ++i) { pending_predict = true; flush_predict(); }
        pending_predict = true;
    }
    dev_predicts = 0;
}

void Tracker::run_epochs(const EpochDets& dets, const int* h_n, const int* h_d0, int frames, const EpochOut& out, hipStream_t s, bool debug) {
    to_device();
    const bool feats = dets.feat_n != nullptr && dim > 0;
    TrkDevParams prm;
    prm.max_cos = (float)this->prm.max_cosine_distance, prm.clamp_cos = (float)(this->prm.max_cosine_distance + 1e-5);   // lsap.cpp:133-134
    prm.max_iou = (float)this->prm.max_iou_distance, prm.clamp_iou = (float)(this->prm.max_iou_distance + 1e-5);
    prm.max_age = this->prm.max_age, prm.n_init = this->prm.n_init, prm.gmax = gmax, prm.dim = dim, prm.cap = cap;
    prm.no_fast = lsap_fast ? 0 : 1;
    prm.no_wave = wave_cascade ? 0 : 1;
    d_costs.ensure((size_t)3 * TRK_DEV_TMAX * TRK_DEV_NMAX);
    d_sub.ensure((size_t)TRK_DEV_TMAX * TRK_DEV_NMAX);
    d_appends.ensure((size_t)3 * TRK_DEV_DNMAX + 4);          // + the list's length (gallery_commit_kernel)
    d_dbg.ensure(16 + 2 * TRK_DEV_TMAX);
    static const int k_env = [] { const char* e = getenv("AICAM_TRK_K"); return e ? atoi(e) : 16; }();   // frames per epoch (measured: 8 / 16 / 32)
    const int kmax = std::max(1, std::min(std::min(TRK_KMAX, epoch_frames > 0 ? epoch_frames : k_env), gmax));
    int f = 0;
    while (f < frames) {
        int k = 0, dn = 0, nmax = 0;
        while (f + k < frames && k < kmax && dn + h_n[f + k] <= TRK_DEV_DNMAX) {
            dn += h_n[f + k];
            nmax = std::max(nmax, h_n[f + k]);
            ++k;
        }
        AIC_REQUIRE(k > 0 && nmax <= TRK_DEV_NMAX, AIC_ERR_CAPACITY, "device association: more than 512 detections in one frame");
        const int dn_pad = std::max(32, (dn + 31) / 32 * 32);
        const int d_begin = h_d0[f];
        const bool has_sm = feats && dn > 0;
        EpochScratch scr{nullptr, nullptr, d_costs.p, d_sub.p, d_appends.p};
        if (has_sm) {
            d_sm.ensure((size_t)cap * (TRK_KMAX + 1) * dn_pad);
            d_gram.ensure((size_t)dn_pad * dn_pad);
            scr.sm = d_sm.p, scr.gram = d_gram.p;
            Prof pr(*dev, PROF_TRK, s, 2.0 * ((double)cap * gmax + dn) * dn * dim, 0);
            static const int rep = [] { const char* e = getenv("AICAM_TRK_PREP_REPEAT"); return e ? std::max(1, atoi(e)) : 1; }();   // measurement: the prep launch's share of the interference
            for (int r = 0; r < rep; ++r)
                launch_trk_epoch_prep(tbl_hdr(), tbl_trk(), d_gal_n.p, gmax, dim, cap, dets.feat_n + (size_t)d_begin * dim, dn, dn_pad, k, d_sm.p, d_gram.p, s);
        }
        EpochOut o = out;
        const bool last = f + k >= frames;
        if (!(debug && last)) { o.dbg_tn = nullptr; if (!o.dbg_stride) o.dbg_match = nullptr; }
        {
            Prof pr(*dev, PROF_TRK, s, 0, 0);
            launch_trk_epoch(tbl_hdr(), tbl_trk(), tbl_free(), d_mean.p, d_cov.p, d_gal_raw.p, d_gal_n.p, prm, dets, f, k, d_begin, dn_pad, nmax,
                             has_sm ? 1 : 0, scr, o, s);
        }
        f += k;
    }
    HIP_CHECK(hipMemcpyAsync(h_tbl.p, d_tbl.p, sizeof(DevTrkHdr), hipMemcpyDeviceToHost, s));
}

void Tracker::check_epochs() {
    const DevTrkHdr* hh = reinterpret_cast<const DevTrkHdr*>(h_tbl.p);
    n_fast = acc_fast + hh->n_fast, n_lsap = acc_lsap + hh->n_lsap;
    if (hh->err == 0) return;
    const std::string at = " (frame " + std::to_string(hh->err_frame) + " of the launch group; the tracker state is the frame before it)";
    AIC_REQUIRE(hh->err != 1, AIC_ERR_CAPACITY, "track capacity exhausted (raise max_tracks)" + at);
    AIC_REQUIRE(hh->err != 3, AIC_ERR_CAPACITY, "device association: frame beyond the epoch kernel's capacity" + at);
    AIC_REQUIRE(false, AIC_ERR_RUNTIME, "device association: the assignment problem has no finite solution" + at);
}

// aic_tracker_update through the device path: one frame = one epoch of length 1.
void Tracker::update_device(const float* det_tlwh, const float* conf, const int32_t* cls, const float* feat, int feat_mem,
                            const uint8_t* has_feat, int n, int dim_in) {
    AIC_REQUIRE(n >= 0 && n <= TRK_DEV_NMAX, AIC_ERR_CAPACITY, "device association: more than 512 detections in one frame");
    hipStream_t s = dev->s_trk;
    const bool any_feat = feat != nullptr && n > 0;
    const int max_rows = TRK_DEV_TMAX;
    // staging layout (host == device): n | d0 | tlwh[n*4] | conf[n] | cls[n] | valid[n] || out: n_tracks | rows[max_rows*6] | conf[max_rows]
    const size_t o_tlwh = 16, o_conf = o_tlwh + (size_t)n * 16, o_cls = o_conf + (size_t)n * 4, o_valid = o_cls + (size_t)n * 4;
    const size_t o_out = ((o_valid + (size_t)n * 4 + 15) / 16) * 16, o_rows = o_out + 16, o_oconf = o_rows + (size_t)max_rows * 24;
    const size_t bytes = o_oconf + (size_t)max_rows * 4;
    HIP_CHECK(hipStreamSynchronize(s));
    h_api.ensure(bytes);
    d_api.ensure(bytes);
    int* hi = reinterpret_cast<int*>(h_api.p);
    hi[0] = n, hi[1] = 0;
    if (n) {
        std::memcpy(h_api.p + o_tlwh, det_tlwh, (size_t)n * 16);
        std::memcpy(h_api.p + o_conf, conf, (size_t)n * 4);
        std::memcpy(h_api.p + o_cls, cls, (size_t)n * 4);
        int* hv = reinterpret_cast<int*>(h_api.p + o_valid);
        for (int j = 0; j < n; ++j) hv[j] = (any_feat && (!has_feat || has_feat[j])) ? 1 : 0;
    }
    HIP_CHECK(hipMemcpyAsync(d_api.p, h_api.p, o_out, hipMemcpyHostToDevice, s));
    const float* d_featp = nullptr;
    const float* d_featn = nullptr;
    if (any_feat) {
        if (feat_mem == AIC_DEVICE) d_featp = feat;
        else {
            d_feat.ensure((size_t)n * dim);
            HIP_CHECK(hipMemcpyAsync(d_feat.p, feat, (size_t)n * dim * 4, hipMemcpyHostToDevice, s));
            d_featp = d_feat.p;
        }
        d_detn.ensure((size_t)n * dim);
        launch_normalize_rows(d_featp, d_detn.p, n, dim, s);
        d_featn = d_detn.p;
    }
    EpochDets dets{reinterpret_cast<const int*>(d_api.p), reinterpret_cast<const int*>(d_api.p + 4),
                   reinterpret_cast<const float*>(d_api.p + o_tlwh), reinterpret_cast<const float*>(d_api.p + o_conf),
                   reinterpret_cast<const int*>(d_api.p + o_cls), reinterpret_cast<const int*>(d_api.p + o_valid), d_featp, d_featn};
    d_dbg.ensure(16 + 2 * TRK_DEV_TMAX);
    EpochOut out{reinterpret_cast<int*>(d_api.p + o_out), reinterpret_cast<int*>(d_api.p + o_rows), reinterpret_cast<float*>(d_api.p + o_oconf),
                 max_rows, d_dbg.p + 8, d_dbg.p, 0};
    const int zero = 0;
    run_epochs(dets, &n, &zero, 1, out, s, true);
    HIP_CHECK(hipMemcpyAsync(h_api.p + o_out, d_api.p + o_out, bytes - o_out, hipMemcpyDeviceToHost, s));
    std::vector<int> dbg(16 + 2 * TRK_DEV_TMAX);
    HIP_CHECK(hipMemcpyAsync(dbg.data(), d_dbg.p, dbg.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    check_epochs();
    const int no = *reinterpret_cast<const int*>(h_api.p + o_out);
    const int* rows = reinterpret_cast<const int*>(h_api.p + o_rows);
    const float* oc = reinterpret_cast<const float*>(h_api.p + o_oconf);
    outputs.clear();
    for (int k2 = 0; k2 < no && k2 < max_rows; ++k2) {
        const int* r = rows + (size_t)k2 * 6;
        outputs.push_back(TrackOut{r[0], r[1], r[2], r[3], r[4], r[5], oc[k2]});
    }
    last_t = dbg[0], last_n = dbg[1];
    last_matches.clear();
    for (int m = 0; m < dbg[8]; ++m) last_matches.emplace_back(dbg[9 + 2 * m], dbg[10 + 2 * m]);
    const size_t tn = (size_t)last_t * last_n;
    last_app.assign(tn, kInfty), last_maha.assign(tn, 0.f), last_iou.assign(tn, kInfty);
    if (tn) {
        const size_t stride = (size_t)TRK_DEV_TMAX * TRK_DEV_NMAX;
        HIP_CHECK(hipMemcpyAsync(last_app.data(), d_costs.p, tn * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(last_maha.data(), d_costs.p + stride, tn * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(last_iou.data(), d_costs.p + 2 * stride, tn * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
}

// aic_tracker_update_batch: k frames, each TrackerCore.predict() + update() (tracker_core.py:44-81), as epochs of the device
// association -- what the pipeline does per launch group, with the embeddings supplied by the caller.
void Tracker::update_batch(int k, const int32_t* counts, const float* det_tlwh, const float* conf, const int32_t* cls, const float* feat,
                           int feat_mem, const uint8_t* has_feat, int dim_in, int cap_rows, int32_t* n_out, int32_t* out6, float* out_conf,
                           int32_t* n_match, int32_t* match_tid, int32_t* match_det) {
    dev->use();
    AIC_REQUIRE(k >= 0 && cap_rows >= 0, AIC_ERR_INVALID, "negative frame count / row capacity");
    if (k == 0) return;
    AIC_REQUIRE(dev_predicts == 0 && !pending_predict, AIC_ERR_INVALID, "update_batch: a predict() without its update() is still open");
    long total = 0;
    for (int f = 0; f < k; ++f) {
        AIC_REQUIRE(counts[f] >= 0 && counts[f] <= TRK_DEV_NMAX, AIC_ERR_CAPACITY, "device association: more than 512 detections in one frame");
        total += counts[f];
    }
    const bool any_feat = feat != nullptr && total > 0;
    if (any_feat) ensure_dim(dim_in);
    AIC_REQUIRE(dev_capable(), AIC_ERR_INVALID, "update_batch runs on the device: needs nn_budget > 0, max_tracks <= 512 and a feature dimension divisible by 4");
    hipStream_t s = dev->s_trk;
    const int n = (int)total, stride = 2 * TRK_DEV_TMAX + 8;
    // staging (host == device layout): frame_n[k] | frame_d0[k] | tlwh[n*4] | conf[n] | cls[n] | valid[n] || n_tracks[k] | rows[k*cap*6] | conf[k*cap] | matches[k*stride]
    const size_t o_d0 = (size_t)k * 4, o_tlwh = (((size_t)k * 8 + 15) / 16) * 16, o_conf = o_tlwh + (size_t)n * 16, o_cls = o_conf + (size_t)n * 4;
    const size_t o_valid = o_cls + (size_t)n * 4, o_out = ((o_valid + (size_t)n * 4 + 15) / 16) * 16;
    const size_t o_rows = o_out + (((size_t)k * 4 + 15) / 16) * 16, o_oconf = o_rows + (size_t)k * cap_rows * 24;
    const size_t o_dbg = ((o_oconf + (size_t)k * cap_rows * 4 + 15) / 16) * 16, bytes = o_dbg + (size_t)k * stride * 4;
    HIP_CHECK(hipStreamSynchronize(s));
    h_api.ensure(bytes);
    d_api.ensure(bytes);
    int* hn = reinterpret_cast<int*>(h_api.p);
    int* hd = reinterpret_cast<int*>(h_api.p + o_d0);
    int d0 = 0;
    for (int f = 0; f < k; ++f) { hn[f] = counts[f]; hd[f] = d0; d0 += counts[f]; }
    if (n) {
        std::memcpy(h_api.p + o_tlwh, det_tlwh, (size_t)n * 16);
        std::memcpy(h_api.p + o_conf, conf, (size_t)n * 4);
        std::memcpy(h_api.p + o_cls, cls, (size_t)n * 4);
        int* hv = reinterpret_cast<int*>(h_api.p + o_valid);
        for (int j = 0; j < n; ++j) hv[j] = (any_feat && (!has_feat || has_feat[j])) ? 1 : 0;
    }
    HIP_CHECK(hipMemcpyAsync(d_api.p, h_api.p, o_out, hipMemcpyHostToDevice, s));
    const float* d_featp = nullptr;
    const float* d_featn = nullptr;
    if (any_feat) {
        if (feat_mem == AIC_DEVICE) d_featp = feat;
        else {
            d_feat.ensure((size_t)n * dim);
            HIP_CHECK(hipMemcpyAsync(d_feat.p, feat, (size_t)n * dim * 4, hipMemcpyHostToDevice, s));
            d_featp = d_feat.p;
        }
        d_detn.ensure((size_t)n * dim);
        launch_normalize_rows(d_featp, d_detn.p, n, dim, s);
        d_featn = d_detn.p;
    }
    EpochDets dets{reinterpret_cast<const int*>(d_api.p), reinterpret_cast<const int*>(d_api.p + o_d0),
                   reinterpret_cast<const float*>(d_api.p + o_tlwh), reinterpret_cast<const float*>(d_api.p + o_conf),
                   reinterpret_cast<const int*>(d_api.p + o_cls), reinterpret_cast<const int*>(d_api.p + o_valid), d_featp, d_featn};
    EpochOut out{reinterpret_cast<int*>(d_api.p + o_out), reinterpret_cast<int*>(d_api.p + o_rows), reinterpret_cast<float*>(d_api.p + o_oconf),
                 cap_rows, reinterpret_cast<int*>(d_api.p + o_dbg), nullptr, stride};
    run_epochs(dets, hn, hd, k, out, s, false);
    HIP_CHECK(hipMemcpyAsync(h_api.p + o_out, d_api.p + o_out, bytes - o_out, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    check_epochs();
    const int* on = reinterpret_cast<const int*>(h_api.p + o_out);
    const int* rows = reinterpret_cast<const int*>(h_api.p + o_rows);
    const float* oc = reinterpret_cast<const float*>(h_api.p + o_oconf);
    const int* dbg = reinterpret_cast<const int*>(h_api.p + o_dbg);
    for (int f = 0; f < k; ++f) {
        const int kk = std::min(on[f], cap_rows);
        if (n_out) n_out[f] = on[f];                       // the true count: rows beyond cap_rows are not stored
        if (out6) std::copy(rows + (size_t)f * cap_rows * 6, rows + ((size_t)f * cap_rows + kk) * 6, out6 + (size_t)f * cap_rows * 6);
        if (out_conf) std::copy(oc + (size_t)f * cap_rows, oc + (size_t)f * cap_rows + kk, out_conf + (size_t)f * cap_rows);
        const int* dm = dbg + (size_t)f * stride;
        if (n_match) n_match[f] = dm[0];
        for (int m = 0; m < dm[0] && m < cap_rows; ++m) {
            if (match_tid) match_tid[(size_t)f * cap_rows + m] = dm[1 + 2 * m];
            if (match_det) match_det[(size_t)f * cap_rows + m] = dm[2 + 2 * m];
        }
    }
    outputs.clear();
    const int kl = std::min(on[k - 1], cap_rows);
    for (int r = 0; r < kl; ++r) {
        const int* q = rows + ((size_t)(k - 1) * cap_rows + r) * 6;
        outputs.push_back(TrackOut{q[0], q[1], q[2], q[3], q[4], q[5], oc[(size_t)(k - 1) * cap_rows + r]});
    }
}

// aic_tracker_import_state: the inverse of aic_tracker_export + aic_tracker_export_gallery.
void Tracker::import_state(int n, const int32_t* track_id, const int32_t* state, const int32_t* hits, const int32_t* age, const int32_t* tsu,
                           const int32_t* cls, const float* conf, const int32_t* gallery_len, const float* mean, const float* cov,
                           const float* galleries, int dim_in, int next_track_id) {
    dev->use();
    AIC_REQUIRE(n >= 0 && n <= cap, AIC_ERR_CAPACITY, "more tracks than the tracker's capacity (max_tracks)");
    AIC_REQUIRE(n == 0 || (track_id && state && hits && age && tsu && cls && conf && gallery_len && mean && cov), AIC_ERR_INVALID, "NULL state array");
    long rows = 0;
    int longest = 0;
    for (int i = 0; i < n; ++i) {
        AIC_REQUIRE(state[i] == TRK_TENTATIVE || state[i] == TRK_CONFIRMED, AIC_ERR_INVALID, "track state must be Tentative or Confirmed (deleted tracks are pruned, tracker_core.py:75)");
        AIC_REQUIRE(gallery_len[i] >= 0, AIC_ERR_INVALID, "negative gallery length");
        AIC_REQUIRE(track_id[i] < next_track_id, AIC_ERR_INVALID, "next_track_id must be above every imported id (track.py:21)");
        rows += gallery_len[i];
        longest = std::max(longest, gallery_len[i]);
    }
    AIC_REQUIRE(rows == 0 || (galleries && dim_in > 0), AIC_ERR_INVALID, "galleries need their rows and a feature dimension");
    if (rows) ensure_dim(dim_in);
    if (!unlimited) AIC_REQUIRE(longest <= gmax, AIC_ERR_CAPACITY, "a gallery is longer than nn_budget");
    while (unlimited && dim > 0 && longest + 1 > gmax) grow_galleries();
    hipStream_t s = dev->s_trk;
    HIP_CHECK(hipStreamSynchronize(s));
    // every check passed: from here on the old state is gone
    on_device = false, dev_predicts = 0, pending_predict = false, pre_rows = false, pre_predicted = false;
    pend.active = false;
    tracks.clear();
    free_slots.clear();
    for (int sl = cap - 1; sl >= n; --sl) free_slots.push_back(sl);     // the constructor's order: the lowest free slot is handed out first
    next_id = next_track_id;
    DevBuf<float> stage;
    if (rows) {
        stage.alloc((size_t)rows * dim);
        HIP_CHECK(hipMemcpyAsync(stage.p, galleries, (size_t)rows * dim * 4, hipMemcpyHostToDevice, s));
    }
    size_t r0 = 0;
    for (int i = 0; i < n; ++i) {
        TrackRec r{};
        r.id = track_id[i], r.state = state[i], r.hits = hits[i], r.age = age[i], r.tsu = tsu[i], r.cls = cls[i], r.conf = conf[i];
        r.slot = i, r.glen = gallery_len[i], r.ghead = 0;
        tracks.push_back(r);
        if (r.glen) {
            float* raw = d_gal_raw.p + (size_t)r.slot * gmax * dim;
            HIP_CHECK(hipMemcpyAsync(raw, stage.p + r0 * dim, (size_t)r.glen * dim * 4, hipMemcpyDeviceToDevice, s));
            launch_normalize_rows(raw, d_gal_n.p + (size_t)r.slot * gmax * dim, r.glen, dim, s);    // matching.py:126-130, as at append time
            r0 += r.glen;
        }
    }
    if (n) {
        HIP_CHECK(hipMemcpyAsync(d_mean.p, mean, (size_t)n * 8 * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_cov.p, cov, (size_t)n * 64 * 4, hipMemcpyHostToDevice, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    outputs.clear(), resolved.clear(), last_matches.clear();
    last_t = last_n = 0;
}

}  // namespace aic

// =================================================================================================
using namespace aic;


namespace {

// Run `fn(d_in..., stream)` on temporary device copies of host arrays (API-level KF / cost calls).
struct Tmp {
    Device& d;
    std::vector<DevBuf<char>> bufs;
    explicit Tmp(Device& dev) : d(dev) {}
    template <class T> T* up(const T* host, size_t count) {
        bufs.emplace_back(count * sizeof(T) + 16);
        if (count) HIP_CHECK(hipMemcpyAsync(bufs.back().p, host, count * sizeof(T), hipMemcpyHostToDevice, d.s_trk));
        return reinterpret_cast<T*>(bufs.back().p);
    }
    template <class T> T* raw(size_t count) {
        bufs.emplace_back(count * sizeof(T) + 16);
        return reinterpret_cast<T*>(bufs.back().p);
    }
    template <class T> void down(T* host, const T* dev, size_t count) {
        if (count) HIP_CHECK(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, d.s_trk));
    }
    void sync() { HIP_CHECK(hipStreamSynchronize(d.s_trk)); }
};

}  // namespace

extern "C" {

int aic_kf_initiate(int device_id, const float* z, int n, float* mean, float* cov) {
    return guarded([&] {
        AIC_REQUIRE(n >= 0 && (n == 0 || (z && mean && cov)), AIC_ERR_INVALID, "bad argument");
        Device& d = device(device_id);
        Tmp t(d);
        float* dz = t.up(z, (size_t)n * 4);
        float* dm = t.raw<float>((size_t)n * 8);
        float* dc = t.raw<float>((size_t)n * 64);
        launch_kf_initiate(dz, n, dm, dc, nullptr, d.s_trk);
        t.down(mean, dm, (size_t)n * 8);
        t.down(cov, dc, (size_t)n * 64);
        t.sync();
    });
}

int aic_kf_predict_dt(int device_id, float* mean, float* cov, int n, float dt) {
    return guarded([&] {
        AIC_REQUIRE(n >= 0 && (n == 0 || (mean && cov)), AIC_ERR_INVALID, "bad argument");
        AIC_REQUIRE(dt == dt && dt - dt == 0.f, AIC_ERR_INVALID, "dt must be finite");
        Device& d = device(device_id);
        Tmp t(d);
        float* dm = t.up(mean, (size_t)n * 8);
        float* dc = t.up(cov, (size_t)n * 64);
        launch_kf_predict(dm, dc, nullptr, n, d.s_trk, dt);
        t.down(mean, dm, (size_t)n * 8);
        t.down(cov, dc, (size_t)n * 64);
        t.sync();
    });
}

int aic_kf_predict(int device_id, float* mean, float* cov, int n) { return aic_kf_predict_dt(device_id, mean, cov, n, 1.f); }

int aic_kf_project(int device_id, const float* mean, const float* cov, int n, float* pmean, float* pcov) {
    return guarded([&] {
        AIC_REQUIRE(n >= 0 && (n == 0 || (mean && cov && pmean && pcov)), AIC_ERR_INVALID, "bad argument");
        Device& d = device(device_id);
        Tmp t(d);
        float* dm = t.up(mean, (size_t)n * 8);
        float* dc = t.up(cov, (size_t)n * 64);
        float* pm = t.raw<float>((size_t)n * 4);
        float* pc = t.raw<float>((size_t)n * 16);
        launch_kf_project(dm, dc, n, pm, pc, d.s_trk);
        t.down(pmean, pm, (size_t)n * 4);
        t.down(pcov, pc, (size_t)n * 16);
        t.sync();
    });
}

int aic_kf_update(int device_id, float* mean, float* cov, const float* z, int n) {
    return guarded([&] {
        AIC_REQUIRE(n >= 0 && (n == 0 || (mean && cov && z)), AIC_ERR_INVALID, "bad argument");
        Device& d = device(device_id);
        Tmp t(d);
        float* dm = t.up(mean, (size_t)n * 8);
        float* dc = t.up(cov, (size_t)n * 64);
        float* dz = t.up(z, (size_t)n * 4);
        launch_kf_update(dm, dc, nullptr, dz, nullptr, n, nullptr, d.s_trk);
        t.down(mean, dm, (size_t)n * 8);
        t.down(cov, dc, (size_t)n * 64);
        t.sync();
    });
}

int aic_kf_gating(int device_id, const float* mean, const float* cov, int n, const float* zs, int m, int shared_z,
                  int only_position, float* d2) {
    return guarded([&] {
        AIC_REQUIRE(n >= 0 && m >= 0, AIC_ERR_INVALID, "bad argument");
        if (!n || !m) return;
        AIC_REQUIRE(mean && cov && zs && d2, AIC_ERR_INVALID, "NULL argument");
        Device& d = device(device_id);
        Tmp t(d);
        float* dm = t.up(mean, (size_t)n * 8);
        float* dc = t.up(cov, (size_t)n * 64);
        float* dz = t.up(zs, (size_t)(shared_z ? m : (size_t)n * m) * 4);
        float* dd = t.raw<float>((size_t)n * m);
        launch_kf_gating(dm, dc, nullptr, n, dz, m, shared_z, only_position, dd, d.s_trk);
        t.down(d2, dd, (size_t)n * m);
        t.sync();
    });
}

int aic_iou_cost(int device_id, const float* track_tlwh, int tn, const float* det_tlwh, int n, float* cost) {
    return guarded([&] {
        AIC_REQUIRE(tn >= 0 && n >= 0, AIC_ERR_INVALID, "bad argument");
        if (!tn || !n) return;
        AIC_REQUIRE(track_tlwh && det_tlwh && cost, AIC_ERR_INVALID, "NULL argument");
        Device& d = device(device_id);
        Tmp t(d);
        float* a = t.up(track_tlwh, (size_t)tn * 4);
        float* b = t.up(det_tlwh, (size_t)n * 4);
        float* c = t.raw<float>((size_t)tn * n);
        launch_iou_cost(a, nullptr, nullptr, tn, b, n, c, d.s_trk);
        t.down(cost, c, (size_t)tn * n);
        t.sync();
    });
}

int aic_appearance_cost(int device_id, const float* galleries, const int32_t* gallery_len, int tn, int gmax, int dim,
                        const float* det_feat, const uint8_t* has_feat, int n, float* cost) {
    return guarded([&] {
        AIC_REQUIRE(tn >= 0 && n >= 0 && gmax >= 0 && dim > 0, AIC_ERR_INVALID, "bad argument");
        if (!tn || !n) return;
        AIC_REQUIRE(gallery_len && det_feat && cost && (gmax == 0 || galleries), AIC_ERR_INVALID, "NULL argument");
        Device& d = device(device_id);
        Tmp t(d);
        std::vector<int> iota(tn);
        for (int i = 0; i < tn; ++i) iota[i] = i;
        float* g = t.up(galleries, (size_t)tn * gmax * dim);
        int* sl = t.up(iota.data(), tn);
        int* gl = t.up(gallery_len, tn);
        float* f = t.up(det_feat, (size_t)n * dim);
        float* fn = t.raw<float>((size_t)n * dim);
        float* gn = t.raw<float>((size_t)tn * gmax * dim);
        unsigned char* hf = has_feat ? t.up(has_feat, n) : nullptr;
        float* c = t.raw<float>((size_t)tn * n);
        launch_fill(c, kInfty, (size_t)tn * n, d.s_trk);
        launch_normalize_rows(f, fn, n, dim, d.s_trk);
        launch_normalize_rows(g, gn, tn * gmax, dim, d.s_trk);
        launch_cosine_min_mfma(gn, sl, gl, tn, gmax, dim, fn, hf, n, c, d.s_trk);
        t.down(cost, c, (size_t)tn * n);
        t.sync();
    });
}

int aic_tracker_create(int device_id, const aic_tracker_params* p, aic_tracker** out) {
    return guarded([&] {
        AIC_REQUIRE(p && out, AIC_ERR_INVALID, "NULL argument");
        AIC_REQUIRE(p->max_age >= 0 && p->n_init >= 0, AIC_ERR_INVALID, "negative tracker parameter");
        *out = new aic_tracker(device(device_id), *p);
    });
}

int aic_tracker_destroy(aic_tracker* t) {
    return guarded([&] { delete t; });
}

int aic_tracker_option(aic_tracker* t, const char* key, int value) {
    return guarded([&] {
        AIC_REQUIRE(t && key, AIC_ERR_INVALID, "NULL argument");
        const std::string k(key);
        if (k == "device_assoc") {
            AIC_REQUIRE(!value || t->t.dev_capable(), AIC_ERR_INVALID,
                        "device association needs nn_budget > 0, max_tracks <= 512 and a feature dimension divisible by 4");
            if (!value) t->t.to_host();
            t->t.dev_assoc = value != 0;
        } else if (k == "wave_cascade") {
            t->t.wave_cascade = value != 0;
        } else if (k == "lsap_fast") {
            t->t.lsap_fast = value != 0;
        } else if (k == "epoch_frames") {
            AIC_REQUIRE(value >= 0 && value <= TRK_KMAX, AIC_ERR_INVALID, "epoch_frames must be in 0..16 (0 = default)");
            t->t.epoch_frames = value;
        } else AIC_REQUIRE(false, AIC_ERR_INVALID, "unknown tracker option: " + k);
    });
}

// The device cascade + LSAP of kernels_trk_dev.hip on caller-provided matrices of one frame (parity tests).
int aic_match_cascade_device(int device_id, const float* app, const float* maha, const float* iou, int t, int n, const int32_t* state,
                             const int32_t* tsu, double max_cosine_distance, double max_iou_distance, int max_age, int flags,
                             int32_t* match_det_of_track, int32_t* n_fast_lsap) {
    return guarded([&] {
        AIC_REQUIRE(t >= 0 && n >= 0 && t <= TRK_DEV_TMAX && n <= TRK_DEV_NMAX, AIC_ERR_CAPACITY, "at most 512 tracks x 512 detections");
        if (t == 0) return;
        AIC_REQUIRE(state && tsu && match_det_of_track && (n == 0 || (app && maha && iou)), AIC_ERR_INVALID, "NULL argument");
        Device& d = device(device_id);
        d.use();
        hipStream_t s = d.s_trk;
        const size_t stride = (size_t)TRK_DEV_TMAX * TRK_DEV_NMAX, tn = (size_t)t * n;
        DevBuf<float> costs(3 * stride), sub(stride);
        DevBuf<int> st(t), ts(t), md(t + 3);
        if (tn) {
            HIP_CHECK(hipMemcpyAsync(costs.p, app, tn * 4, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(costs.p + stride, maha, tn * 4, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(costs.p + 2 * stride, iou, tn * 4, hipMemcpyHostToDevice, s));
        }
        HIP_CHECK(hipMemcpyAsync(st.p, state, (size_t)t * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(ts.p, tsu, (size_t)t * 4, hipMemcpyHostToDevice, s));
        TrkDevParams prm{};
        prm.max_cos = (float)max_cosine_distance, prm.clamp_cos = (float)(max_cosine_distance + 1e-5);
        prm.max_iou = (float)max_iou_distance, prm.clamp_iou = (float)(max_iou_distance + 1e-5);
        prm.max_age = max_age, prm.n_init = 3, prm.gmax = 1, prm.dim = 0, prm.cap = TRK_DEV_TMAX;
        prm.no_fast = (flags & 2) ? 1 : 0;
        prm.no_wave = (flags & 4) ? 1 : 0;
        EpochScratch scr{nullptr, nullptr, costs.p, sub.p, nullptr};
        launch_trk_cascade_test(prm, scr, t, n, st.p, ts.p, md.p, md.p + t, flags & 1, s);
        std::vector<int> out(t + 3);
        HIP_CHECK(hipMemcpyAsync(out.data(), md.p, (size_t)(t + 3) * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        AIC_REQUIRE(out[t] == 0, AIC_ERR_INVALID, "cost matrix contains NaN/-inf or is infeasible");
        std::copy(out.begin(), out.begin() + t, match_det_of_track);
        if (n_fast_lsap) { n_fast_lsap[0] = out[t + 1]; n_fast_lsap[1] = out[t + 2]; }
    });
}

int aic_tracker_predict(aic_tracker* t) {
    return guarded([&] {
        AIC_REQUIRE(t, AIC_ERR_INVALID, "NULL tracker");
        t->t.predict();
    });
}

int aic_tracker_update(aic_tracker* t, const float* det_tlwh, const float* conf, const int32_t* cls, const float* feat,
                       int feat_mem, const uint8_t* has_feat, int n, int dim) {
    return guarded([&] {
        AIC_REQUIRE(t, AIC_ERR_INVALID, "NULL tracker");
        AIC_REQUIRE(n == 0 || (det_tlwh && conf && cls), AIC_ERR_INVALID, "NULL detection arrays");
        t->t.update(det_tlwh, conf, cls, feat, feat_mem, has_feat, n, dim);
    });
}

int aic_tracker_update_batch(aic_tracker* t, int k, const int32_t* counts, const float* det_tlwh, const float* conf, const int32_t* cls,
                             const float* feat, int feat_mem, const uint8_t* has_feat, int dim, int cap_rows, int32_t* n_out, int32_t* out6,
                             float* out_conf, int32_t* n_match, int32_t* match_track_id, int32_t* match_det) {
    return guarded([&] {
        AIC_REQUIRE(t && (k == 0 || counts), AIC_ERR_INVALID, "NULL argument");
        long total = 0;
        for (int f = 0; f < k; ++f) total += counts[f];
        AIC_REQUIRE(total == 0 || (det_tlwh && conf && cls), AIC_ERR_INVALID, "NULL detection arrays");
        t->t.update_batch(k, counts, det_tlwh, conf, cls, feat, feat_mem, has_feat, dim, cap_rows, n_out, out6, out_conf, n_match, match_track_id, match_det);
    });
}

int aic_tracker_import_state(aic_tracker* t, int n, const int32_t* track_id, const int32_t* state, const int32_t* hits, const int32_t* age,
                             const int32_t* time_since_update, const int32_t* cls, const float* conf, const int32_t* gallery_len,
                             const float* mean, const float* cov, const float* galleries, int dim, int next_track_id) {
    return guarded([&] {
        AIC_REQUIRE(t, AIC_ERR_INVALID, "NULL tracker");
        t->t.import_state(n, track_id, state, hits, age, time_since_update, cls, conf, gallery_len, mean, cov, galleries, dim, next_track_id);
    });
}

int aic_tracker_next_track_id(aic_tracker* t, int32_t* next_id) {
    return guarded([&] {
        AIC_REQUIRE(t && next_id, AIC_ERR_INVALID, "NULL argument");
        t->t.to_host();
        *next_id = t->t.next_id;
    });
}

int aic_tracker_assoc_counters(aic_tracker* t, int64_t* n_fast, int64_t* n_lsap) {
    return guarded([&] {
        AIC_REQUIRE(t, AIC_ERR_INVALID, "NULL tracker");
        if (n_fast) *n_fast = t->t.n_fast;
        if (n_lsap) *n_lsap = t->t.n_lsap;
    });
}

int aic_tracker_outputs(aic_tracker* t, int32_t* out6, float* conf, int cap, int32_t* n_out) {
    return guarded([&] {
        AIC_REQUIRE(t && n_out, AIC_ERR_INVALID, "NULL argument");
        const auto& o = t->t.outputs;
        *n_out = (int32_t)o.size();
        AIC_REQUIRE((int)o.size() <= cap, AIC_ERR_CAPACITY, "output capacity too small");
        for (size_t k = 0; k < o.size(); ++k) {
            int32_t* r = out6 + k * 6;
            r[0] = o[k].x1, r[1] = o[k].y1, r[2] = o[k].x2, r[3] = o[k].y2, r[4] = o[k].id, r[5] = o[k].cls;
            if (conf) conf[k] = o[k].conf;
        }
    });
}

int aic_tracker_num_tracks(const aic_tracker* t, int32_t* n) {
    return guarded([&] {
        AIC_REQUIRE(t && n, AIC_ERR_INVALID, "NULL argument");
        const_cast<aic_tracker*>(t)->t.to_host();
        *n = (int32_t)t->t.tracks.size();
    });
}

int aic_tracker_export(aic_tracker* t, int cap, int32_t* track_id, int32_t* state, int32_t* hits, int32_t* age,
                       int32_t* tsu, int32_t* cls, float* conf, int32_t* gallery_len, float* mean, float* cov) {
    return guarded([&] {
        AIC_REQUIRE(t, AIC_ERR_INVALID, "NULL tracker");
        Tracker& k = t->t;
        k.to_host();
        const int T = (int)k.tracks.size();
        AIC_REQUIRE(T <= cap, AIC_ERR_CAPACITY, "export capacity too small");
        std::vector<float> hm, hc;
        if (mean || cov) {
            k.dev->use();
            k.flush_predict();
            hm.resize((size_t)k.cap * 8);
            hc.resize((size_t)k.cap * 64);
            HIP_CHECK(hipMemcpyAsync(hm.data(), k.d_mean.p, hm.size() * 4, hipMemcpyDeviceToHost, k.dev->s_trk));
            HIP_CHECK(hipMemcpyAsync(hc.data(), k.d_cov.p, hc.size() * 4, hipMemcpyDeviceToHost, k.dev->s_trk));
            HIP_CHECK(hipStreamSynchronize(k.dev->s_trk));
        }
        for (int i = 0; i < T; ++i) {
            const TrackRec& r = k.tracks[i];
            if (track_id) track_id[i] = r.id;
            if (state) state[i] = r.state;
            if (hits) hits[i] = r.hits;
            if (age) age[i] = r.age;
            if (tsu) tsu[i] = r.tsu;
            if (cls) cls[i] = r.cls;
            if (conf) conf[i] = r.conf;
            if (gallery_len) gallery_len[i] = r.glen;
            if (mean) std::copy(hm.begin() + (size_t)r.slot * 8, hm.begin() + (size_t)r.slot * 8 + 8, mean + (size_t)i * 8);
            if (cov) std::copy(hc.begin() + (size_t)r.slot * 64, hc.begin() + (size_t)r.slot * 64 + 64, cov + (size_t)i * 64);
        }
    });
}

int aic_tracker_export_gallery(aic_tracker* t, int index, float* out, int cap_rows) {
    return guarded([&] {
        AIC_REQUIRE(t && out, AIC_ERR_INVALID, "NULL argument");
        Tracker& k = t->t;
        k.to_host();
        AIC_REQUIRE(index >= 0 && index < (int)k.tracks.size(), AIC_ERR_INVALID, "track index out of range");
        const TrackRec& r = k.tracks[index];
        AIC_REQUIRE(r.glen <= cap_rows, AIC_ERR_CAPACITY, "gallery capacity too small");
        k.dev->use();
        for (int g = 0; g < r.glen; ++g) {
            const int pos = (r.ghead + g) % k.gmax;
            HIP_CHECK(hipMemcpyAsync(out + (size_t)g * k.dim, k.d_gal_raw.p + ((size_t)r.slot * k.gmax + pos) * k.dim,
                                     (size_t)k.dim * 4, hipMemcpyDeviceToHost, k.dev->s_trk));
        }
        HIP_CHECK(hipStreamSynchronize(k.dev->s_trk));
    });
}

int aic_tracker_last_matches(aic_tracker* t, int32_t* track_id, int32_t* det, int cap, int32_t* n) {
    return guarded([&] {
        AIC_REQUIRE(t && n, AIC_ERR_INVALID, "NULL argument");
        const auto& m = t->t.last_matches;
        *n = (int32_t)m.size();
        AIC_REQUIRE((int)m.size() <= cap, AIC_ERR_CAPACITY, "match capacity too small");
        for (size_t k = 0; k < m.size(); ++k) track_id[k] = m[k].first, det[k] = m[k].second;
    });
}

int aic_tracker_last_costs(aic_tracker* t, float* app, float* maha, float* iou, int cap, int32_t* t_n, int32_t* d_n) {
    return guarded([&] {
        AIC_REQUIRE(t && t_n && d_n, AIC_ERR_INVALID, "NULL argument");
        Tracker& k = t->t;
        *t_n = k.last_t, *d_n = k.last_n;
        const size_t tn = (size_t)k.last_t * k.last_n;
        AIC_REQUIRE((long)tn <= cap, AIC_ERR_CAPACITY, "cost capacity too small");
        if (app) std::copy(k.last_app.begin(), k.last_app.end(), app);
        if (maha) std::copy(k.last_maha.begin(), k.last_maha.end(), maha);
        if (iou) std::copy(k.last_iou.begin(), k.last_iou.end(), iou);
    });
}

}  // extern "C"
