// pipeline.cpp -- the timed loop body of the reference (detect + track,
// src/aicamera_tracker.py:169-207) over frames resident in HBM.
//
// Frame-independent work (letterbox, YOLO, decode+NMS, crop+resize, ReID) is issued for `batch`
// frames per launch group on the device's main stream; association is a per-stream recurrence and
// runs strictly frame by frame on the tracker stream + host (tracker.cpp).  Two chunk contexts
// ping-pong so that the launch group of chunk k+1 is already queued on the GPU while the host walks
// the frames of chunk k.
//
// inject = 1: the seeded engines cannot "see" the planted persons, so the
// detector runs in full on every frame (its outputs are returned) while crop/ReID/association
// consume the planted boxes handed over with aic_pipeline_inject.  inject = 0: the detector's own
// boxes feed the tracker (one extra host round trip per chunk for the order-preserving filter of
// src/tracker/deepsort_tracker.py:88-101).
#include "engine.hpp"
#include "tracker.hpp"
#include "conv_common.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

namespace aic {

struct FrameDets {
    int n = 0, crop0 = 0;
    std::vector<float> tlwh, xyxy, conf;
    std::vector<int32_t> cls;
};

// A LANE = one instance of each engine (activation arena, detector workspace) + the three streams a launch group is issued on.  Lane 0:
// the caller's engines and the device's streams.  Lane 1 (built on first use): second instances for SMALL launch groups, so that the
// groups of the two chunk contexts run side by side on the GPU -- a 16-frame group is a chain of ~60 dependent 25-us kernels that keeps
// the chip 40 % busy; two chains fill the gaps (aic_pipeline_option "dual_lane_frames": largest group that may take lane 1, 0 = off).
struct Lane { Model* yolo = nullptr; Model* reid = nullptr; hipStream_t s_main = nullptr, s_det = nullptr, s_reid = nullptr; };

struct Chunk {
    Lane* ln = nullptr;              // the lane this context's current group was issued on
    int frames = 0, first_slot = 0, n_crops = 0;
    bool dev_mode = false;           // this group's association runs on the device (decided when the group is issued)
    bool prev_dev_mode = true;       // ... and the association of the group this context held before (the device filter is chosen from it)
    int tracks_after = 0;            // live tracks when this context's last group was released (or at the start of the call)
    PinBuf<float> h_boxes;
    PinBuf<int> h_frame_of, h_valid;
    DevBuf<float> d_boxes, d_emb, d_emb_n;
    DevBuf<int> d_frame_of, d_valid;
    // association on the device: per-crop detection arrays + per-frame counts of the group, and the group's output rows
    PinBuf<char> h_meta, h_out;
    DevBuf<char> d_meta, d_out;
    size_t m_n = 0, m_d0 = 0, m_tlwh = 0, m_conf = 0, m_cls = 0, m_bytes = 0;   // offsets into h_meta / d_meta
    PinBuf<int> h_numdets, h_labels;
    PinBuf<float> h_detboxes, h_scores;
    // detection filter on the device (inject = 0): the group's surviving detections are compacted in HBM by det_filter_*_kernel,
    // the host only ever sees the per-frame counts (for planning the association epochs), behind ev_det, on the consumer thread
    bool filt_dev = false;           // this group went through the device filter
    int f_cap = 0;                   // rows the arrays below hold (batch * max_det: the filter cannot overflow them)
    int reid_rows = 0;               // rows the producer's (bounded) ReID round covered
    hipStream_t s_reid_used = nullptr;   // ... and the stream it ran on
    DevBuf<int> d_rank, d_fn, d_fd0, d_total, d_fcls;
    DevBuf<float> d_ftlwh, d_fconf;
    PinBuf<int> h_fn, h_fd0, h_total;
    std::vector<FrameDets> dets;
    hipEvent_t done = nullptr, ev_yolo = nullptr, ev_det = nullptr, ev_reid = nullptr, ev_extra = nullptr;
    hipEvent_t t_begin = nullptr, t_end = nullptr, t_yolo = nullptr;   // AICAM_PIPE_TIMES: GPU timeline of the group on the main stream
};

struct Pipeline {
    Device* dev;
    Model* yolo;
    Model* reid;
    aic_pipeline_params prm;
    std::unique_ptr<aic_tracker> trk_handle;
    Tracker& trk;
    LetterboxGeom geom;
    size_t frame_bytes;
    DevBuf<uint8_t> ring;
    std::vector<int> inj_count, inj_cls;
    std::vector<float> inj_boxes, inj_conf;
    Lane lane[2];
    std::unique_ptr<Model> yolo2, reid2;
    int dual_max = [] { const char* e = getenv("AICAM_DUAL_MAX"); return e ? atoi(e) : 128; }();   // aic_pipeline_option("dual_lane_frames")
    long n_lane1_groups = 0, n_lane1_failed = 0;
    // Launch groups in flight (chunk contexts in use; aic_pipeline_option("in_flight"), 2 or 3; default 2).  4 was measured in round 1 with
    // the association on the host: the GPU never idles, but the tracker chain then queues behind more conv work (84 -> 106 us/frame) and
    // becomes the bound.  3 was measured in round 5 with the association on the device (the consumer releases a context only when the
    // group's epochs have run, so with two contexts the next detector is issued ~20 ms later): 10 755 against 10 960 frames/s with planted
    // boxes, 9 561 against 9 950 with the trained detector's own (tools/own_trained.py, same box) -- slower both ways; 2 stays.
    static constexpr int NCK = 3;
    int nck = [] { const char* e = getenv("AICAM_IN_FLIGHT"); return e ? std::min(NCK, std::max(2, atoi(e))) : 2; }();
    Chunk ck[NCK];
    int dim;
    std::vector<float> last_emb;
    int last_emb_n = 0;
    bool final_group = true;         // the group being walked is the last one of the call: its last frame's embeddings are read back
                                     // (aic_pipeline_last_embeddings).  Every group did that until round 4 -- a BLOCKING copy on the consumer
                                     // thread, queued behind the copy engine's 44 MB frame uploads: ~1.8 ms per 16-frame group
    // host-side wall time (seconds): issuing launch groups, waiting for a group, walking frames through the tracker
    double t_issue = 0, t_wait = 0, t_track = 0;
    long n_frames_done = 0;
    // association on the device, k frames per launch (aic_pipeline_option("device_assoc")): 0 = host C++ cascade / LSAP, one launch +
    // sync per frame; 2 = always on the device; 1 (default) = on the device while a frame's assignment problems are at most assoc_limit x assoc_limit
    // (lsap_wave64 / lsap_wave_reg<2>, and the unique-optimum check in front of them; 192 x 192 since the end of round 3, see assoc_limit), else on the host for that launch group.
    // configs[2] (100 x 100, YOLOv8m at 1080p), round 3: 2 205 frames/s on the device, 2 246 on the host -- both bound by the convs; the
    // device path leaves the host 1 us per frame of work instead of 213
    // auto mode: largest assignment problem side the device takes (aic_pipeline_option("device_assoc_limit")).  192 since the end of round 3:
    // configs[2] (100 persons: 100 detections, ~105 tracks) runs 2 280 / 2 266 frames/s on the device against 2 255 / 2 257 on the host
    // chain, which also costs a core 214 us per frame; the 300-detection frames of the own-detections scene stay on the host (3 740
    // against 6 350 frames/s when forced onto the device: the one-wavefront LSAP over 300 columns).
    int assoc_limit = 192;
    int dev_assoc = getenv("AICAM_TRK_HOST") ? 0 : (getenv("AICAM_TRK_DEV") ? 2 : 1);
    std::atomic<int> tracks_seen{0};      // live tracks after the most recent launch group
    // `tracks_before`: the track count the decision may use.  It must not depend on timing -- the producer issues group k while the
    // consumer may or may not have finished group k - 1 -- so it is the count recorded in the group's chunk context when that
    // context's PREVIOUS group (k - NCK) was released, or the count at the start of the call for the first NCK groups.  The same
    // frames then always take the same path (both give the same rows; a moving choice made a defect of one of them look random).
    bool use_device(int n_max, int tracks_before) const {
        if (!dev_assoc || !trk.dev_capable() || n_max > TRK_DEV_NMAX) return false;   // beyond 512 detections in a frame only the host chain applies (it takes 1536)
        return dev_assoc == 2 || (n_max <= assoc_limit && tracks_before + n_max / 2 <= assoc_limit) || x_shard[0] != nullptr;
    }
    // configs[4]: gallery shard of this stream on the tracker stream, ordered behind the group's association; every launch group
    // counts, whichever path associated it (the ranks' exchange counts must agree).  Double-buffered: the stream's association only
    // waits when the consumer is TWO exchanges behind, and then with a bound -- a stuck peer ends the run loudly instead of hanging it.
    void pack_shard_if_due(hipStream_t s) {
        if (!x_shard[0] || (x_groups++ % x_every) != 0) return;
        std::unique_lock<std::mutex> lk(x_mu);
        static const int wait_s = [] { const char* e = getenv("AICAM_XCHG_WAIT_S"); return e ? std::max(1, atoi(e)) : 60; }();
        const bool ok = x_cv.wait_for(lk, std::chrono::seconds(wait_s), [&] { return x_done >= x_packed - 1; });   // the buffer's previous exchange has been consumed
        AIC_REQUIRE(ok, AIC_ERR_RUNTIME, "gallery exchange: the consumer has not released a shard buffer for " + std::to_string(wait_s) +
                                             " s (a peer rank is stuck or gone)");
        trk.to_device();                                       // (a group associated on the host: the table goes up for the pack)
        const int b = (int)(x_packed & 1);
        launch_gallery_shard(trk.tbl_hdr(), trk.tbl_trk(), trk.d_gal_n.p, trk.gmax, trk.dim, x_shard[b], x_tmax, s);
        HIP_CHECK(hipEventRecord(ev_shard[b], s));
        x_packed += 1;
        lk.unlock();
        x_cv.notify_all();
    }
    bool taper = getenv("AICAM_NO_TAPER") == nullptr;   // aic_pipeline_option("taper")
    bool head_ramp = getenv("AICAM_NO_TAPER") == nullptr && getenv("AICAM_NO_RAMP") == nullptr;   // aic_pipeline_option("head_ramp")
    // inject = 0: the tracker's confidence / class filter (deepsort_tracker.py:88-101) runs on the device behind NMS and ReID is
    // launched for a bound with the count read on the device -- no host synchronisation between YOLO and ReID.  0 = the filter on
    // the host (one hipEventSynchronize per launch group on the producer thread); 1 = per launch group (see stage_a); 2 = always on the
    // device.  aic_pipeline_option("device_filter").
    // Needs the fp16 engine's fused crop + stem (the crop list is read where it was written, in HBM); other engines use the host filter.
    int dev_filter = getenv("AICAM_HOST_FILTER") ? 0 : 1;
    std::mutex reid_mu;              // the ReID engine's host-side launch state: producer (stage A) and consumer (overflow rounds)
    long n_overflow_rounds = 0;      // extra ReID rounds the consumer launched: groups whose crops outnumbered the engine's max_items
    long n_filter_dev_groups = 0, n_filter_host_groups = 0;
    // configs[4]: cross-camera gallery exchange. The pipeline packs a shard of the stream's confirmed tracks on
    // the tracker stream every x_every launch groups; a consumer thread (ai-camera_amd/distributed.py) all-gathers it over
    // RCCL on the exchange stream. Double-buffered: the pipeline only ever waits if the consumer is two exchanges behind.
    float* x_shard[2] = {nullptr, nullptr};
    int x_tmax = 0, x_every = 1;
    long x_groups = 0, x_packed = 0, x_done = 0;
    hipStream_t s_xchg = nullptr;
    hipEvent_t ev_shard[2] = {nullptr, nullptr};
    std::mutex x_mu;
    std::condition_variable x_cv;
    int group_frames = 0;            // frames per launch group (aic_pipeline_option("group_frames")); 0 = prm.batch
    struct GroupTime { int frames; double submit, done; };
    std::vector<GroupTime> group_times;   // per launch group of the last call: stage A start -> track tuples on the host
    std::vector<double> submit_t;
    int last_chunk = -1;             // chunk context of the most recently finished launch group (aic_pipeline_group_embeddings)
    long n_grow = 0;                 // launch groups whose crop count outgrew the buffers sized from max_persons
    long n_rows_clipped = 0;         // frames with more confirmed tracks than the caller's max_persons output rows
    long n_assoc_dev = 0, n_assoc_host = 0;   // frames whose association ran in the epoch kernels / in host C++
    bool split_streams = getenv("AICAM_SPLIT_STREAMS") != nullptr;
    bool pipe_times = getenv("AICAM_PIPE_TIMES") != nullptr;
    hipEvent_t prev_end = nullptr;   // measured: no gain on MI355X
    const uint8_t* host_frames = nullptr;   // run_from_host: frames of the current call in (pinned) host memory
    hipStream_t s_copy = nullptr;
    static constexpr int NCOPY = 8;   // H2D copies in flight: a group's frames cross PCIe up to COPY_AHEAD groups before its launch group is issued
    int copy_ahead = [] { const char* e = getenv("AICAM_COPY_AHEAD"); return e ? std::min(std::max(atoi(e), 1), NCOPY - 2) : 2; }();
    hipEvent_t ev_copy[NCOPY] = {};
    std::vector<int> plan_off, plan_len;   // launch groups of the running call (ring slot offset, frames)
    int plan_slot = 0, copies_issued = 0;
    int host_slot0 = 0;

    Pipeline(Model* y, Model* r, const aic_pipeline_params& p)
        : dev(y->dev), yolo(y), reid(r), prm(p), trk_handle(new aic_tracker(*y->dev, p.tracker)), trk(trk_handle->t) {
        AIC_REQUIRE(y->kind == KIND_YOLO && r->kind == KIND_REID, AIC_ERR_INVALID, "pipeline needs a YOLO and a ReID engine");
        AIC_REQUIRE(y->dev == r->dev, AIC_ERR_INVALID, "engines live on different devices");
        AIC_REQUIRE(p.frame_h > 0 && p.frame_w > 0 && p.batch > 0 && p.ring_frames >= p.batch && p.max_persons > 0,
                    AIC_ERR_INVALID, "bad pipeline geometry");
        AIC_REQUIRE(p.batch <= y->max_items, AIC_ERR_CAPACITY, "batch exceeds the YOLO engine's max_items");
        AIC_REQUIRE(p.max_det > 0 && p.max_det <= y->max_det_cap, AIC_ERR_CAPACITY, "max_det out of range");
        dev->use();
        lane[0] = Lane{y, r, dev->s_main, dev->s_det, dev->s_reid};
        for (Chunk& c : ck) c.ln = &lane[0];
        geom = letterbox_geometry(p.frame_h, p.frame_w, y->in_h, y->in_w);
        frame_bytes = (size_t)p.frame_h * p.frame_w * 3;
        ring.alloc(frame_bytes * p.ring_frames + 64);     // + slack: the crop kernel's 12-byte tap loads may run past the last frame's last byte
        dim = r->out_dim;
        inj_count.assign(p.ring_frames, 0);
        inj_boxes.assign((size_t)p.ring_frames * p.max_persons * 4, 0.f);
        inj_conf.assign((size_t)p.ring_frames * p.max_persons, 0.f);
        inj_cls.assign((size_t)p.ring_frames * p.max_persons, 0);
        HIP_CHECK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
        for (auto& e : ev_copy) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const size_t maxc = (size_t)p.batch * p.max_persons;
        for (Chunk& c : ck) {
            c.h_boxes.alloc(maxc * 4), c.h_frame_of.alloc(maxc), c.h_valid.alloc(maxc);
            c.d_boxes.alloc(maxc * 4), c.d_frame_of.alloc(maxc), c.d_valid.alloc(maxc), c.d_emb.alloc(maxc * dim), c.d_emb_n.alloc(maxc * dim);
            c.h_numdets.alloc(p.batch), c.h_labels.alloc((size_t)p.batch * p.max_det);
            c.h_detboxes.alloc((size_t)p.batch * p.max_det * 4), c.h_scores.alloc((size_t)p.batch * p.max_det);
            c.dets.resize(p.batch);
            HIP_CHECK(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
            HIP_CHECK(hipEventCreate(&c.t_begin)); HIP_CHECK(hipEventCreate(&c.t_end)); HIP_CHECK(hipEventCreate(&c.t_yolo));
            HIP_CHECK(hipEventCreateWithFlags(&c.ev_yolo, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&c.ev_det, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&c.ev_reid, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&c.ev_extra, hipEventDisableTiming));
        }
    }
    ~Pipeline() {
        for (hipStream_t st : {lane[1].s_main, lane[1].s_det, lane[1].s_reid}) if (st) (void)hipStreamDestroy(st);
        for (auto& e : ev_copy) if (e) (void)hipEventDestroy(e);
        if (s_copy) (void)hipStreamDestroy(s_copy);
        for (Chunk& c : ck) {
            if (c.done) (void)hipEventDestroy(c.done);
            if (c.ev_yolo) (void)hipEventDestroy(c.ev_yolo);
            if (c.ev_det) (void)hipEventDestroy(c.ev_det);
            if (c.ev_reid) (void)hipEventDestroy(c.ev_reid);
            if (c.ev_extra) (void)hipEventDestroy(c.ev_extra);
        }
    }

    bool tracked_class(int c) const { return c >= 0 && c < 128 && ((prm.track_class_mask[c >> 6] >> (c & 63)) & 1ull); }

    // deepsort_tracker.py:88-101: order-preserving confidence / class filter. EVERY surviving detection goes on to ReID and
    // the tracker, as in the reference; max_persons only sizes the buffers a launch group starts with (they grow).
    void collect(FrameDets& fd, int n, const float* boxes_xyxy, const float* conf, const int* cls) {
        fd.n = 0;
        fd.tlwh.clear(), fd.xyxy.clear(), fd.conf.clear(), fd.cls.clear();
        for (int i = 0; i < n; ++i) {
            if (!(conf[i] >= prm.min_confidence) || !tracked_class(cls[i])) continue;
            const float* b = boxes_xyxy + (size_t)i * 4;
            fd.tlwh.insert(fd.tlwh.end(), {b[0], b[1], b[2] - b[0], b[3] - b[1]});   // deepsort_tracker.py:185-186
            fd.xyxy.insert(fd.xyxy.end(), b, b + 4);                                    // crops use the xyxy box (:148)
            fd.conf.push_back(conf[i]);
            fd.cls.push_back(cls[i]);
            ++fd.n;
        }
    }

    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // H2D of the launch groups up to index `upto` (one copy + one event each).  A group's ring slots may be overwritten once
    // every earlier group that used an overlapping slot range has finished computing: the groups k-1 .. j-1 around the one
    // being issued (k) are still in flight, so a copy that would touch their slots waits for its own turn.
    void issue_copies(int upto) {
        const int n = (int)plan_off.size();
        while (copies_issued < n && copies_issued <= upto) {
            const int j = copies_issued;
            bool clash = false;
            for (int i = std::max(0, cur_group - (nck - 1)); i < j && !clash; ++i)
                clash = plan_off[i] < plan_off[j] + plan_len[j] && plan_off[j] < plan_off[i] + plan_len[i];
            if (clash && j > cur_group) break;               // copy it when its own group is issued
            if (clash)                                        // a short clip looped: groups k - nck + 1 .. k - 1 may still read these slots -> behind their launch groups
                for (int i = std::max(0, cur_group - (nck - 1)); i < cur_group; ++i)
                    if (plan_off[i] < plan_off[j] + plan_len[j] && plan_off[j] < plan_off[i] + plan_len[i])
                        HIP_CHECK(hipStreamWaitEvent(s_copy, ck[i % nck].done, 0));
            const int slot = plan_slot + plan_off[j];
            HIP_CHECK(hipMemcpyAsync(ring.p + (size_t)slot * frame_bytes, host_frames + (size_t)(slot - host_slot0) * frame_bytes,
                                     (size_t)plan_len[j] * frame_bytes, hipMemcpyHostToDevice, s_copy));
            HIP_CHECK(hipEventRecord(ev_copy[j % NCOPY], s_copy));
            copies_issued += 1;
        }
    }
    int cur_group = 0;

    // second instances of both engines + their streams (first small group of a call; the producer thread has not started yet)
    void ensure_lane1() {
        if (lane[1].yolo) return;
        dev->use();
        const int yi = std::min(yolo->max_items, std::max(dual_max, 1));
        const int ri = std::min(reid->max_items, std::max(yi * prm.max_persons, 64));
        yolo2.reset(new Model(*dev, yolo->blob_copy->data(), yolo->blob_copy->size(), yolo->dtype, yi));
        reid2.reset(new Model(*dev, reid->blob_copy->data(), reid->blob_copy->size(), reid->dtype, ri));
        Lane l{yolo2.get(), reid2.get(), nullptr, nullptr, nullptr};
        HIP_CHECK(hipStreamCreateWithFlags(&l.s_main, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&l.s_det, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&l.s_reid, hipStreamNonBlocking));
        HIP_CHECK(hipDeviceSynchronize());                        // (the new arenas' memsets ran on the device's main stream)
        lane[1] = l;
    }

    void stage_a(Chunk& c, int slot, int frames, int group_index = 0, int lane_index = 0) {
        c.ln = &lane[lane_index];
        if (lane_index) n_lane1_groups += 1;
        cur_group = group_index;
        const double t0 = now();
        hipStream_t s = c.ln->s_main;
        c.frames = frames, c.first_slot = slot;
        if (host_frames) {   // the reference's span: H2D of the group's frames on the copy stream, under the previous groups' compute
            issue_copies(group_index + copy_ahead);
            HIP_CHECK(hipStreamWaitEvent(s, ev_copy[group_index % NCOPY], 0));
            if (split_streams) HIP_CHECK(hipStreamWaitEvent(c.ln->s_reid, ev_copy[group_index % NCOPY], 0));
        }
        const uint8_t* f0 = ring.p + (size_t)slot * frame_bytes;
        if (pipe_times) HIP_CHECK(hipEventRecord(c.t_begin, s));
        c.ln->yolo->reduce_cls = true;           // the pipeline only ever decodes: the class tails store max logit + label themselves
        c.ln->yolo->run_frames(f0, frames, geom, s);
        // decode + NMS + read-back on the side stream: a few latency-bound blocks that overlap the
        // (CU-filling) ReID launch group instead of serialising the main stream
        hipStream_t sd = c.ln->s_det;
        if (pipe_times) HIP_CHECK(hipEventRecord(c.t_yolo, s));
        HIP_CHECK(hipEventRecord(c.ev_yolo, s));
        HIP_CHECK(hipStreamWaitEvent(sd, c.ev_yolo, 0));
        c.ln->yolo->decode_nms(frames, prm.conf_thresh, prm.iou_thresh, prm.max_det, &geom, sd);
        HIP_CHECK(hipMemcpyAsync(c.h_numdets.p, c.ln->yolo->d_numdets.p, (size_t)frames * 4, hipMemcpyDeviceToHost, sd));
        HIP_CHECK(hipMemcpyAsync(c.h_detboxes.p, c.ln->yolo->d_out_boxes_orig.p, (size_t)frames * prm.max_det * 16, hipMemcpyDeviceToHost, sd));
        HIP_CHECK(hipMemcpyAsync(c.h_scores.p, c.ln->yolo->d_out_scores.p, (size_t)frames * prm.max_det * 4, hipMemcpyDeviceToHost, sd));
        HIP_CHECK(hipMemcpyAsync(c.h_labels.p, c.ln->yolo->d_out_labels.p, (size_t)frames * prm.max_det * 4, hipMemcpyDeviceToHost, sd));
        c.filt_dev = false;
        // dev_filter 1 (default): per launch group -- on the device while this context's previous group was associated on the device.  A
        // group associated on the host needs the detection lists and the crop validity on the host anyway (prepare_b rebuilds them on the
        // consumer thread, that mode's critical thread), so the stream-ordered filter only saves a round trip where the association stays
        // on the device too.  The choice reads state the consumer wrote before it released this context: the same frames always take the
        // same path.  dev_filter 2: always on the device.
        if (!prm.inject && dev_filter && (dev_filter == 2 || c.prev_dev_mode) && c.ln->reid->dtype == AIC_F16 && getenv("AICAM_NO_FUSE_CROP") == nullptr) {
            c.ln->reid->in_pix4 = c.ln->reid->input_pix4_ok();
            c.filt_dev = c.ln->reid->in_pix4;
        }
        if (c.filt_dev) {
            stage_a_device_filter(c, f0, frames, s, sd);
            t_issue += now() - t0;
            return;
        }
        HIP_CHECK(hipEventRecord(c.ev_det, sd));
        if (!prm.inject) { HIP_CHECK(hipEventSynchronize(c.ev_det)); n_filter_host_groups += 1; }
        // detection set handed to ReID + association
        int nc = 0;
        for (int f = 0; f < frames; ++f) {
            FrameDets& fd = c.dets[f];
            if (prm.inject) {
                const int sl = slot + f;
                collect(fd, inj_count[sl], inj_boxes.data() + (size_t)sl * prm.max_persons * 4,
                        inj_conf.data() + (size_t)sl * prm.max_persons, inj_cls.data() + (size_t)sl * prm.max_persons);
            } else {
                const int nd = std::min(c.h_numdets.p[f], prm.max_det);
                collect(fd, nd, c.h_detboxes.p + (size_t)f * prm.max_det * 4, c.h_scores.p + (size_t)f * prm.max_det,
                        c.h_labels.p + (size_t)f * prm.max_det);
            }
            fd.crop0 = nc;
            nc += fd.n;
        }
        c.n_crops = nc;
        if ((size_t)nc > c.h_frame_of.n) {          // a crowded group: grow the crop buffers (this context is idle: stage B released it)
            HIP_CHECK(hipStreamSynchronize(s));
            const size_t cap = (size_t)nc + nc / 4;
            c.h_boxes.alloc(cap * 4), c.h_frame_of.alloc(cap), c.h_valid.alloc(cap);
            c.d_boxes.alloc(cap * 4), c.d_frame_of.alloc(cap), c.d_valid.alloc(cap), c.d_emb.alloc(cap * dim), c.d_emb_n.alloc(cap * dim);
            n_grow += 1;
        }
        for (int f = 0; f < frames; ++f) {
            const FrameDets& fd = c.dets[f];
            for (int i = 0; i < fd.n; ++i) {
                std::copy(fd.xyxy.begin() + i * 4, fd.xyxy.begin() + i * 4 + 4, c.h_boxes.p + (size_t)(fd.crop0 + i) * 4);
                c.h_frame_of.p[fd.crop0 + i] = f;
            }
        }
        int n_max = 0;
        for (int f = 0; f < frames; ++f) n_max = std::max(n_max, c.dets[f].n);
        const bool dev_mode = use_device(n_max, c.tracks_after);
        c.dev_mode = dev_mode;
        if (dev_mode) {   // what the epoch kernels read: frame_n[frames] | frame_d0[frames] | tlwh[nc,4] | conf[nc] | cls[nc]
            c.m_n = 0, c.m_d0 = (size_t)frames * 4, c.m_tlwh = (((size_t)frames * 8 + 15) / 16) * 16;
            c.m_conf = c.m_tlwh + (size_t)nc * 16, c.m_cls = c.m_conf + (size_t)nc * 4, c.m_bytes = c.m_cls + (size_t)nc * 4 + 16;
            if (c.m_bytes > c.h_meta.n) { HIP_CHECK(hipStreamSynchronize(s)); c.h_meta.alloc(c.m_bytes + c.m_bytes / 4), c.d_meta.alloc(c.m_bytes + c.m_bytes / 4); }
            int* hn = reinterpret_cast<int*>(c.h_meta.p + c.m_n);
            int* hd = reinterpret_cast<int*>(c.h_meta.p + c.m_d0);
            float* ht = reinterpret_cast<float*>(c.h_meta.p + c.m_tlwh);
            float* hc = reinterpret_cast<float*>(c.h_meta.p + c.m_conf);
            int* hk = reinterpret_cast<int*>(c.h_meta.p + c.m_cls);
            for (int f = 0; f < frames; ++f) {
                const FrameDets& fd = c.dets[f];
                hn[f] = fd.n, hd[f] = fd.crop0;
                if (fd.n) {
                    std::copy(fd.tlwh.begin(), fd.tlwh.end(), ht + (size_t)fd.crop0 * 4);
                    std::copy(fd.conf.begin(), fd.conf.end(), hc + fd.crop0);
                    std::copy(fd.cls.begin(), fd.cls.end(), hk + fd.crop0);
                }
            }
        }
        // crop + ReID on their own stream: in inject mode they do not depend on the detector, and their CU-filling
        // launches backfill the CUs that YOLO's thin layers (50-400 blocks per launch) leave idle
        hipStream_t sr = split_streams ? c.ln->s_reid : s;
        if (nc) {
            HIP_CHECK(hipMemcpyAsync(c.d_boxes.p, c.h_boxes.p, (size_t)nc * 16, hipMemcpyHostToDevice, sr));
            HIP_CHECK(hipMemcpyAsync(c.d_frame_of.p, c.h_frame_of.p, (size_t)nc * 4, hipMemcpyHostToDevice, sr));
            // the engine's host-side launch state (in_pix4, crop_src, n_items_dev) is shared with the consumer thread, which may be running
            // the overflow rounds of a device-filtered group of the OTHER chunk context on the same Model (the filter is chosen per group
            // since round 4: a device-filtered group k and a host-filtered group k + 1 can be in flight together)
            std::lock_guard<std::mutex> lk(reid_mu);
            c.ln->reid->in_pix4 = c.ln->reid->input_pix4_ok();
            static const bool fuse_crop = getenv("AICAM_NO_FUSE_CROP") == nullptr;
            const int round = c.ln->reid->fast_items();
            for (int c0 = 0; c0 < nc; c0 += round) {   // more crops than the ReID arena holds (or than its fast kernels address): several launch groups, nothing dropped
                const int k = std::min(round, nc - c0);
                if (fuse_crop && c.ln->reid->in_pix4) {
                    // crop + resize + normalise inside the ReID stem kernel: the crop tensor (1 GB per 15 360 crops) never exists
                    c.ln->reid->crop_src = CropSrc{f0, prm.frame_h, prm.frame_w, c.d_boxes.p + (size_t)c0 * 4, c.d_frame_of.p + c0, c.d_valid.p + c0};
                } else {
                    Prof pr(*dev, PROF_CROP, sr, 0, (double)k * c.ln->reid->in_h * c.ln->reid->in_w * 19);
                    launch_crop_resize(f0, prm.frame_h, prm.frame_w, c.d_boxes.p + (size_t)c0 * 4, c.d_frame_of.p + c0, k, nullptr, c.ln->reid->in_h,
                                       c.ln->reid->in_w, c.ln->reid->in_pix4 ? 2 : 1, c.ln->reid->dtype, c.ln->reid->input(), c.d_valid.p + c0, sr, true);
                }
                c.ln->reid->run(k, sr);
                c.ln->reid->crop_src.frames = nullptr;
                HIP_CHECK(hipMemcpyAsync(c.d_emb.p + (size_t)c0 * dim, c.ln->reid->embeddings(), (size_t)k * dim * 4, hipMemcpyDeviceToDevice, sr));
            }
            {   // matching.py:126-130 for every detection of the launch group at once
                Prof pr(*dev, PROF_TRK, sr, 0, (double)nc * dim * 8);
                launch_normalize_rows(c.d_emb.p, c.d_emb_n.p, nc, dim, sr);
            }
            HIP_CHECK(hipMemcpyAsync(c.h_valid.p, c.d_valid.p, (size_t)nc * 4, hipMemcpyDeviceToHost, sr));
        }
        if (dev_mode) HIP_CHECK(hipMemcpyAsync(c.d_meta.p, c.h_meta.p, c.m_bytes, hipMemcpyHostToDevice, sr));
        if (split_streams) {
            HIP_CHECK(hipEventRecord(c.ev_reid, sr));
            HIP_CHECK(hipStreamWaitEvent(s, c.ev_reid, 0));
        }
        if (prm.inject) {
            // planted detections: the association needs this group's embeddings (and the detector to be through with the frames), not the
            // detector's OUTPUTS, which are only returned to the caller -- `done` does not wait for decode + NMS + their read-back on the
            // side stream (0.45 ms of a 16-frame group's 4.5 ms chain); stage B waits for ev_det on the host where it copies them out
            if (pipe_times) HIP_CHECK(hipEventRecord(c.t_end, s));
            HIP_CHECK(hipEventRecord(c.done, s));
            HIP_CHECK(hipStreamWaitEvent(s, c.ev_det, 0));   // join: this lane's next group reuses the head buffers
        } else {
            HIP_CHECK(hipStreamWaitEvent(s, c.ev_det, 0));   // join: the next chunk's YOLO reuses the head buffers
            if (pipe_times) HIP_CHECK(hipEventRecord(c.t_end, s));
            HIP_CHECK(hipEventRecord(c.done, s));
        }
        t_issue += now() - t0;
    }

    // inject = 0 with the filter on the device: NMS -> det_filter kernels -> crop + ReID, stream-ordered, no host round trip.
    // The ReID round is launched for a BOUND (the engine's max_items, at most frames * max_det) and reads the crop count on the
    // device (Model::n_items_dev); a group with more crops than that gets its remaining rounds from the consumer thread, which
    // is where the counts first reach the host (prepare_b).
    void stage_a_device_filter(Chunk& c, const uint8_t* f0, int frames, hipStream_t s, hipStream_t sd) {
        const int cap = prm.batch * prm.max_det;
        if (c.f_cap < cap) {                  // first use of this context (it is idle: stage B released it)
            HIP_CHECK(hipStreamSynchronize(s));
            c.d_rank.alloc((size_t)cap), c.d_fn.alloc(prm.batch), c.d_fd0.alloc(prm.batch), c.d_total.alloc(4), c.d_fcls.alloc(cap);
            c.d_ftlwh.alloc((size_t)cap * 4), c.d_fconf.alloc(cap);
            c.h_fn.alloc(prm.batch), c.h_fd0.alloc(prm.batch), c.h_total.alloc(4);
            c.h_boxes.alloc((size_t)cap * 4), c.h_frame_of.alloc(cap), c.h_valid.alloc(cap);
            c.d_boxes.alloc((size_t)cap * 4), c.d_frame_of.alloc(cap), c.d_valid.alloc(cap), c.d_emb.alloc((size_t)cap * dim), c.d_emb_n.alloc((size_t)cap * dim);
            c.f_cap = cap;
        }
        DetFilterArgs fa{};
        fa.num_dets = c.ln->yolo->d_numdets.p, fa.boxes = c.ln->yolo->d_out_boxes_orig.p, fa.scores = c.ln->yolo->d_out_scores.p, fa.labels = c.ln->yolo->d_out_labels.p;
        fa.batch = frames, fa.max_det = prm.max_det, fa.min_conf = prm.min_confidence;
        fa.mask[0] = prm.track_class_mask[0], fa.mask[1] = prm.track_class_mask[1];
        fa.cap = c.f_cap, fa.rank = c.d_rank.p, fa.frame_n = c.d_fn.p, fa.frame_d0 = c.d_fd0.p, fa.total = c.d_total.p;
        fa.xyxy = c.d_boxes.p, fa.tlwh = c.d_ftlwh.p, fa.conf = c.d_fconf.p, fa.cls = c.d_fcls.p, fa.frame_of = c.d_frame_of.p;
        {
            Prof pr(*dev, PROF_DET, sd, 0, (double)frames * prm.max_det * 24);
            launch_det_filter(fa, sd);
        }
        HIP_CHECK(hipMemcpyAsync(c.h_fn.p, c.d_fn.p, (size_t)frames * 4, hipMemcpyDeviceToHost, sd));
        HIP_CHECK(hipMemcpyAsync(c.h_fd0.p, c.d_fd0.p, (size_t)frames * 4, hipMemcpyDeviceToHost, sd));
        HIP_CHECK(hipMemcpyAsync(c.h_total.p, c.d_total.p, 8, hipMemcpyDeviceToHost, sd));
        HIP_CHECK(hipEventRecord(c.ev_det, sd));
        HIP_CHECK(hipStreamWaitEvent(s, c.ev_det, 0));     // the crop list is in HBM: a stream dependency, not a host wait (and this lane's next group reuses the head buffers)
        // split_streams: crop + ReID of this group on the lane's second stream, beside the DETECTOR OF THE NEXT GROUP on the main stream (inside
        // a group ReID depends on the detector; across groups the thin YOLOv8n layers and the CU-filling ReID tiles overlap as they do with
        // planted boxes).  The second stream follows the same event; every later piece of the group (overflow rounds, `done`) stays on it.
        hipStream_t sr = split_streams ? c.ln->s_reid : s;
        c.s_reid_used = sr;
        if (sr != s) HIP_CHECK(hipStreamWaitEvent(sr, c.ev_det, 0));
        // (fast_items: a bound of 16 384 crops would put layer1's tensors at 2^31 elements, past what the fused block / patch kernels address;
        // rows beyond the bound get their overflow rounds from prepare_b like any crowded group's)
        const int bound = std::min(c.ln->reid->fast_items(), frames * prm.max_det);
        c.reid_rows = bound;
        {
            std::lock_guard<std::mutex> lk(reid_mu);
            c.ln->reid->in_pix4 = true;
            c.ln->reid->crop_src = CropSrc{f0, prm.frame_h, prm.frame_w, c.d_boxes.p, c.d_frame_of.p, c.d_valid.p};
            c.ln->reid->n_items_dev = c.d_total.p;
            c.ln->reid->run(bound, sr);
            c.ln->reid->crop_src.frames = nullptr;
            c.ln->reid->n_items_dev = nullptr;
            HIP_CHECK(hipMemcpyAsync(c.d_emb.p, c.ln->reid->embeddings(), (size_t)bound * dim * 4, hipMemcpyDeviceToDevice, sr));
        }
        {
            Prof pr(*dev, PROF_TRK, sr, 0, (double)bound * dim * 8);
            launch_normalize_rows(c.d_emb.p, c.d_emb_n.p, bound, dim, sr, c.d_total.p);
        }
        // crop validity of the round's rows for a group that ends up on the host chain: queued behind the round (61 KB at 15 360 rows), so
        // that prepare_b has no blocking copy of its own
        HIP_CHECK(hipMemcpyAsync(c.h_valid.p, c.d_valid.p, (size_t)bound * 4, hipMemcpyDeviceToHost, sr));
        if (pipe_times) HIP_CHECK(hipEventRecord(c.t_end, sr));
        HIP_CHECK(hipEventRecord(c.done, sr));
        n_filter_dev_groups += 1;
    }

    // Consumer side of a device-filtered group: the per-frame counts are on the host behind ev_det (recorded right after NMS, long
    // before the group's ReID ends).  Plans the association (device / host, deepsort_tracker.py:88-101 redone on the host only for
    // the host path), and launches the ReID rounds the producer's bounded one did not cover.
    void prepare_b(Chunk& c) {
        const double t0 = now();
        HIP_CHECK(hipEventSynchronize(c.ev_det));
        const int total = c.h_total.p[1];
        AIC_REQUIRE(total <= c.f_cap, AIC_ERR_RUNTIME, "detection filter: more rows than batch * max_det");
        c.n_crops = total;
        int n_max = 0;
        for (int f = 0; f < c.frames; ++f) {
            FrameDets& fd = c.dets[f];
            fd.n = c.h_fn.p[f], fd.crop0 = c.h_fd0.p[f];
            n_max = std::max(n_max, fd.n);
        }
        c.dev_mode = use_device(n_max, c.tracks_after);
        if (total > c.reid_rows) {             // a crowded group: every surviving detection is embedded (deepsort_tracker.py:104-113), in further rounds
            hipStream_t s = c.s_reid_used ? c.s_reid_used : c.ln->s_main;
            const uint8_t* f0 = ring.p + (size_t)c.first_slot * frame_bytes;
            std::lock_guard<std::mutex> lk(reid_mu);
            dev->use();
            const int round = c.ln->reid->fast_items();
            for (int c0 = c.reid_rows; c0 < total; c0 += round) {
                const int k = std::min(round, total - c0);
                c.ln->reid->in_pix4 = true;
                c.ln->reid->crop_src = CropSrc{f0, prm.frame_h, prm.frame_w, c.d_boxes.p + (size_t)c0 * 4, c.d_frame_of.p + c0, c.d_valid.p + c0};
                c.ln->reid->run(k, s);
                c.ln->reid->crop_src.frames = nullptr;
                HIP_CHECK(hipMemcpyAsync(c.d_emb.p + (size_t)c0 * dim, c.ln->reid->embeddings(), (size_t)k * dim * 4, hipMemcpyDeviceToDevice, s));
                launch_normalize_rows(c.d_emb.p + (size_t)c0 * dim, c.d_emb_n.p + (size_t)c0 * dim, k, dim, s);
                HIP_CHECK(hipMemcpyAsync(c.h_valid.p + c0, c.d_valid.p + c0, (size_t)k * 4, hipMemcpyDeviceToHost, s));
                n_overflow_rounds += 1;
            }
            HIP_CHECK(hipEventRecord(c.ev_extra, s));
            HIP_CHECK(hipEventRecord(c.done, s));          // `done` now covers the extra rounds too
        }
        if (!c.dev_mode) {                      // association on the host: it needs the detection lists and the crop validity there
            HIP_CHECK(hipEventSynchronize(c.done));
            for (int f = 0; f < c.frames; ++f) {
                FrameDets& fd = c.dets[f];
                const int crop0 = fd.crop0, n_dev = fd.n;
                const int nd = std::min(c.h_numdets.p[f], prm.max_det);
                collect(fd, nd, c.h_detboxes.p + (size_t)f * prm.max_det * 4, c.h_scores.p + (size_t)f * prm.max_det, c.h_labels.p + (size_t)f * prm.max_det);
                fd.crop0 = crop0;
                AIC_REQUIRE(fd.n == n_dev, AIC_ERR_RUNTIME, "detection filter: device and host counts differ");
            }
            // (c.h_valid arrived with the ReID rounds, behind `done`)
        }
        t_wait += now() - t0;
    }

    void stage_b(Chunk& c, int out_base, int32_t* n_tracks, int32_t* tracks6, float* track_conf, int32_t* n_dets,
                 float* det_boxes, float* det_scores, int32_t* det_labels) {
        const double t0 = now();
        n_assoc_host += c.frames;
        HIP_CHECK(hipEventSynchronize(c.done));
        HIP_CHECK(hipEventSynchronize(c.ev_det));              // (inject: `done` does not cover the detector's read-back)
        if (pipe_times) {
            float a = 0, b = 0, g = 0;
            (void)hipEventElapsedTime(&a, c.t_begin, c.t_yolo);
            (void)hipEventElapsedTime(&b, c.t_yolo, c.t_end);
            if (prev_end) (void)hipEventElapsedTime(&g, prev_end, c.t_begin);
            fprintf(stderr, "[pipe_times] group of %d frames: idle before %.2f ms, letterbox+YOLO %.2f ms, crop+ReID %.2f ms\n", c.frames, g, a, b);
            prev_end = c.t_end;
        }
        const double t1 = now();
        t_wait += t1 - t0;
        std::vector<uint8_t> has;
        trk.defer_outputs = true;
        auto emit = [&](int o) {               // the tracker's resolved outputs -> row o of the caller's arrays
            const std::vector<TrackOut>& outs = trk.resolved;
            if (n_tracks) n_tracks[o] = (int32_t)outs.size();       // the true count: rows beyond max_persons are not stored
            if ((int)outs.size() > prm.max_persons) n_rows_clipped += 1;
            for (size_t k = 0; k < outs.size() && (int)k < prm.max_persons; ++k) {
                const TrackOut& t = outs[k];
                if (tracks6) {
                    int32_t* r = tracks6 + ((size_t)o * prm.max_persons + k) * 6;
                    r[0] = t.x1, r[1] = t.y1, r[2] = t.x2, r[3] = t.y2, r[4] = t.id, r[5] = t.cls;
                }
                if (track_conf) track_conf[(size_t)o * prm.max_persons + k] = t.conf;
            }
        };
        std::vector<std::vector<uint8_t>> hasv(c.frames);
        for (int f = 0; f < c.frames; ++f) {
            const FrameDets& fd = c.dets[f];
            hasv[f].resize(fd.n);
            for (int i = 0; i < fd.n; ++i) hasv[f][i] = c.h_valid.p[fd.crop0 + i] ? 1 : 0;   // empty crop -> feature None
        }
        for (int f = 0; f < c.frames; ++f) {
            FrameDets& fd = c.dets[f];
            // the next frame of the group is known: its association rides on this frame's commit launch (one launch per frame)
            NextDets nx{};
            const bool have_next = f + 1 < c.frames;
            if (have_next) {
                const FrameDets& fn = c.dets[f + 1];
                nx.tlwh = fn.tlwh.data(), nx.has = hasv[f + 1].data(), nx.n = fn.n;
                nx.feat_n = fn.n ? c.d_emb_n.p + (size_t)fn.crop0 * dim : nullptr;
            }
            trk.predict();
            trk.update(fd.tlwh.data(), fd.conf.data(), fd.cls.data(), fd.n ? c.d_emb.p + (size_t)fd.crop0 * dim : nullptr,
                       AIC_DEVICE, hasv[f].data(), fd.n, dim, fd.n ? c.d_emb_n.p + (size_t)fd.crop0 * dim : nullptr,
                       have_next ? &nx : nullptr);
            const int o = out_base + f;
            if (f > 0) emit(o - 1);            // frame f-1's boxes came back behind frame f's cost-matrix sync
            if (n_dets) n_dets[o] = c.h_numdets.p[f];
            const size_t md = prm.max_det;
            if (det_boxes) std::copy(c.h_detboxes.p + f * md * 4, c.h_detboxes.p + (f + 1) * md * 4, det_boxes + (size_t)o * md * 4);
            if (det_scores) std::copy(c.h_scores.p + f * md, c.h_scores.p + (f + 1) * md, det_scores + (size_t)o * md);
            if (det_labels) std::copy(c.h_labels.p + f * md, c.h_labels.p + (f + 1) * md, det_labels + (size_t)o * md);
            if (f == c.frames - 1) {
                trk.finish_outputs();          // also fences the chunk's embedding buffers before the producer reuses them
                emit(o);
                if (final_group) {
                    last_emb_n = fd.n;
                    last_emb.resize((size_t)fd.n * dim);
                    if (fd.n) {
                        HIP_CHECK(hipMemcpy(last_emb.data(), c.d_emb.p + (size_t)fd.crop0 * dim, last_emb.size() * 4, hipMemcpyDeviceToHost));
                    }
                }
            }
        }
        trk.defer_outputs = false;             // direct users of the tracker handle get synchronous outputs
        if (x_shard[0]) { pack_shard_if_due(dev->s_trk); HIP_CHECK(hipStreamSynchronize(dev->s_trk)); }
        tracks_seen = (int)trk.tracks.size();
        c.tracks_after = tracks_seen;
        last_chunk = (int)(&c - &ck[0]);
        t_track += now() - t1;
        n_frames_done += c.frames;
    }

    // Stage B with the association on the device: the frames of the group go through the tracker in epochs of k frames, two
    // launches per epoch on the tracker stream and NO host round trip; the host picks up the group's output rows at the end.
    void stage_b_device(Chunk& c, int out_base, int32_t* n_tracks, int32_t* tracks6, float* track_conf, int32_t* n_dets,
                        float* det_boxes, float* det_scores, int32_t* det_labels) {
        const double t0 = now();
        n_assoc_dev += c.frames;
        hipStream_t s = dev->s_trk;
        const int mp = prm.max_persons;
        const size_t o_rows = (((size_t)c.frames * 4 + 15) / 16) * 16, o_conf = o_rows + (size_t)c.frames * mp * 24;
        const size_t obytes = o_conf + (size_t)c.frames * mp * 4;
        if (obytes > c.h_out.n) { c.h_out.alloc(obytes), c.d_out.alloc(obytes); }
        trk.ensure_dim(dim);
        HIP_CHECK(hipStreamWaitEvent(s, c.done, 0));           // the group's embeddings, crop validity and detection arrays are in HBM
        EpochDets dets{reinterpret_cast<const int*>(c.d_meta.p + c.m_n), reinterpret_cast<const int*>(c.d_meta.p + c.m_d0),
                       reinterpret_cast<const float*>(c.d_meta.p + c.m_tlwh), reinterpret_cast<const float*>(c.d_meta.p + c.m_conf),
                       reinterpret_cast<const int*>(c.d_meta.p + c.m_cls), c.d_valid.p, c.d_emb.p, c.d_emb_n.p};
        const int* h_n = reinterpret_cast<const int*>(c.h_meta.p + c.m_n);
        const int* h_d0 = reinterpret_cast<const int*>(c.h_meta.p + c.m_d0);
        if (c.filt_dev) {                                      // the arrays the device filter wrote; counts came back behind ev_det (prepare_b)
            dets = EpochDets{c.d_fn.p, c.d_fd0.p, c.d_ftlwh.p, c.d_fconf.p, c.d_fcls.p, c.d_valid.p, c.d_emb.p, c.d_emb_n.p};
            h_n = c.h_fn.p, h_d0 = c.h_fd0.p;
        }
        EpochOut out{reinterpret_cast<int*>(c.d_out.p), reinterpret_cast<int*>(c.d_out.p + o_rows), reinterpret_cast<float*>(c.d_out.p + o_conf),
                     mp, nullptr, nullptr, 0};
        trk.run_epochs(dets, h_n, h_d0, c.frames, out, s);
        HIP_CHECK(hipMemcpyAsync(c.h_out.p, c.d_out.p, obytes, hipMemcpyDeviceToHost, s));
        pack_shard_if_due(s);
        const double t1 = now();
        t_track += t1 - t0;                                    // host time of the association: planning + launches
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipEventSynchronize(c.ev_det));              // the detector's outputs of the group are on the host (inject: `done` did not cover them)
        trk.check_epochs();
        tracks_seen = reinterpret_cast<const DevTrkHdr*>(trk.h_tbl.p)->n_tracks;
        c.tracks_after = tracks_seen;
        const double t2 = now();
        t_wait += t2 - t1;
        const int* on = reinterpret_cast<const int*>(c.h_out.p);
        const int* orow = reinterpret_cast<const int*>(c.h_out.p + o_rows);
        const float* oc = reinterpret_cast<const float*>(c.h_out.p + o_conf);
        const size_t md = prm.max_det;
        for (int f = 0; f < c.frames; ++f) {
            const int o = out_base + f;
            const int k = std::min(on[f], mp);
            if (n_tracks) n_tracks[o] = on[f];                 // the true count: rows beyond max_persons are not stored
            if (on[f] > mp) n_rows_clipped += 1;
            if (tracks6) std::copy(orow + (size_t)f * mp * 6, orow + ((size_t)f * mp + k) * 6, tracks6 + (size_t)o * mp * 6);
            if (track_conf) std::copy(oc + (size_t)f * mp, oc + (size_t)f * mp + k, track_conf + (size_t)o * mp);
            if (n_dets) n_dets[o] = c.h_numdets.p[f];
            if (det_boxes) std::copy(c.h_detboxes.p + f * md * 4, c.h_detboxes.p + (f + 1) * md * 4, det_boxes + (size_t)o * md * 4);
            if (det_scores) std::copy(c.h_scores.p + f * md, c.h_scores.p + (f + 1) * md, det_scores + (size_t)o * md);
            if (det_labels) std::copy(c.h_labels.p + f * md, c.h_labels.p + (f + 1) * md, det_labels + (size_t)o * md);
        }
        const FrameDets& fl = c.dets[c.frames - 1];
        if (final_group) {
            last_emb_n = fl.n;
            last_emb.resize((size_t)fl.n * dim);
            if (fl.n) HIP_CHECK(hipMemcpy(last_emb.data(), c.d_emb.p + (size_t)fl.crop0 * dim, last_emb.size() * 4, hipMemcpyDeviceToHost));
        }
        last_chunk = (int)(&c - &ck[0]);
        t_track += now() - t2;
        n_frames_done += c.frames;
    }

    // passes > 1: the same ring range is walked `passes` times back to back as ONE continuous stream (outputs of a
    // later pass overwrite the rows of the earlier one): only the very last group of the call has an un-overlapped tail.
    void run(int slot, int count, int32_t* n_tracks, int32_t* tracks6, float* track_conf, int32_t* n_dets, float* det_boxes,
             float* det_scores, int32_t* det_labels, int passes = 1) {
        AIC_REQUIRE(slot >= 0 && count >= 0 && slot + count <= prm.ring_frames, AIC_ERR_INVALID, "slot range outside the ring");
        AIC_REQUIRE(passes >= 1, AIC_ERR_INVALID, "passes must be >= 1");
        dev->use();
        if (count <= 0) return;
        // the association epoch kernel holds one CU while the next group's convs run: persistent conv grids leave it free
        set_conv_cu_budget(dev_assoc && trk.dev_capable() ? dev->n_cu - 1 : dev->n_cu);
        tracks_seen = trk.on_device ? tracks_seen.load() : (int)trk.tracks.size();
        for (auto& c : ck) c.tracks_after = tracks_seen;
        // Launch groups: full batches, then the last batch tapered (1/2, 1/4, ... down to 16 frames): stage B of the
        // final group cannot overlap any GPU work, so a short final group shortens the un-overlapped tail of the call.
        std::vector<int> goff, glen;
        {
            const int gf = group_frames > 0 ? std::min(group_frames, prm.batch) : prm.batch;
            for (int pass = 0; pass < passes; ++pass) {
                const bool last = pass == passes - 1;
                int done = 0;
                // frames coming from host memory: the call's FIRST group cannot start before its frames have crossed PCIe (512 frames:
                // 1.4 GB, 25 ms at the 57 GB/s of this pool -- 3 % of a four-step bench call with nothing to hide it under).  The call
                // therefore opens with a ramp of small groups (gf/16, gf/8, ... gf/2): compute starts after the first 32 frames' copy and
                // every later copy runs under the group before it.  aic_pipeline_option("head_ramp", 0): off.
                if (pass == 0 && host_frames && head_ramp && gf >= 64)
                    for (int g = std::max(16, gf / 16); g < gf && count - done > g + gf; g *= 2) { goff.push_back(done); glen.push_back(g); done += g; }
                while (count - done > gf) { goff.push_back(done); glen.push_back(gf); done += gf; }
                int rem = count - done;
                while (taper && last && rem > 16) {
                    const int g = std::max(16, rem / 2);
                    if (rem - g < 8) break;
                    goff.push_back(done); glen.push_back(g); done += g; rem -= g;
                }
                if (rem > 0) { goff.push_back(done); glen.push_back(rem); }
            }
        }
        const int nchunks = (int)goff.size();
        // lane of every group: odd groups that are small enough take lane 1 (their context's previous group has been released, and the
        // other context's group -- on lane 0 or busy elsewhere -- does not share a buffer or a stream with them)
        std::vector<int> glane(nchunks, 0);
        if (dual_max > 0 && yolo->blob_copy && reid->blob_copy) {
            bool any = false;
            for (int k = 0; k < nchunks; ++k)
                if ((k & 1) && glen[k] <= std::min(dual_max, yolo->max_items) && glen[k] * prm.max_persons <= reid->max_items) { glane[k] = 1; any = true; }
            if (any) {
                try {
                    ensure_lane1();
                } catch (const std::exception&) {        // e.g. no memory for second arenas: one lane does everything, as before round 4
                    yolo2.reset(), reid2.reset();
                    lane[1] = Lane{};
                    dual_max = 0;
                    n_lane1_failed += 1;
                    (void)hipGetLastError();
                }
            }
            for (int k = 0; k < nchunks; ++k)
                if (glane[k] && (!lane[1].yolo || glen[k] > yolo2->max_items)) glane[k] = 0;
        }
        group_times.clear();
        submit_t.assign(nchunks, 0.0);
        plan_off = goff, plan_len = glen, plan_slot = slot, copies_issued = 0;
        // Two host threads: the producer issues the detection/ReID launch groups (stage A, ~100 launches
        // per group), this thread walks the frames of each finished group through the tracker (stage B:
        // small launches + syncs). A chunk context is reissued only after stage B released it.
        std::mutex mu;
        std::condition_variable cv;
        int issued = 0, consumed = 0;
        std::exception_ptr perr;
        std::thread producer([&] {
            try {
                dev->use();
                for (int k = 0; k < nchunks; ++k) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return k < consumed + nck; });
                    }
                    submit_t[k] = now();
                    stage_a(ck[k % nck], slot + goff[k], glen[k], k, glane[k]);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        issued = k + 1;
                    }
                    cv.notify_all();
                }
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                perr = std::current_exception();
                issued = nchunks;
                cv.notify_all();
            }
        });
        std::exception_ptr cerr;
        try {
            for (int k = 0; k < nchunks; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return issued > k; });
                    if (perr) break;
                }
                final_group = k == nchunks - 1;
                if (ck[k % nck].filt_dev) prepare_b(ck[k % nck]);
                ck[k % nck].prev_dev_mode = ck[k % nck].dev_mode;
                if (ck[k % nck].dev_mode) {
                    trk.dev_assoc = true;
                    stage_b_device(ck[k % nck], goff[k], n_tracks, tracks6, track_conf, n_dets, det_boxes, det_scores, det_labels);
                } else {
                    trk.dev_assoc = false;
                    stage_b(ck[k % nck], goff[k], n_tracks, tracks6, track_conf, n_dets, det_boxes, det_scores, det_labels);
                }
                group_times.push_back(GroupTime{glen[k], submit_t[k], now()});
                {
                    std::lock_guard<std::mutex> lk(mu);
                    consumed = k + 1;
                }
                cv.notify_all();
            }
        } catch (...) {
            cerr = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            consumed = nchunks + nck;   // never block the producer again
        }
        cv.notify_all();
        producer.join();
        if (perr) std::rethrow_exception(perr);
        if (cerr) std::rethrow_exception(cerr);
    }
};

}  // namespace aic

using namespace aic;

struct aic_pipeline { Pipeline p; aic_pipeline(Model* y, Model* r, const aic_pipeline_params& q) : p(y, r, q) {} };

extern "C" {

int aic_pipeline_create(aic_model* yolo, aic_model* reid, const aic_pipeline_params* p, aic_pipeline** out) {
    return guarded([&] {
        AIC_REQUIRE(yolo && reid && p && out, AIC_ERR_INVALID, "NULL argument");
        *out = new aic_pipeline(&yolo->m, &reid->m, *p);
    });
}

int aic_pipeline_destroy(aic_pipeline* p) {
    return guarded([&] {
        if (p) { p->p.dev->use(); (void)hipDeviceSynchronize(); }
        delete p;
    });
}

int aic_pipeline_upload(aic_pipeline* p, int slot, const uint8_t* frames, int count) {
    return guarded([&] {
        AIC_REQUIRE(p && frames && count >= 0, AIC_ERR_INVALID, "bad argument");
        Pipeline& q = p->p;
        AIC_REQUIRE(slot >= 0 && slot + count <= q.prm.ring_frames, AIC_ERR_INVALID, "slot range outside the ring");
        q.dev->use();
        HIP_CHECK(hipMemcpy(q.ring.p + (size_t)slot * q.frame_bytes, frames, (size_t)count * q.frame_bytes, hipMemcpyHostToDevice));
    });
}

int aic_pipeline_inject(aic_pipeline* p, int slot, int count, const int32_t* counts, const float* boxes, const float* conf,
                        const int32_t* cls) {
    return guarded([&] {
        AIC_REQUIRE(p && counts && boxes && conf && cls && count >= 0, AIC_ERR_INVALID, "bad argument");
        Pipeline& q = p->p;
        AIC_REQUIRE(slot >= 0 && slot + count <= q.prm.ring_frames, AIC_ERR_INVALID, "slot range outside the ring");
        const int mp = q.prm.max_persons;
        for (int f = 0; f < count; ++f) {
            AIC_REQUIRE(counts[f] >= 0 && counts[f] <= mp, AIC_ERR_CAPACITY, "injected count exceeds max_persons");
            q.inj_count[slot + f] = counts[f];
            std::copy(boxes + (size_t)f * mp * 4, boxes + (size_t)(f + 1) * mp * 4, q.inj_boxes.begin() + (size_t)(slot + f) * mp * 4);
            std::copy(conf + (size_t)f * mp, conf + (size_t)(f + 1) * mp, q.inj_conf.begin() + (size_t)(slot + f) * mp);
            std::copy(cls + (size_t)f * mp, cls + (size_t)(f + 1) * mp, q.inj_cls.begin() + (size_t)(slot + f) * mp);
        }
    });
}

int aic_pipeline_run(aic_pipeline* p, int slot, int count, int32_t* n_tracks, int32_t* tracks6, float* track_conf,
                     int32_t* n_dets, float* det_boxes, float* det_scores, int32_t* det_labels) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL pipeline");
        p->p.run(slot, count, n_tracks, tracks6, track_conf, n_dets, det_boxes, det_scores, det_labels);
    });
}

int aic_pipeline_run_passes(aic_pipeline* p, int slot, int count, int passes, int32_t* n_tracks, int32_t* tracks6, float* track_conf,
                            int32_t* n_dets) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL handle");
        p->p.run(slot, count, n_tracks, tracks6, track_conf, n_dets, nullptr, nullptr, nullptr, passes);
    });
}

int aic_pipeline_run_from_host(aic_pipeline* p, const uint8_t* frames_bgr, int slot, int count, int32_t* n_tracks,
                               int32_t* tracks6, float* track_conf, int32_t* n_dets) {
    return aic_pipeline_run_from_host_passes(p, frames_bgr, slot, count, 1, n_tracks, tracks6, track_conf, n_dets);
}

int aic_pipeline_group_times(aic_pipeline* p, int32_t* frames, double* submit_s, double* done_s, int cap, int32_t* n) {
    return guarded([&] {
        AIC_REQUIRE(p && n, AIC_ERR_INVALID, "NULL argument");
        const auto& g = p->p.group_times;
        *n = (int32_t)g.size();
        AIC_REQUIRE((int)g.size() <= cap || (!frames && !submit_s && !done_s), AIC_ERR_CAPACITY, "group capacity too small");
        for (size_t i = 0; i < g.size(); ++i) {
            if (frames) frames[i] = g[i].frames;
            if (submit_s) submit_s[i] = g[i].submit;
            if (done_s) done_s[i] = g[i].done;
        }
    });
}

int aic_pipeline_run_from_host_passes(aic_pipeline* p, const uint8_t* frames_bgr, int slot, int count, int passes, int32_t* n_tracks,
                                      int32_t* tracks6, float* track_conf, int32_t* n_dets) {
    return guarded([&] {
        AIC_REQUIRE(p && frames_bgr, AIC_ERR_INVALID, "NULL argument");
        Pipeline& q = p->p;
        q.host_frames = frames_bgr;
        q.host_slot0 = slot;
        try {
            q.run(slot, count, n_tracks, tracks6, track_conf, n_dets, nullptr, nullptr, nullptr, passes);
        } catch (...) {
            q.host_frames = nullptr;
            throw;
        }
        q.host_frames = nullptr;
    });
}

int aic_host_register(void* ptr, size_t bytes) {
    return guarded([&] {
        AIC_REQUIRE(ptr && bytes, AIC_ERR_INVALID, "NULL argument");
        HIP_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    });
}

int aic_host_unregister(void* ptr) {
    return guarded([&] {
        AIC_REQUIRE(ptr, AIC_ERR_INVALID, "NULL argument");
        HIP_CHECK(hipHostUnregister(ptr));
    });
}

int aic_pipeline_tracker(aic_pipeline* p, aic_tracker** out) {
    return guarded([&] {
        AIC_REQUIRE(p && out, AIC_ERR_INVALID, "NULL argument");
        *out = p->p.trk_handle.get();   // owned by the pipeline: do not destroy
    });
}

int aic_pipeline_stats(aic_pipeline* p, double* issue_s, double* wait_s, double* track_s, int64_t* frames, int reset) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL pipeline");
        if (issue_s) *issue_s = p->p.t_issue;
        if (wait_s) *wait_s = p->p.t_wait;
        if (track_s) *track_s = p->p.t_track;
        if (frames) *frames = p->p.n_frames_done;
        if (reset) p->p.t_issue = p->p.t_wait = p->p.t_track = 0, p->p.n_frames_done = 0;
    });
}

int aic_pipeline_exchange_enable(aic_pipeline* p, float* shard0_dev, float* shard1_dev, int t_max, int every_groups) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL pipeline");
        Pipeline& q = p->p;
        q.dev->use();
        std::lock_guard<std::mutex> lk(q.x_mu);
        if (!shard0_dev) { q.x_shard[0] = q.x_shard[1] = nullptr; return; }
        AIC_REQUIRE(shard1_dev && t_max > 0 && t_max <= TRK_DEV_TMAX && every_groups >= 1, AIC_ERR_INVALID, "bad exchange arguments");
        AIC_REQUIRE(q.dev_assoc && q.trk.dev_capable(), AIC_ERR_INVALID, "the gallery exchange reads the HBM-resident track table (device association)");
        if (!q.s_xchg) {
            HIP_CHECK(hipStreamCreateWithFlags(&q.s_xchg, hipStreamNonBlocking));
            for (auto& e : q.ev_shard) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        q.x_shard[0] = shard0_dev, q.x_shard[1] = shard1_dev, q.x_tmax = t_max, q.x_every = every_groups;
        q.x_groups = q.x_packed = q.x_done = 0;
    });
}

int aic_pipeline_exchange_stream(aic_pipeline* p, void** stream) {
    return guarded([&] {
        AIC_REQUIRE(p && stream && p->p.s_xchg, AIC_ERR_INVALID, "exchange not enabled");
        *stream = (void*)p->p.s_xchg;
    });
}

int aic_pipeline_exchange_wait(aic_pipeline* p, int64_t seq, int timeout_ms, int32_t* buffer, int32_t* ready) {
    return guarded([&] {
        AIC_REQUIRE(p && buffer && ready, AIC_ERR_INVALID, "NULL argument");
        Pipeline& q = p->p;
        std::unique_lock<std::mutex> lk(q.x_mu);
        *ready = q.x_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return q.x_packed > seq; }) ? 1 : 0;
        if (!*ready) return;
        *buffer = (int32_t)(seq & 1);
        q.dev->use();
        HIP_CHECK(hipStreamWaitEvent(q.s_xchg, q.ev_shard[seq & 1], 0));   // the collective reads the shard behind the pack kernel
    });
}

int aic_gallery_annotate(int device_id, void* stream, const float* gathered_dev, int world, int rank, int t_max, int dim,
                         double max_cosine_distance, int32_t* track_id, int32_t* near_row, float* near_dist, float* annotation) {
    return guarded([&] {
        AIC_REQUIRE(gathered_dev && world >= 1 && rank >= 0 && rank < world && t_max > 0 && dim > 0 && dim % 2 == 0 && dim <= 8192, AIC_ERR_INVALID, "bad argument");
        Device& d = device(device_id);
        d.use();
        hipStream_t s = stream ? (hipStream_t)stream : d.s_trk;
        const int n = world * t_max;
        DevBuf<int> d_i((size_t)2 * n);
        DevBuf<float> d_f(n);
        std::vector<int> ids(n), nr(n);
        std::vector<float> nd(n);
        {
            Prof pr(d, PROF_TRK, s, 2.0 * n * (double)(n - t_max) * dim, (double)n * (2 + dim) * 4);
            launch_gallery_nearest(gathered_dev, world, t_max, dim, d_i.p, d_i.p + n, d_f.p, s);
        }
        HIP_CHECK(hipMemcpyAsync(ids.data(), d_i.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(nr.data(), d_i.p + n, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(nd.data(), d_f.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (track_id) std::copy(ids.begin(), ids.end(), track_id);
        if (near_row) std::copy(nr.begin(), nr.end(), near_row);
        if (near_dist) std::copy(nd.begin(), nd.end(), near_dist);
        if (annotation) {                       // this rank's rows: (rank, track id, distance) of the closest track of another camera within the threshold
            const float thr = (float)max_cosine_distance;
            for (int r = 0; r < t_max; ++r) {
                const int i = rank * t_max + r, j = nr[i];
                float* o = annotation + (size_t)r * 3;
                if (ids[i] >= 0 && j >= 0 && nd[i] <= thr) { o[0] = (float)(j / t_max); o[1] = (float)ids[j]; o[2] = nd[i]; }
                else { o[0] = -1.f; o[1] = -1.f; o[2] = -1.f; }
            }
        }
    });
}

int aic_pipeline_exchange_done(aic_pipeline* p, int64_t seq) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL pipeline");
        {
            std::lock_guard<std::mutex> lk(p->p.x_mu);
            p->p.x_done = std::max<long>(p->p.x_done, (long)seq + 1);
        }
        p->p.x_cv.notify_all();
    });
}

int aic_pipeline_option(aic_pipeline* p, const char* key, int value) {
    return guarded([&] {
        AIC_REQUIRE(p && key, AIC_ERR_INVALID, "NULL argument");
        const std::string k(key);
        if (k == "taper") {                          // like AICAM_NO_TAPER: 0 switches the tail taper AND the head ramp off ("head_ramp" sets the ramp alone)
            p->p.taper = value != 0;
            if (!value) p->p.head_ramp = false;
        }
        else if (k == "split_streams") p->p.split_streams = value != 0;
        else if (k == "device_assoc") {
            AIC_REQUIRE(value >= 0 && value <= 2, AIC_ERR_INVALID, "device_assoc: 0 host, 1 auto, 2 always on the device");
            p->p.dev_assoc = value;
        }
        else if (k == "device_assoc_limit") {
            AIC_REQUIRE(value >= 1 && value <= TRK_DEV_NMAX, AIC_ERR_INVALID, "device_assoc_limit must be in 1..512");
            p->p.assoc_limit = value;
        }
        else if (k == "device_filter") {
            AIC_REQUIRE(value >= 0 && value <= 2, AIC_ERR_INVALID, "device_filter: 0 host filter, 1 on the device while the association is, 2 always on the device");
            p->p.dev_filter = value;
        }
        else if (k == "head_ramp") {
            p->p.head_ramp = value != 0;
        }
        else if (k == "dual_lane_frames") {
            AIC_REQUIRE(value >= 0 && value <= 4096, AIC_ERR_INVALID, "dual_lane_frames must be in 0..4096 (0 = one lane; clamped to the engines' max_items)");
            AIC_REQUIRE(!p->p.lane[1].yolo || value <= p->p.yolo2->max_items, AIC_ERR_INVALID, "dual_lane_frames: the second lane already exists with a smaller arena");
            p->p.dual_max = value;
        }
        else if (k == "in_flight") {
            AIC_REQUIRE(value >= 2 && value <= Pipeline::NCK, AIC_ERR_INVALID, "in_flight: 2 or 3 launch groups");
            p->p.nck = value;
        }
        else if (k == "group_frames") {
            AIC_REQUIRE(value >= 0 && value <= p->p.prm.batch, AIC_ERR_INVALID, "group_frames must be in 0..batch");
            p->p.group_frames = value;
        }
        else AIC_REQUIRE(false, AIC_ERR_INVALID, "unknown pipeline option: " + k);
    });
}

int aic_pipeline_counters(aic_pipeline* p, int64_t* grown_groups, int64_t* clipped_frames) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL pipeline");
        if (grown_groups) *grown_groups = p->p.n_grow;
        if (clipped_frames) *clipped_frames = p->p.n_rows_clipped;
    });
}

int aic_pipeline_lane_groups(aic_pipeline* p, int64_t* lane1_groups) {
    return guarded([&] {
        AIC_REQUIRE(p && lane1_groups, AIC_ERR_INVALID, "NULL argument");
        *lane1_groups = p->p.n_lane1_groups;
    });
}

int aic_pipeline_filter_counters(aic_pipeline* p, int64_t* device_groups, int64_t* host_groups, int64_t* overflow_rounds) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL pipeline");
        if (device_groups) *device_groups = p->p.n_filter_dev_groups;
        if (host_groups) *host_groups = p->p.n_filter_host_groups;
        if (overflow_rounds) *overflow_rounds = p->p.n_overflow_rounds;
    });
}

int aic_pipeline_assoc_frames(aic_pipeline* p, int64_t* device_frames, int64_t* host_frames) {
    return guarded([&] {
        AIC_REQUIRE(p, AIC_ERR_INVALID, "NULL pipeline");
        if (device_frames) *device_frames = p->p.n_assoc_dev;
        if (host_frames) *host_frames = p->p.n_assoc_host;
    });
}

int aic_pipeline_group_embeddings(aic_pipeline* p, float* emb, int cap_rows, int32_t* crops_per_frame, int cap_frames,
                                  int32_t* n_rows, int32_t* n_frames, int32_t* dim) {
    return guarded([&] {
        AIC_REQUIRE(p && n_rows && n_frames && dim, AIC_ERR_INVALID, "NULL argument");
        Pipeline& q = p->p;
        *dim = q.dim, *n_rows = 0, *n_frames = 0;
        if (q.last_chunk < 0) return;
        const Chunk& c = q.ck[q.last_chunk];
        *n_rows = c.n_crops, *n_frames = c.frames;
        if (crops_per_frame) {
            AIC_REQUIRE(c.frames <= cap_frames, AIC_ERR_CAPACITY, "frame capacity too small");
            for (int f = 0; f < c.frames; ++f) crops_per_frame[f] = c.dets[f].n;
        }
        if (emb && c.n_crops) {
            AIC_REQUIRE(c.n_crops <= cap_rows, AIC_ERR_CAPACITY, "embedding capacity too small");
            q.dev->use();
            HIP_CHECK(hipMemcpy(emb, c.d_emb.p, (size_t)c.n_crops * q.dim * 4, hipMemcpyDeviceToHost));
        }
    });
}

int aic_pipeline_last_embeddings(aic_pipeline* p, float* emb, int cap_rows, int32_t* n, int32_t* dim) {
    return guarded([&] {
        AIC_REQUIRE(p && n && dim, AIC_ERR_INVALID, "NULL argument");
        *n = p->p.last_emb_n, *dim = p->p.dim;
        AIC_REQUIRE(p->p.last_emb_n <= cap_rows, AIC_ERR_CAPACITY, "embedding capacity too small");
        if (emb) std::copy(p->p.last_emb.begin(), p->p.last_emb.end(), emb);
    });
}

}  // extern "C"
