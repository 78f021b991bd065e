// tracker.hpp -- DeepSORT core with Kalman state and galleries in HBM, lifecycle + cascade on host.
#pragma once
#include <cstdlib>

#include "kernels.hpp"
#include "trk_dev.hpp"

namespace aic {

enum { TRK_TENTATIVE = 1, TRK_CONFIRMED = 2, TRK_DELETED = 3 };   // src/tracker/core/track.py:10-14

struct TrackRec {
    int id, state, hits, age, tsu, cls;
    float conf;
    int slot;          // row in the device SoA
    int glen, ghead;   // gallery ring: oldest row = ghead, length glen
    int c_kind = 0, c_det = 0, c_kout = 0, c_appos = -1, c_apdet = 0;   // this frame's commit (pipelined launch): 0 none, 1 update, 2 initiate
};

// Detections of the NEXT frame (pipelined form): lets update(f) launch commit(f) + association(f+1) as one kernel
struct NextDets { const float* tlwh; const uint8_t* has; const float* feat_n; int n; };

struct TrackOut { int x1, y1, x2, y2, id, cls; float conf; };

struct Tracker {
    Device* dev = nullptr;
    aic_tracker_params prm{};
    int cap = 0, dim = 0, gmax = 0;
    bool unlimited = false;
    DevBuf<float> d_mean, d_cov, d_gal_raw, d_gal_n;   // galleries: raw rows (export) + unit rows (cost kernel)
    bool pending_predict = false;                       // Kalman predict is folded into the next association launch
    std::vector<TrackRec> tracks;
    std::vector<int> free_slots;
    int next_id = 1;
    // staging
    PinBuf<char> h_stage;
    DevBuf<char> d_stage;
    PinBuf<int> h_slots;
    DevBuf<int> d_slots;
    PinBuf<float> h_cost;
    DevBuf<float> d_cost, d_detn, d_feat, d_tlwh;
    PinBuf<float> h_tlwh;
    // Deferred outputs (the HBM-resident pipeline): update() does not wait for the commit kernel; the boxes of frame f
    // are read back behind the cost-matrix sync of frame f+1 (same in-order stream), or by finish_outputs().
    bool defer_outputs = false;
    PinBuf<char> h_stage2;                 // phase-2 staging: may still be the source of a copy when the next frame starts
    DevBuf<char> d_stage2;
    PinBuf<float> h_tlwh2[2];
    struct OutMeta { int k, id, cls; float conf; };
    struct { bool active = false; int buf = 0; std::vector<OutMeta> meta; } pend;
    int out_parity = 0;
    std::vector<TrackOut> resolved;        // outputs of the most recently resolved frame
    void resolve_pending(bool need_sync);
    void finish_outputs() { resolve_pending(true); }
    // last frame
    std::vector<float> last_app, last_maha, last_iou;
    int last_t = 0, last_n = 0;
    std::vector<std::pair<int, int>> last_matches;   // (track id, det)
    std::vector<TrackOut> outputs;

    // ---- association on the device, k frames per launch (kernels_trk_dev.hip): the track table lives in HBM between launches
    bool dev_assoc = false;            // aic_tracker_option("device_assoc"); the pipeline turns it on when dev_capable()
    bool on_device = false;            // the HBM table is the current one; `tracks` / `free_slots` / `next_id` are stale
    bool wave_cascade = getenv("AICAM_TRK_NOWAVE") == nullptr;  // aic_tracker_option("wave_cascade"): kernels_trk_dev.hip::cascade_wave
    bool lsap_fast = getenv("AICAM_TRK_NOFAST") == nullptr;   // aic_tracker_option("lsap_fast"): unique optima skip the LSAP (kernels_trk_dev.hip::unique_optimum)
    int dev_predicts = 0;              // predict() calls not yet consumed by an update (device path: the epoch kernel predicts)
    DevBuf<char> d_tbl;                // DevTrkHdr | DevTrack[cap] | int free_slots[cap]
    PinBuf<char> h_tbl;
    DevBuf<float> d_sm, d_gram, d_costs, d_sub;
    DevBuf<int> d_appends, d_dbg;
    PinBuf<char> h_api;                // single-frame API staging (aic_tracker_update through the device path)
    DevBuf<char> d_api;
    long acc_fast = 0, acc_lsap = 0;
    long n_fast = 0, n_lsap = 0;       // assignment problems settled by the unique-optimum check / the LSAP (device path, since the table last went up)
    int epoch_frames = 0;              // frames per epoch launch; 0 = AICAM_TRK_K / 16 (aic_tracker_option("epoch_frames"), aic_tracker_update_batch)
    bool dev_capable() const { return !unlimited && cap <= TRK_DEV_TMAX && (dim == 0 || dim % 4 == 0); }
    bool use_device() const { return dev_assoc && dev_capable(); }
    DevTrkHdr* tbl_hdr() { return reinterpret_cast<DevTrkHdr*>(d_tbl.p); }
    DevTrack* tbl_trk() { return reinterpret_cast<DevTrack*>(d_tbl.p + sizeof(DevTrkHdr)); }
    int* tbl_free() { return reinterpret_cast<int*>(d_tbl.p + sizeof(DevTrkHdr) + sizeof(DevTrack) * (size_t)cap); }
    size_t tbl_bytes() const { return sizeof(DevTrkHdr) + (sizeof(DevTrack) + 4) * (size_t)cap; }
    void to_device();
    void to_host();
    // frames [0, frames) of one launch group (h_n / h_d0: host copies of dets.frame_n / frame_d0), all launches on stream s,
    // no host synchronisation; the header copy lands in h_tbl behind them (check_epochs() after the caller's sync)
    void run_epochs(const EpochDets& dets, const int* h_n, const int* h_d0, int frames, const EpochOut& out, hipStream_t s, bool debug = false);
    void check_epochs();
    // k frames (predict + update each) as epochs on the device; features given by the caller (aic_tracker_update_batch)
    void update_batch(int k, const int32_t* counts, const float* det_tlwh, const float* conf, const int32_t* cls, const float* feat, int feat_mem,
                      const uint8_t* has_feat, int dim_in, int cap_rows, int32_t* n_out, int32_t* out6, float* out_conf, int32_t* n_match,
                      int32_t* match_tid, int32_t* match_det);
    // replace the whole state (aic_tracker_import_state): tracks in list order, galleries concatenated in FIFO order
    void import_state(int n, const int32_t* track_id, const int32_t* state, const int32_t* hits, const int32_t* age, const int32_t* tsu,
                      const int32_t* cls, const float* conf, const int32_t* gallery_len, const float* mean, const float* cov,
                      const float* galleries, int dim_in, int next_track_id);
    void update_device(const float* det_tlwh, const float* conf, const int32_t* cls, const float* feat, int feat_mem,
                       const uint8_t* has_feat, int n, int dim_in);

    Tracker(Device& d, const aic_tracker_params& p);
    void ensure_dim(int d);
    void grow_galleries();
    void predict();
    // feat may be host or device memory ([n, dim] fp32)
    void flush_predict();
    // feat_n: the same rows already normalised on the device (rows / max(||row||, 1e-7)), or NULL
    void update(const float* det_tlwh, const float* conf, const int32_t* cls, const float* feat, int feat_mem,
                const uint8_t* has_feat, int n, int dim_in, const float* feat_n = nullptr, const NextDets* nx = nullptr);
    // pipelined form: the association rows of the coming frame were requested (and its Kalman predict applied) by the
    // previous update's launch
    bool pre_rows = false, pre_predicted = false;
    int pre_T = 0, pre_n = 0;
    PinBuf<char> h_step;
    void match(int T, int N, const float* app, const float* maha, const float* iou,
               std::vector<std::pair<int, int>>& matches, std::vector<int>& unmatched_t, std::vector<int>& unmatched_d);
};

}  // namespace aic

struct aic_tracker {
    aic::Tracker t;
    aic_tracker(aic::Device& d, const aic_tracker_params& p) : t(d, p) {}
};
