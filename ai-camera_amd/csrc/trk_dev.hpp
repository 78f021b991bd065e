// trk_dev.hpp -- association on the device, k frames per launch: the structures shared by kernels_trk_dev.hip (kernels) and
// tracker.cpp / pipeline.cpp (host).  SURVEY.md §8(f)-4; what it replaces per frame: one launch + one stream sync + the host
// cascade/LSAP/lifecycle of Tracker::update (src/tracker/core/tracker_core.py:51-81,83-177, linear_assignment.py:19-157).
//
// An EPOCH is k <= TRK_KMAX (16) consecutive frames of one video stream.  Per epoch two launches, no host round trip:
//   trk_epoch_prep_kernel   (many blocks)  every cosine distance the k frames can need, in bulk on the matrix cores:
//                           SM[t][e][d]  = min over the rows of track t's gallery that are still alive after e evictions
//                                          (suffix minima in FIFO order) of max(0, 1 - <row, det d>),
//                           GRAM[a][d]   = max(0, 1 - <det a, det d>) for the rows the epoch itself will append;
//   trk_epoch_kernel        (ONE block)    the per-stream recurrence, frame by frame: Kalman predict, gating + IoU + appearance
//                           rows, matching cascade + rectangular LSAP (SciPy tie rules), lifecycle, Kalman update / initiate,
//                           gallery bookkeeping, output rows.
// The track table (what Tracker::tracks holds on the host path) lives in HBM between launches.
#pragma once
#include <cstdint>

namespace aic {

constexpr int TRK_KMAX = 16;        // frames per epoch (also <= gallery budget: rows appended in an epoch are never evicted in it).  32 until round 4: measured slower than 16
                                    // at every load, and every per-pair loop of the epoch kernel is unrolled to it (33 loads and registers per pair)
constexpr int TRK_DEV_TMAX = 512;   // tracks the single-block kernel handles (= its thread count)
constexpr int TRK_DEV_NMAX = 512;   // detections per frame
constexpr int TRK_DEV_DNMAX = 2048; // detections per epoch

struct DevTrack {                   // one row of the device-resident track table, list order = TrackerCore.tracks order
    int32_t id, state, hits, age, tsu, cls;
    float conf;
    int32_t slot, glen, ghead;
    int32_t pad[2];
};

struct DevTrkHdr {
    int32_t n_tracks, next_id, n_free;
    int32_t err;                    // 0 ok; 1 track slots exhausted; 2 LSAP infeasible / invalid cost; 3 capacity of the kernel exceeded
    int32_t err_frame;              // group frame index the error was raised at (state = the frame before it)
    int32_t frames_done;            // frames processed by the last epoch launch
    int32_t n_fast, n_lsap;         // assignment problems settled by the unique-optimum check / by the LSAP, since the table went to the device
};

struct TrkDevParams {
    float max_cos, clamp_cos, max_iou, clamp_iou;   // fp32 thresholds and clamp values of linear_assignment.py:55-58,76
    int32_t max_age, n_init, gmax, dim, cap;
    int32_t no_fast;                // 1: every assignment problem goes through the LSAP (tests / A-B); 0: unique optima are read off directly
    int32_t no_wave;                // 1: every cascade level goes through the block-wide form (tests / A-B); 0: frames with <= 64 detections walk their levels on one wavefront
};

struct EpochDets {                  // detection arrays of one launch group, device memory, rows = crops in frame order
    const int32_t* frame_n;         // [frames] detections of the frame
    const int32_t* frame_d0;        // [frames] first row of the frame
    const float* tlwh;              // [rows, 4]
    const float* conf;              // [rows]
    const int32_t* cls;             // [rows]
    const int32_t* valid;           // [rows] crop non-empty (feature exists); NULL = all
    const float* feat;              // [rows, dim] raw embeddings (gallery export); NULL = no features at all
    const float* feat_n;            // [rows, dim] unit embeddings
};

struct EpochOut {                   // per frame of the launch group, device memory
    int32_t* n_tracks;              // [frames] confirmed tracks updated in the frame (true count)
    int32_t* rows;                  // [frames, max_rows, 6] x1 y1 x2 y2 id cls
    float* conf;                    // [frames, max_rows]
    int32_t max_rows;
    // debug of the LAST frame of the launch (tracker-level API: aic_tracker_last_matches / _last_costs); may be NULL
    int32_t* dbg_match;             // [0] = count, then (track id, det) pairs
    int32_t* dbg_tn;                // T, N of that frame
    int32_t dbg_stride;             // > 0: dbg_match holds EVERY frame of the launch group, frame f at dbg_match + f * dbg_stride (aic_tracker_update_batch)
};

struct EpochScratch {
    float* sm;                      // [cap, TRK_KMAX + 1, dn_pad]
    float* gram;                    // [dn_pad, dn_pad]
    float* cost;                    // 3 x [TRK_DEV_TMAX * TRK_DEV_NMAX] full matrices of the current frame (app, maha, iou)
    float* sub;                     // LSAP sub-matrix when it does not fit the LDS arena
    int32_t* appends;               // [TRK_DEV_DNMAX, 3] (slot, ring position, epoch row) gallery rows written at the end of the epoch; [3 * TRK_DEV_DNMAX] = their number
};

}  // namespace aic
