/*
 * aicam.h -- C ABI of libaicam.so, the MI355X (gfx950) detect+track hot path.
 *
 * The reference (abdur75648/AI-Camera) is pure Python and has no FFI; its hot path sits
 * behind Python plugin classes.  Each entry point below replaces the
 * arithmetic of the reference call site named in its comment (paths relative to the
 * reference repo) and is what the reference-side ctypes stub in INTEGRATION.md binds.
 *
 * Conventions
 *   - every function returns 0 (AIC_OK) or a negative AIC_ERR_* code; aic_last_error()
 *     returns the message of the last failure on the calling thread;
 *   - plain pointers and sizes only; `mem` arguments say where a buffer lives
 *     (AIC_HOST = host memory, AIC_DEVICE = HBM of the handle's device);
 *   - outputs are caller-owned buffers with explicit capacities;
 *   - one handle = one HIP stream; a handle is not thread-safe, distinct handles are
 *     independent (one process per GPU, one pipeline per video stream);
 *   - there is NO CPU fallback: without a gfx950 device every compute entry point
 *     fails with AIC_ERR_NO_DEVICE.  Host-side integer logic (aic_lsap,
 *     aic_min_cost_matching) is the only code that runs without a GPU, exactly as
 *     the reference keeps it on the host.
 */
#ifndef AICAM_H
#define AICAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIC_ABI_VERSION 2

#define AIC_OK 0
#define AIC_ERR_INVALID (-1)   /* bad argument            (reference: ValueError / TypeError)  */
#define AIC_ERR_NOT_FOUND (-2) /* engine file missing     (reference: FileNotFoundError)        */
#define AIC_ERR_RUNTIME (-3)   /* HIP call / kernel failed (reference: RuntimeError)            */
#define AIC_ERR_NO_DEVICE (-4) /* no gfx950 device visible                                      */
#define AIC_ERR_CAPACITY (-5)  /* output buffer / static capacity too small                     */
#define AIC_ERR_FORMAT (-6)    /* malformed engine file                                         */

#define AIC_HOST 0
#define AIC_DEVICE 1

#define AIC_F32 0 /* fp32 activations, v_mfma_f32_16x16x4_f32: the parity mode           */
#define AIC_F16 1 /* fp16 activations / fp32 accumulate, v_mfma_f32_16x16x32_f16          */

#define AIC_MODEL_YOLO 1
#define AIC_MODEL_REID 2

typedef struct aic_model aic_model;       /* one engine file resident on one GPU          */
typedef struct aic_tracker aic_tracker;   /* DeepSORT core state of one video stream      */
typedef struct aic_pipeline aic_pipeline; /* detector + ReID + tracker over resident frames */

/* ------------------------------------------------------------------ library / device */
const char* aic_last_error(void);
int aic_abi_version(void);
int aic_device_count(int* count);
int aic_device_sync(int device);

/* ------------------------------------------------------------------ engines
 * Replaces TRTEngine (src/trt_utils/trt_engine.py:15-216): deserialise an engine file
 * (here: graph IR + fp32 weights written by ai-camera_amd/engine_file.py), keep it
 * resident, run it on a stream.  max_items = largest batch (frames for YOLO, crops for
 * ReID) the activation arena is sized for. */
int aic_model_load(const char* path, int device, int dtype, int max_items, aic_model** out);
int aic_model_load_mem(const void* blob, size_t nbytes, int device, int dtype, int max_items,
                       aic_model** out);
/* debugging / tests: the first `bytes` bytes of activation buffer `buf` (engine-file buffer index; NHWC, the engine's activation dtype,
 * items of the last run first) -- what a TensorRT user gets by marking a layer as an output (src/trt_utils/trt_engine.py:62-120 lists
 * only the marked I/O tensors). */
int aic_model_read_buffer(aic_model* m, int buf, void* out, size_t bytes);
int aic_model_destroy(aic_model* m);
/* kind, input H/W, classes (YOLO) or feature dim (ReID), anchors per image, conv FLOPs per item */
int aic_model_info(const aic_model* m, int* kind, int* in_h, int* in_w, int* out_dim,
                   int* n_anchors, double* flops_per_item, int* n_convs);

/* TRTEngine.infer for the YOLO engine (trt_engine.py:151-203 called at
 * src/detector/yolo_detector.py:97): images fp32 NCHW RGB in [0,1] -> the four NMS-plugin
 * tensors the detector reads (yolo_detector.py:44-54,108-112): num_dets[B],
 * bboxes[B,max_det,4] (xyxy, letterbox space), scores[B,max_det], labels[B,max_det]. */
int aic_yolo_infer(aic_model* m, const float* images_nchw, int batch, int mem, float conf_thresh,
                   float iou_thresh, int max_det, int32_t* num_dets, float* bboxes, float* scores,
                   int32_t* labels);
/* Raw head of the same engine for parity tests: per anchor 4*reg_max DFL logits and nc class
 * logits, anchors ordered level-major then row-major (8400 at 640x640). Host outputs. */
int aic_yolo_head(aic_model* m, const float* images_nchw, int batch, int mem, float* dfl_logits,
                  float* cls_logits);
/* Decode of the same head (DFL expectation, ltrb->xyxy*stride, arg-max class): boxes[B,A,4],
 * max class logit[B,A], label[B,A]. Host outputs. */
int aic_yolo_decode(aic_model* m, const float* images_nchw, int batch, int mem, float* boxes,
                    float* max_logit, int32_t* labels);

/* TRTEngine.infer for the ReID engine (src/tracker/reid_model.py:111-126): crops fp32 NCHW,
 * ImageNet-normalised -> embeddings[N, feature_dim] fp32. */
int aic_reid_infer(aic_model* m, const float* crops_nchw, int n, int mem, float* embeddings,
                   int out_mem);

/* ------------------------------------------------------------------ pre / post processing
 * letterbox + preprocess_yolo_input (src/utils/image_processing.py:7-70,73-102): u8 BGR HWC
 * frame -> fp32 NCHW RGB /255, padded with 114; also returns r and (pad_w, pad_h). */
int aic_letterbox(int device, const uint8_t* frame_bgr, int h, int w, int out_h, int out_w,
                  float* out_nchw, float* ratio, float* pad_w, float* pad_h);
/* letterbox() itself, any mode (image_processing.py:7-70: auto / scaleFill / scaleup / color): the caller works out the
 * geometry the mode produces (:33-67, host integer logic) and this resamples the frame to unpad_h x unpad_w (cv2.resize
 * INTER_LINEAR, :64) and surrounds it with the constant border (cv2.copyMakeBorder, :68).
 * out: u8 BGR HWC [(unpad_h + top + bottom), (unpad_w + left + right), 3], caller-owned. */
int aic_letterbox_image(int device, const uint8_t* frame_bgr, int h, int w, int unpad_h, int unpad_w, int top, int bottom,
                        int left, int right, int color_b, int color_g, int color_r, uint8_t* out_bgr);
/* _extract_image_crops + preprocess_reid_input (src/tracker/deepsort_tracker.py:143-159,
 * image_processing.py:105-138): int-truncate + clamp boxes, bilinear resize to out_h x out_w,
 * BGR->RGB, (x/255-mean)/std, NCHW. valid[i]=0 for empty crops (their tensor is zero). */
int aic_crop_resize(int device, const uint8_t* frame_bgr, int h, int w, const float* boxes_xyxy,
                    int n, int out_h, int out_w, float* out_nchw, int32_t* valid);

/* YOLODetector.detect (src/detector/yolo_detector.py:68-149) for a batch of same-size frames:
 * letterbox -> engine -> decode+NMS -> score filter -> scale_bboxes
 * (image_processing.py:141-183). boxes are xyxy in original-frame pixels, clipped. */
int aic_detect(aic_model* yolo, const uint8_t* frames_bgr, int batch, int h, int w, int mem,
               float conf_thresh, float iou_thresh, int max_det, int32_t* num_dets, float* boxes_xyxy,
               float* scores, int32_t* labels);

/* crop -> ReID engine (deepsort_tracker.py:104-113 + reid_model.py:67-126) for one frame. */
int aic_reid_embed(aic_model* reid, const uint8_t* frame_bgr, int h, int w, int mem,
                   const float* boxes_xyxy, int n, float* embeddings, int32_t* valid);

/* ------------------------------------------------------------------ Kalman filter (batched)
 * KalmanFilter.initiate/predict/project/update/gating_distance
 * (src/tracker/core/kalman_filter.py:55-83,85-120,122-151,153-204,206-249), n independent
 * filters per launch. mean[n,8], cov[n,8,8], z[n,4] fp32 host arrays. */
int aic_kf_initiate(int device, const float* z, int n, float* mean, float* cov);
int aic_kf_predict(int device, float* mean, float* cov, int n);
/* the same with the time step of KalmanFilter(dt) (kalman_filter.py:34-44: fp32(dt) on the position/velocity diagonal of the
 * motion matrix); aic_kf_predict is dt = 1, the only value the reference's tracker uses (tracker_core.py:30). */
int aic_kf_predict_dt(int device, float* mean, float* cov, int n, float dt);
int aic_kf_project(int device, const float* mean, const float* cov, int n, float* pmean, float* pcov);
int aic_kf_update(int device, float* mean, float* cov, const float* z, int n);
/* d2[n,m]: squared Mahalanobis distance of filter i to measurement zs[i*m+j] (shared_z=0) or
 * zs[j] (shared_z=1); +inf where S is not positive definite (kalman_filter.py:241-247). */
int aic_kf_gating(int device, const float* mean, const float* cov, int n, const float* zs, int m,
                  int shared_z, int only_position, float* d2);

/* ------------------------------------------------------------------ association costs
 * iou_cost (src/tracker/core/matching.py:13-106): 1-IoU[T,N] of tlwh boxes. */
int aic_iou_cost(int device, const float* track_tlwh, int t, const float* det_tlwh, int n, float* cost);
/* appearance_cost_metric + cosine_distance (matching.py:109-217): galleries[T,gmax,dim] with
 * gallery_len[T] valid rows each, det_feat[N,dim], has_feat[N]; cost[T,N] = min over gallery of
 * max(0, 1 - cos), INFTY_COST (1e5) where the gallery is empty or the detection has no feature. */
int aic_appearance_cost(int device, const float* galleries, const int32_t* gallery_len, int t,
                        int gmax, int dim, const float* det_feat, const uint8_t* has_feat, int n,
                        float* cost);

/* scipy.optimize.linear_sum_assignment as called at
 * src/tracker/core/linear_assignment.py:62 (SciPy 1.15.3, rectangular shortest augmenting
 * path, same tie-breaking). HOST code. row_ind/col_ind hold min(nr,nc) entries. */
int aic_lsap(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind);
/* min_cost_matching on a precomputed fp32 cost block (linear_assignment.py:55-88): clamp
 * cost>max to max+1e-5, LSAP, accept iff cost<=max. HOST code. Outputs index into rows/cols. */
int aic_min_cost_matching(const float* cost, int nr, int nc, double max_distance, int32_t* match_row,
                          int32_t* match_col, int32_t* n_match);

/* matching_cascade + the IoU stage of TrackerCore._match (linear_assignment.py:91-157, tracker_core.py:83-177) on the
 * full cost matrices of one frame: app/maha/iou[T,N] (appearance cost, squared Mahalanobis distance, 1-IoU), state[T]
 * (1 tentative, 2 confirmed) and time_since_update[T] after predict().  HOST code.  Outputs: matches as (track index,
 * detection index) in the reference's order, unmatched tracks, unmatched detections; capacities min(T,N), T, N. */
int aic_match_cascade(const float* app, const float* maha, const float* iou, int t, int n, const int32_t* state,
                      const int32_t* time_since_update, double max_cosine_distance, double max_iou_distance,
                      int max_age, int32_t* match_track, int32_t* match_det, int32_t* n_match,
                      int32_t* unmatched_tracks, int32_t* n_unmatched_tracks, int32_t* unmatched_dets,
                      int32_t* n_unmatched_dets);

/* The same cascade run by the DEVICE association kernels (csrc/kernels_trk_dev.hip: wave-parallel restatement of the same
 * SciPy LSAP, preceded by a unique-optimum check that reads the answer off the matrix when it is the only optimum) on one
 * frame's matrices; at most 512 x 512.  flags: bit 0 = appearance stage alone, bit 1 = every problem through the LSAP (no
 * unique-optimum check).  match_det_of_track[T]: the detection each track got, or -1.  n_fast_lsap (may be NULL): [0] =
 * assignment problems settled by the check, [1] = by the LSAP.  Parity-test entry point of SURVEY.md §8(f)-4. */
int aic_match_cascade_device(int device, const float* app, const float* maha, const float* iou, int t, int n,
                             const int32_t* state, const int32_t* time_since_update, double max_cosine_distance,
                             double max_iou_distance, int max_age, int flags, int32_t* match_det_of_track,
                             int32_t* n_fast_lsap);

/* ------------------------------------------------------------------ tracker
 * TrackerCore (src/tracker/core/tracker_core.py:11-198) + Track lifecycle
 * (src/tracker/core/track.py:16-171).  Kalman state and feature galleries live in HBM;
 * lifecycle counters and the matching cascade run on the host. */
typedef struct aic_tracker_params {
    double max_cosine_distance; /* 0.2  src/config.py:23 (a Python float: kept in fp64)  */
    double max_iou_distance;    /* 0.7  src/config.py:26 */
    int32_t nn_budget;         /* 100  src/config.py:29; <=0 means unlimited up to capacity */
    int32_t max_age;           /* 70   src/config.py:27 */
    int32_t n_init;            /* 3    src/config.py:28 */
    int32_t max_tracks;        /* slot capacity (0 -> 512) */
    int32_t feature_dim;       /* 0 -> fixed by the first update */
    int32_t first_track_id;    /* 1    src/tracker/core/track.py:21 (per tracker) */
} aic_tracker_params;

int aic_tracker_create(int device, const aic_tracker_params* p, aic_tracker** out);
int aic_tracker_destroy(aic_tracker* t);
/* "device_assoc" = 1: predict/update run the whole frame (gating, cascade, LSAP, lifecycle, Kalman update) on the device,
 * the track table stays in HBM between calls (what the pipeline does k frames per launch); 0 (default for this entry
 * point): cost matrices on the device, cascade/LSAP/lifecycle in host C++.  Same results either way.
 * "lsap_fast" = 0: every assignment problem of the device path goes through the wave LSAP (default 1: unique optima are read
 * off the matrix).  "epoch_frames" = 1..16: frames per epoch launch of the device path (0 = default 16). */
int aic_tracker_option(aic_tracker* t, const char* key, int value);
/* TrackerCore.predict (tracker_core.py:44-49). */
int aic_tracker_predict(aic_tracker* t);
/* TrackerCore.update (tracker_core.py:51-81) on N detections: tlwh[N,4], conf[N], class id[N],
 * feat[N,dim] (host or device), has_feat[N] (NULL = all). */
int aic_tracker_update(aic_tracker* t, const float* det_tlwh, const float* conf, const int32_t* cls,
                       const float* feat, int feat_mem, const uint8_t* has_feat, int n, int dim);
/* k consecutive frames in ONE call, each a TrackerCore.predict() + TrackerCore.update() (tracker_core.py:44-49, 51-81), run
 * as epochs of the device association (csrc/kernels_trk_dev.hip; frames per epoch: aic_tracker_option "epoch_frames", default
 * 16) -- the pipeline's per-launch-group association with the embeddings supplied by the caller.  counts[k]; the rows of all
 * frames concatenated: det_tlwh[sum,4], conf[sum], class id[sum], feat[sum,dim] (host or device), has_feat[sum] (NULL = all).
 * Per frame (any may be NULL): n_out[k] confirmed tracks updated in the frame (true count), out6[k,cap_rows,6] + out_conf[k,
 * cap_rows] as aic_tracker_outputs; n_match[k], match_track_id / match_det [k,cap_rows] as aic_tracker_last_matches.  n_out and
 * n_match are the TRUE counts: only the first cap_rows rows / matches of a frame are stored, a caller compares the counts with
 * cap_rows to see a clipped frame.  Device
 * path only (nn_budget > 0, max_tracks <= 512, dim % 4 == 0): AIC_ERR_INVALID otherwise.  Same results as k predict/update calls. */
int aic_tracker_update_batch(aic_tracker* t, int k, const int32_t* counts, const float* det_tlwh, const float* conf,
                             const int32_t* cls, const float* feat, int feat_mem, const uint8_t* has_feat, int dim,
                             int cap_rows, int32_t* n_out, int32_t* out6, float* out_conf, int32_t* n_match,
                             int32_t* match_track_id, int32_t* match_det);
/* Confirmed tracks updated this frame, formatted as deepsort_tracker.py:126-141:
 * out[k] = {x1,y1,x2,y2 (round-half-even ints), track_id, class_id}, conf[k]. */
int aic_tracker_outputs(aic_tracker* t, int32_t* out6, float* conf, int cap, int32_t* n_out);
int aic_tracker_num_tracks(const aic_tracker* t, int32_t* n);
/* Attribute surface of Track for callers/tests (track.py): per live track, in list order.
 * Any pointer may be NULL. mean[T,8], cov[T,8,8] are fetched from HBM. */
int aic_tracker_export(aic_tracker* t, int cap, int32_t* track_id, int32_t* state, int32_t* hits,
                       int32_t* age, int32_t* time_since_update, int32_t* cls, float* conf,
                       int32_t* gallery_len, float* mean, float* cov);
/* Gallery of live track `index` in FIFO order (track.py:70-74): out[gallery_len, dim]. */
int aic_tracker_export_gallery(aic_tracker* t, int index, float* out, int cap_rows);
/* The id the next new track will get (Track._next_id, track.py:21; per tracker here). */
int aic_tracker_next_track_id(aic_tracker* t, int32_t* next_id);
/* Inverse of aic_tracker_export + aic_tracker_export_gallery ("export / import of tracker state"): replaces the
 * whole state of TrackerCore.tracks (tracker_core.py:28) by n tracks in list order -- the arrays of aic_tracker_export, the
 * galleries of all tracks concatenated in FIFO order [sum(gallery_len), dim], and the next track id.  A tracker that
 * imports another's export continues exactly as the exporter would have (checkpoint / resume, moving a stream between GPUs).
 * Every argument is checked before the old state is touched. */
int aic_tracker_import_state(aic_tracker* t, int n, const int32_t* track_id, const int32_t* state, const int32_t* hits,
                             const int32_t* age, const int32_t* time_since_update, const int32_t* cls, const float* conf,
                             const int32_t* gallery_len, const float* mean, const float* cov, const float* galleries,
                             int dim, int next_track_id);
/* Device association: assignment problems (one per cascade level + the IoU stage) settled by the unique-optimum check /
 * solved by the wave LSAP since the tracker was created.  Either pointer may be NULL. */
int aic_tracker_assoc_counters(aic_tracker* t, int64_t* n_unique, int64_t* n_lsap);
/* (track_id, detection index) pairs of the last update, and its full cost matrices [T,N]
 * (appearance, squared Mahalanobis, 1-IoU), T = tracks alive before the update. */
int aic_tracker_last_matches(aic_tracker* t, int32_t* track_id, int32_t* det, int cap, int32_t* n);
int aic_tracker_last_costs(aic_tracker* t, float* app, float* maha, float* iou, int cap, int32_t* t_n,
                           int32_t* d_n);

/* ------------------------------------------------------------------ end-to-end pipeline
 * The loop body of src/aicamera_tracker.py:169-207 (detect + track, the reference's own FPS
 * span) over frames that are already resident in HBM, batched: detection and ReID of
 * `batch` frames per launch group, association strictly frame by frame. */
typedef struct aic_pipeline_params {
    int32_t frame_h, frame_w;
    int32_t batch;          /* frames per detection/ReID launch group                    */
    int32_t ring_frames;    /* frames kept resident in HBM                                */
    int32_t max_persons;    /* rows per frame in the caller's track arrays; also sizes the crop buffers a launch
                             * group starts with (they grow: EVERY detection that passes the filter of
                             * deepsort_tracker.py:88-101 is embedded and tracked, none is dropped)      */
    float conf_thresh;      /* 0.3 src/config.py:17 */
    float iou_thresh;       /* 0.5 src/config.py:18 (unused by the reference, F4)        */
    int32_t max_det;        /* 300 (build decision D4)                                    */
    float min_confidence;   /* 0.3 src/config.py:24 */
    int32_t inject;         /* 1: association consumes injected boxes         */
    uint64_t track_class_mask[2]; /* bit c set = class id c is tracked (config.py:53)    */
    aic_tracker_params tracker;
} aic_pipeline_params;

int aic_pipeline_create(aic_model* yolo, aic_model* reid, const aic_pipeline_params* p,
                        aic_pipeline** out);
int aic_pipeline_destroy(aic_pipeline* p);
/* Copy `count` u8 BGR frames into ring slots [slot, slot+count). */
int aic_pipeline_upload(aic_pipeline* p, int slot, const uint8_t* frames_bgr, int count);
/* Planted detections for ring slots (inject=1): counts[count], boxes[count,max_persons,4] ... */
int aic_pipeline_inject(aic_pipeline* p, int slot, int count, const int32_t* counts,
                        const float* boxes_xyxy, const float* conf, const int32_t* cls);
/* Process ring slots [slot, slot+count) in order. Per frame outputs (any may be NULL):
 * n_tracks[count] (the true number of confirmed tracks of the frame; when it exceeds max_persons only the first
 * max_persons rows are stored), tracks[count,max_persons,6] + track_conf as aic_tracker_outputs;
 * n_dets[count], det_boxes[count,max_det,4], det_scores, det_labels from the detector. */
int aic_pipeline_run(aic_pipeline* p, int slot, int count, int32_t* n_tracks, int32_t* tracks6,
                     float* track_conf, int32_t* n_dets, float* det_boxes, float* det_scores,
                     int32_t* det_labels);
/* The range [slot, slot+count) walked `passes` times back to back as one continuous stream (a looped clip): one call,
 * one pipeline fill and one un-overlapped tracker tail for passes*count frames; rows of a later pass overwrite the
 * earlier ones. bench.py's timed region is one such call with passes = K steps. */
int aic_pipeline_run_passes(aic_pipeline* p, int slot, int count, int passes, int32_t* n_tracks, int32_t* tracks6,
                            float* track_conf, int32_t* n_dets);
/* Same, but the frames of this call come from HOST memory (the reference's cap.read() buffers,
 * src/aicamera_tracker.py:170): each launch group's frames are copied into ring slots [slot, slot+count) on a copy
 * stream while the previous group computes. Pin the buffer once with aic_host_register for full PCIe rate. */
int aic_pipeline_run_from_host(aic_pipeline* p, const uint8_t* frames_bgr, int slot, int count, int32_t* n_tracks,
                               int32_t* tracks6, float* track_conf, int32_t* n_dets);
/* The host clip walked `passes` times as one continuous stream (bench.py's timed region: the reference's own span --
 * frame bytes in host memory -> track tuples on the host, src/aicamera_tracker.py:170-207 with yolo_detector.py:91's
 * .to(device) inside). */
int aic_pipeline_run_from_host_passes(aic_pipeline* p, const uint8_t* frames_bgr, int slot, int count, int passes,
                                      int32_t* n_tracks, int32_t* tracks6, float* track_conf, int32_t* n_dets);
/* Launch groups of the last call: frames, wall-clock second at which the group was handed to the pipeline (its H2D /
 * first launch enqueued) and at which its track tuples were on the host. done - submit = latency of every frame of the group. */
int aic_pipeline_group_times(aic_pipeline* p, int32_t* frames, double* submit_s, double* done_s, int cap, int32_t* n);
/* configs[4] of BASELINE.json -- optional cross-camera ReID gallery exchange (not in the reference: README.md:210 lists it
 * as future work; SURVEY.md §8e fixes its form).  enable: every `every_groups` launch groups the pipeline packs, on its tracker
 * stream, a shard fp32 [t_max, 2 + dim] (valid, track id, unit embedding of the newest gallery row) of the stream's first t_max
 * confirmed tracks into the caller's device buffers (two, alternating).  A consumer thread then calls wait(seq) -- blocks
 * until shard `seq` is packed, returns its buffer index and makes the exchange stream (exchange_stream: a hipStream_t the
 * caller runs its RCCL all-gather on, e.g. through torch.cuda.ExternalStream) wait for the pack kernel -- and done(seq) once
 * the collective has consumed the buffer.  shard0_dev = NULL disables.  Two buffers alternate: the stream's association waits
 * only when the consumer is two exchanges behind, and then for at most 60 s (AICAM_XCHG_WAIT_S) before the call fails with
 * AIC_ERR_RUNTIME -- a stuck peer ends the run loudly, it does not hang it. */
int aic_pipeline_exchange_enable(aic_pipeline* p, float* shard0_dev, float* shard1_dev, int t_max, int every_groups);
int aic_pipeline_exchange_stream(aic_pipeline* p, void** stream);
int aic_pipeline_exchange_wait(aic_pipeline* p, int64_t seq, int timeout_ms, int32_t* buffer, int32_t* ready);
int aic_pipeline_exchange_done(aic_pipeline* p, int64_t seq);
/* The annotation pass on an all-gathered set of shards ("consumed read-only by an extra cosine_min_gallery pass"):
 * gathered_dev = fp32 [world, t_max, 2 + dim] in HBM, `stream` = the hipStream_t to run on (the exchange stream; NULL = the
 * device's tracker stream), synchronised before returning.  Host outputs over ALL world * t_max rows (any may be NULL):
 * track_id[i] (-1 = slot empty), near_row[i] = the valid row of ANOTHER rank closest in cosine distance (ties: lowest row; -1 =
 * none), near_dist[i]; annotation[t_max, 3] = (rank, track id, distance) for this rank's rows where that distance is within
 * max_cosine_distance, else -1.  d(i, j) == d(j, i) bit for bit, so every rank derives the same table from the same bytes. */
int aic_gallery_annotate(int device, void* stream, const float* gathered_dev, int world, int rank, int t_max, int dim,
                         double max_cosine_distance, int32_t* track_id, int32_t* near_row, float* near_dist,
                         float* annotation);
/* Cross-camera global-ID policy on top of it (README.md:209 "smarter gallery management in ReID"; BASELINE.json configs[4]).
 * HOST code (csrc/global_id.cpp).  A track's global id is the (rank << 32 | track id) of the first sighting of its identity:
 * new tracks get their own, and two tracks of different cameras that are each other's nearest neighbour within the threshold
 * adopt the smaller of their global ids (transitively).  update() takes the arrays of aic_gallery_annotate; every rank feeds
 * it the same gathered data and therefore holds the same table -- no further communication.  lookup(): -1 = never seen. */
typedef struct aic_gid aic_gid;
int aic_gid_create(int world, aic_gid** out);
int aic_gid_destroy(aic_gid* g);
int aic_gid_update(aic_gid* g, int world, int t_max, const int32_t* track_id, const int32_t* near_row, const float* near_dist,
                   double max_cosine_distance, int32_t* n_links);
int aic_gid_lookup(aic_gid* g, int rank, int track_id, int64_t* global_id);
int aic_gid_size(aic_gid* g, int64_t* n_tracks, int64_t* n_identities, int64_t* n_links);
int aic_host_register(void* ptr, size_t bytes);   /* hipHostRegister: page-lock caller memory */
int aic_host_unregister(void* ptr);
int aic_pipeline_tracker(aic_pipeline* p, aic_tracker** out);
/* Host wall-clock split since the last reset (seconds): issuing launch groups (producer thread), waiting for
 * a group's GPU work, walking its frames through the tracker (association recurrence). */
int aic_pipeline_stats(aic_pipeline* p, double* issue_s, double* wait_s, double* track_s, int64_t* frames, int reset);
/* Runtime options (tests / measurements). "taper": 1 (default) = the last launch group of a call is split into
 * shrinking groups so its un-overlapped tail is short, 0 = full groups only.  "group_frames": frames per launch group
 * (<= batch; 0 = batch).  "device_assoc": 2 = association on the device, k frames per launch (cascade, LSAP and
 * lifecycle in csrc/kernels_trk_dev.hip, no host round trip per frame); 0 = cost matrices on the device, cascade / LSAP /
 * lifecycle in host C++ (csrc/assoc_host.cpp, lsap.cpp), one launch + one sync per frame; 1 (default) = per launch group, on
 * the device while the assignment problems are at most 192 tracks x 192 detections (a few columns per lane of the wave
 * LSAP; unique optima never reach it), else on the host;
 * a group with a frame of more than 512 detections always takes the host chain ("device_assoc_limit": the 192 of the auto mode).  "device_filter" (inject = 0): 1 (default) = the
 * tracker's confidence / class filter runs on the device and ReID is sized from a device-side count, 0 = filter on the host.
 * "split_streams": 1 = crop + ReID of a launch group on a stream of their own beside the next group's detector (more frames/s; the
 * kernels of the two streams stretch each other, so per-launch durations no longer describe the kernels), 0 (default) = one stream.
 * "dual_lane_frames" (default 128): launch groups of at most that many frames alternate between TWO instances of each engine (the second
 * one built on first use: own activation arena, detector workspace and streams), so that the groups of the two chunk contexts run
 * side by side -- a small group is a chain of ~60 short dependent kernels that leaves most of the chip idle; 0 = one lane.
 * Same results in every mode. */
int aic_pipeline_option(aic_pipeline* p, const char* key, int value);
/* launch groups issued on the second lane since the pipeline was created */
int aic_pipeline_lane_groups(aic_pipeline* p, int64_t* lane1_groups);
/* Launch groups whose crop count outgrew the buffers sized from max_persons (handled, not dropped), and frames
 * whose confirmed tracks outnumbered the caller's max_persons rows (n_tracks reports the true count). */
int aic_pipeline_counters(aic_pipeline* p, int64_t* grown_groups, int64_t* clipped_frames);
/* inject = 0: launch groups whose detection filter (src/tracker/deepsort_tracker.py:88-101) ran on the device behind NMS (no host
 * synchronisation between YOLO and ReID) / on the host (one event wait per group), and the extra ReID rounds launched for groups
 * whose surviving detections outnumbered the ReID engine's max_items.  Any pointer may be NULL. */
int aic_pipeline_filter_counters(aic_pipeline* p, int64_t* device_groups, int64_t* host_groups, int64_t* overflow_rounds);
/* Frames, since creation, whose association (src/tracker/core/tracker_core.py:83-177) ran on the device in the epoch
 * kernels / on the host in C++: what the "device_assoc" auto mode actually chose. Either pointer may be NULL. */
int aic_pipeline_assoc_frames(aic_pipeline* p, int64_t* device_frames, int64_t* host_frames);
/* ReID embeddings of every crop of the most recently finished launch group, frame-major (parity tests of the
 * production-size kernel mix): emb[n_rows, dim] host, crops_per_frame[n_frames]. Any output pointer may be NULL. */
int aic_pipeline_group_embeddings(aic_pipeline* p, float* emb, int cap_rows, int32_t* crops_per_frame, int cap_frames,
                                  int32_t* n_rows, int32_t* n_frames, int32_t* dim);
/* Embeddings of the last processed frame (parity tests): emb[n,dim] host. */
int aic_pipeline_last_embeddings(aic_pipeline* p, float* emb, int cap_rows, int32_t* n, int32_t* dim);

/* ------------------------------------------------------------------ overlay (the step after the path)
 * draw_tracks / draw_detections / draw_info_panel (src/utils/visualization.py:9-124,170-228) as ONE kernel on the frame:
 * prims[n,8] = (kind, x0, y0, x1, y1, color B|G<<8|R<<16, text offset, text length | scale<<16); kind 0 = box outline of
 * thickness 2, 1 = filled rectangle (corners inclusive), 2 = 5x7 bitmap text with its top-left corner at (x0, y0).  Painter's
 * order = list order.  The frame (u8 BGR, host or device) is modified in place.  Pixel spec: csrc/kernels_overlay.hip. */
int aic_overlay(int device, uint8_t* frame_bgr, int h, int w, int mem, const int32_t* prims, int n, const uint8_t* text,
                int text_bytes);

/* ------------------------------------------------------------------ measurement
 * HIP-event timing of kernel classes on the streams they are launched on (bench.py roofline).
 * Classes: 0 conv_igemm (MFMA), 1 conv_direct (3-channel stems), 2 pool/upsample/misc,
 * 3 letterbox, 4 crop_resize, 5 decode+nms, 6 tracker kernels. */
#define AIC_PROF_CLASSES 7
/* class_mask: bit c set = time class c; 0 = off; -1 = all classes */
int aic_prof_enable(int device, int class_mask);
int aic_prof_reset(int device);
/* total ms, launches, algorithmic FLOPs and algorithmic bytes accumulated for a class. */
int aic_prof_read(int device, int cls, double* ms, int64_t* launches, double* flops, double* bytes);
/* length (ms, device clock) of the UNION of the class's bracketed intervals since aic_prof_reset: equal to aic_prof_read's ms while
 * one stream carries the class; with brackets open on two streams at once (aic_pipeline_option "split_streams") the overlap is
 * counted once.  -1 when cross-stream event timestamps are unavailable.  (bench.py's roofline denominator; no reference counterpart:
 * the reference times its loop with time.time(), src/aicamera_tracker.py:175,201.) */
int aic_prof_read_union(int device, int cls, double* ms_union);

#ifdef __cplusplus
}
#endif
#endif /* AICAM_H */
