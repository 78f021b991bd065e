// kernels.hpp -- host-callable launchers of the gfx950 kernels (defined in the .hip files).
#pragma once
#include "common.hpp"

namespace aic {

typedef _Float16 half_t;

// ------------------------------------------------------------------ conv / graph ops (kernels_conv.hip)
struct ConvArgs {
    const void* x;      // NHWC input, element type T
    const void* w;      // packed weights [CoutPad][Kp], element type T, K = (kh, kw, cin)
    const float* bias;  // [CoutPad]
    void* y;            // NHWC output (T, or float when out_f32)
    const void* res;    // residual, same layout family as y (type T)
    int x_cs, x_coff, H, W, Cin;       // pixel stride (elements), channel offset, spatial dims, cin (multiple of 16B/sizeof(T))
    int y_cs, y_coff, Ho, Wo, Cout;
    int r_cs, r_coff, res_mode, act;
    int KH, KW, stride, pad;
    int Kp;             // K padded to a multiple of the K-step
    int M;              // N * Ho * Wo
    int out_f32;
    int cout_pad;       // rows in w / bias (multiple of 128)
    unsigned tap_rows;  // bit kh*KW set for kh < KH (replication pattern of the tap-validity mask)
    const void* zero;   // 64 bytes of zeros in HBM: LDS-DMA source for padded / out-of-range chunks
    int xcd_map;        // 1: blocks take tiles through xcd_tile() (set by launch_conv_igemm)
    const float* bias_init;   // non-NULL (k_order 2 only): the accumulators START from this bias and `bias` points at zeros -- the
                              // weights-resident kernels add their products onto the bias, (bias + sum) and (sum + bias) round differently
    const int* n_dev;   // optional DEVICE-side item count (images / crops of this launch): M then is only an upper bound the grid was sized
                        // for, and tiles past n_dev[0] * Ho * Wo leave at once (ReID behind the on-device detection filter, where the
                        // host does not know the count when it launches).  NULL: M is exact
    int k_order;        // order in which the K-steps (tap row kh, tap column kw, channel chunk cc) are accumulated:
                        //   0 = (kh, kw, cc)  tap-major, the memory order of the packed weights (default);
                        //   1 = (cc, kh, kw)  the order of the ping-pong patch kernel (conv3x3_pp_patch_kernel);
                        //   2 = (kw, cc, kh)  the order of the weights-resident 64-channel kernels (conv3x3_c64_resident / _block).
                        // Set by launch_conv_igemm from the layer SHAPE: a layer one of those kernels can take is accumulated in that
                        // kernel's order by EVERY kernel its batch size may select, so embeddings do not depend on the batch
    // ---- optional SECOND SOURCE: a 1x1 conv of another tensor accumulated into the same outputs (ResNet's downsample branch folded into
    // the block's last conv: relu(conv3x3(t) + b + conv1x1/s(x) + b') is ONE GEMM over K = [window of t | channels of x]).  The packed
    // weight rows hold the window's K columns, then Cin2 more; `bias` is the sum of both.  Only layers walked chunk-major (k_order 1)
    // by the LDS-DMA implicit-GEMM kernels take it; x2's K-steps come after the window's.  x2 == NULL: none.
    const void* x2;       // NHWC, element type T; output pixel (oh, ow) reads x2 pixel (oh * s2, ow * s2)
    int x2_cs, x2_coff, H2, W2, s2, Cin2;
    // ---- optional SPLIT SOURCE of a 1x1 / stride 1 conv: its first Cs input channels are not in x but in another tensor of half the
    // resolution, read at (ih >> 1, iw >> 1) -- a 2x nearest-neighbour upsample that was only ever the first slice of this conv's
    // concatenated input (YOLOv8's neck: up(P5) | P4 -> C2f.cv1).  The upsample launch, its write and three quarters of this conv's read
    // of those channels go away; same values, same K order: bit-identical.  x / x_coff address channel 0 of the concat buffer as
    // before (channels Cs .. Cin-1 are read from it).  xs == NULL: none.
    const void* xs;
    int xs_cs, xs_coff, Hs, Ws, Cs;
    // ---- optional 1x1 "tail" conv run in this conv's epilogue (fp16 only; conv_tail_supported()).  This conv's own output
    // (SiLU(acc + bias) rounded to fp16, exactly what it would have stored) never leaves the registers: it is the B operand of
    // the tail's MFMAs.  y / y_cs / y_coff of THIS conv are then unused.  w_tail == NULL: no tail.
    const void* w_tail;   // packed weights of the 1x1 [t_cout_pad][t_kp], K = this conv's Cout
    const float* b_tail;  // [t_cout_pad]
    void* y_tail;         // NHWC output of the tail (fp16, or float when t_out_f32)
    int t_cout, t_kp, t_y_cs, t_y_coff, t_out_f32, t_act;
    // ---- optional CLASS REDUCTION of the tail (the detect head's class branch, whose logits only ever feed an arg-max): instead of
    // storing t_cout fp32 logits per pixel (320 bytes per anchor written here, read back by decode_kernel) the tail stores their
    // maximum and the FIRST channel that reaches it (np.argmax's rule, as decode_kernel) -- the very fp32 values it would have stored,
    // compared in registers.  Pixel m of the tail's map goes to t_max / t_arg[(m / t_hw) * t_na + t_a0 + m % t_hw]
    // (image-major anchor order of DetArgs).  t_max == NULL: none (y_tail is then written as usual).
    float* t_max; int* t_arg;
    int t_hw, t_a0, t_na;
    // ---- optional BOX DECODE of the tail (the detect head's box branch: 4 sides x 16 DFL bins per pixel): instead of 64 fp32 logits
    // per pixel (256 bytes per anchor written, read back by decode_kernel) the tail stores the decoded xyxy box -- decode_kernel's own
    // arithmetic on the same fp32 values in the same order (max, then exp / running sums bin 0 .. 15, one division per side), so the
    // boxes are the bits decode_kernel would have produced.  Same anchor indexing as t_max; t_w / t_stride: the level's map width and
    // stride.  t_box == NULL: none.
    float* t_box; int t_w, t_stride;
};
// true when launch_conv_igemm can run `lead` with `tail` (a 1x1 / stride 1 / pad 0 conv reading exactly lead's output) in its epilogue:
// fp16, lead = SiLU without residual with Cout 64 or 80 (a wave then owns every channel of its pixels), tail.Cout <= lead.Cout
bool conv_tail_supported(int dtype, const ConvArgs& lead, const ConvArgs& tail);
// true when launch_conv_igemm takes this 1x1 conv with the first cs channels of its input read from a half-resolution tensor (ConvArgs::xs)
bool conv_xs_supported(int dtype, const ConvArgs& a, int cs);
// true when launch_conv_igemm takes a conv of this shape (x2 fields ignored) with a second source (ConvArgs::x2) of cin2 channels
bool conv_x2_supported(int dtype, const ConvArgs& a, int cin2);
// dtype: AIC_F16 or AIC_F32 (type of x / w / res and, unless out_f32, y)
void launch_conv_igemm(int dtype, const ConvArgs& a, hipStream_t s);
// a whole 64-channel BasicBlock (c1: conv3x3+ReLU, c2: conv3x3 + block input, ReLU) in one fp16 kernel with the intermediate in
// LDS (kernels_conv_block.hip); false = pattern / geometry not supported, nothing launched
bool conv_try_c64_block(const ConvArgs& c1, const ConvArgs& c2, hipStream_t s);
// a whole C2f block with 16-channel halves (cv1 1x1 32->32, m.cv1 / m.cv2 3x3 16->16 with shortcut, cv2 1x1 48->32) in one fp16 kernel,
// concat buffer and intermediate in LDS (kernels_conv_c2f.hip); false = pattern / geometry not supported, nothing launched
bool conv_try_c2f16(const ConvArgs& cv1, const ConvArgs& m_cv1, const ConvArgs& m_cv2, const ConvArgs& cv2, hipStream_t s);

// letterbox + conv 3x3/2 (3->16) + SiLU fused (fp16 YOLOv8 stem); false = geometry not supported, nothing launched
struct LetterboxGeom;
bool launch_yolo_stem_fused(const uint8_t* frames, int n, const LetterboxGeom& g, const void* w, const float* bias, int Kp, void* y,
                            int y_cs, int y_coff, int Ho, int Wo, hipStream_t s);

// conv 3x3/1 (3->64) + ReLU + max-pool 3x3/2 fused (fp16, W == 64, H % 8 == 0): ReID stem
// in_stride: halves per input pixel, 8 (NHWC8) or 4 (NHWC4 = RGB0; only where reid_stem2_usable(H, W))
// frames != NULL: the fused fp16 ReID stem resamples every crop from the u8 frames itself (boxes [n,4] xyxy, frame_of[n] or NULL, valid[n] written)
struct CropSrc { const uint8_t* frames; int fh, fw; const float* boxes; const int* frame_of; int* valid; };
// n_dev: optional device-side crop count (see ConvArgs::n_dev)
void launch_reid_stem_pool(const void* x, const void* w, const float* bias, void* y, int n, int H, int W, int Kp, int y_cs,
                           int y_coff, int in_stride, hipStream_t s, const CropSrc* crop = nullptr, const int* n_dev = nullptr);
bool reid_stem2_usable(int H, int W);

struct EltArgs {
    const void* src; void* dst;
    int n, h, w, c;            // source extent, channels processed
    int s_cs, s_coff, d_cs, d_coff;
    const int* n_dev;          // optional device-side item count (n is then the bound the grid was sized for); NULL: n is exact
};
void launch_sppf_pool(int dtype, const EltArgs& a, hipStream_t s);     // writes 3 pooled copies at d_coff + k*c
void launch_upsample2x(int dtype, const EltArgs& a, hipStream_t s);
void launch_maxpool3s2(int dtype, const EltArgs& a, hipStream_t s);
void launch_avgpool(int dtype, const EltArgs& a, hipStream_t s);
void launch_l2norm(int dtype, const EltArgs& a, hipStream_t s);        // dst is float
// fp32 NCHW [n,3,h,w] -> NHWC8 (RGB + zeros) of the activation dtype
void launch_nchw_to_nhwc8(int dtype, const float* src, void* dst, int n, int h, int w, hipStream_t s);
// gather float NHWC buffer rows -> dense float [n, hw, c] (head export)
void launch_copy_f32(const float* src, float* dst, size_t count, hipStream_t s);

// ------------------------------------------------------------------ pre-processing (kernels_pre.hip)
struct LetterboxGeom {
    int src_h, src_w, out_h, out_w, unpad_h, unpad_w, top, left;
    float ratio, pad_w, pad_h;
};
LetterboxGeom letterbox_geometry(int h, int w, int out_h, int out_w);
// frames u8 [n,h,w,3] BGR (device). mode 0: fp32 NCHW RGB/255 ; mode 1: NHWC8 activation dtype
void launch_letterbox(const uint8_t* frames, int n, const LetterboxGeom& g, int mode, int dtype, void* out,
                      hipStream_t s);
void launch_letterbox_u8(const uint8_t* frame, const LetterboxGeom& g, const int color_bgr[3], uint8_t* out, hipStream_t s);
// boxes [n,4] xyxy (device), frame_of[n] (device, may be NULL = frame 0), frames u8 [*,h,w,3].
// n_dev (device int, may be NULL) caps the number of live crops. valid[n] written.
// slack: the caller owns >= 16 readable bytes behind the last frame (12-byte tap loads instead of byte loads; same bytes used)
void launch_crop_resize(const uint8_t* frames, int h, int w, const float* boxes, const int* frame_of, int n,
                        const int* n_dev, int out_h, int out_w, int mode, int dtype, void* out, int* valid,
                        hipStream_t s, bool slack = false);

// ------------------------------------------------------------------ detection head (kernels_det.hip)
struct HeadLevel { const float* box; const float* cls; int h, w, stride, a0; int cls_reduced, box_decoded; };   // cls_reduced / box_decoded: max_logit + labels / boxes of this level were written by the branch's own tail (ConvArgs::t_max / t_box)
struct DetArgs {
    HeadLevel lvl[4];
    int n_levels, n_anchors, nc, reg_max, batch;
    int fast_exp;      // fp16 engines: the DFL softmax's exponential as v_exp_f32(x log2 e), the form the box branches' tails use (tail_1x1); fp32 engines: expf
    float logit_thr, iou_thr;
    int max_det;
    // letterbox undo (K4)
    float pad_w, pad_h, ratio; int orig_w, orig_h;
    // outputs / workspace (device)
    float* boxes;      // [B,A,4]
    float* max_logit;  // [B,A]
    int* labels;       // [B,A]
    unsigned long long* keys;  // [B,A] sort keys
    int* n_cand;       // [B]
    int* num_dets;     // [B]
    float* out_boxes;  // [B,max_det,4] letterbox space
    float* out_boxes_orig;  // [B,max_det,4] original frame space (may be NULL)
    float* out_scores; // [B,max_det]
    int* out_labels;   // [B,max_det]
};
void launch_decode(const DetArgs& a, hipStream_t s);
// deepsort_tracker.py:88-101 on the device: order-preserving confidence / class filter of a launch group's NMS outputs, compacted
// into the arrays crop + ReID + the association epochs read (kernels_det.hip)
struct DetFilterArgs {
    const int* num_dets; const float* boxes; const float* scores; const int* labels;   // [B], [B,max_det,4] original-frame xyxy, [B,max_det] x 2
    int batch, max_det;
    float min_conf;
    unsigned long long mask[2];     // bit c = class c is tracked
    int cap;                        // rows the compact arrays hold (batch * max_det never overflows)
    int* rank;                      // scratch [B, max_det]
    int* frame_n; int* frame_d0;    // [B] kept detections of the frame, first row of the frame
    int* total;                     // [2]: rows present (<= cap), rows the filter passed
    float* xyxy; float* tlwh; float* conf; int* cls; int* frame_of;   // [cap, 4] x 2, [cap] x 3
};
void launch_det_filter(const DetFilterArgs& a, hipStream_t s);
void launch_select_sort_nms(const DetArgs& a, hipStream_t s);

// ------------------------------------------------------------------ tracker (kernels_trk.hip)
void launch_kf_initiate(const float* z, int n, float* mean, float* cov, const int* slots, hipStream_t s);
void launch_kf_initiate_idx(const float* z, const int* zidx, int n, float* mean, float* cov, const int* slots, hipStream_t s);
void launch_kf_predict(float* mean, float* cov, const int* slots, int n, hipStream_t s, float dt = 1.f);
void launch_kf_project(const float* mean, const float* cov, int n, float* pmean, float* pcov, hipStream_t s);
void launch_kf_update(float* mean, float* cov, const int* slots, const float* z, const int* zidx, int n,
                      float* out_tlwh, hipStream_t s);
void launch_kf_gating(const float* mean, const float* cov, const int* slots, int n, const float* zs, int m,
                      int shared_z, int only_position, float* d2, hipStream_t s);
void launch_iou_cost(const float* trk_tlwh, const float* mean, const int* slots, int t, const float* det_tlwh,
                     int n, float* cost, hipStream_t s);
void launch_fill(float* p, float v, size_t n, hipStream_t s);
void launch_normalize_rows(const float* src, float* dst, int n, int dim, hipStream_t s, const int* n_dev = nullptr);
// galleries: base [slots][gmax][dim]; per row t: slot index + valid length; det_n normalised rows.
void launch_cosine_min(const float* gal, const int* slots, const int* glen, int t, int gmax, int dim,
                       const float* det_n, const unsigned char* has_feat, int n, float* cost, hipStream_t s);
void launch_gallery_append(float* gal, int gmax, int dim, const int* slot, const int* pos, const int* det,
                           const float* feat, int count, hipStream_t s);

// fused per-frame tracker kernels
void launch_trk_assoc(float* mean, float* cov, const int* slots, int t, int do_predict, const float* det_tlwh,
                      const float* det_xyah, int n, float* app, float* d2, float* iouc, hipStream_t s);
void launch_cosine_min_mfma(const float* gal_n, const int* slots, const int* glen, int t, int gmax, int dim, const float* det_n,
                            const unsigned char* has_feat, int n, float* cost, hipStream_t s);
void launch_trk_assoc_all(float* mean, float* cov, const int* slots, const int* glen, int t, int do_predict, const float* det_tlwh,
                          const float* det_xyah, const float* gal_n, int gmax, int dim, const float* det_n,
                          const unsigned char* has_feat, int n, float* app, float* d2, float* iouc, hipStream_t s);
// the same launch preceded, per track, by the commit of the previous frame (Kalman update / initiate / gallery row);
// c_kind == nullptr: association only; n == 0: commit only
void launch_trk_step(float* mean, float* cov, const int* slots, const int* glen, int t, int do_predict, const float* det_tlwh,
                     const float* det_xyah, const float* gal_n, int gmax, int dim, const float* det_n,
                     const unsigned char* has_feat, int n, float* app, float* d2, float* iouc,
                     const int* c_kind, const int* c_det, const int* c_kout, const int* c_appos, const int* c_apdet,
                     const float* c_xyah, const float* c_feat, const float* c_feat_n, float* c_out_tlwh, float* c_gal_raw, float* c_gal_w,
                     hipStream_t s);
void launch_trk_commit(float* mean, float* cov, const int* lists, int M, int U, int A, const float* xyah, float* out_tlwh,
                       float* gal_raw, float* gal_n, int gmax, int dim, const float* feat, const float* feat_n, hipStream_t s);
// association on the device, k frames per launch (kernels_trk_dev.hip, trk_dev.hpp)
struct DevTrkHdr; struct DevTrack; struct TrkDevParams; struct EpochDets; struct EpochScratch; struct EpochOut;
void launch_trk_epoch_prep(const DevTrkHdr* hdr, const DevTrack* trk, const float* gal_n, int gmax, int dim, int cap, const float* featn,
                           int dn, int dn_pad, int k, float* sm, float* gram, hipStream_t s);
void launch_trk_epoch(DevTrkHdr* hdr, DevTrack* trk, int* free_slots, float* mean, float* cov, float* gal_raw, float* gal_n,
                      const TrkDevParams& prm, const EpochDets& dets, int f0, int k, int d_begin, int dn_pad, int nmax, int has_sm,
                      const EpochScratch& scr, const EpochOut& out, hipStream_t s);
void launch_gallery_shard(const DevTrkHdr* hdr, const DevTrack* trk, const float* gal_n, int gmax, int dim, float* out, int t_max, hipStream_t s);
// configs[4] annotation pass: gathered [world, t_max, 2 + dim] -> per row (all ranks) track id or -1, nearest valid row of another rank or -1, its cosine distance
void launch_gallery_nearest(const float* gathered, int world, int t_max, int dim, int* ids, int* near_row, float* near_dist, hipStream_t s);
void launch_trk_cascade_test(const TrkDevParams& prm, const EpochScratch& scr, int T, int n, const int* state, const int* tsu,
                             int* out_mdet, int* out_err, int stage1_only, hipStream_t s);


}  // namespace aic
