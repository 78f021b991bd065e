// engine.hpp -- executor of an .aicw engine file (graph IR + weights) on one GPU.
#pragma once
#include "kernels.hpp"

#include <algorithm>
#include <memory>
#include <vector>

namespace aic {

enum { OP_CONV = 1, OP_SPPF_POOL = 2, OP_UPSAMPLE2X = 3, OP_MAXPOOL3S2 = 4, OP_AVGPOOL = 5, OP_L2NORM = 6 };
enum { KIND_YOLO = 1, KIND_REID = 2 };

struct BufDesc { int h, w, c, f32; void* p; size_t per_item; int esize; };
struct ConvWeights { DevBuf<char> w; DevBuf<float> bias; int cout, cin, cin_eff, kh, kw, K, Kp, cout_pad; int cin2 = 0; };   // cin2: K columns of a folded 1x1 second source behind the window's
// fuse: 1 = conv fused with the max-pool that follows, 2 = skipped (absorbed).  v[16..19] (0 in engine files) are set at load time for a
// conv that took a 1x1 conv of another tensor in as a second source (ConvArgs::x2): v[16] = that tensor's buffer + 1, v[17] = its channel
// offset, v[18] = its channels, v[19] = the 1x1's stride
// xs_*: set at load time for a 1x1 conv whose first xs_c input channels were a 2x upsample's output slice (ConvArgs::xs): that op's source
struct OpDesc { int v[20]; int fuse = 0; int xs_buf = -1, xs_coff = 0, xs_c = 0; };
struct OutDesc { int v[8]; };

struct Model {
    Device* dev = nullptr;
    int kind = 0, in_h = 0, in_w = 0, dtype = AIC_F16, max_items = 0;
    int meta[8] = {0};
    std::vector<BufDesc> bufs;
    std::vector<DevBuf<char>> storage;
    std::vector<OpDesc> ops;
    std::vector<ConvWeights> weights;
    std::vector<OutDesc> outs;
    double flops_per_item = 0;
    int n_convs = 0, n_anchors = 0, out_dim = 0;

    // detector workspace (YOLO)
    int max_det_cap = 0;
    DevBuf<float> d_boxes, d_maxlogit, d_out_boxes, d_out_boxes_orig, d_out_scores;
    DevBuf<int> d_labels, d_ncand, d_numdets, d_out_labels;
    // staging
    DevBuf<float> d_in_f32;
    DevBuf<uint8_t> d_frames;
    DevBuf<float> d_crop_boxes;
    DevBuf<int> d_valid;
    DevBuf<char> d_zero;   // zero page for the conv kernel's LDS-DMA

    // the engine file's bytes, kept: the pipeline builds a second instance of the engine (own activation arena, own detector workspace)
    // for its second lane (pipeline.cpp: small launch groups of two chunk contexts run side by side)
    std::shared_ptr<const std::vector<char>> blob_copy;
    Model(Device& d, const void* blob, size_t nbytes, int dtype_, int max_items_);
    void* input() { return bufs[0].p; }
    // fp16 ReID engines with the fused stem take NHWC4 crops (8 bytes per pixel: half the crop traffic); whoever fills
    // input() says which layout it wrote
    bool in_pix4 = false;
    // set around run(): the fused stem takes its crops straight from these frames / boxes (pipeline; no crop tensor in HBM)
    CropSrc crop_src{nullptr, 0, 0, nullptr, nullptr, nullptr};
    // set around run(): the number of items is only known on the device (ReID behind the on-device detection filter): run(n) then
    // sizes every launch for the bound n and the kernels leave past n_items_dev[0] (ConvArgs::n_dev)
    const int* n_items_dev = nullptr;
    // set before run() by callers that go on to decode (aic_detect, aic_yolo_infer, aic_yolo_decode, the pipeline) and never look at the
    // raw head logits: the detect branches' tails then store max logit + label (class branch, ConvArgs::t_max) and the decoded box (box
    // branch, ConvArgs::t_box) per anchor themselves, and decode_kernel skips what the levels named in cls_reduced / box_decoded
    // already have (bit l = level l of the LAST run).  aic_yolo_head leaves it off.
    bool reduce_cls = false;
    unsigned cls_reduced = 0, box_decoded = 0;
    bool input_pix4_ok() const;
    // The largest launch (<= max_items) every activation tensor of which stays below 2^31 elements: the patch / block / space-to-depth conv
    // kernels address with 32-bit offsets and hand a larger tensor to the general kernels (conv_try_c64_block & co. check M * cs < 2^31).
    // A caller that is free to choose its launch size (the pipeline's bounded ReID round) stays at or below this: 16 384 crops of the
    // ReID engine put layer1's 64 x 32 x 64-channel tensors at exactly 2^31 and cost 3.8 ms per launch group on the slower kernels.
    int fast_items() const {
        size_t per = 1;
        for (const OpDesc& o : ops) {          // tensors a conv kernel of its own reads or writes (a fused stem's pre-pool output never exists)
            if (o.v[0] != OP_CONV || o.fuse != 0) continue;
            for (int bi : {o.v[1], o.v[4]}) per = std::max(per, bufs[bi].per_item / (size_t)std::max(bufs[bi].esize, 1));
        }
        const size_t n = std::min<size_t>((size_t)max_items, ((1ull << 31) - 1) / per);
        return (int)(n > 64 && n < (size_t)max_items ? n / 64 * 64 : n);          // a round count: whole tiles in every layer
    }
    void run(int n_items, hipStream_t s);
    // u8 BGR frames -> letterbox -> the whole graph; fp16 YOLO engines fuse the letterbox into the stem conv
    void run_frames(const uint8_t* frames, int n, const LetterboxGeom& g, hipStream_t s);
    void run_range(size_t op0, size_t op1, int i0, int n, hipStream_t s);
    // Launches of a FEW frames (the per-frame plugin loop: one 640 x 640 image is 6 - 150 workgroups per layer on 256 CUs, ~10 us per dependent
    // launch): the detect branches of every level but the last leave the main stream -- level l's ops (those from which only output l is
    // reachable) run on a side stream from the moment their feature map exists, beside the rest of the neck, and the main stream joins them
    // before decode.  Planned at load time from the op table's read / write slices (plan_side_heads); same kernels, same arguments, same bits.
    struct SideHead { size_t a, b, cut; };      // ops [a, b) on the side stream, forked once the main stream has issued every op below `cut`
    std::vector<SideHead> side_heads;
    hipStream_t side_stream[2] = {nullptr, nullptr};
    hipEvent_t side_fork[2] = {nullptr, nullptr}, side_join[2] = {nullptr, nullptr};
    bool side_ok = false;                       // set around run() by the per-call entry points (aic_detect, aic_yolo_*): the pipeline keeps its streams to itself
    int side_max_items = 2;                     // aic_model_option-free: AICAM_SIDE_HEADS=0 turns the schedule off, =N moves the item bound
    void plan_side_heads();
    void run_ops(size_t op0, int n, hipStream_t s);     // ops [op0, end) of the whole batch: the side-head schedule where it applies, else run_range
    ~Model();
    size_t lead_ops = 0;   // leading ops whose activations are large: run in sub-batches (Infinity-Cache residency)
    int sub_items = 0;
    // YOLO post-processing on the buffers left by run()
    DetArgs det_args(int batch, float conf, float iou, int max_det, const LetterboxGeom* g);
    // host_out: the kept detections (counts, original-pixel boxes, scores, labels) are stored straight into this page-locked host block by
    // the NMS kernel -- [batch] int32 | [batch, max_det, 4] f32 | [batch, max_det] f32 | [batch, max_det] int32 -- instead of the device
    // arrays: aic_detect's per-frame calls then end in one stream sync and no copy (four pageable D2H copies cost ~60 us of a 1.65 ms frame)
    void decode_nms(int batch, float conf, float iou, int max_det, const LetterboxGeom* g, hipStream_t s, char* host_out = nullptr);
    PinBuf<char> h_det;
    // set around run() by aic_reid_embed's per-frame calls: the graph's LAST op (the embedding's L2 normalisation) stores into this
    // page-locked host block instead of its arena buffer -- no copy behind the run, one stream sync
    float* emb_host_out = nullptr;
    PinBuf<char> h_emb;
    const float* embeddings() const { return reinterpret_cast<const float*>(bufs[outs[0].v[0]].p); }
};

}  // namespace aic

struct aic_model { aic::Model m; aic_model(aic::Device& d, const void* b, size_t n, int dt, int mi) : m(d, b, n, dt, mi) {} };
