"""GPU parity of the engines against the fp32 PyTorch-CPU oracle on the same engine file:
fp32 mode (v_mfma_f32_16x16x4_f32) is the parity gate (north_star: boxes / embeddings within
1e-3), fp16 mode reports its deviation against documented bounds; decode and NMS integer outcomes
are compared exactly on identical head tensors."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import image_oracle as I
from oracle import nets_oracle as N

pytestmark = pytest.mark.gpu
ef = pkg("engine_file")
syn = pkg("synthetic")
HipEngine = pkg("hip_engine").HipEngine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def frames():
    sc = syn.Scene(seed=0)
    return np.stack([sc.render(f) for f in (0, 40)])


@pytest.fixture(scope="module")
def yolo_ref(engines, frames):
    eo = N.EngineOracle(engines[0])
    x = np.concatenate([I.preprocess_yolo_input(f)[0] for f in frames])
    torch.set_num_threads(8)
    dfl, cls = eo.yolo_head(torch.from_numpy(x))
    return eo, x, dfl.numpy(), cls.numpy()


@pytest.fixture(scope="module")
def yolo_ref64(engines, yolo_ref):
    """The same engine file evaluated in fp64 (torch CPU double): the anchor for the fp32 box bound."""
    eo64 = N.EngineOracle(engines[0], dtype=torch.float64)
    d64, c64 = (t.numpy() for t in eo64.yolo_head(torch.from_numpy(yolo_ref[1])))
    return d64, c64, eo64.decode(d64, c64, ft=np.float64)[0]


def test_small_graph_every_op_fp32(gpu, tmp_path):
    """A small engine that exercises every op kind, tile variant and slice/residual mode."""
    g = ef.Graph(ef.KIND_REID, 24, 16)
    wg = ef._WeightGen(3)
    def conv(name, src, dst, cin, cout, k, s, act, **kw):
        g.conv(name, src, dst, cin, cout, k, s, act, wb=wg(cout, cin, k, act), **kw)
    inp = g.buf(24, 16, 8)
    a = g.buf(24, 16, 48); conv("c0", inp, a, 3, 48, 3, 1, ef.ACT_SILU)             # stem (cin 3), 48-ch tile
    b = g.buf(12, 8, 80); conv("c1", a, b, 48, 80, 3, 2, ef.ACT_RELU)               # stride 2, 80-ch tile
    cat = g.buf(12, 8, 80 + 32 + 32)
    conv("c2", b, cat, 80, 32, 1, 1, ef.ACT_SILU, dst_coff=80)                       # 1x1 into a slice, 32-ch tile
    conv("c3", cat, cat, 32, 32, 3, 1, ef.ACT_SILU, src_coff=80, dst_coff=112, res=(cat, 80), res_mode=ef.RES_ACT_THEN_ADD)
    g.simple(ef.OP_UPSAMPLE2X, cat, a, 32, src_coff=112, dst_coff=0)                  # overwrite 32 channels of a
    c = g.buf(24, 16, 16); conv("c4", a, c, 48, 16, 3, 1, ef.ACT_NONE)              # 16-ch tile
    d = g.buf(24, 16, 4 * 16); conv("c5", c, d, 16, 16, 1, 1, ef.ACT_SILU)
    g.simple(ef.OP_SPPF_POOL, d, d, 16, dst_coff=16)
    e = g.buf(12, 8, 64); g.simple(ef.OP_MAXPOOL3S2, d, e, 64)
    f = g.buf(12, 8, 128); conv("c6", e, f, 64, 128, 3, 1, ef.ACT_RELU)
    h = g.buf(12, 8, 128); conv("c7", f, h, 128, 128, 3, 1, ef.ACT_RELU, res=(f, 0), res_mode=ef.RES_ADD_THEN_ACT)
    p = g.buf(1, 1, 128); g.simple(ef.OP_AVGPOOL, h, p, 128)
    q = g.buf(1, 1, 64); conv("fc", p, q, 128, 64, 1, 1, ef.ACT_NONE)
    emb = g.buf(1, 1, 64, ef.DT_F32); g.simple(ef.OP_L2NORM, q, emb, 64)
    g.outputs.append([emb, 64, 0, 0, 0, 0, 0, 0]); g.meta = [64, 0, 0, 0, 0, 0, 0, 0]
    path = str(tmp_path / "small.aicw")
    ef.write_engine(path, g)
    x = np.random.default_rng(0).standard_normal((37, 3, 24, 16)).astype(np.float32)
    ref = N.EngineOracle(path).run(torch.from_numpy(x))[emb][:, :, 0, 0].numpy()
    for dtype, tol in (("fp32", 2e-5), ("fp16", 2e-2)):
        eng = HipEngine(path, dtype=dtype, max_items=64, warm_up=False)
        out = eng.reid_infer_np(x)
        assert out.shape == ref.shape and np.isfinite(out).all()
        assert np.abs(out - ref).max() < tol, (dtype, np.abs(out - ref).max())
        eng.close()


# fp32 (north_star: box coords within 1e-3): two fp32 evaluations of a 63-conv net differ by their summation order (K <= 2304
# per conv: MFMA 16x16x4 chains here, MKL-DNN blocking in torch); the DFL expectation times the stride (<= 32) amplifies a
# ~4e-5 logit difference to a few 1e-3 px on the stride-32 level.  Neither side is "the" fp32 result, so both are measured
# against the fp64 evaluation of the same engine file: the HIP boxes must be within 1e-3 px of it (the fp32 MFMA path sums
# each K-step of 16 products from zero and adds the partial sum to the accumulator -- two-level summation, conv_common.hpp --
# which lands it at 6.7e-4 px, closer than torch's own fp32 evaluation at 1.5e-3 px).  fp16 gates: ~2x the measured
# deviation (logits 2.5e-2, boxes 1.7 px).
@pytest.mark.parametrize("dtype,tol_logit,tol_box", [("fp32", 2e-4, None), ("fp16", 0.06, 4.0)])
def test_yolo_head_and_decode(gpu, engines, yolo_ref, yolo_ref64, dtype, tol_logit, tol_box):
    eo, x, dfl_ref, cls_ref = yolo_ref
    eng = HipEngine(engines[0], dtype=dtype, max_items=2, warm_up=False)
    assert (eng.n_anchors, eng.out_dim, eng.n_convs) == (8400, 80, 63) and abs(eng.flops_per_item - 8.742912e9) < 1e3
    dfl, cls = eng.yolo_head_np(x)
    e_d, e_c = np.abs(dfl - dfl_ref).max(), np.abs(cls - cls_ref).max()
    print(f"[{dtype}] max |dfl logit err| {e_d:.2e}  max |cls logit err| {e_c:.2e}")
    assert e_d < tol_logit and e_c < tol_logit
    boxes, ml, lab = eng.yolo_decode_np(x)
    rb, rml, rlab = eo.decode(dfl_ref, cls_ref)
    e_b = np.abs(boxes - rb).max()
    print(f"[{dtype}] max |box err| {e_b:.2e} px")
    assert np.abs(ml - rml).max() < tol_logit
    if dtype == "fp32":
        d64, c64, b64 = yolo_ref64
        e_hip, e_cpu = np.abs(boxes - b64).max(), np.abs(rb - b64).max()
        l_hip, l_cpu = max(np.abs(dfl - d64).max(), np.abs(cls - c64).max()), max(np.abs(dfl_ref - d64).max(), np.abs(cls_ref - c64).max())
        small = np.abs(boxes - b64)[:, eo.anchors()[1] <= 16].max()
        print(f"[fp32] vs fp64 evaluation: box err HIP {e_hip:.2e} px / torch-CPU fp32 {e_cpu:.2e} px (strides 8+16 only: HIP {small:.2e}); "
              f"logit err HIP {l_hip:.2e} / torch {l_cpu:.2e}")
        assert e_hip <= 1e-3                        # north_star: box coords within 1e-3 (measured 6.7e-4 px; torch-CPU fp32: 1.5e-3)
        assert l_hip <= max(5e-5, l_cpu)
    else:
        assert e_b < tol_box
    if dtype == "fp32":
        gap = np.sort(cls_ref, -1)[..., -1] - np.sort(cls_ref, -1)[..., -2]
        assert (lab == rlab)[gap > 1e-3].all()                 # labels identical away from arg-max near-ties
    # decode kernel on ITS OWN head tensor vs the oracle's decode of that same tensor: kernel-level parity
    kb, kml, klab = eo.decode(dfl, cls)
    assert np.abs(boxes - kb).max() < 2e-3 and np.array_equal(ml, kml) and np.array_equal(lab, klab)
    eng.close()


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_nms_identical_on_identical_decode(gpu, engines, yolo_ref, dtype):
    """Integer outcome of select+sort+NMS: feed the oracle NMS the GPU's own decoded boxes/logits; the
    kept set, its order and num_dets must be identical (the IoU arithmetic is bit-reproducible)."""
    eo, x, _, _ = yolo_ref
    eng = HipEngine(engines[0], dtype=dtype, max_items=2, warm_up=False)
    boxes, ml, lab = eng.yolo_decode_np(x)
    for conf, iou, md in ((0.3, 0.5, 300), (0.25, 0.45, 100), (0.5, 0.7, 300), (0.3, 0.5, 7)):
        nd, ob, osc, ol = eng.yolo_infer_np(x, conf=conf, iou=iou, max_det=md)
        for b in range(len(x)):
            keep, margin = N.nms(boxes[b], ml[b], lab[b], conf, iou, md, return_margin=True)
            n_cand = int((ml[b] >= N.logit_threshold(conf)).sum())
            print(f"[{dtype}] conf {conf} iou {iou} max_det {md}: candidates {n_cand}, kept {len(keep)}, near-tie margin {margin:.2e}")
            assert nd[b] == len(keep)
            assert np.array_equal(ob[b, :nd[b]], boxes[b][keep])
            assert np.array_equal(ol[b, :nd[b]], lab[b][keep])
            assert np.allclose(osc[b, :nd[b]], N.sigmoid32(ml[b][keep]), atol=2e-7)
            if conf == 0.3:
                assert n_cand > 200                      # the seeded head gives NMS real work
    eng.close()


def test_detector_plugin_vs_oracle_fp32(gpu, engines, frames):
    """YOLODetector.detect end to end (letterbox -> engine -> NMS -> scale_bboxes) vs the oracle chain."""
    det = pkg("detector").YOLODetector(engines[0], dtype="fp32")
    eo = N.EngineOracle(engines[0])
    for f in frames:
        boxes, scores, cids, idx = det.detect(f)
        x, ratios, pad = I.preprocess_yolo_input(f)
        dfl, cls = eo.yolo_head(torch.from_numpy(x))
        rb, rml, rlab = eo.decode(dfl.numpy(), cls.numpy())
        keep, margin = N.nms(rb[0], rml[0], rlab[0], 0.3, 0.5, 300, return_margin=True)
        ref_boxes = I.scale_bboxes(rb[0][keep], f.shape[:2], ratios, pad)
        assert boxes.dtype == np.float32 and cids.dtype == np.int32 and len(idx) == len(boxes)
        if margin > 5e-3:     # away from threshold near-ties the two chains keep exactly the same anchors
            assert len(boxes) == len(keep)
            assert np.array_equal(cids, rlab[0][keep])
            # two fp32 chains (see test_yolo_head_and_decode: each is within ~2e-3 letterbox px of the fp64 result), / ratio 0.5
            assert np.abs(boxes - ref_boxes).max() < 5e-3 / 0.5
            assert np.allclose(scores, N.sigmoid32(rml[0][keep]), atol=1e-4)
        else:
            assert abs(len(boxes) - len(keep)) <= 3
    e = det.detect(np.zeros((0, 0, 3), np.uint8))
    assert e[0].shape == (0, 4) and e[3].dtype == int          # bad frame -> empties, no raise (yolo_detector.py:113-126)


def test_fp16_frames_path_fused_stem(gpu, engines, frames):
    """fp16 detect from u8 frames runs the fused letterbox+stem kernel and the 16-channel direct kernels; the same
    engine fed the oracle's letterboxed NCHW tensor runs the letterbox-free generic stem. Same detections up to fp16
    rounding: every confident detection of one side has an IoU > 0.9 partner of the same class on the other."""
    eng = HipEngine(engines[0], dtype="fp16", max_items=2, warm_up=False)
    x = np.concatenate([I.preprocess_yolo_input(f)[0] for f in frames])
    nd0, b0, s0, l0 = eng.detect_np(frames)
    nd1, b1, s1, l1 = eng.yolo_infer_np(x)
    assert (nd0 > 0).all() and np.abs(nd0 - nd1).max() <= max(3, int(0.02 * nd0.max()))
    for i, f in enumerate(frames):
        _, ratios, pad = I.preprocess_yolo_input(f)
        ref = I.scale_bboxes(b1[i, :nd1[i]], f.shape[:2], ratios, pad)
        got = b0[i, :nd0[i]]
        iou = np.stack([N.box_iou_xyxy(g, ref) for g in got]) if len(got) and len(ref) else np.zeros((len(got), len(ref)), np.float32)
        same = l0[i, :nd0[i], None] == l1[i, None, :nd1[i]]
        best = np.where(same, iou, 0).max(1)
        strong = s0[i, :nd0[i]] > 0.35                               # away from the 0.3 threshold
        frac = (best[strong] > 0.9).mean()
        # the matched pairs' scores: both paths feed the same network the same letterboxed pixels (the fused stem resamples with the
        # letterbox kernel's arithmetic), so a pair differs by fp16 rounding in the stem's K order only -- a wrong halo or a shifted
        # tap in a fused kernel moves scores by 1e-2 and more while still passing the IoU criterion above
        partner = np.where(same, iou, 0).argmax(1)
        ok = strong & (best > 0.9)
        ds = np.abs(s0[i, :nd0[i]][ok] - s1[i, :nd1[i]][partner[ok]])
        print(f"frame {i}: {nd0[i]} vs {nd1[i]} detections, {frac:.4f} of the confident ones matched, max |score difference| of matched pairs {ds.max():.2e}")
        assert frac > 0.95       # max_det saturates on seeded heads: near-ties at the rank-300 cut swap a few boxes
        assert ds.max() < 6e-3   # measured 2.5e-3 (fp16 engines); a fused stem + 1.conv kernel with a bug somewhere in its halo: 1.7e-2
    eng.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("fp16", 5e-4)])     # measured 1.4e-7 / 9.8e-5 (north_star: 1e-3)
def test_reid_embeddings(gpu, engines, frames, dtype, tol):
    sc = syn.Scene(seed=0)
    boxes = sc.detections(0)[0]
    reid = pkg("reid_model").ReIDModel(engines[1], dtype=dtype, max_batch=64)
    assert reid.feature_dim == 512
    emb, valid = reid.embed_boxes(frames[0], boxes)
    crops, ovalid = I.crops_to_batch(frames[0], boxes)
    eo = N.EngineOracle(engines[1])
    ref = eo.run(torch.from_numpy(crops))[eo.outputs[0][0]][:, :, 0, 0].numpy()
    err = np.abs(emb - ref).max()
    cos = 1 - (emb * ref).sum(1)
    print(f"[{dtype}] ReID max |emb err| {err:.2e}, max cosine distance to oracle {cos.max():.2e}")
    assert valid.tolist() == ovalid.tolist() and err < tol
    assert np.allclose(np.linalg.norm(emb, axis=1), 1, atol=1e-3)
    # plugin surface (reid_model.py:128-236): list of crops, invalid ones skipped, empty -> (0, dim)
    crops_list = [frames[0][int(b[1]):int(b[3]), int(b[0]):int(b[2])] for b in boxes[:5]]
    out = reid.extract_features_batched(crops_list + [np.zeros((0, 5, 3), np.uint8), "x", np.zeros((4, 4), np.uint8)])
    assert out.shape == (5, 512) and np.abs(out - emb[:5]).max() < 1e-5
    assert reid.extract_features_batched([]).shape == (0, 512)
    assert reid.extract_features_batched([np.zeros((3, 3, 1), np.uint8)]).shape == (0, 512)


def test_embed_boxes_fused_crop_equals_crop_kernel_plus_stem(gpu, engines, frames):
    """aic_reid_embed resamples every crop INSIDE the stem kernel (fp16 engines, frame handed over in host memory); the crop kernel
    followed by the stem on its tensor must give the same bits -- boxes that are clipped, one pixel wide, empty, exactly 2x the crop
    (cv2's INTER_AREA branch) and the whole frame included."""
    ip = pkg("image_processing")
    sc = syn.Scene(seed=4, n_targets=30)
    frame = sc.render(3)
    extra = np.array([[-20.5, -3.2, 40.9, 90.1], [1270.2, 700.7, 1300, 760], [100, 100, 100.9, 180], [50, 60, 178, 316],
                      [0, 0, 1280, 720], [640.99, 10.01, 641.99, 11.5], [300, 200, 290, 260], [5, 5, 69, 133]], np.float32)
    boxes = np.concatenate([sc.detections(3)[0], extra])
    eng = HipEngine(engines[1], dtype="fp16", max_items=64, warm_up=False)
    emb, valid = eng.embed_boxes_np(frame, boxes)
    crops, cvalid = ip.crops_from_boxes(frame, boxes)
    ref = eng.reid_infer_np(crops)
    eng.close()
    assert valid.tolist() == cvalid.tolist() and 0 < valid.sum() < len(boxes)
    ok = valid.astype(bool)
    assert np.array_equal(emb[ok], ref[ok])


@pytest.mark.parametrize("n_crops", [960, 950])
@pytest.mark.parametrize("dtype,tol_split,tol", [("fp32", 2e-5, 1e-5), ("fp16", 1e-3, 5e-4)])
def test_reid_large_batch_kernels(gpu, engines, dtype, tol_split, tol, n_crops):
    """The big-tile conv kernels only engage at production batch sizes (ping-pong 256x256 / 512x128 tiles, the
    patch forms, one block per CU, the persistent weights-resident kernel): 960 crops in ONE launch group must give
    the embeddings of the same crops run in groups of 16 (small-tile kernels; differences = K summation order only)
    and match the fp32 oracle on a sample.  950 crops: ragged last tiles / a partial last trip of the persistent kernel."""
    rng = np.random.default_rng(7)
    sc = syn.Scene(seed=3)
    frame = sc.render(0)
    boxes = sc.detections(0)[0]
    crops, _ = I.crops_to_batch(frame, boxes)                       # [30, 3, 128, 64] fp32
    x = np.concatenate([crops] * 32)[:n_crops].copy()
    x += rng.standard_normal(x.shape).astype(np.float32) * 0.05     # 960 distinct inputs
    big = HipEngine(engines[1], dtype=dtype, max_items=n_crops, warm_up=False)
    small = HipEngine(engines[1], dtype=dtype, max_items=16, warm_up=False)
    e_big = big.reid_infer_np(x)
    idx = np.r_[0:16, 472:488, n_crops - 16:n_crops]
    e_small = np.concatenate([small.reid_infer_np(x[i:i + 16]) for i in (0, 472, n_crops - 16)])
    d = np.abs(e_big[idx] - e_small).max()
    eo = N.EngineOracle(engines[1])
    torch.set_num_threads(8)
    ref = eo.run(torch.from_numpy(x[idx[:8]]))[eo.outputs[0][0]][:, :, 0, 0].numpy()
    err = np.abs(e_big[idx[:8]] - ref).max()
    print(f"[{dtype}] {n_crops}-crop group vs 16-crop groups {d:.2e}; vs oracle {err:.2e}")
    assert d < tol_split and err < tol
    assert np.allclose(np.linalg.norm(e_big, axis=1), 1, atol=1e-3)
    big.close(), small.close()


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_reid_embeddings_do_not_depend_on_the_batch(gpu, engines, dtype):
    """A crop's embedding must not depend on how many crops shared its launch: kernel variants are chosen by batch size (ping-pong
    patch kernels at 832 and 256, the LDS-DMA implicit GEMMs below), and every variant a layer can get walks K in the same order --
    that of the ping-pong patch kernel for the shapes it takes, that of the weights-resident 64-channel kernels for theirs (ConvArgs::k_order), memory order for the rest.
    Bit-identical rows, not a tolerance: a track's gallery must not change with the group size its frames were batched in."""
    x = np.random.default_rng(3).standard_normal((832, 3, 128, 64)).astype(np.float32)
    big = HipEngine(engines[1], dtype=dtype, max_items=832, warm_up=False)
    e_big = big.reid_infer_np(x)
    big.close()
    for n in (256, 64, 30, 8):           # 30 and 8: the wide-step kernel, layer1 in the weights-resident kernels' K order on 256- and 128-pixel tiles
        eng = HipEngine(engines[1], dtype=dtype, max_items=n, warm_up=False)
        e = eng.reid_infer_np(x[:n])
        eng.close()
        assert np.array_equal(e, e_big[:n]), (dtype, n, float(np.abs(e - e_big[:n]).max()))


@pytest.mark.parametrize("env", [{"AICAM_C64_BLOCK": "0"}, {"AICAM_PP_MIN": "0"}], ids=["unfused_layer1", "pp_everywhere"])
def test_reid_large_batch_kernel_switches(gpu, engines, env):
    """Kernel choices that are read once per process, hence a child process: with the fused BasicBlock kernel off layer1 runs on
    the persistent weights-resident kernel (with and without residual); with AICAM_PP_MIN=0 the ping-pong kernels take every
    layer they apply to, including layer4's 8 x 4 maps (four images per 16-pixel MFMA tile).  Same check as above, 950 crops."""
    import subprocess
    import sys
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
he = importlib.import_module("ai-camera_amd.hip_engine")
rng = np.random.default_rng(7)
x = rng.standard_normal((950, 3, 128, 64)).astype(np.float32)
big = he.HipEngine(%r, dtype="fp16", max_items=950, warm_up=False)
small = he.HipEngine(%r, dtype="fp16", max_items=16, warm_up=False)
e_big = big.reid_infer_np(x)
d = max(np.abs(e_big[i:i + 16] - small.reid_infer_np(x[i:i + 16])).max() for i in (0, 472, 934))
print("DIFF", d)
sys.exit(0 if d < 3e-3 and np.allclose(np.linalg.norm(e_big, axis=1), 1, atol=1e-3) else 1)
""" % (ROOT, engines[1], engines[1])
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    print(r.stdout[-300:], r.stderr[-300:])
    assert r.returncode == 0


@pytest.mark.parametrize("scale", ["n", "m"])
def test_detect_heads_on_side_streams_bit_exact(gpu, engines, tmp_path, scale):
    """Launches of one or two frames (the per-frame plugin loop): the detect branches of every level but the last run on side streams from
    the moment their feature map exists, beside the rest of the neck (Model::plan_side_heads / run_ops, csrc/engine.cpp).  Same kernels,
    same arguments: decoded boxes, max logits, labels and the NMS output of twelve different images, one and two per call, must be
    IDENTICAL to the one-stream order (a child process with AICAM_SIDE_HEADS=0: read once per process), and the plan must exist --
    two side heads for a three-level YOLOv8 (printed under AICAM_SIDE_DBG)."""
    import subprocess
    import sys
    ef = pkg("engine_file")
    epath = engines[0] if scale == "n" else ef.ensure_seeded_engines(ROOT, scale="m")[0]
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
he = importlib.import_module("ai-camera_amd.hip_engine")
rng = np.random.default_rng(11)
out = {}
eng = he.HipEngine(%r, dtype="fp16", max_items=2, warm_up=False)
for it in range(12):
    n = 1 + it %% 2
    x = rng.random((n, 3, 640, 640), dtype=np.float32)
    b, ml, lab = eng.yolo_decode_np(x)
    nd, ob, sc, ol = eng.yolo_infer_np(x, conf=0.25, iou=0.45, max_det=300)
    out.update({f"b{it}": b, f"ml{it}": ml, f"lab{it}": lab, f"nd{it}": nd, f"ob{it}": ob, f"sc{it}": sc, f"ol{it}": ol})
eng.close()
np.savez(sys.argv[1], **out)
""" % (ROOT, epath)
    files = []
    for name, env in (("side", {"AICAM_SIDE_DBG": "1"}), ("one_stream", {"AICAM_SIDE_HEADS": "0", "AICAM_SIDE_DBG": "1"})):
        f = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        print(r.stdout[-300:], r.stderr[-600:])
        assert r.returncode == 0
        assert r.stderr.count("side head:") == (2 if name == "side" else 0), r.stderr[-600:]
        files.append(np.load(f))
    a, b = files
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert int(sum(a[f"nd{it}"].sum() for it in range(12))) > 0


@pytest.mark.parametrize("scale", ["n", "x"])
def test_detect_branch_tails_decode_in_place_bit_exact(gpu, engines, tmp_path, scale):
    """fp16 engines, calls that go on to decode: the class branch's 1x1 tail stores max logit + first arg-max per anchor and the box
    branch's tail the decoded box (ConvArgs::t_max / t_box) instead of 80 + 64 fp32 logits that decode_kernel would read back.  Same
    fp32 values, same order of operations: boxes, max logits and labels must be IDENTICAL to the two-step form (a child process with the
    switches off: they are read once per process), at 2 frames (128-pixel tail tiles) and at 40 (512-pixel tiles, the patch kernel's tail).
    Scale 'x' (ADVICE r4): its box branch LEADS with 80 channels (cb = max(16, 320 / 4, 64)) into the 64-output 1x1 -- the five-tile tail,
    which has no in-place box decode; the engine must leave that level to decode_kernel instead of marking it decoded (a 320 x 320
    engine keeps the 68 M-parameter net cheap)."""
    import subprocess
    import sys
    epath, hw = engines[0], 640
    if scale == "x":
        epath, hw = str(tmp_path / "yolov8x_320.aicw"), 320
        ef = pkg("engine_file")
        ef.write_engine(epath, ef.build_yolov8("x", in_hw=(320, 320), seed=3))
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
he = importlib.import_module("ai-camera_amd.hip_engine")
rng = np.random.default_rng(5)
out = {}
HW = %d
for n in (2, 40):
    x = rng.random((n, 3, HW, HW), dtype=np.float32)
    eng = he.HipEngine(%r, dtype="fp16", max_items=n, warm_up=False)
    b, ml, lab = eng.yolo_decode_np(x)
    nd, ob, sc, ol = eng.yolo_infer_np(x, conf=0.25, iou=0.45, max_det=300)
    eng.close()
    out.update({f"b{n}": b, f"ml{n}": ml, f"lab{n}": lab, f"nd{n}": nd, f"ob{n}": ob, f"sc{n}": sc, f"ol{n}": ol})
np.savez(sys.argv[1], **out)
""" % (ROOT, hw, epath)
    files = []
    for name, env in (("tails", {}), ("two_step", {"AICAM_NO_CLS_REDUCE": "1", "AICAM_NO_BOX_DECODE": "1"})):
        f = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        print(r.stdout[-300:], r.stderr[-300:])
        assert r.returncode == 0
        files.append(np.load(f))
    a, b = files
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert a["nd40"].min() > 0 and len(np.unique(a["lab40"])) > 10          # the check is not vacuous


def test_stride2_direct_kernel_with_tail_bit_exact(gpu, engines, tmp_path):
    """YOLOv8n's `3.conv` + `4.c2f.cv1` at large batch: conv3x3_c32s2_tail_kernel (3x3 / 2, 32 -> 64 channels, the input patch read once into
    LDS, the 1x1 in its epilogue) against the LDS-DMA implicit GEMM with the same tail (AICAM_NO_C32S2=1, a child process: read once per
    process); `4.c2f.cv2` and `15.c2f.cv2` likewise: conv1x1_stream_kernel (weights in registers, pixels straight from memory, no LDS)
    against the implicit GEMM (AICAM_NO_1X1_STREAM=1); `22.cls0.0` on the patch kernel's 80-channel form and `22.cls0.1` + `.2` on
    conv3x3_c80_patch_tail_kernel (the patch form for ten chunks per pixel, K-steps that straddle taps) against the implicit GEMM's
    tiles (AICAM_NO_PATCH_C80=1: the same switch takes every conv3x3_pm_patch_kernel form off -- the box / class tails, the 40 x 8 strips of the
    40 x 40 level's bottleneck convs and the merged 128 -> 144 head conv `22.box1.0` + `22.cls1.0`); the 80 x 80 level's 32 -> 32 bottleneck convs on
    conv3x3_patch_kernel against the implicit GEMM (AICAM_NO_PATCH_C32=1).  Same products in the same order, same roundings: the raw head of 32 frames (every one of these kernels
    engages at that size) must be IDENTICAL, and it must be a real head."""
    import subprocess
    import sys
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
he = importlib.import_module("ai-camera_amd.hip_engine")
x = np.random.default_rng(13).uniform(0, 1, (32, 3, 640, 640)).astype(np.float32)
eng = he.HipEngine(%r, dtype="fp16", max_items=32, warm_up=False)
dfl, cls = eng.yolo_head_np(x)
eng.close()
np.savez(sys.argv[1], dfl=dfl, cls=cls)
""" % (ROOT, engines[0])
    files = []
    for name, env in (("direct", {}), ("igemm", {"AICAM_NO_C32S2": "1", "AICAM_NO_1X1_STREAM": "1", "AICAM_NO_PATCH_C80": "1", "AICAM_NO_PATCH_C32": "1"})):
        f = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        print(r.stdout[-300:], r.stderr[-300:])
        assert r.returncode == 0
        files.append(np.load(f))
    a, b = files
    assert np.array_equal(a["dfl"], b["dfl"]) and np.array_equal(a["cls"], b["cls"])
    assert np.isfinite(a["dfl"]).all() and a["cls"].std() > 0.05 and a["dfl"].std() > 0.05


@pytest.mark.parametrize("dtype,tol", [("fp32", 5e-4), ("fp16", 6e-2)])
def test_yolo_large_batch_kernels(gpu, engines, dtype, tol):
    """The detector's big-tile kernels (512 x 80 class-branch tiles, 16-channel direct kernel, patch forms) engage only
    when a launch group has tiles for every CU: the raw head of 48 frames in ONE group must equal the head of the same
    frames run 4 at a time (small tiles; differences = K summation order and fp16 rounding of intermediate tensors)."""
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 1, (48, 3, 640, 640)).astype(np.float32)
    big = HipEngine(engines[0], dtype=dtype, max_items=48, warm_up=False)
    small = HipEngine(engines[0], dtype=dtype, max_items=4, warm_up=False)
    dfl_b, cls_b = big.yolo_head_np(x)
    worst = 0.0
    for i in (0, 20, 44):
        dfl_s, cls_s = small.yolo_head_np(x[i:i + 4])
        worst = max(worst, np.abs(dfl_b[i:i + 4] - dfl_s).max(), np.abs(cls_b[i:i + 4] - cls_s).max())
    print(f"[{dtype}] 48-frame group vs 4-frame groups: max |diff| of the raw head {worst:.2e}")
    assert worst < tol and np.isfinite(dfl_b).all() and np.isfinite(cls_b).all()
    assert cls_b.std() > 0.05 and dfl_b.std() > 0.05          # a real head, not zeros
    big.close(), small.close()


def test_hip_engine_trt_surface(gpu, engines):
    """TRTEngine-compatible dict API on torch tensors (trt_engine.py:151-216)."""
    eng = HipEngine(engines[0], dtype="fp32", max_items=2)
    assert [i.name for i in eng.get_input_details()] == ["images"]
    assert [o.name for o in eng.get_output_details()] == ["num_dets", "bboxes", "scores", "labels"]
    with pytest.raises(TypeError):
        eng(torch.zeros(1, 3, 640, 640))
    with pytest.raises(ValueError):
        eng.infer({"wrong": torch.zeros(1, 3, 640, 640)})
    x = torch.rand(1, 3, 640, 640)
    out = eng({"images": x.double()})                     # wrong dtype/device: warned, cast, moved
    assert out["num_dets"].shape == (1, 1) and out["bboxes"].shape == (1, 300, 4) and out["bboxes"].is_cuda
    nd, ob, _, _ = eng.yolo_infer_np(x.numpy())
    assert int(out["num_dets"][0, 0]) == nd[0] and np.array_equal(out["bboxes"][0].cpu().numpy(), ob[0])
    r = HipEngine(engines[1], dtype="fp32", max_items=16)
    o = r({"input": torch.randn(3, 3, 128, 64)})["output"]
    assert o.shape == (3, 512) and torch.allclose(o.norm(dim=1), torch.ones(3, device=o.device), atol=1e-4)
    with pytest.raises(RuntimeError):
        r({"input": torch.randn(17, 3, 128, 64)})         # beyond engine capacity -> loud, not silent


def _c2f_graph(path):
    """stem conv 3 -> 32, then a C2f wired exactly like engine_file.build_yolov8's layer 2 (cv1 32->32, m.cv1 / m.cv2 3x3 16->16 with
    shortcut, cv2 48->32) on a 24 x 64 map (3 x 2 tiles of the fused kernel: every border and corner case), pooled to an embedding."""
    g = ef.Graph(ef.KIND_REID, 24, 64)
    wg = ef._WeightGen(5)
    def conv(name, src, dst, cin, cout, k, s, act, **kw):
        g.conv(name, src, dst, cin, cout, k, s, act, wb=wg(cout, cin, k, act), **kw)
    inp = g.buf(24, 64, 8)
    b1 = g.buf(24, 64, 32); conv("stem", inp, b1, 3, 32, 3, 1, ef.ACT_SILU)
    cat = g.buf(24, 64, 48); tmp = g.buf(24, 64, 16); out = g.buf(24, 64, 32)
    conv("c2f.cv1", b1, cat, 32, 32, 1, 1, ef.ACT_SILU)
    conv("c2f.m0.cv1", cat, tmp, 16, 16, 3, 1, ef.ACT_SILU, src_coff=16)
    conv("c2f.m0.cv2", tmp, cat, 16, 16, 3, 1, ef.ACT_SILU, dst_coff=32, res=(cat, 16), res_mode=ef.RES_ACT_THEN_ADD)
    conv("c2f.cv2", cat, out, 48, 32, 1, 1, ef.ACT_SILU)
    # a strided "probe": keep spatial structure in the output (an average pool would hide a wrong border ring)
    probe = g.buf(12, 32, 16); conv("probe", out, probe, 32, 16, 3, 2, ef.ACT_NONE)
    p = g.buf(1, 1, 16); g.simple(ef.OP_AVGPOOL, probe, p, 16)
    q = g.buf(1, 1, 64); conv("fc", p, q, 16, 64, 1, 1, ef.ACT_NONE)
    emb = g.buf(1, 1, 64, ef.DT_F32); g.simple(ef.OP_L2NORM, q, emb, 64)
    g.outputs.append([emb, 64, 0, 0, 0, 0, 0, 0]); g.meta = [64, 0, 0, 0, 0, 0, 0, 0]
    ef.write_engine(path, g)
    return out, probe


def test_fused_c2f_block(gpu, tmp_path):
    """The one-kernel C2f (csrc/kernels_conv_c2f.hip) against the fp32 oracle on a map where every tile touches a border, and
    against the four-launch form of the same engine (child process, AICAM_NO_C2F=1): differences = fp16 rounding only."""
    import subprocess
    import sys
    path = str(tmp_path / "c2f.aicw")
    _c2f_graph(path)
    x = np.random.default_rng(2).standard_normal((5, 3, 24, 64)).astype(np.float32)
    eo = N.EngineOracle(path)
    ref = eo.run(torch.from_numpy(x))[eo.outputs[0][0]][:, :, 0, 0].numpy()
    eng = HipEngine(path, dtype="fp16", max_items=8, warm_up=False)
    got = eng.reid_infer_np(x)
    eng.close()
    err = np.abs(got - ref).max()
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
he = importlib.import_module("ai-camera_amd.hip_engine")
x = np.load(%r)
np.save(%r, he.HipEngine(%r, dtype="fp16", max_items=8, warm_up=False).reid_infer_np(x))
""" % (ROOT, str(tmp_path / "x.npy"), str(tmp_path / "unfused.npy"), path)
    np.save(tmp_path / "x.npy", x)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AICAM_NO_C2F="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    unf = np.load(tmp_path / "unfused.npy")
    print(f"fused C2f: err vs fp32 oracle {err:.2e} (four-launch form: {np.abs(unf - ref).max():.2e}); fused vs four launches {np.abs(got - unf).max():.2e}")
    assert err < 5e-3 and np.abs(got - unf).max() < 2e-3


_TAIL_CHILD = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
he = importlib.import_module("ai-camera_amd.hip_engine")
L = importlib.import_module("ai-camera_amd._lib")
x = np.load(%r)
eng = he.HipEngine(%r, dtype="fp16", max_items=%d, warm_up=False)
L.call("aic_prof_reset", 0)
L.call("aic_prof_enable", 0, 1)
out = eng.%s(x)
n = L.prof_read(0)["conv_igemm"]["launches"]
L.call("aic_prof_enable", 0, 0)
np.savez(%r, n=n, **{"o%%d" %% i: o for i, o in enumerate(out if isinstance(out, tuple) else (out,))})
"""


def _run_with_and_without_tail(tmp_path, path, x, items, method, child_env=None):
    """-> (outputs, conv launches) of this process (1x1 tails fused) and of a child with AICAM_NO_TAIL=1 (every conv on its own),
    or with `child_env` instead (another kernel switch that is read once per process)."""
    import subprocess
    import sys
    L = pkg("_lib")
    eng = HipEngine(path, dtype="fp16", max_items=items, warm_up=False)
    L.call("aic_prof_reset", 0)                       # counters are cumulative per process
    L.call("aic_prof_enable", 0, 1)
    out = getattr(eng, method)(x)
    n_fused = L.prof_read(0)["conv_igemm"]["launches"]
    L.call("aic_prof_enable", 0, 0)
    eng.close()
    out = out if isinstance(out, tuple) else (out,)
    np.save(tmp_path / "x.npy", x)
    code = _TAIL_CHILD % (ROOT, str(tmp_path / "x.npy"), path, items, method, str(tmp_path / "unfused.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **(child_env or {"AICAM_NO_TAIL": "1"})), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    z = np.load(tmp_path / "unfused.npz")
    return out, n_fused, tuple(z["o%d" % i] for i in range(len(out))), int(z["n"])


def test_fused_head_tail(gpu, engines, tmp_path):
    """YOLOv8n's detect branches end in a 1x1 conv (22.box*.2, 22.cls*.2) that now runs in the epilogue of the 3x3 before it, on the
    tile in registers (tail_1x1, csrc/conv_common.hpp).  Same products, same K order, same roundings: the raw head must be
    BIT-IDENTICAL to the engine run with every conv on its own (child process, AICAM_NO_TAIL=1), with seven launches fewer: the six
    branch ends, and 3.conv (3x3 / stride 2, 32 -> 64) whose only reader is 4.c2f.cv1 (1x1, 64 -> 64) -- the same graph property.
    48 images: level 0 takes the patch kernel (box) and the 512-pixel DMA tile (cls), levels 1 and 2 the smaller DMA tiles."""
    x = np.random.default_rng(7).standard_normal((48, 3, 640, 640)).astype(np.float32) * 0.5
    (dfl, cls), n_f, (dfl_u, cls_u), n_u = _run_with_and_without_tail(tmp_path, engines[0], x, 48, "yolo_head_np")
    print(f"fused head tails: conv launches {n_u} -> {n_f}; max |dfl| {np.abs(dfl).max():.2f}, max |cls| {np.abs(cls).max():.2f}")
    assert n_u - n_f == 7, (n_u, n_f)
    assert np.isfinite(dfl).all() and np.abs(dfl).max() > 0.1 and np.abs(cls).max() > 0.1
    assert np.array_equal(dfl, dfl_u) and np.array_equal(cls, cls_u)


def test_merged_detect_branch_heads(gpu, engines, tmp_path):
    """22.box{l}.0 and 22.cls{l}.0 read the same feature map: the engine runs them as ONE conv with 64 + 80 output channels side by side
    (Model::Model merges them at load time, on the levels where that is faster: maps up to 40 x 40; their readers take channel
    slices).  Per output channel nothing changes, so the raw head of 48 images must be BIT-IDENTICAL to the engine loaded with
    AICAM_NO_MERGE=1 in a child process, with two conv launches fewer (levels 1 and 2) -- and again with AICAM_MERGE_MAXPX=6400 in the
    parent's place, which merges the 80 x 80 level too (three fewer: the 256 x 144 tile)."""
    x = np.random.default_rng(9).standard_normal((48, 3, 640, 640)).astype(np.float32) * 0.5
    (dfl, cls), n_f, (dfl_u, cls_u), n_u = _run_with_and_without_tail(tmp_path, engines[0], x, 48, "yolo_head_np", {"AICAM_NO_MERGE": "1"})
    assert n_u - n_f == 2, (n_u, n_f)
    (_, _), _, (dfl_a, cls_a), n_a = _run_with_and_without_tail(tmp_path, engines[0], x, 48, "yolo_head_np", {"AICAM_MERGE_MAXPX": "6400"})
    assert n_u - n_a == 3 and np.array_equal(dfl, dfl_a) and np.array_equal(cls, cls_a)
    assert np.abs(dfl).max() > 0.1 and np.abs(cls).max() > 0.1
    assert np.array_equal(dfl, dfl_u) and np.array_equal(cls, cls_u)


@pytest.mark.parametrize("n_crops", [24, 416])
def test_downsample_branch_folded_into_last_conv(gpu, engines, tmp_path, n_crops):
    """ReID layer{2,3,4}.0: the 1x1 / stride-2 downsample conv is read only as the residual of the block's last conv.  The engine folds
    it into that conv as a SECOND SOURCE (Model::Model fold, ConvArgs::x2): relu(conv3x3(t) + b + ds(x) + b') is one GEMM over
    K = [window of t | channels of x], three launches fewer, and the branch never exists as a tensor.  Its sum no longer passes through
    an fp16 rounding, so the embeddings are NOT bit-identical to the unfolded engine (child process, AICAM_NO_DS_FOLD=1): they must
    agree with it within that rounding, be within the usual fp16 tolerance of the fp32 oracle, and be no further from it than the
    unfolded engine is plus 5e-5 (measured: 9.8e-5 folded, 8.5e-5 unfolded, 8.3e-5 between the two).  24 crops: the LDS-DMA implicit GEMMs walk the second source (set_tap / xs); 416: the ping-pong patch kernel's
    extra step (conv3x3_pp_patch_kernel<..., X2>) -- test_reid_embeddings_do_not_depend_on_the_batch holds the two to the same bits."""
    x = np.random.default_rng(21).standard_normal((n_crops, 3, 128, 64)).astype(np.float32)
    (got,), n_f, (unf,), n_u = _run_with_and_without_tail(tmp_path, engines[1], x, n_crops, "reid_infer_np", {"AICAM_NO_DS_FOLD": "1"})
    assert n_u - n_f == 3, (n_u, n_f)
    eo = N.EngineOracle(engines[1])
    k = min(n_crops, 32)
    ref = eo.run(torch.from_numpy(x[:k]))[eo.outputs[0][0]][:, :, 0, 0].numpy()
    e_f, e_u, d = np.abs(got[:k] - ref).max(), np.abs(unf[:k] - ref).max(), np.abs(got - unf).max()
    print(f"downsample fold, {n_crops} crops: conv launches {n_u} -> {n_f}; |folded - oracle| {e_f:.2e}, |unfolded - oracle| {e_u:.2e}, |folded - unfolded| {d:.2e}")
    assert e_f < 5e-4 and d < 5e-4
    assert e_f <= e_u + 5e-5


@pytest.mark.parametrize("n_img", [4, 48])
def test_upsample_folded_into_its_reader(gpu, engines, tmp_path, n_img):
    """YOLOv8's neck: up(P5) | P4 -> 12.c2f.cv1 and up(12) | P3 -> 15.c2f.cv1.  Each 2x upsample writes the first channels of a concat
    buffer that exactly one 1x1 conv reads; the engine folds it into that conv, which takes those channels from the half-resolution
    tensor at (y >> 1, x >> 1) (Model::Model fold, ConvArgs::xs, the memory-order walk of conv_igemm_dma_kernel).  Same values in the
    same K order: the raw head must be BIT-IDENTICAL to the engine loaded with AICAM_NO_UP_FOLD=1 in a child process.  4 images: the
    small tiles; 48: the tiles of the production batch."""
    x = np.random.default_rng(13).standard_normal((n_img, 3, 640, 640)).astype(np.float32) * 0.5
    (dfl, cls), n_f, (dfl_u, cls_u), n_u = _run_with_and_without_tail(tmp_path, engines[0], x, n_img, "yolo_head_np", {"AICAM_NO_UP_FOLD": "1"})
    assert n_u == n_f                                   # conv launches: the upsample was never one
    assert np.isfinite(dfl).all() and np.abs(dfl).max() > 0.1 and np.abs(cls).max() > 0.1
    assert np.array_equal(dfl, dfl_u) and np.array_equal(cls, cls_u)


def test_fused_tail_wide_patch_and_narrow_tail(gpu, tmp_path):
    """The forms the YOLOv8 engines do not reach: the 8 x 32-tile patch kernel with a tail (map width a multiple of 32), a tail
    with FEWER output channels than its lead (24 of 64: part of the MFMA tiles is padding), SiLU on the tail and fp16 output,
    an 80-channel lead whose tail has 48 outputs.  Bit-identical to the unfused engine; oracle within fp16 rounding."""
    g = ef.Graph(ef.KIND_REID, 64, 64)
    wg = ef._WeightGen(11)

    def conv(name, src, dst, cin, cout, k, s, act, **kw):
        g.conv(name, src, dst, cin, cout, k, s, act, wb=wg(cout, cin, k, act), **kw)
    inp = g.buf(64, 64, 8)
    a = g.buf(64, 64, 64); conv("c0", inp, a, 3, 64, 3, 1, ef.ACT_SILU)
    b = g.buf(64, 64, 64); conv("lead64", a, b, 64, 64, 3, 1, ef.ACT_SILU)
    c = g.buf(64, 64, 24); conv("tail24", b, c, 64, 24, 1, 1, ef.ACT_SILU)
    d = g.buf(64, 64, 80); conv("lead80", c, d, 24, 80, 3, 1, ef.ACT_SILU)
    e = g.buf(64, 64, 48); conv("tail48", d, e, 80, 48, 1, 1, ef.ACT_NONE)
    p = g.buf(1, 1, 48); g.simple(ef.OP_AVGPOOL, e, p, 48)
    q = g.buf(1, 1, 64); conv("fc", p, q, 48, 64, 1, 1, ef.ACT_NONE)
    emb = g.buf(1, 1, 64, ef.DT_F32); g.simple(ef.OP_L2NORM, q, emb, 64)
    g.outputs.append([emb, 64, 0, 0, 0, 0, 0, 0]); g.meta = [64, 0, 0, 0, 0, 0, 0, 0]
    path = str(tmp_path / "tail.aicw")
    ef.write_engine(path, g)
    x = np.random.default_rng(5).standard_normal((56, 3, 64, 64)).astype(np.float32)     # 56 x 4096 pixels: the patch kernel applies
    (got,), n_f, (unf,), n_u = _run_with_and_without_tail(tmp_path, path, x, 56, "reid_infer_np")
    assert n_u - n_f == 2, (n_u, n_f)
    assert np.array_equal(got, unf)
    eo = N.EngineOracle(path)
    torch.set_num_threads(8)
    ref = eo.run(torch.from_numpy(x[:8]))[eo.outputs[0][0]][:, :, 0, 0].numpy()
    err = np.abs(got[:8] - ref).max()
    print(f"fused tails (wide patch, narrow tails): conv launches {n_u} -> {n_f}; err vs fp32 oracle {err:.2e}")
    assert err < 5e-3
