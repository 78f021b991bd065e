#!/usr/bin/env python3
"""Train YOLOv8n on the synthetic workload so that the detector SEES the planted persons (BUILD CONTAINER ONLY, torch CPU autograd).

The reference's tracker is driven by its detector's own boxes (src/aicamera_tracker.py:180,193-195) and its weights come from a
download (scripts/download_models.sh:7-8) that is not reachable here.  The seeded engines of ai-camera_amd/engine_file.py fire on
background texture, so every own-detections statement (fp16 vs fp32 id parity, the inject=0 throughput leg) was a statement about a
near-degenerate scene.  This script produces weights for which class 0 ('person', src/config.py:36) means "a planted rectangle of
ai-camera_amd/synthetic.Scene":

  * architecture = engine_file.build_yolov8('n') (the pinned Ultralytics yolov8.yaml graph), executed here by a small torch
    interpreter of the same op list with Conv -> BatchNorm -> SiLU blocks (Ultralytics' Conv module); BatchNorm is folded into the
    conv when the weights are written, so the engine file has the same 63 conv + bias ops as every other engine;
  * data: Scene(seed) frames at 1280x720, letterboxed the way the engine's preprocess does for this size (2x2 integer mean, 114
    border), cropped to the 384 letterbox rows that carry the frame; ground truth = the planted boxes that are at least
    `--min-visible` visible under the painter's order of Scene.render;
  * loss: the YOLOv8 detection loss -- task-aligned assignment (top-10, alpha 0.5, beta 6), BCE on IoU-weighted class targets,
    CIoU + distribution focal loss on the positives (gains 7.5 / 0.5 / 1.5);
  * output: an ONNX file with fp16 initializers (what `yolo export format=onnx half=True` writes; 6.3 MB) through the repo's own
    exporter, re-read by ai-camera_amd/onnx_import.py wherever an engine is needed -- the f1 path exercised on weights that matter.

Deterministic given --seed and the thread count (torch CPU reductions are not bit-reproducible across thread counts; the committed
ONNX file is the artefact, this script is its provenance).

    python tools/train_synthetic_detector.py --steps 2500 --out weights/yolov8n_synth.onnx
"""
from __future__ import annotations

import argparse
import importlib
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
oi = importlib.import_module("ai-camera_amd.onnx_import")

ROW0 = 128          # letterbox rows [128, 512) of the 640 x 640 input hold the 360 frame rows (pad 140) + 12 border rows each side
ROWS = 384          # three batches of four are these rows only (0.6 x the convolution work); every fourth is the whole 640 x 640 letterbox,
                    # so that the flat 114-grey borders are seen as background (a net that never saw them fires all over them)


# ------------------------------------------------------------------------------------------------ the graph as a torch module
class GraphNet(nn.Module):
    """Interpreter of an engine_file.Graph with trainable convs.  Activated convs are conv(no bias) + BatchNorm + act, the six
    linear head convs are conv + bias.  Buffers are kept as channel slices (no in-place writes under autograd)."""

    def __init__(self, g: ef.Graph):
        super().__init__()
        self.g = g
        self.convs = nn.ModuleList()
        self.bns = nn.ModuleList()
        for o in g.ops:
            if o[0] != ef.OP_CONV:
                continue
            cin, cout, k, s, p, act = o[3], o[6], o[7], o[9], o[10], o[11]
            lin = act == ef.ACT_NONE
            self.convs.append(nn.Conv2d(cin, cout, k, s, p, bias=lin))
            self.bns.append(nn.Identity() if lin else nn.BatchNorm2d(cout, eps=1e-3, momentum=0.03))
        nc, reg_max = g.meta[0], g.meta[1]
        wi = 0
        for o in g.ops:                       # Ultralytics Detect.bias_init: box 1.0, cls log(5 / nc / (640 / s)^2)
            if o[0] != ef.OP_CONV:
                continue
            if o[11] == ef.ACT_NONE:
                stride = g.in_h // g.buffers[o[4]][0]
                with torch.no_grad():
                    if o[6] == 4 * reg_max:
                        self.convs[wi].bias.fill_(1.0)
                    else:
                        self.convs[wi].bias.fill_(math.log(5 / nc / (640 / stride) ** 2))
            wi += 1

    def forward(self, x):
        g = self.g
        bufs = {0: [(0, 3, x)]}

        def get(bi, coff, c):
            parts = []
            for s0, sc, t in sorted(bufs[bi], key=lambda e: e[0]):
                lo, hi = max(coff, s0), min(coff + c, s0 + sc)
                if lo < hi:
                    parts.append(t if (lo, hi) == (s0, s0 + sc) else t[:, lo - s0:hi - s0])
            assert sum(p.shape[1] for p in parts) == c, (bi, coff, c)
            return parts[0] if len(parts) == 1 else torch.cat(parts, 1)

        def put(bi, coff, t):
            c = t.shape[1]
            keep = [e for e in bufs.get(bi, []) if e[0] + e[1] <= coff or e[0] >= coff + c]
            bufs[bi] = keep + [(coff, c, t)]

        wi = 0
        for o in g.ops:
            typ, sb, sc, cin, db, dc, cout, kh, kw, st, pad, act, rb, rc, rmode = o[:15]
            if typ == ef.OP_CONV:
                y = self.bns[wi](self.convs[wi](get(sb, sc, cin)))
                wi += 1
                if rmode == ef.RES_ADD_THEN_ACT:
                    y = y + get(rb, rc, cout)
                y = F.silu(y) if act == ef.ACT_SILU else (F.relu(y) if act == ef.ACT_RELU else y)
                if rmode == ef.RES_ACT_THEN_ADD:
                    y = y + get(rb, rc, cout)
                put(db, dc, y)
            elif typ == ef.OP_SPPF_POOL:
                y = get(sb, sc, cin)
                for k in range(3):
                    y = F.max_pool2d(y, 5, 1, 2)
                    put(db, dc + k * cin, y)
            elif typ == ef.OP_UPSAMPLE2X:
                put(db, dc, F.interpolate(get(sb, sc, cin), scale_factor=2, mode="nearest"))
            else:
                raise ValueError(typ)
        out = []
        for box_b, cls_b, s, h, w, *_ in g.outputs:
            out.append((get(box_b, 0, g.buffers[box_b][2]), get(cls_b, 0, g.buffers[cls_b][2]), s))
        return out

    def folded_graph(self) -> ef.Graph:
        """The same graph with BatchNorm folded: w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps)."""
        g = self.g
        ws = []
        for conv, bn in zip(self.convs, self.bns):
            w = conv.weight.detach().double()
            if isinstance(bn, nn.BatchNorm2d):
                sc = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
                w = w * sc.view(-1, 1, 1, 1)
                b = bn.bias.detach().double() - bn.running_mean.double() * sc
            else:
                b = conv.bias.detach().double()
            ws.append((w.float().numpy(), b.float().numpy()))
        g.weights = ws
        return g


# ------------------------------------------------------------------------------------------------ data
def letterbox_rows(frame_bgr, row0=ROW0, rows=ROWS):
    """u8 [720, 1280, 3] BGR -> f32 [3, rows, 640] RGB / 255: the engine's preprocess for this frame size (2x2 integer mean with
    rounding, border 114: image_processing.py:37-68,93-99), rows row0 .. row0 + rows of the 640 x 640 letterbox."""
    a = frame_bgr.astype(np.int32)
    small = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.float32)
    img = np.full((rows, 640, 3), 114.0, np.float32)
    img[140 - row0:140 - row0 + 360] = small
    return np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1)) / np.float32(255.0)


def visible_fraction(sc, frame):
    """Share of every target's rectangle that Scene.render's painter's order leaves visible (later targets cover earlier ones)."""
    owner = np.full((sc.height, sc.width), -1, np.int32)
    boxes = sc.boxes_at(frame)
    vis = sc.visible(frame)
    area = np.zeros(sc.n_targets)
    for t in range(sc.n_targets):
        if not vis[t]:
            continue
        x1, y1, x2, y2 = (int(v) for v in boxes[t])
        x1, y1, x2, y2 = max(0, x1), max(0, y1), min(sc.width, x2), min(sc.height, y2)
        if x2 <= x1 or y2 <= y1:
            continue
        owner[y1:y2, x1:x2] = t
        area[t] = (x2 - x1) * (y2 - y1)
    seen = np.bincount(owner[owner >= 0], minlength=sc.n_targets).astype(np.float64)
    return np.where(area > 0, seen / np.maximum(area, 1), 0.0)


def sample(rng, min_visible, row0=ROW0, rows=ROWS):
    """One training image: a random scene at a random time.  Person counts 4 .. 40 so that the head does not learn the count."""
    n = int(rng.integers(4, 41))
    sc = syn.Scene(seed=int(rng.integers(1 << 20, 1 << 30)), n_targets=n)      # seeds far from the bench / test scenes (0 .. 10^4)
    f = int(rng.integers(0, 2048))
    frame = sc.render(f)
    vf = visible_fraction(sc, f)
    b = sc.detections(f)[0][vf >= min_visible]                                 # no gaps / births: detections() = every target
    gt = np.stack([b[:, 0] / 2, b[:, 1] / 2 + 140 - row0, b[:, 2] / 2, b[:, 3] / 2 + 140 - row0], 1).astype(np.float32)
    return letterbox_rows(frame, row0, rows), gt


def batch(rng, bs, min_visible, max_gt=48, full=False):
    xs, gts = zip(*(sample(rng, min_visible, *((0, 640) if full else (ROW0, ROWS))) for _ in range(bs)))
    gt = np.zeros((bs, max_gt, 4), np.float32)
    mask = np.zeros((bs, max_gt), bool)
    for i, b in enumerate(gts):
        gt[i, :len(b)] = b
        mask[i, :len(b)] = True
    return torch.from_numpy(np.stack(xs)).contiguous(memory_format=torch.channels_last), torch.from_numpy(gt), torch.from_numpy(mask)


# ------------------------------------------------------------------------------------------------ loss
def bbox_iou(a, b, ciou=False, eps=1e-7):
    """IoU / CIoU of xyxy boxes, broadcast over leading dims."""
    ax1, ay1, ax2, ay2 = a.unbind(-1)
    bx1, by1, bx2, by2 = b.unbind(-1)
    w1, h1, w2, h2 = ax2 - ax1, ay2 - ay1 + eps, bx2 - bx1, by2 - by1 + eps
    inter = (torch.minimum(ax2, bx2) - torch.maximum(ax1, bx1)).clamp_(0) * (torch.minimum(ay2, by2) - torch.maximum(ay1, by1)).clamp_(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if not ciou:
        return iou
    cw = torch.maximum(ax2, bx2) - torch.minimum(ax1, bx1)
    chh = torch.maximum(ay2, by2) - torch.minimum(ay1, by1)
    c2 = cw ** 2 + chh ** 2 + eps
    rho2 = ((bx1 + bx2 - ax1 - ax2) ** 2 + (by1 + by2 - ay1 - ay2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def anchors_for(outs):
    pts, strides = [], []
    for box, _, s in outs:
        h, w = box.shape[2:]
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
        pts.append(torch.stack([xs.flatten(), ys.flatten()], 1))
        strides.append(torch.full((h * w,), float(s)))
    return torch.cat(pts), torch.cat(strides)


def detection_loss(outs, gt, gt_mask, reg_max=16, topk=10, alpha=0.5, beta=6.0):
    B = gt.shape[0]
    dfl = torch.cat([b.flatten(2) for b, _, _ in outs], 2).permute(0, 2, 1)            # [B, A, 64]
    cls = torch.cat([c.flatten(2) for _, c, _ in outs], 2).permute(0, 2, 1)            # [B, A, nc]
    pts, strides = anchors_for(outs)                                                   # grid units, px per unit
    A = pts.shape[0]
    proj = torch.arange(reg_max, dtype=torch.float32)
    dist = (dfl.view(B, A, 4, reg_max).softmax(-1) * proj).sum(-1)                    # l t r b in grid units
    pbox = torch.cat([pts - dist[..., :2], pts + dist[..., 2:]], -1)                  # xyxy, grid units
    pbox_px = pbox * strides.view(1, A, 1)
    ctr = pts * strides.view(A, 1)                                                     # [A, 2] px
    with torch.no_grad():                                                              # task-aligned assignment
        lt = ctr.view(1, 1, A, 2) - gt[:, :, None, :2]
        rb = gt[:, :, None, 2:] - ctr.view(1, 1, A, 2)
        inside = (torch.cat([lt, rb], -1).amin(-1) > 1e-9) & gt_mask[:, :, None]      # [B, M, A]
        score0 = cls[..., 0].sigmoid()[:, None, :].expand(-1, gt.shape[1], -1)        # person class
        ov = bbox_iou(gt[:, :, None, :], pbox_px.detach()[:, None, :, :], ciou=True).clamp_(0) * inside
        metric = score0.pow(alpha) * ov.pow(beta) * inside
        top = torch.topk(metric, topk, dim=-1).indices
        mtop = torch.zeros_like(metric, dtype=torch.bool).scatter_(-1, top, True)
        pos = mtop & inside                                                            # [B, M, A]
        multi = pos.sum(1) > 1
        if multi.any():                                                                # an anchor claimed by several boxes: the best overlap wins
            best = ov.argmax(1)                                                        # [B, A]
            only = F.one_hot(best, gt.shape[1]).permute(0, 2, 1).bool()
            pos = torch.where(multi[:, None, :], only & pos.any(1, keepdim=True), pos)
        fg = pos.any(1)                                                                # [B, A]
        gi = pos.float().argmax(1)                                                     # [B, A] index of the box
        tbox = torch.gather(gt, 1, gi[..., None].expand(-1, -1, 4))                    # [B, A, 4] px
        metric = metric * pos
        pos_metric = metric.amax(-1, keepdim=True)
        pos_ov = (ov * pos).amax(-1, keepdim=True)
        norm = (metric * pos_ov / (pos_metric + 1e-9)).amax(1)                         # [B, A]
        tscore = torch.zeros_like(cls)
        tscore[..., 0] = norm * fg
        tsum = tscore.sum().clamp_(min=1.0)
    l_cls = F.binary_cross_entropy_with_logits(cls, tscore, reduction="sum") / tsum
    if fg.any():
        wgt = tscore.sum(-1)[fg]
        iou = bbox_iou(pbox_px[fg], tbox[fg], ciou=True)
        l_box = ((1.0 - iou) * wgt).sum() / tsum
        tb = tbox / strides.view(1, A, 1)
        tdist = torch.cat([pts - tb[..., :2], tb[..., 2:] - pts], -1).clamp_(0, reg_max - 1 - 0.01)[fg]      # [P, 4]
        tl = tdist.long()
        wl = tl + 1 - tdist
        logits = dfl.view(B, A, 4, reg_max)[fg].reshape(-1, reg_max)
        ce_l = F.cross_entropy(logits, tl.view(-1), reduction="none").view(tl.shape)
        ce_r = F.cross_entropy(logits, (tl + 1).view(-1), reduction="none").view(tl.shape)
        l_dfl = ((ce_l * wl + ce_r * (1 - wl)).mean(-1) * wgt).sum() / tsum
    else:
        l_box = l_dfl = dfl.sum() * 0
    return (7.5 * l_box + 0.5 * l_cls + 1.5 * l_dfl) * B, (float(l_box.detach()), float(l_cls.detach()), float(l_dfl.detach()), int(fg.sum()))


# ------------------------------------------------------------------------------------------------ evaluation (this script's own decode)
@torch.no_grad()
def evaluate(net, rng, n_images=8, conf=0.3, iou_thr=0.5, min_visible=0.0):
    """Recall of the planted boxes at IoU >= 0.5 by (class-0 score >= conf) predictions after a plain greedy NMS, and the count of
    predictions matching nothing."""
    net.eval()
    hit = tot = extra = 0
    for _ in range(n_images):
        x, gt, m = batch(rng, 1, min_visible, full=True)
        outs = net(x)
        dfl = torch.cat([b.flatten(2) for b, _, _ in outs], 2).permute(0, 2, 1)[0]
        cls = torch.cat([c.flatten(2) for _, c, _ in outs], 2).permute(0, 2, 1)[0]
        pts, st = anchors_for(outs)
        d = (dfl.view(-1, 4, 16).softmax(-1) * torch.arange(16.0)).sum(-1)
        box = torch.cat([pts - d[:, :2], pts + d[:, 2:]], -1) * st[:, None]
        s = cls[:, 0].sigmoid()
        k = s >= conf
        box, s = box[k], s[k]
        order = s.argsort(descending=True)
        keep = []
        for i in order.tolist():
            if all(float(bbox_iou(box[i], box[j])) <= 0.5 for j in keep):
                keep.append(i)
        pb = box[keep]
        g = gt[0][m[0]]
        tot += len(g)
        if len(pb):
            iou = bbox_iou(g[:, None, :], pb[None, :, :])
            hit += int((iou.amax(1) >= iou_thr).sum())
            extra += int((iou.amax(0) < iou_thr).sum())
    net.train()
    return hit / max(tot, 1), extra / n_images


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2500)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--min-visible", type=float, default=0.25)
    ap.add_argument("--out", default=os.path.join(ROOT, "weights", "yolov8n_synth.onnx"))
    ap.add_argument("--ckpt", default="/tmp/yolov8n_synth_ckpt.pt")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--export-only", action="store_true", help="write --out from the checkpoint as it stands (no training step)")
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    torch.set_num_threads(args.threads)
    rng = np.random.default_rng(args.seed)
    g = ef.build_yolov8("n", calibrate=False)
    net = GraphNet(g).to(memory_format=torch.channels_last)        # mkldnn's NHWC convs: 3x faster than NCHW on these thin layers
    decay, no_decay = [], []
    for n_, p in net.named_parameters():
        (decay if p.ndim == 4 else no_decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 5e-4}, {"params": no_decay, "weight_decay": 0.0}], lr=args.lr)
    warm = 100
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(1.0, (s + 1) / warm) * (0.02 + 0.98 * 0.5 * (1 + math.cos(math.pi * min(s, args.steps) / args.steps))))
    step0 = 0
    if (args.resume or args.export_only) and os.path.exists(args.ckpt):
        ck = torch.load(args.ckpt)
        net.load_state_dict(ck["net"]); opt.load_state_dict(ck["opt"]); sched.load_state_dict(ck["sched"]); step0 = ck["step"]
    net.train()
    t0 = time.time()
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(3)                    # batches are rendered ahead on three threads, each from its own (seed, step) stream
    ahead = {}

    def want(s):
        if s < args.steps and s not in ahead:
            full = s % 4 == 3
            ahead[s] = pool.submit(batch, np.random.default_rng([args.seed, s]), max(1, args.batch * 5 // 8) if full else args.batch, args.min_visible, 48, full)
    for step in range(step0, 0 if args.export_only else args.steps):
        for s in range(step, step + 4):
            want(s)
        x, gt, m = ahead.pop(step).result()
        loss, parts = detection_loss(net(x), gt, m)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
        opt.step()
        sched.step()
        if step % 20 == 0 or step == args.steps - 1:
            print(f"step {step:5d} loss {float(loss) / args.batch:8.4f} box {parts[0]:.4f} cls {parts[1]:.4f} dfl {parts[2]:.4f} pos {parts[3]:5d} "
                  f"lr {sched.get_last_lr()[0]:.2e} {time.time() - t0:7.0f}s", flush=True)
        if (step + 1) % 250 == 0 or step == args.steps - 1:
            rec, extra = evaluate(net, np.random.default_rng(12345), 6)
            print(f"   eval @ {step + 1}: recall(IoU>=0.5, conf 0.3) {rec:.4f}, unmatched predictions per image {extra:.2f}", flush=True)
            torch.save({"net": net.state_dict(), "opt": opt.state_dict(), "sched": sched.state_dict(), "step": step + 1}, args.ckpt)
    net.eval()
    gf = net.folded_graph()
    # fp16-representable weights: the ONNX initializers are fp16 (half export), every engine built from them holds the same values
    gf.weights = [(w.astype(np.float16).astype(np.float32), b.astype(np.float16).astype(np.float32)) for w, b in gf.weights]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    blob = oi.export_onnx(gf, nms={"score_threshold": 0.3, "iou_threshold": 0.5, "max_output_boxes": 300}, module_names=True, half=True)
    with open(args.out, "wb") as f:
        f.write(blob)
    back, info = oi.onnx_to_engine(blob)
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(gf.weights, back.weights))
    print(f"wrote {args.out}: {len(blob) / 1e6:.2f} MB, {info['mapping']}")


if __name__ == "__main__":
    main()
