"""PyTorch-CPU fp32 interpreter of an ``.aicw`` engine file + decode + NMS (TEST INFRASTRUCTURE).

PARITY UNPINNED against the reference: its conv arithmetic, DFL decode and NMS live inside
TensorRT engines built from ONNX files that are not in the repository
(src/trt_utils/trt_engine.py:49-58,172-191; scripts/download_models.sh:7-8).  This oracle is
the plain fp32 restatement of the same graph the HIP executor runs:

* graph ops: conv+bias(+SiLU/ReLU)(+residual), SPPF max-pools, nearest 2x upsample,
  max-pool 3x3/2, global average pool, L2 normalise -- NCHW torch tensors, fp32;
* decode: DFL softmax-expectation over reg_max bins, ltrb -> xyxy * stride, anchor centre
  (i + 0.5); class score = sigmoid(logit), label = arg-max class;
* NMS (build decision D4): candidates with max-class logit >= logit(conf) sorted by
  (logit desc, anchor index asc), class-aware greedy suppression at IoU > iou_thresh,
  at most max_det survivors -- the four tensors the reference detector reads
  (src/detector/yolo_detector.py:44-54,108-112).

It has its own parser of the engine file so that a serialisation bug in the product cannot
cancel out.
"""
from __future__ import annotations

import math
import struct

import numpy as np
import torch
import torch.nn.functional as F

OP_CONV, OP_SPPF_POOL, OP_UPSAMPLE2X, OP_MAXPOOL3S2, OP_AVGPOOL, OP_L2NORM = 1, 2, 3, 4, 5, 6


class EngineOracle:
    def __init__(self, path_or_bytes, dtype=torch.float32):
        """dtype=torch.float64: the same graph evaluated in double precision -- the anchor that tells an fp32 summation-order
        difference (HIP MFMA order vs torch's CPU conv order) from an error: both fp32 evaluations are compared with it."""
        self.dtype = dtype
        blob = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
        hdr = struct.unpack_from("<II7i8i", blob, 0)
        assert hdr[0] == 0x57434941 and hdr[1] == 1, "bad engine file"
        self.kind, self.in_h, self.in_w, nb, no, nw, nout = hdr[2:9]
        self.meta = hdr[9:]
        off = struct.calcsize("<II7i8i")
        self.buffers = np.frombuffer(blob, "<i4", nb * 4, off).reshape(nb, 4).tolist(); off += nb * 16
        self.ops = np.frombuffer(blob, "<i4", no * 20, off).reshape(no, 20).tolist(); off += no * 80
        tab = np.frombuffer(blob, "<i8", nw * 6, off).reshape(nw, 6).tolist(); off += nw * 48
        self.outputs = np.frombuffer(blob, "<i4", nout * 8, off).reshape(nout, 8).tolist(); off += nout * 32
        payload = np.frombuffer(blob, "<f4", -1, off)
        self.weights = []
        for co, ci, kh, kw, wo, bo in tab:
            w = torch.from_numpy(payload[wo:wo + co * ci * kh * kw].reshape(co, ci, kh, kw).copy()).to(dtype)
            b = torch.from_numpy(payload[bo:bo + co].copy()).to(dtype)
            self.weights.append((w, b))

    # ------------------------------------------------------------------ graph
    @torch.no_grad()
    def run(self, x_nchw: torch.Tensor, keep=None):
        """x: fp32 [N,3,H,W]. Returns the list of buffers (NCHW, channels = buffer width)."""
        n = x_nchw.shape[0]
        bufs = [None] * len(self.buffers)
        dt = self.dtype
        bufs[0] = torch.cat([x_nchw.to(dt), torch.zeros(n, self.buffers[0][2] - 3, self.in_h, self.in_w, dtype=dt)], 1)

        def get(bi, coff, c):
            return bufs[bi][:, coff:coff + c]

        def put(bi, coff, val):
            h, w, c, _ = self.buffers[bi]
            if bufs[bi] is None:
                bufs[bi] = torch.zeros(n, c, h, w, dtype=dt)
            bufs[bi][:, coff:coff + val.shape[1]] = val

        for o in self.ops:
            typ, sb, sc, cin, db, dc, cout, kh, kw, st, pad, act, rb, rc, rmode, wi = o[:16]
            if typ == OP_CONV:
                w, b = self.weights[wi]
                y = F.conv2d(get(sb, sc, cin), w, b, stride=st, padding=pad)
                if rmode == 1:
                    y = y + get(rb, rc, cout)
                y = F.silu(y) if act == 1 else (F.relu(y) if act == 2 else y)
                if rmode == 2:
                    y = y + get(rb, rc, cout)
                put(db, dc, y)
            elif typ == OP_SPPF_POOL:
                y = get(sb, sc, cin)
                for k in range(3):
                    y = F.max_pool2d(y, 5, 1, 2)
                    put(db, dc + k * cin, y)
            elif typ == OP_UPSAMPLE2X:
                put(db, dc, F.interpolate(get(sb, sc, cin), scale_factor=2, mode="nearest"))
            elif typ == OP_MAXPOOL3S2:
                put(db, dc, F.max_pool2d(get(sb, sc, cin), 3, 2, 1))
            elif typ == OP_AVGPOOL:
                put(db, dc, get(sb, sc, cin).mean(dim=(2, 3), keepdim=True))
            elif typ == OP_L2NORM:
                v = get(sb, sc, cin)
                put(db, dc, v / v.norm(p=2, dim=1, keepdim=True).clamp_min(1e-12))
            else:
                raise ValueError(typ)
        return bufs

    # ------------------------------------------------------------------ YOLO
    @torch.no_grad()
    def yolo_head(self, x_nchw):
        """-> (dfl logits [N,A,4*reg_max], class logits [N,A,nc]); anchors level-major, row-major."""
        bufs = self.run(x_nchw)
        dfl, cls = [], []
        for box_b, cls_b, *_ in self.outputs:
            dfl.append(bufs[box_b].permute(0, 2, 3, 1).flatten(1, 2))
            cls.append(bufs[cls_b].permute(0, 2, 3, 1).flatten(1, 2))
        return torch.cat(dfl, 1), torch.cat(cls, 1)

    def anchors(self):
        pts, strides = [], []
        for _, _, s, h, w, *_ in self.outputs:
            ys, xs = np.meshgrid(np.arange(h, dtype=np.float32) + 0.5, np.arange(w, dtype=np.float32) + 0.5,
                                 indexing="ij")
            pts.append(np.stack([xs.ravel(), ys.ravel()], 1))
            strides.append(np.full(h * w, s, np.float32))
        return np.concatenate(pts), np.concatenate(strides)

    def decode(self, dfl_logits, cls_logits, ft=np.float32):
        """numpy fp32 (ft=np.float64: the double-precision anchor): boxes [N,A,4] xyxy (letterbox px), max logit [N,A], label [N,A]."""
        nc, reg_max = self.meta[0], self.meta[1]
        d = np.asarray(dfl_logits, ft)
        n, a, _ = d.shape
        d = d.reshape(n, a, 4, reg_max)
        e = np.exp(d - d.max(-1, keepdims=True)).astype(ft)
        p = e / e.sum(-1, keepdims=True, dtype=ft)
        dist = (p * np.arange(reg_max, dtype=ft)).sum(-1, dtype=ft)      # l t r b
        pts, st = self.anchors()
        x1 = (pts[:, 0] - dist[..., 0]) * st
        y1 = (pts[:, 1] - dist[..., 1]) * st
        x2 = (pts[:, 0] + dist[..., 2]) * st
        y2 = (pts[:, 1] + dist[..., 3]) * st
        c = np.asarray(cls_logits, ft)
        return np.stack([x1, y1, x2, y2], -1).astype(ft), c.max(-1), c.argmax(-1).astype(np.int32)


def logit_threshold(conf: float) -> np.float32:
    """fp32 logit such that sigmoid(l) >= conf  <=>  l >= thr (computed in fp64, rounded once)."""
    return np.float32(math.log(conf / (1.0 - conf)))


def box_iou_xyxy(a, b):
    """fp32 IoU of one box against many, the arithmetic the NMS kernel uses (no FMA)."""
    iw = np.maximum(np.float32(0), np.minimum(a[2], b[:, 2]) - np.maximum(a[0], b[:, 0]))
    ih = np.maximum(np.float32(0), np.minimum(a[3], b[:, 3]) - np.maximum(a[1], b[:, 1]))
    inter = iw * ih
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(area_a + area_b - inter, np.float32(1e-9))


def nms(boxes, max_logit, labels, conf, iou_thresh, max_det, return_margin=False):
    """Build decision D4 on one image. Returns (keep anchor indices in output order[, margin])."""
    thr = logit_threshold(conf)
    cand = np.nonzero(max_logit >= thr)[0]
    order = cand[np.lexsort((cand, -max_logit[cand].astype(np.float64)))]   # logit desc, index asc
    keep, margin = [], np.inf
    if len(cand):
        margin = float(np.abs(max_logit.astype(np.float64) - float(thr)).min())
    b = boxes[order]
    lab = labels[order]
    alive = np.ones(len(order), bool)
    for i in range(len(order)):
        if not alive[i]:
            continue
        keep.append(int(order[i]))
        if len(keep) >= max_det:
            break
        rest = np.nonzero(alive[i + 1:] & (lab[i + 1:] == lab[i]))[0] + i + 1
        if len(rest):
            iou = box_iou_xyxy(b[i], b[rest])
            margin = min(margin, float(np.abs(iou - np.float32(iou_thresh)).min()))
            alive[rest[iou > np.float32(iou_thresh)]] = False
    keep = np.asarray(keep, np.int64)
    return (keep, margin) if return_margin else keep


def sigmoid32(x):
    x = np.asarray(x, np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)
