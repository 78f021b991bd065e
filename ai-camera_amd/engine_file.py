"""Engine files: graph IR + fp32 weights, the analogue of the reference's TensorRT engines.

The reference builds its two engines with ``trtexec`` from ONNX files it downloads
(``scripts/download_models.sh:7-8``, ``scripts/export_trt_engines.sh:25-37,57-89``); neither
the ONNX files nor any architecture definition is in the repository.  This
module pins the architectures in source (the public Ultralytics
``yolov8.yaml`` graph and the DeepSORT ReID trunk) and serialises them as a flat op list
that ``libaicam.so`` executes with hand-written HIP kernels.  BatchNorm is folded, so every
conv is conv + bias (+ SiLU / ReLU) (+ residual).

Layout of an ``.aicw`` file (little endian):

    u32 magic 'AICW', u32 version, i32 kind, i32 in_h, i32 in_w, i32 n_buf, i32 n_op,
    i32 n_w, i32 n_out, i32 meta[8]
    n_buf x i32[4]   (h, w, c, dtype)          buffer 0 is the input, NHWC, c = 8 (RGB + 5 zero lanes)
    n_op  x i32[20]  (see OP_* below)
    n_w   x i64[6]   (cout, cin, kh, kw, weight offset, bias offset)   offsets in floats
    n_out x i32[8]   YOLO: (box_buf, cls_buf, stride, h, w, 0,0,0) per level; ReID: (emb_buf, dim,...)
    f32 payload      weights in PyTorch OIHW order, biases

All tensors are NHWC; concatenations are realised by producers writing channel slices of a
shared buffer (``dst_coff``), never by copies.
"""
from __future__ import annotations

import math
import os
import struct
from dataclasses import dataclass, field

import numpy as np

MAGIC = 0x57434941  # 'AICW'
VERSION = 1
KIND_YOLO, KIND_REID = 1, 2
DT_ACT, DT_F32 = 0, 1
OP_CONV, OP_SPPF_POOL, OP_UPSAMPLE2X, OP_MAXPOOL3S2, OP_AVGPOOL, OP_L2NORM = 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
RES_NONE, RES_ADD_THEN_ACT, RES_ACT_THEN_ADD = 0, 1, 2
OP_WORDS = 20
IN_C = 8   # input pixels are stored RGB00000 (16 B in fp16) so the 3-channel stems run through the MFMA implicit GEMM

YOLO_SCALES = {  # depth, width, max_channels (Ultralytics yolov8.yaml)
    "n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512),
}
# Fraction of anchors whose best class passes conf 0.3 on the calibration frame of the seeded
# engines (-> O(10^3) NMS candidates out of 8400).
SEEDED_PASS_FRACTION = 0.10


@dataclass
class Graph:
    kind: int
    in_h: int
    in_w: int
    buffers: list = field(default_factory=list)   # (h, w, c, dtype)
    ops: list = field(default_factory=list)       # int lists of OP_WORDS
    weights: list = field(default_factory=list)   # (w[O,I,kh,kw] f32, b[O] f32)
    names: list = field(default_factory=list)     # conv names, parallel to weights
    outputs: list = field(default_factory=list)   # int[8] each
    meta: list = field(default_factory=lambda: [0] * 8)

    # ---- builders
    def buf(self, h, w, c, dtype=DT_ACT):
        self.buffers.append((int(h), int(w), int(c), int(dtype)))
        return len(self.buffers) - 1

    def conv(self, name, src, dst, cin, cout, k, s, act, *, src_coff=0, dst_coff=0, res=None,
             res_mode=RES_NONE, wb=None):
        sb, db = self.buffers[src], self.buffers[dst]
        p = k // 2
        assert (sb[0] + 2 * p - k) // s + 1 == db[0] and (sb[1] + 2 * p - k) // s + 1 == db[1], name
        assert src_coff + cin <= sb[2] and dst_coff + cout <= db[2], name
        res_buf, res_coff = (-1, 0) if res is None else res
        self.weights.append(wb)
        self.names.append(name)
        self.ops.append([OP_CONV, src, src_coff, cin, dst, dst_coff, cout, k, k, s, p, act,
                         res_buf, res_coff, res_mode, len(self.weights) - 1, 0, 0, 0, 0])

    def simple(self, op, src, dst, c, *, src_coff=0, dst_coff=0):
        self.ops.append([op, src, src_coff, c, dst, dst_coff, c] + [0] * 5 + [-1, 0, 0, -1, 0, 0, 0, 0])

    # ---- derived figures
    def conv_macs(self):
        total = 0
        for o in self.ops:
            if o[0] == OP_CONV:
                h, w, _, _ = self.buffers[o[4]]
                total += h * w * o[6] * o[3] * o[7] * o[8]
        return total

    def n_params(self):
        return sum(w.size + b.size for w, b in self.weights)


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


class _WeightGen:
    """Seeded fp32 weights, generated in graph order. Variance-preserving for the
    activation that follows so fp16 activations stay O(1) through 60+ layers."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def __call__(self, cout, cin, k, act, gain=1.0, bias_std=0.05, bias_mean=0.0):
        fan_in = cin * k * k
        # SiLU keeps ~0.36 of a unit-variance input's second moment, ReLU 0.5
        act_gain = {ACT_SILU: 1.0 / math.sqrt(0.356), ACT_RELU: math.sqrt(2.0), ACT_NONE: 1.0}[act]
        std = gain * act_gain / math.sqrt(fan_in)
        w = (self.rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32)
        b = (bias_mean + self.rng.standard_normal(cout) * bias_std).astype(np.float32)
        return w, b


def build_yolov8(scale="n", nc=80, in_hw=(640, 640), seed=0, reg_max=16, calibrate=True,
                 frame_hw=(720, 1280)) -> Graph:
    depth, width, max_ch = YOLO_SCALES[scale]
    ch = lambda c: make_divisible(min(c, max_ch) * width, 8)        # noqa: E731
    rep = lambda n: max(round(n * depth), 1)                          # noqa: E731
    H, W = in_hw
    assert H % 32 == 0 and W % 32 == 0
    g = Graph(KIND_YOLO, H, W)
    wg = _WeightGen(seed)

    def conv(name, src, dst, cin, cout, k, s, act=ACT_SILU, **kw):
        gain = kw.pop("gain", 1.0)
        bmean = kw.pop("bias_mean", 0.0)
        g.conv(name, src, dst, cin, cout, k, s, act, wb=wg(cout, cin, k, act, gain, bias_mean=bmean), **kw)

    def c2f(name, src, src_coff, c1, dst, dst_coff, c2, n, shortcut, h, w):
        c = c2 // 2
        cat = g.buf(h, w, (2 + n) * c)
        conv(f"{name}.cv1", src, cat, c1, 2 * c, 1, 1, src_coff=src_coff)
        tmp = g.buf(h, w, c)
        for i in range(n):
            conv(f"{name}.m{i}.cv1", cat, tmp, c, c, 3, 1, src_coff=(1 + i) * c)
            if shortcut:  # x + cv2(cv1(x)); halve the branch so the sum keeps unit variance
                conv(f"{name}.m{i}.cv2", tmp, cat, c, c, 3, 1, dst_coff=(2 + i) * c, gain=0.5,
                     res=(cat, (1 + i) * c), res_mode=RES_ACT_THEN_ADD)
            else:
                conv(f"{name}.m{i}.cv2", tmp, cat, c, c, 3, 1, dst_coff=(2 + i) * c)
        conv(f"{name}.cv2", cat, dst, (2 + n) * c, c2, 1, 1, dst_coff=dst_coff)

    c1, c2_, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    inp = g.buf(H, W, IN_C)
    b0 = g.buf(H // 2, W // 2, c1)
    conv("0.conv", inp, b0, 3, c1, 3, 2)
    b1 = g.buf(H // 4, W // 4, c2_)
    conv("1.conv", b0, b1, c1, c2_, 3, 2)
    b2 = g.buf(H // 4, W // 4, c2_)
    c2f("2.c2f", b1, 0, c2_, b2, 0, c2_, rep(3), True, H // 4, W // 4)
    b3 = g.buf(H // 8, W // 8, c3)
    conv("3.conv", b2, b3, c2_, c3, 3, 2)
    # concat buffers of the neck (producers write slices)
    cat14 = g.buf(H // 8, W // 8, c4 + c3)       # [up(12) | 4]
    cat11 = g.buf(H // 16, W // 16, c5 + c4)     # [up(9)  | 6]
    cat17 = g.buf(H // 16, W // 16, c3 + c4)     # [16     | 12]
    cat20 = g.buf(H // 32, W // 32, c4 + c5)     # [19     | 9]
    c2f("4.c2f", b3, 0, c3, cat14, c4, c3, rep(6), True, H // 8, W // 8)
    b5 = g.buf(H // 16, W // 16, c4)
    conv("5.conv", cat14, b5, c3, c4, 3, 2, src_coff=c4)
    c2f("6.c2f", b5, 0, c4, cat11, c5, c4, rep(6), True, H // 16, W // 16)
    b7 = g.buf(H // 32, W // 32, c5)
    conv("7.conv", cat11, b7, c4, c5, 3, 2, src_coff=c5)
    b8 = g.buf(H // 32, W // 32, c5)
    c2f("8.c2f", b7, 0, c5, b8, 0, c5, rep(3), True, H // 32, W // 32)
    # 9 SPPF -> cat20[c4:]
    ch_ = c5 // 2
    sp = g.buf(H // 32, W // 32, 4 * ch_)
    conv("9.sppf.cv1", b8, sp, c5, ch_, 1, 1)
    g.simple(OP_SPPF_POOL, sp, sp, ch_, dst_coff=ch_)
    conv("9.sppf.cv2", sp, cat20, 4 * ch_, c5, 1, 1, dst_coff=c4)
    # 10-12
    g.simple(OP_UPSAMPLE2X, cat20, cat11, c5, src_coff=c4, dst_coff=0)
    c2f("12.c2f", cat11, 0, c5 + c4, cat17, c3, c4, rep(3), False, H // 16, W // 16)
    # 13-15
    g.simple(OP_UPSAMPLE2X, cat17, cat14, c4, src_coff=c3, dst_coff=0)
    p3 = g.buf(H // 8, W // 8, c3)
    c2f("15.c2f", cat14, 0, c4 + c3, p3, 0, c3, rep(3), False, H // 8, W // 8)
    # 16-18
    conv("16.conv", p3, cat17, c3, c3, 3, 2, dst_coff=0)
    p4 = g.buf(H // 16, W // 16, c4)
    c2f("18.c2f", cat17, 0, c3 + c4, p4, 0, c4, rep(3), False, H // 16, W // 16)
    # 19-21
    conv("19.conv", p4, cat20, c4, c4, 3, 2, dst_coff=0)
    p5 = g.buf(H // 32, W // 32, c5)
    c2f("21.c2f", cat20, 0, c4 + c5, p5, 0, c5, rep(3), False, H // 32, W // 32)
    # 22 Detect
    cb = max(16, c3 // 4, reg_max * 4)
    cc = max(c3, min(nc, 100))
    for lvl, (src, cin, s) in enumerate(((p3, c3, 8), (p4, c4, 16), (p5, c5, 32))):
        h, w = H // s, W // s
        t0, t1 = g.buf(h, w, cb), g.buf(h, w, cb)
        box = g.buf(h, w, 4 * reg_max, DT_F32)
        conv(f"22.box{lvl}.0", src, t0, cin, cb, 3, 1)
        conv(f"22.box{lvl}.1", t0, t1, cb, cb, 3, 1)
        conv(f"22.box{lvl}.2", t1, box, cb, 4 * reg_max, 1, 1, ACT_NONE, gain=1.5)
        u0, u1 = g.buf(h, w, cc), g.buf(h, w, cc)
        cls = g.buf(h, w, nc, DT_F32)
        conv(f"22.cls{lvl}.0", src, u0, cin, cc, 3, 1)
        conv(f"22.cls{lvl}.1", u0, u1, cc, cc, 3, 1)
        conv(f"22.cls{lvl}.2", u1, cls, cc, nc, 1, 1, ACT_NONE)
        g.outputs.append([box, cls, s, h, w, 0, 0, 0])
    g.meta = [nc, reg_max, sum(o[3] * o[4] for o in g.outputs), 0, 0, 0, 0, 0]
    if calibrate:
        calibrate_seeded(g, seed, frame_hw=frame_hw)
    return g


def build_reid(in_hw=(128, 64), seed=1, dim=512, fc=True, calibrate=True) -> Graph:
    """DeepSORT ReID trunk + optional embed FC (D2) + L2 norm."""
    H, W = in_hw
    g = Graph(KIND_REID, H, W)
    wg = _WeightGen(seed)

    def conv(name, src, dst, cin, cout, k, s, act, **kw):
        gain = kw.pop("gain", 1.0)
        g.conv(name, src, dst, cin, cout, k, s, act, wb=wg(cout, cin, k, act, gain), **kw)

    inp = g.buf(H, W, IN_C)
    a = g.buf(H, W, 64)
    conv("conv0", inp, a, 3, 64, 3, 1, ACT_RELU)
    h, w = H // 2, W // 2
    x = g.buf(h, w, 64)
    g.simple(OP_MAXPOOL3S2, a, x, 64)
    cin = 64
    for li, cout in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            ds = li > 1 and bi == 0
            if ds:
                h, w = h // 2, w // 2
            t = g.buf(h, w, cout)
            conv(f"layer{li}.{bi}.conv1", x, t, cin, cout, 3, 2 if ds else 1, ACT_RELU)
            if ds:
                sk = g.buf(h, w, cout)
                conv(f"layer{li}.{bi}.ds", x, sk, cin, cout, 1, 2, ACT_NONE, gain=math.sqrt(0.5))
                skip = (sk, 0)
            else:
                skip = (x, 0)
            y = g.buf(h, w, cout)
            # relu(conv2(t) + skip): both halves scaled so the sum keeps unit variance
            conv(f"layer{li}.{bi}.conv2", t, y, cout, cout, 3, 1, ACT_RELU, gain=0.5, res=skip,
                 res_mode=RES_ADD_THEN_ACT)
            x, cin = y, cout
    pooled = g.buf(1, 1, 512)
    g.simple(OP_AVGPOOL, x, pooled, 512)
    feat = pooled
    if fc:
        f = g.buf(1, 1, dim)
        conv("embed_fc", pooled, f, 512, dim, 1, 1, ACT_NONE, gain=2.0)
        feat = f
    emb = g.buf(1, 1, dim, DT_F32)
    g.simple(OP_L2NORM, feat, emb, dim)
    g.outputs.append([emb, dim, 0, 0, 0, 0, 0, 0])
    g.meta = [dim, 0, 0, 0, 0, 0, 0, 0]
    if calibrate:
        calibrate_seeded(g, seed)
    return g


# ------------------------------------------------------------------------------------ calibration
def _resize_bilinear_f32(img, dh, dw):
    """Plain half-pixel bilinear resize in fp32 (calibration only; the exact u8 spec lives in
    the HIP kernels / the oracle)."""
    sh, sw = img.shape[:2]
    fy = np.clip((np.arange(dh) + 0.5) * sh / dh - 0.5, 0, sh - 1)
    fx = np.clip((np.arange(dw) + 0.5) * sw / dw - 0.5, 0, sw - 1)
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    y1, x1 = np.minimum(y0 + 1, sh - 1), np.minimum(x0 + 1, sw - 1)
    wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
    im = img.astype(np.float32)
    top = im[y0][:, x0] * (1 - wx) + im[y0][:, x1] * wx
    bot = im[y1][:, x0] * (1 - wx) + im[y1][:, x1] * wx
    return top * (1 - wy) + bot * wy


def _calibration_input(g: Graph, seed: int, frame_hw=(720, 1280)):
    """fp32 NCHW calibration batch drawn from the synthetic workload (pure NumPy): a
    letterboxed frame of the resolution the engine is meant for, or 16 resized crops."""
    from . import synthetic
    fh, fw = frame_hw
    if g.kind == KIND_YOLO:
        sc = synthetic.Scene(seed=1000 + seed, n_targets=30, width=fw, height=fh)
        fr = sc.render(0)
        r = min(g.in_h / fh, g.in_w / fw, 1.0)
        uh, uw = int(round(fh * r)), int(round(fw * r))
        if fh == 2 * uh and fw == 2 * uw:
            a = fr.astype(np.int32)
            small = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.float32)
        else:
            small = np.rint(_resize_bilinear_f32(fr, uh, uw))
        img = np.full((g.in_h, g.in_w, 3), 114.0, np.float32)
        top, left = (g.in_h - uh) // 2, (g.in_w - uw) // 2
        img[top:top + uh, left:left + uw] = small
        x = img[:, :, ::-1].transpose(2, 0, 1)[None] / 255.0
        return np.ascontiguousarray(x, dtype=np.float32)
    sc = synthetic.Scene(seed=2000 + seed, n_targets=16, width=fw, height=fh)
    fr = sc.render(0)
    crops = []
    for b in sc.boxes_at(0):
        x1, y1, x2, y2 = (int(v) for v in b)
        c = np.rint(_resize_bilinear_f32(fr[y1:y2, x1:x2], g.in_h, g.in_w))[:, :, ::-1] / 255.0
        c = (c - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
        crops.append(c.transpose(2, 0, 1))
    return np.ascontiguousarray(np.stack(crops), dtype=np.float32)


def calibrate_seeded(g: Graph, seed: int, target_rms: float = 1.0, frame_hw=(720, 1280)):
    """Data-dependent rescale of the seeded weights (LSUV style): walk the ops once on a
    synthetic calibration batch and scale each conv so that what it writes has RMS ~1
    (class logits: unit spread around a bias that lets SEEDED_PASS_FRACTION of the anchors
    pass conf 0.3).  Keeps fp16 activations well inside range and makes the random head
    produce a realistic NMS load.  Offline export-time code (torch CPU), not the hot path.
    Scales are rounded to 4 significant digits so the file does not depend on BLAS threading."""
    import torch
    import torch.nn.functional as F

    x = torch.from_numpy(_calibration_input(g, seed, frame_hw))
    n = x.shape[0]
    bufs = [None] * len(g.buffers)
    bufs[0] = torch.cat([x, torch.zeros(n, g.buffers[0][2] - 3, g.in_h, g.in_w)], 1)
    final_cls = {o[1] for o in g.outputs} if g.kind == KIND_YOLO else set()
    final_box = {o[0] for o in g.outputs} if g.kind == KIND_YOLO else set()

    def put(bi, coff, val):
        h, w, c, _ = g.buffers[bi]
        if bufs[bi] is None:
            bufs[bi] = torch.zeros(n, c, h, w)
        bufs[bi][:, coff:coff + val.shape[1]] = val

    def sig4(v):
        return float(f"{v:.4g}")

    cls_raw = []
    with torch.no_grad():
        for o in g.ops:
            typ, sb, sc_, cin, db, dc, cout, kh, kw, st, pad, act, rb, rc, rmode, wi = o[:16]
            src = bufs[sb][:, sc_:sc_ + cin] if bufs[sb] is not None else None
            if typ == OP_CONV:
                w, b = g.weights[wi]
                wt, bt = torch.from_numpy(w.copy()), torch.from_numpy(b.copy())
                z = F.conv2d(src, wt, None, stride=st, padding=pad)
                res = bufs[rb][:, rc:rc + cout] if rmode else None
                tgt = 1.5 if db in final_box else (1.2 if rmode else target_rms)

                def fwd(s):
                    y = z * s + bt.view(1, -1, 1, 1)
                    if rmode == RES_ADD_THEN_ACT:
                        y = y + res
                    y = F.silu(y) if act == ACT_SILU else (F.relu(y) if act == ACT_RELU else y)
                    if rmode == RES_ACT_THEN_ADD:
                        y = y + res
                    return y
                s = 1.0
                for _ in range(4):
                    y = fwd(s)
                    if db in final_cls:
                        r = float((y - bt.view(1, -1, 1, 1)).std())
                    else:
                        r = float(y.pow(2).mean().sqrt())
                    s = sig4(min(max(s * tgt / max(r, 1e-6), 1e-3), 1e3))
                g.weights[wi] = ((w * np.float32(s)).astype(np.float32), b)
                y = fwd(s)
                if db in final_cls:
                    cls_raw.append((wi, y))
                put(db, dc, y)
            elif typ == OP_SPPF_POOL:
                y = src
                for k in range(3):
                    y = F.max_pool2d(y, 5, 1, 2)
                    put(db, dc + k * cin, y)
            elif typ == OP_UPSAMPLE2X:
                put(db, dc, F.interpolate(src, scale_factor=2, mode="nearest"))
            elif typ == OP_MAXPOOL3S2:
                put(db, dc, F.max_pool2d(src, 3, 2, 1))
            elif typ == OP_AVGPOOL:
                put(db, dc, src.mean(dim=(2, 3), keepdim=True))
            elif typ == OP_L2NORM:
                put(db, dc, src / src.norm(p=2, dim=1, keepdim=True).clamp_min(1e-12))
        if cls_raw:
            best = torch.cat([y.amax(1).flatten() for _, y in cls_raw])
            shift = math.log(0.3 / 0.7) - float(torch.quantile(best, 1.0 - SEEDED_PASS_FRACTION))
            shift = float(f"{shift:.3f}")
            for wi, _ in cls_raw:
                w, b = g.weights[wi]
                g.weights[wi] = (w, (b + np.float32(shift)).astype(np.float32))
    return g


# ------------------------------------------------------------------------------------ file I/O
def serialize(g: Graph) -> bytes:
    head = struct.pack("<II7i8i", MAGIC, VERSION, g.kind, g.in_h, g.in_w, len(g.buffers), len(g.ops),
                       len(g.weights), len(g.outputs), *g.meta)
    parts = [head, np.asarray(g.buffers, dtype="<i4").tobytes(),
             np.asarray(g.ops, dtype="<i4").reshape(-1, OP_WORDS).tobytes()]
    table, off, blobs = [], 0, []
    for w, b in g.weights:
        co, ci, kh, kw = w.shape
        table.append([co, ci, kh, kw, off, off + w.size])
        off += w.size + b.size
        blobs += [np.ascontiguousarray(w, "<f4").tobytes(), np.ascontiguousarray(b, "<f4").tobytes()]
    parts.append(np.asarray(table, dtype="<i8").reshape(-1, 6).tobytes())
    parts.append(np.asarray(g.outputs, dtype="<i4").reshape(-1, 8).tobytes())
    return b"".join(parts + blobs)


def write_engine(path, g: Graph):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(serialize(g))
    os.replace(tmp, path)
    return path


def parse(blob: bytes) -> Graph:
    magic, ver, kind, in_h, in_w, nb, no, nw, nout, *meta = struct.unpack_from("<II7i8i", blob, 0)
    if magic != MAGIC or ver != VERSION:
        raise ValueError("not an AICW engine file")
    off = struct.calcsize("<II7i8i")
    g = Graph(kind, in_h, in_w)
    g.meta = list(meta)
    bufs = np.frombuffer(blob, "<i4", nb * 4, off).reshape(nb, 4); off += nb * 16
    ops = np.frombuffer(blob, "<i4", no * OP_WORDS, off).reshape(no, OP_WORDS); off += no * OP_WORDS * 4
    tab = np.frombuffer(blob, "<i8", nw * 6, off).reshape(nw, 6); off += nw * 48
    outs = np.frombuffer(blob, "<i4", nout * 8, off).reshape(nout, 8); off += nout * 32
    payload = np.frombuffer(blob, "<f4", -1, off)
    g.buffers = [tuple(int(v) for v in r) for r in bufs]
    g.ops = [[int(v) for v in r] for r in ops]
    g.outputs = [[int(v) for v in r] for r in outs]
    for co, ci, kh, kw, wo, bo in tab:
        g.weights.append((payload[wo:wo + co * ci * kh * kw].reshape(co, ci, kh, kw), payload[bo:bo + co]))
    g.names = [f"conv{i}" for i in range(nw)]
    return g


def read_engine(path) -> Graph:
    with open(path, "rb") as f:
        return parse(f.read())


def engine_nms_defaults(path):
    """(conf, iou, max_det) an importer stored in the engine's meta[3..5] (the NMS plugin attributes of an ONNX file with an
    embedded EfficientNMS node, ai-camera_amd/onnx_import.py), or None."""
    with open(path, "rb") as f:
        head = f.read(struct.calcsize("<II7i8i"))
    if len(head) < struct.calcsize("<II7i8i"):
        return None
    meta = struct.unpack("<II7i8i", head)[9:]
    if meta[3] <= 0:
        return None
    conf = struct.unpack("<f", struct.pack("<i", meta[4]))[0]
    iou = struct.unpack("<f", struct.pack("<i", meta[5]))[0]
    return conf, iou, int(meta[3])


def default_engine_paths(root=None):
    """The reference's default locations (src/config.py:12-13) with this build's extension."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return (os.path.join(root, "models/detection/yolov8n.aicw"),
            os.path.join(root, "models/reid/deepsort_reid.aicw"))


def ensure_seeded_engines(root=None, scale="n", yolo_seed=0, reid_seed=1, force=False):
    """Write the seeded engines if absent (there is no network for real weights)."""
    ypath, rpath = default_engine_paths(root)
    if scale != "n":
        ypath = ypath.replace("yolov8n", f"yolov8{scale}")
    if force or not os.path.exists(ypath):
        write_engine(ypath, build_yolov8(scale, seed=yolo_seed))
    if force or not os.path.exists(rpath):
        write_engine(rpath, build_reid(seed=reid_seed))
    return ypath, rpath


TRAINED_ONNX = "weights/yolov8n_synth.onnx"


def ensure_trained_detector(root=None, force=False):
    """The YOLOv8n detector trained on the synthetic workload (tools/train_synthetic_detector.py: class 0 = a planted person of
    ai-camera_amd/synthetic.Scene) as an engine file.  The committed artefact is an ONNX file with fp16 initializers and an embedded
    EfficientNMS node, like the model files the reference downloads (scripts/download_models.sh:7-8); it comes in through the same
    importer as any other ONNX file (onnx_import.onnx_to_engine), so this is the f1 path on weights that matter.  -> engine path."""
    from . import onnx_import
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, TRAINED_ONNX)
    dst = os.path.join(root, "models/detection/yolov8n_synth.aicw")
    if not os.path.exists(src):
        raise FileNotFoundError(f"{src}: the trained detector's ONNX file is missing (tools/train_synthetic_detector.py writes it)")
    if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        g, _ = onnx_import.onnx_to_engine(open(src, "rb").read(), "yolo")
        write_engine(dst, g)
    return dst
