"""HipEngine: the drop-in for the reference's TRTEngine (src/trt_utils/trt_engine.py:15-216).

Same surface -- ``HipEngine(engine_path, device)``, ``infer(dict[name -> torch.Tensor]) ->
dict[name -> torch.Tensor]``, ``__call__`` requiring a dict, ``get_input_details()`` /
``get_output_details()`` returning ``TensorInfo(name, dtype, shape, is_dynamic)`` -- but the engine
file is this build's graph IR + weights and the execution is the hand-written HIP graph executor
behind the C ABI (aic_model_load, aic_yolo_infer, aic_reid_infer).  torch is used only as the
caller-visible tensor container (device memory + dtype), as in the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from pathlib import Path
from typing import Dict, List, NamedTuple, Tuple

import numpy as np

from . import _lib as L
from . import config

TensorInfo = NamedTuple('TensorInfo', [('name', str), ('dtype', object), ('shape', Tuple[int, ...]), ('is_dynamic', bool)])


def _torch():
    import torch
    return torch


class HipEngine:
    #: knobs the reference fixes at trtexec time (scripts/export_trt_engines.sh:37) and inside the
    #: NMS plugin (unpinned); exposed as attributes here
    default_conf, default_iou, default_max_det = config.YOLO_CONF_THRESHOLD, config.YOLO_NMS_THRESHOLD, config.YOLO_MAX_DET

    def __init__(self, engine_path, device=None, dtype="fp16", max_items=None, warm_up=True):
        self.engine_path = Path(engine_path)
        if not self.engine_path.exists():                       # trt_engine.py:46-47
            raise FileNotFoundError(f"Engine file not found: {self.engine_path}")
        self.device_index = config.resolve_device(device)
        self.device = device
        self.dtype = L.F16 if str(dtype).lower() in ("fp16", "f16", "half", "1") else L.F32
        self._h = C.c_void_p()
        kind = self._peek_kind()
        if max_items is None:
            max_items = 8 if kind == L.MODEL_YOLO else 128
        self.max_items = int(max_items)
        L.call("aic_model_load", str(self.engine_path).encode(), self.device_index, self.dtype, self.max_items, C.byref(self._h))
        k, ih, iw, od, na, fl, ncv = (C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_double(), C.c_int())
        L.call("aic_model_info", self._h, C.byref(k), C.byref(ih), C.byref(iw), C.byref(od), C.byref(na), C.byref(fl), C.byref(ncv))
        self.kind, self.in_h, self.in_w, self.out_dim = k.value, ih.value, iw.value, od.value
        self.n_anchors, self.flops_per_item, self.n_convs = na.value, fl.value, ncv.value
        self.conf_thresh, self.iou_thresh, self.max_det = self.default_conf, self.default_iou, self.default_max_det
        if self.kind == L.MODEL_YOLO:                       # an engine imported from an ONNX file with an embedded NMS plugin carries its thresholds
            from .engine_file import engine_nms_defaults
            nms = engine_nms_defaults(str(self.engine_path))
            if nms:
                self.conf_thresh, self.iou_thresh, self.max_det = nms[0], nms[1], min(nms[2], self.default_max_det)
        t = None
        try:
            t = _torch()
        except Exception:   # torch is optional for the C-ABI paths
            pass
        f32 = t.float32 if t else np.float32
        i32 = t.int32 if t else np.int32
        if self.kind == L.MODEL_YOLO:
            self.input_info_list = [TensorInfo("images", f32, (1, 3, self.in_h, self.in_w), False)]
            self.output_info_list = [TensorInfo("num_dets", i32, (1, 1), False),
                                     TensorInfo("bboxes", f32, (1, self.max_det, 4), False),
                                     TensorInfo("scores", f32, (1, self.max_det), False),
                                     TensorInfo("labels", i32, (1, self.max_det), False)]
        else:   # dynamic batch like the reference's ReID profile (export_trt_engines.sh:32-34)
            self.input_info_list = [TensorInfo("input", f32, (-1, 3, self.in_h, self.in_w), True)]
            self.output_info_list = [TensorInfo("output", f32, (-1, self.out_dim), True)]
        if warm_up:
            self._warm_up()

    def _peek_kind(self):
        with open(self.engine_path, "rb") as f:
            head = np.frombuffer(f.read(12), "<u4")
        if len(head) < 3 or head[0] != 0x57434941:
            raise RuntimeError(f"Failed to deserialize engine from {self.engine_path}")   # trt_engine.py:55-56
        return int(head[2])

    def close(self):
        if getattr(self, "_h", None):
            L.call("aic_model_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- raw C-ABI level (NumPy in / NumPy out) ----------------------------------------------
    def yolo_infer_np(self, images_nchw, conf=None, iou=None, max_det=None):
        x = L.as_f32(images_nchw)
        b = x.shape[0]
        md = int(max_det or self.max_det)
        nd = np.zeros(b, np.int32)
        boxes, scores, labels = np.zeros((b, md, 4), np.float32), np.zeros((b, md), np.float32), np.zeros((b, md), np.int32)
        L.call("aic_yolo_infer", self._h, L.ptr(x), b, L.HOST, float(conf if conf is not None else self.conf_thresh),
               float(iou if iou is not None else self.iou_thresh), md, L.ptr(nd), L.ptr(boxes), L.ptr(scores), L.ptr(labels))
        return nd, boxes, scores, labels

    def yolo_head_np(self, images_nchw):
        x = L.as_f32(images_nchw)
        b = x.shape[0]
        dfl = np.zeros((b, self.n_anchors, 64), np.float32)
        cls = np.zeros((b, self.n_anchors, self.out_dim), np.float32)
        L.call("aic_yolo_head", self._h, L.ptr(x), b, L.HOST, L.ptr(dfl), L.ptr(cls))
        return dfl, cls

    def yolo_decode_np(self, images_nchw):
        x = L.as_f32(images_nchw)
        b = x.shape[0]
        boxes, ml = np.zeros((b, self.n_anchors, 4), np.float32), np.zeros((b, self.n_anchors), np.float32)
        lab = np.zeros((b, self.n_anchors), np.int32)
        L.call("aic_yolo_decode", self._h, L.ptr(x), b, L.HOST, L.ptr(boxes), L.ptr(ml), L.ptr(lab))
        return boxes, ml, lab

    def reid_infer_np(self, crops_nchw):
        x = L.as_f32(crops_nchw)
        n = x.shape[0]
        out = np.zeros((n, self.out_dim), np.float32)
        if n:
            L.call("aic_reid_infer", self._h, L.ptr(x), n, L.HOST, L.ptr(out), L.HOST)
        return out

    def detect_np(self, frames_bgr, conf=None, iou=None, max_det=None):
        f = np.ascontiguousarray(frames_bgr, dtype=np.uint8)
        if f.ndim == 3:
            f = f[None]
        b, h, w, _ = f.shape
        md = int(max_det or self.max_det)
        nd = np.zeros(b, np.int32)
        boxes, scores, labels = np.zeros((b, md, 4), np.float32), np.zeros((b, md), np.float32), np.zeros((b, md), np.int32)
        L.call("aic_detect", self._h, L.ptr(f), b, h, w, L.HOST, float(conf if conf is not None else self.conf_thresh),
               float(iou if iou is not None else self.iou_thresh), md, L.ptr(nd), L.ptr(boxes), L.ptr(scores), L.ptr(labels))
        return nd, boxes, scores, labels

    def embed_boxes_np(self, frame_bgr, boxes_xyxy):
        f = np.ascontiguousarray(frame_bgr, dtype=np.uint8)
        b = L.as_f32(boxes_xyxy).reshape(-1, 4)
        n = len(b)
        emb, valid = np.zeros((n, self.out_dim), np.float32), np.zeros(n, np.int32)
        if n:
            L.call("aic_reid_embed", self._h, L.ptr(f), f.shape[0], f.shape[1], L.HOST, L.ptr(b), n, L.ptr(emb), L.ptr(valid))
        return emb, valid

    # ---- TRTEngine surface (torch tensors) -----------------------------------------------------
    def _warm_up(self, iterations=5):   # trt_engine.py:119-149
        start = time.time()
        n = 1
        x = np.zeros((n, 3, self.in_h, self.in_w), np.float32)
        for _ in range(iterations):
            if self.kind == L.MODEL_YOLO:
                self.yolo_infer_np(x)
            else:
                self.reid_infer_np(x)
        print(f"Warm-up for '{self.engine_path.name}' finished in {time.time() - start:.3f} seconds.")

    def infer(self, inputs: Dict[str, "object"]) -> Dict[str, "object"]:
        """trt_engine.py:151-203: validate / cast / make contiguous, run, return fresh output tensors
        owned by the caller (device tensors, like the reference)."""
        torch = _torch()
        info = self.input_info_list[0]
        x = inputs.get(info.name)
        if x is None:
            raise ValueError(f"Missing input: '{info.name}'")                       # trt_engine.py:159-160
        dev = torch.device(f"cuda:{self.device_index}")
        if x.device != dev:
            x = x.to(dev)
        if x.dtype != torch.float32:
            print(f"Warning: Input tensor '{info.name}' dtype mismatch. Expected {torch.float32}, got {x.dtype}. Casting...")
            x = x.to(torch.float32)
        x = x.contiguous()
        torch.cuda.current_stream(dev).synchronize()   # the library runs on its own HIP stream
        n = int(x.shape[0])
        if n > self.max_items:
            raise RuntimeError(f"batch {n} exceeds engine capacity {self.max_items}")
        if self.kind == L.MODEL_YOLO:
            md = self.max_det
            out = {"num_dets": torch.empty((n, 1), dtype=torch.int32, device=dev),
                   "bboxes": torch.empty((n, md, 4), dtype=torch.float32, device=dev),
                   "scores": torch.empty((n, md), dtype=torch.float32, device=dev),
                   "labels": torch.empty((n, md), dtype=torch.int32, device=dev)}
            L.call("aic_yolo_infer", self._h, C.c_void_p(x.data_ptr()), n, L.DEVICE, float(self.conf_thresh),
                   float(self.iou_thresh), md, C.c_void_p(out["num_dets"].data_ptr()), C.c_void_p(out["bboxes"].data_ptr()),
                   C.c_void_p(out["scores"].data_ptr()), C.c_void_p(out["labels"].data_ptr()))
            return out
        out = {"output": torch.empty((n, self.out_dim), dtype=torch.float32, device=dev)}
        if n:
            L.call("aic_reid_infer", self._h, C.c_void_p(x.data_ptr()), n, L.DEVICE, C.c_void_p(out["output"].data_ptr()), L.DEVICE)
        return out

    def __call__(self, inputs):
        if not isinstance(inputs, dict):                                             # trt_engine.py:205-210
            raise TypeError(f"Input to {self.engine_path.name} engine must be a dictionary mapping input names to torch.Tensors.")
        return self.infer(inputs)

    def get_input_details(self) -> List[TensorInfo]:
        return self.input_info_list

    def get_output_details(self) -> List[TensorInfo]:
        return self.output_info_list


TRTEngine = HipEngine   # drop-in name
