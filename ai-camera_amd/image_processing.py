"""Pre/post-processing glue with the signatures of the reference's src/utils/image_processing.py,
executed by the HIP kernels of csrc/kernels_pre.hip (no cv2, no NumPy arithmetic on pixels)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _frame(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or a.shape[0] == 0 or a.shape[1] == 0:
        raise ValueError("expected a non-empty uint8 HxWx3 BGR image")
    return a


def preprocess_yolo_input(image_bgr, target_shape=(640, 640), device=0):
    """image_processing.py:73-102 (letterbox :7-70 with auto=False, scaleup=False inside):
    returns (fp32 [1,3,H,W] RGB/255, (r, r), (pad_w, pad_h))."""
    f = _frame(image_bgr)
    oh, ow = int(target_shape[0]), int(target_shape[1])
    out = np.empty((1, 3, oh, ow), np.float32)
    r, pw, ph = C.c_float(), C.c_float(), C.c_float()
    L.call("aic_letterbox", device, L.ptr(f), f.shape[0], f.shape[1], oh, ow, L.ptr(out), C.byref(r), C.byref(pw), C.byref(ph))
    # the reference returns Python floats computed in double (image_processing.py:40-61)
    h, w = f.shape[:2]
    ratio = min(min(oh / h, 1.0), min(ow / w, 1.0))
    dw = (ow - int(round(w * ratio))) / 2
    dh = (oh - int(round(h * ratio))) / 2
    return out, (ratio, ratio), (dw, dh)


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=False, scaleFill=False, scaleup=False, stride=32):
    """image_processing.py:7-70 for the only mode the hot path uses (auto=False, scaleup=False):
    returns the padded uint8 BGR image, (r, r), (dw, dh)."""
    if auto or scaleFill or scaleup or tuple(color) != (114, 114, 114):
        raise NotImplementedError("only the mode used by preprocess_yolo_input is accelerated")
    t, ratios, pad = preprocess_yolo_input(im, new_shape)
    img = np.rint(t[0] * 255.0).astype(np.uint8)[::-1].transpose(1, 2, 0)   # RGB planes -> BGR HWC
    return np.ascontiguousarray(img), ratios, pad


def preprocess_reid_input(image_crop_bgr, target_shape=(128, 64), device=0):
    """image_processing.py:105-138: fp32 [1,3,H,W], bilinear resize + ImageNet normalisation."""
    f = _frame(image_crop_bgr)
    h, w = f.shape[:2]
    box = np.array([[0, 0, w, h]], np.float32)
    out = np.empty((1, 3, int(target_shape[0]), int(target_shape[1])), np.float32)
    L.call("aic_crop_resize", device, L.ptr(f), h, w, L.ptr(box), 1, int(target_shape[0]), int(target_shape[1]), L.ptr(out), None)
    return out


def crops_from_boxes(frame_bgr, boxes_xyxy, target_shape=(128, 64), device=0):
    """deepsort_tracker.py:143-159 + per-crop preprocess_reid_input in one launch:
    (fp32 [N,3,H,W], valid int32 [N])."""
    f = _frame(frame_bgr)
    b = L.as_f32(boxes_xyxy).reshape(-1, 4)
    n = len(b)
    out = np.zeros((n, 3, int(target_shape[0]), int(target_shape[1])), np.float32)
    valid = np.zeros(n, np.int32)
    if n:
        L.call("aic_crop_resize", device, L.ptr(f), f.shape[0], f.shape[1], L.ptr(b), n, int(target_shape[0]),
               int(target_shape[1]), L.ptr(out), L.ptr(valid))
    return out, valid


def scale_bboxes(bboxes_letterboxed, original_shape, letterbox_shape, ratio, padding):
    """image_processing.py:141-183.  Host-side convenience on <= max_det boxes; inside
    YOLODetector.detect the same arithmetic is fused into the NMS kernel's epilogue."""
    b = np.asarray(bboxes_letterboxed)
    if b.size == 0:
        return np.empty((0, 4), dtype=np.float32)
    out = b.astype(np.float32).copy()
    pad_w, pad_h = padding
    ratio_h, ratio_w = ratio
    out[:, [0, 2]] -= np.float32(pad_w)
    out[:, [1, 3]] -= np.float32(pad_h)
    out[:, [0, 2]] /= np.float32(ratio_w)
    out[:, [1, 3]] /= np.float32(ratio_h)
    oh, ow = original_shape
    out[:, [0, 2]] = np.clip(out[:, [0, 2]], 0, ow)
    out[:, [1, 3]] = np.clip(out[:, [1, 3]], 0, oh)
    return out
