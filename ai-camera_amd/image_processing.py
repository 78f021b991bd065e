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


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleFill=False, scaleup=True, stride=32):
    """The integer / float geometry of image_processing.py:33-67 for every mode, in Python arithmetic like the reference's:
    returns r, (unpad_h, unpad_w), (dw, dh) as the reference returns them, (top, bottom, left, right)."""
    shape = tuple(int(v) for v in shape[:2])
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r_h = new_shape[0] / shape[0]
    r_w = new_shape[1] / shape[1]
    if not scaleup:
        r_h = min(r_h, 1.0)
        r_w = min(r_w, 1.0)
    r = min(r_h, r_w)
    new_unpad = (int(round(shape[0] * r)), int(round(shape[1] * r)))
    dw, dh = new_shape[1] - new_unpad[1], new_shape[0] - new_unpad[0]
    if auto:                                   # :50-51 minimum rectangle
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:                            # :52-54 stretch
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[0], new_shape[1])
    dw /= 2
    dh /= 2
    if shape[::-1] == new_unpad:               # :63 compares (W, H) with (H, W): when it holds the frame is NOT resized
        new_unpad = shape
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return r, new_unpad, (dw, dh), (top, bottom, left, right)


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32, device=0):
    """image_processing.py:7-70, same defaults and every mode: the padded uint8 BGR image, (r, r), (dw, dh).  The geometry is host
    arithmetic (as in the reference); the resize + border is one HIP launch (aic_letterbox_image)."""
    f = _frame(im)
    r, (uh, uw), (dw, dh), (top, bottom, left, right) = letterbox_geometry(f.shape, new_shape, auto, scaleFill, scaleup, stride)
    if min(uh, uw) <= 0:
        raise ValueError("letterbox: the resized image would be empty")       # cv2.resize raises on an empty destination
    if isinstance(color, (int, float)):
        color = (color, color, color)                                         # cv2 would take a scalar as (v, 0, 0, 0); the reference never passes one
    c = [int(v) for v in tuple(color)[:3]] + [0] * (3 - len(tuple(color)[:3]))
    out = np.empty((uh + top + bottom, uw + left + right, 3), np.uint8)
    L.call("aic_letterbox_image", device, L.ptr(f), f.shape[0], f.shape[1], uh, uw, top, bottom, left, right, c[0], c[1], c[2], L.ptr(out))
    return out, (r, r), (dw, dh)


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
