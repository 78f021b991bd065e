"""NumPy restatement of the reference's pre/post-processing glue (TEST INFRASTRUCTURE).

Formulas follow ``/root/reference/src/utils/image_processing.py`` and
``src/tracker/deepsort_tracker.py:143-159`` as text (they import cv2, which is absent here).

``cv2.resize`` is a third-party dependency (opencv-python==4.11.0.86, requirements.txt:3)
that is not under /root/reference and not installed: its 8-bit INTER_LINEAR algorithm is
RESTATED here from OpenCV's published source (modules/imgproc/src/resize.cpp):

* half-pixel centres, ``fx = (float)((dx+0.5)*scale - 0.5)``, ``sx = floor(fx)``, taps clamped
  at the borders, 11-bit fixed-point coefficients ``cvRound(c*2048)`` (round-half-even);
* horizontal pass in int32 (``S[sx]*a0 + S[sx+1]*a1``), vertical pass
  ``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``;
* when both scales are exactly 2, INTER_LINEAR is replaced by the 2x2 area average
  ``(a+b+c+d+2)>>2``.

PARITY UNPINNED for the cv2 bit pattern (no cv2 here, no fixture in the reference); this
restated spec is the contract the HIP kernels are held to, bit-exactly.
"""
from __future__ import annotations

import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)   # image_processing.py:129
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)    # image_processing.py:130


def _coeffs(dst, src):
    """Tap index, and the two 11-bit weights, per destination coordinate."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f, scale


def resize_linear_u8(img: np.ndarray, dst_h: int, dst_w: int) -> np.ndarray:
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for uint8 HWC."""
    src_h, src_w = img.shape[:2]
    sx, fx, scale_x = _coeffs(dst_w, src_w)
    sy, fy, scale_y = _coeffs(dst_h, src_h)
    eps = np.finfo(np.float64).eps
    if abs(scale_x - 2) < eps and abs(scale_y - 2) < eps and int(scale_x) == 2 and int(scale_y) == 2:
        a = img.astype(np.int32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    # horizontal taps: clamp with the weight forced to the surviving tap
    lo, hi = sx < 0, sx >= src_w - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, src_w - 1, sx))
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int32)
    sx1 = np.minimum(sx + 1, src_w - 1)
    # vertical taps: rows clipped, weights kept
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int32)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int32)
    r0 = np.clip(sy, 0, src_h - 1)
    r1 = np.clip(sy + 1, 0, src_h - 1)
    src = img.astype(np.int32)
    hrow = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]     # [src_h, dst_w, C]
    s0, s1 = hrow[r0], hrow[r1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def letterbox_geometry(h, w, new_shape=(640, 640)):
    """image_processing.py:33-67 with auto=False, scaleup=False (as called at :92)."""
    r = min(min(new_shape[0] / h, 1.0), min(new_shape[1] / w, 1.0))
    unpad_h, unpad_w = int(round(h * r)), int(round(w * r))
    dw, dh = (new_shape[1] - unpad_w) / 2, (new_shape[0] - unpad_h) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return r, (unpad_h, unpad_w), (dw, dh), (top, bottom, left, right)


def letterbox_u8(frame_bgr, new_shape=(640, 640), color=114):
    """image_processing.py:7-70 -> (padded u8 BGR image, (r, r), (dw, dh))."""
    h, w = frame_bgr.shape[:2]
    r, (uh, uw), (dw, dh), (top, bottom, left, right) = letterbox_geometry(h, w, new_shape)
    im = resize_linear_u8(frame_bgr, uh, uw)
    out = np.full((uh + top + bottom, uw + left + right, 3), color, dtype=np.uint8)
    out[top:top + uh, left:left + uw] = im
    return out, (r, r), (dw, dh)


def letterbox_any(frame_bgr, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """image_processing.py:7-70 with the reference's own defaults and every mode, followed line by line (:33-70)."""
    shape = frame_bgr.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r_h, r_w = new_shape[0] / shape[0], new_shape[1] / shape[1]
    if not scaleup:
        r_h, r_w = min(r_h, 1.0), min(r_w, 1.0)
    r = min(r_h, r_w)
    new_unpad = (int(round(shape[0] * r)), int(round(shape[1] * r)))
    dw, dh = new_shape[1] - new_unpad[1], new_shape[0] - new_unpad[0]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[0], new_shape[1])
    dw /= 2
    dh /= 2
    im = frame_bgr
    if tuple(shape[::-1]) != tuple(new_unpad):
        im = resize_linear_u8(frame_bgr, new_unpad[0], new_unpad[1])
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, 3), np.uint8)
    out[:] = np.asarray(color, np.uint8)[:3]
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return out, (r, r), (dw, dh)


def preprocess_yolo_input(frame_bgr, target_shape=(640, 640)):
    """image_processing.py:73-102 -> (fp32 [1,3,H,W] RGB /255, ratios, (pad_w, pad_h))."""
    img, ratios, pad = letterbox_u8(frame_bgr, target_shape)
    chw = np.transpose(img[:, :, ::-1], (2, 0, 1))
    return np.ascontiguousarray(chw[None].astype(np.float32) / 255.0), ratios, pad


def preprocess_reid_input(crop_bgr, target_shape=(128, 64)):
    """image_processing.py:105-138 -> fp32 [1,3,H,W]."""
    rs = resize_linear_u8(crop_bgr, target_shape[0], target_shape[1])
    rgb = rs[:, :, ::-1]
    norm = (rgb.astype(np.float32) / 255.0 - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(np.transpose(norm, (2, 0, 1))[None], dtype=np.float32)


def crops_to_batch(frame_bgr, boxes_xyxy, target_shape=(128, 64)):
    """deepsort_tracker.py:143-159 + reid_model.py:84-101: (fp32 [N,3,H,W], valid [N]).
    Invalid (empty) crops leave a zero tensor and valid=0."""
    fh, fw = frame_bgr.shape[:2]
    n = len(boxes_xyxy)
    out = np.zeros((n, 3, target_shape[0], target_shape[1]), np.float32)
    valid = np.zeros(n, np.int32)
    for i, b in enumerate(boxes_xyxy):
        x1, y1, x2, y2 = map(int, b)
        x1, y1, x2, y2 = max(0, x1), max(0, y1), min(fw, x2), min(fh, y2)
        if x1 < x2 and y1 < y2:
            out[i] = preprocess_reid_input(frame_bgr[y1:y2, x1:x2], target_shape)[0]
            valid[i] = 1
    return out, valid


def scale_bboxes(boxes, original_shape, ratio, padding):
    """image_processing.py:141-183 -- un-letterbox and clip, fp32."""
    if boxes.size == 0:
        return np.empty((0, 4), dtype=np.float32)
    b = boxes.astype(np.float32).copy()
    pad_w, pad_h = padding
    ratio_h, ratio_w = ratio
    b[:, 0] -= pad_w
    b[:, 1] -= pad_h
    b[:, 2] -= pad_w
    b[:, 3] -= pad_h
    b[:, 0] /= ratio_w
    b[:, 1] /= ratio_h
    b[:, 2] /= ratio_w
    b[:, 3] /= ratio_h
    oh, ow = original_shape
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, ow)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, oh)
    return b
