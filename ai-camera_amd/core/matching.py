"""Cost functions with the signatures of src/tracker/core/matching.py:13-217, computed by the HIP
kernels iou_cost_kernel / cosine_min_kernel (csrc/kernels_trk.hip) through the C ABI."""
import numpy as np

from .. import _lib as L
from .linear_assignment import INFTY_COST

_DEVICE = 0


def iou(bbox_tlwh, candidates_tlwh):
    """IoU of one tlwh box against N candidates (matching.py:13-54)."""
    c = L.as_f32(candidates_tlwh).reshape(-1, 4)
    if c.size == 0:
        return np.array([], dtype=np.float32)
    b = L.as_f32(bbox_tlwh).reshape(1, 4)
    cost = np.empty((1, len(c)), np.float32)
    L.call("aic_iou_cost", _DEVICE, L.ptr(b), 1, L.ptr(c), len(c), L.ptr(cost))
    return (np.float32(1.0) - cost[0]).astype(np.float32)


def iou_cost(tracks, detections, track_indices, detection_indices):
    """1 - IoU matrix [len(track_indices), len(detection_indices)] (matching.py:57-106)."""
    t, n = len(track_indices), len(detection_indices)
    if t == 0 or n == 0:
        return np.empty((t, n), dtype=np.float32)
    tb = np.stack([tracks[i].to_tlwh() for i in track_indices]).astype(np.float32)
    db = np.stack([detections[j].tlwh for j in detection_indices]).astype(np.float32)
    cost = np.empty((t, n), np.float32)
    L.call("aic_iou_cost", _DEVICE, L.ptr(tb), t, L.ptr(db), n, L.ptr(cost))
    return cost


def cosine_distance(features_a, features_b, data_is_normalized=False):
    """Pairwise max(0, 1 - cos) [M, N] (matching.py:109-141). Rows are re-normalised on the device
    either way (a no-op up to rounding for unit rows)."""
    a, b = L.as_f32(features_a), L.as_f32(features_b)
    if a.size == 0 or b.size == 0:
        return np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    m, dim = a.shape
    n = b.shape[0]
    glen = np.ones(m, np.int32)
    cost = np.empty((m, n), np.float32)
    L.call("aic_appearance_cost", _DEVICE, L.ptr(a), L.ptr(glen), m, 1, dim, L.ptr(b), None, n, L.ptr(cost))
    return cost


def appearance_cost_metric(tracks, detections, track_indices, detection_indices, metric_type="cosine"):
    """Min over each track's gallery of the cosine distance to each detection (matching.py:144-217);
    INFTY_COST where the gallery is empty or the detection has no feature."""
    if metric_type != "cosine":
        raise ValueError(f"Unsupported appearance metric_type: {metric_type}")
    t, n = len(track_indices), len(detection_indices)
    if t == 0 or n == 0:
        return np.empty((t, n), dtype=np.float32)
    cost = np.full((t, n), INFTY_COST, dtype=np.float32)
    has = np.array([detections[j].feature is not None for j in detection_indices], np.uint8)
    if not has.any():
        return cost
    dim = next(len(detections[j].feature) for j in detection_indices if detections[j].feature is not None)
    feats = np.zeros((n, dim), np.float32)
    for c, j in enumerate(detection_indices):
        if detections[j].feature is not None:
            feats[c] = detections[j].feature
    glen = np.array([len(tracks[i].features) for i in track_indices], np.int32)
    gmax = int(glen.max())
    if gmax == 0:
        return cost
    gal = np.zeros((t, gmax, dim), np.float32)
    for r, i in enumerate(track_indices):
        if glen[r]:
            gal[r, :glen[r]] = np.asarray(tracks[i].features, np.float32)
    L.call("aic_appearance_cost", _DEVICE, L.ptr(gal), L.ptr(glen), t, gmax, dim, L.ptr(feats), L.ptr(has), n, L.ptr(cost))
    return cost
