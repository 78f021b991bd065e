"""Assignment helpers with the signatures of src/tracker/core/linear_assignment.py:19-212.
Thresholding + the rectangular LSAP run in host C++ (csrc/lsap.cpp), gating distances on the GPU."""
import numpy as np

from .. import _lib as L
from .kalman_filter import CHI2INV95

INFTY_COST = 1e5   # linear_assignment.py:9


def linear_sum_assignment(cost_matrix):
    """scipy.optimize.linear_sum_assignment replacement (same optimum, same tie-breaking)."""
    c = np.ascontiguousarray(cost_matrix, dtype=np.float64)
    if c.ndim != 2:
        raise ValueError("expected a matrix (2-D array)")
    nr, nc = c.shape
    k = min(nr, nc)
    rows, cols = np.zeros(k, np.int64), np.zeros(k, np.int64)
    try:
        L.call("aic_lsap", L.ptr(c), nr, nc, L.ptr(rows), L.ptr(cols))
    except L.AicError as e:
        raise ValueError(str(e)) from None
    return rows, cols


def min_cost_matching(distance_metric, max_distance, tracks, detections, track_indices=None, detection_indices=None):
    """linear_assignment.py:19-88."""
    if track_indices is None:
        track_indices = list(range(len(tracks)))
    if detection_indices is None:
        detection_indices = list(range(len(detections)))
    if not detection_indices or not track_indices:
        return [], track_indices, detection_indices
    cost = np.ascontiguousarray(distance_metric(tracks, detections, track_indices, detection_indices), dtype=np.float32)
    nr, nc = cost.shape
    mr, mc, nm = np.zeros(min(nr, nc), np.int32), np.zeros(min(nr, nc), np.int32), np.zeros(1, np.int32)
    L.call("aic_min_cost_matching", L.ptr(cost), nr, nc, float(max_distance), L.ptr(mr), L.ptr(mc), L.ptr(nm))
    matches = [(track_indices[r], detection_indices[c]) for r, c in zip(mr[:nm[0]], mc[:nm[0]])]
    mt, md = {m[0] for m in matches}, {m[1] for m in matches}
    return (matches, [t for t in track_indices if t not in mt], [d for d in detection_indices if d not in md])


def matching_cascade(distance_metric, max_distance, cascade_depth, tracks, detections, track_indices=None,
                     detection_indices=None, kf=None, gated_cost_weight=1.0):
    """linear_assignment.py:91-157: levels by time_since_update, shrinking detection set."""
    if track_indices is None:
        track_indices = list(range(len(tracks)))
    if detection_indices is None:
        detection_indices = list(range(len(detections)))
    unmatched = list(detection_indices)
    matches = []
    for level in range(cascade_depth):
        if not unmatched:
            break
        rows = [i for i in track_indices if tracks[i].time_since_update == level + 1]
        if not rows:
            continue
        m, _, unmatched = min_cost_matching(distance_metric, max_distance, tracks, detections, rows, unmatched)
        matches.extend(m)
    got = {t for t, _ in matches}
    return matches, [i for i in track_indices if i not in got], unmatched


def gate_cost_matrix_by_mahalanobis(kf, cost_matrix, tracks, detections, track_indices, detection_indices,
                                    only_position=False, gating_threshold_override=None):
    """linear_assignment.py:160-212: entries whose squared Mahalanobis distance exceeds the chi-square
    95% quantile become INFTY_COST (in place)."""
    thr = gating_threshold_override if gating_threshold_override is not None else CHI2INV95.get(2 if only_position else 4, INFTY_COST)
    if not len(track_indices):
        return cost_matrix
    if not len(detection_indices):
        cost_matrix[:, :] = INFTY_COST
        return cost_matrix
    z = np.stack([detections[j].to_xyah() for j in detection_indices]).astype(np.float32)
    mean = np.stack([tracks[i].mean for i in track_indices]).astype(np.float32)
    cov = np.stack([tracks[i].covariance for i in track_indices]).astype(np.float32)
    d2 = np.empty((len(track_indices), len(z)), np.float32)
    L.call("aic_kf_gating", kf.device, L.ptr(mean), L.ptr(cov), len(mean), L.ptr(z), len(z), 1, int(bool(only_position)), L.ptr(d2))
    cost_matrix[d2 > np.float32(thr)] = INFTY_COST
    return cost_matrix
