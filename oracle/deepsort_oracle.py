"""NumPy/SciPy restatement of the reference DeepSORT core (TEST INFRASTRUCTURE).

Follows, function by function, the arithmetic of
``/root/reference/src/tracker/core/*`` and the wrapper logic of
``src/tracker/deepsort_tracker.py``.  Written as flat functions over arrays
(not a copy of the reference's classes) so that each piece is the direct
counterpart of one HIP kernel / one C-ABI entry point.

Pinned against the imported reference core by ``tests/golden/make_golden.py``
(fixtures ``tests/golden/*.npz``) and by the known-answer values of the
reference's ``__main__`` self-tests (``tests/test_oracle_tracker.py``).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

# src/tracker/core/linear_assignment.py:9
INFTY_COST = 1e5
# src/tracker/core/kalman_filter.py:12-22 (only the 2- and 4-dof entries are used)
CHI2INV95 = {1: 3.841458820694124, 2: 5.991464547107979, 3: 7.814727903251179,
             4: 9.487729036781154, 5: 11.070497693516351, 6: 12.591587243743977,
             7: 14.067140449349192, 8: 15.50731305586545, 9: 16.918977604620448}

TENTATIVE, CONFIRMED, DELETED = 1, 2, 3  # src/tracker/core/track.py:10-14

_STD_POS = 1.0 / 20    # kalman_filter.py:52
_STD_VEL = 1.0 / 160   # kalman_filter.py:53

_F = np.eye(8, dtype=np.float32)   # kalman_filter.py:42-44 (dt = 1)
for _i in range(4):
    _F[_i, 4 + _i] = 1.0
_H = np.eye(4, 8, dtype=np.float32)  # kalman_filter.py:47


# --------------------------------------------------------------------------- boxes
def tlwh_to_xyah(tlwh):
    """detection.py:36-47 -- centre, aspect (0 when h<=0), height; fp32."""
    t = np.asarray(tlwh, dtype=np.float32).copy()
    t[:2] += t[2:] / 2.0
    t[2] = t[2] / t[3] if t[3] > 0 else 0
    return t


def mean_to_tlwh(mean):
    """track.py:133-151 -- w = a*h when h>0 else 0, h clamped at 0."""
    p = np.asarray(mean[:4], dtype=np.float32).copy()
    if p[3] > 0:
        w = p[2] * p[3]
    else:
        w = 0
        p[3] = max(0, p[3])
    return np.array([p[0] - w / 2.0, p[1] - p[3] / 2.0, w, p[3]], dtype=np.float32)


# --------------------------------------------------------------------------- Kalman
def kf_initiate(z):
    """kalman_filter.py:55-83."""
    z = np.asarray(z)
    mean = np.concatenate((z, np.zeros_like(z, dtype=np.float32)))
    h = z[3]
    std = [2 * _STD_POS * h, 2 * _STD_POS * h, 1e-2, 2 * _STD_POS * h,
           10 * _STD_VEL * h, 10 * _STD_VEL * h, 1e-5, 10 * _STD_VEL * h]
    return mean, np.diag(np.square(std)).astype(np.float32)


def motion_matrix(dt=1.0):
    """kalman_filter.py:41-44 -- identity with fp32(dt) on the position/velocity diagonal."""
    f = np.eye(8, dtype=np.float32)
    for i in range(4):
        f[i, 4 + i] = dt
    return f


def kf_predict(mean, cov, dt=1.0):
    """kalman_filter.py:85-120 -- x<-Fx, P<-F P F^T + Q(h); F from KalmanFilter(dt) (:34-44)."""
    _F = motion_matrix(dt)
    h = mean[3]
    std = [_STD_POS * h, _STD_POS * h, 1e-2, _STD_POS * h,
           _STD_VEL * h, _STD_VEL * h, 1e-5, _STD_VEL * h]
    q = np.diag(np.square(np.asarray(std))).astype(np.float32)
    return np.dot(_F, mean), np.linalg.multi_dot((_F, cov, _F.T)) + q


def kf_project(mean, cov):
    """kalman_filter.py:122-151 -- S = H P H^T + R(h)."""
    h = mean[3]
    std = [_STD_POS * h, _STD_POS * h, 1e-1, _STD_POS * h]
    r = np.diag(np.square(std)).astype(np.float32)
    return np.dot(_H, mean), np.linalg.multi_dot((_H, cov, _H.T)) + r


def kf_update(mean, cov, z):
    """kalman_filter.py:153-204 -- Cholesky gain, P - K S K^T form."""
    pm, s = kf_project(mean, cov)
    cf = scipy.linalg.cho_factor(s, lower=True, check_finite=False)
    k = scipy.linalg.cho_solve(cf, np.dot(cov, _H.T).T, check_finite=False).T
    return mean + np.dot(k, z - pm), cov - np.linalg.multi_dot((k, s, k.T))


def kf_gating_distance(mean, cov, zs, only_position=False):
    """kalman_filter.py:206-249 -- squared Mahalanobis distance to each row of zs."""
    pm, s = kf_project(mean, cov)
    if only_position:
        pm, s, zs = pm[:2], s[:2, :2], zs[:, :2]
    d = zs - pm
    try:
        c, _ = scipy.linalg.cho_factor(s, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        # kalman_filter.py:241-247 -- S not positive definite (e.g. h == 0): reject everything
        return np.full(zs.shape[0], np.inf, dtype=np.float32)
    y = scipy.linalg.solve_triangular(c, d.T, lower=True, check_finite=False)
    return np.sum(y * y, axis=0)


# --------------------------------------------------------------------------- costs
def iou_one_to_many(box, cands):
    """matching.py:13-54 (tlwh, union floored at 1e-7)."""
    if cands.size == 0:
        return np.array([], dtype=np.float32)
    tl, br = box[:2], box[:2] + box[2:]
    ctl, cbr = cands[:, :2], cands[:, :2] + cands[:, 2:]
    w = np.maximum(0., np.minimum(br[0], cbr[:, 0]) - np.maximum(tl[0], ctl[:, 0]))
    h = np.maximum(0., np.minimum(br[1], cbr[:, 1]) - np.maximum(tl[1], ctl[:, 1]))
    inter = w * h
    union = box[2] * box[3] + cands[:, 2] * cands[:, 3] - inter
    return inter / np.maximum(union, 1e-7)


def iou_cost_matrix(track_tlwh, det_tlwh):
    """matching.py:57-106 -- 1 - IoU, [T,N] fp32."""
    t, n = len(track_tlwh), len(det_tlwh)
    if t == 0 or n == 0:
        return np.empty((t, n), dtype=np.float32)
    out = np.full((t, n), INFTY_COST, dtype=np.float32)
    d = np.asarray(det_tlwh, dtype=np.float32)
    for i in range(t):
        out[i] = 1.0 - iou_one_to_many(track_tlwh[i], d)
    return out


def cosine_distance(a, b, normalized=False):
    """matching.py:109-141."""
    if a.size == 0 or b.size == 0:
        return np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    if not normalized:
        a = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-7)
        b = b / np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-7)
    return np.maximum(1.0 - np.dot(a, b.T), 0.0)


def appearance_cost_matrix(galleries, det_feats):
    """matching.py:144-217 -- per track min over its gallery of the cosine distance.

    galleries: list (len T) of [G_t,D] arrays (G_t may be 0);
    det_feats: list (len N) of [D] arrays or None.
    """
    t, n = len(galleries), len(det_feats)
    if t == 0 or n == 0:
        return np.empty((t, n), dtype=np.float32)
    out = np.full((t, n), INFTY_COST, dtype=np.float32)
    cols = [j for j in range(n) if det_feats[j] is not None]
    if not cols:
        return out
    d = np.asarray([det_feats[j] for j in cols], dtype=np.float32)
    for i, g in enumerate(galleries):
        if len(g) == 0:
            continue
        out[i, cols] = np.min(cosine_distance(np.asarray(g, dtype=np.float32), d), axis=0)
    return out


def gate_by_mahalanobis(cost, means, covs, det_xyah, only_position=False):
    """linear_assignment.py:160-212 -- entries with d2 > chi2(4) -> INFTY_COST (in place)."""
    thr = CHI2INV95[2 if only_position else 4]
    z = np.asarray(det_xyah)
    for i in range(len(means)):
        if not z.size:
            cost[i, :] = INFTY_COST
            continue
        cost[i, kf_gating_distance(means[i], covs[i], z, only_position) > thr] = INFTY_COST
    return cost


# --------------------------------------------------------------------------- assignment
def threshold_and_assign(cost, max_distance, rows, cols):
    """linear_assignment.py:55-88 -- clamp, SciPy LSAP, accept iff cost <= max."""
    if not rows or not cols:
        return [], list(rows), list(cols)
    cost = cost.copy()
    cost[cost > max_distance] = max_distance + 1e-5
    ri, ci = linear_sum_assignment(cost)
    matches, ur, uc = [], list(rows), list(cols)
    for r, c in zip(ri, ci):
        if cost[r, c] <= max_distance:
            matches.append((rows[r], cols[c]))
            ur.remove(rows[r])
            uc.remove(cols[c])
    return matches, ur, uc


def cascade_on_matrices(full_app, full_gate, full_iou, state, tsu, max_cosine_distance, max_iou_distance, max_age):
    """linear_assignment.py:91-157 + tracker_core.py:83-177 on the full [T,N] cost matrices of one frame.
    Returns (matches [(track, det)], unmatched tracks, unmatched dets) in the reference's order."""
    n_t, n = len(state), (full_app.shape[1] if len(state) else full_iou.shape[1] if full_iou.ndim == 2 else 0)
    confirmed = [i for i in range(n_t) if state[i] == CONFIRMED]
    tentative = [i for i in range(n_t) if state[i] == TENTATIVE]
    # stage 1: matching cascade, linear_assignment.py:91-157
    unmatched_d = list(range(n))
    matches = []
    for level in range(max_age):
        if not unmatched_d:
            break
        rows = [i for i in confirmed if tsu[i] == level + 1]
        if not rows:
            continue
        c = full_app[np.ix_(rows, unmatched_d)].copy()
        c[full_gate[np.ix_(rows, unmatched_d)] > CHI2INV95[4]] = INFTY_COST
        m, _, unmatched_d = threshold_and_assign(c, max_cosine_distance, rows, unmatched_d)
        matches += m
    got = {i for i, _ in matches}
    unmatched_confirmed = [i for i in confirmed if i not in got]
    # stage 2: IoU on tentative + just-missed confirmed, tracker_core.py:138-166
    cand = tentative + [i for i in unmatched_confirmed if tsu[i] == 1]
    stale = [i for i in unmatched_confirmed if tsu[i] > 1]
    if cand and unmatched_d:
        c = full_iou[np.ix_(cand, unmatched_d)]
        m2, ut, unmatched_d = threshold_and_assign(c, max_iou_distance, cand, unmatched_d)
    else:
        m2, ut = [], cand
    return matches + m2, stale + ut, unmatched_d


class OracleTrack:
    """Plain record with the attribute surface of track.py:16-171."""
    __slots__ = ("track_id", "mean", "covariance", "class_name", "confidence", "hits", "age",
                 "time_since_update", "state", "features")

    def is_confirmed(self):
        return self.state == CONFIRMED

    def is_tentative(self):
        return self.state == TENTATIVE

    def to_tlwh(self):
        return mean_to_tlwh(self.mean)


class OracleTracker:
    """tracker_core.py:11-198 + track.py lifecycle, with a per-instance id counter."""

    def __init__(self, max_cosine_distance=0.2, nn_budget=100, max_iou_distance=0.7,
                 max_age=70, n_init=3):
        self.max_cosine_distance = max_cosine_distance
        self.nn_budget = nn_budget
        self.max_iou_distance = max_iou_distance
        self.max_age = max_age
        self.n_init = n_init
        self.tracks: list[OracleTrack] = []
        self.next_id = 1
        self.last_matches = []
        self.last_costs = None

    # tracker_core.py:44-49 -> track.py:76-80
    def predict(self):
        for t in self.tracks:
            t.mean, t.covariance = kf_predict(t.mean, t.covariance)
            t.age += 1
            t.time_since_update += 1

    # tracker_core.py:83-177
    def _match(self, det_tlwh, det_feats):
        n = len(det_tlwh)
        tr = self.tracks
        det_xyah = [tlwh_to_xyah(b) for b in det_tlwh]
        confirmed = [i for i, t in enumerate(tr) if t.state == CONFIRMED]
        tentative = [i for i, t in enumerate(tr) if t.state == TENTATIVE]
        # full matrices once (entry-wise independent; the reference recomputes sub-blocks,
        # linear_assignment.py:143-150 -> tracker_core.py:93-109, with identical values)
        full_app = appearance_cost_matrix([t.features for t in tr], det_feats)
        full_gate = np.zeros((len(tr), n), dtype=np.float32)
        if len(tr) and n:
            z = np.asarray(det_xyah)
            for i, t in enumerate(tr):
                full_gate[i] = kf_gating_distance(t.mean, t.covariance, z)
        full_iou = iou_cost_matrix([t.to_tlwh() for t in tr], det_tlwh)
        self.last_costs = (full_app, full_gate, full_iou)

        return cascade_on_matrices(full_app, full_gate, full_iou, [t.state for t in tr], [t.time_since_update for t in tr],
                                   self.max_cosine_distance, self.max_iou_distance, self.max_age)

    # tracker_core.py:51-81
    def update(self, det_tlwh, det_conf, det_class, det_feats):
        det_tlwh = [np.asarray(b, dtype=np.float32) for b in det_tlwh]
        det_feats = [None if f is None else np.asarray(f, dtype=np.float32) for f in det_feats]
        matches, unmatched_t, unmatched_d = self._match(det_tlwh, det_feats)
        self.last_matches = [(self.tracks[i].track_id, j) for i, j in matches]
        for i, j in matches:                       # track.py:82-104
            t = self.tracks[i]
            t.mean, t.covariance = kf_update(t.mean, t.covariance, tlwh_to_xyah(det_tlwh[j]))
            if det_feats[j] is not None:
                self._add_feature(t, det_feats[j])
            t.hits += 1
            t.time_since_update = 0
            t.confidence = float(det_conf[j])
            t.class_name = det_class[j]
            if t.state == TENTATIVE and t.hits >= self.n_init:
                t.state = CONFIRMED
        for i in unmatched_t:                      # track.py:106-119
            t = self.tracks[i]
            if t.state == TENTATIVE:
                t.state = DELETED
            elif t.state == CONFIRMED and t.time_since_update > self.max_age:
                t.state = DELETED
        for j in unmatched_d:                      # tracker_core.py:180-194, track.py:23-67
            t = OracleTrack()
            t.mean, t.covariance = kf_initiate(tlwh_to_xyah(det_tlwh[j]))
            t.track_id = self.next_id
            self.next_id += 1
            t.class_name, t.confidence = det_class[j], float(det_conf[j])
            t.hits, t.age, t.time_since_update, t.state = 1, 1, 0, TENTATIVE
            t.features = []
            if det_feats[j] is not None:
                self._add_feature(t, det_feats[j])
            self.tracks.append(t)
        self.tracks = [t for t in self.tracks if t.state != DELETED]

    def _add_feature(self, t, f):                  # track.py:70-74 (FIFO)
        t.features.append(f)
        if self.nn_budget is not None and len(t.features) > self.nn_budget:
            t.features.pop(0)

    # deepsort_tracker.py:126-141
    def output_tuples(self):
        out = []
        self.last_output_float = []            # the four fp32 coordinates BEFORE int(round(.)), same order: what a test needs to tell a
        for t in self.tracks:                  # legitimate one-pixel flip (coordinate on a rounding edge) from a wrong box
            if t.state == CONFIRMED and t.time_since_update == 0:
                x1, y1, w, h = t.to_tlwh()
                w, h = max(0, w), max(0, h)
                out.append((int(round(x1)), int(round(y1)), int(round(x1 + w)), int(round(y1 + h)),
                            t.track_id, t.class_name, float(t.confidence)))
                self.last_output_float.append((float(x1), float(y1), float(x1 + w), float(y1 + h)))
        return out


def filter_detections(boxes_xyxy, confs, class_ids, classes, classes_to_track, min_conf):
    """deepsort_tracker.py:88-101 -- order-preserving conf/class filter; returns kept indices."""
    keep = []
    for i in range(len(boxes_xyxy)):
        cid = int(class_ids[i])
        name = classes[cid] if 0 <= cid < len(classes) else "Unknown"
        if confs[i] >= min_conf and name in classes_to_track:
            keep.append(i)
    return keep


def crop_rect(box_xyxy, frame_h, frame_w):
    """deepsort_tracker.py:143-159 -- int() truncation then clamp; None when empty."""
    x1, y1, x2, y2 = map(int, box_xyxy)
    x1, y1, x2, y2 = max(0, x1), max(0, y1), min(frame_w, x2), min(frame_h, y2)
    if x1 < x2 and y1 < y2:
        return x1, y1, x2, y2
    return None
