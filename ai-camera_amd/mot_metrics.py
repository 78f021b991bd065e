"""MOT quality harness ("MOT metrics harness for quality regression"; the reference has none -- README.md:209-212
lists evaluation as future work).  CLEAR-MOT (MOTA, FP / FN / ID switches) and the identity measures (IDF1, IDP, IDR) for sequences
whose ground truth is known -- the synthetic scenes of ai-camera_amd/synthetic.py carry an identity per planted box.

Definitions (Bernardin & Stiefelhagen 2008; Ristani et al. 2016), as the MOTChallenge devkit applies them:
  * per frame, ground-truth boxes and tracker outputs are matched one-to-one by maximum total IoU among pairs with IoU >= 0.5,
    keeping a pair matched in the previous frame when it is still valid (CLEAR-MOT continuity rule);
  * FN = unmatched ground truth, FP = unmatched outputs, IDSW = a ground-truth identity matched to a different track id than at
    its previous match; MOTA = 1 - (FN + FP + IDSW) / GT;
  * IDF1: ONE global bipartite assignment of ground-truth identities to track ids maximising the number of frames in which the
    assigned pair overlaps with IoU >= 0.5 (IDTP); IDP = IDTP / outputs, IDR = IDTP / GT, IDF1 = 2 IDTP / (GT + outputs).
The assignment problems go through this package's own rectangular LSAP (csrc/lsap.cpp, host code: no GPU needed)."""
from __future__ import annotations

import numpy as np

from .core.linear_assignment import linear_sum_assignment


def iou_matrix(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1, 4), np.asarray(b, np.float64).reshape(-1, 4)
    if not len(a) or not len(b):
        return np.zeros((len(a), len(b)))
    iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    ua = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ub = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(ua[:, None] + ub[None, :] - inter, 1e-12)


def evaluate(gt_frames, out_frames, iou_thr=0.5):
    """gt_frames[f] = (boxes_xyxy [n,4], identities [n]); out_frames[f] = list of (x1, y1, x2, y2, track_id, ...) tuples.
    -> dict(mota, idf1, idp, idr, fp, fn, idsw, gt, outputs, matches)."""
    fp = fn = idsw = n_gt = n_out = n_match = 0
    last = {}                                  # gt identity -> track id at its previous match
    prev = {}                                  # gt identity -> track id matched in the previous FRAME (continuity rule)
    overlap = {}                               # (gt identity, track id) -> frames with IoU >= thr
    gt_ids, tr_ids = set(), set()
    for (gb, gi), outs in zip(gt_frames, out_frames):
        gb, gi = np.asarray(gb, np.float64).reshape(-1, 4), [int(v) for v in gi]
        ob = np.array([o[:4] for o in outs], np.float64).reshape(-1, 4)
        oi = [int(o[4]) for o in outs]
        n_gt += len(gi)
        n_out += len(oi)
        gt_ids.update(gi), tr_ids.update(oi)
        iou = iou_matrix(gb, ob)
        for a in range(len(gi)):
            for b in range(len(oi)):
                if iou[a, b] >= iou_thr:
                    overlap[(gi[a], oi[b])] = overlap.get((gi[a], oi[b]), 0) + 1
        pairs = {}
        used_g, used_o = set(), set()
        for a, g in enumerate(gi):             # keep last frame's pairs that are still valid
            t = prev.get(g)
            if t is not None and t in oi:
                b = oi.index(t)
                if iou[a, b] >= iou_thr and b not in used_o:
                    pairs[a] = b
                    used_g.add(a), used_o.add(b)
        rg = [a for a in range(len(gi)) if a not in used_g]
        ro = [b for b in range(len(oi)) if b not in used_o]
        if rg and ro:
            cost = np.where(iou[np.ix_(rg, ro)] >= iou_thr, 1.0 - iou[np.ix_(rg, ro)], 1e3)
            rows, cols = linear_sum_assignment(cost)
            for r, c in zip(rows, cols):
                if cost[r, c] < 1e3:
                    pairs[rg[r]] = ro[c]
        prev = {}
        for a, b in pairs.items():
            g, t = gi[a], oi[b]
            if g in last and last[g] != t:
                idsw += 1
            last[g] = t
            prev[g] = t
        n_match += len(pairs)
        fn += len(gi) - len(pairs)
        fp += len(oi) - len(pairs)
    idtp = 0
    if gt_ids and tr_ids:
        G, T = sorted(gt_ids), sorted(tr_ids)
        w = np.zeros((len(G), len(T)))
        for (g, t), c in overlap.items():
            w[G.index(g), T.index(t)] = c
        rows, cols = linear_sum_assignment(-w)
        idtp = int(w[rows, cols].sum())
    return {"mota": 1.0 - (fn + fp + idsw) / max(n_gt, 1), "idf1": 2.0 * idtp / max(n_gt + n_out, 1), "idp": idtp / max(n_out, 1),
            "idr": idtp / max(n_gt, 1), "fp": fp, "fn": fn, "idsw": idsw, "gt": n_gt, "outputs": n_out, "matches": n_match}


def scene_ground_truth(scene, n_frames):
    """Ground truth of a synthetic Scene: the boxes and identities it plants in each frame."""
    return [(lambda d: (d[0], d[3]))(scene.detections(f)) for f in range(n_frames)]
