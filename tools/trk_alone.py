#!/usr/bin/env python3
"""The association epochs ALONE on an idle GPU (no conv kernels beside them): 30 persons, 512-d features, galleries at the 100-row
budget, epochs of 16 frames through aic_tracker_update_batch.  Under rocprofv3 --kernel-trace --stats: what trk_epoch_prep_kernel and
trk_epoch_kernel cost when nothing competes for their CUs."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("ai-camera_amd.synthetic")
TC = importlib.import_module("ai-camera_amd.core.tracker_core").TrackerCore
n_t, dim, K = 30, 512, 16
sc = syn.Scene(seed=0, n_targets=n_t)
trk = TC()
trk.option("device_assoc", 2)
rng = np.random.default_rng(0)
def frame(f):
    boxes, conf, cls, ids = sc.detections(f)
    feats = syn.identity_features(ids, f, dim=dim, seed=5)
    tlwh = np.stack([boxes[:, 0], boxes[:, 1], boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).astype(np.float32)
    return (tlwh, conf.astype(np.float32), np.zeros(len(ids), np.int32), feats.astype(np.float32), np.ones(len(ids), np.uint8))
f = 0
for e in range(12):                       # fill the galleries
    trk.update_batch([frame(f + i) for i in range(K)]); f += K
t0 = time.perf_counter()
n = 60
for e in range(n):
    trk.update_batch([frame(f + i) for i in range(K)]); f += K
dt = time.perf_counter() - t0
print(f"{n} epochs of {K} frames alone: {1e6 * dt / n:.0f} us per epoch (host side included)")
