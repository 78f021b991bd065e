"""Scenes of the trajectory fixtures (traj8 / traj30 / traj100): shared by make_golden.py (which runs
the reference on them) and by the tests (which replay them through the oracle and the GPU tracker)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
synthetic = importlib.import_module("ai-camera_amd.synthetic")

TRAJ = {
    # name: (Scene kwargs, tracker kwargs, frames, feature dim, featureless detection period)
    "traj30": (dict(seed=11, n_targets=30, jitter=1.5, shuffle=True,
                    gaps=[(3, 40, 52), (7, 60, 140), (12, 100, 103), (12, 110, 180), (20, 30, 31),
                          (25, 200, 299), (5, 150, 222)],
                    births={28: 25, 29: 90, 27: 160}),
               dict(), 300, 512, 0),
    "traj100": (dict(seed=12, n_targets=100, width=1920, height=1080, w_range=(30, 60), h_range=(90, 150),
                     y_range=(50, 850), jitter=1.0, shuffle=True,
                     gaps=[(i, 20 + i, 25 + 2 * i) for i in range(0, 40, 3)], births={90 + i: 10 * i for i in range(10)}),
                dict(), 120, 128, 0),
    "traj8": (dict(seed=13, n_targets=8, width=640, height=480, w_range=(30, 60), h_range=(60, 120),
                   y_range=(20, 300), jitter=2.0, shuffle=True,
                   gaps=[(0, 10, 14), (1, 20, 27), (2, 5, 5), (3, 30, 60), (4, 12, 13), (4, 16, 17), (6, 40, 47)],
                   births={7: 33}),
              dict(max_age=5, n_init=2, nn_budget=4, max_cosine_distance=0.25, max_iou_distance=0.8), 90, 32, 7),
}


def scene_inputs(name, f):
    kw, _, _, dim, featless = TRAJ[name]
    sc = scene_inputs.cache.setdefault(name, synthetic.Scene(**kw))
    boxes, conf, cls, ids = sc.detections(f)
    feats = synthetic.identity_features(ids, f, dim=dim, seed=kw["seed"])
    tlwh = boxes.copy()
    tlwh[:, 2:] -= tlwh[:, :2]
    has = np.ones(len(ids), bool)
    if featless:
        has[(np.arange(len(ids)) + f) % featless == 0] = False
    return tlwh.astype(np.float32), conf, ids, feats, has


scene_inputs.cache = {}


