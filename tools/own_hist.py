#!/usr/bin/env python3
"""Distribution of the own-detections leg's per-frame problem sizes (bench.py's inject=0 scene): detections above the tracker floor per
frame, the maximum per 16-frame epoch and per 512-frame launch group -- what the device/host choice of the association sees."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
cfg = importlib.import_module("ai-camera_amd.config")
TP = importlib.import_module("ai-camera_amd.pipeline").TrackingPipeline

R = 1024
ypath, rpath = ef.ensure_seeded_engines(ROOT)
sc = syn.Scene(seed=0, n_targets=30)
frames = sc.render_batch(0, R)
cfg.CLASSES_TO_TRACK.clear()
cfg.CLASSES_TO_TRACK.update(cfg.CLASSES)
p = TP(ypath, rpath, (720, 1280), batch=64, ring_frames=R, max_persons=64, dtype="fp16", inject=False, min_confidence=0.999999, max_tracks=512)
p.upload(0, frames)
_, dets = p.run(0, R, want_dets=True)
sc_all = np.sort(np.concatenate([d[1] for d in dets[:64]]))[::-1]
floor = float(sc_all[min(len(sc_all) - 1, 64 * 30)])
n = np.array([int((d[1] >= floor).sum()) for d in dets])
print("floor", floor, "mean", n.mean(), "max", n.max())
print("per-frame histogram (edges 0,16,32,64,96,128,192,256,301):", np.histogram(n, [0, 16, 32, 64, 96, 128, 192, 256, 301])[0].tolist())
e = n.reshape(-1, 16).max(1)
print("per 16-frame epoch max histogram:", np.histogram(e, [0, 16, 32, 64, 96, 128, 192, 256, 301])[0].tolist(), "of", len(e))
g = n.reshape(-1, 512).max(1)
print("per 512-frame group max:", g.tolist())
print("first 64 frames:", n[:64].tolist())
