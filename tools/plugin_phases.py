#!/usr/bin/env python3
"""Where a frame of the per-frame plugin loop (YOLODetector.detect + DeepSORT.update, src/aicamera_tracker.py:169-207) spends its time:
wall time per call of the three parts, and -- with the library's HIP-event brackets -- the GPU time inside them.  python tools/plugin_phases.py [frames]"""
import contextlib, importlib, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
L = importlib.import_module("ai-camera_amd._lib")
det_mod = importlib.import_module("ai-camera_amd.detector")
ds_mod = importlib.import_module("ai-camera_amd.deepsort_tracker")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ypath, rpath = ef.ensure_seeded_engines(ROOT)
if os.environ.get("TRAINED", "1") != "0":
    ypath = ef.ensure_trained_detector(ROOT)
sc = syn.Scene(seed=0, n_targets=30)
frames = sc.render_batch(0, n + 20)
with contextlib.redirect_stdout(io.StringIO()):
    det = det_mod.YOLODetector(engine_path=ypath, device="cuda:0", dtype="fp16", max_batch=1)
    trk = ds_mod.DeepSORT(reid_model_path=rpath, device="cuda:0", dtype="fp16", reid_max_batch=32)
t = dict(detect=0.0, embed=0.0, track=0.0, total=0.0)
L.call("aic_prof_enable", 0, 0x7f)
for f in range(n + 20):
    if f == 20:
        for k in t: t[k] = 0.0
        L.call("aic_prof_reset", 0)
    boxes, conf, cids, _ = sc.detections(f)
    t0 = time.perf_counter()
    d = det.detect(frames[f])
    t1 = time.perf_counter()
    if os.environ.get("TRAINED", "1") != "0":
        boxes, conf, cids = d[0], d[1], d[2]
    feats, valid = trk.reid_model.embed_boxes(frames[f].copy(), boxes)
    t2 = time.perf_counter()
    trk.tracker_core.predict()
    tlwh = np.stack([boxes[:, 0], boxes[:, 1], boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).astype(np.float32)
    trk.tracker_core.update_arrays(tlwh, conf, cids.astype(np.int32), feats, valid.astype(np.uint8))
    rows, cf = trk.tracker_core.outputs()
    t3 = time.perf_counter()
    t["detect"] += t1 - t0; t["embed"] += t2 - t1; t["track"] += t3 - t2; t["total"] += t3 - t0
p = L.prof_read(0)
print("wall us per frame:", {k: round(1e6 * v / n, 1) for k, v in t.items()}, " fps", round(n / t["total"], 1))
print("GPU bracket us per frame:", {k: (round(1e3 * v["ms"] / n, 1), v["launches"] // n) for k, v in p.items() if isinstance(v, dict) and v.get("launches")})
