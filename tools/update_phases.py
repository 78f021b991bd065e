import contextlib, importlib, io, os, sys, time
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
det_mod = importlib.import_module("ai-camera_amd.detector")
ds_mod = importlib.import_module("ai-camera_amd.deepsort_tracker")
cfg = importlib.import_module("ai-camera_amd.config")
n = 300
_, rpath = ef.ensure_seeded_engines(ROOT)
ypath = ef.ensure_trained_detector(ROOT)
sc = syn.Scene(seed=0, n_targets=30)
frames = sc.render_batch(0, n + 20)
with contextlib.redirect_stdout(io.StringIO()):
    det = det_mod.YOLODetector(engine_path=ypath, device="cuda:0", dtype="fp16", max_batch=1)
    trk = ds_mod.DeepSORT(reid_model_path=rpath, device="cuda:0", dtype="fp16", reid_max_batch=32)
T = dict(detect=0, copy=0, predict=0, filt=0, embed=0, glue=0, upd=0, outs=0, tup=0)
for f in range(n + 20):
    if f == 20:
        for k in T: T[k] = 0
    t = [time.perf_counter()]
    d = det.detect(frames[f]); t.append(time.perf_counter())
    fr = frames[f].copy(); t.append(time.perf_counter())
    trk.frame_count += 1
    trk.tracker_core.predict(); t.append(time.perf_counter())
    boxes = np.asarray(d[0], dtype=np.float32).reshape(-1, 4); confs = np.asarray(d[1], dtype=np.float32).reshape(-1); cids = np.asarray(d[2]).reshape(-1).astype(np.int64)
    lut = ds_mod._tracked_lut(); known = (cids >= 0) & (cids < len(lut))
    keep = np.nonzero((confs >= trk.min_detection_confidence) & known & lut[np.where(known, cids, 0)])[0]
    b, c, k = boxes[keep], confs[keep], cids[keep].astype(np.int32); t.append(time.perf_counter())
    feats, valid = trk.reid_model.embed_boxes(fr, b); t.append(time.perf_counter())
    tlwh = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1).astype(np.float32); v8 = valid.astype(np.uint8); t.append(time.perf_counter())
    trk.tracker_core.update_arrays(tlwh, c, k, feats, v8); t.append(time.perf_counter())
    rows, conf = trk.tracker_core.outputs(); t.append(time.perf_counter())
    out = [(r[0], r[1], r[2], r[3], r[4], cfg.class_name(r[5]), cf) for r, cf in zip(rows.tolist(), conf.tolist())]; t.append(time.perf_counter())
    for key, (a, bb) in zip(T, zip(t, t[1:])): T[key] += bb - a
print({k: round(1e6 * v / n, 1) for k, v in T.items()}, "sum", round(1e6 * sum(T.values()) / n, 1))
