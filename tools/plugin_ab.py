#!/usr/bin/env python3
"""The per-frame plugin loop without the HIP-event brackets: wall time of YOLODetector.detect / DeepSORT.update per frame and a digest of
everything they returned (two runs under different switches must print the same digest).  python tools/plugin_ab.py [frames]"""
import contextlib, hashlib, importlib, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
det_mod = importlib.import_module("ai-camera_amd.detector")
ds_mod = importlib.import_module("ai-camera_amd.deepsort_tracker")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
_, rpath = ef.ensure_seeded_engines(ROOT)
ypath = ef.ensure_trained_detector(ROOT)
sc = syn.Scene(seed=0, n_targets=30)
frames = sc.render_batch(0, n + 20)
with contextlib.redirect_stdout(io.StringIO()):
    det = det_mod.YOLODetector(engine_path=ypath, device="cuda:0", dtype="fp16", max_batch=1)
    trk = ds_mod.DeepSORT(reid_model_path=rpath, device="cuda:0", dtype="fp16", reid_max_batch=32)
h = hashlib.sha1()
td = tu = 0.0
lat = []
for f in range(n + 20):
    t0 = time.perf_counter()
    d = det.detect(frames[f])
    t1 = time.perf_counter()
    out = trk.update(d[0], d[1], d[2], frames[f].copy())
    t2 = time.perf_counter()
    if f >= 20:
        td += t1 - t0; tu += t2 - t1; lat.append(t2 - t0)
    for a in d[:3]:
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(repr([tuple(int(x) for x in o[:5]) for o in out]).encode())
lat = np.sort(np.asarray(lat))
print(f"detect {1e6 * td / n:7.1f} us  update {1e6 * tu / n:7.1f} us  fps {n / (td + tu):7.1f}  p50 {1e3 * lat[len(lat) // 2]:.3f} ms  digest {h.hexdigest()[:16]}  env "
      f"{ {k: v for k, v in os.environ.items() if k.startswith('AICAM_')} }")
