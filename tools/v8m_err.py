#!/usr/bin/env python3
"""Where YOLOv8m's fp32 box error (1.6e-3 px from the fp64 evaluation) comes from: the head's logits or the decode arithmetic."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
he = importlib.import_module("ai-camera_amd.hip_engine")
syn = importlib.import_module("ai-camera_amd.synthetic")
from oracle import image_oracle as I, nets_oracle as N
scale = sys.argv[1] if len(sys.argv) > 1 else "m"
ypath, _ = ef.ensure_seeded_engines(ROOT, scale=scale)
sc = syn.Scene(seed=2, n_targets=100, width=1920, height=1080, w_range=(30.0, 60.0), h_range=(90.0, 150.0), y_range=(50.0, 850.0))
x, _, _ = I.preprocess_yolo_input(sc.render(0))
torch.set_num_threads(16)
eo64 = N.EngineOracle(ypath, dtype=torch.float64)
d64, c64 = (t.numpy() for t in eo64.yolo_head(torch.from_numpy(x)))
b64 = eo64.decode(d64, c64, ft=np.float64)[0]
eng = he.HipEngine(ypath, dtype="fp32", max_items=2, warm_up=False)
dfl, cls = eng.yolo_head_np(x)
boxes, ml, lab = eng.yolo_decode_np(x)
print("HIP fp32 logits vs fp64: dfl", np.abs(dfl - d64).max(), "cls", np.abs(cls - c64).max())
b_hip_logits_64dec = eo64.decode(dfl.astype(np.float64), cls.astype(np.float64), ft=np.float64)[0]
e = np.abs(boxes - b64).ravel()
print("boxes: HIP head + HIP decode vs fp64: max", e.max(), " rms %.3e  p99 %.3e  p99.9 %.3e  anchors over 1e-3: %d of %d" % (np.sqrt((e ** 2).mean()), np.percentile(e, 99), np.percentile(e, 99.9), int((e > 1e-3).sum()), e.size))
el = np.abs(dfl - d64).ravel()
print("dfl logits vs fp64: rms %.3e  p99.9 %.3e" % (np.sqrt((el ** 2).mean()), np.percentile(el, 99.9)))
print("boxes: HIP head + fp64 decode vs fp64:", np.abs(b_hip_logits_64dec - b64).max(), " (= what the logits alone cost)")
print("boxes: HIP decode vs fp64 decode on the SAME (HIP) logits:", np.abs(boxes - b_hip_logits_64dec).max(), " (= what the decode arithmetic costs)")
st = eo64.anchors()[1] if hasattr(eo64, "anchors") else None
if st is not None:
    for s_ in (8, 16, 32):
        m = st == s_
        print(f"   stride {s_}: logits-only {np.abs(b_hip_logits_64dec - b64)[:, m].max():.2e}  decode-only {np.abs(boxes - b_hip_logits_64dec)[:, m].max():.2e}  dfl err {np.abs(dfl - d64)[:, m].max():.2e}")
