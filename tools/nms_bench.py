#!/usr/bin/env python3
"""Decode + select / sort / NMS of a launch group alone: B frames of the bench's scene through aic_detect, the decode_nms class's own HIP-event
bracket read back (AICAM_NMS_DBG=1 / 2: the NMS kernel leaves after the selection / after the sort -- timing only).
  python tools/nms_bench.py [frames = 256]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
L = importlib.import_module("ai-camera_amd._lib")
he = importlib.import_module("ai-camera_amd.hip_engine")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ypath, _ = ef.ensure_seeded_engines(ROOT)
if os.environ.get("TRAINED", "0") != "0":            # the trained detector (~30 candidates per frame instead of ~3 000)
    ypath = ef.ensure_trained_detector(ROOT)
sc = syn.Scene(seed=0, n_targets=30)
frames = sc.render_batch(0, n)
eng = he.HipEngine(ypath, dtype="fp16", max_items=n, warm_up=False)
nd, boxes, scores, labels = eng.detect_np(frames)
L.call("aic_prof_enable", 0, 0x7f)
L.call("aic_prof_reset", 0)
R = 3
for _ in range(R):
    nd, boxes, scores, labels = eng.detect_np(frames)
p = L.prof_read(0)
L.call("aic_prof_enable", 0, 0)
d = p["decode_nms"]
lb, cv = p["letterbox"], p["conv_igemm"]
print(f"{n} frames: letterbox + stem (+ 1.conv) {1e3 * lb['ms'] / R:.1f} us, conv class {1e3 * cv['ms'] / R:.1f} us ({cv['launches'] // R} launches); "
      f"decode + NMS {1e3 * d['ms'] / R:.1f} us per launch group ({d['launches'] // R} launches), detections per frame {nd.mean():.1f}, "
      f"checksum {float(boxes.sum()):.3f} {int(labels.sum())}")
eng.close()
