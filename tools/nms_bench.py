#!/usr/bin/env python3
"""Decode + NMS microbench through the C ABI: the seeded YOLOv8n engine of bench.py on synthetic frames, time of the
decode_nms profiling class per frame (HIP events inside the library) and candidate / detection counts.
  python tools/nms_bench.py [frames]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
L = importlib.import_module("ai-camera_amd._lib")
he = importlib.import_module("ai-camera_amd.hip_engine")
syn = importlib.import_module("ai-camera_amd.synthetic")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    path, _ = ef.ensure_seeded_engines(ROOT, scale="n")          # the engines bench.py runs
    eng = he.HipEngine(path, dtype="fp16", max_items=n, warm_up=False)
    frames = syn.Scene(seed=0, n_targets=30, width=1280, height=720).render_batch(0, n)
    eng.detect_np(frames)
    L.call("aic_prof_enable", 0, 0xFF)
    L.call("aic_prof_reset", 0)
    reps = 5
    for _ in range(reps):
        nd, *_ = eng.detect_np(frames)
    pr = L.prof_read(0)
    L.call("aic_prof_enable", 0, 0)
    d = pr["decode_nms"]
    print(f"frames {n}: decode+nms {d['ms'] * 1e3 / (reps * n):.2f} us/frame ({d['launches']} launches), dets/frame mean {np.mean(nd):.1f} max {np.max(nd)}")


if __name__ == "__main__":
    main()
