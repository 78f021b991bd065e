#!/usr/bin/env python3
"""bench.py's own-detections leg (inject = 0) with the association forced onto the device or the host, AICAM_TRK_PHASES=1 prints the
epoch kernel's per-phase cycles at exit:  AICAM_TRK_PHASES=1 python tools/own_phases.py [device_assoc 0/1/2] [passes]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
cfg = importlib.import_module("ai-camera_amd.config")
L = importlib.import_module("ai-camera_amd._lib")
TP = importlib.import_module("ai-camera_amd.pipeline").TrackingPipeline

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8597
R = 1024
ypath, rpath = ef.ensure_seeded_engines(ROOT)
sc = syn.Scene(seed=0, n_targets=30)
host = np.empty((2 * R, 720, 1280, 3), np.uint8)
host[:R] = sc.render_batch(0, R)
host[R:] = host[:R][::-1]
TP.pin(host)
cfg.CLASSES_TO_TRACK.clear()
cfg.CLASSES_TO_TRACK.update(cfg.CLASSES)
p = TP(ypath, rpath, (720, 1280), batch=512, ring_frames=2 * R, max_persons=64, dtype="fp16", inject=False, min_confidence=floor, max_tracks=512)
p.option("device_assoc", mode)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    p.option(k, int(v))
p.run_raw_from_host_passes(host, 1)
L.call("aic_device_sync", 0)
p.stats(reset=True)
t0 = time.perf_counter()
nt, _, nd = p.run_raw_from_host_passes(host, passes)
L.call("aic_device_sync", 0)
dt = time.perf_counter() - t0
st = p.stats()
c = p.counters()
print(f"device_assoc={mode}: {passes * 2 * R / dt:.1f} frames/s; confirmed/frame {nt.mean():.1f}; assoc frames (dev, host) = ({c['assoc_device_frames']}, {c['assoc_host_frames']}); "
      f"filter groups (dev, host) = ({c['filter_device_groups']}, {c['filter_host_groups']}); host us/frame issue {1e6*st['issue_s']/st['frames']:.1f} wait {1e6*st['wait_s']/st['frames']:.1f} track {1e6*st['track_s']/st['frames']:.1f}")
a = p.tracker_core.export_arrays()
print("tracks alive at the end:", len(a["track_id"]), "states:", np.bincount(a["state"], minlength=4).tolist())
p.close()
