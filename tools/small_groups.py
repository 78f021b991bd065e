#!/usr/bin/env python3
"""The bench's pipeline with small launch groups (the latency end of by_launch_group_frames):
   python tools/small_groups.py [group_frames=16] [frames=512] [passes=4] [key=value pipeline options ...]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
syn = importlib.import_module("ai-camera_amd.synthetic")
L = importlib.import_module("ai-camera_amd._lib")
TP = importlib.import_module("ai-camera_amd.pipeline").TrackingPipeline

g = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
batch = int(os.environ.get("SG_BATCH", "512"))
ypath, rpath = ef.ensure_seeded_engines(ROOT)
sc = syn.Scene(seed=0, n_targets=30)
host = np.ascontiguousarray(sc.render_batch(0, n))
TP.pin(host)
p = TP(ypath, rpath, (720, 1280), batch=batch, ring_frames=max(n, batch), max_persons=32, dtype="fp16", inject=True)
p.inject(0, [sc.detections(f)[:3] for f in range(n)])
p.option("split_streams", 1)
p.option("group_frames", g)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    p.option(k, int(v))
p.run_raw_from_host_passes(host, 1)
L.call("aic_device_sync", 0)
p.stats(reset=True)
t0 = time.perf_counter()
p.run_raw_from_host_passes(host, passes)
L.call("aic_device_sync", 0)
dt = time.perf_counter() - t0
gf, gl = p.group_times()
full = gf == g
lat = np.sort(gl[full])
st = p.stats()
print(f"group_frames={g} batch={batch}: {passes * n / dt:.1f} frames/s, latency p50 {1e3 * lat[len(lat) // 2]:.2f} ms p99 {1e3 * lat[int(0.99 * len(lat))]:.2f} ms; "
      f"host us/frame issue {1e6 * st['issue_s'] / st['frames']:.1f} wait {1e6 * st['wait_s'] / st['frames']:.1f} track {1e6 * st['track_s'] / st['frames']:.1f}")
p.close()
