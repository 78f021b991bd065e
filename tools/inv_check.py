"""Does the fp16 ReID engine give the same bits for a crop whatever the batch it is in?  (kernel variants are chosen by batch size)
    python tools/inv_check.py        [AICAM_NO_SIDE=1 / AICAM_NO_TAIL=1 to rule a fusion in or out]"""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
he = importlib.import_module("ai-camera_amd.hip_engine")
ef = importlib.import_module("ai-camera_amd.engine_file")
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
yp, rp = ef.ensure_seeded_engines(root, scale="n")
x = np.random.default_rng(3).standard_normal((832, 3, 128, 64)).astype(np.float32)
DT = os.environ.get("INV_DTYPE", "fp16")
big = he.HipEngine(rp, dtype=DT, max_items=832, warm_up=False).reid_infer_np(x)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("AICAM_")) or "(defaults)"
prev = None
for n in (256, 64, 8):
    e = he.HipEngine(rp, dtype=DT, max_items=n, warm_up=False).reid_infer_np(x[:n])
    d = np.abs(e - big[:n])
    line = f"{tag}: {n:4d} vs 832: equal {np.array_equal(e, big[:n])}, max diff {d.max():.2e}, rows differing {int((d.max(1) > 0).sum())}"
    if prev is not None:
        line += f"; vs the {len(prev)}-batch: equal {np.array_equal(e, prev[:n])}"
    print(line)
    prev = e
