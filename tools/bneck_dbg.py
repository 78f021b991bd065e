import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
he = importlib.import_module("ai-camera_amd.hip_engine")
L = importlib.import_module("ai-camera_amd._lib")
ypath, _ = ef.ensure_seeded_engines(ROOT)
n = 1
x = np.random.default_rng(13).standard_normal((n, 3, 640, 640)).astype(np.float32) * 0.5
eng = he.HipEngine(ypath, dtype="fp16", max_items=n, warm_up=False)
def cat(buf, ch):
    a = np.zeros((n, 80, 80, ch), np.float16)
    L.call("aic_model_read_buffer", eng._h, buf, L.ptr(a), a.nbytes)
    return a.astype(np.float32)
for k, v in (kv.split("=") for kv in sys.argv[1:]):
    os.environ[k] = v
os.environ["AICAM_BNECK_DBG_T"] = "1"
os.environ["AICAM_BNECK_MODE"] = "2"          # only 15.c2f.m0 fused: tmp (buffer 25) then holds ITS intermediate
eng.yolo_head_np(x)
f11, f24, ft = cat(11, 128), cat(24, 96), cat(25, 32)
os.environ["AICAM_NO_BNECK"] = "1"
eng.yolo_head_np(x)
u11, u24, ut = cat(11, 128), cat(24, 96), cat(25, 32)
for nm, a, b, sl in (("15.c2f.m0 T (tmp)", ft, ut, slice(0, 32)), ("4.c2f input x (cat[32:64])", f11, u11, slice(32, 64)), ("4.c2f.m0 out (cat[64:96])", f11, u11, slice(64, 96)),
                     ("4.c2f.m1 out (cat[96:128])", f11, u11, slice(96, 128)), ("15.c2f.m0 in (cat[32:64])", f24, u24, slice(32, 64)), ("15.c2f.m0 out (cat[64:96])", f24, u24, slice(64, 96))):
    d = np.abs(a[..., sl] - b[..., sl])
    idx = np.argwhere(d > 0)
    print(nm, "differing", len(idx), "of", d.size, "max", float(d.max()))
    if len(idx):
        ys, xs, cs = idx[:, 1], idx[:, 2], idx[:, 3]
        print("   rows hist (y%8):", np.bincount(ys % 8, minlength=8).tolist(), " cols (x%40):", np.bincount(xs % 40, minlength=40).tolist()[:40])
        print("   first:", idx[:6].tolist(), [ (float(a[tuple(i[:3])+(sl.start+i[3],)]), float(b[tuple(i[:3])+(sl.start+i[3],)])) for i in idx[:4]])
