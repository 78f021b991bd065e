"""Unit check of the fused ReID stem kernel through its (C++-mangled) launcher symbol, against torch."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "ai-camera_amd", "libaicam.so"))
fn = getattr(lib, "_ZN3aic21launch_reid_stem_poolEPKvS1_PKfPviiiiiiiP12ihipStream_t")
fn.restype = None
fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p]
torch.manual_seed(0)
n, H, W, Kp = 4, 128, 64, 96
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
x = torch.randn(n, 3, H, W) if mode == "rand" else torch.zeros(n, 3, H, W)
w = torch.randn(64, 3, 3, 3) * 0.3
b = torch.randn(64) * 0.2
xh = torch.zeros(n, H, W, 8, dtype=torch.float16); xh[..., :3] = x.permute(0, 2, 3, 1).half()
wp = torch.zeros(64, Kp, dtype=torch.float16)
wp[:, :72].view(64, 9, 8)[:, :, :3] = w.permute(0, 2, 3, 1).reshape(64, 9, 3).half()
dev = "cuda"
xd, wd, bd = xh.to(dev), wp.to(dev), b.to(dev)
yd = torch.full((n, H // 2, W // 2, 64), -7.0, dtype=torch.float16, device=dev)
torch.cuda.synchronize()
fn(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), n, H, W, Kp, 64, 0, 8, None)
torch.cuda.synchronize()
y = yd.cpu().float()
ref = torch.nn.functional.conv2d(xh[..., :3].permute(0, 3, 1, 2).float(), wp[:, :72].view(64, 9, 8)[:, :, :3].reshape(64, 3, 3, 3).permute(0, 3, 1, 2).float(), b, padding=1)
ref = torch.relu(ref).half().float()
ref = torch.nn.functional.max_pool2d(ref, 3, 2, 1).permute(0, 2, 3, 1)
err = (y - ref).abs()
print("max err", err.max().item(), "bad frac", (err > 0.02).float().mean().item())
bad = (err > 0.02)
print("bad per channel", bad.sum((0, 1, 2)).tolist())
print("bad per pooled col", bad.sum((0, 1, 3)).tolist())
print("bad per pooled row (first 20)", bad.sum((0, 2, 3)).tolist()[:20])
idx = bad.nonzero()[:10]
for i in idx: print(i.tolist(), y[tuple(i)].item(), ref[tuple(i)].item())
for ch in (0, 1, 9, 33):
    bm = bad[0, :, :, ch]
    print("ch", ch, "rows", sorted(set(bm.nonzero()[:, 0].tolist()))[:12], "cols", sorted(set(bm.nonzero()[:, 1].tolist())))
    print("   vals", sorted(set(y[0, :, :, ch][bm].tolist()))[:8])
