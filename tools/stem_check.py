import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
he = importlib.import_module("ai-camera_amd.hip_engine")
H, W = 128, 64
g = ef.Graph(ef.KIND_REID, H, W)
wg = ef._WeightGen(5)
inp = g.buf(H, W, ef.IN_C); a = g.buf(H, W, 64); x = g.buf(H // 2, W // 2, 64)
g.conv("conv0", inp, a, 3, 64, 3, 1, ef.ACT_RELU, wb=wg(64, 3, 3, ef.ACT_RELU))
g.simple(ef.OP_MAXPOOL3S2, a, x, 64)
# 1x1 conv to 64 linear f32-ish then avgpool: keep spatial info by pooling per 8x8? just avgpool
p = g.buf(1, 1, 64); g.simple(ef.OP_AVGPOOL, x, p, 64)
e = g.buf(1, 1, 64, ef.DT_F32); g.simple(ef.OP_L2NORM, p, e, 64)
g.outputs.append([e, 64, 0, 0, 0, 0, 0, 0]); g.meta = [64, 0, 0, 0, 0, 0, 0, 0]
path = f"/tmp/stem_{os.getpid()}.aicw"; ef.write_engine(path, g)
eng = he.HipEngine(path, dtype="fp16", max_items=8, warm_up=False)
rng = np.random.default_rng(0)
xs = rng.standard_normal((8, 3, H, W)).astype(np.float32)
# localized probes: crop i>=4 has a single bright pixel
for i in range(4, 8):
    xs[i] = 0
    xs[i, :, [0, 5, 64, 127][i - 4], [0, 17, 33, 63][i - 4]] = 3.0
xs[0] = 0; xs[1] = 1.0; xs[2] = 0; xs[2, 0] = 1.0
out = eng.reid_infer_np(xs)
np.save(sys.argv[1], out)
np.set_printoptions(linewidth=200, precision=3, suppress=True)
print(out[:4, :24])
