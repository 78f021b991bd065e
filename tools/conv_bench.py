#!/usr/bin/env python3
"""Single-layer conv microbench through the C ABI: builds a throw-away ReID-kind engine whose body is
`reps` copies of one conv (ping-pong buffers), runs it on n items and reports the conv-class TFLOP/s
measured by the library's own HIP-event brackets.
  python tools/conv_bench.py H W CIN COUT K [items] [reps] [res]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")
L = importlib.import_module("ai-camera_amd._lib")
he = importlib.import_module("ai-camera_amd.hip_engine")


def graph(H, W, cin, cout, k, reps, res, stride=1):
    g = ef.Graph(ef.KIND_REID, H, W)
    wg = ef._WeightGen(3)
    inp = g.buf(H, W, ef.IN_C)
    a, b = g.buf(H, W, cin), g.buf(H // stride, W // stride, cout)
    g.conv("stem", inp, a, 3, cin, 1, 1, ef.ACT_RELU, wb=wg(cin, 3, 1, ef.ACT_RELU))
    src = a
    if res == 2:        # BasicBlock pairs: conv1 src -> mid, conv2 mid -> dst adding src (the fused 64-channel block kernel's pattern)
        mid = g.buf(H, W, cout)
        for i in range(reps // 2):
            dst = b if src == a else a
            g.conv(f"b{i}c1", src, mid, cin, cout, k, 1, ef.ACT_RELU, wb=wg(cout, cin, k, ef.ACT_RELU))
            g.conv(f"b{i}c2", mid, dst, cout, cout, k, 1, ef.ACT_RELU, wb=wg(cout, cout, k, ef.ACT_RELU, 0.5), res=(src, 0), res_mode=ef.RES_ADD_THEN_ACT)
            src = dst
        reps = 0
    for i in range(reps):
        dst = b if src == a else a
        if (cin != cout or stride != 1) and i:
            break
        g.conv(f"t{i}", src, dst, cin, cout, k, stride, ef.ACT_RELU, wb=wg(cout, cin, k, ef.ACT_RELU, 0.5 if res else 1.0),
               **(dict(res=(dst, 0), res_mode=ef.RES_ADD_THEN_ACT) if res and cin == cout else {}))
        src = dst
    co = cin if src == a else cout               # (no test conv at all -- the CB_NET baseline -- pools the stem's output)
    p = g.buf(1, 1, co)
    g.simple(ef.OP_AVGPOOL, src, p, co)
    e = g.buf(1, 1, co, ef.DT_F32)
    g.simple(ef.OP_L2NORM, p, e, co)
    g.outputs.append([e, co, 0, 0, 0, 0, 0, 0])
    g.meta = [co, 0, 0, 0, 0, 0, 0, 0]
    return g


def conv_ms(g, items, H, W):
    path = f"/tmp/convbench_{os.getpid()}.aicw"
    ef.write_engine(path, g)
    eng = he.HipEngine(path, dtype=os.environ.get("DTYPE", "fp16"), max_items=items, warm_up=False)
    x = np.random.default_rng(0).standard_normal((items, 3, H, W)).astype(np.float32)
    eng.reid_infer_np(x)
    L.call("aic_prof_enable", 0, 1)
    L.call("aic_prof_reset", 0)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.reid_infer_np(x)
    dt = time.perf_counter() - t0
    p = L.prof_read(0)["conv_igemm"]
    L.call("aic_prof_enable", 0, 0)
    eng.close()
    os.remove(path)
    return p["ms"] / 5, p["flops"] / 5, dt / 5


def main():
    H, W, cin, cout, k = (int(v) for v in sys.argv[1:6])
    items = int(sys.argv[6]) if len(sys.argv) > 6 else 960
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 8
    res = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    stride = int(sys.argv[9]) if len(sys.argv) > 9 else 1
    g = graph(H, W, cin, cout, k, reps, res, stride)
    n_conv = sum(1 for o in g.ops if o[0] == 1)
    ms, fl, wall = conv_ms(g, items, H, W)
    us = ms * 1e3 / n_conv
    extra = ""
    if os.environ.get("CB_NET"):            # the layer alone: the same graph without the test convs (stem only) subtracted
        ms0, fl0, _ = conv_ms(graph(H, W, cin, cout, k, 0, 0, stride), items, H, W)
        n_t = n_conv - 1
        extra = f"  NET {1e3 * (ms - ms0) / max(n_t, 1):8.1f} us per conv = {(fl - fl0) / max(ms - ms0, 1e-9) / 1e9:7.1f} TF"
    print(f"H{H} W{W} cin{cin} cout{cout} k{k} s{stride} items{items} res{res}: M={items*(H//stride)*(W//stride)} K={cin*k*k}  "
          f"{fl / ms / 1e9:7.1f} TF  (~{us:.1f} us per conv, wall {wall*1e3:.2f} ms/iter){extra} env={ {k_: v for k_, v in os.environ.items() if k_.startswith('AICAM')} }")


if __name__ == "__main__":
    main()
