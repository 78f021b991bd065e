#!/usr/bin/env python3
"""What a conv pays for reading / writing a channel SLICE of a wider NHWC buffer (YOLOv8's C2f concat buffers): the 3x3 / 1, c -> c SiLU conv
of a C2f bottleneck, `reps` times back to back, in four arrangements of its tensors -- dense -> dense, slice -> dense, dense -> slice,
dense -> slice with the shortcut read from another slice -- through a throw-away engine (tools/conv_bench.py's harness), conv-class time by
the library's own HIP-event brackets, the stem-only graph subtracted.
  python tools/slice_bench.py [H = 80] [W = 80] [c = 32] [wide = 128] [items = 512] [reps = 8]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
ef = importlib.import_module("ai-camera_amd.engine_file")
cb = importlib.import_module("conv_bench")


def graph(H, W, c, wide, reps, mode):
    g = ef.Graph(ef.KIND_REID, H, W)
    wg = ef._WeightGen(3)
    inp = g.buf(H, W, ef.IN_C)
    dense = [g.buf(H, W, c), g.buf(H, W, c)]
    cat = g.buf(H, W, wide)
    # the stem fills slice 1 of the wide buffer and one dense buffer (ReLU: the stem kernel's form)
    g.conv("stem", inp, dense[0], 3, c, 1, 1, ef.ACT_RELU, wb=wg(c, 3, 1, ef.ACT_RELU))
    g.conv("stem2", inp, cat, 3, c, 1, 1, ef.ACT_RELU, wb=wg(c, 3, 1, ef.ACT_RELU), dst_coff=c)
    for i in range(reps):
        w = wg(c, c, 3, ef.ACT_SILU, 0.5)
        if mode == "dense->dense":
            g.conv(f"t{i}", dense[i & 1], dense[(i & 1) ^ 1], c, c, 3, 1, ef.ACT_SILU, wb=w)
        elif mode == "slice->dense":
            g.conv(f"t{i}", cat, dense[1], c, c, 3, 1, ef.ACT_SILU, wb=w, src_coff=c)
        elif mode == "dense->slice":
            g.conv(f"t{i}", dense[0], cat, c, c, 3, 1, ef.ACT_SILU, wb=w, dst_coff=2 * c)
        elif mode == "dense->slice+res":
            g.conv(f"t{i}", dense[0], cat, c, c, 3, 1, ef.ACT_SILU, wb=w, dst_coff=2 * c, res=(cat, c), res_mode=ef.RES_ACT_THEN_ADD)
    src = dense[0]
    p = g.buf(1, 1, c)
    g.simple(ef.OP_AVGPOOL, src, p, c)
    e = g.buf(1, 1, c, ef.DT_F32)
    g.simple(ef.OP_L2NORM, p, e, c)
    g.outputs.append([e, c, 0, 0, 0, 0, 0, 0])
    g.meta = [c, 0, 0, 0, 0, 0, 0, 0]
    return g


def main():
    a = [int(v) for v in sys.argv[1:]]
    H, W, c, wide, items, reps = (a + [80, 80, 32, 128, 512, 8][len(a):])[:6]
    ms0, _, _ = cb.conv_ms(graph(H, W, c, wide, 0, ""), items, H, W)
    for mode in ("dense->dense", "slice->dense", "dense->slice", "dense->slice+res"):
        ms, _, _ = cb.conv_ms(graph(H, W, c, wide, reps, mode), items, H, W)
        print(f"{H}x{W} c{c} in a {wide}-channel buffer, {items} items: {mode:18s} {1e3 * (ms - ms0) / reps:8.1f} us per conv")


if __name__ == "__main__":
    main()
