#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run of bench.py: per-kernel totals and a per-layer table of the
conv launches (dispatch order within a launch group = graph order: YOLO convs, then ReID convs)."""
import collections
import csv
import glob
import importlib
import os
import re
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ef = importlib.import_module("ai-camera_amd.engine_file")


def main(d, frames=16, crops=480, top=30, fused_stem=1, fused_yolo_stem=1, fused_block=1):
    trace = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        n = re.sub(r"<.*", "", n)
        m = re.search(r"aic(\d+)([a-z_0-9]+?)(I|E)", n)
        if n.startswith("_ZN3aic"):
            n = re.sub(r"^_ZN3aic\d+", "", n)
            n = re.sub(r"(IDF16_|If|EvN|EEv|Ev).*", "", n)
        t = tot[n]
        t[0] += 1
        t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    all_us = sum(v[1] for v in tot.values())
    print(f"{'kernel':42s} {'calls':>7s} {'total ms':>9s} {'%':>6s} {'avg us':>8s}")
    for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{n[:42]:42s} {c:7d} {us / 1e3:9.2f} {100 * us / all_us:6.1f} {us / c:8.1f}")
    conv = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv3x3_patch", "conv3x3_pp_patch", "conv3x3_sp_patch", "conv3x3s2_sp_patch", "conv3x3_c16", "conv1x1_stream", "conv3x3_c32s2_tail", "conv3x3_pm_patch", "conv3x3_c64_resident", "conv3x3_c64_block", "c2f16_fused"))]
    # the conv class over the steady-state middle of the trace: summed kernel durations against the UNION of their intervals (with the class
    # on two streams -- bench.py's default since round 4 -- the sum counts every overlapped microsecond twice; bench.py's roofline uses
    # the union, from paired HIP events per stream), and the class's rate over each, from the full launch groups that START in the window
    if conv:
        t_lo, t_hi = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
        a_, b_ = t_lo + 0.25 * (t_hi - t_lo), t_lo + 0.85 * (t_hi - t_lo)
        win = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in conv if a_ <= int(r["Start_Timestamp"]) < b_)
        stems = [r for r in rows if "yolo_stem_fused" in r["Kernel_Name"] and a_ <= int(r["Start_Timestamp"]) < b_]
        grid = lambda r: int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        full = max((grid(r) for r in stems), default=0)
        n_full = sum(1 for r in stems if grid(r) == full)
        n_frames = sum(grid(r) for r in stems) * frames / max(full, 1)          # every group that starts in the window, ramp and tail groups too
        if win and n_full:
            summed = sum(e - b for b, e in win) / 1e3
            union, lo, hi = 0.0, win[0][0], win[0][1]
            for b, e in win[1:]:
                if b > hi:
                    union += hi - lo
                    lo, hi = b, e
                else:
                    hi = max(hi, e)
            union = (union + hi - lo) / 1e3                                    # microseconds
            gf = 76.046 - 0.088 - 30 * 0.0283      # conv-class GFLOP per frame at the headline configuration (SURVEY 8d minus the two fused 3-channel stems)
            print(f"\nconv class, middle 60 % of the trace (launch groups of {n_frames:.0f} frames start in it, {n_full} of them full {frames}-frame groups): "
                  f"summed kernel time {summed / 1e3:.1f} ms, union of the kernels' intervals {union / 1e3:.1f} ms; per {frames} frames {summed * frames / n_frames / 1e3:.2f} / "
                  f"{union * frames / n_frames / 1e3:.2f} ms; class rate over the union ~{n_frames * gf * 1e3 / union:.0f} TFLOP/s (over the sum ~{n_frames * gf * 1e3 / summed:.0f})")
    layers = []
    for name, g, n in (("yolo", ef.build_yolov8("n", calibrate=False), frames), ("reid", ef.build_reid(calibrate=False), crops)):
        for o in g.ops:
            if o[0] == 1 and not (name == "reid" and g.names[o[15]] == "conv0" and fused_stem) \
                    and not (name == "yolo" and g.names[o[15]] == "0.conv" and fused_yolo_stem):
                h, w, _, _ = g.buffers[o[4]]
                layers.append((name, g.names[o[15]], n * h * w, o[6], o[3] * o[7] * o[8]))
    if fused_block:     # a 64-channel BasicBlock (layer1.N.conv1 + conv2) is ONE launch of conv3x3_c64_block_kernel: K doubled = both convs' FLOPs
        merged = []
        for L in layers:
            if merged and L[0] == "reid" and L[1].startswith("layer1.") and L[1].endswith(".conv2") and merged[-1][1] == L[1][:-1] + "1":
                p = merged.pop()
                merged.append((p[0], L[1][:-6] + " (block)", p[2], p[3], p[4] + L[4]))
            else:
                merged.append(L)
        layers = merged
    if fused_block:     # YOLOv8n's 160 x 160 C2f (cv1, m0.cv1, m0.cv2, cv2) is ONE launch of c2f16_fused_kernel: N*K = sum over the four convs
        merged, i = [], 0
        while i < len(layers):
            L = layers[i]
            if L[0] == "yolo" and L[1] == "2.c2f.cv1" and i + 3 < len(layers) and layers[i + 3][1] == "2.c2f.cv2":
                nk = sum(x[3] * x[4] for x in layers[i:i + 4])
                merged.append(("yolo", "2.c2f (fused x4)", L[2], 1, nk))
                i += 4
            else:
                merged.append(L)
                i += 1
        layers = merged
    if fused_block:     # the detect branches' last 1x1 (22.box*.2 / 22.cls*.2, n-scale: 64 / 80 channels) runs in the epilogue of the 3x3 before it
        merged = []
        for L in layers:
            head = L[1].startswith("22.") and L[1].endswith(".2") and bool(merged) and merged[-1][1] == L[1][:-1] + "1"
            c2f_in = L[1] == "4.c2f.cv1" and bool(merged) and merged[-1][1] == "3.conv"      # same graph property: 3.conv's only reader is this 1x1
            if merged and L[0] == "yolo" and (head or c2f_in) and merged[-1][3] in (64, 80) and L[3] <= merged[-1][3]:
                p = merged.pop()
                merged.append((p[0], p[1] + ("+.2" if head else "+4.c2f.cv1"), p[2], 1, p[3] * p[4] + L[3] * L[4]))
            else:
                merged.append(L)
        layers = merged
    if fused_block:     # 22.box{l}.0 and 22.cls{l}.0 read the same map: one conv with the output channels side by side (Model::Model merge)
        merged = []
        for L in layers:
            if L[0] == "yolo" and L[1].startswith("22.cls") and L[1].endswith(".0"):
                k = next((i for i, q in enumerate(merged) if q[1] == L[1].replace("cls", "box")), None)
                if k is not None and merged[k][2] == L[2] and merged[k][4] == L[4] and L[2] // frames <= 1600:     # maps up to 40 x 40 (engine.cpp)
                    q = merged[k]
                    merged[k] = (q[0], q[1] + "+cls.0", q[2], q[3] + L[3], q[4])
                    continue
            merged.append(L)
        layers = merged
    if fused_block and not os.environ.get("AICAM_NO_DS_FOLD"):     # a ResNet downsample 1x1 is a second source of the block's last conv (engine.cpp fold): K columns behind the window's
        merged = []
        for L in layers:
            hw = L[2] // crops      # conv_x2_supported: the 512 x 128 tile (32 x 16 maps, Cout 128) and the 256 x 256 tile on 16 x 8 / 8 x 4 maps
            if merged and L[0] == "reid" and L[1].endswith(".0.conv2") and merged[-1][1] == L[1][:-5] + "ds" and \
                    ((L[3] == 128 and hw % 512 == 0) or (L[3] % 256 == 0 and hw % 32 == 0)):
                p = merged.pop()
                merged.append((L[0], L[1] + "+ds", L[2], L[3], L[4] + p[4]))
            else:
                merged.append(L)
        layers = merged
    per = len(layers)
    # a launch group starts at its (fused) YOLO stem; groups of other sizes (tapered tail of a call: fewer frames, and below
    # the fused-block threshold two more launches) are dropped: keep the groups with `per` conv launches and the modal grid
    is_conv = lambda r: any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv3x3_patch", "conv3x3_pp_patch", "conv3x3_sp_patch", "conv3x3s2_sp_patch", "conv3x3_c16", "conv1x1_stream", "conv3x3_c32s2_tail", "conv3x3_pm_patch", "conv3x3_c64_resident", "conv3x3_c64_block", "c2f16_fused"))
    glist, cur = [], None
    for r in rows:
        if "yolo_stem_fused" in r["Kernel_Name"] or "letterbox" in r["Kernel_Name"]:
            if cur:
                glist.append(cur)
            cur = []
        elif cur is not None and is_conv(r):
            cur.append(r)
    if cur:
        glist.append(cur)
    glist = [g for g in glist if len(g) == per]
    gkey = [int(g[0].get("Grid_Size_X", g[0].get("Grid_Size", 0))) for g in glist]
    modal = collections.Counter(gkey).most_common(1)[0][0]
    dur = [[] for _ in range(per)]
    for g, k in zip(glist, gkey):
        if k != modal:
            continue
        for i, r in enumerate(g):
            dur[i].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    groups = sum(1 for k in gkey if k == modal)
    out = []
    for (name, ln, m, co, k), ts in zip(layers, dur):
        t = statistics.median(ts)
        out.append((t, name, ln, m, co, k, 2.0 * m * co * k / t / 1e6))
    total = sum(o[0] for o in out)
    print(f"\nconv launches per group {per}, groups {groups}, conv us per group {total:.0f} "
          f"(yolo {sum(o[0] for o in out if o[1] == 'yolo'):.0f}, reid {sum(o[0] for o in out if o[1] == 'reid'):.0f}); "
          f"overall {sum(2.0 * o[3] * o[4] * o[5] for o in out) / total / 1e6:.0f} TFLOP/s")
    for t, name, ln, m, co, k, tf in sorted(out, reverse=True)[:top]:
        print(f"{t:8.1f} us {100 * t / total:5.1f}%  {name} {ln:18s} M={m:8d} N={co:4d} K={k:5d} {tf:7.1f} TF")


if __name__ == "__main__":
    main(sys.argv[1], *(int(v) for v in sys.argv[2:]))
