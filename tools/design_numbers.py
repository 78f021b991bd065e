#!/usr/bin/env python3
"""Rewrite DESIGN.md §0's "current numbers" rows (between the numbers:begin / numbers:end markers) from the files under profiles/:
   python tools/design_numbers.py [tag] ["value range text"]
tag (default r05) selects profiles/<tag>_bench.json, _conv_layers.txt, _sq_counters.txt, _kernel_summary.txt and profiles/pmc_traffic.json."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
RANGE = sys.argv[2] if len(sys.argv) > 2 else "10 923 – 11 494"
P = lambda n: os.path.join(ROOT, "profiles", f"{TAG}_{n}")

TEMPLATE = r"""| Quantity | Value | Source |
|---|---|---|
| `value`: end-to-end frames/s, 1×MI355X, 1280×720, 30 planted persons, YOLOv8n (the trained weights since the end of round 5: NMS sees ~30 candidates per frame, +0.8 %) + ReID + DeepSORT, fp16, frame bytes in page-locked host memory → track tuples on host, two streams | **@VALUE@** (boxes of this pool differ by ±3 %: @VALUE_RANGE@ over the round's runs) | `profiles/@TAG@_bench.json` |
| conv class (the `roofline` object): TFLOP/s over the union of both streams' conv intervals ÷ 2 500 | **@ACH@ TFLOP/s = @FRAC@**; one stream, full groups of the rocprofv3 trace: @ACH1@ TFLOP/s = @FRAC1@ | `profiles/@TAG@_bench.json`, `profiles/@TAG@_conv_layers.txt` |
| conv time per 512-frame launch group (one stream): total / YOLOv8n / ReID | **@CONV_MS@ / @YOLO_MS@ / @REID_MS@ ms** (round 4: 42.16 / 11.70 / 30.46) | `profiles/@TAG@_conv_layers.txt` |
| per-half rate: ReID 34.0 TFLOP per group, YOLOv8n 4.43 | ReID @REID_TF@ TFLOP/s = @REID_FRAC@; YOLOv8n @YOLO_TF@ = @YOLO_FRAC@ | same |
| MFMA busy, time-weighted: class / ReID trunk kernels | @BUSY@ % / @BUSY_REID@ % | `profiles/@TAG@_sq_counters.txt` |
| HBM traffic per full-group conv launch (PMC, FETCH×2 + WRITE) vs algorithmic | @TRAFFIC@ GB vs @TRAFFIC_ALG@ GB | `profiles/@TAG@_pmc_traffic.json` |
| tracker stream per 512-frame group (beside the convs) | @TRK_MS@ ms | `profiles/@TAG@_kernel_summary.txt` |
| **own detections on the TRAINED detector** (inject = 0: the reference's data flow; same clip, same span, streams as the headline) | **@OWN_FPS@ frames/s = @OWN_FRAC@ of `value`**; association frames (device, host) = (@OWN_DEV@, 0); detector recall of the planted boxes @OWN_RECALL@; fp16 vs fp32 of the same engines on their own detections: @OWN_REPRO@ of the track outputs reproduced, @OWN_IDSW@ id switches in 256 frames (test scene, 300 frames: 100 %, 0) | `profiles/@TAG@_bench.json` (`config.own_detections_trained_detector…`), `tests/test_trained_detector.py` |
| own detections, seeded texture scene (stress leg: 100–258 detections in a fifth of the frames) | @TEX_DEV@ / @TEX_HOST@ frames/s (filter per group / on the host) | same |
| per-frame plugin loop (`YOLODetector.detect` + `DeepSORT.update` on the detector's own boxes, one frame per call, host arrays in and out) | **@PLUG_FPS@ frames/s, p50 @PLUG_P50@ ms** (round 4: 532, 1.87) | `profiles/@TAG@_bench.json` (`config.plugin_loop`) |
| launch groups of 16 / 64 / 128 / 256 / 512 frames: frames/s (p50 latency) | @CURVE@ | same |
| CPU oracle chain on the box's 16 cores / 1 thread | @CPU16@ / @CPU1@ frames/s | same |
| YOLOv8m fp32 boxes vs the fp64 evaluation of the same engine (33 600 coordinates) | rms 5.3e-5 px, 99.9th percentile 5.2e-4, max 1.05e-3 (1 coordinate above 1e-3); round 4: max 1.6e-3 | `gpurun_out/r5_v8m_err.txt`, `tests/test_gpu_configs.py` |
| GPU test suite | 162 passed (`pytest -m gpu`, 6 min); CPU suite 46 passed | `GPUTEST_r05.json` |"""


def sp(x, nd=0):
    """12345.6 -> '12 346' (thin grouping as the rest of the document writes it)."""
    s = f"{x:,.{nd}f}".replace(",", " ")
    return s


def main():
    b = json.loads(open(P("bench.json")).read().strip().splitlines()[-1])
    c, r = b["config"], b["roofline"]
    layers = open(P("conv_layers.txt")).read()
    m = re.search(r"conv us per group (\d+) \(yolo (\d+), reid (\d+)\); overall (\d+) TFLOP/s", layers)
    conv, yolo, reid, ach1 = (float(m.group(i)) for i in range(1, 5))
    sq = open(P("sq_counters.txt")).read()
    m = re.search(r"MFMA busy ([\d.]+) % of the busy cycles \(ReID trunk kernels[^:]*: ([\d.]+) %", sq)
    busy, busy_r = m.group(1), m.group(2)
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    ks = open(P("kernel_summary.txt")).read()
    trk = 0.0
    for name in ("trk_epoch_kernel", "trk_epoch_prep_kernel", "gallery_commit_kernel"):
        mm = re.search(name + r"\s+(\d+)\s+[\d.]+\s+[\d.]+\s+([\d.]+)", ks)
        trk += float(mm.group(2)) * 32 / 1e3 if mm else 0.0          # 32 epochs of 16 frames per 512-frame group
    own = next(v for k, v in c.items() if k.startswith("own_detections_trained"))
    tex = next(v for k, v in c.items() if k.startswith("own_detections_seeded"))
    pl = c["plugin_loop"]
    curve = " / ".join(f"{sp(v['fps'])} ({v['latency_ms_p50']:.1f} ms)" for k, v in sorted(c["by_launch_group_frames(from host, 3 passes each)"].items(), key=lambda kv: int(kv[0])))
    cpu = b["cpu_baseline"]
    f16 = own.get("fp16_vs_fp32_same_engine", {})
    sub = {
        "VALUE": sp(b["value"]), "VALUE_RANGE": RANGE, "ACH": f"{r['achieved']:.0f}", "FRAC": f"{r['frac']:.3f}",
        "ACH1": f"{ach1:.0f}", "FRAC1": f"{ach1 / 2500:.3f}",
        "CONV_MS": f"{conv / 1e3:.2f}", "YOLO_MS": f"{yolo / 1e3:.2f}", "REID_MS": f"{reid / 1e3:.2f}",
        "REID_TF": sp(34.0e3 / (reid / 1e3)), "REID_FRAC": f"{34.0e3 / (reid / 1e3) / 2500:.3f}",
        "YOLO_TF": f"{4.43e3 / (yolo / 1e3):.0f}", "YOLO_FRAC": f"{4.43e3 / (yolo / 1e3) / 2500:.3f}",
        "BUSY": busy, "BUSY_REID": busy_r,
        "TRAFFIC": f"{t['hbm_bytes_per_launch'] / 1e9:.3f}", "TRAFFIC_ALG": f"{t['algorithmic_bytes_per_launch_same_basis'] / 1e9:.3f}",
        "TRK_MS": f"{trk:.1f}",
        "OWN_FPS": sp(own["fps"]), "OWN_FRAC": f"{own['fraction_of_value']:.3f}", "OWN_DEV": sp(own["association_frames(device, host)"][0]),
        "OWN_RECALL": f"{100 * own['detector_vs_planted(fp16, first 256 frames)']['recall']:.1f} %",
        "OWN_REPRO": f"{100 * f16.get('reproduced_fraction', 0):.2f} %", "OWN_IDSW": str(f16.get("id_switches")),
        "TEX_DEV": sp(tex["device_filter"]["fps"]), "TEX_HOST": sp(tex["host_filter"]["fps"]),
        "PLUG_FPS": sp(pl["fps"]), "PLUG_P50": f"{pl['latency_ms_p50']:.2f}",
        "CURVE": curve, "CPU16": f"{cpu['value']:.2f}", "CPU1": f"{cpu['one_thread']['value']:.2f}", "TAG": TAG,
    }
    rows = TEMPLATE
    for k, v in sub.items():
        rows = rows.replace("@" + k + "@", v)
    left = re.findall(r"@[A-Z0-9_]+@", rows)
    assert not left, left
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a, z = s.index("<!-- numbers:begin -->"), s.index("<!-- numbers:end -->")
    s = s[:a] + "<!-- numbers:begin -->\n\n" + rows + "\n\n" + s[z:]
    open(path, "w").write(s)
    print(rows)


if __name__ == "__main__":
    main()
