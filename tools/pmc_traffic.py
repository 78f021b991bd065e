#!/usr/bin/env python3
"""HBM bytes per conv launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, as
MI355X_MICROARCH.md prescribes) -> profiles/pmc_traffic.json (read by bench.py's roofline.traffic).
  python tools/pmc_traffic.py <fetch_dir> <write_dir> "<method note>" [tag]   (tag: also profiles/<tag>_pmc_traffic.json) """
import csv, glob, json, os, sys

CONV = ("conv_igemm_dma_kernel", "conv3x3_patch_kernel", "conv_igemm_pp_kernel", "conv3x3_pp_patch_kernel", "conv3x3_sp_patch_kernel", "conv3x3s2_sp_patch_kernel", "conv_igemm_kernel", "conv3x3_c16_kernel", "conv1x1_stream_kernel", "conv3x3_c32s2_tail_kernel", "conv3x3_pm_patch_kernel", "conv3x3_c64_resident_kernel", "conv3x3_c64_block_kernel", "c2f16_fused_kernel")


def per_launch(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in CONV):
            tot += float(r["Counter_Value"])
            n += 1
    return tot / max(n, 1), n


fetch_kib, n1 = per_launch(sys.argv[1], "FETCH_SIZE")
write_kib, n2 = per_launch(sys.argv[2], "WRITE_SIZE")
out = {
    "kernel": "conv class of bench.py: conv_igemm_dma / conv_igemm_pp / conv3x3_pp_patch / conv3x3_sp_patch / conv3x3s2_sp_patch / conv3x3_patch / conv3x3_c16 / conv1x1_stream / conv3x3_c32s2_tail / conv3x3_pm_patch / conv3x3_c64_resident / conv3x3_c64_block / c2f16_fused kernels",
    "launches_sampled": n1,
    "fetch_bytes_per_launch_raw": fetch_kib * 1024,
    "fetch_bytes_per_launch_corrected_x2": fetch_kib * 2048,
    "write_bytes_per_launch": write_kib * 1024,
    "hbm_bytes_per_launch": fetch_kib * 2048 + write_kib * 1024,
    "method": sys.argv[3],
}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import importlib
out["sources_digest"] = importlib.import_module("ai-camera_amd.build").sources_digest()   # bench.py quotes this file only for these sources
try:    # algorithmic bytes of the same workload (library counter, AICAM_NO_TAPER=1 run).  What does not change with the kernels is the
        # TOTAL over the sampled launch groups; per launch it moves whenever launches are merged, so it is re-derived from the total
    prev = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    total = prev.get("algorithmic_bytes_sampled_total")
    if total is None and "algorithmic_bytes_per_launch_same_basis" in prev:
        total = prev["algorithmic_bytes_per_launch_same_basis"] * prev["launches_sampled"]
    if total is not None and n1:
        out["algorithmic_bytes_sampled_total"] = total
        out["algorithmic_bytes_per_launch_same_basis"] = total / n1
except Exception:
    pass
tag = sys.argv[4] if len(sys.argv) > 4 else "r03"
for name in ("pmc_traffic.json", f"{tag}_pmc_traffic.json"):
    json.dump(out, open(os.path.join(root, "profiles", name), "w"), indent=1)
print(json.dumps(out, indent=1))
