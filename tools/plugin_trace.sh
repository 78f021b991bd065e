#!/bin/bash
# per-frame plugin loop under the rocprofv3 kernel trace: kernel time against gaps inside one frame.   gpurun -- bash tools/plugin_trace.sh
set -e
R=$PWD
O=$R/gpurun_out/plug_trace
rm -rf $O && mkdir -p $O
python3 tools/plugin_phases.py 200 > $O/phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/plugin_phases.py 60 > $O/phases_traced.txt 2> $O/trace.err
cd $R
python3 - <<'PY' > $O/frames.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/plug_trace/t/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames: split at yolo_stem_fused launches
idx = [i for i, r in enumerate(rows) if "yolo_stem_fused" in r["Kernel_Name"]]
for a, b in list(zip(idx, idx[1:]))[-3:]:
    fr = rows[a:b]
    t0 = int(fr[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr) / 1e3
    span = (int(fr[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"frame: {len(fr)} launches, kernel time {busy:.1f} us, first start -> last end {span:.1f} us, next frame starts {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us after this one")
    prev = t0
    for r in fr:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"   +{(s - t0) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:6.1f}  grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>4s}  {r['Kernel_Name'].split('(')[0][-60:]}")
        prev = e
PY
rm -rf $O/t
cat $O/phases.txt | tail -12
