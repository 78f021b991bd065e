#!/bin/bash
# What the NMS kernel of a launch group runs beside: kernel trace of the headline workload on ONE stream, every select_sort_nms launch with its
# grid, duration and the kernels whose intervals overlap it.   gpurun -- bash tools/nms_in_pipeline.sh [ENV=1]
R=$PWD; O=$R/gpurun_out/nms_pipe; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$1" ] && export "$1"
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --cpu-frames 0 --no-curve --no-own --no-plugin --single-stream --steps 2 --warmup 1 > $O/bench.json 2> $O/trace.log
cd $R
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/t/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:48], r['Queue_Id'], r.get('Grid_Size_X', r.get('Grid_Size', '?'))) for r in rows]
iv.sort()
nms = [v for v in iv if 'select_sort_nms' in v[2]]
for s, e, n, q, g in nms:
    ov = [(max(s, a), min(e, b), nm, qq) for a, b, nm, qq, _ in iv if a < e and b > s and 'select_sort_nms' not in nm]
    prev = [v for v in iv if v[3] == q and v[1] <= s][-1:]
    print(f"nms grid {g:>8} q{q} dur {(e - s) / 1e3:8.1f} us  gap after previous on its queue {((s - prev[0][1]) / 1e3) if prev else -1:7.1f} us ; beside: " +
          ", ".join(f"{nm.split('(')[0][-28:]}[q{qq}] {(b - a) / 1e3:.0f}us" for a, b, nm, qq in ov[:6]))
PY
rm -rf $O/t
