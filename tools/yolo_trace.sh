#!/bin/bash
# per-kernel stats of the detector alone (512 frames, one stream) with and without a kernel switch: tools/yolo_trace.sh "AICAM_NO_TAIL=1"
R=$PWD; cd /tmp; export TMPDIR=/tmp
for e in "X=1" "$1"; do
  O=$R/gpurun_out/ytrace_$(echo $e | tr -c 'A-Za-z0-9' '_')
  rm -rf $O
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --no-curve --no-own --no-plugin --cpu-frames 0 --single-stream --steps 1 --warmup 1 > /dev/null 2>&1
  echo "== $e"; python3 - $O <<'PY'
import csv,glob,sys,re,collections
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('conv3x3_patch_kernel','c2f16')):
        print(n[:110], r['Calls'], r['TotalDurationNs'], 'avg us', float(r['AverageNs'])/1e3)
PY
done
