#!/bin/bash
# per-layer conv table of the headline workload on ONE stream (the refresh script's second trace alone):  gpurun -- bash tools/layers_now.sh [ENV=1]
R=$PWD; O=$R/gpurun_out/layers_now; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$1" ] && export "$1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --cpu-frames 0 --no-curve --no-own --no-plugin --single-stream > $O/bench.json 2> $O/trace.log
cd $R
python tools/prof_layers.py $(ls -d $O/t/*/ | head -1) 512 15360 100 > $O/conv_layers.txt
python - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/t/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('stem','nms','avgpool','sppf')): print(r['Name'][:60], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
rm -rf $O/t
sed -n 24,75p $O/conv_layers.txt
