#!/bin/bash
# own detections (trained detector) against planted boxes, same clip, same streams: kernel totals of both under the rocprofv3 kernel trace,
# then both without the profiler.   gpurun -- bash tools/ab_own.sh
set -e
R=$PWD
O=$R/gpurun_out/ab_own
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for inj in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$inj -- python3 $R/tools/own_trained.py $inj 2 split_streams=1 > $O/run$inj.txt 2> $O/run$inj.err
  cp $(ls $O/t$inj/*/*_kernel_stats.csv | head -1) $O/kernel_stats_inject$inj.csv
  python3 $R/tools/trace_streams.py $(ls $O/t$inj/*/*_kernel_trace.csv | head -1) 0.34 timeline > $O/streams_inject$inj.txt 2>&1 || true
  rm -rf $O/t$inj
done
cd $R
for inj in 0 1; do python3 tools/own_trained.py $inj 2 split_streams=1 >> $O/plain.txt 2>&1; done
cat $O/run0.txt $O/run1.txt $O/plain.txt
