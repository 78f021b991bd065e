#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh other.so [rounds] -- the headline bench under the in-tree libaicam.so
# ("new") and under other.so ("old") copied over it, interleaved; the in-tree file is restored at the end.  CB="H W CIN COUT K items reps res":
# tools/conv_bench.py on that shape under each as well.
cd "$(dirname "$0")/.."
B="--no-curve --no-own --no-plugin --cpu-frames 0"
L=ai-camera_amd/libaicam.so
cp $L /tmp/ab_new_lib.so
for i in $(seq 1 ${2:-3}); do
for cfg in new old; do
  if [ $cfg = old ]; then cp "$1" $L; else cp /tmp/ab_new_lib.so $L; fi
  python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
  [ -n "$CB" ] && CB_NET=1 python tools/conv_bench.py $CB | sed 's/wall.*NET/NET/'
done
done
cp /tmp/ab_new_lib.so $L
