#!/bin/bash
B="--no-curve --no-own --no-plugin --cpu-frames 0"
# images per block of the fused BasicBlock kernel (AICAM_BLK_IPB): headline bench + the layer alone (tools/conv_bench.py, res = 2), two rounds.   gpurun -- bash tools/ab_ipb.sh
for i in 1 2; do
for ipb in 16 15 20 12 30; do
  AICAM_BLK_IPB=$ipb python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ipb $ipb', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
  AICAM_BLK_IPB=$ipb CB_NET=1 python tools/conv_bench.py 64 32 64 64 3 15360 8 2 | sed 's/.*NET/NET/'
done
done
