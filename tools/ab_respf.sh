#!/bin/bash
# residual prefetch of the ping-pong patch kernel (AICAM_PPP_PF = taps of the last chunk that carry a pass): ReID layer2/3/4 conv2 shapes alone
set -e
for shape in "32 16 128 128" "16 8 256 256" "8 4 512 512"; do
  for pf in 0 0xC0 0x180 0x41 0x81 0x1C0; do
    echo -n "shape $shape pf=$pf: "
    CB_NET=1 AICAM_PPP_PF=$pf python tools/conv_bench.py $shape 3 15360 8 1 2>&1 | tail -1
  done
done
