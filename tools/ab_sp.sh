#!/bin/bash
# A/B of the software-pipelined patch kernel (v6, kernels_conv_sp.hip; in step and with the half-body offset AICAM_SP_SKEW=1) against the
# ping-pong one (v5, AICAM_NO_SP=1) on the ReID layer2 / 3 / 4 shapes, each layer alone (tools/conv_bench.py, CB_NET: the stem
# subtracted), 15 360 crops = one 512-frame launch group: tools/ab_sp.sh
cd "$(dirname "$0")/.."
export CB_NET=1
for shape in "32 16 128 128" "16 8 256 256" "8 4 512 512"; do
  for res in 0 1; do
    for sw in "" "AICAM_SP_SKEW=1" "AICAM_NO_SP=1"; do
      env $sw python tools/conv_bench.py $shape 3 15360 8 $res | sed 's/wall.*NET/NET/'
    done
  done
done
