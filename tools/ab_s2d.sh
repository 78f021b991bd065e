#!/bin/bash
# A/B of the space-to-depth patch kernel (conv3x3s2_sp_patch_kernel, kernels_conv_sp.hip) against the im2col ping-pong kernel (v4, AICAM_NO_S2D=1) on
# the stride-2 3x3 convs of the ReID trunk (layerN.0.conv1), each layer alone, 15 360 crops: tools/ab_s2d.sh
cd "$(dirname "$0")/.."
export CB_NET=1
for shape in "64 32 64 128" "32 16 128 256" "16 8 256 512"; do
  for sw in "" "AICAM_NO_S2D=1"; do
    env $sw python tools/conv_bench.py $shape 3 15360 1 0 2 | sed 's/wall.*NET/NET/'
  done
done
