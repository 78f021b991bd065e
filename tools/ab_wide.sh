#!/bin/bash
# ring shapes of the wide-step kernel on the small ReID shapes, and the plugin loop by grid threshold.   gpurun -- bash tools/ab_wide.sh
for cfg in 0 1 2 3; do
  for sh in "8 4 512 512 3 28 20" "16 8 256 256 3 28 20" "32 16 128 128 3 28 20" "20 20 256 256 3 1 20" "40 40 128 128 3 1 20"; do
    AICAM_WIDE_BLOCKS=1024 AICAM_WIDE_CFG=$cfg timeout -k 10 60 python tools/conv_bench.py $sh 2>&1 | tail -1 | cut -c1-110
  done
done
for w in 0 192 512 1024; do
  echo "AICAM_WIDE_BLOCKS=$w"
  AICAM_WIDE_BLOCKS=$w timeout -k 10 120 python tools/plugin_phases.py 200 2>&1 | tail -2
done
