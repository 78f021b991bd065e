#!/bin/bash
# A/B of the LDS-DMA implicit GEMM's ring depth on YOLOv8n-shaped thin layers (512 frames), run on a GPU box: tools/yolo_layers_ab.sh > gpurun_out/ab.txt
export CB_NET=1
for ns in 4 3 2; do
  for sh in "80 80 96 64 1" "80 80 128 64 1" "80 80 192 64 1" "40 40 192 128 1" "40 40 384 128 1" "80 80 64 80 3" "80 80 80 80 3" "40 40 64 64 1" "20 20 384 256 1"; do
    AICAM_DMA_NSTAGE=$ns python tools/conv_bench.py $sh 512 2
  done
done
