set -e
AICAM_PP=9 AICAM_PP_MIN=0 timeout -k 10 300 python -m pytest tests/test_gpu_nets.py -x -q 2>&1 | tail -3
for cfg in "32 16 128 128 3" "32 16 128 128 1"; do
  python tools/conv_bench.py $cfg 960 8 0
  AICAM_PP=3 timeout -k 10 120 python tools/conv_bench.py $cfg 960 8 0
  AICAM_PP=9 timeout -k 10 120 python tools/conv_bench.py $cfg 960 8 0
done
