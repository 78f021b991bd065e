set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
AICAM_PP_MIN=0 timeout -k 10 300 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
for cfg in "8 4 512 512 3" "16 8 256 256 3" "32 16 128 128 3" "64 32 64 64 3"; do
  timeout -k 10 120 python tools/conv_bench.py $cfg 960 8 0 2>&1 | tail -1
done
python bench.py --steps 8 --warmup 2 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | cut -c1-200
