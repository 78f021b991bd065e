set -e
AICAM_PPP=7 AICAM_PP_MIN=0 timeout -k 10 300 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
for cfg in "32 16 128 128 3" "16 8 256 256 3" "8 4 512 512 3"; do
  AICAM_PPP=6 timeout -k 10 120 python tools/conv_bench.py $cfg 1920 8 0 2>&1 | tail -1
  AICAM_PPP=14 timeout -k 10 120 python tools/conv_bench.py $cfg 1920 8 0 2>&1 | tail -1
  AICAM_PPP=6 timeout -k 10 120 python tools/conv_bench.py $cfg 1920 8 1 2>&1 | tail -1
done
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; }
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f base
AICAM_PPP=6 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f ppp6
AICAM_PPP=2 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f ppp2
