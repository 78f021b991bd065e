set -e
for cfg in "40 40 64 64 3" "20 20 128 128 3"; do
  timeout -k 10 120 python tools/conv_bench.py $cfg 128 8 0 2>&1 | tail -1
  AICAM_PATCH_ALL=1 timeout -k 10 120 python tools/conv_bench.py $cfg 128 8 0 2>&1 | tail -1
done
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; }
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f new
AICAM_NO_PATCH_C32=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f noc32
