set -e
timeout -k 10 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; }
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f c16
AICAM_NO_C16=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f noc16
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f c16
AICAM_NO_C16=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f noc16
