set -e
timeout -k 10 600 python -m pytest tests/test_gpu_pre_tracker.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
AICAM_TRK_KS=1 timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['config']['host_us_per_frame'])"; }
for i in 1 2; do
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f ks2
AICAM_TRK_KS=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f ks1
done
