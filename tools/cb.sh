set -e
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['confirmed_tracks_per_frame'], d['config']['host_us_per_frame'])"; }
python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f cont
python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-pcie --per-step-calls 2>&1 | tail -1 | f perstep
python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f cont
