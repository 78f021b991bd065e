set -e
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['host_us_per_frame'])"; }
time (python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f r128)
time (python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie --ring 256 2>&1 | tail -1 | f r256)
time (python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pcie --ring 512 2>&1 | tail -1 | f r512)
time (python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pcie --ring 512 --batch 128 2>&1 | tail -1 | f r512b128)
