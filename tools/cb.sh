set -e
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; }
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f base
AICAM_C64R=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f c64r1
AICAM_C64R=2 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f c64r2
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f base
AICAM_C64R=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f c64r1
