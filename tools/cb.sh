set -e
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['host_us_per_frame'])"; }
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f prio
AICAM_NO_TRK_PRIO=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f noprio
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f prio
AICAM_NO_TRK_PRIO=1 python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f noprio
