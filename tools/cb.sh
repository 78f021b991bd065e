set -e
python tools/stem_unit.py rand 2>&1 | grep "max err"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['achieved'])"; }
python bench.py --steps 8 --warmup 2 --cpu-frames 0 --no-pcie 2>&1 | tail -1 | f stem2
python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-pcie --batch 64 --ring 128 2>&1 | tail -1 | f stem2_b64
