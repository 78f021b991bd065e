set -e
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['confirmed_tracks_per_frame'], d['config']['host_us_per_frame'])"; }
python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pcie --model m --width 1920 --height 1080 --persons 100 --ring 128 --batch 64 2>&1 | tail -1 | f cfg2_b64
python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pcie --model m --width 1920 --height 1080 --persons 100 --ring 256 --batch 128 2>&1 | tail -1 | f cfg2_b128
