set -e
timeout -k 10 600 python -m pytest tests/test_gpu_nets.py -x -q -s -k "large_batch" 2>&1 | tail -6
f() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline'], d['config']['host_us_per_frame'])"; }
python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pcie --dtype fp32 --ring 128 --batch 64 2>&1 | tail -1 | f fp32
python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pcie --model m --width 1920 --height 1080 --persons 100 --ring 64 --batch 32 2>&1 | tail -1 | f cfg2
