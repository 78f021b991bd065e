B="--no-curve --no-own --no-plugin --cpu-frames 0"
for i in 1 2; do
for cfg in "--batch 512" "--batch 256" "--batch 384" "--batch 128"; do
  python bench.py $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['roofline']['frac'], d['config']['frame_latency_ms(handed to the pipeline -> tuples on host, full launch groups of the timed run)'])"
done
done
AICAM_CONV_CUS=240 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CUS240', d['value'])"
AICAM_CONV_CUS=224 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CUS224', d['value'])"
AICAM_TRK_NOWAVE=1 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NOWAVE', d['value'])"
