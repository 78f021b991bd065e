B="--no-curve --no-own --no-plugin --cpu-frames 0"
for i in 1 2 3; do
for cfg in "" "AICAM_NO_RAMP=1"; do
  env $cfg python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['roofline']['frac'])"
done
done
