#!/bin/bash
# interleaved A/B of the headline bench under an environment switch: tools/ab_env.sh "AICAM_NO_TAIL=1" [rounds] [rev]
# (a third argument runs the switch first in every pair: a first-of-pair effect shows up as the sign flipping)
B="--no-curve --no-own --no-plugin --cpu-frames 0"
for i in $(seq 1 ${2:-3}); do
for cfg in $([ -n "$3" ] && echo "$1 X=1" || echo "X=1 $1"); do
  env $cfg python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
done
done
