#!/bin/bash
# GPU box: A/B of one build under two environment settings, alternating, same box.
# usage: tools/ab_env.sh "<VAR=a>" "<VAR=b>" [reps] [bench args...]
A=$1; B=$2; n=${3:-3}; shift 3
for i in $(seq $n); do
  for v in "$A" "$B"; do
    env $v python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), 'kernel_us %.2f' % d['roofline']['kernel_us'], 'ms/step %.4f' % d['ms_per_step'])"
  done
done
