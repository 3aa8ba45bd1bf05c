#!/bin/bash
# On the GPU box: alternate old/new builds (tools/ab_build.sh) through bench.py, print value + kernel µs.
n=${1:-3}; shift
for i in $(seq $n); do
  for v in old new; do
    CRAFTER_HIP_LIB=gpurun_ab/$v.so python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra "$@" 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), d['roofline'].get('kernel_us'), d['roofline'].get('reset_kernel_us'))"
  done
done
