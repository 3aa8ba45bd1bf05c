#!/bin/bash
for i in 1 2; do for v in "$@"; do
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 200 python bench.py --steps 1500 --warmup 300 --no-cpu-baseline 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['roofline']['kernel_us'], 2))"
done; done
