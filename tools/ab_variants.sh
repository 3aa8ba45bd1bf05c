#!/bin/bash
# On the GPU box: alternate any set of library builds (gpurun_ab/<name>.so) through bench.py.
# usage: tools/ab_variants.sh name1 name2 ...   -> value, us per step, kernel us
for i in 1 2; do for v in "$@"; do
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 200 python bench.py --steps 1500 --warmup 300 --no-cpu-baseline 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['ms_per_step'] * 1000, 2), round(d['roofline']['kernel_us'], 2), round(d['roofline']['reset_kernel_us'], 2))"
done; done
