#!/bin/bash
# On the GPU box: alternate library builds (gpurun_ab/<name>.so) through a short bench.py run (no parity, no extras).
# usage: tools/ab_quick.sh name1 name2 ...   -> M env-steps/s, us per step, step kernel us, queue kernel us
for i in 1 2; do for v in "$@"; do
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 200 python bench.py --steps 1500 --warmup 300 --no-cpu-baseline --no-parity --no-extra --sustained-steps 0 2> /tmp/ab_err.txt | python -c "
import sys, json
try:
  d = json.loads(sys.stdin.read().strip().splitlines()[-1])
  print('$v', round(d['value'] / 1e6, 2), round(d['ms_per_step'] * 1000, 2), round(d['roofline']['kernel_us'], 2), round(d['roofline']['reset_kernel_us'], 2))
except Exception as e:
  print('$v', 'FAILED', e); print(open('/tmp/ab_err.txt').read()[-600:])"
done; done
