#!/bin/bash
# GPU box: same-box A/B of library builds gpurun_ab/<name>.so at several batch sizes (closed loop, 1500 steps + 1000 sustained)
# usage: tools/r6_ab.sh "4096 1024" name1 name2 ...
sizes=$1; shift
for n in $sizes; do for i in 1 2; do for v in "$@"; do
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 200 python bench.py --envs $n --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2> /tmp/ab_err.txt | python -c "
import sys, json
try:
  d = json.loads(sys.stdin.read().strip().splitlines()[-1])
  print('%-10s envs %5d  value %.2f M  sustained %.2f M  kernel %.2f us' % ('$v', $n, d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))
except Exception as e:
  print('$v', 'FAILED', e); print(open('/tmp/ab_err.txt').read()[-600:])"
done; done; done
