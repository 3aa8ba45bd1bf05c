#!/bin/bash
tag=${1:-r4s}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
C="--envs 16384 --no-render --no-extra --steps 1000 --warmup 100 --burn-in 300 --kernel-reps 100 --no-cpu-baseline --no-parity --sustained-steps 0"
for i in 1 2; do for v in "$@"; do
  CRAFTER_HIP_LIB=$root/gpurun_ab/$v.so timeout 200 python bench.py $C 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], 'requeue %.2f' % d['roofline']['reset_kernel_us'])" | tee -a $out/${tag}_cfg5_bisect.txt
done; done
