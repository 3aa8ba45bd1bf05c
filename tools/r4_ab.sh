#!/bin/bash
# GPU box: A/B of library variants (gpurun_ab/*.so) and of the regeneration kernel's grid on the metric workload.
tag=${1:-r4k}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
Q="--no-cpu-baseline --no-extra --no-parity --steps 1500 --warmup 300 --sustained-steps 0 --kernel-reps 100"
run() {   # label, env assignments...
  label=$1; shift
  env "$@" timeout 200 python bench.py $Q 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], 'requeue %.2f' % d['roofline']['reset_kernel_us'])" | tee -a $out/${tag}_ab.txt
}
for i in 1 2; do
  for v in "$@"; do run $v CRAFTER_HIP_LIB=$root/gpurun_ab/$v.so; done
  run base_requeue1 CRAFTER_HIP_LIB=$root/gpurun_ab/base.so CRAFTER_REQUEUE_GRID=1
done
