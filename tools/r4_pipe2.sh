#!/bin/bash
tag=${1:-r4u}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
Q="--no-cpu-baseline --no-extra --steps 1000 --warmup 200 --sustained-steps 0 --kernel-reps 100"
run() { label=$1; shift
  env "$@" timeout 200 python bench.py $Q 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], d['roofline']['kernel'], 'parity', d['parity']['bit_exact'])" | tee -a $out/${tag}_pipe2.txt
}
for i in 1 2; do
run fused CRAFTER_HIP_LIB=$root/gpurun_ab/pipe_cap.so
run pipe_cap_static1024 CRAFTER_HIP_LIB=$root/gpurun_ab/pipe_cap.so CRAFTER_PIPE=1 CRAFTER_PIPE_STATIC=1 CRAFTER_PIPE_GRID=1024
run pipe_nocap_static1024 CRAFTER_HIP_LIB=$root/gpurun_ab/pipe_nocap.so CRAFTER_PIPE=1 CRAFTER_PIPE_STATIC=1 CRAFTER_PIPE_GRID=1024
run pipe_nocap_tickets1024 CRAFTER_HIP_LIB=$root/gpurun_ab/pipe_nocap.so CRAFTER_PIPE=1 CRAFTER_PIPE_GRID=1024
done
C="--envs 16384 --no-render --no-extra --steps 1000 --warmup 100 --burn-in 300 --kernel-reps 100 --no-cpu-baseline --no-parity --sustained-steps 0"
for i in 1 2; do
CRAFTER_HIP_LIB=$root/gpurun_ab/pipe_cap.so timeout 200 python bench.py $C 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg5 head', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], 'requeue %.2f' % d['roofline']['reset_kernel_us'])" | tee -a $out/${tag}_pipe2.txt
done
