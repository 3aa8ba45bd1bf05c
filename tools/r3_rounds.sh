#!/bin/bash
# GPU box: does a step launch last ceil(envs per CU / resident workgroups) rounds?  Step kernel time at batch sizes around
# the multiples of 1280 (= 256 CUs x 5 resident step workgroups).
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
Q="--no-cpu-baseline --no-parity --no-extra --steps 600 --warmup 100 --sustained-steps 0 --kernel-reps 200"
for n in 2560 3072 3584 3840 4096 4352 5120 6400; do
  timeout 120 python bench.py $Q --envs $n 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($n, 'envs: step kernel %.1f us, %.2f us/step, %.1f M env-steps/s, %.2f ns/env' % (d['roofline']['kernel_us'], 1000*d['ms_per_step'], d['value']/1e6, 1e6*d['ms_per_step']/$n))"
done | tee $out/rounds.txt
