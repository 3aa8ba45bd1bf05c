#!/bin/bash
# GPU box: the 512-thread step kernel for small batches (crafter_step_wide_kernel, CRAFTER_STEP_WIDE=1 / 0): the GPU suite
# with it (every test of <= 768 envs on the default instance runs it), then the A/B by batch size.
tag=${1:-r4zy}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
grep -n "passed\|failed\|rc " $out/${tag}_pytest_gpu.txt | tail -3
Q="--no-cpu-baseline --no-extra --no-parity --steps 1500 --warmup 300 --sustained-steps 0 --kernel-reps 100"
run() { label=$1; shift; timeout 200 python bench.py $Q "$@" 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], d['roofline'].get('kernel'))" | tee -a $out/${tag}_wide_ab.txt; }
for i in 1 2; do
  for n in 512 768 256 1024; do
    CRAFTER_STEP_WIDE=0 run "256 threads envs=$n" --envs $n
    CRAFTER_STEP_WIDE=1 run "512 threads envs=$n" --envs $n
  done
done
