#!/bin/bash
tag=${1:-r4l}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
tail -8 $out/${tag}_pytest_gpu.txt
Q="--no-cpu-baseline --no-extra --steps 1500 --warmup 300 --sustained-steps 0 --kernel-reps 100"
for v in 1 0 1 0; do
  for n in 4096 1024; do
  CRAFTER_NOISE_AHEAD=$v timeout 200 python bench.py $Q --envs $n 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('noise ahead $v envs $n', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], 'requeue %.2f' % d['roofline']['reset_kernel_us'], 'parity', d['parity']['bit_exact'], d['world_pool']['regenerated_inline'])" | tee -a $out/${tag}_noise_ab.txt
  done
done
timeout 200 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1; head -10 $out/${tag}_phases_4096.txt
