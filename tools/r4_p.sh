#!/bin/bash
tag=${1:-r4p}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -x > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
tail -4 $out/${tag}_pytest_gpu.txt
Q="--no-cpu-baseline --no-extra --steps 1500 --warmup 300 --sustained-steps 0 --kernel-reps 100"
for v in 1 0 1; do
  for n in 4096 1024 512; do
  CRAFTER_NOISE_AHEAD=$v timeout 200 python bench.py $Q --envs $n 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('noise ahead $v envs $n', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], 'requeue %.2f' % d['roofline']['reset_kernel_us'], 'parity', d['parity']['bit_exact'], d['world_pool']['regenerated_inline'])" | tee -a $out/${tag}_noise_ab.txt
  done
done
timeout 200 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1; head -10 $out/${tag}_phases_4096.txt
for v in 1 0; do
CRAFTER_NOISE_AHEAD=$v timeout 400 python bench.py --envs 8192 --area 256 --no-extra --steps 800 --warmup 100 --burn-in 300 --kernel-reps 50 --no-cpu-baseline --sustained-steps 0 2> /dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg4 noise ahead $v: %.2f M  ms/step %.4f kernel_us %.1f requeue %.1f parity %s' % (j['value'] / 1e6, j['ms_per_step'], j['roofline']['kernel_us'], j['roofline']['reset_kernel_us'], j['parity']['bit_exact']))" | tee -a $out/${tag}_noise_ab.txt
done
