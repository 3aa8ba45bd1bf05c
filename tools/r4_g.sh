#!/bin/bash
tag=${1:-r4g}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
CRAFTER_HIP_LIB=$root/gpurun_ab/balance_probe.so timeout 300 python tools/gpu_balance_probe.py 8192 --area 256 > $out/${tag}_balance_probe_cfg4.txt 2>&1; tail -3 $out/${tag}_balance_probe_cfg4.txt
CRAFTER_HIP_LIB=$root/gpurun_ab/balance_probe.so timeout 300 python tools/gpu_balance_probe.py 4096 > $out/${tag}_balance_probe_4096.txt 2>&1; tail -3 $out/${tag}_balance_probe_4096.txt
Q="--no-cpu-baseline --no-extra --steps 1000 --warmup 200 --sustained-steps 0 --kernel-reps 100"
for v in 0 1 0 1; do
  CRAFTER_SIMD_BALANCE=$v timeout 200 python bench.py $Q > $out/${tag}_ab_bal$v.json 2> $out/${tag}_ab.err
  python - $out/${tag}_ab_bal$v.json $v <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('simd balance', sys.argv[2], 'value %.2f M' % (j['value'] / 1e6), 'kernel_us %.2f' % j['roofline']['kernel_us'], 'requeue_us %.2f' % j['roofline']['reset_kernel_us'], 'parity', j['parity']['bit_exact'])
PY
done
timeout 400 python bench.py --envs 8192 --area 256 --no-extra --steps 1000 --warmup 100 --burn-in 300 --kernel-reps 100 --no-cpu-baseline --sustained-steps 0 > $out/${tag}_bench_cfg4.json 2> $out/${tag}_bench_cfg4.err
python - $out/${tag}_bench_cfg4.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('cfg4 %.2f M  ms/step %.4f kernel_us %.1f requeue %.1f parity %s pool %s' % (j['value'] / 1e6, j['ms_per_step'], j['roofline']['kernel_us'], j['roofline']['reset_kernel_us'], j['parity']['bit_exact'], j['world_pool']))
PY
