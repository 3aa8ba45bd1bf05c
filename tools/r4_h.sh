#!/bin/bash
tag=${1:-r4h}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
tail -8 $out/${tag}_pytest_gpu.txt
timeout 400 python bench.py --envs 8192 --area 256 --no-extra --steps 1000 --warmup 100 --burn-in 300 --kernel-reps 100 --no-cpu-baseline --sustained-steps 0 > $out/${tag}_bench_cfg4.json 2> $out/${tag}_bench_cfg4.err
python - $out/${tag}_bench_cfg4.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('cfg4 %.2f M  ms/step %.4f kernel_us %.1f requeue %.1f parity %s pool %s' % (j['value'] / 1e6, j['ms_per_step'], j['roofline']['kernel_us'], j['roofline']['reset_kernel_us'], j['parity']['bit_exact'], j['world_pool']))
PY
Q="--no-cpu-baseline --no-extra --steps 1000 --warmup 200 --sustained-steps 0 --kernel-reps 100"
timeout 200 python bench.py $Q > $out/${tag}_bench_quick.json 2> $out/${tag}_ab.err
python - $out/${tag}_bench_quick.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('default value %.2f M' % (j['value'] / 1e6), 'kernel_us %.2f' % j['roofline']['kernel_us'], 'requeue_us %.2f' % j['roofline']['reset_kernel_us'], 'parity', j['parity']['bit_exact'])
PY
timeout 200 python tools/host_overhead_dist.py 512 > $out/${tag}_host_overhead_dist.txt 2>&1; grep "envs\|scatter" $out/${tag}_host_overhead_dist.txt
