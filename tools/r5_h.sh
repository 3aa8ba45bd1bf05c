#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
for rep in 1 2; do
for v in CRAFTER_LDS_PAD=280 CRAFTER_LDS_PAD=0 ; do
  env $v python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.2f' % d['roofline']['kernel_us'], d['world_pool'])"
done
done 2>&1 | tee $out/r5h_closed_occupancy_ab.txt
for v in CRAFTER_LDS_PAD=280 CRAFTER_LDS_PAD=0 ; do
  env $v python bench.py --envs 1024 --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1024 envs $v', 'value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.2f' % d['roofline']['kernel_us'])"
done 2>&1 | tee -a $out/r5h_closed_occupancy_ab.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q --timeout 180 2>&1 | tail -3
