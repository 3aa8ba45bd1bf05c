#!/bin/bash
# GPU box: regeneration beside the step launch (CRAFTER_REGEN_BESIDE=1, crafter_regen_server_kernel) against the kernel behind
# it (=0): the pool tests both ways first (wrapped in timeouts: a protocol error must not hang the box), then the A/B at
# 4096 / 1024 / 512 envs and on configs[4].
tag=${1:-r4v}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_configs.py -x -q -m gpu > $out/${tag}_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/${tag}_pytest.txt
tail -5 $out/${tag}_pytest.txt
Q="--no-cpu-baseline --no-extra --no-parity --steps 1500 --warmup 300 --sustained-steps 0 --kernel-reps 100"
run() {   # label, bench args; env from the caller
  label=$1; shift
  timeout 200 python bench.py $Q "$@" 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', '%.2f M' % (d['value'] / 1e6), 'us/step %.2f' % (d['ms_per_step'] * 1000), 'kernel %.2f' % d['roofline']['kernel_us'], 'requeue %.2f' % d['roofline']['reset_kernel_us'])" | tee -a $out/${tag}_ab.txt
}
for i in 1 2; do
  for b in 1 0; do
    export CRAFTER_REGEN_BESIDE=$b
    run "beside=$b envs=4096"
    run "beside=$b envs=1024" --envs 1024
    run "beside=$b envs=512" --envs 512
  done
done
for b in 1 0; do
  export CRAFTER_REGEN_BESIDE=$b
  run "beside=$b envs=16384 render off" --envs 16384 --no-render
  run "beside=$b envs=8192 area 256" --envs 8192 --area 256 --steps 200 --warmup 50
done
