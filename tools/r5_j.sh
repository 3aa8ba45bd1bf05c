#!/bin/bash
# GPU box: the N > 1 code path of bench.py as far as one GPU allows: world size 1 forced through the exchange (native, then
# torch.distributed), and the 2-rank gloo plumbing run; host overhead again after the Python trimming.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
export HSA_ENABLE_IPC_MODE_LEGACY=0
P="--envs 512 --steps 300 --warmup 50 --burn-in 200 --sustained-steps 300 --kernel-reps 20 --no-cpu-baseline --no-extra"
CRAFTER_BENCH_NATIVE_EXCHANGE=1 CRAFTER_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py $P > $out/r5j_forced_native.json 2> $out/r5j_forced_native.err; echo rc $?
CRAFTER_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py $P > $out/r5j_forced_py.json 2> $out/r5j_forced_py.err; echo rc $?
CRAFTER_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --envs 128 --steps 20 --warmup 5 --burn-in 50 --sustained-steps 40 --kernel-reps 10 > $out/r5j_gloo2.log 2>&1; echo rc $?
python - <<'PY'
import json
for f in ('r5j_forced_native', 'r5j_forced_py'):
  try:
    d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, 'value %.2f M' % (d['value'] / 1e6), 'host_us %.1f' % d['host_us_per_step'], 'ms/step %.4f' % d['ms_per_step'], d['config']['exchange_enqueued_by'], 'blocks16', d['exchange_blocks_of_16'] and round(d['exchange_blocks_of_16']['value'] / 1e6, 2), d['parity']['bit_exact'])
  except Exception as e:
    print(f, 'FAILED', e)
PY
tail -3 $out/r5j_gloo2.log | cut -c1-600
python tools/host_overhead_dist.py 512 2>&1 | grep "envs," | tee $out/r5j_host_overhead_dist.txt
