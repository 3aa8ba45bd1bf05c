#!/bin/bash
# GPU box: configs[3] (8192 envs, 256x256) against the world pool's two knobs on the round-4 layouts.
tag=${1:-r4i}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
C="--envs 8192 --area 256 --no-extra --steps 800 --warmup 100 --burn-in 300 --kernel-reps 50 --no-cpu-baseline --no-parity --sustained-steps 0"
for g in 256 512 1024; do for p in 16 32; do
  CRAFTER_GEN_CLASSIFY_GRID=$g timeout 300 python bench.py $C --gen-period $p > $out/${tag}_k.json 2> $out/${tag}_k.err
  python - $out/${tag}_k.json $g $p <<'PY' | tee -a $out/${tag}_cfg4_knobs.txt
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('classify grid %s period %s: %.2f M env-steps/s, ms/step %.4f, kernel_us %.1f, requeue %.1f, pool %s' % (sys.argv[2], sys.argv[3], j['value'] / 1e6, j['ms_per_step'], j['roofline']['kernel_us'], j['roofline']['reset_kernel_us'], j['world_pool']))
PY
done; done
