#!/bin/bash
# GPU box (round 6): from which batch size on does crafter_step_early_kernel pay?  (product build; CRAFTER_STEP_EARLY override)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for n in 1536 2048 3072 4096 8192; do for i in 1 2; do for e in 0 1; do
  CRAFTER_STEP_EARLY=$e timeout 200 python bench.py --envs $n --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('envs %5d  early %d  value %.2f M  sustained %.2f M  kernel %.2f us' % ($n, $e, d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
done; done; done
