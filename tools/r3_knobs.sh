#!/bin/bash
# GPU box: the world pool's two scheduling knobs again, now that classification is a third cheaper and the step launch is
# ordered: batches every P steps (--gen-period), classification kernel width (CRAFTER_GEN_CLASSIFY_GRID).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
Q="--no-cpu-baseline --no-parity --no-extra --steps 1500 --warmup 300 --sustained-steps 0"
run() { python bench.py $Q "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M env-steps/s, %.2f us/step, step kernel %.2f us, inline regenerations %d' % (d['value']/1e6, 1000*d['ms_per_step'], d['roofline']['kernel_us'], d['world_pool']['regenerated_inline']))"; }
for rep in 1 2; do
  for p in 8 16 24 32; do echo -n "gen-period $p: "; run --gen-period $p; done
  for g in 128 192 256 384 512; do echo -n "classify grid $g: "; CRAFTER_GEN_CLASSIFY_GRID=$g run; done
done
