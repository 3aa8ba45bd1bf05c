#!/bin/bash
# GPU box: dispatch order on / off (CRAFTER_ORDER), same box, alternating; then the GPU tests that go through it.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
Q="--no-cpu-baseline --no-parity --no-extra --steps 1500 --warmup 200 --sustained-steps 0"
for i in 1 2; do for o in 0 1; do
  CRAFTER_ORDER=$o timeout 120 python bench.py $Q "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order=$o: %.2f M env-steps/s, %.2f us/step, step kernel %.2f us + queue kernel %.2f us' % (d['value']/1e6, 1000*d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['reset_kernel_us']))"
done; done
