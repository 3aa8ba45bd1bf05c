#!/bin/bash
# GPU box: quick look at the metric workload -- GPU suite, phase means, rollout, two closed-loop bench lines
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 300 python tools/gpu_phase_means.py 4096 2>&1 | head -9 | tail -7
timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | cut -c1-200
for i in 1 2; do python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('closed value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.2f' % d['roofline']['kernel_us'])"; done
