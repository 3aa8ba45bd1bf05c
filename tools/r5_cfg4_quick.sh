#!/bin/bash
# GPU box: configs[3] (8192 envs x 256x256) quick look -- phases, the balance pass's parts (needs gpurun_ab/balance_probe.so), bench
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout 300 python tools/gpu_phase_means.py 8192 --area 256 2>&1 | head -9
CRAFTER_HIP_LIB=gpurun_ab/balance_probe.so timeout 300 python tools/gpu_balance_probe.py 8192 --area 256 2>&1 | grep -v amdgpu | head -12
for i in 1 2; do timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('cfg4 value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.1f' % d['roofline']['kernel_us'])"; done
