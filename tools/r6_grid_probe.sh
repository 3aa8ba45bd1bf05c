#!/bin/bash
# GPU box (round 6): width of the classification kernel against the driver's 20-step window and the steady state (probe build)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
for g in 256 320 384 512 768; do
  for i in 1 2; do
    CRAFTER_GEN_CLASSIFY_GRID=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2> /tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid $g: value %.2f M  stream %.2f M  sustained %.2f M  kernel %.2f us' % (d['value'] / 1e6, 4096 / d['launch_stream_ms_per_step'] / 1e3, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
  done
  CRAFTER_GEN_CLASSIFY_GRID=$g timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | cut -c1-100 | tail -1
done
