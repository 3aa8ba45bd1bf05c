#!/bin/bash
# GPU box: the headline (4096 envs) against the classification grid, re-measured on the final kernels (default instance at 20 LDS
# granules, classification at 80 VGPRs, one-wave kernels at priority 1).  Needs gpurun_ab/probes.so.
cd ${GRAFT_REPO_ROOT:-.}
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
for i in 1 2 3 4; do for g in 256 272 288 304; do
  CRAFTER_GEN_CLASSIFY_GRID=$g timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid $g: window %.2f M  sustained %.2f M  kernel %.2f us' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
done; done
