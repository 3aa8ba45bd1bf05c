#!/bin/bash
# GPU box: the slot table of configs[3] in global memory (FarSlot) -- parity of the 256x256 tests, then A/B of builds in gpurun_ab/
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rollout.py -x -q -m gpu -k "config4 or 256 or generic or large or mutated" 2>&1 | tail -5
for lib in "$@"; do
  echo "== $lib"
  export CRAFTER_HIP_LIB=gpurun_ab/$lib.so
  timeout 300 python tools/gpu_phase_means.py 8192 --area 256 2>&1 | head -9
  for i in 1 2; do timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('cfg4 value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.1f' % d['roofline']['kernel_us'])"; done
done
