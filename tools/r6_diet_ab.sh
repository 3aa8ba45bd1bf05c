#!/bin/bash
# GPU box: the default instance's workgroup at 20 LDS granules instead of 21 (a classification workgroup of the world pool fits
# BESIDE six of them), with the classification kernel left at 110 VGPRs / bounded to 96 / 80 (six early-frame step waves leave a
# SIMD 80 registers), and with the plain step kernel (62 VGPRs: six leave 128).  Needs gpurun_ab/cw1.so (before) diet.so diet_cw5.so diet_cw6.so.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_pool.py tests/test_gpu_rollout.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -3
line() {
  v=$1; shift
  env CRAFTER_HIP_LIB=gpurun_ab/$v.so "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-22s window %.2f M  sustained %.2f M  kernel %.2f us' % ('$v', '$*', d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
  env CRAFTER_HIP_LIB=gpurun_ab/$v.so "$@" timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | python -c "
import sys, ast
for l in sys.stdin:
  if l.startswith('{'):
    d = ast.literal_eval(l.strip()); print('%-10s %-22s open loop %.2f M' % ('$v', '$*', d['open_loop_M'])); break"
}
for i in 1 2; do
  line cw1 X=1
  line diet X=1
  line diet_cw5 X=1
  line diet_cw6 X=1
  line cw1 CRAFTER_STEP_EARLY=0
  line diet CRAFTER_STEP_EARLY=0
done
