#!/bin/bash
# GPU box: the classification kernel bounded to 96 / 80 VGPRs (five / six waves per SIMD: it spills a dozen / two dozen registers)
# against the 110 it takes when left alone -- what a resident generation wave displaces is step waves, by its registers.
# headline (4096 envs): driver's window + sustained; configs[3] (8192 x 256x256): sustained.  Needs gpurun_ab/cw1.so cw5.so cw6.so.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for i in 1 2; do for v in cw1 cw5 cw6; do
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v headline: value %.2f M  sustained %.2f M  kernel %.2f us' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('$v cfg4: value %.2f M  sustained %.2f M  kernel_us %.1f' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | cut -c1-120
done; done
