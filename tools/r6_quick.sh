#!/bin/bash
# GPU box (round 6): generation phase clocks (probe build), pool + parity tests, driver-style and sustained bench lines of the product build
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
CRAFTER_HIP_LIB=gpurun_ab/probes.so python tools/gpu_gen_probe.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_pool.py tests/test_gpu_parity.py tests/test_gpu_noise.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2> /tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('driver-style: value %.2f M  stream %.2f M  sustained %.2f M  kernel %.2f us' % (d['value'] / 1e6, 4096 / d['launch_stream_ms_per_step'] / 1e3, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
done
timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | cut -c1-120
