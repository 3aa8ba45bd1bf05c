#!/bin/bash
# GPU box (round 6, probe build): wave priority of the generation kernels against the steady state and the driver's window
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
line() {
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2> /tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s value %.2f M  stream %.2f M  sustained %.2f M  kernel %.2f us' % ('$label', d['value'] / 1e6, 4096 / d['launch_stream_ms_per_step'] / 1e3, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
  done
  env "$@" timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | cut -c1-100 | tail -1
}
line base X=1
line classify_prio1 CRAFTER_GEN_CLASSIFY_PRIO=1
line classify_prio2 CRAFTER_GEN_CLASSIFY_PRIO=2
line classify_prio3 CRAFTER_GEN_CLASSIFY_PRIO=3
line all_prio2 CRAFTER_GEN_CLASSIFY_PRIO=2 CRAFTER_GEN_SERIAL_PRIO=2
line all_prio3 CRAFTER_GEN_CLASSIFY_PRIO=3 CRAFTER_GEN_SERIAL_PRIO=3
