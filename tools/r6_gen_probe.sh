#!/bin/bash
# GPU box (round 6): what does world generation cost the step loop, and what moves the driver's 20-step window?
# Needs gpurun_ab/probes.so (tools/ab_make.sh probes tree -DCRAFTER_PROBES).  Prints, per variant: driver-style value
# (--steps 20 --warmup 5, device-sync clock), launch-stream value, sustained (1000 steps), step kernel us.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
line() {   # label, env assignments...
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2> /tmp/err.txt | python -c "
import sys, json
try:
  d = json.loads(sys.stdin.read().strip().splitlines()[-1])
  print('%-28s value %.2f M  stream %.2f M  sustained %.2f M  kernel %.2f us  requeue %.2f us' % ('$label', d['value'] / 1e6, 4096 / d['launch_stream_ms_per_step'] / 1e3, d['sustained']['value'] / 1e6, d['roofline']['kernel_us'], d['roofline']['reset_kernel_us']))
except Exception as e:
  print('$label', 'FAILED', e); print(open('/tmp/err.txt').read()[-800:])"
  done
}
line base X=1
line free_gen CRAFTER_PROBE_FREE_GEN=1
line serial_prio2 CRAFTER_GEN_SERIAL_PRIO=2
line serial_prio3 CRAFTER_GEN_SERIAL_PRIO=3
line classify_grid512 CRAFTER_GEN_CLASSIFY_GRID=512
line classify_grid1024 CRAFTER_GEN_CLASSIFY_GRID=1024
for p in 4 8 32; do
  for i in 1 2; do
    timeout 300 python bench.py --steps 20 --warmup 5 --gen-period $p --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2> /tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s value %.2f M  stream %.2f M  sustained %.2f M  kernel %.2f us' % ('gen_period $p', d['value'] / 1e6, 4096 / d['launch_stream_ms_per_step'] / 1e3, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
  done
done
# open loop with and without the generator
for v in 0 1; do
  CRAFTER_PROBE_FREE_GEN=$v timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | cut -c1-220
done
