#!/bin/bash
# GPU box: configs[3] (8192 envs x 256x256): wave priority of the one-wave-per-world generation kernels (seeding, ordered draws: the
# draws of a 256x256 world are a 7 ms chain) behind a closed-loop step.  Needs gpurun_ab/probes.so (the tree, -DCRAFTER_PROBES).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
for i in 1 2 3; do for p in 0 1 2 3; do
  CRAFTER_GEN_SERIAL_PRIO=$p timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('serial prio $p  value %.2f M  sustained %.2f M  kernel_us %.1f' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
done; done
