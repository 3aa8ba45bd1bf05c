#!/bin/bash
# GPU box: configs[3] with the one-wave generation kernels at wave priority 1: classification priority x classification grid.  Needs gpurun_ab/probes.so.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
for i in 1 2; do for cp in 0 1; do for g in 384 512 768; do
  CRAFTER_GEN_SERIAL_PRIO=1 CRAFTER_GEN_CLASSIFY_PRIO=$cp CRAFTER_GEN_CLASSIFY_GRID=$g timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('serial 1 classify prio $cp grid $g  value %.2f M  sustained %.2f M  kernel_us %.1f' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
done; done; done
