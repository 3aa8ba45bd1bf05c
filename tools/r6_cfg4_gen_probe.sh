#!/bin/bash
# GPU box: configs[3] (8192 envs x 256x256) with and without its world generator (CRAFTER_PROBE_FREE_GEN=1: batches stamp their
# requests ready without generating -- a TIMING probe), at six and seven step workgroups per CU.  Needs gpurun_ab/probes_b6.so / probes_b7.so.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for lib in probes_b6 probes_b7; do
  for fg in 0 1; do
    for i in 1 2; do
      CRAFTER_HIP_LIB=gpurun_ab/$lib.so CRAFTER_PROBE_FREE_GEN=$fg timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('$lib free_gen=$fg  value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.1f' % d['roofline']['kernel_us'])"
    done
  done
done
