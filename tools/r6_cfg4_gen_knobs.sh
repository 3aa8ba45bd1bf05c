#!/bin/bash
# GPU box: configs[3] (8192 envs x 256x256), the world pool's knobs re-measured now that six step workgroups share a CU:
# classification grid (the library: 2 x 256 for worlds whose maps stay in HBM), generation period, wave priority.  Needs gpurun_ab/probes_b6.so.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export CRAFTER_HIP_LIB=gpurun_ab/probes_b6.so
line() {
  label=$1; shift
  env "$@" timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('%-24s value %.2f M  sustained %.2f M  kernel_us %.1f' % ('$label', d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
}
line base X=1
for g in 128 256 384 768 1024 2048; do line grid_$g CRAFTER_GEN_CLASSIFY_GRID=$g; done
for p in 8 32; do EXTRA="--gen-period $p" line period_$p X=1; done
EXTRA=
line serial_prio2 CRAFTER_GEN_SERIAL_PRIO=2
line lag2 CRAFTER_GEN_LAG=2
line lag5 CRAFTER_GEN_LAG=5
line base X=1
