#!/bin/bash
# GPU box (round 6): which resource of a resident generation workgroup displaces step workgroups -- LDS or registers?
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
line() {
  label=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 --sustained-steps 0 2> /tmp/err.txt | python -c "
import sys, json
try:
  d = json.loads(sys.stdin.read().strip().splitlines()[-1])
  print('%-44s closed %.2f M  kernel %.2f us' % ('$label', d['value'] / 1e6, d['roofline']['kernel_us']))
except Exception as e:
  print('$label', 'FAILED', e); print(open('/tmp/err.txt').read()[-800:])"
  done
}
O="CRAFTER_PROBE_FREE_GEN=2 CRAFTER_PROBE_OCCUPY=390,136,123,361"
line free_gen CRAFTER_PROBE_FREE_GEN=1
line occupy_as_is_smallregs $O
line occupy_as_is_bigregs $O CRAFTER_PROBE_OCCUPY_BIG=1
line occupy_lds_in_hole_smallregs $O CRAFTER_PROBE_OCCUPY_LDS=1024,2048,2048
line occupy_lds_in_hole_bigregs $O CRAFTER_PROBE_OCCUPY_LDS=1024,2048,2048 CRAFTER_PROBE_OCCUPY_BIG=1
line occupy_lds_zero_bigregs $O CRAFTER_PROBE_OCCUPY_LDS=0,0,0 CRAFTER_PROBE_OCCUPY_BIG=1
