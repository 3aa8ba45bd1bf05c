#!/bin/bash
# GPU box (round 6): is the generator's cost to the step loop RESIDENCY (LDS / register slots its workgroups hold) or EXECUTION?
# probes.so: CRAFTER_PROBE_FREE_GEN=1 no generation at all; =2 sleeping workgroups with the generation kernels' footprints.
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
  print('%-34s closed %.2f M  kernel %.2f us' % ('$label', d['value'] / 1e6, d['roofline']['kernel_us']))
except Exception as e:
  print('$label', 'FAILED', e); print(open('/tmp/err.txt').read()[-800:])"
  done
  env "$@" timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | python -c "
import sys, ast
for l in sys.stdin:
  try:
    d = ast.literal_eval(l.strip()); print('%-34s open %.2f M' % ('$label', d['open_loop_M']))
  except Exception as e: print(l[:200])"
}
line base X=1
line free_gen CRAFTER_PROBE_FREE_GEN=1
line occupy_as_measured CRAFTER_PROBE_FREE_GEN=2 CRAFTER_PROBE_OCCUPY=390,136,123,361
line occupy_classify_only CRAFTER_PROBE_FREE_GEN=2 CRAFTER_PROBE_OCCUPY=390,0,123,0
line occupy_resolve_only CRAFTER_PROBE_FREE_GEN=2 CRAFTER_PROBE_OCCUPY=390,0,0,361
line occupy_seed_only CRAFTER_PROBE_FREE_GEN=2 CRAFTER_PROBE_OCCUPY=390,136,0,0
line pad5percu_free CRAFTER_PROBE_FREE_GEN=1 CRAFTER_LDS_PAD=1280 CRAFTER_ROLLOUT_LDS_PAD=1280
