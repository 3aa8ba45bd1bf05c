#!/bin/bash
# On the GPU box: SQ / instruction-cache counters of the step kernel (one rocprofv3 --pmc pass per group).
# usage: tools/pmc_sq.sh [bench args...]   ->  gpurun_out/pmc_sq.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_sq; rm -rf $out; mkdir -p $out
groups=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
        "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
        "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_CVT"
        "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM")
if [ -n "$PMC_SQ_GROUPS" ]; then   # e.g. PMC_SQ_GROUPS="1 3 4": only these groups
  sel=(); for k in $PMC_SQ_GROUPS; do sel+=("${groups[$k]}"); done; groups=("${sel[@]}")
fi
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --output-format csv -d $out/g$i -- python $root/bench.py --steps 200 --warmup 50 --burn-in 200 --kernel-reps 10 --no-cpu-baseline --no-parity --no-extra --sustained-steps 0 "$@" > $out/g$i.log 2>&1
  i=$((i+1))
done
python - $out <<'PY' | tee $root/gpurun_out/pmc_sq.txt
import csv, glob, sys, collections
import re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + '/g*/**/*counter_collection.csv', recursive=True):
  for row in csv.DictReader(open(f)):
    m = re.search(r'crafter_[a-z_]+_kernel', row['Kernel_Name'])
    if not m:
      continue
    a = acc[m.group(0)][row['Counter_Name']]
    a[0] += float(row['Counter_Value']); a[1] += 1
for kern in sorted(acc):
  print(kern)
  for k in sorted(acc[kern]):
    v = acc[kern][k]
    print(f'  {k:32s} {v[0] / v[1]:16.0f}  (per launch, {v[1]} launches)')
PY
