#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/r5_sq; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_INSTS_[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_ACTIVE[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_[A-Z_]*\|SQ_BUSY[A-Z_]*\|SQ_WAVES[A-Z_]*\|SQ_LEVEL_WAVES\|SQ_VALU_MFMA_BUSY_CYCLES\|GRBM_GUI_ACTIVE\|SQ_THREAD_CYCLES_VALU" | sort -u > $out/avail.txt
wc -l $out/avail.txt
groups=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
        "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES"
        "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES"
        "SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_DCACHE_HITS SQ_INST_LEVEL_LDS")
i=0
for g in "${groups[@]}"; do
  timeout 200 rocprofv3 --pmc $g --output-format csv -d $out/g$i -- python $root/bench.py --steps 200 --warmup 50 --burn-in 200 --kernel-reps 10 --no-cpu-baseline --no-parity --no-extra --sustained-steps 0 > $out/g$i.log 2>&1
  i=$((i+1))
done
python - $out <<'PY' | tee $root/gpurun_out/r5_sq_counters.txt
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + '/g*/**/*counter_collection.csv', recursive=True):
  for row in csv.DictReader(open(f)):
    m = re.search(r'crafter_[a-z_]+_kernel', row['Kernel_Name'])
    if not m or m.group(0) not in ('crafter_step_kernel', 'crafter_gen_classify_kernel', 'crafter_gen_resolve_kernel'):
      continue
    a = acc[m.group(0)][row['Counter_Name']]
    a[0] += float(row['Counter_Value']); a[1] += 1
for kern in sorted(acc):
  print(kern)
  for k in sorted(acc[kern]):
    v = acc[kern][k]
    print(f'  {k:32s} {v[0] / v[1]:16.0f}  (per launch, {v[1]} launches)')
PY
cat $out/avail.txt | tr '\n' ' ' | head -c 3000
rm -rf $out/g*
