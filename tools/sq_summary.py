"""LDS / issue counters of the step and generation kernels from rocprofv3 --pmc directories (tools/r6_evidence.sh)."""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + '/g*/**/*counter_collection.csv', recursive=True):
  for row in csv.DictReader(open(f)):
    m = re.search(r'crafter_[a-z_]+_kernel', row['Kernel_Name'])
    if not m or m.group(0) not in ('crafter_step_early_kernel', 'crafter_gen_classify_kernel', 'crafter_gen_resolve_kernel', 'crafter_gen_seed_kernel'):
      continue
    a = acc[m.group(0)][row['Counter_Name']]
    a[0] += float(row['Counter_Value']); a[1] += 1
for kern in sorted(acc):
  print(kern)
  c = acc[kern]
  for k in sorted(c):
    print(f'  {k:32s} {c[k][0] / c[k][1]:16.0f}  (per launch, {c[k][1]} launches)')
  if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c:
    print('  => bank-conflict cycles / LDS-active cycles = %.1f %% (classification kernel, round 5: 61 %%)' % (
        100 * (c['SQ_LDS_BANK_CONFLICT'][0] / c['SQ_LDS_BANK_CONFLICT'][1]) / max(c['SQ_LDS_IDX_ACTIVE'][0] / c['SQ_LDS_IDX_ACTIVE'][1], 1)))
