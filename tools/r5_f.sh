#!/bin/bash
# GPU box: HBM counters of the resident rollout kernel (two --pmc passes over tools/gpu_rollout_ab.py), summarised.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/r5f_$c
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $out/r5f_$c -- python $root/tools/gpu_rollout_ab.py 4096 default > $out/r5f_$c.log 2>&1
done
python - $out <<'PY'
import csv, sys, pathlib, collections
out = pathlib.Path(sys.argv[1])
res = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
  acc = collections.defaultdict(list)
  for f in (out / f'r5f_{c}').rglob('*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
      if r['Counter_Name'] == c and 'crafter' in r['Kernel_Name']:
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        acc[name].append(float(r['Counter_Value']))
  for k, v in acc.items():
    res[k][c] = (sum(v) / len(v), len(v))
for k, v in sorted(res.items()):
  f, w = v.get('FETCH_SIZE', (0, 0)), v.get('WRITE_SIZE', (0, 0))
  print(f'{k:60s} reads {2 * f[0] * 1024 / 1e6:10.2f} MB  writes {w[0] * 1024 / 1e6:10.2f} MB  per launch ({f[1]} / {w[1]} launches)')
PY
find $out/r5f_FETCH_SIZE $out/r5f_WRITE_SIZE -name '*.csv' -size +4M -delete
