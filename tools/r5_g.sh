#!/bin/bash
# GPU box: full GPU suite (with the soak), closed loop + rollout A/B, step-kernel traffic after the state array's conditional write-back.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 --durations=8 > $out/r5g_pytest.log 2>&1; echo "pytest rc $?"; tail -14 $out/r5g_pytest.log
timeout 300 python tools/gpu_rollout_ab.py 4096 default > $out/r5g_rollout_ab.txt 2>&1; cat $out/r5g_rollout_ab.txt
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-parity --no-extra --sustained-steps 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/r5g_$c
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $out/r5g_$c -- python $root/bench.py $P --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/r5g_$c.log 2>&1
done
python - $out <<'PY'
import csv, sys, pathlib, collections
out = pathlib.Path(sys.argv[1])
res = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
  acc = collections.defaultdict(list)
  for f in (out / f'r5g_{c}').rglob('*counter_collection.csv'):
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
find $out/r5g_FETCH_SIZE $out/r5g_WRITE_SIZE -name '*.csv' -size +4M -delete
cd $root
timeout 600 python bench.py --no-big-extra --no-cpu-baseline > $out/r5g_bench.json 2> $out/r5g_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5g_bench.json').read().strip().splitlines()[-1])
print('value %.2f M  sustained %.2f M  kernel_us %.2f  open_loop %.2f M  launch-stream ms %.4f sync ms %.4f' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us'], d['open_loop']['value'] / 1e6, d['launch_stream_ms_per_step'], d['device_sync_ms_per_step']), d['parity']['bit_exact'])
PY
