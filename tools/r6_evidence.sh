#!/bin/bash
# GPU box: the evidence set of round 6's final build (one gpurun call on the final sources) -- GPU suite (with the soak), kernel
# trace + HBM counters of the headline step kernel (crafter_step_early_kernel) and of the resident rollout kernel, of configs[1] /
# [3] / [4], launch gaps and the generator's share of kernel time from the trace, LDS bank-conflict counters of the generation
# kernels (VERDICT r5 #1b), phase stamps of the step kernels and of the three generation kernels (probe build), the host's share
# of the N > 1 loop body, the bench lines.  Profiles are summarised ON the box (-> gpurun_out/<tag>_profiles/) so that the bench
# lines that follow quote the traffic measured on these very sources.
# usage: tools/r6_evidence.sh <tag>     (needs gpurun_ab/probes.so: tools/ab_make.sh probes tree -DCRAFTER_PROBES)
tag=${1:-r6}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out $out/${tag}_profiles
cd $root
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --durations=6 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt; tail -4 $out/${tag}_pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-parity --no-extra --sustained-steps 0"
prof() {   # name, envs, area, render, bench args
  name=$1; envs=$2; area=$3; render=$4; shift 4
  rm -rf ${out:?}/${name}_stats $out/${name}_fetch $out/${name}_write
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_stats -- python $root/bench.py $P "$@" --steps 600 --warmup 100 --burn-in 300 --kernel-reps 50 > $out/${name}_stats.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${name}_fetch -- python $root/bench.py $P "$@" --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${name}_write -- python $root/bench.py $P "$@" --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_write.log 2>&1
  (cd $root && python tools/summarize_profile.py $name $out/${name}_stats $out/${name}_fetch $out/${name}_write $envs $area $render > $out/${name}_summary.log 2>&1)
  cp $root/profiles/${name}_kernel_stats.csv $root/profiles/${name}_hbm_traffic.json $out/${tag}_profiles/ 2> /dev/null
  if [ "$name" = "$tag" ]; then python $root/tools/trace_gaps.py $out/${name}_stats > $out/${tag}_profiles/${tag}_launch_gaps.txt 2>&1; fi
  rm -rf ${out:?}/${name}_stats $out/${name}_fetch $out/${name}_write
}
prof ${tag} 4096 64 1
prof ${tag}_cfg2 1024 64 1 --envs 1024
prof ${tag}_cfg4 8192 256 1 --envs 8192 --area 256
prof ${tag}_cfg5 16384 64 0 --envs 16384 --no-render
# the resident rollout kernel (crafter_step_n): the same three passes over the open-loop driver
for c in stats FETCH_SIZE WRITE_SIZE; do
  rm -rf ${out:?}/${tag}_ro_$c
  if [ $c = stats ]; then a="--kernel-trace --stats"; else a="--pmc $c"; fi
  timeout 400 rocprofv3 $a --output-format csv -d $out/${tag}_ro_$c -- python $root/tools/gpu_rollout_ab.py 4096 default > $out/${tag}_ro_$c.log 2>&1
done
python - $out $tag <<'PY'
import csv, sys, pathlib, collections, json
out, tag = pathlib.Path(sys.argv[1]), sys.argv[2]
res = collections.defaultdict(dict)
def short(n): return n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
  acc = collections.defaultdict(list)
  for f in (out / f'{tag}_ro_{c}').rglob('*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
      if r['Counter_Name'] == c and 'crafter' in r['Kernel_Name']:
        acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
  for k, v in acc.items():
    res[k][c + '_KiB_mean'] = sum(v) / len(v)
    res[k][c + '_launches'] = len(v)
for k, v in res.items():
  v['read_bytes_per_launch'] = 2.0 * v.get('FETCH_SIZE_KiB_mean', 0.0) * 1024
  v['write_bytes_per_launch'] = v.get('WRITE_SIZE_KiB_mean', 0.0) * 1024
  v['hbm_bytes_per_launch'] = v['read_bytes_per_launch'] + v['write_bytes_per_launch']
stats = {}
for f in (out / f'{tag}_ro_stats').rglob('*kernel_stats.csv'):
  for r in csv.DictReader(open(f)):
    if 'crafter' in r['Name']:
      stats[short(r['Name'])] = {'calls': int(r['Calls']), 'average_ns': float(r['AverageNs']), 'min_ns': float(r['MinNs']), 'max_ns': float(r['MaxNs'])}
doc = {'workload': 'tools/gpu_rollout_ab.py 4096 default: 400 closed-loop burn-in steps, then 26 BatchedEnv.rollout calls of 64 steps (launches of 16 steps x 4096 envs)',
       'note': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction); a rollout launch = 16 steps of 4096 envs: algorithmic 16 x 80.9 MB',
       'counters': res, 'kernel_stats': stats}
(out / f'{tag}_profiles' / f'{tag}_rollout_profile.json').write_text(json.dumps(doc, indent=1) + '\n')
k = next((x for x in res if 'rollout_kernel<1, 1, 1>' in x), None)
if k:
  print('rollout kernel per 16-step launch: reads %.1f MB writes %.1f MB; algorithmic 1294 MB; ratio %.2f; avg %.1f us' % (
      res[k]['read_bytes_per_launch'] / 1e6, res[k]['write_bytes_per_launch'] / 1e6, res[k]['hbm_bytes_per_launch'] / (16 * 80863232), stats.get(k, {}).get('average_ns', 0) / 1e3))
PY
rm -rf ${out:?}/${tag}_ro_stats $out/${tag}_ro_FETCH_SIZE $out/${tag}_ro_WRITE_SIZE
cd $root
# LDS counters of the generation kernels (the classification kernel's bank conflicts: 61 % of its LDS-active cycles in round 5)
sq=$out/${tag}_sq; rm -rf ${sq:?}; mkdir -p $sq
i=0
for g in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $g --output-format csv -d $sq/g$i -- python $root/bench.py --steps 200 --warmup 50 --burn-in 200 --kernel-reps 10 $P > $sq/g$i.log 2>&1)
  i=$((i+1))
done
python tools/sq_summary.py $sq > $out/${tag}_profiles/${tag}_sq_counters.txt
rm -rf ${sq:?}
CRAFTER_HIP_LIB=gpurun_ab/probes.so timeout 200 python tools/gpu_gen_probe.py 2>&1 | grep -v amdgpu > $out/${tag}_profiles/${tag}_gen_phases.txt
timeout 200 python tools/host_overhead_dist.py 512 2>&1 | grep "envs,\|scatter\|native\|from C" > $out/${tag}_profiles/${tag}_host_overhead_dist.txt
timeout 200 python tools/gpu_phase_means.py 1024 > $out/${tag}_profiles/${tag}_phases_1024.txt 2>&1
CRAFTER_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --envs 512 --steps 600 --warmup 50 --no-extra --no-cpu-baseline > $out/${tag}_profiles/${tag}_bench_exchange_1rank.json 2> $out/${tag}_bench_exchange.err
timeout 200 python tools/gpu_phase_means.py 4096 > $out/${tag}_profiles/${tag}_phases_4096.txt 2>&1
timeout 200 python tools/gpu_rollout_phases.py 4096 > $out/${tag}_profiles/${tag}_rollout_phases_4096.txt 2>&1
timeout 300 python tools/gpu_rollout_ab.py 1536,3072,4096,4608,6144,8192 default > $out/${tag}_profiles/${tag}_rollout_batch_sweep.txt 2>&1
timeout 900 python bench.py > $out/${tag}_profiles/${tag}_bench.json 2> $out/${tag}_bench.err
for i in 1 2 3; do timeout 900 python bench.py --steps 20 --warmup 5 --no-extra > $out/${tag}_profiles/${tag}_bench_driver_$i.json 2> $out/${tag}_bench_driver.err; done
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_profiles/${tag}_bench_driver.json 2>> $out/${tag}_bench_driver.err
cp $out/${tag}_pytest_gpu.txt $out/${tag}_profiles/
python - <<PY
import json
for f in ('${tag}_bench', '${tag}_bench_driver'):
  d = json.loads(open('gpurun_out/${tag}_profiles/' + f + '.json').read().strip().splitlines()[0])
  print(f, 'value %.2f M  sustained %.2f M  kernel_us %.2f frac %.3f traffic %s open_loop %.2f M' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us'], d['roofline']['frac'], d['roofline']['traffic'], d['open_loop']['value'] / 1e6), d['parity']['bit_exact'])
  for k, v in d.get('extra', {}).items():
    print('  ', k, 'value %.2f M sustained %.2f M kernel_us %.1f frac %.3f traffic %s' % (v['value'] / 1e6, v['sustained']['value'] / 1e6, v['kernel_us'], v['roofline_frac'], v['traffic']), v['parity']['bit_exact'])
PY
ls $out/${tag}_profiles
