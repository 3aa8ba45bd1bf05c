#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests -m gpu -x -q --timeout 300 > $out/r5k_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/r5k_pytest.log | head -2
timeout 300 python tools/gpu_phase_means.py 4096 > $out/r5k_phases_4096.txt 2>&1; head -9 $out/r5k_phases_4096.txt
timeout 300 python tools/gpu_rollout_ab.py 4096 default > $out/r5k_rollout_ab.txt 2>&1; cat $out/r5k_rollout_ab.txt
timeout 600 python bench.py --no-cpu-baseline > $out/r5k_bench.json 2> $out/r5k_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5k_bench.json').read().strip().splitlines()[0])
print('value %.2f M  sustained %.2f M  kernel_us %.2f  open_loop %.2f M' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us'], d['open_loop']['value'] / 1e6), d['parity']['bit_exact'])
for k, v in d['extra'].items():
  print(k, 'value %.2f M sustained %.2f M kernel_us %.1f' % (v['value'] / 1e6, v['sustained']['value'] / 1e6, v['kernel_us']), v['kernel'], v['parity']['bit_exact'])
PY
