#!/bin/bash
# GPU box: the GPU suite, the driver-style bench line (extras with parity + sustained windows), phase stamps of the
# configs[3] step kernel.  usage: tools/r4_round.sh <tag>
tag=${1:-r4e}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
tail -8 $out/${tag}_pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.json 2> $out/${tag}_bench.err; tail -3 $out/${tag}_bench.err
python - $out/${tag}_bench_driver.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline %.2f M  sustained %.2f M  kernel_us %.2f requeue %.2f  parity %s' % (j['value'] / 1e6, j['sustained']['value'] / 1e6, j['roofline']['kernel_us'], j['roofline']['reset_kernel_us'], j['parity']['bit_exact']))
for k, v in j['extra'].items():
  print(k, '%.2f M' % (v['value'] / 1e6), 'sustained %.2f M' % (v['sustained']['value'] / 1e6), 'kernel_us %.1f' % v['kernel_us'], 'requeue %.1f' % v['reset_kernel_us'], 'parity', v['parity']['bit_exact'], v['parity']['problems'])
PY
timeout 300 python tools/gpu_phase_means.py 8192 --area 256 --steps 700 > $out/${tag}_cfg4_phases.txt 2>&1; head -12 $out/${tag}_cfg4_phases.txt
timeout 200 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1; head -10 $out/${tag}_phases_4096.txt
