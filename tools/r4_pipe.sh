#!/bin/bash
# GPU box: first run of the pipelined step kernel -- its tests, the whole GPU suite with it as the default, A/B against the
# fused kernel at several grids, phase stamps of its two halves.
tag=${1:-r4b}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -x -q --timeout 120 -k "pipelined or fused" > $out/${tag}_pytest_pipe.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_pipe.txt
tail -12 $out/${tag}_pytest_pipe.txt
if grep -q "pytest rc 0" $out/${tag}_pytest_pipe.txt; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
  tail -8 $out/${tag}_pytest_gpu.txt
fi
Q="--no-cpu-baseline --no-extra --steps 1000 --warmup 200 --sustained-steps 0 --kernel-reps 100"
for v in "0 0" "1 0" "1 1024"; do
  set -- $v
  CRAFTER_PIPE=$1 CRAFTER_PIPE_GRID=$2 timeout 200 python bench.py $Q > $out/${tag}_ab_pipe$1_grid$2.json 2> $out/${tag}_ab.err
  python - $out/${tag}_ab_pipe$1_grid$2.json "$v" <<'PY'
import json, sys
try:
  j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print('pipe/grid', sys.argv[2], 'value %.2f M' % (j['value'] / 1e6), 'ms/step %.4f' % j['ms_per_step'], 'kernel_us %.2f' % j['roofline']['kernel_us'],
        'requeue_us %.2f' % j['roofline']['reset_kernel_us'], j['roofline']['kernel'], 'parity', j['parity']['bit_exact'], j['parity']['problems'])
except Exception as e:
  print('pipe/grid', sys.argv[2], 'FAILED', e)
PY
done
timeout 300 python tools/gpu_split_phases.py 4096 > $out/${tag}_pipe_phases_4096.txt 2>&1; head -14 $out/${tag}_pipe_phases_4096.txt
CRAFTER_PIPE_STATIC=1 CRAFTER_PIPE_GRID=1024 timeout 200 python bench.py $Q > $out/${tag}_ab_static1024.json 2>> $out/${tag}_ab.err
python - $out/${tag}_ab_static1024.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('static walks, grid 1024: value %.2f M' % (j['value'] / 1e6), 'kernel_us %.2f' % j['roofline']['kernel_us'], 'parity', j['parity']['bit_exact'])
PY
