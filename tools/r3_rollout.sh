#!/bin/bash
# GPU box: the rollout path -- its GPU tests, then the default bench line (which carries `open_loop`).  usage: tools/r3_rollout.sh <tag>
tag=${1:-r3r}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_noise.py -q --timeout 300 -x > $out/${tag}_pytest_rollout.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_rollout.txt
tail -25 $out/${tag}_pytest_rollout.txt | cut -c1-300
timeout 600 python bench.py --no-big-extra --no-cpu-baseline > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python - <<PY
import json
d = json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'kernel_us', d['roofline']['kernel_us'])
print('open_loop', d.get('open_loop'))
PY
tail -5 $out/${tag}_bench.err
