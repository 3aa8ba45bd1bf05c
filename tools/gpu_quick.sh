#!/bin/bash
# GPU box: quick iteration set -- GPU tests, the default bench line, phase means at 4096 / 1024 envs.
# usage: tools/gpu_quick.sh <tag> [notest]
tag=${1:-q}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
if [ "$2" != "notest" ]; then
  timeout 300 python -m pytest tests -m gpu -x -q --timeout 120 > $out/${tag}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/${tag}_pytest.log
  tail -4 $out/${tag}_pytest.log
fi
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 2500 $out/${tag}_bench.json
timeout 300 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1; head -12 $out/${tag}_phases_4096.txt
timeout 300 python tools/gpu_phase_means.py 1024 > $out/${tag}_phases_1024.txt 2>&1; head -12 $out/${tag}_phases_1024.txt
