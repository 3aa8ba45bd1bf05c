#!/bin/bash
# GPU box, round 5 first call: the GPU suite on the pruned build with the resident rollout, rollout occupancy A/B, the bench line.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 400 python -m pytest tests -m gpu -x -q --timeout 180 > $out/r5a_pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/r5a_pytest.log
tail -6 $out/r5a_pytest.log
timeout 300 python tools/gpu_rollout_ab.py 4096 0 300 6000 > $out/r5a_rollout_ab.txt 2>&1; cat $out/r5a_rollout_ab.txt
timeout 600 python bench.py --no-big-extra > $out/r5a_bench.json 2> $out/r5a_bench.err; tail -c 3000 $out/r5a_bench.json
