#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 300 python -m pytest tests/test_gpu_rollout.py -x -q --timeout 180 > $out/r5d_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/r5d_pytest.log
timeout 300 python tools/gpu_rollout_ab.py 4096 default CRAFTER_ROLLOUT_LDS_PAD=300 > $out/r5d_rollout_ab.txt 2>&1; cat $out/r5d_rollout_ab.txt
timeout 300 python tools/gpu_rollout_phases.py 4096 > $out/r5d_rollout_phases.txt 2>&1; head -14 $out/r5d_rollout_phases.txt
