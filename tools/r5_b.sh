#!/bin/bash
# GPU box, round 5: resident rollout with the rule wave not waiting + dispatch order: A/B, T(N), phases; rollout tests.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 300 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_c_boundary.py -x -q --timeout 180 > $out/r5b_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $out/r5b_pytest.log
timeout 300 python tools/gpu_rollout_ab.py 4096 CRAFTER_ROLLOUT_ORDER=1 CRAFTER_ROLLOUT_ORDER=0 CRAFTER_ROLLOUT_ORDER=1,CRAFTER_ROLLOUT_LDS_PAD=300 > $out/r5b_rollout_ab.txt 2>&1; cat $out/r5b_rollout_ab.txt
timeout 300 python tools/gpu_rollout_ab.py 1536,3072,4608,6144,8192 default > $out/r5b_rollout_sweep.txt 2>&1; cat $out/r5b_rollout_sweep.txt
timeout 300 python tools/gpu_rollout_phases.py 4096 > $out/r5b_rollout_phases.txt 2>&1; head -12 $out/r5b_rollout_phases.txt
