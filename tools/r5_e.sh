#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python tools/gpu_rollout_ab.py 4096 default CRAFTER_GEN_LAG=1 GEN_PERIOD=32,CRAFTER_GEN_LAG=1 GEN_PERIOD=32,CRAFTER_GEN_LAG=2 GEN_PERIOD=64,CRAFTER_GEN_LAG=1 GEN_PERIOD=24,CRAFTER_GEN_LAG=1 > $out/r5e_rollout_ab.txt 2>&1; cat $out/r5e_rollout_ab.txt
