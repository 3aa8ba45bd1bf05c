#!/bin/bash
# GPU box: the whole GPU suite on the tree's library, then the A/B of tools/r6_far_quick.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r6_far_pytest_gpu.txt
bash tools/r6_far_quick.sh "$@"
