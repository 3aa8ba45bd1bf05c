#!/bin/bash
# build container: the balance probe build -> gpurun_ab/balance_probe.so (then on the GPU box: tools/gpu_balance_probe.py)
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/gpurun_ab
cd $root && python3 -c "
from crafter_amd import build
print(build.build(force=True, out='$root/gpurun_ab/balance_probe.so', defines=['CRAFTER_BALANCE_PROBE']))"
