#!/bin/bash
# Probe build for tools/gpu_simd_probe.py: the shipped sources + one store per wave at the end of step_body.
set -e
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/crafter_amd && cp -r $root/crafter_amd/csrc $tmp/crafter_amd/csrc
cp -r $root/include $tmp/include
python3 - $tmp/crafter_amd/csrc/env_kernels.hpp <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "  stamp(5);\n  return will_reset;\n}\n"
assert s.count(old) == 1
s = s.replace(old, "  stamp(5);\n  if (prof && (threadIdx.x & 63) == 0) ((uint16_t*)&prof[9])[threadIdx.x >> 6] = (uint16_t)__builtin_amdgcn_s_getreg((15 << 11) | 4);   // HW_ID[15:0]\n  return will_reset;\n}\n")
open(p, 'w').write(s)
PY
mkdir -p $root/gpurun_ab
cd $root && python3 -c "
from crafter_amd import build
print(build.build(force=True, out='$root/gpurun_ab/simd_probe.so', root='$tmp'))"
rm -rf $tmp
echo $root/gpurun_ab/simd_probe.so
