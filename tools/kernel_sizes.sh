#!/bin/bash
# Code bytes per kernel / device function of the working tree's crafter_hip.hip, and (with an
# argument N) the per-helper attribution of the step kernel's bytes (tools/code_size.py).
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
out=${TMPDIR:-/tmp}/crafter_code_size; mkdir -p $out; cd $out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -gline-tables-only -c -o g.o $root/crafter_amd/csrc/crafter_hip.hip
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=g.o --output=g.co
/opt/rocm/lib/llvm/bin/llvm-readelf -sW g.co | awk '$4=="FUNC" {print $3, $8}' | sort -u | sort -n
if [ -n "$1" ]; then
  sym=$(/opt/rocm/lib/llvm/bin/llvm-readelf -sW g.co | awk '$4=="FUNC" {print $8}' | grep "${2:-crafter_step_kernel}" | head -1)
  /opt/rocm/lib/llvm/bin/llvm-objdump -d -l --no-show-raw-insn --disassemble-symbols=$sym g.co > k.s
  python $root/tools/code_size.py k.s $1
fi
