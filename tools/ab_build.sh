#!/bin/bash
# Builds gpurun_ab/old.so from HEAD and gpurun_ab/new.so from the working tree (A/B kernel timing
# on one GPU box: CRAFTER_HIP_LIB=gpurun_ab/old.so python bench.py ...).
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared"
tmp=$(mktemp -d)
git archive HEAD crafter_amd/csrc include | tar -x -C "$tmp"
hipcc $FLAGS -o gpurun_ab/old.so "$tmp/crafter_amd/csrc/crafter_hip.hip" &
hipcc $FLAGS -o gpurun_ab/new.so crafter_amd/csrc/crafter_hip.hip &
wait
rm -rf "$tmp"
ls -la gpurun_ab
