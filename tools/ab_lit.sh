#!/bin/bash
# Builds gpurun_ab/lit.so (the renderer's 79 MB table of lit sprite rows, as built by default) and gpurun_ab/nolit.so
# (-DCRAFTER_LIT_SPRITES=0: sprite rows blended and lit per frame) for a same-box A/B: tools/ab_variants.sh lit nolit
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared"
hipcc $FLAGS -DCRAFTER_LIT_SPRITES=1 -o gpurun_ab/lit.so crafter_amd/csrc/crafter_hip.hip &
hipcc $FLAGS -DCRAFTER_LIT_SPRITES=0 -o gpurun_ab/nolit.so crafter_amd/csrc/crafter_hip.hip &
wait
ls -la gpurun_ab
