#!/bin/bash
# A/B builds of the library into gpurun_ab/<name>.so (loaded through CRAFTER_HIP_LIB; tools/ab_variants.sh alternates them on
# one GPU box).  usage: tools/ab_make.sh <name> [HEAD|tree] [-DMACRO ...]   (HEAD: the committed sources; tree: the working tree)
set -e
cd "$(dirname "$0")/.."
name=$1; what=${2:-tree}; shift; shift || true
defs=""; for d in "$@"; do defs="$defs '${d#-D}',"; done
root=None
if [ "$what" = HEAD ]; then
  tmp=$(mktemp -d); git archive HEAD crafter_amd/csrc include | tar -x -C "$tmp"; root="'$tmp'"
fi
python -c "
from crafter_amd import build
print(build.build(force=True, out='gpurun_ab/$name.so', defines=($defs), root=$root))"
[ -n "$tmp" ] && rm -rf "$tmp"
