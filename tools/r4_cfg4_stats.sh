#!/bin/bash
tag=${1:-r4j}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CFG4="--envs 8192 --area 256 --no-extra --no-cpu-baseline --no-parity --sustained-steps 0"
rm -rf $out/${tag}_cfg4_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cfg4_stats -- python $root/bench.py $CFG4 --steps 300 --warmup 20 --burn-in 300 --kernel-reps 50 > $out/${tag}_cfg4_stats.log 2>&1
find $out/${tag}_cfg4_stats -name '*kernel_trace.csv' -size +8M -delete
f=$(find $out/${tag}_cfg4_stats -name '*kernel_stats.csv' | head -1); head -12 $f | cut -c1-200
tail -c 600 $out/${tag}_cfg4_stats.log
