#!/bin/bash
# On the GPU box: the per-round evidence set.  usage: tools/profile_round.sh <tag>
#   gpurun_out/<tag>_bench.json            the default bench.py line
#   gpurun_out/<tag>_stats / _fetch / _write   rocprofv3 outputs (kernel trace + stats; PMC passes on their own)
# then locally: python tools/summarize_profile.py <tag> gpurun_out/<tag>_stats gpurun_out/<tag>_fetch gpurun_out/<tag>_write
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
python $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $root/bench.py --steps 1500 --warmup 200 --no-cpu-baseline > $out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python $root/bench.py --steps 300 --warmup 100 --no-cpu-baseline > $out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python $root/bench.py --steps 300 --warmup 100 --no-cpu-baseline > $out/${tag}_write.log 2>&1
tail -c 600 $out/${tag}_bench.json
