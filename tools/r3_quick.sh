#!/bin/bash
# GPU box: quick iteration set of round 3 -- GPU tests, the default bench line (split step), the fused twin, per-kernel
# times of the pair from rocprofv3, configs[4] and configs[3] short.   usage: tools/r3_quick.sh <tag> [notest]
tag=${1:-q}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
if [ "$2" != "notest" ]; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
  tail -6 $out/${tag}_pytest_gpu.txt
fi
timeout 600 python bench.py --no-big-extra > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 1800 $out/${tag}_bench.json
Q="--no-cpu-baseline --no-parity --no-extra --steps 1000 --warmup 200 --sustained-steps 0"
CRAFTER_SPLIT=0 timeout 300 python bench.py $Q > $out/${tag}_bench_fused.json 2> $out/${tag}_bench_fused.err; tail -c 700 $out/${tag}_bench_fused.json
timeout 300 python bench.py $Q --envs 16384 --no-render > $out/${tag}_bench_cfg5.json 2> $out/${tag}_bench_cfg5.err; tail -c 700 $out/${tag}_bench_cfg5.json
timeout 300 python bench.py $Q --envs 1024 > $out/${tag}_bench_1024.json 2> $out/${tag}_bench_1024.err; tail -c 700 $out/${tag}_bench_1024.json
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-extra --sustained-steps 0 --envs 8192 --area 256 --steps 200 --warmup 20 --burn-in 300 --kernel-reps 50 > $out/${tag}_bench_cfg4.json 2> $out/${tag}_bench_cfg4.err; tail -c 700 $out/${tag}_bench_cfg4.json
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $root/bench.py $Q > $out/${tag}_stats.log 2>&1
find $out/${tag}_stats -name '*kernel_trace.csv' -size +8M -delete
find $out/${tag}_stats -name '*kernel_stats.csv' | xargs head -8 | cut -c1-260
