#!/bin/bash
# On the GPU box: the per-round evidence set.  usage: tools/profile_round.sh <tag>
#   gpurun_out/<tag>_bench.json                the default bench.py line (metric workload: 4096 envs)
#   gpurun_out/<tag>_bench_driver.json         the driver's invocation (--steps 20 --warmup 5)
#   gpurun_out/<tag>_bench_cfg4.json / _cfg5   BASELINE configs[3] (8192 envs, 256x256) / configs[4] (16384 envs, render off)
#   gpurun_out/<tag>_gloo2.log                 2 ranks on the one GPU, gloo exchange (plumbing of bench.py --gpus N)
#   gpurun_out/<tag>_stats / _fetch / _write   rocprofv3 outputs (kernel trace + stats; PMC passes on their own)
# then locally: python tools/summarize_profile.py <tag> gpurun_out/<tag>_stats gpurun_out/<tag>_fetch gpurun_out/<tag>_write
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 200 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.json 2> $out/${tag}_bench_driver.err
timeout 300 python bench.py --envs 16384 --no-render --no-extra --steps 1000 > $out/${tag}_bench_cfg5.json 2> $out/${tag}_bench_cfg5.err
timeout 400 python bench.py --envs 8192 --area 256 --no-extra --steps 200 --warmup 20 --burn-in 300 --kernel-reps 100 --no-cpu-baseline --sustained-steps 0 > $out/${tag}_bench_cfg4.json 2> $out/${tag}_bench_cfg4.err
CRAFTER_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 200 --warmup 20 --no-parity --sustained-steps 100 > $out/${tag}_gloo2.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $root/bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-parity --no-extra > $out/${tag}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python $root/bench.py --steps 300 --warmup 100 --burn-in 200 --kernel-reps 50 --no-cpu-baseline --no-parity --no-extra --sustained-steps 0 > $out/${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python $root/bench.py --steps 300 --warmup 100 --burn-in 200 --kernel-reps 50 --no-cpu-baseline --no-parity --no-extra --sustained-steps 0 > $out/${tag}_write.log 2>&1
find $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write -name '*kernel_trace.csv' -size +8M -delete
find $out/${tag}_fetch $out/${tag}_write -name '*counter_collection.csv' -size +30M -exec sh -c 'head -200000 "$1" > "$1.head" && mv "$1.head" "$1"' _ {} \;
tail -c 400 $out/${tag}_bench.json
