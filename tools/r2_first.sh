#!/bin/bash
# GPU box, first call of round 2: tests, smoke, the new bench line (default + the driver's short invocation), 2-rank
# gloo plumbing run, rocprofv3 kernel stats, phase means.   usage: tools/r2_first.sh <tag>
tag=${1:-r2a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/${tag}_pytest.log
tail -5 $out/${tag}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; tail -2 $out/${tag}_smoke.log
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 1500 $out/${tag}_bench.json
timeout 300 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.json 2> $out/${tag}_bench_driver.err; tail -c 700 $out/${tag}_bench_driver.json
CRAFTER_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 200 --warmup 20 > $out/${tag}_gloo2.log 2>&1; tail -c 900 $out/${tag}_gloo2.log
timeout 300 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1; cat $out/${tag}_phases_4096.txt | head -12
timeout 300 python tools/gpu_phase_means.py 1024 > $out/${tag}_phases_1024.txt 2>&1; cat $out/${tag}_phases_1024.txt | head -12
timeout 300 python tools/gpu_phase_means.py 16384 --no-render > $out/${tag}_phases_16384nr.txt 2>&1; cat $out/${tag}_phases_16384nr.txt | head -12
timeout 400 python bench.py --envs 16384 --no-render --no-extra --steps 1000 > $out/${tag}_bench_cfg5.json 2> $out/${tag}_bench_cfg5.err; tail -c 600 $out/${tag}_bench_cfg5.json
timeout 600 python bench.py --envs 8192 --area 256 --no-extra --steps 200 --warmup 20 --burn-in 300 --kernel-reps 100 --no-cpu-baseline > $out/${tag}_bench_cfg4.json 2> $out/${tag}_bench_cfg4.err; tail -c 600 $out/${tag}_bench_cfg4.json
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $root/bench.py --steps 1500 --warmup 200 --no-cpu-baseline --no-parity --no-extra > $out/${tag}_stats.log 2>&1
find $out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs head -8
# keep the merged output small: the raw kernel trace is big
find $out/${tag}_stats -name '*kernel_trace.csv' -size +20M -delete
