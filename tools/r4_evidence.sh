#!/bin/bash
# GPU box: the round-4 evidence set on the final build.  usage: tools/r4_evidence.sh <tag>
# then locally (build container):
#   python tools/summarize_profile.py <tag> gpurun_out/<tag>_stats gpurun_out/<tag>_fetch gpurun_out/<tag>_write 4096 64 1
#   python tools/summarize_profile.py <tag>_cfg4 gpurun_out/<tag>_cfg4_stats gpurun_out/<tag>_cfg4_fetch gpurun_out/<tag>_cfg4_write 8192 256 1
#   python tools/summarize_profile.py <tag>_cfg5 gpurun_out/<tag>_cfg5_stats gpurun_out/<tag>_cfg5_fetch gpurun_out/<tag>_cfg5_write 16384 64 0
tag=${1:-r4z}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt; tail -3 $out/${tag}_pytest_gpu.txt
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.json 2> $out/${tag}_bench_driver.err
CRAFTER_PIPE=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 1000 --warmup 200 --sustained-steps 0 > $out/${tag}_bench_pipe.json 2> /dev/null
CRAFTER_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 200 --warmup 20 --no-parity --sustained-steps 100 > $out/${tag}_gloo2.log 2>&1
timeout 200 python tools/host_overhead_dist.py 512 > $out/${tag}_host_overhead_dist.txt 2>&1
timeout 200 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1
timeout 300 python tools/gpu_phase_means.py 8192 --area 256 --steps 700 > $out/${tag}_cfg4_phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-parity --no-extra --sustained-steps 0"
prof() {   # name, bench args
  name=$1; shift
  rm -rf $out/${name}_stats $out/${name}_fetch $out/${name}_write
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_stats -- python $root/bench.py $P "$@" --steps 600 --warmup 100 --burn-in 300 --kernel-reps 50 > $out/${name}_stats.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${name}_fetch -- python $root/bench.py $P "$@" --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${name}_write -- python $root/bench.py $P "$@" --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_write.log 2>&1
  find $out/${name}_stats $out/${name}_fetch $out/${name}_write -name '*kernel_trace.csv' -size +6M -delete
  find $out/${name}_fetch $out/${name}_write -name '*counter_collection.csv' -size +12M -exec sh -c 'head -80000 "$1" > "$1.head" && mv "$1.head" "$1"' _ {} \;
}
prof ${tag}
prof ${tag}_cfg4 --envs 8192 --area 256
prof ${tag}_cfg5 --envs 16384 --no-render
cd $root
PMC_SQ_GROUPS="1 3 4" timeout 400 bash tools/pmc_sq.sh > /dev/null 2>&1; cp $out/pmc_sq.txt $out/${tag}_sq_counters.txt
rm -rf $out/pmc_sq
du -sh $out | tail -1
ls $out | grep $tag | head -60
