#!/bin/bash
# GPU box: the evidence VERDICT r2 asked for on configs[3] (8192 envs, 256x256 worlds): kernel stats, HBM counters, SQ
# counters and phase stamps of crafter_step_kernel<0, 0, 0> (and of the generation kernels beside it).
# usage: tools/r3_cfg4_profile.sh <tag>;  then locally:
#   python tools/summarize_profile.py <tag>_cfg4 gpurun_out/<tag>_cfg4_stats gpurun_out/<tag>_cfg4_fetch gpurun_out/<tag>_cfg4_write
tag=${1:-r3z}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
CFG4="--envs 8192 --area 256 --no-extra --no-cpu-baseline --no-parity --sustained-steps 0"
timeout 300 python tools/gpu_phase_means.py 8192 --area 256 --steps 700 > $out/${tag}_cfg4_phases.txt 2>&1; head -12 $out/${tag}_cfg4_phases.txt | cut -c1-230
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_cfg4_stats $out/${tag}_cfg4_fetch $out/${tag}_cfg4_write
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cfg4_stats -- python $root/bench.py $CFG4 --steps 200 --warmup 20 --burn-in 300 --kernel-reps 50 > $out/${tag}_cfg4_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_cfg4_fetch -- python $root/bench.py $CFG4 --steps 100 --warmup 20 --burn-in 200 --kernel-reps 20 > $out/${tag}_cfg4_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_cfg4_write -- python $root/bench.py $CFG4 --steps 100 --warmup 20 --burn-in 200 --kernel-reps 20 > $out/${tag}_cfg4_write.log 2>&1
find $out/${tag}_cfg4_stats $out/${tag}_cfg4_fetch $out/${tag}_cfg4_write -name '*kernel_trace.csv' -size +8M -delete
find $out/${tag}_cfg4_fetch $out/${tag}_cfg4_write -name '*counter_collection.csv' -size +30M -exec sh -c 'head -200000 "$1" > "$1.head" && mv "$1.head" "$1"' _ {} \;
cd $root
PMC_SQ_GROUPS="1 3 4 5" timeout 900 bash tools/pmc_sq.sh $CFG4 > /dev/null 2>&1; cp $out/pmc_sq.txt $out/${tag}_cfg4_sq_counters.txt; head -30 $out/${tag}_cfg4_sq_counters.txt
rm -rf $out/pmc_sq
timeout 60 python tools/host_overhead.py 64 > $out/${tag}_host_overhead_64.txt 2>&1; cat $out/${tag}_host_overhead_64.txt
