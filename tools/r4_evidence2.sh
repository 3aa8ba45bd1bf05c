#!/bin/bash
# GPU box: the evidence set on the round's last build (regeneration beside the step launch), leaner than r4_evidence.sh:
# GPU tests, the two bench lines, host overhead of the exchange loop, phase stamps, kernel trace + HBM counters for the
# default instance, configs[3] and configs[4], summarised on the box (-> gpurun_out/<tag>_profiles/) so that the bench lines
# that follow quote the traffic measured on these very sources.  usage: tools/r4_evidence2.sh <tag>
tag=${1:-r4zz}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt; tail -3 $out/${tag}_pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-parity --no-extra --sustained-steps 0"
# (rocprofv3 --pmc serialises kernels ACROSS streams: with CRAFTER_REGEN_BESIDE=1 a counter pass runs into the regeneration
# server's wait at every step, profiles/r4zz_pmc_mode.txt.  The default -- the kernel behind the launch -- is what is profiled.)
PMC_BESIDE=0
prof() {   # name, envs, area, render, bench args
  name=$1; envs=$2; area=$3; render=$4; shift 4
  rm -rf $out/${name}_stats $out/${name}_fetch $out/${name}_write
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_stats -- python $root/bench.py $P "$@" --steps 600 --warmup 100 --burn-in 300 --kernel-reps 50 > $out/${name}_stats.log 2>&1
  CRAFTER_REGEN_BESIDE=$PMC_BESIDE timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${name}_fetch -- python $root/bench.py $P "$@" --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_fetch.log 2>&1
  CRAFTER_REGEN_BESIDE=$PMC_BESIDE timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${name}_write -- python $root/bench.py $P "$@" --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_write.log 2>&1
  # summarised HERE, from the whole files, into this copy's profiles/ -- the bench lines below quote the traffic of exactly these sources
  (cd $root && python tools/summarize_profile.py $name $out/${name}_stats $out/${name}_fetch $out/${name}_write $envs $area $render > $out/${name}_summary.log 2>&1)
  mkdir -p $out/${tag}_profiles && cp $root/profiles/${name}_kernel_stats.csv $root/profiles/${name}_hbm_traffic.json $out/${tag}_profiles/ 2> /dev/null
  find $out/${name}_stats $out/${name}_fetch $out/${name}_write -name '*kernel_trace.csv' -size +6M -delete
  find $out/${name}_fetch $out/${name}_write -name '*counter_collection.csv' -size +12M -exec sh -c 'head -80000 "$1" > "$1.head" && mv "$1.head" "$1"' _ {} \;
}
prof ${tag} 4096 64 1
prof ${tag}_cfg4 8192 256 1 --envs 8192 --area 256
prof ${tag}_cfg5 16384 64 0 --envs 16384 --no-render
cd $root
timeout 200 python tools/host_overhead_dist.py 512 > $out/${tag}_host_overhead_dist.txt 2>&1
timeout 200 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1
for w in 0 1; do   # 512 envs -- one GPU's share of configs[2]: the 256-thread and the 512-thread step kernel
  CRAFTER_STEP_WIDE=$w timeout 300 python bench.py --envs 512 --no-cpu-baseline --no-extra --steps 1500 --warmup 300 > $out/${tag}_bench_512_wide$w.json 2> /dev/null
done
for n in 4096 1024 512; do   # the opt-in form, one handle per process: regeneration beside the launch
  CRAFTER_REGEN_BESIDE=1 timeout 300 python bench.py --envs $n --no-cpu-baseline --no-extra --steps 1500 --warmup 300 > $out/${tag}_bench_beside_$n.json 2> /dev/null
  timeout 300 python bench.py --envs $n --no-cpu-baseline --no-extra --no-parity --steps 1500 --warmup 300 --sustained-steps 0 > $out/${tag}_bench_behind_$n.json 2> /dev/null
done
cd /tmp
prof ${tag}_cfg2 1024 64 1 --envs 1024
cd $root
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.json 2> $out/${tag}_bench_driver.err
du -sh $out | tail -1
ls $out | grep $tag | head -60
