#!/bin/bash
# GPU box: the round-3 evidence set on the final build.  usage: tools/r3_evidence.sh <tag>
tag=${1:-r3z}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt; tail -3 $out/${tag}_pytest_gpu.txt
bash tools/profile_round.sh $tag
timeout 120 python tools/host_overhead.py 512 > $out/${tag}_host_overhead.txt 2>&1
timeout 200 python tools/gpu_phase_means.py 4096 > $out/${tag}_phases_4096.txt 2>&1
timeout 200 python tools/gpu_phase_means.py 1024 > $out/${tag}_phases_1024.txt 2>&1
CRAFTER_SPLIT=1 timeout 200 python tools/gpu_split_phases.py 4096 > $out/${tag}_split_phases_4096.txt 2>&1
Q="--no-cpu-baseline --no-parity --no-extra --steps 1000 --warmup 200 --sustained-steps 0"
CRAFTER_SPLIT=1 timeout 120 python bench.py $Q > $out/${tag}_bench_split.json 2> /dev/null
timeout 120 python bench.py $Q --envs 1024 > $out/${tag}_bench_1024.json 2> /dev/null
PMC_SQ_GROUPS="1 3 4" timeout 400 bash tools/pmc_sq.sh > /dev/null 2>&1; cp $out/pmc_sq.txt $out/${tag}_sq_counters.txt
CRAFTER_SPLIT=1 PMC_SQ_GROUPS="1 3 4" timeout 400 bash tools/pmc_sq.sh > /dev/null 2>&1; cp $out/pmc_sq.txt $out/${tag}_sq_counters_split.txt
rm -rf $out/pmc_sq
# the launch-time law, the dispatch order on / off, homogeneous populations, the long-horizon soak
bash tools/r3_rounds.sh > /dev/null 2>&1; cp $out/rounds.txt $out/${tag}_batch_size_sweep.txt
bash tools/r3_order_ab.sh > $out/${tag}_order_ab.txt 2>&1
timeout 120 python tools/gpu_sync_population.py 4096 300 > $out/${tag}_sync_population.txt 2>&1
timeout 120 python tools/gpu_order_experiment.py 4096 > $out/${tag}_order_experiment.txt 2>&1
timeout 900 python tools/soak_parity.py 20000 4096 > $out/${tag}_soak.txt 2>&1
ls $out | grep $tag | head -40
