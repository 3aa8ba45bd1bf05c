#!/bin/bash
# GPU box, first call of round 4: the GPU suite with the BASELINE-size parity tests, the bench line whose extras carry
# their own parity samples, and the instruction-cache / fetch counters of the step kernel (never collected before).
tag=${1:-r4a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --durations=8 > $out/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/${tag}_pytest_gpu.txt
tail -16 $out/${tag}_pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver.json 2> $out/${tag}_bench.err; tail -c 3000 $out/${tag}_bench_driver.json; tail -5 $out/${tag}_bench.err
PMC_SQ_GROUPS="0 2" timeout 400 bash tools/pmc_sq.sh > /dev/null 2>&1; cp $out/pmc_sq.txt $out/${tag}_sq_icache.txt; grep -A12 crafter_step_kernel $out/${tag}_sq_icache.txt
rm -rf $out/pmc_sq
