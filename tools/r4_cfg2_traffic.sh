#!/bin/bash
# GPU box: counter traffic of configs[1]'s workload (1024 envs) on the final sources, summarised on the box, then the default
# bench line again so that all three extra lines quote traffic measured on these sources.  usage: tools/r4_cfg2_traffic.sh <tag>
tag=${1:-r4zz}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-parity --no-extra --sustained-steps 0"
name=${tag}_cfg2
rm -rf $out/${name}_stats $out/${name}_fetch $out/${name}_write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_stats -- python $root/bench.py $P --envs 1024 --steps 600 --warmup 100 --burn-in 300 --kernel-reps 50 > $out/${name}_stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${name}_fetch -- python $root/bench.py $P --envs 1024 --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${name}_write -- python $root/bench.py $P --envs 1024 --steps 200 --warmup 50 --burn-in 200 --kernel-reps 20 > $out/${name}_write.log 2>&1
(cd $root && python tools/summarize_profile.py $name $out/${name}_stats $out/${name}_fetch $out/${name}_write 1024 64 1 > $out/${name}_summary.log 2>&1)
mkdir -p $out/${tag}_profiles && cp $root/profiles/${name}_kernel_stats.csv $root/profiles/${name}_hbm_traffic.json $out/${tag}_profiles/ 2> /dev/null
find $out/${name}_stats $out/${name}_fetch $out/${name}_write -name '*kernel_trace.csv' -size +6M -delete
find $out/${name}_fetch $out/${name}_write -name '*counter_collection.csv' -size +12M -exec sh -c 'head -80000 "$1" > "$1.head" && mv "$1.head" "$1"' _ {} \;
cd $root
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench.json
