cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT}
rm -rf $root/gpurun_out/gaps_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/gaps_trace -- python $root/bench.py --no-cpu-baseline --no-parity --no-extra --sustained-steps 0 --steps 600 --warmup 100 --burn-in 300 > /dev/null 2>&1
python $root/tools/probes/launch_gaps.py $root/gpurun_out/gaps_trace
rm -rf $root/gpurun_out/gaps_trace
