cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT}
rm -rf $root/gpurun_out/rgaps_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/rgaps_trace -- python $root/tools/gpu_rollout_ab.py 4096 default > /dev/null 2>&1
python $root/tools/probes/rollout_gaps.py $root/gpurun_out/rgaps_trace
rm -rf $root/gpurun_out/rgaps_trace
