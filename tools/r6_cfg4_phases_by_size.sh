cd ${GRAFT_REPO_ROOT:-.}
for n in 1536 3072 8192; do timeout 300 python tools/gpu_phase_means.py $n --area 256 2>&1 | grep -E "envs,|^day  |^all " ; done
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
for n in 1536 8192; do CRAFTER_PROBE_FREE_GEN=1 timeout 300 python tools/gpu_phase_means.py $n --area 256 2>&1 | grep -E "envs,|^day  |^all " ; done
