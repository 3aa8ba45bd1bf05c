cd ${GRAFT_REPO_ROOT:-.}
for k in 4 1 2 3; do echo "FAR_STAMP=$k (4: loads committed, before the first barrier; 1: behind it; 2: scan evaluated; 3: behind the second barrier)"; CRAFTER_HIP_LIB=gpurun_ab/fs$k.so timeout 300 python tools/gpu_phase_means.py 8192 --area 256 2>&1 | grep -E "^day  |^all " ; done
