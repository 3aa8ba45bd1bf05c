cd ${GRAFT_REPO_ROOT:-.}
for v in fin_b5 fin_b6 fin_b7; do echo "== $v"; CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rollout.py -x -q -m gpu -k "config4 or 256" 2>&1 | tail -2; done
