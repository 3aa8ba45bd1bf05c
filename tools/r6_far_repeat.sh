cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rollout.py -x -q -m gpu -k "config4 or 256 or generic or large or mutated" 2>&1 | tail -1; done
bash tools/r6_far_quick.sh ck2 ck2d 2>&1 | grep -E "passed|failed|==|^day |^all |cfg4"
