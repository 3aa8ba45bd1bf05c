cd $GRAFT_REPO_ROOT
for v in "$@"; do CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 200 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | head -2 | python -c "
import sys, ast
for l in sys.stdin:
  d = ast.literal_eval(l.strip()); print('$v', 'open %.2f M' % d['open_loop_M'], 'us/step %.2f' % d['us_per_step'], ('closed %.2f M' % d['closed_loop_M']) if 'closed_loop_M' in d else '')"; done
