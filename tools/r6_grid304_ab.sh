cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for v in t_b6 g304; do
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v headline: window %.2f M  sustained %.2f M  kernel %.2f us' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 300 python tools/gpu_rollout_ab.py 4096 default 2>&1 | grep -v amdgpu | python -c "
import sys, ast
for l in sys.stdin:
  if l.startswith('{'):
    d = ast.literal_eval(l.strip()); print('$v open loop %.2f M' % d['open_loop_M']); break"
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 300 python bench.py --envs 1024 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('$v 1024 envs: value %.2f M sustained %.2f M' % (d['value'] / 1e6, d['sustained']['value'] / 1e6))"
done; done
