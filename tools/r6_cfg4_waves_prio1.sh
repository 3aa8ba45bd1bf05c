cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for v in t_b5 t_b6 t_b7; do
  CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 600 python bench.py --envs 8192 --area 256 --steps 600 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('$v  value %.2f M  sustained %.2f M  kernel_us %.1f' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
done; done
