cd ${GRAFT_REPO_ROOT:-.}
export CRAFTER_HIP_LIB=gpurun_ab/probes.so
for i in 1 2 3; do for p in 0 1; do
  CRAFTER_GEN_SERIAL_PRIO=$p timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra --kernel-reps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline serial prio $p: window %.2f M  sustained %.2f M  kernel %.2f us' % (d['value'] / 1e6, d['sustained']['value'] / 1e6, d['roofline']['kernel_us']))"
done; done
