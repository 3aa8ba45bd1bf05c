cd $GRAFT_REPO_ROOT
for v in "$@"; do for rep in 1 2 3; do CRAFTER_HIP_LIB=gpurun_ab/$v.so timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('$v', '20-step value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.2f' % d['roofline']['kernel_us'])"; done; done
