cd $GRAFT_REPO_ROOT
for g in 1 2 8 64; do CRAFTER_REQUEUE_GRID=$g python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('grid $g closed value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.2f' % d['roofline']['kernel_us'], 'requeue_us %.2f' % d['roofline']['reset_kernel_us'])"; done
