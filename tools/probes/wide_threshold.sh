cd $GRAFT_REPO_ROOT
for n in 1024 1536 2048; do for wide in 0 1; do CRAFTER_STEP_WIDE=$wide python bench.py --envs $n --steps 1500 --warmup 100 --no-cpu-baseline --no-parity --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0])
print('envs $n wide $wide value %.2f M' % (d['value'] / 1e6), 'sustained %.2f M' % (d['sustained']['value'] / 1e6), 'kernel_us %.2f' % d['roofline']['kernel_us'], d['roofline'].get('kernel'))"; done; done
