#!/usr/bin/env python3
"""GPU box: like gpu_tail.py, but under the benchmark's conditions -- steps enqueued back to back with the
world-pool generation kernels running beside them; the stamps of every 40th step are read back."""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
T = 2400
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).cuda()
for t in range(200):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
rows, tot = [], []
for t in range(200, T):
  env.step(tape[t], info=False)
  if t % 40 == 39:
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.int64)
    total = p[:, 5] - p[:, 0]
    i = int(np.argmax(total))
    r = p[i]
    adopt = (r[6] - r[3]) if r[6] > r[3] else 0
    rows.append([r[1] - r[0], r[9] - r[1], r[10] - r[9], r[2] - r[10], r[3] - r[2], adopt, r[7] - max(r[3], r[6] if r[6] > r[3] else 0), r[8] - r[7], r[4] - r[8], r[5] - r[4], total[i]])
    tot.append([total.mean(), np.percentile(total, 50), np.percentile(total, 90), np.percentile(total, 99), total.max()])
    prof.zero_()
names = ['load', 'setup', 'player', 'objects', 'balance+fin', 'adopt', 'tables', 'noise', 'writeout', 'store', 'TOTAL']
a = np.array(rows)
print('pipelined: slowest env of sampled steps, mean (ticks):')
for k, nm in enumerate(names):
  print(f'  {nm:12s} {a[:, k].mean():9.0f}   (p90 {np.percentile(a[:, k], 90):9.0f})')
t = np.array(tot)
print('per-env total: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f' % tuple(t.mean(0)))
