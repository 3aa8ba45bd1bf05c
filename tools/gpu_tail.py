#!/usr/bin/env python3
"""GPU box: what the SLOWEST env of each batched step spends its time on (the kernel's critical path)."""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
area = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env = BatchedEnv(n, area=(area, area), seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(500, n)).astype(np.int32)).cuda()
for t in range(150):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
rows = []
tot = []
bal = []
for t in range(150, 500):
  prof.zero_()
  env.step(tape[t], info=False)
  torch.cuda.synchronize()
  p = prof.cpu().numpy().astype(np.int64)
  total = p[:, 5] - p[:, 0]
  q = p[p[:, 11] > 0]
  bal.append(np.stack([q[:, 11] - np.maximum(q[:, 3], q[:, 6]), q[:, 12] - q[:, 11], q[:, 13] - q[:, 12], q[:, 7] - q[:, 13]], 1))
  i = int(np.argmax(total))
  r = p[i]
  adopt = (r[6] - r[3]) if r[6] > 0 else 0
  rows.append([r[1] - r[0], r[9] - r[1], r[10] - r[9], r[2] - r[10], r[3] - r[2], adopt, r[7] - max(r[3], r[6]), r[8] - r[7], r[4] - r[8], r[5] - r[4], total[i]])
  tot.append([np.mean(total), np.percentile(total, 50), np.percentile(total, 90), np.percentile(total, 99), total.max(), p[:, 5].max() - p[:, 0].min()])
names = ['load', 'setup', 'player', 'objects', 'balance+fin', 'adopt', 'tables', 'noise', 'writeout', 'store', 'TOTAL']
a = np.array(rows)
print('slowest env per step, mean over steps (ticks):')
for k, nm in enumerate(names):
  print(f'  {nm:12s} {a[:, k].mean():9.0f}   (p90 {np.percentile(a[:, k], 90):9.0f})')
b = np.concatenate(bal)
print('rules-end -> render: handoff %.0f  cell table %.0f  item table + barrier %.0f  texel cache %.0f   (all envs, mean)' % tuple(b.mean(0)))
print('   p99: handoff %.0f  cell table %.0f  item table + barrier %.0f  texel cache %.0f' % tuple(np.percentile(b, 99, axis=0)))
t = np.array(tot)
print('per-env total: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f ; kernel span (first start -> last end) %.0f' % tuple(t.mean(0)))
