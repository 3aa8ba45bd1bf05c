#!/usr/bin/env python3
"""GPU box: does a block's time depend on what its CU neighbours (blocks b +- 256 k) are doing?
Per step: class of every env (night = spent time in the noise phase; balance step), block time by
(own class, number of night neighbours)."""
import sys, pathlib, collections
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = 1024
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(700, n)).astype(np.int32)).cuda()
for t in range(200):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
acc = collections.defaultdict(list)
for t in range(200, 700):
  env.step(tape[t], info=False)
  if t % 10 == 9:
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.int64)
    total = p[:, 5] - p[:, 0]
    night = (p[:, 8] - p[:, 7]) > 15000          # the noise phase is long only at night
    grp = night.reshape(4, 256)                   # blocks b, b+256, b+512, b+768 share a CU
    nn = grp.sum(0)[None, :].repeat(4, 0).reshape(-1) - night
    for own in (0, 1):
      for k in range(4):
        m = (night == own) & (nn == k)
        if m.any():
          acc[(own, k)].append((total[m].mean(), total[m].max(), int(m.sum())))
    prof.zero_()
print('own class, night neighbours on the CU: mean block ticks, mean of per-step max, envs per step')
for key in sorted(acc):
  a = np.array(acc[key])
  print(('night' if key[0] else 'day  '), key[1], '%8.0f %8.0f %6.1f' % (a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean()))
