#!/usr/bin/env python3
"""GPU box: which envs' frame workgroups time out waiting for their rule wave (overlapped split step), and when."""
import sys, pathlib, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(400, n)).astype(np.int32)).cuda()
off = env._off['status']
seen = np.zeros(n, bool)
for t in range(400):
  t0 = time.perf_counter()
  env.step(tape[t], info=False)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  st = env._rec_i32[:, off].cpu().numpy()
  bad = (st & 32) != 0
  new = bad & ~seen
  if new.any() or dt > 0.05:
    idx = np.nonzero(new)[0]
    print(f'step {t}: {dt * 1e3:.1f} ms, {new.sum()} new time-outs; mod 8 histogram {np.bincount(idx % 8, minlength=8).tolist()}; '
          f'range {idx.min() if len(idx) else -1}..{idx.max() if len(idx) else -1}; // 1280 histogram {np.bincount(idx // 1280, minlength=4).tolist()}; first {idx[:12].tolist()}')
  seen |= bad
print('total envs timed out', int(seen.sum()), 'pool', env.pool_status())
