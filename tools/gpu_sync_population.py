#!/usr/bin/env python3
"""GPU box: step kernel time launch by launch over the first steps after reset(), when all envs of the batch are in the
same phase (same step counter: all by day, every tenth step all of them balance, from step ~180 all by night) -- a
homogeneous population of fast or slow envs -- against the mixed steady state.  Tests what the fixed part of
T(N) = 22 us + 12 ns x N is made of.  usage: tools/gpu_sync_population.py [envs] [steps]"""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
env = BatchedEnv(n, seed=1000, auto_reset=True, length=10000)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(steps, n)).astype(np.int32)).cuda()
env.set_timing(True)
us = []
for t in range(steps):
  env.step(tape[t], info=False)
  a, b, k = env.get_timing()
  us.append(1000 * a)
us = np.array(us)
step = np.arange(1, steps + 1)   # the env-step counter every env holds during launch t (until the first deaths)
day = step < 140
print(f'{n} envs, all in phase after reset(): step kernel us')
print('  day, no balance (steps 11..139, not multiples of 10): mean %.1f  min %.1f  max %.1f' % (
    us[day & (step % 10 != 0) & (step > 10)].mean(), us[day & (step % 10 != 0) & (step > 10)].min(), us[day & (step % 10 != 0) & (step > 10)].max()))
print('  day, every env balances (steps 20, 30, .. 130): mean %.1f' % us[day & (step % 10 == 0) & (step > 10)].mean())
if steps >= 260:
  night = (step >= 200) & (step < 260)
  print('  night (steps 200..259), no balance: mean %.1f; balance: mean %.1f' % (us[night & (step % 10 != 0)].mean(), us[night & (step % 10 == 0)].mean()))
print('  per launch, steps 1..40:', np.round(us[:40], 1).tolist())
