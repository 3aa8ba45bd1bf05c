#!/usr/bin/env python3
"""GPU box: where a balance pass spends its clocks.  Needs a PROBE BUILD (crafter_amd.build.build(out=..., defines=['CRAFTER_BALANCE_PROBE']))
through CRAFTER_HIP_LIB: balance() accumulates shader clocks per part and leaves them in stamp slots 14, 15, 6, 12.
usage: CRAFTER_HIP_LIB=gpurun_ab/balance_probe.so tools/gpu_balance_probe.py [envs] [--area A]"""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
area = int(sys.argv[sys.argv.index('--area') + 1]) if '--area' in sys.argv else 64
env = BatchedEnv(n, area=(area, area), seed=1000, auto_reset=True)
env.reset()
T = 620
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).cuda()
for t in range(300):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
day = env.tables.daylight
rows = {'day': [], 'night': []}
for t in range(300, T):
  torch.cuda.synchronize(); prof.zero_()
  env.step(tape[t], info=False)
  torch.cuda.synchronize()
  p = prof.cpu().numpy().astype(np.uint64)
  s = env.records()['step'].astype(np.int64)
  bal = (s % 10 == 0) & (p[:, 12] > 0)
  if not bal.any():
    continue
  night = day[np.clip(s, 0, len(day) - 1)] < 0.5
  lo = lambda v: (v & np.uint64(0xFFFFFFFF)).astype(np.int64)
  hi = lambda v: (v >> np.uint64(32)).astype(np.int64)
  parts = np.stack([lo(p[:, 14]), hi(p[:, 14]), lo(p[:, 15]), hi(p[:, 15]), lo(p[:, 6]), hi(p[:, 6]), lo(p[:, 12]), hi(p[:, 12]), p[:, 3].astype(np.int64) - p[:, 2].astype(np.int64)], 1)
  for key, m in (('day', bal & ~night), ('night', bal & night)):
    if m.any():
      rows[key].append(parts[m])
names = ['pair flags', 'speculation', 'serial draws', 'despawn pass', 'cell search', 'apply loop', 'hits', 'spec rounds', 'balance+fin total']
print(f'{n} envs, {area}x{area}: balance steps, mean per env (clocks; hits / rounds are counts)')
for key, v in rows.items():
  if v:
    a = np.concatenate(v)
    print(f'{key:6s} ' + '  '.join(f'{nm} {a[:, k].mean():.0f}' for k, nm in enumerate(names)) + f'   ({len(a)} env-steps)')
