#!/usr/bin/env python3
"""GPU box: phase breakdown of the step kernel from in-kernel shader-clock stamps."""
import sys, pathlib, json
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(400, n)).astype(np.int32)).cuda()
for t in range(100):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
names = ['load', 'dynamics', 'balance+finish', 'render', 'store']
acc = {k: [] for k in names}
night = []
adopt = []
rt_tab, rt_quad, rt_night = [], [], []
dyn_a, dyn_b, dyn_c, nobjs = [], [], [], []
for t in range(100, 400):
  prof[:, 6] = 0
  env.step(tape[t], info=False)
  torch.cuda.synchronize()
  p = prof.cpu().numpy().astype(np.int64)
  d = np.diff(p[:, :6], axis=1)
  rt_tab.append(p[:, 7] - p[:, 3]); rt_quad.append(p[:, 8] - p[:, 7]); rt_night.append(p[:, 4] - p[:, 8])
  dyn_a.append(p[:, 9] - p[:, 1]); dyn_b.append(p[:, 10] - p[:, 9]); dyn_c.append(p[:, 2] - p[:, 10])
  m = p[:, 6] > 0
  adopt.extend((p[m, 6] - p[m, 3]).tolist())
  for i, k in enumerate(names):
    acc[k].append(d[:, i])
# reset kernel phases for the envs that have been regenerated at least once since stamping began
p = prof.cpu().numpy().astype(np.int64)
rows = p[p[:, 15] > 0]
if len(rows):
  rn = ['load+clear+mtseed', 'simplex perm', 'classify noise', 'material draws', 'creature draws', 'finalize+render', 'store']
  d = np.diff(rows[:, 8:16], axis=1)
  print('reset kernel phases (ticks), envs sampled', len(rows))
  for i, k in enumerate(rn):
    print(f'  {k:20s} mean {d[:, i].mean():10.0f}  max {d[:, i].max():10.0f}')
  print(f'  total                mean {(rows[:,15]-rows[:,8]).mean():10.0f}')
print('adopt_world ticks: n', len(adopt), 'mean', float(np.mean(adopt)) if adopt else None, 'max', max(adopt) if adopt else None)
for nm, arr in (('dyn: setup(action,step)', dyn_a), ('dyn: player_update', dyn_b), ('dyn: object loop', dyn_c), ('render: tables', rt_tab), ('render: quad pass', rt_quad), ('render: night pass', rt_night)):
  a = np.stack(arr); a = a[(a > 0) & (a < 10**7)]
  print(f'{nm:22s} mean {a.mean():9.0f} p50 {np.median(a):9.0f} p99 {np.percentile(a, 99):9.0f}')
out = {}
for k in names:
  a = np.stack(acc[k])
  out[k] = {'mean': float(a.mean()), 'p50': float(np.median(a)), 'p99': float(np.percentile(a, 99)), 'max_mean': float(a.max(axis=1).mean())}
print(json.dumps(out, indent=1))
print('units: shader clock ticks (s_memtime); max_mean = mean over steps of the slowest env')
