#!/usr/bin/env python3
"""GPU box: where an env's step goes INSIDE a rollout launch (crafter_step_n, resident state): in-kernel shader-clock
stamps of the last step of every sampled stretch, averaged over all envs, by kind of step; and the launch's own duration.
usage: tools/gpu_rollout_phases.py [envs] [VAR=v ...]"""
import os, sys, pathlib, json
for kv in sys.argv[2:]:
  if '=' in kv:
    k, v = kv.split('=')
    os.environ[k] = v
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
T, calls = 16, 40
total = 400 + T * (calls + 2)
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(total, n)).astype(np.int32)).cuda()
for t in range(400):
  env.step(tape[t], info=False)
out = (torch.empty((T,) + tuple(env.obs.shape), dtype=torch.uint8, device='cuda'), torch.empty((T, n), dtype=torch.float32, device='cuda'),
       torch.empty((T, n), dtype=torch.uint8, device='cuda'))
prof = env.enable_phase_stamps(True)
day = env.tables.daylight
names = ['load', 'setup', 'player', 'objects', 'balance+fin', 'celltab', 'tabsync', 'rows', 'pixels', 'writeout', 'store', 'TOTAL']
cats = {'day': [], 'night': [], 'day+balance': [], 'night+balance': []}
t = 400
wg = []
for c in range(calls):
  torch.cuda.synchronize()
  prof.zero_()
  env.rollout(tape[t:t + T], out=out)
  t += T
  torch.cuda.synchronize()
  p = prof.cpu().numpy().astype(np.int64)
  rec = env.records()
  s = rec['step'].astype(np.int64)
  ok = (p[:, 5] > 0) & (p[:, 6] == 0) & (s >= T)   # no adoption in the stretch's last step (and none at all in the stretch)
  night = day[np.clip(s, 0, len(day) - 1)] < 0.5
  bal = (s % 10) == 0
  ph = np.stack([p[:, 1] - p[:, 0], p[:, 9] - p[:, 1], p[:, 10] - p[:, 9], p[:, 2] - p[:, 10], p[:, 3] - p[:, 2],
                 p[:, 12] - p[:, 11], p[:, 13] - p[:, 12], p[:, 7] - p[:, 13], p[:, 8] - p[:, 7], p[:, 4] - p[:, 8], p[:, 5] - p[:, 4], p[:, 5] - p[:, 0]], 1)
  for key, m in (('day', ~night & ~bal), ('night', night & ~bal), ('day+balance', ~night & bal), ('night+balance', night & bal)):
    cats[key].append(ph[ok & m])
  wg.append((p[:, 15] - p[:, 14])[p[:, 15] > 0])
tot_n = sum(len(x) for v in cats.values() for x in v)
print(f'{n} envs, rollout of {T} steps per launch, LAST step of each stretch; ticks = shader clocks; settings {sys.argv[2:]}')
print(f'{"":14s}' + ''.join(f'{k:>12s}' for k in names) + '   share')
res = {}
for key, v in cats.items():
  a = np.concatenate(v) if v else np.zeros((0, len(names)))
  if len(a) == 0:
    continue
  print(f'{key:14s}' + ''.join(f'{a[:, k].mean():12.0f}' for k in range(len(names))) + f'   {len(a) / tot_n:.3f}')
  res[key] = {nm: float(a[:, k].mean()) for k, nm in enumerate(names)}
allp = np.concatenate([x for v in cats.values() for x in v])
print(f'{"all":14s}' + ''.join(f'{allp[:, k].mean():12.0f}' for k in range(len(names))))
print('TOTAL p50 %.0f p90 %.0f p99 %.0f max %.0f' % tuple(np.percentile(allp[:, -1], [50, 90, 99, 100])))
wga = np.concatenate(wg)
print('workgroup (16 steps) ticks: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f  -> per step %.0f (the steps\' own stamps sum to the TOTAL column: the rest is between steps)' % (
    wga.mean(), *np.percentile(wga, [50, 90, 99, 100]), wga.mean() / T))
env.enable_phase_stamps(False)
env.set_timing(True)
for c in range(20):
  env.rollout(tape[400 + c * T:400 + (c + 1) * T], out=out)
ms, rms, k = env.get_timing()
print(f'rollout kernel {1000 * ms / k:.1f} us per {T}-step launch = {1000 * ms / k / T:.2f} us per step; requeue kernel {1000 * rms / k:.1f} us ({k} launches)')
res['all'] = {nm: float(allp[:, k].mean()) for k, nm in enumerate(names)}
res['kernel_us_per_launch'] = 1000 * ms / k
print(json.dumps(res))
