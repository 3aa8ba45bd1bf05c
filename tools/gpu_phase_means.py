#!/usr/bin/env python3
"""GPU box: where an env's step goes, averaged over ALL envs of a batch under the benchmark's conditions (steps
enqueued back to back, world-pool generation beside them), split by what the env was doing (day / night frame,
balance step or not).  In-kernel shader-clock stamps of every 25th step are read back.
usage: tools/gpu_phase_means.py [envs] [--no-render] [--area A] [--steps T]"""
import sys, pathlib, json
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
render = '--no-render' not in sys.argv
area = int(sys.argv[sys.argv.index('--area') + 1]) if '--area' in sys.argv else 64
env = BatchedEnv(n, area=(area, area), seed=1000, auto_reset=True, render=render)
env.reset()
T = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 1400
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).cuda()
for t in range(400):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
day = env.tables.daylight
names = ['load', 'setup', 'player', 'objects', 'balance+fin', 'celltab', 'tabsync', 'rows', 'pixels', 'writeout', 'store', 'TOTAL']
cats = {'day': [], 'night': [], 'day+balance': [], 'night+balance': []}
for t in range(400, T):
  if t % 25 == 24:
    torch.cuda.synchronize()
    prof.zero_()
    step_before = env.records()['step'].astype(np.int64)
  env.step(tape[t], info=False)
  if t % 25 == 24:
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.int64)
    rec = env.records()
    ok = (p[:, 5] > 0) & (p[:, 6] == 0) & (rec['step'] == step_before + 1)   # no adopt / reset this step
    s = rec['step'].astype(np.int64)
    night = day[np.clip(s, 0, len(day) - 1)] < 0.5
    bal = (s % 10) == 0
    ph = np.stack([p[:, 1] - p[:, 0], p[:, 9] - p[:, 1], p[:, 10] - p[:, 9], p[:, 2] - p[:, 10], p[:, 3] - p[:, 2],
                   p[:, 12] - p[:, 11], p[:, 13] - p[:, 12], p[:, 7] - p[:, 13], p[:, 8] - p[:, 7], p[:, 4] - p[:, 8], p[:, 5] - p[:, 4], p[:, 5] - p[:, 0]], 1)
    for key, m in (('day', ~night & ~bal), ('night', night & ~bal), ('day+balance', ~night & bal), ('night+balance', night & bal)):
      cats[key].append(ph[ok & m])
out = {}
tot_n = sum(len(x) for v in cats.values() for x in v)
kname = 'crafter_step_early_kernel' if (env.step_instance.endswith('<1, 1, 1>') and render and n >= 2048) else env.step_instance
print(f'{n} envs, {area}x{area} world, {kname}, render {"on" if render else "off"}; ticks = shader clocks; phases per env (mean), share = fraction of env-steps')
print(f'{"":14s}' + ''.join(f'{k:>12s}' for k in names) + '   share')
for key, v in cats.items():
  a = np.concatenate(v) if v else np.zeros((0, len(names)))
  if len(a) == 0:
    continue
  print(f'{key:14s}' + ''.join(f'{a[:, k].mean():12.0f}' for k in range(len(names))) + f'   {len(a) / tot_n:.3f}')
  out[key] = {nm: float(a[:, k].mean()) for k, nm in enumerate(names)}
  out[key]['share'] = len(a) / tot_n
  out[key]['p99_total'] = float(np.percentile(a[:, -1], 99))
allp = np.concatenate([x for v in cats.values() for x in v])
print(f'{"all":14s}' + ''.join(f'{allp[:, k].mean():12.0f}' for k in range(len(names))))
print('TOTAL p50 %.0f p90 %.0f p99 %.0f max %.0f' % tuple(np.percentile(allp[:, -1], [50, 90, 99, 100])))
# (No launch-wide span / start offsets: the stamps are s_memtime values, and every XCD counts from a base of its own -- only
# differences between stamps of ONE workgroup mean anything.  The round-4 output printed such spans; they were garbage.)
env.set_timing(True)
for t in range(300):
  env.step(tape[t], info=False)
ms, rms, k = env.get_timing()
print(f'kernel_us {1000 * ms / k:.2f} (timing events, {k} launches)')
out['all'] = {nm: float(allp[:, k].mean()) for k, nm in enumerate(names)}
out['kernel_us'] = 1000 * ms / k
print(json.dumps(out))
