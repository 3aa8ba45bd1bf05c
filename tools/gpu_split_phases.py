#!/usr/bin/env python3
"""GPU box: where the two kernels of the split step spend an env's time -- shader-clock stamps written by the rule
wave (crafter_rules_kernel) and by the frame workgroup (crafter_frame_kernel), every 25th step of a run under the
benchmark's conditions, split by what the env was doing.  Also: how the envs' start times and end times spread over
each launch (the launch lasts until its slowest env is done).
usage: tools/gpu_split_phases.py [envs] [--steps T]"""
import sys, pathlib, json
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
T = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 1400
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).cuda()
for t in range(400):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
day = env.tables.daylight
R = ['load', 'setup', 'player', 'objects', 'balance+fin', 'share/adopt', 'emit', 'store', 'RULES']
F = ['f.load', 'f.tables', 'f.pixels', 'f.tail', 'FRAME']
cats = {}
spans_r, spans_f, ends_r, ends_f = [], [], [], []
for t in range(400, T):
  sample = t % 25 == 24
  if sample:
    torch.cuda.synchronize()
    prof.zero_()
    before = env.records()
  env.step(tape[t], info=False)
  if sample:
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.int64)
    rec = env.records()
    s = rec['step'].astype(np.int64)
    adopted = rec['episode'] != before['episode']
    night = day[np.clip(s, 0, len(day) - 1)] < 0.5
    bal = (s % 10) == 0
    rules = np.stack([p[:, 1] - p[:, 0], p[:, 9] - p[:, 1], p[:, 10] - p[:, 9], p[:, 2] - p[:, 10], p[:, 3] - p[:, 2],
                      p[:, 11] - p[:, 3], p[:, 4] - p[:, 11], p[:, 5] - p[:, 4], p[:, 5] - p[:, 0]], 1)
    frame = np.stack([p[:, 15] - p[:, 14], p[:, 7] - p[:, 15], p[:, 8] - p[:, 7], p[:, 6] - p[:, 8], p[:, 6] - p[:, 14]], 1)
    ok = (p[:, 5] > 0) & (p[:, 6] > 0)
    for key, m in (('day', ~night & ~bal & ~adopted), ('night', night & ~bal & ~adopted), ('day+balance', ~night & bal & ~adopted),
                   ('night+balance', night & bal & ~adopted), ('adopted', adopted)):
      cats.setdefault(key, []).append(np.concatenate([rules, frame], 1)[ok & m])
    spans_r.append(p[ok, 5].max() - p[ok, 0].min())
    spans_f.append(p[ok, 6].max() - p[ok, 14].min())
    ends_r.append(np.percentile(p[ok, 5] - p[ok, 0].min(), [50, 90, 99, 100]))
    ends_f.append(np.percentile(p[ok, 6] - p[ok, 14].min(), [50, 90, 99, 100]))
names = R + F
tot = sum(len(x) for v in cats.values() for x in v)
print(f'{n} envs, split step; ticks = shader clocks; per env (mean); share = fraction of env-steps')
print(f'{"":14s}' + ''.join(f'{k:>12s}' for k in names) + '   share')
out = {}
for key, v in cats.items():
  a = np.concatenate(v)
  if not len(a):
    continue
  print(f'{key:14s}' + ''.join(f'{a[:, k].mean():12.0f}' for k in range(len(names))) + f'   {len(a) / tot:.3f}')
  out[key] = {nm: float(a[:, k].mean()) for k, nm in enumerate(names)}
  out[key]['share'] = len(a) / tot
  out[key]['p99_rules'] = float(np.percentile(a[:, len(R) - 1], 99))
  out[key]['p99_frame'] = float(np.percentile(a[:, -1], 99))
allp = np.concatenate([x for v in cats.values() for x in v])
print(f'{"all":14s}' + ''.join(f'{allp[:, k].mean():12.0f}' for k in range(len(names))))
print('RULES per env p50 %.0f p90 %.0f p99 %.0f max %.0f' % tuple(np.percentile(allp[:, len(R) - 1], [50, 90, 99, 100])))
print('FRAME per env p50 %.0f p90 %.0f p99 %.0f max %.0f' % tuple(np.percentile(allp[:, -1], [50, 90, 99, 100])))
print('rules launch: span (first start .. last end) mean %.0f ticks; env END times at p50 / p90 / p99 / max of the launch: %s'
      % (np.mean(spans_r), ' '.join(f'{x:.0f}' for x in np.mean(ends_r, 0))))
print('frame launch: span mean %.0f ticks; env END times at p50 / p90 / p99 / max: %s'
      % (np.mean(spans_f), ' '.join(f'{x:.0f}' for x in np.mean(ends_f, 0))))
env.enable_phase_stamps(False)
env.set_timing(True)
for t in range(300):
  env.step(tape[t], info=False)
ms, rms, k = env.get_timing()
print(f'pair_us {1000 * ms / k:.2f} (timing events, {k} launches)')
out['all'] = {nm: float(allp[:, k].mean()) for k, nm in enumerate(names)}
out['pair_us'] = 1000 * ms / k
out['span_rules'] = float(np.mean(spans_r)); out['span_frame'] = float(np.mean(spans_f))
print(json.dumps(out))
