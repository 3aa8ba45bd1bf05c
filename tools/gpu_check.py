#!/usr/bin/env python3
"""Quick on-GPU bring-up: parity of a few envs against the oracle + rough timing.
Usage (GPU box): python tools/gpu_check.py [num_envs] [steps]"""
import json
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from crafter_amd import BatchedEnv  # noqa: E402
from oracle.crafter_oracle import OracleEnv  # noqa: E402
from tests.parity import diff_snapshots  # noqa: E402


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
  report = {'device': torch.cuda.get_device_name(0)}
  seeds = [1000 + i for i in range(n)]
  env = BatchedEnv(n, seeds=seeds, auto_reset=False, semantic=True)
  report['lds_bytes'] = env.lds_bytes
  orc = [OracleEnv(seed=s) for s in seeds]
  obs = env.reset().cpu().numpy()
  env.check_errors()
  bad = 0
  for i, o in enumerate(orc):
    want = o.reset()
    d = diff_snapshots(env.snapshot(i), o.snapshot())
    if d or not np.array_equal(obs[i], want):
      bad += 1
      print('RESET MISMATCH env', i, d[:3], int((obs[i] != want).sum()), 'pixels')
  report['reset_mismatch'] = bad
  rs = np.random.RandomState(7)
  alive = [True] * n
  first_bad = None
  for t in range(steps):
    acts = rs.randint(0, 17, size=n).astype(np.int32)
    o_, r_, d_, info = env.step(torch.from_numpy(acts).cuda())
    o_ = o_.cpu().numpy(); r_ = r_.cpu().numpy(); d_ = d_.cpu().numpy()
    sem = info['semantic'].cpu().numpy()
    for i, o in enumerate(orc):
      if not alive[i]:
        continue
      ob, rew, done, inf = o.step(int(acts[i]))
      d = diff_snapshots(env.snapshot(i), o.snapshot()) if (t % 10 == 0 or t < 5) else []
      px = int((o_[i] != ob).sum())
      ok = (not d) and px == 0 and r_[i] == np.float32(rew) and bool(d_[i]) == bool(done) and np.array_equal(sem[i], inf['semantic'])
      if not ok and first_bad is None:
        first_bad = {'step': t, 'env': i, 'diff': d[:4], 'pixels': px, 'reward': [float(r_[i]), float(rew)],
                     'done': [int(d_[i]), bool(done)], 'daylight': float(o.daylight), 'sleeping': bool(o.sleeping)}
        print('STEP MISMATCH', first_bad)
      if done:
        alive[i] = False
    if first_bad:
      break
  report['first_step_mismatch'] = first_bad
  report['parity_steps_checked'] = t + 1
  env.check_errors()

  # rough timing: 1024 envs, auto-reset, random actions resident on device
  for n2 in (1024,):
    e2 = BatchedEnv(n2, seed=1000, auto_reset=True)
    e2.reset()
    tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(400, n2)).astype(np.int32)).cuda()
    for t in range(50):
      e2.step(tape[t], info=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(50, 400):
      e2.step(tape[t], info=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e2.check_errors()
    report[f'steps_per_s_{n2}'] = 350 * n2 / dt
    report[f'ms_per_step_{n2}'] = 1000 * dt / 350
  print(json.dumps(report, indent=1))
  out = ROOT / 'gpurun_out'
  out.mkdir(exist_ok=True)
  (out / 'gpu_check.json').write_text(json.dumps(report, indent=1))


if __name__ == '__main__':
  main()
