"""GPU box: open-loop rollout (BatchedEnv.rollout / crafter_step_n) throughput of the metric workload under LDS paddings
of the resident rollout kernel (= workgroups per CU), next to the closed loop on the same box.
usage: python tools/gpu_rollout_ab.py <envs>[,<envs>...] [VAR=v,VAR=v ...]   (each argument one variant: environment settings)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crafter_amd import BatchedEnv  # noqa: E402


KNOBS = ("CRAFTER_FOLD_MAIN_EVENT", "CRAFTER_GEN_SERIAL_PRIO", 'CRAFTER_ROLLOUT_LDS_PAD', 'CRAFTER_ROLLOUT_ORDER', 'CRAFTER_ROLLOUT_GROUPS', 'CRAFTER_NOISE_AHEAD', 'CRAFTER_GEN_LAG', 'GEN_PERIOD', 'AREA')


def run(n, variant, T=64, calls=24, burn=400, closed=0):
  for k in KNOBS:
    os.environ.pop(k, None)
  for kv in variant.split(','):
    if '=' in kv:
      k, v = kv.split('=')
      os.environ[k] = v
  pad = variant
  area = int(os.environ.get('AREA', '64'))   # (AREA=256: BASELINE configs[3]'s worlds, crafter_rollout_kernel<0, 2, 1>)
  env = BatchedEnv(n, seed=1000, auto_reset=True, gen_period=int(os.environ.get('GEN_PERIOD', '0')), **({} if area == 64 else {'area': (area, area)}))
  total = burn + (calls + 2) * T + closed
  tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(total, n)).astype(np.int32)).cuda()
  env.reset()
  for t in range(burn):
    env.step(tape[t], info=False)
  out = (torch.empty((T,) + tuple(env.obs.shape), dtype=torch.uint8, device='cuda'), torch.empty((T, n), dtype=torch.float32, device='cuda'),
         torch.empty((T, n), dtype=torch.uint8, device='cuda'))
  t = burn
  for _ in range(2):
    env.rollout(tape[t:t + T], out=out)
    t += T
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(calls):
    env.rollout(tape[t:t + T], out=out)
    t += T
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  res = {'envs': n, 'variant': pad, 'open_loop_M': calls * T * n / dt / 1e6, 'us_per_step': 1e6 * dt / (calls * T)}
  if closed:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(closed):
      env.step(tape[t + k], info=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res['closed_loop_M'] = closed * n / dt / 1e6
  env.check_errors()
  res['pool'] = env.pool_status()
  return res


if __name__ == '__main__':
  ns = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [4096]
  variants = sys.argv[2:] or ['default']
  for rep in range(2):
    for n in ns:
      for v in variants:
        print(run(n, v, closed=1000 if rep == 0 and v == variants[0] else 0), flush=True)
