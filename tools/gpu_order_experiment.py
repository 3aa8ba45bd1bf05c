#!/usr/bin/env python3
"""GPU box: does the ORDER in which the envs of a step launch are dispatched matter?  Needs the experimental build with
StepCtl.order / crafter_debug_set_order (workgroup b steps env order[b]) through CRAFTER_HIP_LIB.  Every step the order is
recomputed on the device (torch ops on the launch stream) from each env's next step: slow = night frame or balance step.
Prints the step kernel's own time (HIP events on the kernel) per ordering; the torch ops between the steps do not count.
usage: CRAFTER_HIP_LIB=gpurun_ab/order.so tools/gpu_order_experiment.py [envs]"""
import sys, pathlib, ctypes as C
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv, tables

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
lib, h = env._lib, env._handle
lib.crafter_debug_set_order.argtypes = [C.c_void_p, C.c_void_p]
T = 400 + 6 * 300
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).cuda()
for t in range(400):
  env.step(tape[t], info=False)
day = torch.from_numpy(tables.daylight_table(int(env.cfg.n_daylight))).cuda()
step_col = env._off['step']
ident = torch.arange(n, dtype=torch.int32, device='cuda')

def classes():
  nxt = env._rec_i32[:, step_col] + 1          # (an env that resets this step is misjudged: a few per launch)
  night = day[nxt.clamp(max=day.numel() - 1).long()] < 0.5
  bal = (nxt % 10) == 0
  return night, bal

def order_for(kind):
  if kind in ('none', 'identity'):
    return ident
  night, bal = classes()
  slow = night | bal
  cost = night.int() * 2 + bal.int()           # 3 = night + balance, 2 = night, 1 = balance, 0 = day
  idx_slow = torch.nonzero(slow).flatten()
  idx_fast = torch.nonzero(~slow).flatten()
  if kind == 'slow_first':
    return torch.cat([idx_slow[torch.argsort(-cost[idx_slow], stable=True)], idx_fast]).int()
  if kind == 'slow_last':
    return torch.cat([idx_fast, idx_slow]).int()
  if kind == 'spread':   # the slow envs evenly over the first 65 % of the positions, the last 35 % fast only
    head = int(0.65 * n)
    ns = idx_slow.numel()
    pos = torch.full((n,), -1, dtype=torch.long, device='cuda')
    if ns > 0 and ns < head:
      slots = (torch.arange(ns, device='cuda') * head // ns)
      pos[slots] = idx_slow[torch.argsort(-cost[idx_slow], stable=True)]
      free = torch.nonzero(pos < 0).flatten()
      pos[free] = idx_fast
      return pos.int()
    return ident
  raise ValueError(kind)

t = 400
for kind in ('none', 'identity', 'spread', 'slow_first', 'slow_last', 'none'):
  env.set_timing(True)
  for k in range(300):
    o = order_for(kind).contiguous()
    lib.crafter_debug_set_order(h, C.c_void_p(o.data_ptr()) if kind != 'none' else None)
    env.step(tape[t], info=False)
    t += 1
    keep = o
  a, b, launches = env.get_timing()
  env.set_timing(False)
  night, bal = classes()
  print(f'{kind:11s} step kernel {1000 * a / launches:6.2f} us  (night {float(night.float().mean()):.3f}, balance {float(bal.float().mean()):.3f} of the envs)')
lib.crafter_debug_set_order(h, None)
env.check_errors()
