#!/usr/bin/env python3
"""GPU box: does the ORDER in which the envs of a step launch are dispatched matter, and which order is best?  Every step
the order is recomputed on the device (torch ops on the launch stream) from each env's record and handed to the library
(crafter_debug_set_dispatch_order).  Prints the step kernel's own time (HIP events on the kernel) per ordering; the torch
ops between the steps do not count.  usage: tools/gpu_order_experiment.py [envs]"""
import sys, pathlib, ctypes as C
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv, tables

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
lib, h = env._lib, env._handle
kinds = ('library', 'arrival', 'slow_first', 'by_cost', 'by_cost_objects', 'spread', 'slow_last', 'library')
T = 400 + len(kinds) * 300
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).cuda()
for t in range(400):
  env.step(tape[t], info=False)
day = torch.from_numpy(tables.daylight_table(int(env.cfg.n_daylight))).cuda()
step_col, nobj_col = env._off['step'], env._off['nobj']
ident = torch.arange(n, dtype=torch.int32, device='cuda')

def classes():
  nxt = env._rec_i32[:, step_col] + 1          # (an env that resets this step is misjudged: a few per launch)
  night = day[nxt.clamp(max=day.numel() - 1).long()] < 0.5
  bal = (nxt % 10) == 0
  return night, bal

def order_for(kind):
  if kind == 'arrival':
    return ident
  night, bal = classes()
  slow = night | bal
  cost = night.int() * 26 + bal.int() * 10      # k clocks over a plain day step
  if kind == 'by_cost':
    return torch.argsort(-cost, stable=True).int()
  if kind == 'by_cost_objects':                 # + 0.2 k clocks per live object (the serial object loop)
    c = cost.float() + 0.2 * env._rec_i32[:, nobj_col].float()
    return torch.argsort(-c, stable=True).int()
  idx_slow = torch.nonzero(slow).flatten()
  idx_fast = torch.nonzero(~slow).flatten()
  if kind == 'slow_first':
    return torch.cat([idx_slow, idx_fast.flip(0)]).int()
  if kind == 'slow_last':
    return torch.cat([idx_fast, idx_slow]).int()
  if kind == 'spread':   # the slow envs evenly over the first 65 % of the positions, the last 35 % fast only
    head, ns = int(0.65 * n), idx_slow.numel()
    if 0 < ns < head:
      pos = torch.full((n,), -1, dtype=torch.long, device='cuda')
      pos[torch.arange(ns, device='cuda') * head // ns] = idx_slow
      pos[torch.nonzero(pos < 0).flatten()] = idx_fast
      return pos.int()
    return ident
  raise ValueError(kind)

t = 400
for kind in kinds:
  env.set_timing(True)
  for k in range(300):
    if kind == 'library':
      lib.crafter_debug_set_dispatch_order(h, None)
    else:
      keep = order_for(kind).contiguous()
      lib.crafter_debug_set_dispatch_order(h, C.c_void_p(keep.data_ptr()))
    env.step(tape[t], info=False)
    t += 1
  a, b, launches = env.get_timing()
  env.set_timing(False)
  night, bal = classes()
  print(f'{kind:16s} step kernel {1000 * a / launches:6.2f} us  (night {float(night.float().mean()):.3f}, balance {float(bal.float().mean()):.3f} of the envs)')
lib.crafter_debug_set_dispatch_order(h, None)
env.check_errors()
