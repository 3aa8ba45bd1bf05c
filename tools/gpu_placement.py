#!/usr/bin/env python3
"""GPU box: which CU does workgroup b of a step launch run on?  (probe build with the placement stamp, see gpu_residency.py)
Prints the placement of the first workgroups, how many distinct CUs the workgroups b, b + 256, b + 512, ... share, and
whether the placement repeats from launch to launch."""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv
n = 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(330, n)).astype(np.int32)).cuda()
for t in range(300): env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
keys = []
for t in range(300, 303):
  torch.cuda.synchronize(); prof.zero_()
  env.step(tape[t], info=False)
  torch.cuda.synchronize()
  p = prof.cpu().numpy().astype(np.int64)
  hw = p[:, 14]; xcc = (hw >> 32) & 0xF; hwid = hw & 0xFFFFFFFF
  cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
  keys.append(xcc * 4096 + se * 64 + sh * 16 + cu)
  if t == 300:
    print('b -> (xcc, se, sh, cu):', [(int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])) for b in range(20)])
    print('xcc of b = 0..31:', xcc[:32].tolist())
k = keys[0]
same = np.mean([len(np.unique(k[b::256])) for b in range(256)])
per_cu = np.bincount(np.unique(k, return_inverse=True)[1])
print('distinct CUs among workgroups b, b+256, ...: mean %.1f of 16' % same, '| workgroups per CU: min %d max %d' % (per_cu.min(), per_cu.max()))
print('same placement in the next launches: %.2f %.2f' % ((keys[0] == keys[1]).mean(), (keys[0] == keys[2]).mean()))
first = k[:1280]
print('first 1280 workgroups: per CU min %d max %d' % (np.bincount(np.unique(first, return_inverse=True)[1]).min(), np.bincount(np.unique(first, return_inverse=True)[1]).max()))
