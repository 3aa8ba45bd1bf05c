#!/usr/bin/env python3
"""GPU box: how many step workgroups are resident per CU?  Start / end shader-clock stamps of every workgroup of one launch
(4096 envs back to back with the usual traffic); workgroup b runs on XCD b % 8 (32 CUs, one clock per XCD): the peak
number of overlapping [start, end] intervals per XCD / 32 = resident workgroups per CU.
usage: tools/gpu_residency.py [envs]"""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(400, n)).astype(np.int32)).cuda()
for t in range(300):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
peaks = []
for t in range(300, 400):
  if t % 10 == 9:
    torch.cuda.synchronize()
    prof.zero_()
  env.step(tape[t], info=False)
  if t % 10 == 9:
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.int64)
    per = []
    for x in range(8):
      s, e = p[x::8, 0], p[x::8, 5]
      ok = (s > 0) & (e > s)
      ev = np.concatenate([np.stack([s[ok], np.ones(ok.sum(), np.int64)], 1), np.stack([e[ok], -np.ones(ok.sum(), np.int64)], 1)])
      ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
      per.append(np.cumsum(ev[:, 1]).max())
    peaks.append(per)
peaks = np.array(peaks)
print(f'{n} envs, LDS per step workgroup {env.lds_bytes() if callable(env.lds_bytes) else env.lds_bytes} B: peak resident workgroups per XCD (8 XCDs, mean over {len(peaks)} launches):',
      np.round(peaks.mean(0), 1), '-> per CU %.2f' % (peaks.mean() / 32))
