#!/usr/bin/env python3
"""GPU box: how many step workgroups does a CU hold over a launch?  Needs a PROBE BUILD of the library through
CRAFTER_HIP_LIB: in step_body (env_kernels.hpp), right after stamp(0),
    if (prof && w.leader()) prof[14] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                       ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_ID | XCC_ID << 32
(device pass only; not part of the shipped kernel -- the step kernel is sensitive to every extra live value).
s_memtime is per XCC: intervals [start, end] are compared within one (xcc, se, sh, cu) only.
usage: CRAFTER_HIP_LIB=... tools/gpu_residency.py [envs]"""
import sys, pathlib, collections
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
res = []
for t in range(300, 400):
  if t % 10 == 9:
    torch.cuda.synchronize()
    prof.zero_()
  env.step(tape[t], info=False)
  if t % 10 == 9:
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.int64)
    hw = p[:, 14]
    xcc = (hw >> 32) & 0xF
    hwid = hw & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    key = xcc * 4096 + se * 64 + sh * 16 + cu
    ok = (p[:, 0] > 0) & (p[:, 5] > p[:, 0])
    peaks, means, counts = [], [], []
    span_by_xcc = {}
    for x in np.unique(xcc[ok]):
      m = ok & (xcc == x)
      span_by_xcc[int(x)] = (p[m, 5].max() - p[m, 0].min())
    for k in np.unique(key[ok]):
      m = ok & (key == k)
      s, e = p[m, 0], p[m, 5]
      ev = np.concatenate([np.stack([s, np.ones(len(s), np.int64)], 1), np.stack([e, -np.ones(len(s), np.int64)], 1)])
      ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
      occ = np.cumsum(ev[:, 1])
      peaks.append(occ.max())
      span = span_by_xcc[int(k // 4096)]
      means.append((e - s).sum() / span)   # time-averaged resident workgroups over the XCC's launch span
      counts.append(len(s))
    res.append((len(peaks), np.mean(peaks), np.max(peaks), np.mean(means), np.mean(counts), np.mean(list(span_by_xcc.values())) / 2050.0))
r = np.array(res)
print(f'{n} envs: distinct (xcc, se, sh, cu) {r[:, 0].mean():.0f}; workgroups per CU and launch {r[:, 4].mean():.1f}; '
      f'peak resident per CU mean {r[:, 1].mean():.2f} max {r[:, 2].max():.0f}; time-averaged resident per CU {r[:, 3].mean():.2f}; '
      f'launch span per XCC {r[:, 5].mean():.1f} us')
