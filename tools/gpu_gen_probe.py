"""GPU box, probe build (CRAFTER_HIP_LIB=gpurun_ab/probes.so): shader clocks per phase of the three world-generation kernels,
(a) while the step loop runs beside them (the bench workload), (b) alone on the chip (the steps stopped: what a device-wide
synchronize at the end of a timed window waits for).  usage: python tools/gpu_gen_probe.py [envs]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crafter_amd import BatchedEnv, lib  # noqa: E402

NAMES = {0: 'seed: init_genrand', 1: 'seed: first draw (twist)', 2: 'seed: simplex shuffle', 8: 'classify: head (5 look-ups + exp)',
         9: 'classify: conditional rounds', 16: 'resolve: stage-in', 17: 'resolve: player + window_open', 18: 'resolve: material draws',
         19: 'resolve: creature draws', 20: 'resolve: strip flags', 21: 'resolve: recount_space', 22: 'resolve: write-back'}


def read():
  out = (ctypes.c_uint64 * 32)()
  assert lib.load().crafter_debug_gen_probe(out) == 0
  return np.array(out[:], dtype=np.float64)


def report(tag, v):
  print(tag)
  for grp, cnt in ((range(0, 7), 7), (range(8, 14), 15), (range(16, 31), 31)):
    n = max(v[cnt], 1.0)
    tot = 0.0
    for k in grp:
      if k in NAMES and v[k]:
        print('  %-36s %9.0f clocks per item' % (NAMES[k], v[k] / n))
        tot += v[k] / n
    print('  %-36s %9.0f clocks, %d items%s' % ('= total', tot, int(v[cnt]), (', %.1f rounds per item' % (v[14] / n)) if cnt == 15 else ''))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(1200, n)).astype(np.int32)).cuda()
env.reset()
for t in range(400):
  env.step(tape[t], info=False)
read()
for t in range(400, 1040):
  env.step(tape[t], info=False)
report('beside the step loop (640 steps, %d envs):' % n, read())
# alone: every batch in flight drains with no step kernel beside it -- launch one batch and stop stepping
for rep in range(6):
  for t in range(16):
    env.step(tape[1040 + rep * 16 + t], info=False)
  torch.cuda.synchronize()
  if rep == 0:
    read()
report('batches drained with the step loop stopped right after their launch (5 batches):', read())
env.check_errors()
