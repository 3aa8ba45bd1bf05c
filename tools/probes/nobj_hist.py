"""GPU box: how many slots of the object table the envs of the metric workload use (the stage-in loads a blind prefix of it)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crafter_amd import BatchedEnv
n = 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(800, n)).astype(np.int32)).cuda()
for t in range(800):
  env.step(tape[t], info=False)
  if t in (100, 400, 799):
    torch.cuda.synchronize()
    nobj = env.records()['nobj'].astype(np.int64)
    print('step', t, 'nobj mean %.1f' % nobj.mean(), 'percentiles 10/50/90/99:', np.percentile(nobj, [10, 50, 90, 99]).tolist(),
          'share <= 64: %.3f  <= 96: %.3f  <= 128: %.3f' % ((nobj <= 64).mean(), (nobj <= 96).mean(), (nobj <= 128).mean()))
