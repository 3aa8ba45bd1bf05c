#!/usr/bin/env python3
"""GPU box: repeats the 4096-env sampled parity run and reports every mismatch with the env's episode / step at that time."""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv
from tests.parity import sha8
from tests.rollout import oracle_rollouts
n, T = 4096, 300
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sample = [0, 1, 63, 64, 511, 512, 1023, 1024, 2047, 2048, 3071, 3500, 4094, 4095] + list(range(100, 4000, 177))
tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tapes[:, i], auto_reset=True) for i in sample])
dev_tape = torch.from_numpy(tapes).cuda()
sel = torch.tensor(sample).cuda()
for rep in range(reps):
  env = BatchedEnv(n, seed=1000, auto_reset=True)
  env.reset()
  bad = []
  dead = set()
  for t in range(T):
    obs, rew, done, _ = env.step(dev_tape[t], info=False)
    o = obs[sel].cpu().numpy()
    rec = env.records()
    for k, i in enumerate(sample):
      if k in dead:
        continue
      if sha8(o[k]) != res[k]['obs_sha'][t]:
        bad.append((i, t, int(rec['episode'][i]), int(rec['step'][i]), bool(res[k]['done'][t])))
        dead.add(k)
  print(f'rep {rep}: {len(bad)} envs diverged: (env, batch step, episode, env step, done at that step) {bad[:6]}', flush=True)
  del env
