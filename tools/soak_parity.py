#!/usr/bin/env python3
"""GPU box: long-horizon parity soak.  A few of the benchmark's envs (seed 1000 + i, the benchmark's action tape,
auto-reset through the world pool) stepped for thousands of steps next to the CPU port; obs, reward, done at every
step and the full state every 100 steps.  usage: tools/soak_parity.py [steps] [env indices...]"""
import sys, pathlib, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv
from oracle.crafter_oracle import OracleEnv
from tests.parity import assert_same

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sample = [int(a) for a in sys.argv[2:]] or [0, 7, 512, 1023]
tape = np.random.RandomState(1234).randint(0, 17, size=(steps, 1024)).astype(np.int32)
env = BatchedEnv(len(sample), seeds=[1000 + i for i in sample], auto_reset=True)
orcs = [OracleEnv(seed=1000 + i) for i in sample]
obs = env.reset().cpu().numpy()
for k, o in enumerate(orcs):
  assert np.array_equal(obs[k], o.reset())
t0, episodes, nights = time.time(), 0, 0
for t in range(steps):
  acts = np.ascontiguousarray(tape[t, sample])
  obs, rew, done, _ = env.step(torch.from_numpy(acts).cuda(), info=False)
  obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
  for k, o in enumerate(orcs):
    ob, r, d, _ = o.step(int(acts[k]))
    nights += o.daylight < 0.5
    if d:
      episodes += 1
      ob = o.reset()
    assert np.array_equal(obs[k], ob), (t, sample[k])
    assert rew[k] == np.float32(r) and bool(done[k]) == bool(d), (t, sample[k])
  if t % 100 == 99:
    env.check_errors()
    for k, o in enumerate(orcs):
      assert_same(env.snapshot(k), o.snapshot(), f'step {t} env {sample[k]}')
print(f'soak ok: {len(sample)} envs x {steps} steps bit-exact ({episodes} episode ends, {nights} night frames), {time.time() - t0:.0f} s')
