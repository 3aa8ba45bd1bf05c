#!/usr/bin/env python3
"""GPU box: long-horizon parity soak.  Envs of the benchmark (seed 1000 + i, the benchmark's action tape, auto-reset
through the world pool) stepped for thousands of steps INSIDE a full-size batch, a sample of them against the CPU port:
obs hash, reward, done, inventory, achievements at every step, the full state every 100 steps.  The oracle trajectories are
computed first, one process per sampled env (tests/rollout.py).
usage: tools/soak_parity.py [--area A] [steps] [batch envs] [sampled env indices...]     (--area 256: BASELINE configs[3]'s worlds)"""
import sys, pathlib, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv
from tests.compare import compare_with_rollouts
from tests.rollout import oracle_rollouts

area = 64
if len(sys.argv) > 2 and sys.argv[1] == '--area':
  area = int(sys.argv[2])
  del sys.argv[1:3]
kw = {} if area == 64 else {'area': (area, area)}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sample = [int(a) for a in sys.argv[3:]] or sorted({0, 7, n // 2, n - 1} | set(range(100, min(n, 1000), 97)))
tape = np.random.RandomState(1234).randint(0, 17, size=(steps, n)).astype(np.int32)
t0 = time.time()
res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i, **kw), actions=tape[:, i], snapshots=range(99, steps, 100), auto_reset=True)
                       for i in sample])
t1 = time.time()
env = BatchedEnv(n, seed=1000, auto_reset=True, **kw)
compare_with_rollouts(env, tape, res, index=sample, where='soak')
print(f'soak ok ({env.step_instance}, {area}x{area} worlds): {len(sample)} envs sampled from a {n}-env batch x {steps} steps bit-exact '
      f'({sum(r["episodes"] for r in res)} episode ends, {sum(r["night_steps"] for r in res)} night frames in the sample; '
      f'pool {env.pool_status()}); oracle {t1 - t0:.0f} s, device + compare {time.time() - t1:.0f} s')
