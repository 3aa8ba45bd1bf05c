"""-m gpu: the long-horizon soak, inside the suite the driver runs (the 30 k / 100 k-step soaks of
tools/soak_parity.py were builder-run text files; races that the serial CPU harness cannot show have only ever been found
by long runs under load).  Bounded to about a minute: 14 envs sampled from the metric's 4096-env batch x 3000 steps --
ten nights' worth of night frames in the sample, ~250 episode ends, every reset through the world pool --
  * closed loop: one crafter_step per step -- obs hash, reward, done, inventory, achievements every step, the full state
    every 100 steps;
  * open loop: the same 3000 steps as crafter_step_n calls of 64 (the resident rollout kernel: the state stays in LDS
    over the sixteen steps of a launch) -- every frame, reward and done of every step, the full state after every call.
The oracle trajectories (one process per sampled env) are computed once and shared."""
import numpy as np
import pytest
import torch

from tests.compare import compare_with_rollouts
from tests.rollout import oracle_rollouts

pytestmark = pytest.mark.gpu

N, STEPS, CALL = 4096, 3008, 64   # 47 calls of 64
SAMPLE = sorted({0, 7, N // 2, N - 1} | set(range(100, 1000, 97)))


@pytest.fixture(scope='module')
def soak():
  tape = np.random.RandomState(1234).randint(0, 17, size=(STEPS, N)).astype(np.int32)
  snaps = sorted(set(range(99, STEPS, 100)) | set(range(CALL - 1, STEPS, CALL)))
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tape[:, i], snapshots=snaps, auto_reset=True) for i in SAMPLE])
  assert sum(r['episodes'] for r in res) > 150 and sum(r['night_steps'] for r in res) > 4000   # resets and nights are in it
  return tape, res


def test_soak_closed_loop_3000_steps_of_the_metric_batch(soak):
  from crafter_amd import BatchedEnv
  tape, res = soak
  env = BatchedEnv(N, seed=1000, auto_reset=True)
  compare_with_rollouts(env, tape, res, index=SAMPLE, where='soak')
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] > 50000, ps


def test_soak_open_loop_3000_steps_in_resident_rollouts(soak):
  from crafter_amd import BatchedEnv
  from tests.test_gpu_rollout import _rollouts_against_oracle
  tape, res = soak
  env = BatchedEnv(N, seed=1000, auto_reset=True)
  _rollouts_against_oracle(env, tape, res, SAMPLE, [CALL] * (STEPS // CALL), 'rollout soak')
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] > 50000, ps
