"""-m gpu: the world pool under the schedules ADVICE r2 flagged as racy (found by reading, never reproduced): a pool entry
is shared by the worlds of one env with the same episode parity, so two writers of one entry must never overlap --
(1) Env.reset() in mid-run while a generation batch holding the env's world k + 2 is still in flight (the reset kernel
writes that very entry), (2) very short episodes at full batch width (inline regenerations whose follow-up request
lands in the batch right behind one that still holds the older world).  Both against the oracle, which has no pool."""
import numpy as np
import pytest
import torch

from tests.compare import compare_with_rollouts as _compare
from tests.rollout import oracle_rollouts

pytestmark = pytest.mark.gpu


def _batched(*a, **k):
  from crafter_amd import BatchedEnv
  return BatchedEnv(*a, **k)


def test_manual_reset_right_after_a_generation_batch_was_launched():
  """reset() of all envs one step / a few steps after a batch launch (period 16: batches go out with steps 16, 32, ...),
  twice; the next episodes -- adopted from the pool entries the reset kernel and the batches wrote -- must be the
  oracle's worlds (full state right after every reset and every 10 steps, frames every step)."""
  n, T, length = 128, 100, 12
  seeds = [7000 + 3 * i for i in range(n)]
  tapes = np.random.RandomState(77).choice([0, 1, 2, 3, 4, 5], size=(T, n)).astype(np.int32)
  resets = (16, 35)
  res = oracle_rollouts([dict(kwargs=dict(seed=s, length=length), actions=tapes[:, i], snapshots=range(0, T, 10), auto_reset=True,
                              reset_at=resets) for i, s in enumerate(seeds)])
  env = _batched(n, seeds=seeds, length=length, auto_reset=True)
  _compare(env, tapes, res, where='mid-run reset', reset_at=resets)
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] >= 4 * n, ps


def test_short_episodes_at_full_width_keep_the_pool_consistent():
  """Inline regeneration (envs queued for crafter_requeue_reset_kernel behind the step launch) next to adoption.
  4096 envs, length 9: every env resets every 9 steps, far faster than the pool's look-ahead of two worlds can be
  refilled (a batch every 16 steps), so adoptions and inline regenerations mix and requests of one env follow each
  other through consecutive batches.  32 envs sampled from the batch against the oracle: every frame, the full state
  every 9 steps (right after each auto-reset)."""
  n, T, length = 4096, 150, 9
  sample = sorted(set(np.random.RandomState(5).randint(0, n, size=30).tolist() + [0, n - 1]))
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i, length=length), actions=tapes[:, i], snapshots=range(8, T, 9), auto_reset=True)
                         for i in sample])
  env = _batched(n, seed=1000, length=length, auto_reset=True)
  _compare(env, tapes, res, index=sample, where='short episodes')
  ps = env.pool_status()
  assert ps['state'] == 'running', ps
  assert ps['adopted'] > 0 and ps['regenerated_inline'] > 0, f'both reset paths must have run: {ps}'
