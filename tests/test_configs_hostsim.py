"""CPU: the comparisons of tests/test_gpu_configs.py at reduced size on the CPU harness (kernel bodies run serially,
tests/hostsim) -- checks the test logic itself and the rule code on these configurations without a GPU."""
import numpy as np
import pytest

from tests import scenarios
from tests.compare import compare_with_rollouts
from tests.hostsim.shim import HostSimBatched
from tests.rollout import oracle_rollouts


def test_render_off_through_the_night_with_auto_reset():
  n, T = 3, 300
  seeds = [500, 501, 502]
  rs = np.random.RandomState(55)
  tapes = np.stack([rs.choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if i % 2 == 0 else rs.randint(0, 17, size=T)
                    for i in range(n)], 1).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=s), actions=tapes[:, i], snapshots=range(0, T, 25), auto_reset=True)
                         for i, s in enumerate(seeds)])
  assert max(r['night_steps'] for r in res) >= 100
  compare_with_rollouts(HostSimBatched(n, seeds=seeds, auto_reset=True, render=False), tapes, res, pixels=False)


def test_deep_256x256_world_with_pool():
  T, seeds = 240, [43, 46]
  tapes = np.stack([np.random.RandomState(900 + s).choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if s % 2 == 0 else
                    np.random.RandomState(900 + s).randint(0, 17, size=T) for s in seeds], 1).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(area=(256, 256), seed=s), actions=tapes[:, i], snapshots=(0, 150, 200), auto_reset=True)
                         for i, s in enumerate(seeds)])
  assert max(r['max_objects'] for r in res) > 1000 and max(r['night_balance_steps'] for r in res) >= 5
  compare_with_rollouts(HostSimBatched(len(seeds), area=(256, 256), seeds=seeds, auto_reset=True, pool=True), tapes, res)


def test_gifted_tapes_batched():
  T = 200
  plan = [('builder', 3), ('sleeper', 21), ('fighter', 5)]
  tapes, gifts, seeds = [], [], []
  for kind, seed in plan:
    a, g = scenarios.SCENARIOS[kind](T, seed)
    tapes.append(a), gifts.append(g), seeds.append(seed)
  tapes = np.stack(tapes, 1).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=s), actions=tapes[:, i], gifts=gifts[i], snapshots=range(0, T, 10))
                         for i, s in enumerate(seeds)])
  compare_with_rollouts(HostSimBatched(len(seeds), seeds=seeds, auto_reset=False, semantic=True), tapes, res, gifts=gifts)


def test_sampled_envs_of_a_larger_batch():
  n, T = 96, 200
  sample = [0, 1, 47, 95]
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tapes[:, i], snapshots=range(0, T, 50), auto_reset=True)
                         for i in sample])
  compare_with_rollouts(HostSimBatched(n, seed=1000, auto_reset=True, pool=True), tapes, res, index=sample)


def test_one_long_episode_past_the_lit_row_table():
  """One episode of 1200 steps (the reference's default length is 10000, env.py:27-29): beyond step 1023 the renderer
  has no pre-lit rows (render.hpp kLitSteps) and lights the rows on the fly; day, night and sleeping frames on both
  sides of that step, obs every step, the full state every 50."""
  T, seeds = 1200, [100, 124]
  plan = [scenarios.SCENARIOS['survivor'](T, s) for s in seeds]
  tapes = np.stack([a for a, _ in plan], 1).astype(np.int32)
  gifts = [g for _, g in plan]
  res = oracle_rollouts([dict(kwargs=dict(seed=s), actions=tapes[:, i], gifts=gifts[i], snapshots=range(0, T, 50))
                         for i, s in enumerate(seeds)])
  assert all(r['steps_played'] == T and r['night_steps'] >= 450 for r in res), 'the players must survive the tape'
  compare_with_rollouts(HostSimBatched(len(seeds), seeds=seeds, auto_reset=False), tapes, res, gifts=gifts, where='long')


def test_manual_reset_in_mid_run_with_the_pool():
  """Env.reset() of every env in the middle of its episode while the world pool runs (the schedule of
  tests/test_gpu_pool.py on the CPU harness, which has no concurrency: this checks the protocol -- episode numbering,
  which pool entry is adopted -- and the test helpers)."""
  n, T, length = 6, 60, 12
  seeds = [7000 + 3 * i for i in range(n)]
  tapes = np.random.RandomState(77).choice([0, 1, 2, 3, 4, 5], size=(T, n)).astype(np.int32)
  resets = (16, 35)
  res = oracle_rollouts([dict(kwargs=dict(seed=s, length=length), actions=tapes[:, i], snapshots=range(0, T, 10), auto_reset=True,
                              reset_at=resets) for i, s in enumerate(seeds)])
  compare_with_rollouts(HostSimBatched(n, seeds=seeds, length=length, auto_reset=True, pool=True), tapes, res, where='mid-run reset',
                        reset_at=resets)


@pytest.mark.parametrize('kw,stretch', [(dict(pool=True), 16), (dict(pool=False), 16), (dict(pool=True, length=7), 16),
                                        (dict(pool=True), 5), (dict(pool=True, area=(48, 40)), 16),
                                        (dict(pool=False, area=(40, 48), size=(90, 72)), 7)],
                         ids=['pool', 'requeue', 'short-episodes', 'stretch-5', 'generic-instance', 'direct-mode-frames'])
def test_rollout_equals_the_loop_of_steps(kw, stretch):
  """crafter_step_n's bodies (rollout_body; requeue_rollout_body for envs that run out of pooled worlds -- every reset
  without the pool, the second reset inside one stretch with it) against the same steps one call at a time: every
  observation, reward, done, and the final state."""
  from tests.hostsim.driver import HostSimEnv
  T, n = 230, 5
  seeds = [700 + i for i in range(n)]
  tape = np.random.RandomState(3).randint(0, 17, size=(T, n)).astype(np.int32)
  a = HostSimEnv(seeds, auto_reset=True, **kw)
  b = HostSimEnv(seeds, auto_reset=True, **kw)
  a.reset(), b.reset()
  want = [tuple(x.copy() for x in a.step(tape[t])) for t in range(T)]
  obs, reward, done = b.step_n(tape, stretch=stretch)
  for t in range(T):
    assert np.array_equal(obs[t], want[t][0]), ('obs', t)
    assert np.array_equal(reward[t], want[t][1]), ('reward', t)
    assert np.array_equal(done[t], want[t][2]), ('done', t)
  for i in range(n):   # canonical state (live objects, not the slot table's dead tail: the two runs may take a world from the
    sa, sb = a.snapshot(i), b.snapshot(i)   # pool where the other generated it inline)
    for k in sa:
      same = np.array_equal(sa[k], sb[k]) if isinstance(sa[k], np.ndarray) else sa[k] == sb[k]
      assert same, (i, k)
  assert done.sum() > (40 if kw.get('length') else 3)
