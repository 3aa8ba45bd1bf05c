"""BASELINE config 4 shape (256x256 world): the maps no longer fit LDS and stay in HBM; 128x128 still
stages them.  Kernel bodies on the CPU vs the oracle: reset, steps, auto-reset through the world pool."""
import numpy as np
import pytest

from oracle.crafter_oracle import OracleEnv
from tests.hostsim.driver import HostSimEnv
from tests.parity import assert_same


@pytest.mark.parametrize('side,steps,length', [(128, 45, 20), (256, 26, 11)])
def test_large_world_parity(side, steps, length):
  area = (side, side)
  hs = HostSimEnv([11], area=area, auto_reset=True, length=length, pool=True)
  orc = OracleEnv(area=area, seed=11, length=length)
  assert np.array_equal(hs.reset()[0], orc.reset())
  assert_same(hs.snapshot(0), orc.snapshot(), 'reset')
  rs = np.random.RandomState(0)
  for t in range(steps):
    a = int(rs.randint(0, 17))
    obs, rew, done = hs.step(np.array([a], np.int32))
    ob, r, d, _ = orc.step(a)
    if d:
      ob = orc.reset()
    assert np.array_equal(obs[0], ob), t
    assert rew[0] == np.float32(r) and bool(done[0]) == bool(d)
    if t % 6 == 0 or d:
      assert_same(hs.snapshot(0), orc.snapshot(), f'step {t}')


# The step kernel of worlds whose maps stay in HBM keeps its slot table in global memory too (env_core.hpp FarSlot): the
# object loop runs off a scan's near bits and record cache, holes are squeezed out lazily.
TINY = ('fartiny', ('CRAFTER_FAR_CACHE=3', 'CRAFTER_FAR_HOLES=2'))   # nearly every record misses the cache; the table is squeezed all the time


@pytest.mark.parametrize('variant,kw', [(None, {}), (TINY, {}), (None, dict(max_objects=1200))], ids=['product-constants', 'tiny-cache', 'small-table'])
def test_far_slot_table_through_a_night(variant, kw):
  """240 steps of two 256x256 worlds (one mostly fighting / moving, one random) with auto-reset through the pool: more than
  1000 objects, night balance passes (spawns and despawns far from the player), arrows, removals -- state every 40 steps,
  observation, reward and done every step."""
  from tests.compare import compare_with_rollouts
  from tests.hostsim.shim import HostSimBatched
  from tests.rollout import oracle_rollouts
  T, seeds = 240, [43, 46]
  tapes = np.stack([np.random.RandomState(900 + s).choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if s % 2 == 0 else
                    np.random.RandomState(900 + s).randint(0, 17, size=T) for s in seeds], 1).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(area=(256, 256), seed=s), actions=tapes[:, i], snapshots=range(0, T, 40), auto_reset=True)
                         for i, s in enumerate(seeds)])
  assert max(r['max_objects'] for r in res) > 1000 and max(r['night_balance_steps'] for r in res) >= 5
  # (small-table: 1200 slots for worlds of up to ~1100 objects -- the table is squeezed as soon as it has a hole, the policy's second
  # branch: a table that is about to look three quarters full to the host)
  compare_with_rollouts(HostSimBatched(len(seeds), area=(256, 256), seeds=seeds, auto_reset=True, pool=True, variant=variant, **kw), tapes, res)


@pytest.mark.parametrize('variant', [None, TINY], ids=['product-constants', 'tiny-cache'])
def test_far_slot_table_resident_rollout(variant):
  """crafter_step_n on 256x256 worlds (rollout_body over big_layout: the scan runs again at the head of every resident
  step) against the same steps one call at a time, short episodes so that worlds are adopted inside a stretch."""
  T, seeds = 70, [11, 12]
  kw = dict(area=(256, 256), auto_reset=True, length=23, pool=True, variant=variant)
  tape = np.random.RandomState(5).randint(0, 17, size=(T, len(seeds))).astype(np.int32)
  a, b = HostSimEnv(seeds, **kw), HostSimEnv(seeds, **kw)
  a.reset(), b.reset()
  want = [tuple(x.copy() for x in a.step(tape[t])) for t in range(T)]
  obs, reward, done = b.step_n(tape, stretch=16)
  for t in range(T):
    assert np.array_equal(obs[t], want[t][0]), ('obs', t)
    assert np.array_equal(reward[t], want[t][1]) and np.array_equal(done[t], want[t][2]), t
  for i in range(len(seeds)):
    assert_same(b.snapshot(i), a.snapshot(i), f'env {i}')
  assert done.sum() >= 4
