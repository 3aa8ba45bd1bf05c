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
