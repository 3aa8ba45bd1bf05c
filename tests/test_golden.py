"""CPU: the oracle (and the kernel bodies run through tests/hostsim) against fixtures produced by
the real reference (tests/golden, tools/make_golden.py).  This is what pins the oracle on a machine
without /root/reference, e.g. the GPU box."""
import pytest

from tests import adapters
from tests.parity import golden_cases, replay_golden


@pytest.mark.parametrize('name', golden_cases())
def test_oracle_reproduces_reference_fixture(name):
  replay_golden(name, adapters.OracleAdapter)


@pytest.mark.parametrize('name', golden_cases())
def test_kernel_bodies_on_cpu_reproduce_reference_fixture(name):
  replay_golden(name, adapters.HostSimAdapter)


def test_world_pool_is_unobservable():
  """auto-reset through pre-generated (pooled) worlds == auto-reset through inline regeneration."""
  import numpy as np
  from tests.hostsim.driver import HostSimEnv
  seeds = [40 + i for i in range(5)]
  a = HostSimEnv(seeds, auto_reset=True, length=30, pool=True)
  b = HostSimEnv(seeds, auto_reset=True, length=30, pool=False)
  assert np.array_equal(a.reset(), b.reset())
  rs = np.random.RandomState(5)
  adopted = 0
  for t in range(140):
    acts = rs.randint(0, 17, size=len(seeds))
    oa, ra, da = a.step(acts)
    ob, rb, db = b.step(acts)
    assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db), t
    adopted += int(da.sum())
  for i in range(len(seeds)):
    sa, sb = a.snapshot(i), b.snapshot(i)
    assert all(np.array_equal(np.asarray(sa[k]), np.asarray(sb[k])) if hasattr(sb[k], 'shape') else sa[k] == sb[k]
               for k in sb), i
  assert adopted >= 4 * len(seeds) and (a.pool_hdr['ready'] >> 32 == 1).any()
