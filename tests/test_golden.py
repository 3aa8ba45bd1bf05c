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


@pytest.mark.parametrize('name', ['night_s7', 'rand_s12345', 'collect_s2'])
def test_sprite_cells_beyond_the_row_table_take_the_generic_path(name):
  """The renderer keeps blended rows for up to CRAFTER_SPRITE_ROWS sprite cells per view (16 in the product);
  a harness built with room for ONE puts every second sprite (cow next to the player, zombies at night, arrows)
  through the overflow path, by day and at night, against the reference fixtures."""
  import functools
  replay_golden(name, functools.partial(adapters.HostSimAdapter, variant=('rows1', ('CRAFTER_SPRITE_ROWS=1',))))
