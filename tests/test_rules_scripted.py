"""CPU: kernel bodies (hostsim) vs the oracle on scripted scenarios that reach crafting, placing,
combat and sleeping -- state, reward, done and every pixel, step by step."""
import numpy as np
import pytest

from oracle.crafter_oracle import OracleEnv
from tests import scenarios
from tests.hostsim.driver import HostSimEnv
from tests.parity import assert_same


def run(kind, seed, steps, area=(64, 64)):
  acts, gifts = scenarios.SCENARIOS[kind](steps, seed)
  hs = HostSimEnv([seed], area=area, want_semantic=True)
  orc = OracleEnv(area=area, seed=seed)
  names = list(orc.t.items)
  assert np.array_equal(hs.reset()[0], orc.reset())
  reached = set()
  for t, a in enumerate(acts):
    if t in gifts:
      for item, amount in gifts[t].items():
        k = names.index(item)
        orc.inv[k] = amount
        hs.rec['inv'][0][k] = amount
    obs, rew, done = hs.step(np.array([a], np.int32))
    ob, r, d, info = orc.step(int(a))
    assert np.array_equal(obs[0], ob), f'{kind} step {t}: pixels (daylight {orc.daylight}, sleeping {orc.sleeping})'
    assert rew[0] == np.float32(r) and bool(done[0]) == bool(d), f'{kind} step {t}'
    assert_same(hs.snapshot(0), orc.snapshot(), f'{kind} step {t}')
    reached |= {n for n, c in info['achievements'].items() if c}
    if d:
      break
  return reached


def test_builder_reaches_crafting_and_placing():
  reached = set()
  for seed in (3, 4):
    reached |= run('builder', seed, 260)
  assert {'place_table', 'place_stone', 'place_plant', 'make_wood_pickaxe'} <= reached, reached
  assert reached & {'place_furnace', 'make_iron_pickaxe', 'make_stone_sword', 'make_iron_sword'}


def test_sleeper_covers_night_and_sleep_tint():
  reached = run('sleeper', 21, 330)
  assert 'wake_up' in reached


def test_fighter_combat():
  reached = set()
  for seed in (5, 6, 8):
    reached |= run('fighter', seed, 220)
  assert reached & {'defeat_zombie', 'eat_cow', 'defeat_skeleton'}, reached


@pytest.mark.parametrize('area', [(24, 36), (64, 40)])
def test_odd_areas(area):
  run('builder', 11, 120, area=area)


def test_render_disabled_still_consumes_night_noise():
  """BASELINE config 5 (render off): the night frame's 3087 doubles (engine.py:209) must still be
  drawn or the dynamics diverge from the reference -- state incl. MT19937 position vs the oracle."""
  hs = HostSimEnv([7], render_obs=False)
  orc = OracleEnv(seed=7)
  hs.reset()
  orc.reset()
  rs = np.random.RandomState(1234 + 7)
  acts = rs.choice([0, 0, 0, 6, 1, 2, 3, 4, 5], size=320)
  for t, a in enumerate(acts):
    hs.step(np.array([a], np.int32))
    _, _, d, _ = orc.step(int(a))
    if t % 10 == 0 or t > 145:
      assert_same(hs.snapshot(0), orc.snapshot(), f'step {t}')
    if d:
      break
  assert orc._step > 150, 'scenario must reach the night (steps 148-272)'


@pytest.mark.parametrize('size', [(512, 512), (100, 72), (64, 64)])
def test_render_at_other_sizes_matches_oracle(size):
  """Env.render(size) (env.py:120-130; VideoRecorder uses 512x512): unit 56 / non-square units, border
  offsets, and the night noise of shape (9*ux, 7*uy) drawn from the env RNG by every call."""
  hs = HostSimEnv([7])
  orc = OracleEnv(seed=7)
  hs.reset()
  orc.reset()
  acts = np.random.RandomState(1234 + 7).choice([0, 0, 0, 6, 1, 2, 3, 4, 5], size=200)
  for t, a in enumerate(acts):
    hs.step(np.array([a], np.int32))
    orc.step(int(a))
    if t in (3, 60, 152, 170):   # day, day, night, night
      assert np.array_equal(hs.render(size)[0], orc.render(size)), (t, size)
      assert_same(hs.snapshot(0), orc.snapshot(), f'after render at step {t}')


def test_sticky_status_bits_report_misuse():
  """Errors the reference raises as Python exceptions are sticky per-env status bits on the device
  (EnvRec.status): an action index outside the table (constants.actions[action], env.py:86) is executed as
  noop and flagged; an object table that is too small flags the overflow instead of writing out of bounds."""
  from crafter_amd import abi
  hs = HostSimEnv([3, 4], max_objects=256)
  ref = HostSimEnv([3, 4], max_objects=256)
  hs.reset()
  ref.reset()
  o1, _, _ = hs.step(np.array([99, 0], np.int32))      # env 0: invalid action
  o2, _, _ = ref.step(np.array([0, 0], np.int32))      # noop
  assert np.array_equal(o1, o2)
  assert int(hs.rec['status'][0]) & abi.ST_BAD_ACTION and int(hs.rec['status'][1]) == 0
  small = HostSimEnv([3], max_objects=6)               # worldgen places far more than 4 creatures
  small.reset()
  assert int(small.rec['status'][0]) & abi.ST_OBJ_OVERFLOW
  assert int(small.rec['nobj'][0]) <= 6
