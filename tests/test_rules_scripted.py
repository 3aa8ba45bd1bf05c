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
