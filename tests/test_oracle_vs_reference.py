"""Build-container only (skipped where /root/reference is absent, e.g. the GPU box): the oracle
restatement against the UNTOUCHED reference imported through oracle/reference_harness.py.
This is the primary pin of oracle/crafter_oracle.py; tests/golden carries its results elsewhere."""
import numpy as np
import pytest

from oracle import reference_harness as rh
from oracle.crafter_oracle import OracleEnv
from tests import scenarios

pytestmark = pytest.mark.skipif(not rh.available(), reason='reference tree not mounted')


def ref_objects(env):
  from crafter import objects as robj
  types = {robj.Player: 1, robj.Cow: 2, robj.Zombie: 3, robj.Skeleton: 4, robj.Arrow: 5, robj.Plant: 6}
  out = []
  for o in env._world._objects:
    if o is None:
      continue
    t = types[type(o)]
    f = tuple(int(v) for v in getattr(o, 'facing', (0, 0)))
    aux = {3: getattr(o, 'cooldown', 0), 4: getattr(o, 'reload', 0), 6: getattr(o, 'grown', 0)}.get(t, 0)
    out.append((t, int(o.pos[0]), int(o.pos[1]), int(o.health), f[0], f[1], int(aux)))
  return out


def compare_step(r, o, ra, oa, where):
  assert np.array_equal(ra[0], oa[0]), f'{where}: obs'
  assert ra[1] == oa[1] and type(ra[1]) is type(oa[1]), f'{where}: reward {ra[1]!r} {oa[1]!r}'
  assert (ra[2] is None and not oa[2]) or bool(ra[2]) == bool(oa[2]), f'{where}: done'
  ri, oi = ra[3], oa[3]
  assert ri['inventory'] == oi['inventory'] and ri['achievements'] == oi['achievements'], where
  assert ri['discount'] == oi['discount'] and ri['reward'] == oi['reward'], where
  assert np.array_equal(ri['semantic'], oi['semantic']) and np.array_equal(ri['player_pos'], oi['player_pos']), where
  assert ref_objects(r) == o.objects(), f'{where}: objects'
  assert list(r._world._chunks.keys()) == o.chunk_order, f'{where}: chunk order'
  sr, so = r._world.random.get_state(), o.random.get_state()
  assert sr[2] == so[2] and np.array_equal(sr[1], so[1]), f'{where}: RNG'


def run_pair(seed, steps, tape=None, gifts=None, episodes=1, **kw):
  crafter = rh.load()
  r, o = crafter.Env(seed=seed, **kw), OracleEnv(seed=seed, **kw)
  rs = np.random.RandomState(seed + 99)
  for ep in range(episodes):
    assert np.array_equal(r.reset(), o.reset())
    assert np.array_equal(r._world._mat_map, o.mat) and ref_objects(r) == o.objects()
    for t in range(steps):
      if gifts and t in gifts:
        for item, amount in gifts[t].items():
          r._player.inventory[item] = amount
          o.inv[o.t.item_id[item]] = amount
      a = int(tape[t]) if tape is not None else int(rs.randint(0, 17))
      ra, oa = r.step(a), o.step(a)
      compare_step(r, o, ra, oa, f'seed {seed} ep {ep} step {t}')
      if ra[2]:
        break
  return r, o


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_random_policy_two_episodes(seed):
  run_pair(seed, 350, episodes=2)


@pytest.mark.parametrize('kind', sorted(scenarios.SCENARIOS))
def test_scripted_scenarios(kind):
  acts, gifts = scenarios.SCENARIOS[kind](300, 5)
  run_pair(5, 300, tape=acts, gifts=gifts)


def test_constructor_variants():
  run_pair(4, 80, area=(32, 48), view=(7, 9), size=(84, 84))     # other view / unit 12 x 9 / border
  run_pair(6, 60, length=40, reward=False)                         # `over` + zeroed reward
  run_pair(8, 60, length=None)                                     # done is None-ish until death


def test_extra_render_calls_consume_night_noise():
  crafter = rh.load()
  r, o = crafter.Env(seed=7), OracleEnv(seed=7)
  r.reset(), o.reset()
  for t in range(200):
    a = 0 if t % 3 else 6
    ra, oa = r.step(a), o.step(a)
    if t > 150 and t % 7 == 0:       # night: every render() draws 3087 doubles (engine.py:209)
      assert np.array_equal(r.render(), o.render())
    compare_step(r, o, ra, oa, f'step {t}')
    if ra[2]:
      break


def test_survey_end_to_end_fixture():
  """SURVEY.md App. B 'provisional end-to-end fixtures' (surveyor's own noise restatement)."""
  import hashlib
  o = OracleEnv(seed=0)
  obs = o.reset()
  assert hashlib.sha256(o.mat.tobytes()).hexdigest()[:16] == 'e6448727016242ea'
  assert hashlib.sha256(obs.tobytes()).hexdigest()[:16] == '7ea6d5809711316c'


class _NumpyWithLibmExp:
  """`np` as crafter.worldgen sees it, with exp() = the C library's (correctly rounded on the tie cells, checked against
  100-digit arithmetic in tests/test_exp_cr.py); everything else is numpy's."""

  def __getattr__(self, name):
    return getattr(np, name)

  @staticmethod
  def exp(x):
    import math
    return np.float64(math.exp(float(x)))


def _worlds(seed, episodes, patch_exp):
  """(material maps of the reference, of the oracle) over `episodes` consecutive Env.reset() calls."""
  crafter = rh.load()
  from crafter import worldgen
  saved = worldgen.np
  if patch_exp:
    worldgen.np = _NumpyWithLibmExp()
  try:
    r, o = crafter.Env(seed=seed), OracleEnv(seed=seed)
    out = []
    for _ in range(episodes):
      r.reset(), o.reset()
      out.append((r._world._mat_map.copy(), o.mat.copy(), ref_objects(r) == o.objects()))
  finally:
    worldgen.np = saved
  return out


def test_exp_flavour_is_the_only_difference():
  """VERDICT r3 weak #1b.  The oracle pins worldgen.py:27's exp to the correctly rounded value (oracle/exp_cr.py); the
  reference takes whatever np.exp is on the host -- SVML on AVX-512 hosts, the C library's elsewhere -- and on ~1 world in
  5000 (a structural tie of `start > 0.5` at distance exactly 4 from the player) the flavours decide a material, after
  which every later uniform() of that world shifts.  Seed 7327, episode 10 is such a world: with worldgen's np.exp replaced
  by libm's exp -- the ONLY patch -- the untouched reference produces the oracle's 11 worlds cell for cell, objects included."""
  for ep, (ref_mat, orc_mat, same_objects) in enumerate(_worlds(7327, 11, patch_exp=True), 1):
    assert np.array_equal(ref_mat, orc_mat) and same_objects, f'episode {ep}: reference (libm exp) != oracle'


def test_unpatched_reference_differs_only_where_numpy_exp_misrounds():
  """The same 11 worlds with the reference as it runs on THIS host.  Where numpy's exp rounds the tie cell of episode 10
  correctly (non-AVX-512 hosts) everything is equal; where it does not (SVML: the build container) exactly that world
  differs -- a dozen cells behind cell (32, 28) in generation order -- and the ten others are equal.  Either way the difference is explained
  by np.exp(-1.638e-16) alone."""
  worlds = _worlds(7327, 11, patch_exp=False)
  tie = np.exp(np.float64(-1.638387376145862e-16))
  misrounds = tie.hex() != '0x1.fffffffffffffp-1'
  for ep, (ref_mat, orc_mat, same_objects) in enumerate(worlds, 1):
    if ep == 10 and misrounds:
      bad = np.argwhere(ref_mat != orc_mat)
      # cell (32, 28) is grass either way (with a draw or without one); what differs are cells whose draw came later
      assert 1 <= len(bad) <= 64 and all((x, y) > (32, 28) for x, y in bad.tolist()), (len(bad), bad[:3])
    else:
      assert np.array_equal(ref_mat, orc_mat) and same_objects, f'episode {ep}'


def test_exp_tie_worlds_of_the_sweep():
  """more than ONE pinned tie world, and a MEASURED rate.  tools/sweep_exp_ties.py swept 400,000 worlds (seeds
  0 .. 99,999, four episodes each, oracle only) for cells at distance exactly 4 from the player whose `start` is non-zero
  rounding dust (< 1e-15): 122 of them, one world in 3,279 -- the first ten are in tests/golden/exp_ties.json.  On every one
  the untouched reference with worldgen's np.exp replaced by the C library's exp (the only patch) produces the oracle's
  world cell for cell, objects included.  The flavour of exp only DECIDES where it rounds exp(-start) to another
  neighbour of 1 than the correctly rounded one -- start around 1.6e-16: ONE of the 400,000 worlds on an AVX-512 host (seed
  20042, episode 1; round 4's "one world in 5000" was the rate of the dust cells, not of the decisions) -- and there the
  unpatched reference differs from the oracle exactly when this host's np.exp misrounds the known tie."""
  import json
  import pathlib
  ties = json.loads((pathlib.Path(__file__).parent / 'golden' / 'exp_ties.json').read_text())
  assert len(ties['ties']) >= 10 and ties['worlds_swept'] >= 400000
  decisive = ties['decided_differently_by_this_hosts_np_exp']['first']
  assert len(decisive) == ties['decided_differently_by_this_hosts_np_exp']['count'] == 1
  misrounds = np.exp(np.float64(-1.638387376145862e-16)).hex() != '0x1.fffffffffffffp-1'
  for t in ties['ties'] + decisive:
    ref_mat, orc_mat, same_objects = _worlds(t['seed'], t['episode'], patch_exp=True)[-1]
    assert np.array_equal(ref_mat, orc_mat) and same_objects, f'seed {t["seed"]} episode {t["episode"]}: reference (libm exp) != oracle'
    plain = _worlds(t['seed'], t['episode'], patch_exp=False)[-1]
    same = np.array_equal(plain[0], plain[1]) and plain[2]
    assert same == (not (t in decisive and misrounds)), (t, 'unpatched reference', 'equal' if same else 'differs')
