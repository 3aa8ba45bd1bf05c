"""The reference's object list is unbounded (engine.py:50-58); the device's slot table holds `max_objects` live objects
(256 on a 64x64 world: what the default step instance is compiled for) and doubles when three quarters full
(BatchedEnv._grow_objects, at check_errors()).  How close does a 64x64 world come?  VERDICT r4 #7: measured here, on the
oracle, under the policy that grows the list fastest -- a player who cannot die and plants saplings wherever it walks, for
the default episode length of 10,000 steps -- and asserted with a factor of two to spare."""
import numpy as np

from tests import scenarios
from tests.rollout import oracle_rollouts


def test_live_objects_under_a_planting_survivor_stay_below_half_the_slot_table():
  T = 10000
  specs = []
  for seed in (1, 2, 3, 4):
    acts, gifts = scenarios.planter_tape(T, seed)
    specs.append(dict(kwargs=dict(seed=seed, length=T), actions=acts, gifts=gifts))
  res = oracle_rollouts(specs)
  peak = [r['max_objects'] for r in res]
  played = [r['steps_played'] for r in res]
  print(f'live objects, peak per episode: {peak} over {played} steps (random policy: 91, SURVEY App. C)')
  assert sum(played) > 20000           # (lava and sleeping under zombies still end some episodes early)
  assert max(peak) <= 128, peak        # measured: 92 .. 117 -- half of the 256 slots; growth starts at 192
