"""Scripted action tapes + inventory 'gifts' that drive the rules a random policy rarely reaches:
crafting at tables/furnaces, placing stone/plants, sword combat, sleeping through the night."""
import numpy as np

A = dict(noop=0, left=1, right=2, up=3, down=4, do=5, sleep=6, place_stone=7, place_table=8,
         place_furnace=9, place_plant=10, make_wood_pickaxe=11, make_stone_pickaxe=12,
         make_iron_pickaxe=13, make_wood_sword=14, make_stone_sword=15, make_iron_sword=16)

RICH = dict(wood=9, stone=9, coal=9, iron=9, sapling=9, wood_pickaxe=1, stone_pickaxe=1, iron_pickaxe=1)


def builder_tape(n, seed):
  """Walk a little, then keep placing tables / furnaces / stone / plants, crafting and hitting things."""
  rs = np.random.RandomState(seed)
  pool = [1, 2, 3, 4, 1, 2, 3, 4, 5, 5, 5, 5, 7, 8, 9, 10, 10, 11, 12, 13, 14, 15, 16]
  acts = rs.choice(pool, size=n)
  gifts = {t: dict(RICH) for t in range(0, n, 20)}
  return acts.astype(np.int32), gifts


def sleeper_tape(n, seed):
  """Drain energy artificially, sleep a lot (night render + sleeping tint + wake-up logic)."""
  rs = np.random.RandomState(seed)
  acts = rs.choice([6, 6, 6, 6, 0, 1, 2, 3, 4, 5], size=n)
  gifts = {t: dict(food=9, drink=9, health=9) for t in range(0, n, 35)}
  for t in (0, 170):
    gifts[t] = dict(energy=3, food=9, drink=9, health=9)
  return acts.astype(np.int32), gifts


def fighter_tape(n, seed):
  """Swords in hand, walk and hit: zombie / skeleton / cow damage and achievements."""
  rs = np.random.RandomState(seed)
  acts = rs.choice([1, 2, 3, 4, 5, 5, 5], size=n)
  gifts = {t: dict(wood_sword=1, stone_sword=1, iron_sword=(t // 40) % 2, health=9) for t in range(0, n, 40)}
  return acts.astype(np.int32), gifts


SCENARIOS = {'builder': builder_tape, 'sleeper': sleeper_tape, 'fighter': fighter_tape}


def survivor_tape(n, seed):
  """One long episode (the reference's default length is 10000, env.py:27-29): health / food / drink topped up every
  step so that the player outlives the random policy's ~170 steps by far, energy drained now and then so that
  it sleeps through parts of the day and of the night on both sides of step 1024 (where the renderer's table of
  pre-lit rows ends, render.hpp kLitSteps)."""
  rs = np.random.RandomState(seed)
  acts = rs.choice([0, 0, 1, 2, 3, 4, 5, 6], size=n)
  gifts = {t: dict(health=9, food=9, drink=9) for t in range(n)}
  for t in range(120, n, 290):      # 120, 410, 700, 990, ...: falls asleep in the evening / before step 1024
    gifts[t] = dict(health=9, food=9, drink=9, energy=2)
    acts[t:t + 3] = 6
  for t in range(260, n, 290):      # mornings: wide awake again
    gifts[t] = dict(health=9, food=9, drink=9, energy=9)
  return acts.astype(np.int32), gifts


SCENARIOS['survivor'] = survivor_tape


def planter_tape(n, seed):
  """A player kept alive (health / food / drink / energy and nine saplings topped up every step) who walks, collects and
  plants wherever it stands for a whole default-length episode: the policy that makes the reference's object list
  (engine.py:50-58: append-only, unbounded) as long as the rules allow -- every plant is an object until a cow or a
  zombie next to it eats it (objects.py:405-411)."""
  rs = np.random.RandomState(seed)
  acts = rs.choice([1, 2, 3, 4, 5, 5, 10, 10, 10], size=n)
  gifts = {t: dict(health=9, food=9, drink=9, energy=9, sapling=9) for t in range(n)}
  return acts.astype(np.int32), gifts


SCENARIOS['planter'] = planter_tape
