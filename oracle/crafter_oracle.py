"""ORACLE -- test infrastructure only.

Nothing under crafter_amd/ may import this file; only tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg use it, and only as the checker / the timed CPU port.

A CPU restatement (pure Python + numpy) of the reference hot path
  /root/reference/crafter/env.py      (Env.reset/step/render, _balance_*)
  /root/reference/crafter/engine.py   (World, Textures, LocalView, ItemView, SemanticView)
  /root/reference/crafter/objects.py  (Player, Cow, Zombie, Skeleton, Arrow, Plant)
  /root/reference/crafter/worldgen.py (generate_world)
written as a struct-of-arrays state machine (integer material / object / item codes, flat
slot tables) instead of the reference's object graph, so that its state can be compared
field by field with the device state.  Each function cites the reference lines it follows.

Pinning: this restatement is checked against the UNTOUCHED reference, imported in the build
container by oracle/reference_harness.py (slot-order canonicalisation), in
tests/test_oracle_vs_reference.py, and against the golden fixtures that the same reference
produced (tests/golden/, made by tools/make_golden.py).  The reference itself has no tests,
golden vectors or fixtures.  One ingredient stays "parity unpinned": the terrain noise comes
from the absent third-party package ``opensimplex`` and is a restatement of its published
algorithm (oracle/opensimplex_ref.py).

Third-party arithmetic that IS installed is used directly, like the reference does:
numpy.random.RandomState (the env RNG), numpy cos/exp/linspace (daylight, vignette) and
Pillow's NEAREST resize (texture scaling).  ImageEnhance.Color is restated in integer/f32
arithmetic (see _desaturate) and pinned by the pixel-parity tests.
"""
import json
import pathlib

import numpy as np
from PIL import Image

from . import exp_cr as _exp_cr
from . import noise as _noise

_DATA = pathlib.Path(__file__).resolve().parent.parent / 'crafter_amd' / 'data'

# object type codes (order of env.py:47-49, the SemanticView class list)
PLAYER, COW, ZOMBIE, SKELETON, ARROW, PLANT = 1, 2, 3, 4, 5, 6
TYPE_NAMES = {1: 'Player', 2: 'Cow', 3: 'Zombie', 4: 'Skeleton', 5: 'Arrow', 6: 'Plant'}
DIRS = ((-1, 0), (1, 0), (0, -1), (0, 1))  # objects.py:33-34 all_dirs; also left,right,up,down
CHUNK = 12  # env.py:40


def load_rules(path=None):
  return json.loads(pathlib.Path(path or _DATA / 'rules.json').read_text())


def load_textures(path=None):
  with np.load(path or _DATA / 'textures.npz') as z:
    return {k: z[k] for k in z.files}


class Tables:
  """Integer-coded view of data.yaml (reference constants.py:6-8)."""

  def __init__(self, rules):
    self.rules = rules
    self.actions = list(rules['actions'])
    self.materials = list(rules['materials'])
    self.mat_id = {name: i + 1 for i, name in enumerate(self.materials)}  # engine.py:29-30
    self.mat_id[None] = 0
    self.items = list(rules['items'])
    self.item_id = {n: i for i, n in enumerate(self.items)}
    self.item_max = [rules['items'][n]['max'] for n in self.items]
    self.item_init = [rules['items'][n]['initial'] for n in self.items]
    self.achievements = list(rules['achievements'])
    self.ach_id = {n: i for i, n in enumerate(self.achievements)}
    self.walkable = {self.mat_id[m] for m in rules['walkable']}            # objects.py:21-22
    self.player_walkable = self.walkable | {self.mat_id['lava']}           # objects.py:96-97
    self.arrow_walkable = self.walkable | {self.mat_id['water'], self.mat_id['lava']}  # objects.py:369-371


class Textures:
  """engine.py:120-142: originals in [x][y] order, NEAREST resize with Pillow itself."""

  def __init__(self, originals):
    self._orig = {k: v.transpose((1, 0, 2)) for k, v in originals.items()}
    self._cache = {}

  def get(self, name, size):
    if name is None:
      name = 'unknown'
    size = int(size[0]), int(size[1])
    key = name, size
    if key not in self._cache:
      img = self._orig[name]
      if img.shape[:2] != size:
        img = np.array(Image.fromarray(img).resize(size[::-1], resample=Image.NEAREST))
      self._cache[key] = img
    return self._cache[key]


def daylight_of(step):
  """env.py:135-139, evaluated with the same numpy calls."""
  progress = (step / 300) % 1 + 0.3
  return 1 - np.abs(np.cos(np.pi * progress)) ** 3


def vignette(shape, stddev=0.5):
  """engine.py:213-218."""
  xs, ys = np.meshgrid(np.linspace(-1, 1, shape[0]), np.linspace(-1, 1, shape[1]))
  return 1 - np.exp(-0.5 * (xs ** 2 + ys ** 2) / (stddev ** 2)).T


def luma(rgb_u8):
  """Pillow RGB->L (ImageConvert L24): (19595 R + 38470 G + 7471 B + 0x8000) >> 16."""
  c = rgb_u8.astype(np.int64)
  return (19595 * c[..., 0] + 38470 * c[..., 1] + 7471 * c[..., 2] + 0x8000) >> 16


def _desaturate(rgb_u8, factor):
  """ImageEnhance.Color(img).enhance(factor) == Image.blend(gray, img, factor)
  (engine.py:192-193,199-200).  Pillow's ImagingBlend works in C float:
  out = (u8)((float)g + alpha * (float)(c - g)); alpha == 0 returns the gray image."""
  g = luma(rgb_u8)[..., None]
  if factor == 0.0:
    return np.repeat(g, 3, axis=-1).astype(np.uint8)
  diff = (rgb_u8.astype(np.int64) - g).astype(np.float32)
  out = g.astype(np.float32) + np.float32(factor) * diff
  return out.astype(np.uint8)


def py_world_seed(seed, episode):
  """env.py:74 -- CPython's own tuple hash."""
  return hash((seed, episode)) % (2 ** 31 - 1)


class OracleEnv:
  """Drop-in for the reference ``crafter.Env`` surface (env.py:25-133), SoA inside."""

  def __init__(self, area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000,
               seed=None, rules=None, textures=None):
    view = np.array(view if hasattr(view, '__len__') else (view, view))
    size = np.array(size if hasattr(size, '__len__') else (size, size))
    seed = np.random.randint(0, 2 ** 31 - 1) if seed is None else seed  # env.py:32
    self._area = tuple(int(a) for a in area)
    self._view = view
    self._size = size
    self._reward = reward
    self._length = length
    self._seed = seed
    self._episode = 0
    self.t = Tables(rules or load_rules())
    self._textures = Textures(textures or load_textures())
    item_rows = int(np.ceil(len(self.t.items) / view[0]))       # env.py:42
    self._local_grid = np.array([view[0], view[1] - item_rows])  # env.py:43-44
    self._item_grid = np.array([view[0], item_rows])             # env.py:45-46
    self._step = None
    self.reward_range = None
    self.metadata = None
    self.random = None
    self._clear_world(None)

  # ---------------------------------------------------------------- world state (engine.py:33-39)
  def _clear_world(self, seed):
    W, H = self._area
    self.random = np.random.RandomState(seed)
    self.daylight = 0.0
    self.mat = np.zeros((W, H), np.uint8)
    self.objmap = np.zeros((W, H), np.int64)
    # slot tables, slot 0 = None (engine.py:37); a removed slot has otype 0
    self.otype = [0]
    self.ox = [0]
    self.oy = [0]
    self.ohealth = [0]
    self.ofx = [0]
    self.ofy = [0]
    self.oaux = [0]       # zombie cooldown / skeleton reload / plant grown
    self.chunk_order = []  # chunk keys in first-insertion order (dict order of engine.py:36)
    self._chunk_seen = set()

  def chunk_key(self, x, y):
    """engine.py:112-117."""
    W, H = self._area
    xmin, ymin = (x // CHUNK) * CHUNK, (y // CHUNK) * CHUNK
    return (xmin, min(xmin + CHUNK, W), ymin, min(ymin + CHUNK, H))

  def _touch_chunk(self, x, y):
    key = self.chunk_key(x, y)
    if key not in self._chunk_seen:
      self._chunk_seen.add(key)
      self.chunk_order.append(key)

  def _inside(self, x, y):
    return 0 <= x < self._area[0] and 0 <= y < self._area[1]

  def _cell(self, x, y):
    """engine.py:88-93 World.__getitem__: (material id, slot) or (0, 0) outside.
    NB: 0 doubles as 'None' for both; a removed object's slot is already 0 in objmap."""
    if not self._inside(x, y):
      return 0, 0
    return int(self.mat[x, y]), int(self.objmap[x, y])

  def _add(self, typ, x, y, health=0, fx=0, fy=0, aux=0):
    """engine.py:50-57."""
    assert self.objmap[x, y] == 0
    slot = len(self.otype)
    self.otype.append(typ)
    self.ox.append(x)
    self.oy.append(y)
    self.ohealth.append(health)
    self.ofx.append(fx)
    self.ofy.append(fy)
    self.oaux.append(aux)
    self.objmap[x, y] = slot
    self._touch_chunk(x, y)
    return slot

  def _remove(self, slot):
    """engine.py:59-65.  The caller keeps executing on its own copy of the fields
    (a removed Cow/Zombie/Skeleton finishes its update, objects.py:275-279 etc.)."""
    if self.otype[slot] == 0:
      return
    self.objmap[self.ox[slot], self.oy[slot]] = 0
    self.otype[slot] = 0

  def _move(self, slot, x, y):
    """engine.py:67-80 (no-op for a removed object)."""
    if self.otype[slot] == 0:
      return
    assert self.objmap[x, y] == 0
    self.objmap[x, y] = slot
    self.objmap[self.ox[slot], self.oy[slot]] = 0
    self._touch_chunk(x, y)
    self.ox[slot] = x
    self.oy[slot] = y

  def _damage(self, slot, amount):
    """objects.py:28-30 health setter (clamped at 0); the player's health is an inventory item."""
    if self.otype[slot] == PLAYER:
      self.inv[self.t.item_id['health']] = max(0, self.inv[self.t.item_id['health']] - amount)
    else:
      self.ohealth[slot] = max(0, self.ohealth[slot] - amount)

  @property
  def health(self):
    return self.inv[self.t.item_id['health']]

  # ---------------------------------------------------------------- RNG helpers
  def _uniform(self):
    return self.random.uniform()

  def _randint(self, n):
    return int(self.random.randint(0, n))

  # ---------------------------------------------------------------- Env API (env.py:58-133)
  @property
  def action_names(self):
    return self.t.actions

  def reset(self):
    """env.py:70-81."""
    W, H = self._area
    self._episode += 1
    self._step = 0
    self._clear_world(py_world_seed(self._seed, self._episode))
    self.daylight = daylight_of(self._step)
    # Player.__init__ objects.py:70-82
    self.inv = list(self.t.item_init)
    self.ach = [0] * len(self.t.achievements)
    self.action = 'noop'
    self.sleeping = False
    self.hunger = 0
    self.thirst = 0
    self.fatigue = 0
    self.recover = 0
    self.player_last_health = self.health
    self._env_last_health = self.health
    slot = self._add(PLAYER, W // 2, H // 2, 0, 0, 1)
    assert slot == 1
    self._unlocked = set()
    self._generate_world()
    return self.render()

  def step(self, action):
    """env.py:83-118."""
    self._step += 1
    self.daylight = daylight_of(self._step)
    self.action = self.t.actions[action]
    limit = 2 * int(max(self._view))
    nslots = len(self.otype)  # snapshot: objects created this step are not updated (engine.py:41-44)
    for slot in range(1, nslots):
      if self.otype[slot] == 0:
        continue
      dist = abs(self.ox[slot] - self.ox[1]) + abs(self.oy[slot] - self.oy[1])
      if dist < limit:
        self._update_object(slot)
    if self._step % 10 == 0:
      for key in list(self.chunk_order):
        self._balance_chunk(key)
    obs = self.render()
    reward = (self.health - self._env_last_health) / 10
    self._env_last_health = self.health
    unlocked = {i for i, c in enumerate(self.ach) if c > 0 and i not in self._unlocked}
    if unlocked:
      self._unlocked |= unlocked
      reward += 1.0
    dead = self.health <= 0
    over = self._length and self._step >= self._length
    done = dead or over
    info = {
        'inventory': {n: self.inv[i] for i, n in enumerate(self.t.items)},
        'achievements': {n: self.ach[i] for i, n in enumerate(self.t.achievements)},
        'discount': 1 - float(dead),
        'semantic': self.semantic(),
        'player_pos': np.array([self.ox[1], self.oy[1]]),
        'reward': reward,
    }
    if not self._reward:
      reward = 0.0
    return obs, reward, done, info

  def semantic(self):
    """engine.py:251-264."""
    canvas = self.mat.copy()
    base = len(self.t.materials) + 1
    for slot in range(1, len(self.otype)):
      if self.otype[slot]:
        canvas[self.ox[slot], self.oy[slot]] = base + self.otype[slot] - 1
    return canvas

  # ---------------------------------------------------------------- object rules (objects.py)
  def _free(self, x, y, walkable):
    """objects.py:44-47."""
    m, o = self._cell(x, y)
    return o == 0 and m in walkable

  def _try_move(self, slot, px, py, dx, dy, walkable):
    """objects.py:36-42: (px, py) is the object's own idea of its position (stale if removed)."""
    tx, ty = px + dx, py + dy
    if self._free(tx, ty, walkable):
      self._move(slot, tx, ty)
      return True
    return False

  @staticmethod
  def _toward(px, py, tx, ty, long_axis=True):
    """objects.py:54-62."""
    ox, oy = tx - px, ty - py
    d0, d1 = abs(ox), abs(oy)
    sign = lambda v: (v > 0) - (v < 0)
    if (d0 > d1) if long_axis else (d0 <= d1):
      return sign(ox), 0
    return 0, sign(oy)

  def _random_dir(self):
    """objects.py:64-65."""
    return DIRS[self._randint(4)]

  def _update_object(self, slot):
    typ = self.otype[slot]
    if typ == PLAYER:
      self._update_player()
    elif typ == COW:
      self._update_cow(slot)
    elif typ == ZOMBIE:
      self._update_zombie(slot)
    elif typ == SKELETON:
      self._update_skeleton(slot)
    elif typ == ARROW:
      self._update_arrow(slot)
    elif typ == PLANT:
      self._update_plant(slot)

  def _update_cow(self, slot):
    """objects.py:274-279."""
    x, y = self.ox[slot], self.oy[slot]
    if self.ohealth[slot] <= 0:
      self._remove(slot)
    if self._uniform() < 0.5:
      dx, dy = self._random_dir()
      self._try_move(slot, x, y, dx, dy, self.t.walkable)

  def _update_zombie(self, slot):
    """objects.py:294-312."""
    x, y = self.ox[slot], self.oy[slot]
    if self.ohealth[slot] <= 0:
      self._remove(slot)
    px, py = self.ox[1], self.oy[1]
    dist = abs(px - x) + abs(py - y)
    if dist <= 8 and self._uniform() < 0.9:
      dx, dy = self._toward(x, y, px, py, self._uniform() < 0.8)
    else:
      dx, dy = self._random_dir()
    if self._try_move(slot, x, y, dx, dy, self.t.walkable) and self.otype[slot]:
      x, y = x + dx, y + dy
    dist = abs(px - x) + abs(py - y)
    if dist <= 1:
      if self.oaux[slot]:
        self.oaux[slot] -= 1
      else:
        damage = 7 if self.sleeping else 2
        self._damage(1, damage)
        self.oaux[slot] = 5

  def _update_skeleton(self, slot):
    """objects.py:327-351."""
    x, y = self.ox[slot], self.oy[slot]
    if self.ohealth[slot] <= 0:
      self._remove(slot)
    self.oaux[slot] = max(0, self.oaux[slot] - 1)
    px, py = self.ox[1], self.oy[1]
    dist = abs(px - x) + abs(py - y)
    if dist <= 3:
      dx, dy = self._toward(x, y, px, py, self._uniform() < 0.6)
      if self._try_move(slot, x, y, -dx, -dy, self.t.walkable):
        return
    if dist <= 5 and self._uniform() < 0.5:
      dx, dy = self._toward(x, y, px, py)
      # _shoot objects.py:343-351
      if self.oaux[slot] > 0:
        return
      if dx == 0 and dy == 0:
        return
      if self._free(x + dx, y + dy, self.t.arrow_walkable):
        self._add(ARROW, x + dx, y + dy, 0, dx, dy)
        self.oaux[slot] = 4
    elif dist <= 8 and self._uniform() < 0.3:
      dx, dy = self._toward(x, y, px, py, self._uniform() < 0.6)
      self._try_move(slot, x, y, dx, dy, self.t.walkable)
    elif self._uniform() < 0.2:
      dx, dy = self._random_dir()
      self._try_move(slot, x, y, dx, dy, self.t.walkable)

  def _update_arrow(self, slot):
    """objects.py:373-384."""
    x, y = self.ox[slot], self.oy[slot]
    fx, fy = self.ofx[slot], self.ofy[slot]
    tx, ty = x + fx, y + fy
    m, o = self._cell(tx, ty)
    if o:
      self._damage(o, 2)
      self._remove(slot)
    elif m not in self.t.arrow_walkable:
      self._remove(slot)
      if m in (self.t.mat_id['table'], self.t.mat_id['furnace']):
        self.mat[tx, ty] = self.t.mat_id['path']
    else:
      self._try_move(slot, x, y, fx, fy, self.t.arrow_walkable)

  def _update_plant(self, slot):
    """objects.py:405-411."""
    x, y = self.ox[slot], self.oy[slot]
    self.oaux[slot] += 1
    for dx, dy in DIRS:
      o = self._cell(x + dx, y + dy)[1]
      if o and self.otype[o] in (ZOMBIE, SKELETON, COW):
        self._damage(slot, 1)
        break
    if self.ohealth[slot] <= 0:
      self._remove(slot)

  # -- player (objects.py:99-261)
  def _update_player(self):
    t = self.t
    inv = self.inv
    x, y = self.ox[1], self.oy[1]
    fx, fy = self.ofx[1], self.ofy[1]
    tx, ty = x + fx, y + fy
    material, obj = self._cell(tx, ty)
    action = self.action
    energy = t.item_id['energy']
    if self.sleeping:                                   # objects.py:103-108
      if inv[energy] < t.item_max[energy]:
        action = 'sleep'
      else:
        self.sleeping = False
        self.ach[t.ach_id['wake_up']] += 1
    if action == 'noop':
      pass
    elif action.startswith('move_'):                    # objects.py:174-179
      fx, fy = dict(left=(-1, 0), right=(1, 0), up=(0, -1), down=(0, 1))[action[5:]]
      self.ofx[1], self.ofy[1] = fx, fy
      self._try_move(1, x, y, fx, fy, t.player_walkable)
      if self.mat[self.ox[1], self.oy[1]] == t.mat_id['lava']:
        inv[t.item_id['health']] = 0
    elif action == 'do' and obj:
      self._do_object(obj)
    elif action == 'do':
      self._do_material(tx, ty, material)
    elif action == 'sleep':                             # objects.py:117-119
      if inv[energy] < t.item_max[energy]:
        self.sleeping = True
    elif action.startswith('place_'):
      self._place(action[6:], tx, ty, material, obj)
    elif action.startswith('make_'):
      self._make(action[5:])
    self._life_stats()
    self._health_regen()
    for i in range(len(inv)):                           # objects.py:126-128
      inv[i] = max(0, min(inv[i], t.item_max[i]))
    if self.health < self.player_last_health:           # objects.py:169-172
      self.sleeping = False
    self.player_last_health = self.health

  def _life_stats(self):
    """objects.py:133-151."""
    inv, iid = self.inv, self.t.item_id
    self.hunger += 0.5 if self.sleeping else 1
    if self.hunger > 25:
      self.hunger = 0
      inv[iid['food']] -= 1
    self.thirst += 0.5 if self.sleeping else 1
    if self.thirst > 20:
      self.thirst = 0
      inv[iid['drink']] -= 1
    if self.sleeping:
      self.fatigue = min(self.fatigue - 1, 0)
    else:
      self.fatigue += 1
    if self.fatigue < -10:
      self.fatigue = 0
      inv[iid['energy']] += 1
    if self.fatigue > 30:
      self.fatigue = 0
      inv[iid['energy']] -= 1

  def _health_regen(self):
    """objects.py:153-167."""
    inv, iid = self.inv, self.t.item_id
    ok = inv[iid['food']] > 0 and inv[iid['drink']] > 0 and (inv[iid['energy']] > 0 or self.sleeping)
    if ok:
      self.recover += 2 if self.sleeping else 1
    else:
      self.recover -= 0.5 if self.sleeping else 1
    if self.recover > 25:
      self.recover = 0
      inv[iid['health']] = max(0, inv[iid['health']] + 1)
    if self.recover < -15:
      self.recover = 0
      inv[iid['health']] = max(0, inv[iid['health']] - 1)

  def _do_object(self, slot):
    """objects.py:181-212."""
    inv, iid, ach, aid = self.inv, self.t.item_id, self.ach, self.t.ach_id
    damage = max(1, inv[iid['wood_sword']] and 2, inv[iid['stone_sword']] and 3,
                 inv[iid['iron_sword']] and 5)
    typ = self.otype[slot]
    if typ == PLANT:
      if self.oaux[slot] > 300:
        self.oaux[slot] = 0
        inv[iid['food']] += 4
        ach[aid['eat_plant']] += 1
    if typ == ZOMBIE:
      self._damage(slot, damage)
      if self.ohealth[slot] <= 0:
        ach[aid['defeat_zombie']] += 1
    if typ == SKELETON:
      self._damage(slot, damage)
      if self.ohealth[slot] <= 0:
        ach[aid['defeat_skeleton']] += 1
    if typ == COW:
      self._damage(slot, damage)
      if self.ohealth[slot] <= 0:
        inv[iid['food']] += 6
        ach[aid['eat_cow']] += 1
        self.hunger = 0

  def _do_material(self, tx, ty, material):
    """objects.py:214-229."""
    t = self.t
    if material == t.mat_id['water']:
      self.thirst = 0
    name = t.materials[material - 1] if material else None
    info = t.rules['collect'].get(name)
    if not info:
      return
    for item, amount in info['require'].items():
      if self.inv[t.item_id[item]] < amount:
        return
    self.mat[tx, ty] = t.mat_id[info['leaves']]
    if self._uniform() <= info.get('probability', 1):
      for item, amount in info['receive'].items():
        self.inv[t.item_id[item]] += amount
        self.ach[t.ach_id[f'collect_{item}']] += 1

  def _place(self, name, tx, ty, material, obj):
    """objects.py:231-249."""
    t = self.t
    if obj:
      return
    info = t.rules['place'][name]
    if material not in [t.mat_id[m] for m in info['where']]:
      return
    if any(self.inv[t.item_id[k]] < v for k, v in info['uses'].items()):
      return
    for item, amount in info['uses'].items():
      self.inv[t.item_id[item]] -= amount
    if info['type'] == 'material':
      self.mat[tx, ty] = t.mat_id[name]
    elif info['type'] == 'object':
      assert name == 'plant'
      self._add(PLANT, tx, ty, 1, 0, 0, 0)
    self.ach[t.ach_id[f'place_{name}']] += 1

  def _make(self, name):
    """objects.py:251-261 with World.nearby's numpy slicing (engine.py:95-103): a negative slice
    start wraps, so the 3x3 window is EMPTY when x == 0 or y == 0."""
    t = self.t
    x, y = self.ox[1], self.oy[1]
    nearby = set(self.mat[x - 1: x + 2, y - 1: y + 2].flatten().tolist())
    info = t.rules['make'][name]
    if not all(t.mat_id[util] in nearby for util in info['nearby']):
      return
    if any(self.inv[t.item_id[k]] < v for k, v in info['uses'].items()):
      return
    for item, amount in info['uses'].items():
      self.inv[t.item_id[item]] -= amount
    self.inv[t.item_id[name]] += info['gives']
    self.ach[t.ach_id[f'make_{name}']] += 1

  # ---------------------------------------------------------------- balance (env.py:141-179)
  def _balance_chunk(self, key):
    light = self.daylight
    g, p = self.t.mat_id['grass'], self.t.mat_id['path']
    self._balance_object(key, ZOMBIE, g, 6, 0, 0.3, 0.4,
                         lambda space: (0 if space < 50 else 3.5 - 3 * light, 3.5 - 3 * light))
    self._balance_object(key, SKELETON, p, 7, 7, 0.1, 0.1,
                         lambda space: (0 if space < 6 else 1, 2))
    self._balance_object(key, COW, g, 5, 5, 0.01, 0.1,
                         lambda space: (0 if space < 30 else 1, 1.5 + light))

  def _balance_object(self, key, typ, material, span_dist, despan_dist, spawn_prob, despawn_prob,
                      target_fn):
    xmin, xmax, ymin, ymax = key
    # creatures of this class in the chunk, ascending slot order (canonical set order)
    creatures = [s for s in range(1, len(self.otype))
                 if self.otype[s] == typ and xmin <= self.ox[s] < xmax and ymin <= self.oy[s] < ymax]
    mask = self.mat[xmin:xmax, ymin:ymax] == material
    space = int(mask.sum())
    target_min, target_max = target_fn(space)
    px, py = self.ox[1], self.oy[1]
    if len(creatures) < int(target_min) and self._uniform() < spawn_prob:
      cells = np.argwhere(mask)  # row-major == x-major order of env.py:166-168
      i = self._randint(len(cells))
      x, y = int(cells[i][0]) + xmin, int(cells[i][1]) + ymin
      empty = self.objmap[x, y] == 0
      away = abs(x - px) + abs(y - py) >= span_dist
      if empty and away:
        health = {ZOMBIE: 5, SKELETON: 3, COW: 3}[typ]
        self._add(typ, x, y, health)
    elif len(creatures) > int(target_max) and self._uniform() < despawn_prob:
      s = creatures[self._randint(len(creatures))]
      away = abs(self.ox[s] - px) + abs(self.oy[s] - py) >= despan_dist
      if away:
        self._remove(s)

  # ---------------------------------------------------------------- worldgen (worldgen.py:10-91)
  def _generate_world(self):
    W, H = self._area
    t = self.t
    simplex = _noise.OpenSimplex(seed=self._randint(2 ** 31 - 1))   # worldgen.py:11
    px, py = self.ox[1], self.oy[1]
    uniform = self._uniform
    M = t.mat_id
    tunnels = np.zeros((W, H), bool)

    def S(x, y, z, sizes, normalize=True):                           # worldgen.py:79-91
      if not isinstance(sizes, dict):
        sizes = {sizes: 1}
      value = 0
      for size, weight in sizes.items():
        value += weight * simplex.noise3(x / size, y / size, z)
      if normalize:
        value /= sum(sizes.values())
      return value

    # worldgen.py:25-27 for every cell at once.  The exponential is the PINNED one (oracle/exp_cr.py: correctly rounded):
    # np.exp is SVML on AVX512 hosts and the C library's exp elsewhere, and their last-bit differences decide a material
    # in about one world in 5000 (a cell at distance exactly 4 from the player on which the noise vanishes).
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing='ij')
    start_all = 4 - np.sqrt(((xs - px) ** 2 + (ys - py) ** 2).astype(np.float64))
    start_all = start_all + 2 * simplex.noise3_many(xs / 3, ys / 3, np.full(xs.shape, 8.0))
    start_all = _exp_cr.sigmoid(start_all)
    for x in range(W):                                               # worldgen.py:21-61
      for y in range(H):
        start = start_all[x, y]
        water = S(x, y, 3, {15: 1, 5: 0.15}, False) + 0.1
        water -= 2 * start
        mountain = S(x, y, 0, {15: 1, 5: 0.3})
        mountain -= 4 * start + 0.3 * water
        if start > 0.5:
          m = M['grass']
        elif mountain > 0.15:
          if S(x, y, 6, 7) > 0.15 and mountain > 0.3:
            m = M['path']
          elif S(2 * x, y / 5, 7, 3) > 0.4:
            m = M['path']
            tunnels[x, y] = True
          elif S(x / 5, 2 * y, 7, 3) > 0.4:
            m = M['path']
            tunnels[x, y] = True
          elif S(x, y, 1, 8) > 0 and uniform() > 0.85:
            m = M['coal']
          elif S(x, y, 2, 6) > 0.4 and uniform() > 0.75:
            m = M['iron']
          elif mountain > 0.18 and uniform() > 0.994:
            m = M['diamond']
          elif mountain > 0.3 and S(x, y, 6, 5) > 0.35:
            m = M['lava']
          else:
            m = M['stone']
        elif 0.25 < water <= 0.35 and S(x, y, 4, 9) > -0.2:
          m = M['sand']
        elif 0.3 < water:
          m = M['water']
        else:
          if S(x, y, 5, 7) > 0 and uniform() > 0.8:
            m = M['tree']
          else:
            m = M['grass']
        self.mat[x, y] = m
    for x in range(W):                                               # worldgen.py:64-76
      for y in range(H):
        m = int(self.mat[x, y])
        dist = np.sqrt((x - px) ** 2 + (y - py) ** 2)
        if m not in t.walkable:
          pass
        elif dist > 3 and m == M['grass'] and uniform() > 0.985:
          self._add(COW, x, y, 3)
        elif dist > 10 and uniform() > 0.993:
          self._add(ZOMBIE, x, y, 5)
        elif m == M['path'] and tunnels[x, y] and uniform() > 0.95:
          self._add(SKELETON, x, y, 3)

  # ---------------------------------------------------------------- render (env.py:120-130)
  def _object_texture(self, slot):
    """objects.py:85-93,271,291,323,361-367,395-399."""
    typ = self.otype[slot]
    facing = {(-1, 0): 'left', (1, 0): 'right', (0, -1): 'up', (0, 1): 'down'}
    if typ == PLAYER:
      return 'player-sleep' if self.sleeping else 'player-' + facing[(self.ofx[1], self.ofy[1])]
    if typ == ARROW:
      return 'arrow-' + facing[(self.ofx[slot], self.ofy[slot])]
    if typ == PLANT:
      return 'plant-ripe' if self.oaux[slot] > 300 else 'plant'
    return {COW: 'cow', ZOMBIE: 'zombie', SKELETON: 'skeleton'}[typ]

  @staticmethod
  def _blit_alpha(canvas, x, y, texture):
    """engine.py:276-284."""
    w, h = texture.shape[:2]
    if texture.shape[-1] == 4:
      alpha = texture[..., 3:].astype(np.float32) / 255
      tex = texture[..., :3].astype(np.float32) / 255
      current = canvas[x: x + w, y: y + h].astype(np.float32) / 255
      blended = alpha * tex + (1 - alpha) * current
      texture = (255 * blended).astype(np.uint8)
    canvas[x: x + w, y: y + h] = texture

  def _local_view(self, unit):
    """engine.py:165-211."""
    grid = self._local_grid
    offset = grid // 2
    W, H = self._area
    px, py = self.ox[1], self.oy[1]
    ux, uy = int(unit[0]), int(unit[1])
    canvas = np.zeros((grid[0] * ux, grid[1] * uy, 3), np.uint8) + 127
    for gx in range(grid[0]):
      for gy in range(grid[1]):
        wx, wy = px + gx - offset[0], py + gy - offset[1]
        if not (0 <= wx < W and 0 <= wy < H):
          continue
        m = int(self.mat[wx, wy])
        name = self.t.materials[m - 1] if m else None
        tex = self._textures.get(name, unit)
        canvas[gx * ux: gx * ux + ux, gy * uy: gy * uy + uy] = tex[..., :3]
    for slot in range(1, len(self.otype)):
      if not self.otype[slot]:
        continue
      gx, gy = self.ox[slot] - px + offset[0], self.oy[slot] - py + offset[1]
      if not (0 <= gx < grid[0] and 0 <= gy < grid[1]):
        continue
      self._blit_alpha(canvas, gx * ux, gy * uy, self._textures.get(self._object_texture(slot), unit))
    # _light engine.py:189-196
    daylight = self.daylight
    night = canvas
    if daylight < 0.5:
      amount = 2 * (0.5 - daylight)
      noise = self.random.uniform(32, 127, canvas.shape[:2])[..., None]
      mask = amount * vignette(canvas.shape, 0.5)[..., None]
      night = (1 - mask) * canvas + mask * noise
    night = _desaturate(night.astype(np.uint8), 0.4)
    night = (1 - 0.5) * night + 0.5 * np.array((0, 16, 64))
    out = daylight * canvas + (1 - daylight) * night
    if self.sleeping:                                   # engine.py:198-202
      gray = _desaturate(out.astype(np.uint8), 0.0)
      out = (1 - 0.5) * gray + 0.5 * np.array((0, 0, 16))
    return out

  def _item_view(self, unit):
    """engine.py:221-248."""
    grid = self._item_grid
    unit = np.array(unit)
    canvas = np.zeros(tuple(grid * unit) + (3,), np.uint8)
    for index, item in enumerate(self.t.items):
      amount = self.inv[index]
      if amount < 1:
        continue
      cell = np.array((index % grid[0], index // grid[0]))
      pos = (cell * unit + 0.1 * unit).astype(np.int32)
      self._blit_alpha(canvas, int(pos[0]), int(pos[1]), self._textures.get(item, 0.8 * unit))
      pos = (cell * unit + 0.4 * unit).astype(np.int32)
      text = str(amount) if amount in range(10) else 'unknown'
      self._blit_alpha(canvas, int(pos[0]), int(pos[1]), self._textures.get(text, 0.6 * unit))
    return canvas

  def render(self, size=None):
    """env.py:120-130."""
    size = self._size if size is None else np.array(size)
    unit = size // self._view
    canvas = np.zeros(tuple(size) + (3,), np.uint8)
    view = np.concatenate([self._local_view(unit), self._item_view(unit)], 1)
    border = (size - (size // self._view) * self._view) // 2
    (x, y), (w, h) = border, view.shape[:2]
    canvas[x: x + w, y: y + h] = view
    return canvas.transpose((1, 0, 2))

  # ---------------------------------------------------------------- canonical state dump
  def objects(self):
    """Live objects in slot (creation) order as (type, x, y, health, fx, fy, aux)."""
    out = []
    for s in range(1, len(self.otype)):
      if self.otype[s]:
        health = self.health if self.otype[s] == PLAYER else self.ohealth[s]
        out.append((self.otype[s], self.ox[s], self.oy[s], health, self.ofx[s], self.ofy[s],
                    self.oaux[s]))
    return out

  def snapshot(self):
    kind, key, pos = self.random.get_state()[:3]
    return {
        'step': self._step, 'episode': self._episode,
        'mat': self.mat.copy(), 'occupied': (self.objmap > 0),
        'objects': self.objects(),
        'inventory': list(self.inv), 'achievements': list(self.ach),
        'sleeping': bool(self.sleeping),
        'hunger2': int(round(self.hunger * 2)), 'thirst2': int(round(self.thirst * 2)),
        'fatigue2': int(round(self.fatigue * 2)), 'recover2': int(round(self.recover * 2)),
        'player_last_health': self.player_last_health,
        'chunk_order': list(self.chunk_order),
        'mt_key': np.array(key, np.uint32), 'mt_pos': int(pos),
    }
