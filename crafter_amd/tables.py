"""Host-side table building: everything the kernels read but never compute.

* rule tables   -- the reference's data.yaml (constants.py:6-8), baked to crafter_amd/data/rules.json
                   by tools/bake_assets.py, compiled here into the integer-coded ``abi.Rules``;
* texture atlas -- the reference's 16x16 PNGs (engine.py:120-142) resized with Pillow itself
                   (NEAREST; its column choice is not floor((i+.5)*16/7));
* daylight(step) and the night vignette -- evaluated with the same numpy calls as the reference
                   (env.py:135-139, engine.py:213-218) so no transcendental runs on the device;
* geometry      -- unit/grid/border/icon sizes exactly as env.py:30-46 and engine.py:236-247 compute them.

Pure numpy + Pillow; no torch, no GPU.
"""
import ctypes
import json
import pathlib

import numpy as np
from PIL import Image

from . import abi

DATA = pathlib.Path(__file__).resolve().parent / 'data'
DIR_NAMES = ['left', 'right', 'up', 'down']  # objects.py:175


def load_rules(path=None):
  return json.loads(pathlib.Path(path or DATA / 'rules.json').read_text())


def load_textures(path=None):
  with np.load(path or DATA / 'textures.npz') as z:
    return {k: z[k] for k in z.files}


class RuleError(ValueError):
  pass


def _item_list(dst, mapping, item_id, ach_id=None, ach_prefix=None):
  if len(mapping) > abi.MAX_USES:
    raise RuleError(f'more than {abi.MAX_USES} entries in {mapping}')
  dst.n = len(mapping)
  for i, (name, amount) in enumerate(mapping.items()):
    dst.item[i] = item_id[name]
    dst.amount[i] = int(amount)
    dst.ach[i] = ach_id[f'{ach_prefix}{name}'] if ach_prefix else -1


def build_rules(rules):
  """dict (data.yaml layout) -> abi.Rules.  Unknown names fail here, loudly, instead of as a
  KeyError somewhere inside an episode (where the reference would raise)."""
  R = abi.Rules()
  actions, materials = list(rules['actions']), list(rules['materials'])
  items, achievements = list(rules['items']), list(rules['achievements'])
  if len(actions) > abi.MAX_ACTIONS or len(materials) > abi.MAX_MATERIALS - 1:
    raise RuleError('too many actions / materials')
  if len(items) > abi.MAX_ITEMS or len(achievements) > abi.MAX_ACH:
    raise RuleError('too many items / achievements')
  mat_id = {m: i + 1 for i, m in enumerate(materials)}  # 0 = None (engine.py:29-30)
  item_id = {m: i for i, m in enumerate(items)}
  ach_id = {m: i for i, m in enumerate(achievements)}
  R.n_actions, R.n_materials, R.n_items, R.n_achievements = map(len, (actions, materials, items, achievements))
  place_names, make_names = list(rules['place']), list(rules['make'])
  for i, name in enumerate(actions):
    if name == 'noop':
      kind, arg = abi.A_NOOP, 0
    elif name.startswith('move_'):
      kind, arg = abi.A_MOVE, DIR_NAMES.index(name[5:])
    elif name == 'do':
      kind, arg = abi.A_DO, 0
    elif name == 'sleep':
      kind, arg = abi.A_SLEEP, 0
    elif name.startswith('place_'):
      kind, arg = abi.A_PLACE, place_names.index(name[6:])
    elif name.startswith('make_'):
      kind, arg = abi.A_MAKE, make_names.index(name[5:])
    else:
      kind, arg = abi.A_NOOP, 0  # objects.py:109-123: an unmatched action name does nothing
    R.action_kind[i], R.action_arg[i] = kind, arg
  for i, name in enumerate(items):
    R.item_max[i] = int(rules['items'][name]['max'])
    R.item_init[i] = int(rules['items'][name]['initial'])
  mask = lambda names: sum(1 << mat_id[n] for n in names)
  R.walkable_mask = mask(rules['walkable'])
  R.player_walkable_mask = mask(list(rules['walkable']) + ['lava'])
  R.arrow_walkable_mask = mask(list(rules['walkable']) + ['water', 'lava'])
  R.arrow_breaks_mask = mask(['table', 'furnace'])
  for m in abi._MATS:
    setattr(R, 'mat_' + m, mat_id[m])
  for m in abi._ITEMS:
    setattr(R, 'item_' + m, item_id[m])
  for m in abi._ACHS:
    setattr(R, 'ach_' + m, ach_id[m])
  for name, info in rules['collect'].items():
    cr = R.collect[mat_id[name]]
    cr.valid = 1
    cr.leaves = mat_id[info['leaves']]
    cr.probability = float(info.get('probability', 1))
    _item_list(cr.require, info['require'], item_id)
    _item_list(cr.receive, info['receive'], item_id, ach_id, 'collect_')
  for k, name in enumerate(place_names):
    info = rules['place'][name]
    pr = R.place[k]
    pr.valid = 1
    pr.is_object = int(info['type'] == 'object')
    if info['type'] == 'material':
      pr.material = mat_id[name]
    elif name != 'plant':
      raise RuleError(f'placeable object {name!r} is not supported (reference: plant / fence only)')
    pr.ach = ach_id[f'place_{name}']
    pr.where_mask = mask(info['where'])
    _item_list(pr.uses, info['uses'], item_id)
  for k, name in enumerate(make_names):
    info = rules['make'][name]
    mk = R.make[k]
    mk.valid = 1
    mk.item = item_id[name]
    mk.gives = int(info['gives'])
    mk.ach = ach_id[f'make_{name}']
    mk.nearby_mask = mask(info['nearby'])
    _item_list(mk.uses, info['uses'], item_id)
  return R


def geometry(area, view, size, n_items):
  """env.py:30-46, 122-128 and engine.py:236-247 as plain ints."""
  view = np.array(view if hasattr(view, '__len__') else (view, view))
  size = np.array(size if hasattr(size, '__len__') else (size, size))
  unit = size // view
  item_rows = int(np.ceil(n_items / view[0]))
  local_grid = np.array([view[0], view[1] - item_rows])
  item_grid = np.array([view[0], item_rows])
  border = (size - (size // view) * view) // 2
  icon = (int(0.8 * unit[0]), int(0.8 * unit[1]))
  digit = (int(0.6 * unit[0]), int(0.6 * unit[1]))
  item_pos = np.zeros((abi.MAX_ITEMS, 4), np.int32)
  for index in range(n_items):
    cell = np.array((index % item_grid[0], index // item_grid[0]))
    item_pos[index, 0:2] = (cell * unit + 0.1 * unit).astype(np.int32)
    item_pos[index, 2:4] = (cell * unit + 0.4 * unit).astype(np.int32)
  return dict(view=view, size=size, unit=unit, local_grid=local_grid, item_grid=item_grid, border=border,
              icon=icon, digit=digit, item_pos=item_pos, area=tuple(int(a) for a in area))


def _resized(originals, cache, name, size):
  """engine.py:131-142 Textures.get on the [x][y]-transposed originals."""
  size = int(size[0]), int(size[1])
  key = name, size
  if key not in cache:
    img = originals[name].transpose((1, 0, 2))
    if img.shape[:2] != size:
      img = np.array(Image.fromarray(img).resize(size[::-1], resample=Image.NEAREST))
    cache[key] = img
  return cache[key]


def build_atlas(rules, textures, geo):
  """Packs every texture the render can touch at this unit into one RGBA byte array.
  Returns dict(atlas, tex_tile, tex_icon, tex_digit, tex_alpha)."""
  unit = geo['unit']
  cache = {}
  chunks, offset = [], 0
  alpha = np.zeros(abi.TEX_COUNT + abi.MAX_ITEMS + 11, np.uint8)

  def push(name, size, flag_index):
    nonlocal offset
    img = _resized(textures, cache, name, size)
    rgba = np.full(img.shape[:2] + (4,), 255, np.uint8)
    rgba[..., :img.shape[2]] = img
    alpha[flag_index] = int(img.shape[2] == 4)
    chunks.append(np.ascontiguousarray(rgba).reshape(-1))
    start = offset
    offset += rgba.size
    return start

  tex_tile = np.full(abi.TEX_COUNT, -1, np.int32)
  tex_tile[0] = push('unknown', unit, 0)                          # material None -> 'unknown'
  for i, name in enumerate(rules['materials']):
    tex_tile[1 + i] = push(name, unit, 1 + i)
  for slot, name in abi.SPRITE_NAMES.items():
    tex_tile[slot] = push(name, unit, slot)
  tex_icon = np.full(abi.MAX_ITEMS, -1, np.int32)
  for i, name in enumerate(rules['items']):
    tex_icon[i] = push(name, geo['icon'], abi.TEX_COUNT + i)
  tex_digit = np.full(11, -1, np.int32)
  for d in range(1, 10):
    tex_digit[d] = push(str(d), geo['digit'], abi.TEX_COUNT + abi.MAX_ITEMS + d)
  tex_digit[10] = push('unknown', geo['digit'], abi.TEX_COUNT + abi.MAX_ITEMS + 10)
  tex_digit[0] = tex_digit[10]
  alpha[abi.TEX_COUNT + abi.MAX_ITEMS] = alpha[abi.TEX_COUNT + abi.MAX_ITEMS + 10]
  atlas = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
  return dict(atlas=atlas, tex_tile=tex_tile, tex_icon=tex_icon, tex_digit=tex_digit, tex_alpha=alpha)


UNBOUNDED_DAYLIGHT = 100002   # Env(length=None): steps the table covers to begin with (BatchedEnv grows it before an episode gets there)


def daylight_table(n, head=None):
  """env.py:135-139 for step = 0..n-1, evaluated per step with numpy scalars like the reference.  head: a table for the
  first steps that is already there (only the rest is evaluated)."""
  out = np.empty(n, np.float64)
  start = 0 if head is None else min(len(head), n)
  if start:
    out[:start] = head[:start]
  for step in range(start, n):
    progress = (step / 300) % 1 + 0.3
    out[step] = 1 - np.abs(np.cos(np.pi * progress)) ** 3
  return out


def vignette_table(shape, stddev=0.5):
  """engine.py:213-218 (shape = LocalView canvas (w, h))."""
  xs, ys = np.meshgrid(np.linspace(-1, 1, shape[0]), np.linspace(-1, 1, shape[1]))
  return np.ascontiguousarray(1 - np.exp(-0.5 * (xs ** 2 + ys ** 2) / (stddev ** 2)).T)


def unit255_table():
  """float32(i) / 255 as numpy evaluates ``arr.astype(np.float32) / 255`` (engine.py:279-281)."""
  return (np.arange(256).astype(np.float32) / 255).astype(np.float32)


def default_max_objects(area):
  """Slot-table capacity: live objects only (freed slots are compacted every step).  Random-policy
  maxima: 91 live on 64x64, 1200 on 256x256; balance caps creatures per chunk."""
  cells = int(area[0]) * int(area[1])
  cap = max(256, cells // 32)
  return int(min(cap, 65535))


def make_config(num_envs, rules, area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000,
                max_objects=None, auto_reset=False, want_semantic=False, render_obs=True,
                n_daylight=None):
  geo = geometry(area, view, size, len(rules['items']))
  c = abi.Config()
  c.num_envs = int(num_envs)
  c.W, c.H = geo['area']
  c.view_w, c.view_h = map(int, geo['view'])
  c.size_w, c.size_h = map(int, geo['size'])
  c.unit_x, c.unit_y = map(int, geo['unit'])
  c.local_gw, c.local_gh = map(int, geo['local_grid'])
  c.item_gw, c.item_gh = map(int, geo['item_grid'])
  c.border_x, c.border_y = map(int, geo['border'])
  c.icon_w, c.icon_h = geo['icon']
  c.digit_w, c.digit_h = geo['digit']
  c.max_objects = int(max_objects or default_max_objects(geo['area']))
  c.nchunk_x = -(-c.W // abi.CHUNK)
  c.nchunk_y = -(-c.H // abi.CHUNK)
  c.length = int(length) if length else 0
  c.update_dist = 2 * int(max(geo['view']))
  c.n_daylight = int(n_daylight if n_daylight else (c.length + 2 if c.length else UNBOUNDED_DAYLIGHT))
  c.auto_reset = int(bool(auto_reset))
  c.want_semantic = int(bool(want_semantic))
  c.render_obs = int(bool(render_obs))
  c.reward = int(bool(reward))
  if c.unit_x < 1 or c.unit_y < 1 or c.local_gh < 1:
    raise ValueError('size / view leave no pixels for the local view')
  if c.W * c.H >= 65536 * 4 or c.max_objects > 65535:
    raise ValueError('area too large for 16-bit slot ids')
  return c, geo


class HostTables:
  """All read-only tables as numpy arrays + the ctypes Rules struct (kept alive here)."""

  def __init__(self, rules, textures, config, geo):
    self.rules_dict = rules
    self.rules = build_rules(rules)
    at = build_atlas(rules, textures, geo)
    self.atlas = at['atlas']
    self.tex_tile, self.tex_icon, self.tex_digit, self.tex_alpha = (
        at['tex_tile'], at['tex_icon'], at['tex_digit'], at['tex_alpha'])
    self.item_pos = np.ascontiguousarray(geo['item_pos'])
    self.daylight = daylight_table(config.n_daylight)
    lw, lh = config.local_gw * config.unit_x, config.local_gh * config.unit_y
    self.vignette = vignette_table((lw, lh))
    self.unit255 = unit255_table()

  def rules_bytes(self):
    return np.frombuffer(ctypes.string_at(ctypes.addressof(self.rules), ctypes.sizeof(self.rules)), np.uint8).copy()
