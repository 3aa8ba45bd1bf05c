"""ctypes / numpy mirrors of the plain-data structs in crafter_amd/csrc/types.hpp.

The C-ABI library (include/crafter_hip.h) exchanges these by pointer.  ``check_sizes`` compares
every sizeof with what the loaded library reports, so a layout drift fails loudly at import.
"""
import ctypes as C

import numpy as np

MT_N = 624
CHUNK = 12
MAX_ITEMS = 16
MAX_ACH = 32
MAX_MATERIALS = 16
MAX_ACTIONS = 32
MAX_PLACE = 8
MAX_MAKE = 8
MAX_USES = 4

T_NONE, T_PLAYER, T_COW, T_ZOMBIE, T_SKELETON, T_ARROW, T_PLANT = range(7)
A_NOOP, A_MOVE, A_DO, A_SLEEP, A_PLACE, A_MAKE = range(6)
ST_OBJ_OVERFLOW, ST_BAD_ACTION, ST_STEP_OVERFLOW, ST_CHUNK_OVERFLOW, ST_POOL_MISMATCH, ST_PIPE_STALL = 1, 2, 4, 8, 16, 32
STATUS_NAMES = {
    ST_OBJ_OVERFLOW: 'object table overflow (raise max_objects)',
    ST_BAD_ACTION: 'action index out of range',
    ST_STEP_OVERFLOW: 'step beyond the daylight table',
    ST_CHUNK_OVERFLOW: 'chunk table overflow',
    ST_POOL_MISMATCH: 'world pool handed out the wrong episode',
    ST_PIPE_STALL: 'a bounded in-kernel wait ran out (reserved)',
}

# texture slots of TablePtrs.tex_tile (types.hpp TEX_*)
TEX_MATERIAL0 = 0
(TEX_PLAYER_LEFT, TEX_PLAYER_RIGHT, TEX_PLAYER_UP, TEX_PLAYER_DOWN, TEX_PLAYER_SLEEP, TEX_COW,
 TEX_ZOMBIE, TEX_SKELETON, TEX_ARROW_LEFT, TEX_ARROW_RIGHT, TEX_ARROW_UP, TEX_ARROW_DOWN, TEX_PLANT,
 TEX_PLANT_RIPE, TEX_COUNT) = range(17, 32)
SPRITE_NAMES = {
    TEX_PLAYER_LEFT: 'player-left', TEX_PLAYER_RIGHT: 'player-right', TEX_PLAYER_UP: 'player-up',
    TEX_PLAYER_DOWN: 'player-down', TEX_PLAYER_SLEEP: 'player-sleep', TEX_COW: 'cow',
    TEX_ZOMBIE: 'zombie', TEX_SKELETON: 'skeleton', TEX_ARROW_LEFT: 'arrow-left',
    TEX_ARROW_RIGHT: 'arrow-right', TEX_ARROW_UP: 'arrow-up', TEX_ARROW_DOWN: 'arrow-down',
    TEX_PLANT: 'plant', TEX_PLANT_RIPE: 'plant-ripe',
}

i32, u32, u8, f64 = C.c_int32, C.c_uint32, C.c_uint8, C.c_double


class ItemList(C.Structure):
  _fields_ = [('n', i32), ('item', i32 * MAX_USES), ('amount', i32 * MAX_USES), ('ach', i32 * MAX_USES)]


class CollectRule(C.Structure):
  _fields_ = [('valid', i32), ('leaves', i32), ('probability', f64), ('require', ItemList),
              ('receive', ItemList)]


class PlaceRule(C.Structure):
  _fields_ = [('valid', i32), ('is_object', i32), ('material', i32), ('ach', i32), ('where_mask', u32),
              ('pad', i32), ('uses', ItemList)]


class MakeRule(C.Structure):
  _fields_ = [('valid', i32), ('item', i32), ('gives', i32), ('ach', i32), ('nearby_mask', u32),
              ('pad', i32), ('uses', ItemList)]


_MATS = ['water', 'grass', 'stone', 'path', 'sand', 'tree', 'lava', 'coal', 'iron', 'diamond', 'table',
         'furnace']
_ITEMS = ['health', 'food', 'drink', 'energy', 'wood_sword', 'stone_sword', 'iron_sword']
_ACHS = ['wake_up', 'eat_plant', 'defeat_zombie', 'defeat_skeleton', 'eat_cow']


class Rules(C.Structure):
  _fields_ = (
      [('n_actions', i32), ('n_materials', i32), ('n_items', i32), ('n_achievements', i32),
       ('action_kind', u8 * MAX_ACTIONS), ('action_arg', u8 * MAX_ACTIONS),
       ('item_max', i32 * MAX_ITEMS), ('item_init', i32 * MAX_ITEMS),
       ('walkable_mask', u32), ('player_walkable_mask', u32), ('arrow_walkable_mask', u32),
       ('arrow_breaks_mask', u32)]
      + [('mat_' + m, i32) for m in _MATS]
      + [('item_' + m, i32) for m in _ITEMS]
      + [('ach_' + m, i32) for m in _ACHS]
      + [('collect', CollectRule * (MAX_MATERIALS + 1)), ('place', PlaceRule * MAX_PLACE),
         ('make', MakeRule * MAX_MAKE)])


class Config(C.Structure):
  _fields_ = [(n, i32) for n in (
      'num_envs', 'W', 'H', 'view_w', 'view_h', 'size_w', 'size_h', 'unit_x', 'unit_y', 'local_gw',
      'local_gh', 'item_gw', 'item_gh', 'border_x', 'border_y', 'icon_w', 'icon_h', 'digit_w', 'digit_h',
      'max_objects', 'nchunk_x', 'nchunk_y', 'length', 'update_dist', 'n_daylight', 'auto_reset',
      'want_semantic', 'render_obs', 'reward', 'step_threads', 'reset_threads', 'gen_period')]


class StatePtrs(C.Structure):
  _fields_ = [(n, C.c_void_p) for n in (
      'mat', 'objmap', 'objs', 'mt', 'rec', 'chunk_order', 'chunk_seen', 'census', 'semantic', 'prof', 'reset_q', 'pool_mat', 'pool_objs', 'pool_mt', 'pool_hdr',
      'pool_chunk_order', 'gen_q', 'gen_latest', 'terminal', 'pool_stats', 'pool_perm', 'pool_census')]


class TablePtrs(C.Structure):
  _fields_ = [(n, C.c_void_p) for n in (
      'rules', 'atlas', 'tex_tile', 'tex_icon', 'tex_digit', 'tex_alpha', 'item_pos', 'daylight',
      'vignette', 'unit255', 'render_static')]


# numpy views of the per-env records (same layout as the C structs)
OBJ_DTYPE = np.dtype([('type', 'u1'), ('health', 'i1'), ('fx', 'i1'), ('fy', 'i1'), ('x', '<u2'),
                      ('y', '<u2'), ('aux', '<i4'), ('pad', '<u4')])
REC_DTYPE = np.dtype([
    ('mt_pos', '<i4'), ('step', '<i4'), ('episode', '<i4'), ('nobj', '<i4'), ('seed_lane', '<u8'),
    ('nchunks_seen', '<i4'), ('status', '<u4'), ('inv', '<i4', (MAX_ITEMS,)), ('ach', '<i4', (MAX_ACH,)),
    ('hunger2', '<i4'), ('thirst2', '<i4'), ('fatigue2', '<i4'), ('recover2', '<i4'),
    ('player_last_health', '<i4'), ('env_last_health', '<i4'), ('unlocked', '<u4'), ('sleeping', '<i4'),
    ('dhealth', '<i4'), ('new_unlocked', '<u4'), ('dead', '<i4'), ('done', '<i4'), ('needs_reset', '<i4'),
    ('ep_dhealth', '<i4'), ('ep_unlock_steps', '<i4'), ('pad', '<i4', (1,))])
POOL_HDR_DTYPE = np.dtype([('ready', '<u8'), ('mt_pos', '<i4'), ('nobj', '<i4'), ('nchunks_seen', '<i4'),
                           ('pad', '<i4'), ('pending', '<i4'), ('pad2', '<i4')])
assert POOL_HDR_DTYPE.itemsize == 32
assert OBJ_DTYPE.itemsize == 16
assert REC_DTYPE.itemsize % 16 == 0, REC_DTYPE.itemsize

SIZES = {
    'Obj': OBJ_DTYPE.itemsize, 'EnvRec': REC_DTYPE.itemsize, 'Rules': C.sizeof(Rules),
    'Config': C.sizeof(Config), 'StatePtrs': C.sizeof(StatePtrs), 'TablePtrs': C.sizeof(TablePtrs),
}
SIZE_ORDER = ['Obj', 'EnvRec', 'Rules', 'Config', 'StatePtrs', 'TablePtrs']


def check_sizes(reported):
  """reported: list of ints in SIZE_ORDER from <lib>_struct_sizes()."""
  for name, got in zip(SIZE_ORDER, reported):
    if SIZES[name] != got:
      raise RuntimeError(f'ABI mismatch: sizeof({name}) python={SIZES[name]} library={got}')
