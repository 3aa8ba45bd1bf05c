"""TEST INFRASTRUCTURE: drives libhostsim.so (the kernel bodies on the CPU) with numpy buffers.
Mirrors what crafter_amd.batched.BatchedEnv does with torch tensors + libcrafter_hip.so."""
import ctypes as C
import hashlib

import numpy as np

from crafter_amd import abi, state, tables
from . import build as _build

_libs = {}
_STATIC_BLOCKS = {}   # renderer static blocks by content key (HostSimEnv._table_ptrs)


def lib(variant=None):
  """variant: None, or (tag, defines) for a differently configured build of the same sources."""
  if variant not in _libs:
    path = _build.build() if variant is None else _build.build(tag=variant[0], defines=variant[1])
    l = C.CDLL(str(path))
    sizes = (C.c_int32 * 6)()
    l.hostsim_struct_sizes(sizes)
    abi.check_sizes(list(sizes))
    l.hostsim_world_seed.restype = C.c_uint32
    l.hostsim_render_static_bytes.restype = C.c_longlong
    l.hostsim_world_seed.argtypes = [C.c_uint64, C.c_uint64]
    _libs[variant] = l
  return _libs[variant]


def _ptr(a):
  return a.ctypes.data_as(C.c_void_p)


class HostSimEnv:

  def __init__(self, seeds, area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000,
               rules=None, pool=False, variant=None, **kw):
    self.lib = lib(variant)
    self.variant = variant
    self.pool = int(bool(pool))
    self.rules_dict = rules or tables.load_rules()
    self.cfg, self.geo = tables.make_config(len(seeds), self.rules_dict, area, view, size, reward, length, **kw)
    self.tab = tables.HostTables(self.rules_dict, tables.load_textures(), self.cfg, self.geo)
    self.buf = {k: np.zeros(shape, dt) for k, (shape, dt) in state.state_spec(self.cfg).items()}
    self.rec = state.rec_view(self.buf['rec'])
    self.rec['seed_lane'] = state.seed_lanes(seeds)
    self.rec['mt_pos'] = abi.MT_N
    self.rec['nobj'] = 1
    self.st = abi.StatePtrs(prof=None, **{k: _ptr(v).value for k, v in self.buf.items()})
    self.pool_hdr = self.buf['pool_hdr'].view(abi.POOL_HDR_DTYPE).reshape(2, -1)
    self.terminal = self.buf['terminal']
    t = self.tab
    self._rules_buf = t.rules_bytes()
    self.tb, self._static = self._table_ptrs(self.cfg, t, self._rules_buf)
    n = self.cfg.num_envs
    self.obs = np.zeros((n, self.cfg.size_h, self.cfg.size_w, 3), np.uint8)
    self.reward = np.zeros(n, np.float32)
    self.done = np.zeros(n, np.uint8)

  def reset(self, mask=None):
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    self.lib.hostsim_reset(C.byref(self.cfg), C.byref(self.tb), C.byref(self.st),
                           None if m is None else _ptr(m), self.pool, _ptr(self.obs))
    return self.obs

  def step(self, actions):
    a = np.ascontiguousarray(actions, np.int32)
    self.lib.hostsim_step(C.byref(self.cfg), C.byref(self.tb), C.byref(self.st), _ptr(a), _ptr(self.obs),
                          _ptr(self.reward), _ptr(self.done), self.pool)
    return self.obs, self.reward, self.done

  def step_n(self, actions, stretch=16):
    """crafter_step_n on the harness: actions [T, N] -> obs [T, N, h, w, 3], reward [T, N], done [T, N]."""
    a = np.ascontiguousarray(actions, np.int32)
    T, n = a.shape
    obs = np.zeros((T,) + self.obs.shape, np.uint8)
    reward = np.zeros((T, n), np.float32)
    done = np.zeros((T, n), np.uint8)
    self.lib.hostsim_step_n(C.byref(self.cfg), C.byref(self.tb), C.byref(self.st), T, _ptr(a), _ptr(obs), _ptr(reward), _ptr(done),
                            self.pool, stretch)
    return obs, reward, done

  def _table_ptrs(self, cfg, t, rules_buf):
    """TablePtrs over the host arrays + the renderer's static block (built by the kernel code itself,
    as crafter_upload_tables does on the device)."""
    tb = abi.TablePtrs(
        rules=_ptr(rules_buf).value, atlas=_ptr(t.atlas).value, tex_tile=_ptr(t.tex_tile).value,
        tex_icon=_ptr(t.tex_icon).value, tex_digit=_ptr(t.tex_digit).value, tex_alpha=_ptr(t.tex_alpha).value,
        item_pos=_ptr(t.item_pos).value, daylight=_ptr(t.daylight).value, vignette=_ptr(t.vignette).value,
        unit255=_ptr(t.unit255).value, render_static=None)
    # the block depends on the frame geometry, the daylight table and the textures only: built once per distinct set
    # (tens of MB of lit rows -- seconds on the host)
    h = hashlib.sha1()
    for part in (np.array([cfg.unit_x, cfg.unit_y, cfg.local_gw, cfg.local_gh, cfg.item_gw, cfg.item_gh, cfg.size_w, cfg.size_h,
                           cfg.icon_w, cfg.icon_h, cfg.digit_w, cfg.digit_h, cfg.n_daylight], np.int64),
                 rules_buf, t.atlas, t.tex_tile, t.tex_icon, t.tex_digit, t.tex_alpha, t.item_pos, t.daylight, t.unit255):
      h.update(np.ascontiguousarray(part).tobytes())
    key = (self.variant, h.hexdigest(), int(self.lib.hostsim_render_static_bytes(C.byref(cfg))))   # (+ its size: a handle that never draws keeps no lit sprite rows)
    static = _STATIC_BLOCKS.get(key)
    if static is None:
      static = np.zeros(self.lib.hostsim_render_static_bytes(C.byref(cfg)), np.uint8)
      self.lib.hostsim_build_static(C.byref(cfg), C.byref(tb), _ptr(static))
      _STATIC_BLOCKS[key] = static
    tb.render_static = _ptr(static).value
    return tb, static

  # canonical per-env snapshot, comparable with OracleEnv.snapshot()
  def snapshot(self, i):
    r = self.rec[i]
    cfg = self.cfg
    objs = state.objs_view(self.buf['objs'])[i]
    health = r['inv'][self.tab.rules.item_health]
    return {
        'step': int(r['step']), 'episode': int(r['episode']),
        'mat': self.buf['mat'][i].reshape(cfg.W, cfg.H).copy(),
        'occupied': (state.occupied_cells(objs, r['nobj'], cfg) if self.lib.hostsim_slot_map_derived(C.byref(cfg))
                     else self.buf['objmap'][i].reshape(cfg.W, cfg.H) > 0),
        'objects': state.live_objects(objs, r['nobj'], health),
        'inventory': [int(v) for v in r['inv'][:self.tab.rules.n_items]],
        'achievements': [int(v) for v in r['ach'][:self.tab.rules.n_achievements]],
        'sleeping': bool(r['sleeping']),
        'hunger2': int(r['hunger2']), 'thirst2': int(r['thirst2']), 'fatigue2': int(r['fatigue2']),
        'recover2': int(r['recover2']), 'player_last_health': int(r['player_last_health']),
        'chunk_order': state.chunk_keys(self.buf['chunk_order'][i], r['nchunks_seen'], cfg),
        'mt_key': self.buf['mt'][i].copy(), 'mt_pos': int(r['mt_pos']),
    }

  def render(self, size=None):
    """Env.render(size): a second config over the same state buffers (as BatchedEnv.render does)."""
    if size is None:
      cfg, tb, shape = self.cfg, self.tb, self.obs.shape
    else:
      cfg, geo = tables.make_config(self.cfg.num_envs, self.rules_dict, (self.cfg.W, self.cfg.H),
                                    (self.cfg.view_w, self.cfg.view_h), size, True, self.cfg.length,
                                    max_objects=self.cfg.max_objects, n_daylight=self.cfg.n_daylight)
      t = tables.HostTables(self.rules_dict, tables.load_textures(), cfg, geo)
      self._aux = (t, t.rules_bytes())
      tb, static = self._table_ptrs(cfg, t, self._aux[1])
      self._aux = self._aux + (static,)
      shape = (cfg.num_envs, cfg.size_h, cfg.size_w, 3)
    out = np.zeros(shape, np.uint8)
    self.lib.hostsim_render(C.byref(cfg), C.byref(tb), C.byref(self.st), None, _ptr(out))
    return out
