"""BatchedEnv: N Crafter environments resident on one MI355X, stepped by libcrafter_hip.so.

This is the batched form of the reference's ``crafter.Env`` (env.py:25-133): same constructor
arguments per environment, ``reset()`` / ``step(actions)`` / ``render()`` over tensors.  All world
state lives in caller-owned torch tensors on the GPU (struct-of-arrays, types.hpp StatePtrs); this
class only allocates them, uploads the host-evaluated tables and enqueues kernels on the current
torch stream.  There is no CPU implementation behind it.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import abi, lib as _libmod, state, tables

_TORCH_DTYPE = {np.uint8: torch.uint8, np.uint16: torch.int16, np.uint32: torch.int32, np.int32: torch.int32}


class CrafterDeviceError(RuntimeError):
  pass


class _Handle:
  """One crafter_handle: config + uploaded tables (+ the state buffers bound to it)."""

  def __init__(self, lib, cfg, host_tables, device):
    self.lib, self.cfg, self.tables, self.device = lib, cfg, host_tables, device
    self._pid = os.getpid()   # the handle belongs to this process: a forked child must never call into HIP with it
    self.ptr = C.c_void_p()
    with torch.cuda.device(device):
      if lib.crafter_create(C.byref(cfg), C.byref(self.ptr)):
        raise CrafterDeviceError(_libmod.last_error(lib, None))
      t = host_tables
      self._rules_buf = t.rules_bytes()
      p = lambda a: a.ctypes.data_as(C.c_void_p)
      ht = _libmod.HostTablesC(
          rules=p(self._rules_buf), atlas=p(t.atlas), atlas_bytes=t.atlas.nbytes,
          tex_tile=p(t.tex_tile), n_tex_tile=t.tex_tile.size, tex_icon=p(t.tex_icon), n_tex_icon=t.tex_icon.size,
          tex_digit=p(t.tex_digit), n_tex_digit=t.tex_digit.size, tex_alpha=p(t.tex_alpha),
          n_tex_alpha=t.tex_alpha.size, item_pos=p(t.item_pos), n_item_pos=t.item_pos.size,
          daylight=p(t.daylight), n_daylight=t.daylight.size, vignette=p(t.vignette),
          n_vignette=t.vignette.size, unit255=p(t.unit255), n_unit255=t.unit255.size)
      self.check(lib.crafter_upload_tables(self.ptr, C.byref(ht)))

  def check(self, rc):
    if rc:
      raise CrafterDeviceError(_libmod.last_error(self.lib, self.ptr))

  def bind(self, state_ptrs):
    self.check(self.lib.crafter_bind_state(self.ptr, C.byref(state_ptrs)))

  def close(self):
    if self.ptr and self.ptr.value and os.getpid() == self._pid:
      self.lib.crafter_destroy(self.ptr)
    self.ptr = C.c_void_p()


class BatchedEnv:

  def __init__(self, num_envs, area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000,
               seed=None, seeds=None, device='cuda', auto_reset=True, semantic=False, render=True,
               max_objects=None, rules=None, textures=None, gen_period=0, grow_objects=True):
    if not torch.cuda.is_available():
      raise CrafterDeviceError('BatchedEnv needs a HIP device (torch.cuda.is_available() is False); '
                               'there is no CPU path')
    self._lib = _libmod.load()
    self.device = torch.device(device)
    if self.device.type != 'cuda':
      raise CrafterDeviceError(f'device must be a GPU, got {device!r}')
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    self.num_envs = int(num_envs)
    if seeds is None:
      if seed is None:
        seeds = [int(np.random.randint(0, 2 ** 31 - 1)) for _ in range(self.num_envs)]  # env.py:32
      else:
        seeds = [seed + i for i in range(self.num_envs)]
    self.seeds = list(seeds)
    if len(self.seeds) != self.num_envs:
      raise ValueError('len(seeds) != num_envs')
    self.rules = rules or tables.load_rules()
    self.cfg, self.geo = tables.make_config(
        self.num_envs, self.rules, area, view, size, reward, length, max_objects=max_objects,
        auto_reset=auto_reset, want_semantic=semantic, render_obs=render)
    self.cfg.step_threads = 0    # workgroup sizes are compile-time constants of the library
    self.cfg.reset_threads = 0
    self.cfg.gen_period = int(gen_period)   # world pool: 0 = default, < 0 = off (auto-reset always regenerates inline)
    self.tables = tables.HostTables(self.rules, textures or tables.load_textures(), self.cfg, self.geo)
    self._ctor = dict(area=area, view=view, reward=reward, length=length, max_objects=self.cfg.max_objects)
    self.action_names = list(self.rules['actions'])
    self.item_names = list(self.rules['items'])
    self.achievement_names = list(self.rules['achievements'])
    self._textures = textures or tables.load_textures()
    self._native = _Handle(self._lib, self.cfg, self.tables, self.device)
    self._handle = self._native.ptr
    self._aux = {}   # render(size) handles for other frame sizes, bound to the same state
    with torch.cuda.device(self.device):
      self._alloc_state()
    self._pool_warned = False
    self.grow_objects = bool(grow_objects)   # check_errors() doubles the slot table before an env can fill it (_grow_objects)
    # Env(length=None): no episode may outrun the daylight table (_grow_daylight).  _step_bound >= every env's step counter.
    self._unbounded = self.cfg.length == 0
    self._step_bound = 0
    if self._unbounded and self.cfg.n_daylight < 1024:   # (crafter_extend_daylight refuses shorter tables: say so now, not deep into a run)
      raise ValueError('length=None needs a daylight table of at least 1024 steps to grow from')

  # ------------------------------------------------------------------ setup
  def _check(self, rc):
    if rc:
      raise CrafterDeviceError(_libmod.last_error(self._lib, self._handle))

  def _alloc_state(self):
    cfg = self.cfg
    self.state = {}
    for name, (shape, dt) in state.state_spec(cfg).items():
      if name == 'semantic' and not cfg.want_semantic:
        continue
      if name in state.POOL_BUFFERS and not (cfg.auto_reset and cfg.gen_period >= 0):
        continue
      self.state[name] = torch.zeros(shape, dtype=_TORCH_DTYPE[dt], device=self.device)
    rec = np.zeros(self.num_envs, abi.REC_DTYPE)
    rec['seed_lane'] = state.seed_lanes(self.seeds)
    rec['mt_pos'] = abi.MT_N
    rec['nobj'] = 1
    self.state['rec'].copy_(torch.from_numpy(rec.view(np.uint8).reshape(self.num_envs, -1)))
    ptrs = {k: v.data_ptr() for k, v in self.state.items()}
    for name in ('semantic', 'prof') + state.POOL_BUFFERS:
      ptrs.setdefault(name, None)
    self.terminal = self.state['terminal']
    self._st = abi.StatePtrs(**ptrs)
    self._native.bind(self._st)
    n = self.num_envs
    self.obs = torch.zeros((n, cfg.size_h, cfg.size_w, 3), dtype=torch.uint8, device=self.device)
    self.reward = torch.zeros(n, dtype=torch.float32, device=self.device)
    self.done = torch.zeros(n, dtype=torch.uint8, device=self.device)
    self._rec_i32 = self.state['rec'].view(torch.int32)
    off = {name: abi.REC_DTYPE.fields[name][1] // 4 for name in abi.REC_DTYPE.names}
    self._off = off

  def __del__(self):
    try:
      for h in list(getattr(self, '_aux', {}).values()) + [getattr(self, '_native', None)]:
        if h is not None:
          h.close()
    except Exception:
      pass

  # ------------------------------------------------------------------ spaces (env.py:58-68)
  @property
  def observation_shape(self):
    return (self.cfg.size_h, self.cfg.size_w, 3)

  @property
  def num_actions(self):
    return len(self.action_names)

  @property
  def lds_bytes(self):
    return int(self._lib.crafter_lds_bytes(self._handle))

  @property
  def slot_map_derived(self):
    return bool(self._lib.crafter_slot_map_derived(self._handle))

  @property
  def step_instance(self):
    """Template instance of the step kernel this batch runs, e.g. 'crafter_step_kernel<1, 1, 1>' (maps in LDS,
    default geometry compiled in, default rules compiled in) -- the generic instances are slower."""
    k = int(self._lib.crafter_step_instance(self._handle))
    if k & 8:   # a world whose maps stay in global memory, seen through the default view (compiled in) with the default rules
      return f'crafter_step_kernel<0, 2, {k & 1}>'
    return f'crafter_step_kernel<{(k >> 2) & 1}, {(k >> 1) & 1}, {k & 1}>'

  def _stream(self):
    return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  _DAYLIGHT_MARGIN = 8

  def _grow_daylight(self, steps):
    """Env(length=None) (env.py:29): before `steps` more steps are launched, makes sure the daylight table (env.py:135-139,
    one host-evaluated value per step of an episode) reaches beyond the longest episode under way.  The bound kept
    between calls is the number of steps launched since the last look at the device; only when THAT gets near the end
    of the table (every ~100,000 calls) are the envs' step counters read back (one synchronisation) and, if an episode
    really is that long, the table doubled (crafter_extend_daylight, on every handle over this state)."""
    self._step_bound += steps
    if self._step_bound + self._DAYLIGHT_MARGIN < self.cfg.n_daylight:
      return
    self._step_bound = int(self._rec_i32[:, self._off['step']].max().item()) + steps
    need = self._step_bound + self._DAYLIGHT_MARGIN
    if need < self.cfg.n_daylight - max(1024, self.cfg.n_daylight // 8):   # (only when an episode really nears the table's end: ADVICE r4)
      return
    n = max(2 * self.cfg.n_daylight, need + tables.UNBOUNDED_DAYLIGHT)
    table = tables.daylight_table(n, head=self.tables.daylight)
    for h in [self._native] + list(self._aux.values()):
      with torch.cuda.device(self.device):
        h.check(self._lib.crafter_extend_daylight(h.ptr, table.ctypes.data_as(C.c_void_p), n))
      h.cfg.n_daylight = n
      h.tables.daylight = table
    self.cfg.n_daylight = n
    self.tables.daylight = table

  def _grow_objects(self, new_max):
    """Doubles the slot table (World._objects is an unbounded list in the reference, engine.py:50-58): synchronises, moves
    every env's table -- and its two pooled worlds' -- into larger buffers, and puts a new native handle over the state
    (`max_objects` is part of the handle's configuration and of the kernels' LDS layout: 256 slots with one-byte slot ids are
    what the default instance is compiled for, anything larger runs the generic instance).  The world pool starts afresh --
    its batches in flight belonged to the old handle: headers and request queues are cleared, an env whose next world is
    missing regenerates it inline (same generator, same (seed, episode): unobservable) and asks again."""
    new_max = int(min(new_max, 65535))
    if new_max <= self.cfg.max_objects:
      return
    torch.cuda.synchronize(self.device)
    old = self._native
    for h in self._aux.values():
      h.close()
    self._aux = {}
    cfg = abi.Config.from_buffer_copy(bytes(self.cfg))
    cfg.max_objects = new_max
    with torch.cuda.device(self.device):
      native = _Handle(self._lib, cfg, self.tables, self.device)   # (raises if one env no longer fits the LDS-resident kernels)
      for name in ('objs', 'pool_objs'):
        if name not in self.state:
          continue
        t = self.state[name]
        shape = list(t.shape)
        shape[-2] = new_max
        bigger = torch.zeros(shape, dtype=t.dtype, device=self.device)
        bigger[..., :t.shape[-2], :] = t
        self.state[name] = bigger
      for name in ('pool_hdr', 'gen_latest', 'gen_q', 'reset_q'):
        if name in self.state:
          self.state[name].zero_()
      torch.cuda.synchronize(self.device)
      ptrs = {k: v.data_ptr() for k, v in self.state.items()}
      for name in ('semantic', 'prof') + state.POOL_BUFFERS:
        ptrs.setdefault(name, None)
      if getattr(self, '_prof', None) is not None:
        ptrs['prof'] = self._prof.data_ptr()
      self._st = abi.StatePtrs(**ptrs)
      native.bind(self._st)
    old.close()
    was_default = self.cfg.max_objects == 256
    self._native, self._handle = native, native.ptr
    self.cfg = cfg
    self._ctor['max_objects'] = new_max
    self.objects_grown = getattr(self, 'objects_grown', 0) + 1
    if getattr(self, '_timing_on', False):   # handle-level settings move with the handle
      with torch.cuda.device(self.device):
        self._check(self._lib.crafter_set_timing(self._handle, 1))
    if was_default:
      import warnings
      warnings.warn(f'crafter_amd: an environment holds more than 192 objects: the slot tables grow to {new_max} entries and the batch '
                    'leaves the compiled default instance (256 slots, one-byte slot ids) for the generic kernels, which are slower',
                    RuntimeWarning, stacklevel=3)

  # ------------------------------------------------------------------ Env API
  def reset(self, mask=None):
    """Env.reset() (env.py:70-81) for all envs, or those with a non-zero mask byte.  Returns obs."""
    mptr = None
    if mask is not None:
      mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
      mptr = C.c_void_p(mask.data_ptr())
    with torch.cuda.device(self.device):
      self._check(self._lib.crafter_reset(self._handle, mptr, C.c_void_p(self.obs.data_ptr()), self._stream()))
    self._keep = mask
    return self.obs

  def step(self, actions, info=True, out=None):
    """Env.step() (env.py:83-118) for all envs.  actions: int tensor [N] on the device.
    Returns (obs u8[N,H,W,3], reward f32[N], done u8[N], info dict of device tensor views).
    With auto_reset, a finished env comes back already regenerated (its obs is the first frame
    of the next episode; done/reward still describe the finished step).
    out = (obs or None, reward, done): device tensors the kernels write instead of self.obs / self.reward /
    self.done (e.g. the send buffer of crafter_amd.dist.StepExchange); obs must be 16-byte aligned."""
    if not (torch.is_tensor(actions) and actions.dtype == torch.int32 and actions.is_cuda):
      actions = torch.as_tensor(actions, device=self.device).to(torch.int32)
    actions = actions.contiguous()
    obs, reward, done = self.obs, self.reward, self.done
    if out is not None:
      obs = obs if out[0] is None else out[0]
      reward, done = out[1], out[2]
      self._check_out(obs, self.obs), self._check_out(reward, self.reward), self._check_out(done, self.done)
    if self._unbounded:
      self._grow_daylight(1)
    with torch.cuda.device(self.device):
      self._check(self._lib.crafter_step(
          self._handle, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
          C.c_void_p(reward.data_ptr()), C.c_void_p(done.data_ptr()), self._stream()))
    self._keep = (actions, obs, reward, done)
    return obs, reward, done, (self.info() if info else {})

  def rollout(self, actions, out=None, obs=True):
    """T steps in one call for policies that choose their actions without looking at the observations (random,
    scripted, action repeat; run_random.py:36-44 is such a loop): actions int32 [T, N] on the device.  Returns
    (obs u8[T,N,H,W,3] or None, reward f32[T,N], done u8[T,N]) -- bit-identical to T calls of step(), faster because an env
    starts its step t + 1 without waiting for every other env's step t (crafter_step_n).  out = (obs or None, reward,
    done): tensors of those shapes to write into.  self.obs / self.reward / self.done are NOT updated; the state is."""
    if not (torch.is_tensor(actions) and actions.dtype == torch.int32 and actions.is_cuda):
      actions = torch.as_tensor(actions, device=self.device).to(torch.int32)
    actions = actions.contiguous()
    if actions.dim() != 2 or actions.shape[1] != self.num_envs or actions.shape[0] < 1:
      raise ValueError(f'actions must have shape [T, {self.num_envs}]')
    T = int(actions.shape[0])
    obs = obs and bool(self.cfg.render_obs)   # render=False: no kernel draws, there is no frame to return (step() hands out zeros)
    if out is None:
      o = torch.empty((T,) + tuple(self.obs.shape), dtype=torch.uint8, device=self.device) if obs else None
      r = torch.empty((T, self.num_envs), dtype=torch.float32, device=self.device)
      d = torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device)
    else:
      o, r, d = out
      for t, like in ((o, self.obs), (r, self.reward), (d, self.done)):
        if t is None and like is self.obs:
          continue
        if not (t.is_cuda and t.dtype == like.dtype and tuple(t.shape) == (T,) + tuple(like.shape) and t.is_contiguous() and
                t.data_ptr() % 16 == 0):
          raise ValueError(f'out tensor must be a contiguous, 16-byte aligned {like.dtype} device tensor of shape {(T,) + tuple(like.shape)}')
    if self._unbounded:
      self._grow_daylight(T)
    with torch.cuda.device(self.device):
      self._check(self._lib.crafter_step_n(
          self._handle, T, C.c_void_p(actions.data_ptr()), C.c_void_p(o.data_ptr()) if o is not None else None,
          C.c_void_p(r.data_ptr()), C.c_void_p(d.data_ptr()), self._stream()))
    self._keep = (actions, o, r, d)
    return o, r, d

  @staticmethod
  def _check_out(t, like):
    if not (t.is_cuda and t.dtype == like.dtype and t.shape == like.shape and t.is_contiguous() and t.data_ptr() % 16 == 0):
      raise ValueError(f'out tensor must be a contiguous, 16-byte aligned {like.dtype} device tensor of shape {tuple(like.shape)}')

  def _render_handle(self, size):
    """A second handle with another frame size over the SAME state buffers: Env.render(size)
    (env.py:120-130; the reference's VideoRecorder asks for 512x512, recorder.py:92)."""
    size = tuple(int(v) for v in (size if hasattr(size, '__len__') else (size, size)))
    if size == (self.cfg.size_w, self.cfg.size_h):
      return self._native
    if size not in self._aux:
      k = self._ctor
      cfg, geo = tables.make_config(self.num_envs, self.rules, k['area'], k['view'], size, k['reward'], k['length'],
                                    max_objects=k['max_objects'], auto_reset=False, want_semantic=False,
                                    render_obs=True, n_daylight=self.cfg.n_daylight)
      cfg.gen_period = -1
      h = _Handle(self._lib, cfg, tables.HostTables(self.rules, self._textures, cfg, geo), self.device)
      h.bind(self._st)
      self._aux[size] = h
    return self._aux[size]

  def render(self, size=None, mask=None, out=None):
    """Env.render(size) (env.py:120-130) for all / masked envs; like the reference it re-draws the frame and
    consumes the night noise from each env's RNG again.  Returns uint8 [N, size[1], size[0], 3]."""
    h = self._native if size is None else self._render_handle(size)
    shape = (self.num_envs, h.cfg.size_h, h.cfg.size_w, 3)
    out = torch.zeros(shape, dtype=torch.uint8, device=self.device) if out is None else out
    mptr = None
    if mask is not None:
      mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
      mptr = C.c_void_p(mask.data_ptr())
    with torch.cuda.device(self.device):
      h.check(self._lib.crafter_render(h.ptr, mptr, C.c_void_p(out.data_ptr()), self._stream()))
    self._keep = mask
    return out

  def info(self):
    """Device-tensor views of what the reference puts into ``info`` (env.py:108-115)."""
    o, r = self._off, self._rec_i32
    ni, na = len(self.item_names), len(self.achievement_names)
    objs = self.state['objs']
    out = {
        'inventory': r[:, o['inv']: o['inv'] + ni],
        'achievements': r[:, o['ach']: o['ach'] + na],
        'discount': 1.0 - r[:, o['dead']].to(torch.float32),
        'player_pos': objs[:, 1, 4:8].view(torch.int16).to(torch.int32),
        'step': r[:, o['step']],
        'episode': r[:, o['episode']],
    }
    if self.cfg.want_semantic:
      out['semantic'] = self.state['semantic'].view(self.num_envs, self.cfg.W, self.cfg.H)
    return out

  # ------------------------------------------------------------------ measurement
  def enable_phase_stamps(self, enable=True):
    """Debug aid: the step kernel writes shader-clock stamps of its phases into a [N, 8] buffer."""
    self._prof = torch.zeros((self.num_envs, 16), dtype=torch.int64, device=self.device) if enable else None
    self._st.prof = self._prof.data_ptr() if enable else None
    self._native.bind(self._st)
    return self._prof

  def dispatch_order(self):
    """Diagnostics: the order in which the next step() dispatches the envs (numpy int32 [N]; slow envs -- night frame or
    balance step next -- first), or None when this batch keeps none."""
    out = np.empty(self.num_envs, np.int32)
    rc = self._lib.crafter_debug_dispatch_order(self._handle, out.ctypes.data_as(C.c_void_p))
    if rc == 2:
      return None
    self._check(rc)
    return out

  def set_timing(self, enable):
    """Attach HIP start / stop events to the kernels of every following step() (their own execution time)."""
    self._check(self._lib.crafter_set_timing(self._handle, int(bool(enable))))
    self._timing_on = bool(enable)

  def get_timing(self):
    """(sum step-kernel ms, sum reset-kernel ms, launches) since the last call; synchronises."""
    a, b, n = C.c_double(), C.c_double(), C.c_int32()
    self._check(self._lib.crafter_get_timing(self._handle, C.byref(a), C.byref(b), C.byref(n)))
    return a.value, b.value, n.value

  # ------------------------------------------------------------------ host read-back (sync)
  def records(self):
    """Structured numpy copy of every env's scalar record (synchronises)."""
    return state.rec_view(self.state['rec'].cpu().numpy())

  def pool_status(self, stats=True):
    """World pool diagnostics: {'state': 'off' | 'running' | 'failed', 'launched', 'trusted', 'error'} and, with
    stats=True (a small device -> host copy: synchronises), 'adopted' / 'regenerated_inline'."""
    a, b = C.c_uint32(), C.c_uint32()
    rc = self._lib.crafter_pool_status(self._handle, C.byref(a), C.byref(b))
    err = self._lib.crafter_pool_error(self._handle)
    out = {'state': {0: 'off', 1: 'running', 2: 'failed'}.get(rc, 'unknown'), 'launched': a.value,
           'trusted': b.value, 'error': err.decode() if err else ''}
    if stats and 'pool_stats' in self.state:   # synchronises (small device -> host copy)
      s = self.state['pool_stats'].cpu().tolist()
      out['adopted'], out['regenerated_inline'] = int(s[0]), int(s[1])
    return out

  def check_errors(self):
    """Raises if any env hit a sticky device-side error (object-table overflow, bad action...).  (With length=None the
    daylight table grows ahead of the longest episode, _grow_daylight; 'step beyond the daylight table' can only come
    from a caller of the C boundary who never extends it.)  A world pool that was switched off by a HIP error only warns: stepping
    stays correct (finished envs regenerate inline), it is slower.

    Also where the slot table GROWS (the reference's object list has no bound, engine.py:50-58; here `max_objects` slots):
    when an env's table is three quarters full the capacity doubles -- new buffers, a new native handle over them
    (_grow_objects) -- long before an object could be refused, provided the caller comes by here at least every few
    steps (`crafter_amd.Env` does after every step).  A caller who never does keeps the sticky ST_OBJ_OVERFLOW."""
    ps = self.pool_status(stats=False)   # host-side state only: no extra device -> host copy per call
    if ps['state'] == 'failed' and not self._pool_warned:
      import warnings
      warnings.warn(ps['error'], RuntimeWarning)
      self._pool_warned = True
    status = self._rec_i32[:, self._off['status']]
    if self.grow_objects and 4 * int(self._rec_i32[:, self._off['nobj']].max()) > 3 * self.cfg.max_objects:
      self._grow_objects(2 * self.cfg.max_objects)
      status = self._rec_i32[:, self._off['status']]
    bad = torch.nonzero(status).flatten()
    if bad.numel():
      i = int(bad[0])
      bits = int(status[i]) & 0xFFFFFFFF
      names = [v for k, v in abi.STATUS_NAMES.items() if bits & k]
      raise CrafterDeviceError(f'env {i}: {", ".join(names) or hex(bits)} ({bad.numel()} envs affected)')

  def snapshot(self, i):
    """Canonical host-side dump of env i, comparable with the oracle's snapshot() (tests)."""
    cfg = self.cfg
    r = state.rec_view(self.state['rec'][i:i + 1].cpu().numpy())[0]
    objs = state.objs_view(self.state['objs'][i:i + 1].cpu().numpy())[0]
    R = self.tables.rules
    mat = self.state['mat'][i].cpu().numpy().reshape(cfg.W, cfg.H)
    if self.slot_map_derived:
      occupied = state.occupied_cells(objs, r['nobj'], cfg)
    else:
      occupied = self.state['objmap'][i].cpu().numpy().view(np.uint16).reshape(cfg.W, cfg.H) > 0
    order = self.state['chunk_order'][i].cpu().numpy().view(np.uint16)
    return {
        'step': int(r['step']), 'episode': int(r['episode']), 'mat': mat, 'occupied': occupied,
        'objects': state.live_objects(objs, r['nobj'], r['inv'][R.item_health]),
        'inventory': [int(v) for v in r['inv'][:R.n_items]],
        'achievements': [int(v) for v in r['ach'][:R.n_achievements]],
        'sleeping': bool(r['sleeping']),
        'hunger2': int(r['hunger2']), 'thirst2': int(r['thirst2']), 'fatigue2': int(r['fatigue2']),
        'recover2': int(r['recover2']), 'player_last_health': int(r['player_last_health']),
        'chunk_order': state.chunk_keys(order, r['nchunks_seen'], cfg),
        'mt_key': self.state['mt'][i].cpu().numpy().view(np.uint32), 'mt_pos': int(r['mt_pos']),
    }
