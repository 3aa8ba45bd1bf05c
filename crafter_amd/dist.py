"""Multi-GPU sharding of a batch of environments (one process per GPU, torch.distributed).

Environments are independent, so the batch shards by env index with no data-path collective:
rank r owns global envs [r * n, (r + 1) * n) and seeds them by GLOBAL index, so the union of all
ranks is the same set of worlds whatever the world size.  The one exchange the path has is the
learner-side gather of (obs, reward, done) each step -- ONE all-gather of a packed per-rank byte
record over RCCL / xGMI (gloo in the CPU tests), double-buffered so that it overlaps the next step
(``StepExchange``; ``bench.py --gpus N`` and ``tests/test_dist_gloo.py`` drive this same class), or, by
``mode``, a gather to the learner rank only / an exchange of reward and done alone.
"""
import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
  """Contiguous, balanced [lo, hi) slice of range(total_envs) owned by `rank`."""
  base, extra = divmod(int(total_envs), int(world_size))
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


def shard_seeds(base_seed, total_envs, rank, world_size):
  lo, hi = shard_range(total_envs, rank, world_size)
  return [base_seed + i for i in range(lo, hi)]


def shard_actions(global_actions, rank, world_size):
  """Slice of a [.., total_envs] action tensor owned by `rank`."""
  lo, hi = shard_range(global_actions.shape[-1], rank, world_size)
  return global_actions[..., lo:hi]


def _align(v, a=256):
  return (v + a - 1) // a * a


class _Slot:
  """One of the exchange's buffers: `steps` consecutive records of this rank (the step kernel writes straight into their
  views) and the gathered records of all ranks."""

  def __init__(self, n, world, obs_shape, device, steps=1):
    obs_bytes = n * int(torch.tensor(obs_shape).prod()) if obs_shape is not None else 0
    self.off_reward = _align(obs_bytes)
    self.off_done = self.off_reward + _align(4 * n)
    self.record_bytes = self.off_done + _align(n)
    self.steps = int(steps)
    self.local = torch.zeros(self.steps * self.record_bytes, dtype=torch.uint8, device=device)
    self.gathered = torch.zeros((world, self.steps * self.record_bytes), dtype=torch.uint8, device=device)
    self.n, self.obs_shape, self.obs_bytes = n, obs_shape, obs_bytes
    self.work = None
    self.step = -1        # first step of the block the slot holds
    self.cursor = -1      # the step begun last
    self.launched = False

  def _views(self, rec, lead):
    n = self.n
    obs = None
    if self.obs_shape is not None:
      obs = rec[..., :self.obs_bytes].reshape(lead + (n,) + tuple(self.obs_shape))
    reward = rec[..., self.off_reward:self.off_reward + 4 * n].view(torch.float32).reshape(lead + (n,))
    done = rec[..., self.off_done:self.off_done + n].reshape(lead + (n,))
    return obs, reward, done

  def record(self, buf, k):
    """the k-th record of a block buffer (last dimension = steps * record_bytes)"""
    return buf[..., k * self.record_bytes:(k + 1) * self.record_bytes]

  def outputs(self, k=0):
    """(obs u8[n, ...] or None, reward f32[n], done u8[n]) views of this rank's k-th record: hand them to
    ``BatchedEnv.step(actions, out=...)`` so the kernel's outputs ARE the send buffer (no staging copy)."""
    return self._views(self.record(self.local, k), ())


class StepExchange:
  """Per-step exchange of every rank's (obs, reward, done), overlapped with the following step.

  Protocol per step t (same code for RCCL on GPUs and gloo on CPU tensors)::

      slot = ex.begin(t)                       # slot t % depth; first waits until its previous exchange has finished
      env.step(actions, out=ex.outputs(slot))  # or copy results into ex.outputs(slot)
      ex.launch(slot)                          # async collective over the packed record: ONE per step
      ...
      obs, reward, done = ex.result(t)         # [world, n, ...] views, valid until slot t % depth is begun again

  mode (what the learner side needs decides what crosses xGMI; records are 6.29 MB per rank at 512 envs):
    'allgather'  every rank receives every rank's record (the north star's "RCCL gather of obs/reward/done"): each GPU
                 takes in (world - 1) records per step -- 44 MB at 8 x 512 envs; 41 us direct over seven 153 GB/s links,
                 ~290 us if the library runs it as a ring.
    'gather'     only rank `dst` (the learner) receives them (dist.gather = grouped point-to-point sends: every peer's
                 record travels its own link to dst, ~41 us, and the seven other GPUs receive nothing: no 44 MB of HBM
                 writes next to their step kernels).  result() returns the gathered views on dst and None elsewhere.
    'scalars'    reward / done are all-gathered (2.5 KB per rank: latency only), observations stay on the rank that
                 rendered them -- for a learner that is itself sharded by env index.  result() returns obs = this
                 rank's own frames [n, ...] (the record the kernels wrote), reward / done [world, n].

  The collective is issued with ``async_op=True``: with the NCCL (= RCCL) backend it runs on the backend's own
  stream, ordered after the work already enqueued on the current stream (the step that filled the record), and
  ``Work.wait()`` orders the current stream behind it without blocking the host; with gloo ``wait()`` blocks the
  calling thread.  Equal n on every rank (the env count must divide by the world size)."""

  MODES = ('allgather', 'gather', 'scalars')

  def __init__(self, n, obs_shape=(64, 64, 3), device='cpu', group=None, depth=2, gather_obs=True, mode='allgather', dst=0, steps=1):
    """steps: records per collective.  1 = one exchange per step (the north star's wording).  K > 1: the records of K
    consecutive steps fill one block and travel together -- fewer, larger collectives: the host pays the collective's
    enqueue (~25 us through torch.distributed: more than a 512-env step takes on the GPU, tools/host_overhead_dist.py) once
    per K steps, and the wire sees K x 6.3 MB messages instead of K small ones; result(t) is then available once t's block
    is complete (a learner that consumes trajectories in chunks of K steps anyway loses nothing)."""
    if mode not in self.MODES:
      raise ValueError(f'mode must be one of {self.MODES}')
    self.steps = int(steps)
    if self.steps < 1:
      raise ValueError('steps must be >= 1')
    self._blk, self._pos = 0, 0   # block being filled, records in it so far
    self.group = group
    self.world = dist.get_world_size(group)
    self.rank = dist.get_rank(group)
    self.n = int(n)
    self.mode, self.dst = mode, int(dst)
    if not 0 <= self.dst < self.world:
      raise ValueError(f'dst {dst} is not a rank of the group')
    self.obs_shape = tuple(obs_shape)
    wire_obs = gather_obs and mode != 'scalars'   # do observations travel?
    self.slots = [_Slot(self.n, self.world, self.obs_shape if wire_obs else None, device, self.steps) for _ in range(depth)]
    if mode == 'scalars' and gather_obs:   # the frames still need a home the kernels can write: a plain per-slot buffer
      for s in self.slots:
        s.own_obs = torch.zeros((self.steps, self.n) + self.obs_shape, dtype=torch.uint8, device=device)
    rec = self.slots[0].record_bytes
    self.receives = mode != 'gather' or self.rank == self.dst
    self.bytes_per_step = rec * self.world            # receive buffer of a rank that receives, per step
    # bytes that cross the interconnect per step, summed over ranks / taken in by the busiest rank
    self.wire_bytes_per_step = rec * (self.world - 1) * (self.world if mode != 'gather' else 1)
    self.recv_bytes_per_step = rec * (self.world - 1) if self.receives else 0

  def scatter_actions(self, actions=None, src=0, how='broadcast'):
    """The way back of a closed loop ("actions return as a 16 KB broadcast / scatter"): the learner rank `src`
    has chosen an action for EVERY env -- int32 [world * n], global env index = rank * n + i, on this exchange's device --
    and every rank gets the n actions of its own envs as an int32 [n] tensor to hand to ``BatchedEnv.step``.  `actions`
    is only read on `src`.  how='broadcast': one broadcast of the whole vector (16 KB at 4096 envs: latency only), each
    rank slices; how='scatter': ``dist.scatter`` of the per-rank slices (n * 4 bytes each).  The call is enqueued like any
    collective (RCCL: ordered behind the work on the current stream, the current stream ordered behind it; gloo: blocks)
    and the returned tensor stays valid until the next call."""
    if how not in ('broadcast', 'scatter'):
      raise ValueError("how must be 'broadcast' or 'scatter'")
    dev = self.slots[0].local.device
    if not hasattr(self, '_act_all'):
      self._act_all = torch.zeros(self.world * self.n, dtype=torch.int32, device=dev)
      self._act_own = torch.zeros(self.n, dtype=torch.int32, device=dev)
    src = int(src)
    if not 0 <= src < self.world:
      raise ValueError(f'src {src} is not a rank of the group')
    gsrc = src if self.group is None else dist.get_global_rank(self.group, src)
    if self.rank == src:
      if actions is None or tuple(actions.shape) != (self.world * self.n,):
        raise ValueError(f'the learner rank passes the actions of all {self.world * self.n} envs')
      self._act_all.copy_(actions.to(torch.int32))
    if how == 'broadcast':
      dist.broadcast(self._act_all, src=gsrc, group=self.group)
      return self._act_all[self.rank * self.n:(self.rank + 1) * self.n]
    parts = list(self._act_all.view(self.world, self.n).unbind(0)) if self.rank == src else None
    dist.scatter(self._act_own, scatter_list=parts, src=gsrc, group=self.group)
    return self._act_own

  def begin(self, t):
    """The slot step t's record goes to: the block being filled (steps records per block, blocks rotate through the
    slots).  At the first step of a block the call first waits until the slot's previous exchange has finished."""
    slot = self.slots[self._blk % len(self.slots)]
    if self._pos == 0:
      if slot.work is not None:   # the exchange that last used this slot must have consumed `local` / produced `gathered`
        slot.work.wait()
        slot.work = None
      slot.step = t
      slot.launched = False
    elif t != slot.step + self._pos:
      raise RuntimeError(f'steps must be begun in order (expected {slot.step + self._pos}, got {t})')
    slot.cursor = t
    self._pos += 1
    return slot

  def outputs(self, slot):
    """(obs or None, reward, done) tensors for ``BatchedEnv.step(out=...)``: the record of the step begun last in the
    slot's send block (and, in 'scalars' mode, the slot's own frame buffer)."""
    k = slot.cursor - slot.step
    o, r, d = slot.outputs(k)
    return (slot.own_obs[k] if (o is None and hasattr(slot, 'own_obs')) else o), r, d

  def launch(self, slot, flush=False):
    """Call after every step: issues the block's collective when its last step has been written (or now, with flush:
    the block is then closed short and the next step starts a new one)."""
    if slot.launched or not (flush or self._pos == self.steps):
      return
    slot.launched = True
    self._blk += 1
    self._pos = 0
    if self.mode == 'gather':
      parts = list(slot.gathered.unbind(0)) if self.rank == self.dst else None
      slot.work = dist.gather(slot.local, gather_list=parts, dst=self._global_dst(), group=self.group, async_op=True)
    else:
      slot.work = dist.all_gather_into_tensor(slot.gathered.view(-1), slot.local, group=self.group, async_op=True)

  def _global_dst(self):
    return self.dst if self.group is None else dist.get_global_rank(self.group, self.dst)

  def result(self, t):
    """(obs, reward, done) of step t as zero-copy views of the receive buffer, global env index = rank * n + i:
    'allgather' u8[world, n, ...] / f32[world, n] / u8[world, n]; 'gather' the same on dst, None on the other ranks;
    'scalars' obs = this rank's own u8[n, ...] (or None without frames), reward / done [world, n]."""
    slot = next((sl for sl in self.slots if sl.step >= 0 and sl.step <= t <= sl.cursor), None)
    if slot is None:
      raise RuntimeError(f'step {t} is no longer buffered (slots hold ' + ', '.join(f'{sl.step}..{sl.cursor}' for sl in self.slots) + ')')
    k = t - slot.step
    if not slot.launched:
      raise RuntimeError(f"step {t}'s block has not been exchanged yet (steps={self.steps}: complete it, or launch(slot, flush=True))")
    if slot.work is not None:
      slot.work.wait()
      slot.work = None
    if not self.receives:
      return None
    obs, reward, done = slot._views(slot.record(slot.gathered, k), (self.world,))
    if self.mode == 'scalars':
      obs = slot.own_obs[k] if hasattr(slot, 'own_obs') else None
    return obs, reward, done

  def finish(self):
    """Exchanges a block still being filled (every rank has begun the same steps) and waits for every exchange in flight
    (end of a run)."""
    if self._pos:
      self.launch(self.slots[self._blk % len(self.slots)], flush=True)
    for slot in self.slots:
      if slot.work is not None:
        slot.work.wait()
        slot.work = None


class NativeStepExchange:
  """The per-step all-gather of StepExchange(mode='allgather', steps=1) with the whole loop body -- step kernels writing into
  the send record, the RCCL all-gather of the record behind them -- enqueued by ONE call into libcrafter_hip.so
  (crafter_step_exchange, include/crafter_hip.h) instead of four trips through Python and torch.distributed: at one GPU's
  share of BASELINE configs[2] (512 envs) those cost the host 49-72 us per step against 26-31 us of GPU (round 4).

      ex = NativeStepExchange(env)             # collective: every rank; the communicator id travels through torch.distributed
      for t in ...:
        ex.step(t, actions)                    # enqueue: step kernels -> record of slot t % depth -> ncclAllGather on the exchange's stream
        obs, reward, done = ex.result(t - 1)   # [world, n, ...] views; the current stream waits for that slot's gather

  The communicator is the library's own (ncclCommInitRank from the id rank 0 draws), next to torch.distributed's; devices
  only (RCCL): the gloo plumbing path stays StepExchange."""

  def __init__(self, env, group=None, depth=2, gather_obs=True):
    import ctypes as C
    from . import lib as _libmod
    self.env, self.depth = env, int(depth)
    self._C, self._lib = C, _libmod.load()
    self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
    self.n = env.num_envs
    dev = env.device
    self.slots = [_Slot(self.n, self.world, tuple(env.obs.shape[1:]) if gather_obs else None, dev) for _ in range(self.depth)]
    self.with_obs = bool(gather_obs)
    idbuf = (C.c_uint8 * 128)()
    if self.rank == 0 and self._lib.crafter_exchange_unique_id(idbuf):
      raise RuntimeError(self._lib.crafter_exchange_error(None).decode())
    box = [bytes(idbuf)]
    src = 0 if group is None else dist.get_global_rank(group, 0)
    self._x = C.c_void_p()
    with torch.cuda.device(dev):   # (the object broadcast of the nccl backend stages through the CURRENT device: this rank's, not device 0 -- ADVICE r5)
      dist.broadcast_object_list(box, src=src, group=group)
      idbuf = (C.c_uint8 * 128).from_buffer_copy(box[0])
      if self._lib.crafter_exchange_create(idbuf, self.rank, self.world, self.depth, C.byref(self._x)):
        raise RuntimeError(self._lib.crafter_exchange_error(None).decode())
    self._held = {}   # slot index -> step it holds
    rec = self.slots[0].record_bytes
    self.wire_bytes_per_step = rec * (self.world - 1) * self.world
    # everything a step's call needs, made once: the host's share of a step is what this class is about
    self._views = [s._views(s.gathered, (self.world,)) for s in self.slots]
    self._args = [(C.c_void_p(s.local.data_ptr()), C.c_void_p(s.gathered.data_ptr()), s.record_bytes, s.off_reward, s.off_done, int(self.with_obs))
                  for s in self.slots]
    self._dev_index = env.device.index

  def step(self, t, actions):
    """Enqueues step t of this rank's envs and the exchange of its record.  actions: int32 [n] on the device."""
    env, k = self.env, t % self.depth
    if not (torch.is_tensor(actions) and actions.dtype == torch.int32 and actions.is_cuda and actions.is_contiguous()):
      actions = torch.as_tensor(actions, device=env.device).to(torch.int32).contiguous()
    if env._unbounded:
      env._grow_daylight(1)
    if torch.cuda.current_device() != self._dev_index:
      torch.cuda.set_device(self._dev_index)
    send, recv, rec, off_r, off_d, with_obs = self._args[k]
    if self._lib.crafter_step_exchange(env._handle, self._x, k, actions.data_ptr(), send, recv, rec, off_r, off_d, with_obs,
                                       torch.cuda.current_stream().cuda_stream):
      raise RuntimeError(self._lib.crafter_exchange_error(self._x).decode())
    env._keep = actions
    self._held[k] = t

  def result(self, t):
    """(obs u8[world, n, ...] or None, reward f32[world, n], done u8[world, n]) of step t: views of the slot's receive buffer,
    valid until step t + depth is enqueued; the current stream is ordered behind the slot's gather (no host wait)."""
    k = t % self.depth
    if self._held.get(k) != t:
      raise RuntimeError(f'step {t} is no longer buffered')
    if self._lib.crafter_exchange_wait(self._x, k, torch.cuda.current_stream().cuda_stream):
      raise RuntimeError(self._lib.crafter_exchange_error(self._x).decode())
    return self._views[k]

  def finish(self):
    for k in list(self._held):
      with torch.cuda.device(self.env.device):
        self._lib.crafter_exchange_wait(self._x, k, self.env._stream())

  def close(self):
    if self._x:
      torch.cuda.synchronize(self.env.device)
      self._lib.crafter_exchange_destroy(self._x)
      self._x = self._C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
