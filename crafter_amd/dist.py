"""Multi-GPU sharding of a batch of environments (one process per GPU, torch.distributed).

Environments are independent, so the batch shards by env index with no data-path collective:
rank r owns global envs [r * n, (r + 1) * n) and seeds them by GLOBAL index, so the union of all
ranks is the same set of worlds whatever the world size.  The one exchange the path has is the
learner-side gather of (obs, reward, done) each step -- ONE all-gather of a packed per-rank byte
record over RCCL / xGMI (gloo in the CPU tests), double-buffered so that it overlaps the next step
(``StepExchange``; ``bench.py --gpus N`` and ``tests/test_dist_gloo.py`` drive this same class).
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
  """One of the exchange's buffers: this rank's record (the step kernel writes straight into its views) and the
  gathered records of all ranks."""

  def __init__(self, n, world, obs_shape, device):
    obs_bytes = n * int(torch.tensor(obs_shape).prod()) if obs_shape is not None else 0
    self.off_reward = _align(obs_bytes)
    self.off_done = self.off_reward + _align(4 * n)
    self.record_bytes = self.off_done + _align(n)
    self.local = torch.zeros(self.record_bytes, dtype=torch.uint8, device=device)
    self.gathered = torch.zeros((world, self.record_bytes), dtype=torch.uint8, device=device)
    self.n, self.obs_shape, self.obs_bytes = n, obs_shape, obs_bytes
    self.work = None
    self.step = -1

  def _views(self, rec, lead):
    n = self.n
    obs = None
    if self.obs_shape is not None:
      obs = rec[..., :self.obs_bytes].reshape(lead + (n,) + tuple(self.obs_shape))
    reward = rec[..., self.off_reward:self.off_reward + 4 * n].view(torch.float32).reshape(lead + (n,))
    done = rec[..., self.off_done:self.off_done + n].reshape(lead + (n,))
    return obs, reward, done

  def outputs(self):
    """(obs u8[n, ...] or None, reward f32[n], done u8[n]) views of this rank's record: hand them to
    ``BatchedEnv.step(actions, out=...)`` so the kernel's outputs ARE the send buffer (no staging copy)."""
    return self._views(self.local, ())


class StepExchange:
  """Per-step all-gather of every rank's (obs, reward, done), overlapped with the following step.

  Protocol per step t (same code for RCCL on GPUs and gloo on CPU tensors)::

      slot = ex.begin(t)                       # slot t % depth; first waits until its previous gather has finished
      env.step(actions, out=slot.outputs())    # or copy results into slot.outputs()
      ex.launch(slot)                          # async all-gather of the packed record: ONE collective per step
      ...
      obs, reward, done = ex.result(t)         # [world, n, ...] views, valid until slot t % depth is begun again

  The collective is issued with ``async_op=True``: with the NCCL (= RCCL) backend it runs on the backend's own
  stream, ordered after the work already enqueued on the current stream (the step that filled the record), and
  ``Work.wait()`` orders the current stream behind it without blocking the host; with gloo ``wait()`` blocks the
  calling thread.  Equal n on every rank (the env count must divide by the world size).  xGMI is point-to-point:
  at 512 envs per rank the record is 6.3 MB, 50 MB gathered per rank and step."""

  def __init__(self, n, obs_shape=(64, 64, 3), device='cpu', group=None, depth=2, gather_obs=True):
    self.group = group
    self.world = dist.get_world_size(group)
    self.rank = dist.get_rank(group)
    self.n = int(n)
    self.slots = [_Slot(self.n, self.world, tuple(obs_shape) if gather_obs else None, device) for _ in range(depth)]
    self.bytes_per_step = self.slots[0].record_bytes * self.world

  def begin(self, t):
    slot = self.slots[t % len(self.slots)]
    if slot.work is not None:   # the gather that last used this slot must have consumed `local` / produced `gathered`
      slot.work.wait()
      slot.work = None
    slot.step = t
    return slot

  def launch(self, slot):
    slot.work = dist.all_gather_into_tensor(slot.gathered.view(-1), slot.local, group=self.group, async_op=True)

  def result(self, t):
    """Gathered (obs u8[world, n, ...] or None, reward f32[world, n], done u8[world, n]) of step t: zero-copy views of
    the receive buffer; global env index = rank * n + i."""
    slot = self.slots[t % len(self.slots)]
    if slot.step != t:
      raise RuntimeError(f'step {t} is no longer buffered (slot holds step {slot.step})')
    if slot.work is not None:
      slot.work.wait()
      slot.work = None
    return slot._views(slot.gathered, (self.world,))

  def finish(self):
    """Waits for every gather still in flight (end of a run)."""
    for slot in self.slots:
      if slot.work is not None:
        slot.work.wait()
        slot.work = None
