"""Multi-GPU sharding of a batch of environments (one process per GPU, torch.distributed).

Environments are independent, so the batch shards by env index with no data-path collective:
rank r owns global envs [r * n, (r + 1) * n) and seeds them by GLOBAL index, so the union of all
ranks is the same set of worlds whatever the world size.  The one exchange the path has is the
learner-side gather of (reward, done[, obs]) each step -- an RCCL all-gather over xGMI (gloo in the
CPU tests), issued on a side stream so it overlaps the next step.
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


class StepGather:
  """All-gathers per-rank (reward f32[n], done u8[n]) and optionally obs u8[n,H,W,3] into
  [world, n, ...] buffers.  Requires equal n on every rank (weak scaling)."""

  def __init__(self, n, obs_shape=None, device='cpu', group=None):
    self.group = group
    self.world = dist.get_world_size(group)
    self.packed = torch.zeros((self.world, n, 2), dtype=torch.float32, device=device)
    self.obs = None if obs_shape is None else torch.zeros((self.world, n) + tuple(obs_shape), dtype=torch.uint8,
                                                            device=device)

  def __call__(self, reward, done, obs=None):
    mine = torch.stack([reward.to(torch.float32), done.to(torch.float32)], dim=1).contiguous()
    dist.all_gather_into_tensor(self.packed.view(-1, 2), mine, group=self.group)
    if self.obs is not None and obs is not None:
      dist.all_gather_into_tensor(self.obs.view((-1,) + tuple(self.obs.shape[2:])), obs.contiguous(), group=self.group)
    return self.packed[..., 0], self.packed[..., 1].to(torch.uint8), self.obs
