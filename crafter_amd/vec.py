"""A vector-env view of ``BatchedEnv`` with the method surface Stable-Baselines3's ``VecEnv`` and similar
trainers expect (numpy in / numpy out, ``step_async`` + ``step_wait``, auto-reset with
``infos[i]['terminal_observation']``), SURVEY.md section 8f row 3.

No dependency on SB3 / gymnasium (neither is required by the reference, and neither is in this image): the
class only follows the protocol, so it can be handed to anything that duck-types a VecEnv.

The finished episode's last frame has to be reported, so the wrapped env runs with ``auto_reset=False``
and finished envs are reset here with one masked ``reset`` (the in-kernel auto-reset of ``BatchedEnv``
draws the next episode's first frame over it).  ``info`` follows the reference (env.py:108-115):
inventory / achievements dicts, discount, player_pos, reward (+ semantic when asked for).
"""
import numpy as np
import torch

from .batched import BatchedEnv
from .env import BoxSpace, DiscreteSpace


class VecEnvView:

  def __init__(self, num_envs, area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000, seed=None,
               seeds=None, device='cuda', semantic=False, **kwargs):
    self._batch = BatchedEnv(num_envs, area=area, view=view, size=size, reward=reward, length=length, seed=seed,
                             seeds=seeds, device=device, auto_reset=False, semantic=semantic, **kwargs)
    self.num_envs = int(num_envs)
    b = self._batch
    self.observation_space = BoxSpace(0, 255, tuple(b.observation_shape), np.uint8)
    self.action_space = DiscreteSpace(b.num_actions)
    self.action_names = list(b.action_names)
    self.reward_range = None
    self.metadata = None
    self._semantic = bool(semantic)
    self._length = length
    self._pending = None

  @property
  def batch(self):
    """The underlying ``BatchedEnv`` (device tensors)."""
    return self._batch

  # ------------------------------------------------------------------ VecEnv protocol
  def reset(self):
    return self._batch.reset().cpu().numpy()

  def step_async(self, actions):
    self._pending = torch.as_tensor(np.asarray(actions), dtype=torch.int32).to(self._batch.device)

  def step_wait(self):
    b = self._batch
    obs, reward, done, info = b.step(self._pending)
    self._pending = None
    b.check_errors()
    rec = b.records()
    obs_h, rew_h = obs.cpu().numpy().copy(), reward.cpu().numpy().copy()
    done_h = done.cpu().numpy().astype(bool)
    pos = info['player_pos'].cpu().numpy().astype(np.int64)
    sem = info['semantic'].cpu().numpy() if self._semantic else None
    infos = []
    for i in range(self.num_envs):
      r = rec[i]
      dead = bool(r['dead'])
      d = {
          'inventory': {n: int(r['inv'][k]) for k, n in enumerate(b.item_names)},
          'achievements': {n: int(r['ach'][k]) for k, n in enumerate(b.achievement_names)},
          'discount': 1 - float(dead),
          'player_pos': pos[i],
          'reward': int(r['dhealth']) / 10 + (1.0 if int(r['new_unlocked']) else 0.0),
      }
      if sem is not None:
        d['semantic'] = sem[i].copy()
      if done_h[i]:
        d['terminal_observation'] = obs_h[i].copy()
        d['TimeLimit.truncated'] = not dead   # the episode ran into `length` (env.py:106-107)
      infos.append(d)
    if done_h.any():
      fresh = b.reset(torch.from_numpy(done_h.astype(np.uint8)).to(b.device)).cpu().numpy()
      obs_h[done_h] = fresh[done_h]
    return obs_h, rew_h, done_h, infos

  def step(self, actions):
    self.step_async(actions)
    return self.step_wait()

  def close(self):
    pass

  def seed(self, seed=None):
    """Seeds are fixed at construction (``seed`` / ``seeds``): one RandomState per env, env.py:74."""
    return [None] * self.num_envs

  def render(self, size=None, mode='rgb_array'):
    return self._batch.render(size).cpu().numpy()

  def get_images(self):
    return list(self.render())

  def _indices(self, indices):
    if indices is None:
      return list(range(self.num_envs))
    return [int(indices)] if np.isscalar(indices) else [int(i) for i in indices]

  def get_attr(self, attr_name, indices=None):
    return [getattr(self, attr_name) for _ in self._indices(indices)]

  def set_attr(self, attr_name, value, indices=None):
    raise AttributeError('the batched env has no per-env Python attributes to set')

  def env_method(self, method_name, *args, indices=None, **kwargs):
    raise AttributeError('the batched env has no per-env Python objects to call')

  def env_is_wrapped(self, wrapper_class, indices=None):
    return [False for _ in self._indices(indices)]
