"""``crafter_amd.Env`` -- drop-in for the reference ``crafter.Env`` (env.py:25-133) on one MI355X.

Same constructor, same ``reset() -> obs``, ``step(a) -> (obs, reward, done, info)``,
``render(size=None)``, ``observation_space`` / ``action_space`` / ``action_names``, the same
old-gym conventions (no auto-reset, obs only from reset) and the private attributes in-repo
callers touch (``_step``, ``_world.count``, ``_player.achievements/.sleeping``; run_random.py:32-42,
run_gui.py:70,105-122).  It is ``BatchedEnv(1)`` plus host copies: every call synchronises, so use
``BatchedEnv`` for throughput.
"""
import collections

import numpy as np
import torch

from . import batched, tables

try:  # gym is optional in the reference too (env.py:11-22)
  import gym
  DiscreteSpace = gym.spaces.Discrete
  BoxSpace = gym.spaces.Box
  BaseClass = gym.Env
except ImportError:
  DiscreteSpace = collections.namedtuple('DiscreteSpace', 'n')
  BoxSpace = collections.namedtuple('BoxSpace', 'low, high, shape, dtype')
  BaseClass = object


class _WorldView:
  """What callers read through ``env._world`` (run_random.py:32-34, run_terrain.py:23)."""

  def __init__(self, env):
    self._env = env
    self.area = env._area

  def count(self, material):
    ids = {name: i + 1 for i, name in enumerate(self._env._batch.rules['materials'])}
    mat = self._env._batch.state['mat'][0]
    return int((mat == ids[material]).sum().item())

  @property
  def daylight(self):
    return float(self._env._batch.tables.daylight[self._env._step or 0])


class _PlayerView:
  """What callers read through ``env._player`` (run_gui.py:105,116; recorder via info)."""

  def __init__(self, env):
    self._env = env

  def _rec(self):
    return self._env._batch.records()[0]

  @property
  def inventory(self):
    r = self._rec()
    return {n: int(r['inv'][i]) for i, n in enumerate(self._env._batch.item_names)}

  @property
  def achievements(self):
    r = self._rec()
    return {n: int(r['ach'][i]) for i, n in enumerate(self._env._batch.achievement_names)}

  @property
  def sleeping(self):
    return bool(self._rec()['sleeping'])

  @property
  def health(self):
    return self.inventory['health']

  @property
  def pos(self):
    return self._env._batch.info()['player_pos'][0].cpu().numpy().astype(np.int64)


class Env(BaseClass):

  def __init__(self, area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000, seed=None,
               device='cuda', rules=None):
    view = np.array(view if hasattr(view, '__len__') else (view, view))
    size = np.array(size if hasattr(size, '__len__') else (size, size))
    seed = np.random.randint(0, 2 ** 31 - 1) if seed is None else seed  # env.py:32
    self._area = area
    self._view = view
    self._size = size
    self._reward = reward
    self._length = length
    self._seed = seed
    self._episode = 0
    self._batch = batched.BatchedEnv(
        1, area, tuple(int(v) for v in view), tuple(int(s) for s in size), reward, length, seeds=[seed],
        device=device, auto_reset=False, semantic=True, render=True, rules=rules)
    self._world = _WorldView(self)
    self._player = None
    self._step = None
    self.reward_range = None
    self.metadata = None

  @property
  def observation_space(self):
    return BoxSpace(0, 255, tuple(self._size) + (3,), np.uint8)

  @property
  def action_space(self):
    return DiscreteSpace(len(self._batch.action_names))

  @property
  def action_names(self):
    return self._batch.action_names

  def reset(self):
    self._episode += 1
    self._step = 0
    obs = self._batch.reset()
    self._player = _PlayerView(self)
    out = obs[0].cpu().numpy()
    self._batch.check_errors()
    return out

  def step(self, action):
    b = self._batch
    action = int(action)
    if not 0 <= action < len(b.action_names):
      # the reference indexes a Python list (env.py:86); negative indices are rejected here too
      raise IndexError('list index out of range')
    self._step += 1
    acts = torch.tensor([action], dtype=torch.int32, device=b.device)
    obs, _, _, _ = b.step(acts, info=False)
    obs = obs[0].cpu().numpy()
    b.check_errors()
    r = b.records()[0]
    reward = int(r['dhealth']) / 10              # env.py:97
    if int(r['new_unlocked']):
      reward += 1.0                              # env.py:102-104
    dead = bool(r['dead'])
    over = self._length and self._step >= self._length
    done = dead or over
    info = {
        'inventory': {n: int(r['inv'][i]) for i, n in enumerate(b.item_names)},
        'achievements': {n: int(r['ach'][i]) for i, n in enumerate(b.achievement_names)},
        'discount': 1 - float(dead),
        'semantic': b.state['semantic'][0].cpu().numpy().reshape(b.cfg.W, b.cfg.H),
        'player_pos': self._player.pos,
        'reward': reward,
    }
    if not self._reward:
      reward = 0.0
    return obs, reward, done, info

  def render(self, size=None):
    size = None if size is None else tuple(int(v) for v in (size if hasattr(size, '__len__') else (size, size)))
    return self._batch.render(size)[0].cpu().numpy()
