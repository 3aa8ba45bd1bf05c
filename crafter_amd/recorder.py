"""Batched counterpart of the reference's ``StatsRecorder`` (recorder.py:28-66): one ``stats.jsonl``
row per finished episode -- {'length', 'reward', 'achievement_<name>'...}, the format
``analysis/read_metrics.py`` consumes -- from a ``BatchedEnv``, auto-reset included.

The per-episode totals are kept on the device (EnvRec.ep_dhealth / ep_unlock_steps) and copied into
the ``terminal`` buffer by the step kernel the moment an episode ends, so nothing is lost when the
env is regenerated inside the same step.  ``reward`` is rebuilt as the reference accumulates it:
sum over steps of (health delta) / 10 plus 1.0 per step that unlocked something (env.py:97-104),
rounded to one decimal like recorder.py:57.
"""
import json
import pathlib

import numpy as np

from . import abi


def episode_rows(terminal_rows, achievement_names):
  """int32 [k, MAX_ACH + 4] -> list of dicts in StatsRecorder's layout."""
  rows = []
  for t in terminal_rows:
    length, dh, unlocks = int(t[abi.MAX_ACH]), int(t[abi.MAX_ACH + 1]), int(t[abi.MAX_ACH + 2])
    row = {'length': length, 'reward': round(dh / 10 + unlocks * 1.0, 1)}
    for i, name in enumerate(achievement_names):
      row[f'achievement_{name}'] = int(t[i])
    rows.append(row)
  return rows


class BatchedStatsRecorder:
  """``rec = BatchedStatsRecorder(BatchedEnv(...), directory)``; use ``rec.reset()`` / ``rec.step(actions)``
  like the env.  Everything else is forwarded (recorder.py:22-25 idiom)."""

  def __init__(self, env, directory):
    self._env = env
    self._directory = pathlib.Path(directory).expanduser()
    self._directory.mkdir(exist_ok=True, parents=True)
    self._file = (self._directory / 'stats.jsonl').open('a')
    self.episodes = 0

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return getattr(self._env, name)

  def reset(self, mask=None):
    return self._env.reset(mask)

  def step(self, actions, info=True):
    out = self._env.step(actions, info=info)
    done = out[2]
    idx = done.nonzero().flatten()
    if idx.numel():   # one small device->host copy, only on steps where an episode ended
      rows = self._env.terminal[idx].cpu().numpy()
      for row in episode_rows(rows, self._env.achievement_names):
        self._file.write(json.dumps(row) + '\n')
        self.episodes += 1
      self._file.flush()
    return out

  def close(self):
    self._file.close()


class EnvStatsRecorder:
  """``StatsRecorder`` (recorder.py:28-66) over the single-env facade ``crafter_amd.Env``: same file, same row
  layout, same reset()/step(action) surface -- the row comes from the device-side terminal record instead of
  host-side accumulation.  (The reference's own ``crafter.Recorder`` also works on top of ``crafter_amd.Env``;
  this one exists so that ``python -m crafter_amd.run_random --record`` needs no reference install.)"""

  def __init__(self, env, directory):
    self._env = env
    self._directory = pathlib.Path(directory).expanduser()
    self._directory.mkdir(exist_ok=True, parents=True)
    self._file = (self._directory / 'stats.jsonl').open('a')

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return getattr(self._env, name)

  def reset(self):
    return self._env.reset()

  def step(self, action):
    out = self._env.step(action)
    if out[2]:
      batch = self._env._batch
      row = episode_rows(batch.terminal[:1].cpu().numpy(), batch.achievement_names)[0]
      self._file.write(json.dumps(row) + '\n')
      self._file.flush()
    return out


class BatchedEpisodeRecorder:
  """Batched counterpart of the reference's ``EpisodeRecorder`` (recorder.py:100-152): one compressed
  ``.npz`` per finished episode with the same keys and the same first-row convention --
  ``image, action, reward, done, discount, semantic, player_pos, achievement_<name>, ainventory_<name>``
  (sic: the reference's 'ainventory_' spelling), step 0 holding the reset frame and zeros elsewhere.

  The finished episode's LAST frame must reach the file, so the wrapped ``BatchedEnv`` has to be built
  with ``auto_reset=False, semantic=True``; the recorder resets finished envs itself (masked reset), and
  the obs it returns for them is then the next episode's first frame, as with auto-reset.

  File names follow EpisodeName (recorder.py:155-186) plus the env index:
  ``<timestamp>-env<i>-ep<k>-ach<unlocked>-len<length>.npz`` (k counts that env's episodes: a batch finishes
  several episodes within one timestamp second).
  """

  def __init__(self, env, directory, envs=None):
    import datetime
    self._now = lambda: datetime.datetime.now().strftime('%Y%m%dT%H%M%S')
    if env.cfg.auto_reset:
      raise ValueError('BatchedEpisodeRecorder needs BatchedEnv(auto_reset=False): the last frame of an '
                       'episode is overwritten by an in-kernel auto-reset')
    if not env.cfg.want_semantic:
      raise ValueError('BatchedEpisodeRecorder needs BatchedEnv(semantic=True) (info["semantic"] is recorded)')
    self._env = env
    self._directory = pathlib.Path(directory).expanduser()
    self._directory.mkdir(exist_ok=True, parents=True)
    self._envs = list(range(env.num_envs)) if envs is None else [int(i) for i in envs]
    self._episodes = {}
    self._count = {}
    self.saved = []

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return getattr(self._env, name)

  def reset(self, mask=None):
    obs = self._env.reset(mask)
    host = obs.cpu().numpy()
    picked = None if mask is None else np.asarray(mask.cpu() if hasattr(mask, 'cpu') else mask).astype(bool)
    for i in self._envs:
      if picked is None or picked[i]:
        self._episodes[i] = [{'image': host[i].copy()}]
    return obs

  def step(self, actions):
    env = self._env
    obs, reward, done, info = env.step(actions)
    rec = env.records()
    host_obs = obs.cpu().numpy()
    host_done = done.cpu().numpy().astype(bool)
    acts = np.asarray(actions.cpu() if hasattr(actions, 'cpu') else actions).astype(np.int64)
    sem = info['semantic'].cpu().numpy()
    pos = info['player_pos'].cpu().numpy()
    for i in self._envs:
      r = rec[i]
      rew = int(r['dhealth']) / 10 + (1.0 if int(r['new_unlocked']) else 0.0)   # info['reward'], env.py:97-104,114
      t = {'action': int(acts[i]), 'image': host_obs[i].copy(), 'reward': rew, 'done': bool(host_done[i]),
           'discount': 1 - float(bool(r['dead'])), 'semantic': sem[i].copy(),
           'player_pos': pos[i].astype(np.int64)}
      for k, name in enumerate(env.achievement_names):
        t[f'achievement_{name}'] = int(r['ach'][k])
      for k, name in enumerate(env.item_names):
        t[f'ainventory_{name}'] = int(r['inv'][k])
      self._episodes[i].append(t)
      if host_done[i]:
        unlocked = sum(int(v >= 1) for v in r['ach'][:len(env.achievement_names)])
        self._save(i, unlocked)
    if host_done.any():   # Env.reset for the finished envs; their returned obs becomes the new first frame
      import torch
      mask = torch.from_numpy(host_done.astype(np.uint8)).to(obs.device)
      new_obs = env.reset(mask)
      host_new = new_obs.cpu().numpy()
      for i in self._envs:
        if host_done[i]:
          self._episodes[i] = [{'image': host_new[i].copy()}]
      obs = new_obs
    return obs, reward, done, info

  def _save(self, i, unlocked):
    episode = self._episodes[i]
    k = self._count.get(i, 0)
    self._count[i] = k + 1
    name = f'{self._now()}-env{i}-ep{k}-ach{unlocked}-len{len(episode) - 1}.npz'
    for key, value in episode[1].items():   # zeros for the keys the reset row lacks (recorder.py:141-144)
      if key not in episode[0]:
        episode[0][key] = np.zeros_like(value)
    arrays = {k: np.array([step[k] for step in episode]) for k in episode[0]}
    path = self._directory / name
    np.savez_compressed(str(path), **arrays)
    self.saved.append(path)
