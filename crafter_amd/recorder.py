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
