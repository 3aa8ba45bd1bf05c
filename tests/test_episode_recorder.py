"""BatchedEpisodeRecorder against the reference's EpisodeRecorder format (recorder.py:100-152).

The expected .npz content is rebuilt from the oracle with the reference recorder's own rules: row 0 = reset
frame + zeros, one row per transition with info['reward'] overriding the returned reward, achievement_* /
ainventory_* columns.  CPU: the kernel bodies behind a BatchedEnv-shaped stand-in; -m gpu: the HIP path."""
import types

import numpy as np
import pytest
import torch

from crafter_amd import abi, state
from crafter_amd.recorder import BatchedEpisodeRecorder
from oracle.crafter_oracle import OracleEnv


def reference_episodes(seed, actions, length):
  """What EpisodeRecorder(crafter.Env(seed=..., length=...)) stores for a tape of actions (oracle-backed)."""
  env = OracleEnv(seed=seed, length=length)
  episodes, cur = [], [{'image': env.reset()}]
  for a in actions:
    obs, reward, done, info = env.step(int(a))
    t = {'action': int(a), 'image': obs, 'reward': reward, 'done': done}
    for k, v in info.items():
      if k not in ('inventory', 'achievements'):
        t[k] = v
    for k, v in info['achievements'].items():
      t[f'achievement_{k}'] = v
    for k, v in info['inventory'].items():
      t[f'ainventory_{k}'] = v
    cur.append(t)
    if done:
      for k, v in cur[1].items():
        if k not in cur[0]:
          cur[0][k] = np.zeros_like(v)
      episodes.append({k: np.array([s[k] for s in cur]) for k in cur[0]})
      cur = [{'image': env.reset()}]
  return episodes


def check(saved, seeds, tape, length):
  by_env = {}
  for path in saved:
    i = int(path.name.split('-env')[1].split('-')[0])
    by_env.setdefault(i, []).append(path)
  for i, seed in enumerate(seeds):
    want = reference_episodes(seed, tape[:, i], length)
    got = by_env.get(i, [])
    assert len(got) == len(want) >= 2
    for path, ep in zip(got, want):
      with np.load(path) as f:
        assert set(f.files) == set(ep.keys())
        assert path.name.endswith(f"-ach{int(sum(ep[k][-1] >= 1 for k in ep if k.startswith('achievement_')))}-len{len(ep['action']) - 1}.npz")
        for k, v in ep.items():
          a = f[k]
          assert a.shape == v.shape, k
          if a.dtype.kind == 'f':
            assert np.array_equal(a.astype(np.float64), np.asarray(v, np.float64)), k
          else:
            assert np.array_equal(a, v), k


class HostSimBatched:
  """The slice of BatchedEnv the recorder touches, over the CPU build of the kernel bodies."""

  def __init__(self, seeds, length):
    from tests.hostsim.driver import HostSimEnv
    self.e = HostSimEnv(seeds, length=length, want_semantic=True)
    self.cfg = self.e.cfg
    self.num_envs = len(seeds)
    rules = self.e.tab.rules
    from crafter_amd import tables
    r = tables.load_rules()
    self.achievement_names = list(r['achievements'])
    self.item_names = list(r['items'])

  def reset(self, mask=None):
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask), np.uint8)
    return torch.from_numpy(self.e.reset(m))

  def step(self, actions):
    obs, rew, done = self.e.step(np.asarray(actions, np.int32))
    objs = state.objs_view(self.e.buf['objs'])
    info = {'semantic': torch.from_numpy(self.e.buf['semantic'].reshape(self.num_envs, self.cfg.W, self.cfg.H)),
            'player_pos': torch.from_numpy(np.stack([objs[:, 1]['x'], objs[:, 1]['y']], 1).astype(np.int32))}
    return torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done), info

  def records(self):
    return self.e.rec


def test_episode_files_match_reference_format_cpu(tmp_path):
  seeds, length, steps = [5, 6], 13, 40
  rec = BatchedEpisodeRecorder(HostSimBatched(seeds, length), tmp_path)
  tape = np.random.RandomState(3).randint(0, 17, size=(steps, len(seeds))).astype(np.int32)
  rec.reset()
  for t in range(steps):
    rec.step(torch.from_numpy(tape[t]))
  check(rec.saved, seeds, tape, length)


def test_recorder_refuses_auto_reset():
  fake = types.SimpleNamespace(cfg=types.SimpleNamespace(auto_reset=1, want_semantic=1), num_envs=1)
  with pytest.raises(ValueError):
    BatchedEpisodeRecorder(fake, '/tmp/unused')


@pytest.mark.gpu
def test_episode_files_match_reference_format_gpu(tmp_path):
  from crafter_amd import BatchedEnv
  seeds, length, steps = [21, 22, 23], 17, 60
  env = BatchedEnv(len(seeds), seeds=seeds, length=length, auto_reset=False, semantic=True)
  rec = BatchedEpisodeRecorder(env, tmp_path)
  tape = np.random.RandomState(9).randint(0, 17, size=(steps, len(seeds))).astype(np.int32)
  rec.reset()
  for t in range(steps):
    rec.step(torch.from_numpy(tape[t]).cuda())
  env.check_errors()
  check(rec.saved, seeds, tape, length)
