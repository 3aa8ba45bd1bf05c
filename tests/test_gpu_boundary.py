"""-m gpu: the drop-in boundary driven the way the reference's own wrappers and CLIs drive ``crafter.Env`` --
StatsRecorder / VideoRecorder / EpisodeRecorder call patterns (recorder.py:53-66,87-92,122-152; /root/reference
does not exist on the GPU box, so the patterns are replayed here against the oracle doing the same calls),
``BatchedStatsRecorder`` rows from the kernel-written terminal record, ``python -m crafter_amd.run_random``."""
import json
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.parity import assert_same, sha8
from tests.rollout import oracle_rollouts

pytestmark = pytest.mark.gpu


def test_facade_driven_like_the_reference_recorder_stack(tmp_path):
  """crafter.Recorder(env, dir) = StatsRecorder(VideoRecorder(EpisodeRecorder(env))): every step is followed by
  ``env.render((512, 512))`` (recorder.py:87,92), which at night draws the frame's noise from the env's RNG a second
  time (engine.py:208-211) -- the trajectory only stays on the reference's if the device consumes it identically.
  Rows, frames and per-step info against the oracle driven through the same call sequence."""
  from crafter_amd import Env, EnvStatsRecorder
  T, seed = 230, 70   # this player dies at step 218, 71 steps into the night (one stats row)
  acts = np.random.RandomState(2000 + seed).choice([0, 0, 0, 6, 1, 2, 3, 4, 5], size=T)
  (want,) = oracle_rollouts([dict(kwargs=dict(seed=seed), actions=acts, render_each_step=(512, 512),
                                  snapshots=(5, 150, 160, 175, 199, 217), frames=(160, 175, 217))])
  assert want['night_steps'] >= 60 and len(want['rows']) == 1, 'the run must reach deep into the night and end'
  env = EnvStatsRecorder(Env(seed=seed), tmp_path)   # StatsRecorder idiom incl. __getattr__ forwarding
  assert np.array_equal(env.reset(), want['reset_obs'])
  video = [env.render((512, 512))]                   # VideoRecorder.reset
  assert video[0].shape == (512, 512, 3)
  total = 0.0
  for t in range(want['steps_played']):
    obs, reward, done, info = env.step(int(acts[t]))
    frame = env.render((512, 512))                   # VideoRecorder.step
    total += info['reward']                          # StatsRecorder.step
    assert sha8(obs) == want['obs_sha'][t], f'step {t}: obs'
    assert sha8(frame) == want['extra_sha'][t], f'step {t}: 512x512 frame'
    assert np.float32(reward) == want['reward'][t] and bool(done) == want['done'][t]
    assert list(info['inventory'].values()) == want['inv'][t] and list(info['achievements'].values()) == want['ach'][t]
    if t in want['frames']:
      assert np.array_equal(obs, want['frames'][t])
    if t in want['snapshots']:
      assert_same(env._batch.snapshot(0), want['snapshots'][t], f'step {t}')
    if done:
      break
  if want['rows']:
    rows = [json.loads(l) for l in (tmp_path / 'stats.jsonl').read_text().splitlines()]
    assert rows == [r for _, r in want['rows']]
    assert rows[-1]['reward'] == round(total, 1)


def test_batched_stats_recorder_rows_from_the_kernel_written_terminal_record(tmp_path):
  """stats.jsonl rows (recorder.py:53-66: length, reward rounded to 0.1, achievement_* counts) of every episode a batch
  finishes, auto-reset inside the step kernel included, equal to what StatsRecorder accumulates on the host."""
  from crafter_amd import BatchedEnv, BatchedStatsRecorder
  n, T, length = 16, 130, 40
  seeds = [300 + 7 * i for i in range(n)]
  tapes = np.random.RandomState(21).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=s, length=length), actions=tapes[:, i], auto_reset=True)
                         for i, s in enumerate(seeds)])
  want = sorted(((t, i, row) for i, r in enumerate(res) for t, row in r['rows']), key=lambda x: (x[0], x[1]))
  assert len(want) >= 3 * n
  assert any(row['reward'] != 0 for _, _, row in want) and any(row['length'] < length for _, _, row in want)
  env = BatchedStatsRecorder(BatchedEnv(n, seeds=seeds, length=length, auto_reset=True), tmp_path)
  env.reset()
  dev = torch.from_numpy(tapes).to(env.device)
  for t in range(T):
    env.step(dev[t], info=False)
  env.check_errors()
  env.close()
  rows = [json.loads(l) for l in (tmp_path / 'stats.jsonl').read_text().splitlines()]
  assert rows == [row for _, _, row in want]
  assert env.episodes == len(want)


def test_run_random_cli(tmp_path):
  """BASELINE configs[0] plumbing on the MI355X path: ``python -m crafter_amd.run_random`` with the reference CLI's flags
  (run_random.py:10-44), one env and a batch, --record writing stats.jsonl in both modes."""
  for extra, min_rows in ((['--envs', '8', '--episodes', '1'], 8), ([], 1)):
    out = tmp_path / ('batch' if extra else 'single')
    proc = subprocess.run([sys.executable, '-m', 'crafter_amd.run_random', '--seed', '0', '--length', '60',
                           '--record', str(out)] + extra, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert 'Reset time' in proc.stdout and 'Step time' in proc.stdout
    rows = [json.loads(l) for l in (out / 'stats.jsonl').read_text().splitlines()]
    assert len(rows) >= min_rows and all(1 <= r['length'] <= 60 and 'achievement_collect_wood' in r for r in rows)


def test_step_kernel_instances_and_reference_gym_ids():
  """The fast instance only for crafter.Env()'s defaults with the shipped data.yaml; everything else runs the generic
  code.  register_reference_ids() is importable without gym only as far as gym itself is (ImportError)."""
  import copy
  import crafter_amd
  from crafter_amd import BatchedEnv, tables
  assert BatchedEnv(2, seed=1).step_instance == 'crafter_step_kernel<1, 1, 1>'
  rules = copy.deepcopy(tables.load_rules())
  rules['items']['health'] = {'max': 5, 'initial': 5}
  assert BatchedEnv(2, seed=1, rules=rules).step_instance == 'crafter_step_kernel<1, 1, 0>'
  assert BatchedEnv(2, seed=1, area=(32, 32)).step_instance == 'crafter_step_kernel<1, 0, 0>'
  assert BatchedEnv(2, seed=1, size=(100, 72)).step_instance == 'crafter_step_kernel<1, 0, 0>'
  try:
    import gym  # noqa: F401
  except ImportError:
    with pytest.raises(ImportError):
      crafter_amd.register_reference_ids()
  else:
    crafter_amd.register_reference_ids(force=True)
    env = gym.make('CrafterReward-v1')
    assert env.unwrapped.__class__ is crafter_amd.Env
