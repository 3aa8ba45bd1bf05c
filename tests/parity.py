"""Shared helpers of the parity tests: compare a product snapshot with the oracle's."""
import numpy as np

FIELDS = ['step', 'episode', 'mat', 'occupied', 'objects', 'inventory', 'achievements', 'sleeping',
          'hunger2', 'thirst2', 'fatigue2', 'recover2', 'player_last_health', 'chunk_order', 'mt_key',
          'mt_pos']


def diff_snapshots(got, want):
  """Returns a list of human-readable differences (empty = identical)."""
  out = []
  for k in FIELDS:
    x, y = got[k], want[k]
    if isinstance(y, np.ndarray):
      if not np.array_equal(np.asarray(x), y):
        bad = np.argwhere(np.asarray(x) != y)
        out.append(f'{k}: {len(bad)} entries differ, first {bad[:3].tolist()}')
    elif x != y:
      if k == 'objects':
        firsts = [(i, p, q) for i, (p, q) in enumerate(zip(x, y)) if p != q][:3]
        out.append(f'objects: len {len(x)} vs {len(y)}, first diffs {firsts}')
      else:
        out.append(f'{k}: {x!r} != {y!r}')
  return out


def assert_same(got, want, where=''):
  d = diff_snapshots(got, want)
  assert not d, f'{where}: ' + '; '.join(d)


# ----------------------------------------------------------------------------------------------
# golden replay (tests/golden/*.npz were produced by the real reference, tools/make_golden.py)
# ----------------------------------------------------------------------------------------------
import hashlib
import pathlib

GOLDEN = pathlib.Path(__file__).resolve().parent / 'golden'


def golden_cases():
  return sorted(p.stem for p in GOLDEN.glob('*.npz'))


def sha8(a):
  return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:8], np.uint64)[0]


def state_of_snapshot(s):
  """snapshot() dict (oracle / hostsim / BatchedEnv) -> the golden file's state layout."""
  return dict(
      mat=np.asarray(s['mat']), objects=np.array(s['objects'], np.int32).reshape(-1, 7),
      inventory=np.array(s['inventory'], np.int32), achievements=np.array(s['achievements'], np.int32),
      misc=np.array([int(s['sleeping']), s['hunger2'], s['thirst2'], s['fatigue2'], s['recover2'],
                     s['player_last_health']], np.int32),
      chunk_order=np.array(s['chunk_order'], np.int32).reshape(-1, 4),
      mt_key=np.asarray(s['mt_key'], np.uint32), mt_pos=np.int32(s['mt_pos']))


def check_state(got, gold, prefix, where):
  for k, v in got.items():
    want = gold[prefix + k]
    assert np.array_equal(np.asarray(v), want), f'{where}: {k} differs from the reference fixture'


def replay_golden(name, make_env):
  """make_env(seed, area, length) -> object with reset()->obs, step(a)->(obs, reward, done, semantic),
  snapshot()->dict.  Replays the fixture's action tape (reset right after every done)."""
  g = np.load(GOLDEN / f'{name}.npz')
  env = make_env(int(g['seed']), tuple(int(v) for v in g['area']), int(g['length']))
  obs = env.reset()
  assert np.array_equal(obs, g['reset_obs']), f'{name}: reset obs'
  check_state(state_of_snapshot(env.snapshot()), g, 'reset_', f'{name} reset')
  frames = {int(s): f for s, f in zip(g['frame_steps'], g['frames'])}
  for t, a in enumerate(g['actions']):
    obs, reward, done, sem = env.step(int(a))
    assert reward == g['rewards'][t], f'{name} step {t}: reward {reward} != {g["rewards"][t]}'
    assert bool(done) == bool(g['dones'][t]), f'{name} step {t}: done'
    assert sha8(sem) == g['sem_sha'][t], f'{name} step {t}: semantic view'
    if t in frames:
      assert np.array_equal(obs, frames[t]), f'{name} step {t}: frame pixels'
    if done:
      obs = env.reset()
    assert sha8(obs) == g['obs_sha'][t], f'{name} step {t}: obs hash'
  check_state(state_of_snapshot(env.snapshot()), g, 'final_', f'{name} final')
