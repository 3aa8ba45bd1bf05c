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
