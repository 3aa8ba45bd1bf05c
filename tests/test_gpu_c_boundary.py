"""-m gpu: the drop-in boundary consumed from C: tests/c/boundary_test.c, compiled with
gcc -std=c99 -pedantic against include/crafter_hip.h alone, does create -> upload_tables -> bind_state -> reset -> step
(the second half of the tape in one crafter_step_n call) and writes obs / reward / done; compared with the oracle here."""
import subprocess

import numpy as np
import pytest

from tests import c_boundary
from tests.rollout import oracle_rollouts

pytestmark = pytest.mark.gpu


def test_c_program_steps_the_envs_like_the_oracle(tmp_path):
  n, T, first_seed = 4, 60, 1000
  exe = c_boundary.compile_c(tmp_path / 'boundary_test')
  c_boundary.write_tables(tmp_path / 'tables.bin')
  tape = np.random.RandomState(3).randint(0, 17, size=(T, n)).astype(np.int32)
  tape.tofile(tmp_path / 'tape.bin')
  proc = subprocess.run([str(exe), str(tmp_path / 'tables.bin'), str(tmp_path / 'tape.bin'), str(tmp_path / 'out.bin'), str(n), str(T),
                         str(first_seed)], capture_output=True, text=True, timeout=300)
  assert proc.returncode == 0, proc.stdout + proc.stderr
  raw = np.fromfile(tmp_path / 'out.bin', np.uint8)
  frame = n * 64 * 64 * 3
  per_step = frame + 4 * n + n
  assert raw.size == frame + T * per_step + n * (16 + 32) * 4
  res = oracle_rollouts([dict(kwargs=dict(seed=first_seed + i), actions=tape[:, i], frames=range(T)) for i in range(n)])
  reset = raw[:frame].reshape(n, 64, 64, 3)
  for i, r in enumerate(res):
    assert np.array_equal(reset[i], r['reset_obs']), f'env {i}: reset obs'
  alive = [True] * n
  for t in range(T):
    blk = raw[frame + t * per_step: frame + (t + 1) * per_step]
    obs = blk[:frame].reshape(n, 64, 64, 3)
    rew = blk[frame:frame + 4 * n].view(np.float32)
    done = blk[frame + 4 * n:]
    for i, r in enumerate(res):
      if not alive[i]:
        continue
      assert np.array_equal(obs[i], r['frames'][t]), f'env {i} step {t}: pixels'
      assert rew[i] == r['reward'][t] and bool(done[i]) == r['done'][t], f'env {i} step {t}: reward / done'
      alive[i] = not r['done'][t]
  tail = raw[frame + T * per_step:].view(np.int32).reshape(n, 48)
  for i, r in enumerate(res):
    if alive[i]:
      assert tail[i, :len(r['inv'][-1])].tolist() == r['inv'][-1] and tail[i, 16:16 + len(r['ach'][-1])].tolist() == r['ach'][-1]
  assert sum(alive) >= 2
