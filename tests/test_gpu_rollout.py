"""-m gpu: crafter_step_n / BatchedEnv.rollout -- T steps of every env in one call (one launch per stretch between two
generation batches, an env goes on to its next step without waiting for the others) -- against the oracle: every
observation, reward and done of every step, the full state at the end of every call; with the world pool (envs sampled
from a full-width batch), without it (every reset goes through the regeneration kernel's half of the rollout), and with
episodes so short that an env resets more than once inside one stretch."""
import numpy as np
import pytest
import torch

from tests.compare import assert_same, sha8
from tests.rollout import oracle_rollouts

pytestmark = pytest.mark.gpu


def _batched(*a, **k):
  from crafter_amd import BatchedEnv
  return BatchedEnv(*a, **k)


def _rollouts_against_oracle(env, tapes, results, index, calls, where):
  """calls: lengths of the consecutive rollout() calls that together cover the tape."""
  assert sum(calls) == tapes.shape[0]
  sel = torch.tensor(index, device=env.device)
  dev_tape = torch.from_numpy(np.ascontiguousarray(tapes)).to(env.device)
  obs0 = env.reset()
  host = obs0[sel].cpu().numpy()
  for k, i in enumerate(index):
    assert np.array_equal(host[k], results[k]['reset_obs']), f'{where} env {i}: reset obs'
  t0 = 0
  for T in calls:
    obs, rew, done = env.rollout(dev_tape[t0:t0 + T])
    obs_h, rew_h, done_h = obs[:, sel].cpu().numpy(), rew[:, sel].cpu().numpy(), done[:, sel].cpu().numpy()
    for k, i in enumerate(index):
      r = results[k]
      for t in range(T):
        g = t0 + t
        assert rew_h[t, k] == r['reward'][g] and bool(done_h[t, k]) == r['done'][g], f'{where} env {i} step {g}: reward / done'
        assert sha8(obs_h[t, k]) == r['obs_sha'][g], f'{where} env {i} step {g}: obs pixels'
      g = t0 + T - 1
      if g in r['snapshots']:
        assert_same(env.snapshot(i), r['snapshots'][g], f'{where} env {i} after step {g}')
    t0 += T
  for k, i in enumerate(index):
    assert_same(env.snapshot(i), results[k]['final_snapshot'], f'{where} env {i} final')
  env.check_errors()


def test_rollout_of_the_metric_workload_sampled_in_place():
  n, calls = 4096, [16, 64, 7, 1, 100, 62]   # through the first night and the first auto-resets
  T = sum(calls)
  ends = np.cumsum(calls) - 1
  sample = sorted(set(np.random.RandomState(9).randint(0, n, size=12).tolist() + [0, n - 1]))
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tapes[:, i], snapshots=[int(e) for e in ends], auto_reset=True)
                         for i in sample])
  env = _batched(n, seed=1000, auto_reset=True)
  _rollouts_against_oracle(env, tapes, res, sample, calls, 'rollout 4096')
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] > 0, ps


@pytest.mark.parametrize('kw', [dict(gen_period=-1), dict(length=9), dict(length=3)], ids=['no-pool', 'length-9', 'length-3'])
def test_rollout_when_envs_run_out_of_pooled_worlds(kw):
  """Without the pool every reset, with very short episodes many of them, happen INSIDE a stretch with no world ready:
  the env stops, the regeneration kernel generates its world, draws that step's frame and runs the rest of its steps
  (generating again if it finishes again)."""
  n, calls = 256, [40, 16, 33, 61]
  T = sum(calls)
  ends = np.cumsum(calls) - 1
  length = kw.get('length', 60)
  sample = list(range(0, n, 17))
  tapes = np.random.RandomState(4321).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=50 + i, length=length), actions=tapes[:, i], snapshots=[int(e) for e in ends],
                              auto_reset=True) for i in sample])
  env = _batched(n, seed=50, length=length, auto_reset=True, **{k: v for k, v in kw.items() if k != 'length'})
  _rollouts_against_oracle(env, tapes, res, sample, calls, f'rollout {kw}')


def test_rollout_then_step_then_rollout_and_the_generic_instances():
  """rollout() and step() interleave freely (same state, same scheduler), on an instance other than the default one
  (another area: generic kernel), without frames, and into caller-provided tensors."""
  n = 64
  tapes = np.random.RandomState(8).randint(0, 17, size=(120, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=300 + i, area=(48, 40), length=70), actions=tapes[:, i], snapshots=[39, 59, 119],
                              auto_reset=True) for i in range(n)])
  env = _batched(n, seed=300, area=(48, 40), length=70, auto_reset=True)
  env.reset()
  dev = torch.from_numpy(tapes).to(env.device)
  out = (torch.empty((40,) + tuple(env.obs.shape), dtype=torch.uint8, device=env.device),
         torch.empty((40, n), dtype=torch.float32, device=env.device), torch.empty((40, n), dtype=torch.uint8, device=env.device))
  o, r, d = env.rollout(dev[:40], out=out)
  assert o is out[0]
  got = [(sha8(x), float(a), bool(b)) for x, a, b in zip(o[-1].cpu().numpy(), r[-1].cpu().numpy(), d[-1].cpu().numpy())]
  assert got == [(res[i]['obs_sha'][39], res[i]['reward'][39], res[i]['done'][39]) for i in range(n)]
  for i in range(0, n, 9):
    assert_same(env.snapshot(i), res[i]['snapshots'][39], f'env {i} after the first rollout')
  for t in range(40, 60):
    obs, rew, done, _ = env.step(dev[t], info=False)
  for i in range(0, n, 9):
    assert_same(env.snapshot(i), res[i]['snapshots'][59], f'env {i} after the steps')
  _, r2, d2 = env.rollout(dev[60:], obs=False)
  assert r2.cpu().numpy().tolist() == [[res[i]['reward'][t] for i in range(n)] for t in range(60, 120)]
  for i in range(n):
    assert_same(env.snapshot(i), res[i]['final_snapshot'], f'env {i} final')
  env.check_errors()


@pytest.mark.parametrize('kw', [dict(), dict(view=(7, 9), size=(84, 72))], ids=['default-view', 'other-view'])
def test_rollout_on_256x256_worlds(kw):
  """crafter_rollout_kernel<0, 2, 1> / <0, 0, 0>: worlds whose maps AND slot table stay in global memory (env_core.hpp FarSlot)
  -- a resident step scans the table again at its head.  Through the evening into the night (balance passes with spawns and
  despawns all over the 484 chunks), episodes of 190 steps so that worlds are adopted inside a stretch; with crafter.Env()'s
  view (compiled in) and with another one (the instance with nothing compiled in)."""
  n, T = 64, 260
  sample = [0, 1, 17, 31, 40, 63]
  tapes = np.random.RandomState(21).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(area=(256, 256), seed=2000 + i, length=190, **kw), actions=tapes[:, i], snapshots=[99, 179, 259],
                              auto_reset=True) for i in sample])
  assert sum(r['episodes'] for r in res) >= 6 and max(r['night_steps'] for r in res) >= 20
  env = _batched(n, area=(256, 256), seed=2000, length=190, auto_reset=True, **kw)
  assert not env.slot_map_derived and env.step_instance == ('crafter_step_kernel<0, 0, 0>' if kw else 'crafter_step_kernel<0, 2, 1>')
  _rollouts_against_oracle(env, tapes, res, sample, [100, 80, 80], 'rollout 256^2')
