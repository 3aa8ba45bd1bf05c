"""TEST INFRASTRUCTURE: steps a batched env (BatchedEnv on the GPU, or the CPU harness behind tests/hostsim/shim.py)
through an action tape and compares it with oracle trajectories computed by tests/rollout.py."""
import numpy as np
import torch

from tests.parity import assert_same, sha8


def compare_with_rollouts(env, tapes, results, index=None, gifts=None, pixels=True, snapshots=(), where='', reset_at=(),
                          check_every_step=False):
  """Steps `env` through tapes [T][N] and compares the envs listed in `index` (default: all, in order) with the
  oracle `results` (tests/rollout.py) -- obs hash, reward, done, inventory, achievements every step, the full
  state at `snapshots` and at the end.  Envs without auto-reset drop out of the comparison when they finish."""
  n = env.num_envs
  index = list(range(n)) if index is None else list(index)
  T = tapes.shape[0]
  dev_tape = torch.from_numpy(np.ascontiguousarray(tapes)).to(env.device)
  inv0, na, ni = env._off['inv'], len(env.achievement_names), len(env.item_names)
  ach0 = env._off['ach']
  sel = torch.tensor(index, device=env.device)
  alive = {i: True for i in index}
  obs = env.reset()
  if pixels:
    host = obs[sel].cpu().numpy()
    for k, i in enumerate(index):
      assert np.array_equal(host[k], results[k]['reset_obs']), f'{where} env {i}: reset obs'
  for k, i in enumerate(index):
    assert_same(env.snapshot(i), results[k]['reset_snapshot'], f'{where} env {i} reset')
  for t in range(T):
    if gifts:
      for k, i in enumerate(index):
        for item, amount in (gifts[k].get(t) or {}).items():
          env._rec_i32[i, inv0 + env.item_names.index(item)] = int(amount)
    obs, rew, done, _ = env.step(dev_tape[t], info=False)
    if check_every_step:   # (as crafter_amd.Env does: also where the slot table grows ahead of its objects)
      env.check_errors()
    rec = env._rec_i32[sel].cpu().numpy()
    rew_h, done_h = rew[sel].cpu().numpy(), done[sel].cpu().numpy()
    obs_h = obs[sel].cpu().numpy() if pixels else None
    for k, i in enumerate(index):
      r = results[k]
      if not alive[i]:
        continue
      assert rew_h[k] == r['reward'][t] and bool(done_h[k]) == r['done'][t], f'{where} env {i} step {t}: reward / done'
      if not (env.cfg.auto_reset and r['done'][t]):   # after an auto-reset the record already belongs to the new episode
        assert rec[k, inv0:inv0 + ni].tolist() == r['inv'][t], f'{where} env {i} step {t}: inventory'
        assert rec[k, ach0:ach0 + na].tolist() == r['ach'][t], f'{where} env {i} step {t}: achievements'
      if pixels:
        assert sha8(obs_h[k]) == r['obs_sha'][t], f'{where} env {i} step {t}: obs pixels'
      if t in r['snapshots']:
        assert_same(env.snapshot(i), r['snapshots'][t], f'{where} env {i} step {t}')
      if r['done'][t] and not env.cfg.auto_reset:
        alive[i] = False
    if t in reset_at:   # Env.reset() of every env in the middle of its episode
      obs = env.reset()
      host = obs[sel].cpu().numpy()
      for k, i in enumerate(index):
        r = results[k]
        if pixels:
          assert np.array_equal(host[k], r['manual_reset_obs'][t]), f'{where} env {i}: obs of the reset after step {t}'
        assert_same(env.snapshot(i), r['manual_reset_snapshot'][t], f'{where} env {i} reset after step {t}')
  for k, i in enumerate(index):
    if alive[i]:
      assert_same(env.snapshot(i), results[k]['final_snapshot'], f'{where} env {i} final')
  env.check_errors()


