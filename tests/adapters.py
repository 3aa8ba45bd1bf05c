"""Single-env adapters with one surface for golden replay: oracle, hostsim, HIP."""
import numpy as np


class OracleAdapter:

  def __init__(self, seed, area, length):
    from oracle.crafter_oracle import OracleEnv
    self.e = OracleEnv(area=area, length=length, seed=seed)

  def reset(self):
    return self.e.reset()

  def step(self, a):
    obs, reward, done, info = self.e.step(a)
    return obs, reward, done, info['semantic']

  def snapshot(self):
    return self.e.snapshot()


def _exact_reward(rec):
  reward = int(rec['dhealth']) / 10
  if int(rec['new_unlocked']):
    reward += 1.0
  return reward


class HostSimAdapter:

  def __init__(self, seed, area, length, **kw):
    from tests.hostsim.driver import HostSimEnv
    self.e = HostSimEnv([seed], area=area, length=length, want_semantic=True, **kw)
    self.length = length

  def reset(self):
    return self.e.reset()[0].copy()

  def step(self, a):
    obs, _, done = self.e.step(np.array([a], np.int32))
    cfg = self.e.cfg
    return obs[0].copy(), _exact_reward(self.e.rec[0]), bool(done[0]), self.e.buf['semantic'][0].reshape(cfg.W, cfg.H)

  def snapshot(self):
    return self.e.snapshot(0)


class HipAdapter:
  """crafter_amd.Env (the drop-in facade) on cuda:0."""

  def __init__(self, seed, area, length):
    from crafter_amd import Env
    self.e = Env(area=area, length=length, seed=seed)

  def reset(self):
    return self.e.reset()

  def step(self, a):
    obs, reward, done, info = self.e.step(a)
    return obs, reward, done, info['semantic']

  def snapshot(self):
    return self.e._batch.snapshot(0)
