"""TEST INFRASTRUCTURE: the CPU harness (tests/hostsim, the kernel bodies run serially on the host) behind the part
of the BatchedEnv surface the comparison helpers use, so that the GPU parity tests' logic -- tapes, gifts, sampling,
oracle rollouts -- is exercised in the CPU suite too.  torch CPU tensors alias the harness's numpy buffers."""
import numpy as np
import torch

from crafter_amd import abi
from tests.hostsim.driver import HostSimEnv


class HostSimBatched:

  def __init__(self, num_envs, area=(64, 64), view=(9, 9), size=(64, 64), reward=True, length=10000, seed=None,
               seeds=None, auto_reset=True, semantic=False, render=True, pool=False, **kw):
    seeds = list(seeds) if seeds is not None else [seed + i for i in range(num_envs)]
    self._hs = HostSimEnv(seeds, area=area, view=view, size=size, reward=reward, length=length, pool=pool,
                          auto_reset=auto_reset, want_semantic=semantic, render_obs=render, **kw)
    self.num_envs = num_envs
    self.cfg = self._hs.cfg
    self.device = torch.device('cpu')
    self.item_names = list(self._hs.rules_dict['items'])
    self.achievement_names = list(self._hs.rules_dict['achievements'])
    self._rec_i32 = torch.from_numpy(self._hs.buf['rec'].view(np.int32))
    self._off = {name: abi.REC_DTYPE.fields[name][1] // 4 for name in abi.REC_DTYPE.names}
    self.terminal = torch.from_numpy(self._hs.terminal)

  def reset(self, mask=None):
    return torch.from_numpy(self._hs.reset(None if mask is None else np.asarray(mask)))

  def step(self, actions, info=True):
    obs, rew, done = self._hs.step(np.asarray(actions))
    return torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done), {}

  def snapshot(self, i):
    return self._hs.snapshot(i)

  def records(self):
    return self._hs.rec

  def check_errors(self):
    bad = np.nonzero(self._hs.rec['status'])[0]
    assert not len(bad), f'status bits {self._hs.rec["status"][bad[0]]:#x} on env {bad[0]}'
