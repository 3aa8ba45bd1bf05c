"""The split step (rule half + frame half with the frame record in between: crafter_rules_kernel / crafter_frame_kernel,
env_kernels.hpp) against the fused step on the CPU harness of the same kernel bodies: identical observations, rewards, dones
and state, with and without the world pool, with and without frames."""
import numpy as np
import pytest

from tests.hostsim.driver import HostSimEnv, lib


def _run(split, steps, n, **kw):
  l = lib()
  l.hostsim_set_split(split)
  try:
    env = HostSimEnv([1000 + i for i in range(n)], auto_reset=True, **kw)
    env.reset()
    tape = np.random.RandomState(7).randint(0, 17, size=(steps, n)).astype(np.int32)
    out = []
    for t in range(steps):
      o, r, d = env.step(tape[t])
      out.append((o.copy(), r.copy(), d.copy()))
    return out, {k: env.buf[k].copy() for k in ('mat', 'objs', 'mt', 'rec')}
  finally:
    l.hostsim_set_split(0)


@pytest.mark.parametrize('kw', [dict(pool=False), dict(pool=True), dict(pool=True, render_obs=False)], ids=['requeue', 'pool', 'no-frames'])
def test_split_step_equals_fused_step(kw):
  steps, n = 330, 6   # through the first night (steps 148-272) and the first auto-resets
  a, sa = _run(0, steps, n, **kw)
  b, sb = _run(1, steps, n, **kw)
  for t in range(steps):
    for x, y, what in zip(a[t], b[t], ('obs', 'reward', 'done')):
      assert np.array_equal(x, y), (t, what)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k
