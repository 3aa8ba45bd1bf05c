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
def test_split_step_equals_fused_step(kw, mode=1):
  steps, n = 330, 6   # through the first night (steps 148-272) and the first auto-resets
  a, sa = _run(0, steps, n, **kw)
  b, sb = _run(mode, steps, n, **kw)
  for t in range(steps):
    for x, y, what in zip(a[t], b[t], ('obs', 'reward', 'done')):
      assert np.array_equal(x, y), (t, what)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k


def _run_gifted(split, tapes, gifts, seeds, **kw):
  l = lib()
  l.hostsim_set_split(split)
  try:
    env = HostSimEnv(seeds, auto_reset=False, **kw)
    names = list(env.rules_dict['items'])
    env.reset()
    out = []
    for t in range(tapes.shape[0]):
      for i, g in enumerate(gifts):
        for item, amount in (g.get(t) or {}).items():
          env.rec['inv'][i, names.index(item)] = amount
      o, r, d = env.step(tapes[t])
      out.append((o.copy(), r.copy(), d.copy(), env.buf['semantic'].copy() if kw.get('want_semantic') else None))
    return out, {k: env.buf[k].copy() for k in ('mat', 'objs', 'mt', 'rec', 'chunk_order', 'census')}
  finally:
    l.hostsim_set_split(0)


def _same(a, sa, b, sb):
  for t, (x, y) in enumerate(zip(a, b)):
    for u, v, what in zip(x, y, ('obs', 'reward', 'done', 'semantic')):
      assert (u is None and v is None) or np.array_equal(u, v), (t, what)
  for k in sa:
    assert np.array_equal(sa[k], sb[k]), k


def test_split_rule_wave_on_scripted_tapes():
  """The rule wave of the split step has no cell -> slot map (lane-register occupancy) and only a window of the material
  map: crafting (World.nearby), placing, collecting, combat, arrows, plants, sleeping -- and info['semantic'] -- against
  the fused step on the scripted tapes."""
  from tests import scenarios
  T = 260
  plan = [('builder', 3), ('builder', 12), ('sleeper', 21), ('fighter', 5), ('fighter', 8)]
  made = [scenarios.SCENARIOS[k](T, s) for k, s in plan]
  tapes = np.stack([a for a, _ in made], 1).astype(np.int32)
  gifts, seeds = [g for _, g in made], [s for _, s in plan]
  a, sa = _run_gifted(0, tapes, gifts, seeds, want_semantic=True)
  b, sb = _run_gifted(1, tapes, gifts, seeds, want_semantic=True)
  _same(a, sa, b, sb)


def test_split_rule_wave_window_at_the_map_edges():
  """Players marched into the corners and along the edges of the map (the window is clamped to the map there, and chunks
  far from it are balanced from HBM), kept alive by gifts, through a night."""
  T = 420
  legs = {0: [1] * 45 + [3] * 45, 1: [2] * 45 + [4] * 45, 2: [1] * 45 + [4] * 45, 3: [2] * 45 + [3] * 45}
  rs = np.random.RandomState(4)
  tapes = np.zeros((T, 4), np.int32)
  for i in range(4):
    seq = []
    while len(seq) < T:
      seq += legs[i] + rs.choice([0, 1, 2, 3, 4, 5], size=30).tolist()
    tapes[:, i] = seq[:T]
  gifts = [{t: dict(health=9, food=9, drink=9, energy=9) for t in range(T)} for _ in range(4)]
  seeds = [11, 12, 13, 14]
  a, sa = _run_gifted(0, tapes, gifts, seeds)
  b, sb = _run_gifted(1, tapes, gifts, seeds)
  _same(a, sa, b, sb)
  c, sc = _run_gifted(2, tapes, gifts, seeds)
  _same(a, sa, c, sc)
  xs = sb['objs'].view(np.uint16).reshape(4, -1, 8)[:, 1, 2:4]
  assert ((xs < 8) | (xs > 55)).any(), f'some player must have ended near an edge: {xs.tolist()}'


def test_split_rule_wave_on_one_long_episode():
  from tests import scenarios
  T, seeds = 1200, [100, 124]
  made = [scenarios.SCENARIOS['survivor'](T, s) for s in seeds]
  tapes = np.stack([a for a, _ in made], 1).astype(np.int32)
  gifts = [g for _, g in made]
  a, sa = _run_gifted(0, tapes, gifts, seeds)
  b, sb = _run_gifted(1, tapes, gifts, seeds)
  _same(a, sa, b, sb)
  c, sc = _run_gifted(2, tapes, gifts, seeds)
  _same(a, sa, c, sc)


def test_noise_generated_ahead_equals_the_in_frame_pass():
  """The fused step's night frames take their noise from MT19937 states generated AHEAD of the rules (noise_chain: a copy of
  the staged state regenerated eleven times while the rules run; the frame's quads then light their own pixels in any
  order) -- against the in-frame pass that regenerates epoch by epoch in stream order: identical frames, rewards, dones
  and final states (RNG included) over a night with auto-resets, and with the player asleep (scripted)."""
  l = lib()
  from tests import scenarios

  def run(ahead):
    l.hostsim_set_noise_ahead(ahead)
    try:
      a = _run(0, 330, 6, pool=True)
      made = [scenarios.SCENARIOS[k](260, s_) for k, s_ in (('sleeper', 21), ('survivor', 100))]
      tapes = np.stack([t for t, _ in made], 1).astype(np.int32)
      b = _run_gifted(0, tapes, [g for _, g in made], [21, 100])
      return a, b
    finally:
      l.hostsim_set_noise_ahead(1)

  (a1, sa1), (b1, sb1) = run(1)
  (a0, sa0), (b0, sb0) = run(0)
  for t in range(len(a1)):
    for x, y, what in zip(a1[t], a0[t], ('obs', 'reward', 'done')):
      assert np.array_equal(x, y), (t, what)
  for k in sa1:
    assert np.array_equal(sa1[k], sa0[k]), k
  _same(b1, sb1, b0, sb0)
