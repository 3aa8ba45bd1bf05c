"""-m gpu: the HIP path (through the C ABI, via BatchedEnv / Env) against the oracle.

Bit-exact contract: material map, object table (slot order), inventory, achievements, player
counters, chunk order, MT19937 key+position, reward, done, semantic view and every obs pixel.
(Terrain noise itself is "parity unpinned" against the absent opensimplex package: oracle and
device restate the same published algorithm -- see oracle/opensimplex_ref.py.)
"""
import numpy as np
import pytest
import torch

from oracle.crafter_oracle import OracleEnv
from tests.parity import assert_same

pytestmark = pytest.mark.gpu


def _batched(*a, **k):
  from crafter_amd import BatchedEnv
  return BatchedEnv(*a, **k)


def test_extension_loaded_and_no_cpu_path():
  from crafter_amd import lib
  l = lib.load()
  assert l.crafter_abi_version() == 7
  assert torch.cuda.is_available()
  with pytest.raises(Exception):
    _batched(2, device='cpu')


@pytest.mark.parametrize('kernel', ['default', 'early', 'wide'])
def test_reset_and_step_parity_random_policy(kernel, monkeypatch):
  """kernel: the step kernel of the default instance the batch runs on -- the one its size selects (12 envs: the 512-thread
  wide kernel ... which the ordered launch of CRAFTER_ORDER=1 runs replace by crafter_step_kernel<1, 1, 1>), or forced:
  crafter_step_early_kernel (the material half of a day frame drawn while the object loop runs: render.hpp early_frame, round 6;
  by default only batches of at least 2048 envs) / crafter_step_kernel<1, 1, 1> (CRAFTER_STEP_WIDE=0)."""
  if kernel == 'early':
    monkeypatch.setenv('CRAFTER_STEP_EARLY', '1')
    monkeypatch.setenv('CRAFTER_STEP_WIDE', '0')
  elif kernel == 'default':
    monkeypatch.setenv('CRAFTER_STEP_EARLY', '0')
    monkeypatch.setenv('CRAFTER_STEP_WIDE', '0')
  n, steps = 12, 260   # covers the first night (steps 148-272) for every env that survives
  seeds = [1000 + i for i in range(n)]
  env = _batched(n, seeds=seeds, auto_reset=False, semantic=True)
  orcs = [OracleEnv(seed=s) for s in seeds]
  obs = env.reset().cpu().numpy()
  for i, o in enumerate(orcs):
    want = o.reset()
    assert_same(env.snapshot(i), o.snapshot(), f'reset env {i}')
    assert np.array_equal(obs[i], want), f'reset obs env {i}'
  rs = np.random.RandomState(7)
  alive = [True] * n
  for t in range(steps):
    acts = rs.randint(0, 17, size=n).astype(np.int32)
    obs, rew, done, info = env.step(torch.from_numpy(acts).cuda())
    obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    sem = info['semantic'].cpu().numpy()
    inv = info['inventory'].cpu().numpy()
    for i, o in enumerate(orcs):
      if not alive[i]:
        continue
      ob, r, d, inf = o.step(int(acts[i]))
      assert np.array_equal(obs[i], ob), f'obs step {t} env {i} daylight {o.daylight}'
      assert rew[i] == np.float32(r) and bool(done[i]) == bool(d)
      assert np.array_equal(sem[i], inf['semantic'])
      assert inv[i].tolist() == list(inf['inventory'].values())
      if t % 20 == 0 or d:
        assert_same(env.snapshot(i), o.snapshot(), f'step {t} env {i}')
      if d:
        alive[i] = False
  env.check_errors()


def test_auto_reset_parity_short_episodes():
  n, steps, length = 16, 150, 40
  seeds = [77 + 3 * i for i in range(n)]
  env = _batched(n, seeds=seeds, auto_reset=True, length=length)
  orcs = [OracleEnv(seed=s, length=length) for s in seeds]
  obs = env.reset().cpu().numpy()
  for i, o in enumerate(orcs):
    assert np.array_equal(obs[i], o.reset())
  rs = np.random.RandomState(11)
  resets = 0
  for t in range(steps):
    acts = rs.randint(0, 17, size=n).astype(np.int32)
    obs, rew, done, _ = env.step(torch.from_numpy(acts).cuda(), info=False)
    obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
    for i, o in enumerate(orcs):
      ob, r, d, _ = o.step(int(acts[i]))
      if d:
        ob = o.reset()
        resets += 1
      assert rew[i] == np.float32(r) and bool(done[i]) == bool(d), (t, i)
      assert np.array_equal(obs[i], ob), (t, i, d)
  assert resets >= n * (steps // length)
  for i, o in enumerate(orcs):
    assert_same(env.snapshot(i), o.snapshot(), f'final env {i}')
  env.check_errors()


def test_env_facade_matches_oracle_api():
  from crafter_amd import Env
  env, orc = Env(seed=5), OracleEnv(seed=5)
  assert env.action_names == orc.action_names
  assert np.array_equal(env.reset(), orc.reset())
  rs = np.random.RandomState(0)
  for t in range(120):
    a = int(rs.randint(0, 17))
    o1, r1, d1, i1 = env.step(a)
    o2, r2, d2, i2 = orc.step(a)
    assert np.array_equal(o1, o2)
    assert r1 == r2 and type(r1) is float and bool(d1) == bool(d2)
    assert i1['inventory'] == i2['inventory'] and i1['achievements'] == i2['achievements']
    assert i1['discount'] == i2['discount'] and i1['reward'] == i2['reward']
    assert np.array_equal(i1['semantic'], i2['semantic']) and np.array_equal(i1['player_pos'], i2['player_pos'])
    if d1:
      break
  # Env.render() re-draws and (at night) consumes the RNG exactly like the reference
  assert np.array_equal(env.render(), orc.render())
  assert env._step == orc._step and env._world.count('grass') == int((orc.mat == orc.t.mat_id['grass']).sum())
  with pytest.raises(IndexError):
    env.step(17)


def test_full_size_roundtrip_properties():
  """BASELINE config[1] size (1024 envs): size-independent invariants instead of an oracle run."""
  n = 1024
  env = _batched(n, seed=1000, auto_reset=True)
  env.reset()
  tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(300, n)).astype(np.int32)).cuda()
  total_done = 0
  for t in range(300):
    obs, rew, done, info = env.step(tape[t])
    total_done += int(done.sum())
  env.check_errors()
  rec = env.records()
  assert total_done > 0 and (rec['episode'] >= 1).all()
  # the slot map of these LDS-resident worlds is derived from the slot table at every stage-in:
  # no two live objects may share a cell
  assert env.slot_map_derived
  from crafter_amd import state
  objs = state.objs_view(env.state['objs'].cpu().numpy())
  H = env.cfg.H
  for i in range(0, n, 37):
    k = int(rec['nobj'][i])
    cells = [int(o['x']) * H + int(o['y']) for o in objs[i, 1:k] if o['type']]
    assert len(cells) == len(set(cells))
  inv = info['inventory'].cpu().numpy()
  assert (inv >= 0).all() and (inv <= 9).all()
  # the last row / column of the 64x64 frame stay black (63 = 9 * 7 pixels are drawn, env.py:127-129)
  o = obs.cpu().numpy()
  assert (o[:, 63, :, :] == 0).all() and (o[:, :, 63, :] == 0).all()


def test_render_other_size_and_large_world():
  """Env.render((512, 512)) (what the reference's VideoRecorder asks for) and a 256x256 world (maps in HBM)."""
  from crafter_amd import Env
  env, orc = Env(seed=7), OracleEnv(seed=7)
  env.reset(), orc.reset()
  acts = np.random.RandomState(1234 + 7).choice([0, 0, 0, 6, 1, 2, 3, 4, 5], size=175)
  for t, a in enumerate(acts):
    env.step(int(a)), orc.step(int(a))
    if t in (5, 160, 171):
      assert np.array_equal(env.render((512, 512)), orc.render((512, 512))), t
      assert np.array_equal(env.render(), orc.render()), t
  assert_same(env._batch.snapshot(0), orc.snapshot(), 'after renders')
  n, side, length = 3, 256, 9
  seeds = [21, 22, 23]
  big = _batched(n, area=(side, side), seeds=seeds, auto_reset=True, length=length)
  orcs = [OracleEnv(area=(side, side), seed=s, length=length) for s in seeds]
  obs = big.reset().cpu().numpy()
  for i, o in enumerate(orcs):
    assert np.array_equal(obs[i], o.reset())
  rs = np.random.RandomState(3)
  for t in range(40):   # long enough for the pool (generated 16+ steps behind) to serve resets
    acts = rs.randint(0, 17, size=n).astype(np.int32)
    obs, rew, done, _ = big.step(torch.from_numpy(acts).cuda(), info=False)
    obs = obs.cpu().numpy()
    for i, o in enumerate(orcs):
      ob, r, d, _ = o.step(int(acts[i]))
      if d:
        ob = o.reset()
      assert np.array_equal(obs[i], ob), (t, i)
  for i, o in enumerate(orcs):
    assert_same(big.snapshot(i), o.snapshot(), f'256x256 env {i}')
  big.check_errors()


def test_vec_env_view_protocol_and_terminal_observation():
  """SB3-style VecEnv view: numpy in/out, auto-reset with the finished episode's last frame in infos."""
  from crafter_amd import VecEnvView
  seeds, length, steps = [31, 32, 33, 34], 11, 30
  venv = VecEnvView(len(seeds), seeds=seeds, length=length)
  orcs = [OracleEnv(seed=s, length=length) for s in seeds]
  obs = venv.reset()
  assert obs.shape == (4, 64, 64, 3) and obs.dtype == np.uint8 and venv.action_space.n == 17
  for i, o in enumerate(orcs):
    assert np.array_equal(obs[i], o.reset())
  rs = np.random.RandomState(5)
  ends = 0
  for t in range(steps):
    acts = rs.randint(0, 17, size=len(seeds))
    obs, rew, done, infos = venv.step(acts)
    for i, o in enumerate(orcs):
      ob, r, d, inf = o.step(int(acts[i]))
      assert bool(done[i]) == bool(d) and rew[i] == np.float32(r)
      assert infos[i]['inventory'] == inf['inventory'] and infos[i]['achievements'] == inf['achievements']
      assert infos[i]['reward'] == inf['reward'] and tuple(infos[i]['player_pos']) == tuple(inf['player_pos'])
      if d:
        ends += 1
        assert np.array_equal(infos[i]['terminal_observation'], ob)
        assert infos[i]['TimeLimit.truncated'] == (inf['discount'] == 1.0)
        ob = o.reset()
      assert np.array_equal(obs[i], ob), (t, i)
  assert ends >= 2 * len(seeds)


def test_device_status_bits_become_exceptions():
  from crafter_amd import CrafterDeviceError
  env = _batched(3, seed=5, auto_reset=False)
  env.reset()
  env.step(torch.tensor([0, 42, 0], dtype=torch.int32).cuda())
  with pytest.raises(CrafterDeviceError, match='action index out of range'):
    env.check_errors()
  small = _batched(2, seed=5, auto_reset=False, max_objects=6)
  small.reset()
  with pytest.raises(CrafterDeviceError, match='object table overflow'):
    small.check_errors()


def test_mutated_rules_run_the_generic_rules_instance():
  """run_random.py:21-22 style rule mutation (health max 5): the uploaded rules differ from the compiled-in defaults, so
  the step kernel instance that stages the rules into LDS runs -- bit-exact against the oracle with the same rules."""
  import copy
  from crafter_amd import tables
  rules = copy.deepcopy(tables.load_rules())
  rules['items']['health'] = {'max': 5, 'initial': 5}
  seeds = [61, 62, 63]
  env = _batched(len(seeds), seeds=seeds, auto_reset=False, rules=rules)
  orcs = [OracleEnv(seed=s, rules=copy.deepcopy(rules)) for s in seeds]
  obs = env.reset().cpu().numpy()
  for i, o in enumerate(orcs):
    assert np.array_equal(obs[i], o.reset())
  rs = np.random.RandomState(17)
  for t in range(120):
    acts = rs.randint(0, 17, size=len(seeds)).astype(np.int32)
    obs, rew, done, info = env.step(torch.from_numpy(acts).cuda())
    obs, done = obs.cpu().numpy(), done.cpu().numpy()
    inv = info['inventory'].cpu().numpy()
    for i, o in enumerate(orcs):
      if o is None:
        continue
      ob, r, d, inf = o.step(int(acts[i]))
      assert np.array_equal(obs[i], ob), (t, i)
      assert bool(done[i]) == bool(d) and int(inv[i][0]) == inf['inventory']['health'] <= 5
      if d:
        orcs[i] = None
  env.check_errors()
