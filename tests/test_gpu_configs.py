"""-m gpu: the BASELINE.json configurations and rule paths a short random run does not reach, HIP path (through the
C ABI) vs the oracle -- render-off dynamics (config 5), deep 256x256 runs (config 4), scripted crafting / combat /
sleeping tapes with inventory gifts poked into the device state, and a 4096-env batch (the metric's workload)
whose sampled envs are read back from THAT batch.  The oracle trajectories are computed up front, one process per
env (tests/rollout.py)."""
import os

import numpy as np
import pytest
import torch

from tests import scenarios
from tests.compare import compare_with_rollouts as _compare
from tests.rollout import oracle_rollouts

pytestmark = pytest.mark.gpu


def _batched(*a, **k):
  from crafter_amd import BatchedEnv
  return BatchedEnv(*a, **k)


def test_config5_render_off_advances_the_rng_through_the_night():
  """BASELINE configs[4]: render disabled.  The reference draws the night noise (engine.py:208-211, 3087 doubles per
  night frame) inside step() whoever looks at the pixels; with render=False the device must advance each env's
  MT19937 stream identically or every later draw differs.  State + RNG key/position vs the oracle across a night."""
  n, T = 8, 330
  seeds = [500 + i for i in range(n)]
  rs = np.random.RandomState(55)
  tapes = np.stack([rs.choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if i % 2 == 0 else rs.randint(0, 17, size=T)
                    for i in range(n)], 1).astype(np.int32)
  snaps = list(range(0, T, 20)) + list(range(146, 156)) + list(range(268, 278))
  res = oracle_rollouts([dict(kwargs=dict(seed=s), actions=tapes[:, i], snapshots=snaps, auto_reset=True)
                         for i, s in enumerate(seeds)])
  assert max(r['night_steps'] for r in res) >= 100, 'some env must live through the night (steps 148-272)'
  assert sum(r['night_balance_steps'] for r in res) >= 8
  env = _batched(n, seeds=seeds, auto_reset=True, render=False)
  _compare(env, tapes, res, pixels=False, where='config5')
  assert env.step_instance == 'crafter_step_kernel<1, 1, 1>'


def test_config4_deep_256x256_worlds():
  """BASELINE configs[3] at depth: 256x256 worlds (maps in HBM, generic kernel instance, >128 live objects so the
  slot table's tail is fetched after the blind prefix), default episode length, through the first night with
  balance steps at night, auto-reset through the world pool when the players die."""
  T = 330
  seeds = [42, 43, 49, 51, 40, 45, 46, 55]
  tapes = []
  for s in seeds:
    rs = np.random.RandomState(900 + s)
    tapes.append(rs.choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if s % 2 == 0 else rs.randint(0, 17, size=T))
  tapes = np.stack(tapes, 1).astype(np.int32)
  snaps = list(range(0, T, 30)) + [150, 160, 170, 200, 210, 220]
  res = oracle_rollouts([dict(kwargs=dict(area=(256, 256), seed=s), actions=tapes[:, i], snapshots=snaps, auto_reset=True)
                         for i, s in enumerate(seeds)])
  assert max(r['max_objects'] for r in res) > 1000 and min(r['max_objects'] for r in res) > 128
  assert max(r['night_balance_steps'] for r in res) >= 8 and max(r['night_steps'] for r in res) >= 80
  assert sum(r['episodes'] for r in res) >= 8
  env = _batched(len(seeds), area=(256, 256), seeds=seeds, auto_reset=True)
  assert not env.slot_map_derived and env.step_instance == 'crafter_step_kernel<0, 2, 1>'
  _compare(env, tapes, res, where='config4')


def test_config4_tables_in_global_memory_stay_consistent():
  """256x256 worlds keep their maps, slot table and chunk tables in global memory between the steps (env_core.hpp FarSlot: every
  write goes through, the chunk tables are only stored when a step touched a new chunk, holes stay in the slot table).  After
  EVERY step of 260 (deaths, adoptions of pooled worlds, the evening's balance passes): chunk_order holds no chunk twice and
  every chunk in it has its flag (round 6: one build lost a wave's share of the flags at an adoption -- the next step then listed
  those chunks again), every live record's cell names its slot in the slot map and the slot map names nothing else."""
  from crafter_amd import state
  seeds = [42, 43, 49, 51, 40, 45, 46, 55]
  T = 260
  tapes = np.stack([np.random.RandomState(900 + s).choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if s % 2 == 0 else
                    np.random.RandomState(900 + s).randint(0, 17, size=T) for s in seeds], 1).astype(np.int32)
  env = _batched(len(seeds), area=(256, 256), seeds=seeds, auto_reset=True)
  env.reset()
  dev = torch.from_numpy(tapes).to(env.device)
  episodes = 0
  for t in range(T):
    _, _, done, _ = env.step(dev[t], info=False)
    episodes += int(done.sum())
    rec = state.rec_view(env.state['rec'].cpu().numpy())
    order = env.state['chunk_order'].cpu().numpy().view(np.uint16).reshape(len(seeds), -1)
    seen = env.state['chunk_seen'].cpu().numpy().reshape(len(seeds), -1)
    check_maps = t % 10 == 0 or t > T - 20
    if check_maps:
      objs = state.objs_view(env.state['objs'].cpu().numpy())
      objmap = env.state['objmap'].cpu().numpy().view(np.uint16).reshape(len(seeds), env.cfg.W, env.cfg.H)
    for i in range(len(seeds)):
      n = int(rec[i]['nchunks_seen'])
      o = order[i][:n].astype(np.int64)
      assert len(np.unique(o)) == n, f'step {t} env {i}: a chunk is listed twice'
      assert seen[i][o].all() and int(seen[i].sum()) == n, f'step {t} env {i}: chunk flags and chunk order disagree'
      if check_maps:
        nobj = int(rec[i]['nobj'])
        live = [s_ for s_ in range(1, nobj) if objs[i][s_]['type'] != 0]
        for s_ in live:
          assert objmap[i][int(objs[i][s_]['x']), int(objs[i][s_]['y'])] == s_, f'step {t} env {i}: slot {s_} is not in the slot map where its record says'
        assert int((objmap[i] != 0).sum()) == len(live), f'step {t} env {i}: the slot map names slots that hold nothing'
  assert episodes >= 4
  env.check_errors()


def test_config4_generic_instance_on_another_view():
  """256x256 worlds seen through another view / image size: crafter_step_kernel<0, 0, 0>, the instance with nothing compiled
  in (maps and slot table in global memory, env_core.hpp FarSlot) -- into the night, auto-resets through the pool."""
  T = 230
  seeds = [42, 43, 46, 55]
  tapes = np.stack([np.random.RandomState(900 + s).choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if s % 2 == 0 else
                    np.random.RandomState(900 + s).randint(0, 17, size=T) for s in seeds], 1).astype(np.int32)
  kw = dict(area=(256, 256), view=(7, 9), size=(84, 72))
  res = oracle_rollouts([dict(kwargs=dict(seed=s, **kw), actions=tapes[:, i], snapshots=range(0, T, 45), auto_reset=True)
                         for i, s in enumerate(seeds)])
  assert max(r['night_balance_steps'] for r in res) >= 5
  env = _batched(len(seeds), seeds=seeds, auto_reset=True, **kw)
  assert not env.slot_map_derived and env.step_instance == 'crafter_step_kernel<0, 0, 0>'
  _compare(env, tapes, res, where='config4 generic')


@pytest.mark.parametrize('kernel', ['by size', 'early'])
def test_scripted_tapes_with_gifts_on_the_device(kernel, monkeypatch):
  """Rule paths a random policy all but never reaches, on the lane-parallel device code: crafting next to a table /
  furnace (objects.py:251-261), require-gated collection (objects.py:214-229), placing stone / table / furnace /
  plants (objects.py:231-249), sword damage and kills (objects.py:181-212), sleeping through the night with the
  sleep tint and wake-up logic (objects.py:99-108, engine.py:198-202).  Inventory gifts are written straight into
  the device-side record (state is caller-owned) and into the oracle at the same steps."""
  if kernel == 'early':   # crafter_step_early_kernel (by default: batches of at least 2048 envs): falling asleep and waking, placed
    monkeypatch.setenv('CRAFTER_STEP_EARLY', '1')   # and collected materials, arrows that break things -- everything its early frames must survive
    monkeypatch.setenv('CRAFTER_STEP_WIDE', '0')
  T = 330
  plan = [('builder', 3), ('builder', 4), ('sleeper', 21), ('fighter', 5), ('fighter', 6), ('fighter', 8),
          ('builder', 12), ('sleeper', 23)]
  tapes, gifts, seeds = [], [], []
  for kind, seed in plan:
    a, g = scenarios.SCENARIOS[kind](T, seed)
    tapes.append(a), gifts.append(g), seeds.append(seed)
  tapes = np.stack(tapes, 1).astype(np.int32)
  snaps = list(range(0, T, 10))
  res = oracle_rollouts([dict(kwargs=dict(seed=s), actions=tapes[:, i], gifts=gifts[i], snapshots=snaps)
                         for i, s in enumerate(seeds)])
  names = None
  reached = {}
  from crafter_amd import tables
  names = list(tables.load_rules()['achievements'])
  for (kind, _), r in zip(plan, res):
    got = {names[k] for row in r['ach'] for k, c in enumerate(row) if c}
    reached.setdefault(kind, set()).update(got)
  assert {'place_table', 'place_stone', 'place_plant', 'make_wood_pickaxe'} <= reached['builder'], reached['builder']
  assert reached['builder'] & {'place_furnace', 'make_iron_pickaxe', 'make_stone_sword', 'make_iron_sword'}
  assert reached['builder'] & {'collect_stone', 'collect_coal', 'collect_iron'}, 'require-gated collection'
  assert 'wake_up' in reached['sleeper']
  assert reached['fighter'] & {'defeat_zombie', 'eat_cow', 'defeat_skeleton'}, reached['fighter']
  env = _batched(len(seeds), seeds=seeds, auto_reset=False, semantic=True)
  _compare(env, tapes, res, gifts=gifts, where='scripted')


def test_metric_workload_4096_envs_sampled_in_place():
  """BASELINE.json metric workload (4096 envs on one GPU, seeds 1000+i, RandomState(1234) tape, auto-reset): envs
  sampled FROM the big batch -- same launch, same world-pool traffic, same queue contention as the timed run --
  against the oracle: obs, reward, done, inventory, achievements every step, full state every 50 steps."""
  n, T = 4096, 300
  sample = [0, 1, 63, 64, 511, 512, 1023, 1024, 2047, 2048, 3071, 3500, 4094, 4095]
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tapes[:, i], snapshots=range(0, T, 50), auto_reset=True)
                         for i in sample])
  assert sum(r['episodes'] for r in res) >= len(sample) // 2, 'the sample must contain auto-resets'
  env = _batched(n, seed=1000, auto_reset=True)
  _compare(env, tapes, res, index=sample, where='4096')
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['trusted'] >= 1, ps
  total = int((env.records()['episode'] - 1).sum())
  assert total > n // 2, 'most envs went through at least one auto-reset'


def test_config4_8192_envs_of_256x256_sampled_in_place():
  """BASELINE configs[3] AT ITS SIZE (VERDICT r3 weak #1a): 8192 envs x 256x256 worlds -- the generic step instance with
  the maps in HBM, the dispatch order, world-pool batches of several hundred 256x256 worlds, auto-resets through that pool
  -- with 12 envs sampled FROM the batch against the oracle: obs / reward / done / inventory / achievements every step,
  the full state (map, objects, chunk order, RNG) every 20 steps, 200 steps (through 20 balance steps, into the night)."""
  n, T = 8192, 200
  sample = [0, 1, 255, 256, 1023, 2048, 4095, 4096, 5000, 6143, 8190, 8191]
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(area=(256, 256), seed=1000 + i), actions=tapes[:, i], snapshots=range(0, T, 20), auto_reset=True)
                         for i in sample])
  assert sum(r['episodes'] for r in res) >= 3, 'the sample must contain auto-resets'
  assert max(r['night_steps'] for r in res) >= 40 and min(r['max_objects'] for r in res) > 128
  env = _batched(n, area=(256, 256), seed=1000, auto_reset=True)
  assert not env.slot_map_derived and env.step_instance == 'crafter_step_kernel<0, 2, 1>'
  _compare(env, tapes, res, index=sample, where='8192 x 256^2')
  assert env.dispatch_order() is not None, 'the timed workload runs with the dispatch order'
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] > n // 4, ps


def test_config5_16384_envs_render_off_sampled_in_place():
  """BASELINE configs[4] AT ITS SIZE: 16384 envs, render off -- the rule kernel of the split step in the lane-register
  layout, world adoption HBM -> HBM, pool batches of ~1500 worlds -- 12 envs sampled FROM the batch: reward / done /
  inventory / achievements every step, full state incl. the MT19937 key + position every 20 steps and around dusk, 290
  steps (the night's 3087 doubles per frame are drawn whoever looks at the pixels)."""
  n, T = 16384, 290
  sample = [0, 1, 63, 64, 4095, 4096, 8191, 8192, 12000, 12287, 16382, 16383]
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  snaps = sorted(set(range(0, T, 20)) | set(range(146, 152)) | set(range(270, 276)))
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tapes[:, i], snapshots=snaps, auto_reset=True) for i in sample])
  assert sum(r['episodes'] for r in res) >= len(sample) // 2 and max(r["night_steps"] for r in res) >= 40   # (an episode that reaches the night rarely survives it)
  env = _batched(n, seed=1000, auto_reset=True, render=False)
  _compare(env, tapes, res, index=sample, pixels=False, where='16384 render-off')
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] > n // 2, ps


def test_config2_1024_envs_sampled_in_place():
  """BASELINE configs[1] at its size: 1024 envs (all resident at once: no dispatch order), 12 sampled against the oracle."""
  n, T = 1024, 300
  sample = [0, 1, 63, 64, 255, 256, 511, 512, 700, 767, 1022, 1023]
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tapes[:, i], snapshots=range(0, T, 50), auto_reset=True) for i in sample])
  assert sum(r['episodes'] for r in res) >= len(sample) // 2
  env = _batched(n, seed=1000, auto_reset=True)
  _compare(env, tapes, res, index=sample, where='1024')


def test_world_pool_pipeline_generates_the_reference_worlds():
  """The world pool's three-kernel generation pipeline (seed -> compacted classification -> ordered draws) against
  Env.reset of the oracle: 32 envs x 8 episodes of 20 steps, every adopted world compared in full right after the
  auto-reset (material map, objects in slot order, chunk order, RNG key + position) and through the frames of the
  following episode.  Look-ahead of two worlds: after the first round nothing is regenerated inline."""
  n, T, length = 32, 170, 20
  seeds = [9000 + 13 * i for i in range(n)]
  tapes = np.random.RandomState(31).choice([0, 0, 1, 2, 3, 4, 5], size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=s, length=length), actions=tapes[:, i], snapshots=range(T), auto_reset=True)
                         for i, s in enumerate(seeds)])
  for r in res:   # keep only the snapshots taken right after a reset (plus the last one)
    ends = {t for t, _ in r['rows']}
    r['snapshots'] = {t: v for t, v in r['snapshots'].items() if t in ends}
  assert sum(r['episodes'] for r in res) >= 8 * n
  env = _batched(n, seeds=seeds, length=length, auto_reset=True)
  _compare(env, tapes, res, where='pool')
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] + ps['regenerated_inline'] == sum(r['episodes'] for r in res), ps
  assert ps['adopted'] >= 6 * n, ps


def test_length_none_runs_past_the_first_episode_table():
  """BatchedEnv(length=None) must not count batched steps against the daylight table (the device-side step
  counter restarts with every episode)."""
  env = _batched(4, seed=9, length=None, auto_reset=True)
  res = oracle_rollouts([dict(kwargs=dict(seed=9 + i, length=None), actions=np.zeros(40, np.int32), auto_reset=True)
                         for i in range(4)])
  _compare(env, np.zeros((40, 4), np.int32), res, where='length=None')


def test_stepping_a_finished_env_is_forgiven_by_reset():
  """Stepping a finished env past `length` (legal in the reference) raises the step-overflow status; reset() starts a
  new episode and clears it (other status bits stay sticky)."""
  from crafter_amd import CrafterDeviceError
  env = _batched(2, seed=3, length=5, auto_reset=False)
  env.reset()
  acts = torch.zeros(2, dtype=torch.int32, device=env.device)
  for _ in range(9):
    env.step(acts, info=False)
  with pytest.raises(CrafterDeviceError, match='daylight'):
    env.check_errors()
  env.reset()
  env.check_errors()
  env.step(acts, info=False)
  env.check_errors()


def test_split_step_rules_kernel_then_frame_kernel(monkeypatch):
  """The default instance as two kernels (CRAFTER_SPLIT=1): the rule kernel -- one wave per env, no cell -> slot map (object
  positions in lane registers), a window of the material map -- and the frame kernel, four waves per env from the frame
  record, night pixels through the env's scratch in global memory.  It is what runs when no frame is drawn (config 5); with
  frames the fused kernel is faster and the default.  512 envs of the metric workload through the
  first night with auto-resets, sampled against the oracle -- obs, reward, done, inventory, achievements every step, full
  state every 50."""
  monkeypatch.setenv('CRAFTER_SPLIT', '1')
  n, T = 512, 300
  sample = [0, 1, 63, 64, 100, 127, 128, 200, 255, 256, 300, 383, 384, 450, 510, 511]
  tapes = np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(seed=1000 + i), actions=tapes[:, i], snapshots=range(0, T, 50), auto_reset=True)
                         for i in sample])
  assert sum(r['night_steps'] for r in res) >= 60 and sum(r['episodes'] for r in res) >= 3, 'the sample must see night frames and auto-resets'
  env = _batched(n, seed=1000, auto_reset=True)
  _compare(env, tapes, res, index=sample, where='split')


def test_one_long_episode_past_step_1024():
  """VERDICT r2 weak #1a: ONE episode of 1200 steps on the default geometry.  The reference's default length is 10000
  (env.py:27-29); the device's day frames take rows lit at table upload for steps < 1024 (render.hpp kLitSteps) and
  light them on the fly beyond -- a path random-policy episodes (~170 steps) never reach.  Eight players kept alive by
  gifts (health / food / drink every step; energy drained in the evenings so that they sleep through parts of every
  night), noop / move / do / sleep mixed: day, night and sleeping frames on both sides of step 1024, obs every step,
  the full state every 50 steps, against the oracle (engine.py:189-202)."""
  T, seeds = 1200, [100, 114, 120, 124, 134, 142, 179, 188]
  plan = [scenarios.SCENARIOS['survivor'](T, s) for s in seeds]
  tapes = np.stack([a for a, _ in plan], 1).astype(np.int32)
  gifts = [g for _, g in plan]
  res = oracle_rollouts([dict(kwargs=dict(seed=s), actions=tapes[:, i], gifts=gifts[i], snapshots=range(0, T, 50))
                         for i, s in enumerate(seeds)])
  assert all(r['steps_played'] == T for r in res), 'every player must survive the whole tape'
  assert all(r['night_steps'] >= 450 for r in res)
  for r in res:
    assert any(s['sleeping'] for t, s in r['snapshots'].items() if t > 1024) and any(s['sleeping'] for t, s in r['snapshots'].items() if t < 1024)
  env = _batched(len(seeds), seeds=seeds, auto_reset=False)
  assert env.step_instance == 'crafter_step_kernel<1, 1, 1>'
  _compare(env, tapes, res, gifts=gifts, where='long episode')


def test_no_time_limit_outgrows_the_daylight_table(monkeypatch):
  """Env(length=None) (env.py:29,103): an episode ends with the player and may run past any table of _update_time values
  (env.py:135-139) fixed in advance -- VERDICT r3 missing #6.  BatchedEnv grows the table ahead of the longest episode
  (crafter_extend_daylight on the native handle and on a render(size) handle over the same state).  Here the table
  starts at 1100 steps: three survivors play 1400, through the growth and a night beyond it, frames every step, state
  every 50 steps and a 96x96 render() at the end against the oracle; no env ever reports a step beyond the table."""
  from crafter_amd import tables
  monkeypatch.setattr(tables, 'UNBOUNDED_DAYLIGHT', 1100)
  T, seeds = 1400, [100, 124, 142]
  plan = [scenarios.SCENARIOS['survivor'](T, s) for s in seeds]
  tapes = np.stack([a for a, _ in plan], 1).astype(np.int32)
  gifts = [g for _, g in plan]
  res = oracle_rollouts([dict(kwargs=dict(seed=s, length=None), actions=tapes[:, i], gifts=gifts[i], snapshots=range(0, T, 50))
                         for i, s in enumerate(seeds)])
  assert all(r['steps_played'] == T for r in res), 'every player must survive the whole tape'
  env = _batched(len(seeds), seeds=seeds, length=None, auto_reset=False)
  assert env.cfg.n_daylight == 1100
  env.reset()
  big = env.render(size=(96, 96))   # a second handle over the same state: it must grow with the first
  assert big.shape == (len(seeds), 96, 96, 3)
  env2 = _batched(len(seeds), seeds=seeds, length=None, auto_reset=False)   # (the compared env: no second handle, no extra render)
  _compare(env2, tapes, res, gifts=gifts, where='no time limit')
  assert env2.cfg.n_daylight > T + 2 and env2.tables.daylight.size == env2.cfg.n_daylight
  env2.check_errors()
  # the env with the second handle (reset above; its render() at step 0 drew a day frame: no noise taken from the stream):
  # same tape through the growth, then both handles draw
  names = list(env.item_names)
  for t in range(T):
    for i, g in enumerate(gifts):
      for item, amount in (g.get(t) or {}).items():
        env._rec_i32[i, env._off['inv'] + names.index(item)] = int(amount)
    env.step(torch.as_tensor(tapes[t], device=env.device), info=False)
  assert env.cfg.n_daylight == env2.cfg.n_daylight and all(h.cfg.n_daylight == env.cfg.n_daylight for h in env._aux.values())
  env.check_errors()
  assert torch.equal(env.state['mat'], env2.state['mat']) and torch.equal(env.state['objs'], env2.state['objs'])
  small, big = env.render(), env.render(size=(96, 96))
  assert int(small.sum()) > 0 and int(big.sum()) > 0
  env.check_errors()


def test_dispatch_order_is_a_permutation_with_the_slow_envs_first():
  """The step launch dispatches the envs in the order block 0 of the launch before sorted them into (night frame or balance
  step next: first).  Whatever the order, it must name every env exactly once -- checked every few steps through a night."""
  from crafter_amd import BatchedEnv, tables
  n = 4096
  env = BatchedEnv(n, seed=1000, auto_reset=True)
  env.reset()
  tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(260, n)).astype(np.int32)).to(env.device)
  day = tables.daylight_table(int(env.cfg.n_daylight))
  seen_slow = 0
  for t in range(260):
    env.step(tape[t], info=False)
    if t % 7 == 3 or t > 250:
      order = env.dispatch_order()
      assert order is not None, 'a 4096-env auto-reset batch keeps a dispatch order'
      assert np.array_equal(np.sort(order), np.arange(n)), f'step {t}: not a permutation'
      nxt = env._rec_i32[:, env._off['step']].cpu().numpy() + 1
      slow = (nxt % 10 == 0) | (day[np.minimum(nxt, len(day) - 1)] < 0.5)
      k = int(slow.sum())
      # (the order is built one step ahead: an env that reset in the step just run is filed by its old step counter)
      misfiled = int((~slow[order[:k]]).sum() + slow[order[k:]].sum())
      assert misfiled <= 0.03 * n, f'step {t}: {misfiled} envs on the wrong side of position {k}'
      seen_slow += k
  assert seen_slow > n
  small = BatchedEnv(256, seed=1, auto_reset=True)
  small.reset()
  small.step(tape[0, :256].contiguous(), info=False)
  if os.environ.get('CRAFTER_ORDER') is None:   # (CRAFTER_ORDER=1 forces the order at every batch size: the whole suite runs that way too)
    assert small.dispatch_order() is None   # fewer envs than the chip holds at once: nothing to order
  env.check_errors()


def test_slot_table_grows_before_an_object_is_refused():
  """VERDICT r4 #7.  The reference's object list has no bound (engine.py:50-58); the device's slot table has, and doubles at
  check_errors() when three quarters full (BatchedEnv._grow_objects: larger buffers, a new native handle over the same
  state, the world pool started afresh).  A deliberately small table -- 80 slots where these 64x64 worlds hold 47-55
  objects at reset and up to 83 in their first night -- stepped the way crafter_amd.Env steps (a look at the device after
  every step): the table grows in mid-episode, ST_OBJ_OVERFLOW never surfaces, and every frame / reward / inventory and the
  final state still match the oracle, which knows no slot table at all."""
  T = 300
  seeds = [35, 37, 40, 32]
  made = [scenarios.survivor_tape(T, s) for s in seeds]
  tapes = np.stack([a for a, _ in made], 1).astype(np.int32)
  gifts = [g for _, g in made]
  res = oracle_rollouts([dict(kwargs=dict(seed=s), actions=tapes[:, i], gifts=gifts[i], snapshots=[99, 199]) for i, s in enumerate(seeds)])
  assert max(r['max_objects'] for r in res) > 80 and min(r['steps_played'] for r in res) == T   # more live objects than slots
  env = _batched(len(seeds), seeds=seeds, auto_reset=False, max_objects=80)
  assert env.step_instance.endswith('<1, 0, 0>')   # (not the 256-slot default instance)
  _compare(env, tapes, res, gifts=gifts, where='growing slot table', check_every_step=True)
  assert env.objects_grown >= 1 and env.cfg.max_objects >= 160, (env.objects_grown, env.cfg.max_objects)


def test_slot_table_of_a_large_world_grows_in_global_memory():
  """The same for 256x256 worlds, whose slot table lives in global memory (env_core.hpp FarSlot: holes stay in it, squeezed
  before the host could mistake them for objects): 1000 slots where these worlds hold 750-1100 objects -- the table doubles
  at the first look at the device, the world pool starts afresh, and 120 steps with auto-resets still match the oracle."""
  T = 120
  seeds = [42, 43, 46, 55]
  tapes = np.stack([np.random.RandomState(900 + s).choice([0, 0, 0, 1, 2, 3, 4, 5, 6], size=T) if s % 2 == 0 else
                    np.random.RandomState(900 + s).randint(0, 17, size=T) for s in seeds], 1).astype(np.int32)
  res = oracle_rollouts([dict(kwargs=dict(area=(256, 256), seed=s, length=70), actions=tapes[:, i], snapshots=[39, 79, 119], auto_reset=True)
                         for i, s in enumerate(seeds)])
  assert max(r['max_objects'] for r in res) > 750
  env = _batched(len(seeds), area=(256, 256), seeds=seeds, length=70, auto_reset=True, max_objects=1000)
  assert env.step_instance == 'crafter_step_kernel<0, 2, 1>'
  _compare(env, tapes, res, where='growing slot table, 256x256', check_every_step=True)
  assert env.objects_grown >= 1 and env.cfg.max_objects >= 2000, (env.objects_grown, env.cfg.max_objects)


def test_slot_table_grows_while_the_world_pool_runs():
  """_grow_objects with auto_reset=True and the world pool running -- a new native handle over the same state in
  mid-run, pool headers and request queues cleared, batches in flight abandoned.  A deliberately small table (80 slots: it
  grows when an env holds 60 objects; these 64x64 worlds hold 36-69) over envs whose FIRST world is small, with short episodes:
  worlds are adopted from the pool, one of them crosses the mark, the table grows, and worlds keep being adopted; every
  frame / reward / done, and the full state at every episode boundary, against the oracle, which knows neither a slot table
  nor a pool."""
  n, T, length = 10, 260, 40
  cand = [7100 + 7 * i for i in range(16)]
  tapes16 = np.random.RandomState(77).choice([0, 0, 1, 2, 3, 4, 5], size=(T, 16)).astype(np.int32)
  res16 = oracle_rollouts([dict(kwargs=dict(seed=s, length=length), actions=tapes16[:, i], snapshots=range(T), auto_reset=True)
                           for i, s in enumerate(cand)])
  keep = [i for i, r in enumerate(res16) if len(r['reset_snapshot']['objects']) < 56][:n]   # (no growth before the first auto-reset)
  assert len(keep) == n
  seeds, tapes, res = [cand[i] for i in keep], np.ascontiguousarray(tapes16[:, keep]), [res16[i] for i in keep]
  assert max(r['max_objects'] for r in res) >= 62   # ... but later
  for r in res:
    ends = {t for t, _ in r['rows']}
    r['snapshots'] = {t: v for t, v in r['snapshots'].items() if t in ends}
  env = _batched(n, seeds=seeds, length=length, auto_reset=True, max_objects=80)
  env.set_timing(True)
  _compare(env, tapes, res, where='pooled growth', check_every_step=True)
  assert env.objects_grown >= 1 and env.cfg.max_objects >= 160, (env.objects_grown, env.cfg.max_objects)
  step_ms, reset_ms, launches = env.get_timing()   # the timing setting moved to the new handle with everything else
  assert launches > 0
  ps = env.pool_status()
  assert ps['state'] == 'running' and ps['adopted'] >= 3 * n, ps
