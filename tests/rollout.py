"""TEST INFRASTRUCTURE: oracle trajectories computed ahead of a GPU comparison, one process per env.

A parity test hands over, per env, the constructor arguments, the action tape, optional inventory gifts and the
steps at which it wants a full snapshot; the oracle (oracle/crafter_oracle.py, the CPU restatement pinned against
the reference) plays the tape and returns what the device results are then compared with: per-step obs hash /
reward / done / inventory / achievements, the requested snapshots, and a few facts about the trajectory the test
asserts on (did it reach the night, a balance step at night, how many objects, which achievements).
Processes, because the GPU box has many host cores and the GPU budget is wall-clock.
"""
import multiprocessing as mp
import os

import numpy as np

from tests.parity import sha8


def _play(spec):
  from oracle.crafter_oracle import OracleEnv
  kw = dict(spec.get('kwargs', {}))
  env = OracleEnv(**kw)
  acts = np.asarray(spec['actions'])
  gifts = spec.get('gifts') or {}
  snaps_at = set(int(t) for t in spec.get('snapshots', ()))
  frames_at = set(int(t) for t in spec.get('frames', ()))
  auto_reset = bool(spec.get('auto_reset', False))
  reset_at = set(int(t) for t in spec.get('reset_at', ()))   # a manual Env.reset() right after these steps
  extra_render = spec.get('render_each_step')   # (w, h): call render(size) after every step, like VideoRecorder
  names = list(env.t.items)
  out = {'obs_sha': [], 'reward': [], 'done': [], 'inv': [], 'ach': [], 'snapshots': {}, 'frames': {},
         'extra_sha': [], 'max_objects': 0, 'night_steps': 0, 'night_balance_steps': 0, 'episodes': 0,
         'rows': [], 'manual_reset_obs': {}, 'manual_reset_snapshot': {}}
  obs = env.reset()
  out['reset_obs'] = obs.copy()
  out['reset_snapshot'] = env.snapshot()
  ep_len, ep_reward = 0, 0.0
  for t, a in enumerate(acts):
    if t in gifts:
      for item, amount in gifts[t].items():
        env.inv[names.index(item)] = amount
    obs, r, d, info = env.step(int(a))
    ep_len += 1
    ep_reward += info['reward']
    night = env.daylight < 0.5
    out['night_steps'] += int(night)
    out['night_balance_steps'] += int(night and env._step % 10 == 0)
    out['max_objects'] = max(out['max_objects'], sum(1 for v in env.otype[1:] if v))
    out['reward'].append(np.float32(r))
    out['done'].append(bool(d))
    out['inv'].append([int(v) for v in info['inventory'].values()])
    out['ach'].append([int(v) for v in info['achievements'].values()])
    if extra_render is not None:
      out['extra_sha'].append(sha8(env.render(tuple(extra_render))))
    if t in snaps_at and not (d and auto_reset):
      out['snapshots'][t] = env.snapshot()
    if d:   # recorder.py:53-66 row of the finished episode
      row = {'length': ep_len, 'reward': round(ep_reward, 1)}
      row.update({f'achievement_{k}': int(v) for k, v in info['achievements'].items()})
      out['rows'].append((t, row))
      out['episodes'] += 1
      ep_len, ep_reward = 0, 0.0
      if auto_reset:
        out['terminal_sha'] = out.get('terminal_sha', []) + [(t, sha8(obs))]
        obs = env.reset()
        if t in snaps_at:   # the device state after this step is already the next episode's
          out['snapshots'][t] = env.snapshot()
      else:
        out['obs_sha'].append(sha8(obs))
        if t in frames_at:
          out['frames'][t] = obs.copy()
        out['stopped_at'] = t
        break
    out['obs_sha'].append(sha8(obs))
    if t in frames_at:
      out['frames'][t] = obs.copy()
    if t in reset_at:   # the caller resets in the middle of an episode (env.py:70-81: the episode counter moves on)
      obs = env.reset()
      ep_len, ep_reward = 0, 0.0
      out['manual_reset_obs'][t] = obs.copy()
      out['manual_reset_snapshot'][t] = env.snapshot()
  out['final_snapshot'] = env.snapshot()
  out['steps_played'] = len(out['reward'])
  return out


def oracle_rollouts(specs, workers=None):
  """specs: list of dicts {kwargs, actions, gifts?, snapshots?, frames?, auto_reset?, render_each_step?}."""
  workers = workers or min(len(specs), max(1, len(os.sched_getaffinity(0))))
  if workers <= 1 or len(specs) == 1:
    return [_play(s) for s in specs]
  ctx = mp.get_context('fork')   # the children only run numpy; they never touch the HIP runtime of the parent
  with ctx.Pool(workers, initializer=_worker_init) as pool:
    # a worker that dies (it must not, but a hang here costs GPU-box minutes) surfaces as a timeout, not as a wait forever
    return pool.map_async(_play, specs, chunksize=1).get(timeout=600)


def _worker_init():
  import gc
  gc.disable()   # nothing inherited from the parent (GPU handles, tensors) is ever finalised in a worker
