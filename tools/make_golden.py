#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the UNTOUCHED reference (/root/reference, imported through
oracle/reference_harness.py: 3 import shims + slot-order canonicalisation).

The reference has no tests, fixtures or golden vectors of its own, so these files are the pin:
each holds, for one (seed, area, length, tape) case, what the real ``crafter.Env`` produced --
the reset state, per-step reward/done/hashes, sampled full frames and the final state.
tests/test_golden.py replays them against the oracle (CPU, here and on the GPU box) and
tests/test_gpu_golden.py against the HIP path.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import hashlib
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True

from oracle import reference_harness as rh  # noqa: E402

OUT = ROOT / 'tests' / 'golden'

# (name, seed, area, length, n_steps, tape kind)
CASES = [
    ('rand_s0', 0, (64, 64), 10000, 400, 'uniform'),
    ('rand_s1', 1, (64, 64), 10000, 400, 'uniform'),
    ('rand_s12345', 12345, (64, 64), 10000, 600, 'uniform'),
    ('night_s7', 7, (64, 64), 10000, 320, 'noop_heavy'),     # survives into the night (steps 148-272)
    ('short_s3', 3, (64, 64), 50, 120, 'uniform'),           # length limit -> done by `over`
    ('area32_s5', 5, (32, 32), 10000, 200, 'uniform'),       # small map: edges, partial chunks
    ('area80x48_s9', 9, (80, 48), 10000, 150, 'uniform'),    # non-square, chunk remainder 8 / 0
    ('collect_s2', 2, (64, 64), 10000, 400, 'do_heavy'),
]


def tape(kind, n, seed):
  rs = np.random.RandomState(1234 + seed)
  if kind == 'uniform':
    return rs.randint(0, 17, size=n)
  if kind == 'noop_heavy':   # mostly noop / sleep so the player survives long
    return rs.choice([0, 0, 0, 6, 1, 2, 3, 4, 5], size=n)
  if kind == 'do_heavy':     # move + do + place + make
    return rs.choice([1, 2, 3, 4, 5, 5, 5, 8, 10, 11, 14, 7, 9], size=n)
  raise ValueError(kind)


def sha(a):
  return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:8], np.uint64)[0]


def ref_state(env):
  from crafter import objects as robj
  types = {robj.Player: 1, robj.Cow: 2, robj.Zombie: 3, robj.Skeleton: 4, robj.Arrow: 5, robj.Plant: 6}
  objs = []
  for o in env._world._objects:
    if o is None:
      continue
    t = types[type(o)]
    f = tuple(int(v) for v in getattr(o, 'facing', (0, 0)))
    aux = {3: getattr(o, 'cooldown', 0), 4: getattr(o, 'reload', 0), 6: getattr(o, 'grown', 0)}.get(t, 0)
    objs.append((t, int(o.pos[0]), int(o.pos[1]), int(o.health), f[0], f[1], int(aux)))
  p = env._player
  kind, key, pos = env._world.random.get_state()[:3]
  return dict(
      mat=env._world._mat_map.copy(), objects=np.array(objs, np.int32),
      inventory=np.array(list(p.inventory.values()), np.int32),
      achievements=np.array(list(p.achievements.values()), np.int32),
      misc=np.array([int(p.sleeping), int(round(p._hunger * 2)), int(round(p._thirst * 2)),
                     int(round(p._fatigue * 2)), int(round(p._recover * 2)), int(p._last_health)], np.int32),
      chunk_order=np.array(list(env._world._chunks.keys()), np.int32).reshape(-1, 4),
      mt_key=np.array(key, np.uint32), mt_pos=np.int32(pos))


def main():
  crafter = rh.load()
  OUT.mkdir(parents=True, exist_ok=True)
  for name, seed, area, length, n, kind in CASES:
    env = crafter.Env(area=area, length=length, seed=seed)
    acts = tape(kind, n, seed).astype(np.int32)
    out = {'seed': np.int64(seed), 'area': np.array(area, np.int32), 'length': np.int32(length), 'actions': acts}
    obs = env.reset()
    out['reset_obs'] = obs
    for k, v in ref_state(env).items():
      out['reset_' + k] = v
    rewards, dones, obs_sha, sem_sha, frames, frame_steps, episodes = [], [], [], [], [], [], []
    for t, a in enumerate(acts):
      obs, reward, done, info = env.step(int(a))
      rewards.append(reward)
      dones.append(bool(done))
      obs_sha.append(sha(obs))
      sem_sha.append(sha(info['semantic']))
      if t % 40 == 0 or (148 <= env._step <= 272 and t % 15 == 0):
        frames.append(obs)
        frame_steps.append(t)
      if done:
        obs = env.reset()          # the replay does the same: reset right after done
        episodes.append(t)
        obs_sha[-1] = sha(obs)     # hash of what an auto-resetting caller sees
    out.update(rewards=np.array(rewards, np.float64), dones=np.array(dones, np.uint8),
               obs_sha=np.array(obs_sha, np.uint64), sem_sha=np.array(sem_sha, np.uint64),
               frames=np.array(frames, np.uint8), frame_steps=np.array(frame_steps, np.int32),
               reset_at=np.array(episodes, np.int32))
    for k, v in ref_state(env).items():
      out['final_' + k] = v
    np.savez_compressed(OUT / f'{name}.npz', **out)
    print(f'{name}: {n} steps, resets at {episodes}, night frames '
          f'{sum(1 for s in frame_steps if s % 40)}; {(OUT / (name + ".npz")).stat().st_size} bytes')


if __name__ == '__main__':
  main()
