"""``python -m crafter_amd.run_random`` -- the reference's ``crafter/run_random.py`` (lines 10-44) on the
MI355X path: same flags, same prints (reset time, material counts, step time / FPS, episode length).
``--envs N`` (not in the reference) runs N environments at once through BatchedEnv."""
import argparse
import copy
import time

import numpy as np


def main():
  parser = argparse.ArgumentParser()
  parser.add_argument('--seed', type=int, default=None)
  parser.add_argument('--area', nargs=2, type=int, default=(64, 64))
  parser.add_argument('--length', type=int, default=10000)
  parser.add_argument('--health', type=int, default=9)
  parser.add_argument('--record', type=str, default=None)
  parser.add_argument('--episodes', type=int, default=1)
  parser.add_argument('--envs', type=int, default=1)
  args = parser.parse_args()

  import torch
  from . import BatchedEnv, Env, tables
  from .recorder import BatchedStatsRecorder, EnvStatsRecorder
  rules = copy.deepcopy(tables.load_rules())
  rules['items']['health']['max'] = args.health       # run_random.py:21-22
  rules['items']['health']['initial'] = args.health
  random = np.random.RandomState(args.seed)

  if args.envs == 1:
    env = Env(area=tuple(args.area), length=args.length, seed=args.seed, rules=rules)
    if args.record:   # run_random.py:24: crafter.Recorder(env, args.record) -> stats.jsonl
      env = EnvStatsRecorder(env, args.record)
    for _ in range(args.episodes):
      start = time.time()
      env.reset()
      print('')
      print(f'Reset time: {1000 * (time.time() - start):.2f}ms')
      print('Coal exist:    ', env._world.count('coal'))
      print('Iron exist:    ', env._world.count('iron'))
      print('Diamonds exist:', env._world.count('diamond'))
      start = time.time()
      done = False
      while not done:
        action = random.randint(0, env.action_space.n)
        _, _, done, _ = env.step(action)
      duration = time.time() - start
      step = env._step
      print(f'Step time: {1000 * duration / step:.2f}ms ({int(step / duration)} FPS)')
      print('Episode length:', step)
    return

  seed = 0 if args.seed is None else args.seed
  env = BatchedEnv(args.envs, area=tuple(args.area), length=args.length, seed=seed, rules=rules, auto_reset=True)
  if args.record:
    env = BatchedStatsRecorder(env, args.record)
  start = time.time()
  env.reset()
  torch.cuda.synchronize()
  print(f'Reset time: {1000 * (time.time() - start):.2f}ms for {args.envs} envs')
  finished, steps = 0, 0
  start = time.time()
  while finished < args.episodes * args.envs:
    actions = torch.from_numpy(random.randint(0, env.num_actions, size=args.envs).astype(np.int32)).to(env.device)
    _, _, done, _ = env.step(actions, info=False)
    finished += int(done.sum())
    steps += args.envs
  torch.cuda.synchronize()
  duration = time.time() - start
  env.check_errors()
  print(f'Step time: {1000 * duration / (steps / args.envs):.3f}ms per batched step ({int(steps / duration)} env-steps/s)')
  print('Episodes finished:', finished)


if __name__ == '__main__':
  main()
