"""GPU box: a process's second (third, ...) handle.  Steps/s of a 1024-env batch created AFTER a 4096-env batch that stays alive (what
bench.py's extra lines do), with inline regeneration beside the launch and behind it.  (r4zz_handles_dedicated.txt: CRAFTER_AUX_DEDICATED=1 was a build whose
regeneration stream came from hipExtStreamCreateWithCUMask with a full mask -- measured, removed.)  usage: python tools/gpu_two_handles.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crafter_amd import BatchedEnv


def rate(env, steps=1500, warm=300):
  a = torch.randint(0, 17, (steps + warm, env.num_envs), dtype=torch.int32, device=env.device)
  for t in range(warm):
    env.step(a[t], info=False)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(warm, warm + steps):
    env.step(a[t], info=False)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  return env.num_envs * steps / dt / 1e6, dt / steps * 1e6


for beside, dedicated in (('1', '0'), ('1', '1'), ('0', '0')):
  os.environ['CRAFTER_REGEN_BESIDE'] = beside
  os.environ['CRAFTER_AUX_DEDICATED'] = dedicated
  first = BatchedEnv(4096, seed=1, auto_reset=True)
  first.reset()
  r1 = rate(first, 600, 100)
  second = BatchedEnv(1024, seed=9000, auto_reset=True)
  second.reset()
  r2 = rate(second)
  third = BatchedEnv(512, seed=19000, auto_reset=True)
  third.reset()
  r3 = rate(third)
  more = [BatchedEnv(1024, seed=30000 + 5000 * k, auto_reset=True) for k in range(3)]
  for e in more:
    e.reset()
  rm = [rate(e, 800, 200) for e in more]
  r1b = rate(first, 600, 100)
  print(f'CRAFTER_REGEN_BESIDE={beside} CRAFTER_AUX_DEDICATED={dedicated}: first handle (4096 envs) {r1[0]:.2f} M, {r1[1]:.1f} us/step; second (1024) {r2[0]:.2f} M, {r2[1]:.1f} us/step; '
        f'third (512) {r3[0]:.2f} M, {r3[1]:.1f} us/step; first again {r1b[0]:.2f} M; three more of 1024: ' + ', '.join(f'{r[1]:.1f} us' for r in rm), flush=True)
  del first, second, third, more
