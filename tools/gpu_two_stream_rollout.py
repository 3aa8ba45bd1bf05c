"""GPU box experiment: do two rollout streams with staggered stretch boundaries fill each other's launch tails?  Two handles
of N/2 envs each on two torch streams (each with its own world pool), the second one 8 steps ahead, against one handle of N.
usage: python tools/gpu_two_stream_rollout.py [envs] [stagger]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
stagger = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T, calls, burn = 64, 24, 400
dev = torch.device('cuda', 0)


def make(m, seed0):
  env = BatchedEnv(m, seeds=[seed0 + i for i in range(m)], auto_reset=True)
  tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(burn + (calls + 3) * T + 64, m)).astype(np.int32)).cuda()
  env.reset()
  for t in range(burn):
    env.step(tape[t], info=False)
  out = (torch.empty((T,) + tuple(env.obs.shape), dtype=torch.uint8, device=dev), torch.empty((T, m), dtype=torch.float32, device=dev),
         torch.empty((T, m), dtype=torch.uint8, device=dev))
  return env, tape, out


one, tape1, out1 = make(n, 1000)
t = burn
for _ in range(2):
  one.rollout(tape1[t:t + T], out=out1); t += T
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
  one.rollout(tape1[t:t + T], out=out1); t += T
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'one handle of {n}: {calls * T * n / dt / 1e6:.2f} M env-steps/s ({1e6 * dt / (calls * T):.1f} us per step)')
del one, tape1, out1
torch.cuda.synchronize()

for stag in (0, stagger):
  a, tapea, outa = make(n // 2, 1000)
  b, tapeb, outb = make(n // 2, 1000 + n // 2)
  sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
  torch.cuda.synchronize()
  ta = tb = burn
  with torch.cuda.stream(sa):
    a.rollout(tapea[ta:ta + T], out=outa); ta += T
  with torch.cuda.stream(sb):
    b.rollout(tapeb[tb:tb + T], out=outb); tb += T
    if stag:
      b.rollout(tapeb[tb:tb + stag]); tb += stag
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(calls):
    with torch.cuda.stream(sa):
      a.rollout(tapea[ta:ta + T], out=outa); ta += T
    with torch.cuda.stream(sb):
      b.rollout(tapeb[tb:tb + T], out=outb); tb += T
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  print(f'two handles of {n // 2} on two streams, second {stag} steps ahead: {calls * T * n / dt / 1e6:.2f} M env-steps/s ({1e6 * dt / (calls * T):.1f} us per step)', a.pool_status()['regenerated_inline'], b.pool_status()['regenerated_inline'])
  a.check_errors(); b.check_errors()
  del a, b, tapea, tapeb, outa, outb
  torch.cuda.synchronize()
