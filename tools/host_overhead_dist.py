#!/usr/bin/env python3
"""GPU box: host-side cost of the FULL loop body of an N > 1 run (bench.py --gpus N) -- StepExchange.begin + BatchedEnv.step(out=
the send record) + StepExchange.launch + the late result() -- on the `nccl` backend at world size 1 (one GPU per box: the
collective moves nothing, its enqueue cost is what is measured), for the three exchange modes, at configs[2]'s shard size.
If the host needs longer per step than the GPU (~33 us at 512 envs), the 8-GPU run is host-bound whatever xGMI does.
Round 5: the same per-step loop body enqueued from C (NativeStepExchange / crafter_step_exchange).
usage: tools/host_overhead_dist.py [envs]"""
import os, socket, sys, time, pathlib
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv
from crafter_amd import dist as cdist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
with socket.socket() as s:
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
K = 2000
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(K + 300, n)).astype(np.int32)).to(dev)
env = BatchedEnv(n, seed=1000, device=dev, auto_reset=True)
env.reset()
for t in range(300):
  env.step(tape[t], info=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(K):
  env.step(tape[300 + t], info=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{n} envs, no exchange:        host {1e6 * (t1 - t0) / K:6.1f} us/step, drained after {1e6 * (t2 - t0) / K:6.1f} us/step')
for mode in cdist.StepExchange.MODES:
  ex = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device=dev, mode=mode, dst=0)
  for t in range(100):
    slot = ex.begin(t)
    env.step(tape[t], info=False, out=ex.outputs(slot))
    ex.launch(slot)
  ex.finish()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(K):
    slot = ex.begin(t)
    env.step(tape[300 + t], info=False, out=ex.outputs(slot))
    ex.launch(slot)
    if t >= 1:
      ex.result(t - 1)
  t1 = time.perf_counter()
  ex.finish()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print(f'{n} envs, exchange {mode:9s}: host {1e6 * (t1 - t0) / K:6.1f} us/step, drained after {1e6 * (t2 - t0) / K:6.1f} us/step')
ex = cdist.NativeStepExchange(env)   # the same loop body as ONE call into the library (crafter_step_exchange)
for t in range(100):
  ex.step(t, tape[t])
ex.finish()
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(K):
  ex.step(t, tape[300 + t])
  if t >= 1:
    ex.result(t - 1)
t1 = time.perf_counter()
ex.finish()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{n} envs, exchange enqueued from C (crafter_step_exchange, per step): host {1e6 * (t1 - t0) / K:6.1f} us/step, drained after {1e6 * (t2 - t0) / K:6.1f} us/step')
ex.close()
for K in (4, 16):   # K steps' records per collective: the enqueue cost is paid once per K steps
  ex = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device=dev, mode='allgather', dst=0, steps=K)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(K * (2000 // K)):
    slot = ex.begin(t)
    env.step(tape[300 + t], info=False, out=ex.outputs(slot))
    ex.launch(slot)
  t1 = time.perf_counter()
  ex.finish()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print(f'{n} envs, allgather, {K:2d} steps per collective: host {1e6 * (t1 - t0) / (K * (2000 // K)):6.1f} us/step, drained after {1e6 * (t2 - t0) / (K * (2000 // K)):6.1f} us/step')
ex = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device=dev, mode='scalars')
a = torch.zeros(n, dtype=torch.int32, device=dev)
for how in ('broadcast', 'scatter'):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(K):
    ex.scatter_actions(a, src=0, how=how)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  print(f'scatter_actions({how}): host {1e6 * (t1 - t0) / K:6.1f} us/call')
dist.destroy_process_group()
