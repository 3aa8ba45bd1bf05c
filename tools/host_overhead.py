#!/usr/bin/env python3
"""GPU box: host-side cost of one BatchedEnv.step() call (Python + ctypes + HIP launches), measured
with a small batch so the GPU is never the bottleneck.  usage: tools/host_overhead.py [envs]"""
import sys, time, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv
import ctypes as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
env = BatchedEnv(n, seed=1, auto_reset=True)
env.reset()
a = torch.zeros(n, dtype=torch.int32, device='cuda')
for _ in range(200):
  env.step(a, info=False)
torch.cuda.synchronize()
K = 3000
t0 = time.perf_counter()
for _ in range(K):
  env.step(a, info=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{n} envs: ', end='')
print(f'step() host time {1e6 * (t1 - t0) / K:.1f} us/call; drained after {1e6 * (t2 - t1) / K:.1f} us/call more')
# raw C call only
lib, h = env._lib, env._handle
args = (h, C.c_void_p(a.data_ptr()), C.c_void_p(env.obs.data_ptr()), C.c_void_p(env.reward.data_ptr()),
        C.c_void_p(env.done.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
  lib.crafter_step(*args)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f'raw crafter_step() {1e6 * (t1 - t0) / K:.1f} us/call')
t0 = time.perf_counter()
for _ in range(K):
  with torch.cuda.device(env.device):
    pass
t1 = time.perf_counter()
print(f'torch.cuda.device ctx {1e6 * (t1 - t0) / K:.1f} us; ', end='')
t0 = time.perf_counter()
for _ in range(K):
  torch.cuda.current_stream(env.device).cuda_stream
t1 = time.perf_counter()
print(f'current_stream {1e6 * (t1 - t0) / K:.1f} us; ', end='')
t0 = time.perf_counter()
for _ in range(K):
  torch._C._cuda_getCurrentRawStream(0)
t1 = time.perf_counter()
print(f'raw stream {1e6 * (t1 - t0) / K:.2f} us')
