#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched Crafter hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one BatchedEnv.step() over every env of every rank: Env.step (dynamics, balance,
reward/done, 64x64x3 render) with auto-reset (Env.reset worldgen) of finished envs, uniform random
actions from a device-resident tape.  Workload at N=1 = BASELINE.json
configs[1]: 1024 envs, 64x64 world, 64x64x3 obs.  For N>1 every rank owns --envs-per-gpu envs (weak
scaling, envs shard by index, seeds 1000+global index) and the per-step exchange named by the
north star -- an RCCL all-gather of reward/done (and obs with --gather-obs) -- runs on a side stream
overlapped with the next step.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (crafter_step_kernel): algorithmic bytes/launch (19,742 B per
                  env-step) / mean kernel duration measured with HIP start/stop events
                  attached to the kernel on the launch stream, against the 8 TB/s HBM peak
  cpu_baseline -- the CPU port (oracle/crafter_oracle.py) timed on this host's cores on a bounded
                  sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ALGO_BYTES_PER_ENV_STEP = 19742   # 64x64 world, render on
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(num_envs, seconds=12.0):
  """Times the CPU port on len(sched_getaffinity) processes: each steps its own envs with the same
  seeds/action tape convention (seed 1000+i, RandomState(1234) tape, reset after done)."""
  import multiprocessing as mp
  cores = len(os.sched_getaffinity(0))
  procs = max(1, min(cores, num_envs))
  ctx = mp.get_context('fork')
  q = ctx.Queue()

  def worker(rank):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle.crafter_oracle import OracleEnv
    envs = [OracleEnv(seed=1000 + i) for i in range(rank, min(num_envs, procs * 2), procs)]
    tape = np.random.RandomState(1234).randint(0, 17, size=(100000,)).astype(np.int32)
    for e in envs:
      e.reset()
    steps, t0, t = 0, time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
      for e in envs:
        _, _, done, _ = e.step(int(tape[t % len(tape)]))
        steps += 1
        if done:
          e.reset()
      t += 1
    q.put((steps, time.perf_counter() - t0))

  ps = [ctx.Process(target=worker, args=(r,)) for r in range(procs)]
  for p in ps:
    p.start()
  res = [q.get() for _ in ps]
  for p in ps:
    p.join()
  total = sum(s / dt for s, dt in res)
  return {'value': total, 'unit': 'env-steps/s', 'cores': procs, 'kind': 'port',
          'sample': f'{procs} processes x 2 envs of the CPU port (oracle/crafter_oracle.py, C noise helper), '
                    f'{seconds:.0f} s wall, random actions, resets included'}


def parity_sample(tape_np, n, dev, steps=300):
  """Bit-exactness spot check inside the bench run: a sample of the benchmark's own envs
  (same seeds, same action tape, auto-reset) replayed on the GPU and by the CPU port, compared on obs,
  reward and done at every step and on the full state at the end.  Envs are independent, so a fresh
  small batch reproduces exactly what those envs did inside the big one."""
  import torch
  from crafter_amd import BatchedEnv
  from oracle.crafter_oracle import OracleEnv
  from tests.parity import assert_same
  sample = sorted({0, 1, n // 2, n - 1})
  steps = min(steps, tape_np.shape[0])
  env = BatchedEnv(len(sample), seeds=[1000 + i for i in sample], device=dev, auto_reset=True)
  orcs = [OracleEnv(seed=1000 + i) for i in sample]
  ok = True
  try:
    obs = env.reset().cpu().numpy()
    for k, o in enumerate(orcs):
      ok &= bool(np.array_equal(obs[k], o.reset()))
    for t in range(steps):
      acts = tape_np[t, sample]
      obs, rew, done, _ = env.step(torch.from_numpy(np.ascontiguousarray(acts)).to(dev), info=False)
      obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
      for k, o in enumerate(orcs):
        ob, r, d, _ = o.step(int(acts[k]))
        if d:
          ob = o.reset()
        ok &= bool(np.array_equal(obs[k], ob)) and rew[k] == np.float32(r) and bool(done[k]) == bool(d)
    env.check_errors()
    for k, o in enumerate(orcs):
      assert_same(env.snapshot(k), o.snapshot(), f'env {sample[k]}')
  except AssertionError:
    ok = False
  return {'bit_exact': bool(ok), 'envs': sample, 'steps': steps,
          'checked': 'obs, reward, done every step (auto-reset included); full state + RNG at the end; vs the CPU port'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2000)
  ap.add_argument('--warmup', type=int, default=200)
  ap.add_argument('--envs-per-gpu', type=int, default=1024)
  ap.add_argument('--area', type=int, default=64)
  ap.add_argument('--no-render', action='store_true')
  ap.add_argument('--gather-obs', action='store_true')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-seconds', type=float, default=12.0)
  ap.add_argument('--gen-period', type=int, default=0)
  args = ap.parse_args()

  import torch
  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  if args.gpus != world and world > 1:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
  ndev = torch.cuda.device_count()
  dev = torch.device('cuda', local_rank % max(ndev, 1))
  torch.cuda.set_device(dev)
  dist = None
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # RCCL ("nccl") over xGMI is the product path; CRAFTER_BENCH_BACKEND=gloo only exists so that the
    # multi-process plumbing can be exercised on a box with a single GPU.
    backend = os.environ.get('CRAFTER_BENCH_BACKEND', 'nccl')
    kw = {'device_id': dev} if backend == 'nccl' else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)

  from crafter_amd import BatchedEnv
  from crafter_amd import dist as cdist
  n = args.envs_per_gpu
  seeds = cdist.shard_seeds(1000, world * n, rank, world)   # global env index -> seed 1000 + index
  env = BatchedEnv(n, area=(args.area, args.area), seeds=seeds, device=dev,
                   auto_reset=True, render=not args.no_render, gen_period=args.gen_period)
  total = args.warmup + args.steps
  tape_np = np.random.RandomState(1234).randint(0, 17, size=(total, world * n)).astype(np.int32)
  tape = cdist.shard_actions(torch.from_numpy(tape_np), rank, world).contiguous().to(dev)
  env.reset()

  # Per-step exchange named by the north star: all-gather of reward/done (and obs with --gather-obs).
  # Double-buffered: step t copies its outputs into slot t % 2 on the launch stream, the gather of that
  # slot runs on a side stream and overlaps step t + 1; slot reuse waits on the gather's event.
  side = torch.cuda.Stream(device=dev) if world > 1 else None
  if world > 1:
    slots = []
    for _ in range(2):
      slots.append({
          'rd': torch.zeros((n, 2), dtype=torch.float32, device=dev),
          'obs': torch.zeros_like(env.obs) if args.gather_obs else None,
          'g_rd': torch.zeros((world * n, 2), dtype=torch.float32, device=dev),
          'g_obs': torch.zeros((world * n,) + tuple(env.obs.shape[1:]), dtype=torch.uint8, device=dev) if args.gather_obs else None,
          'done': None})

  def run(t):
    env.step(tape[t], info=False)
    if side is None:
      return
    s = slots[t & 1]
    main = torch.cuda.current_stream(dev)
    if s['done'] is not None:
      main.wait_event(s['done'])
    s['rd'][:, 0].copy_(env.reward)
    s['rd'][:, 1].copy_(env.done)
    if s['obs'] is not None:
      s['obs'].copy_(env.obs)
    side.wait_stream(main)
    with torch.cuda.stream(side):
      dist.all_gather_into_tensor(s['g_rd'], s['rd'])
      if s['obs'] is not None:
        dist.all_gather_into_tensor(s['g_obs'], s['obs'])
      ev = torch.cuda.Event()
      ev.record(side)
      s['done'] = ev

  for t in range(args.warmup):
    run(t)
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  for t in range(args.warmup, total):
    run(t)
  ev1.record()
  if side is not None:
    torch.cuda.current_stream(dev).wait_stream(side)
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
  dt = time.perf_counter() - t0
  gpu_ms = ev0.elapsed_time(ev1)
  env.check_errors()
  if dist is not None:
    tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
    dt = float(tdt.item())

  # dominant kernel: mean duration of crafter_step_kernel (and of the auto-reset kernel that
  # follows it in the same call), HIP events recorded by the library on the launch stream
  kern_us = reset_us = None
  if rank == 0:
    reps = min(300, args.steps)
    env.set_timing(True)
    for i in range(reps):
      env.step(tape[args.warmup + i], info=False)
    step_ms, reset_ms, launches = env.get_timing()
    env.set_timing(False)
    kern_us = 1000.0 * step_ms / launches
    reset_us = 1000.0 * reset_ms / launches

  traffic = None
  if rank == 0:
    # HBM bytes per launch from the latest committed PMC passes (profiles/*_hbm_traffic.json,
    # tools/summarize_profile.py); only quoted when it was collected on this exact workload.
    import glob
    import pathlib
    files = sorted(glob.glob(str(pathlib.Path(__file__).resolve().parent / 'profiles' / '*_hbm_traffic.json')))
    if files and not args.no_render and args.area == 64:
      tj = json.load(open(files[-1]))
      t = tj.get('crafter_step_kernel<1, 1, 1>') or tj.get('crafter_step_kernel<1, 1>') or tj.get('crafter_step_kernel<1>') or tj.get('crafter_step_kernel')
      if t and t.get('grid_threads') == n * t.get('workgroup', 256):
        traffic = t['hbm_bytes_per_launch']

  if rank == 0:
    value = args.steps * n * world / dt
    bytes_per_launch = ALGO_BYTES_PER_ENV_STEP * n if not args.no_render else 7454 * n
    achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9 if kern_us else None
    line = {
        'metric': 'env-steps/sec (whole node), random policy', 'value': value, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8/i32 (+f64 render filters)',
        'data': 'synthetic',
        'config': {'workload': f'{n} envs/GPU x {world} GPU, {args.area}x{args.area} world, view 9x9, '
                               f'obs 64x64x3, random actions, auto-reset, render {"off" if args.no_render else "on"}',
                   'envs_per_gpu': n, 'parallelism': f'env-index sharding x{world}',
                   'exchange': None if world == 1 else ('all_gather reward/done' + ('+obs' if args.gather_obs else ''))},
        'gpu_ms_per_step': gpu_ms / args.steps,
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': (achieved / HBM_PEAK_GBS) if achieved else None, 'traffic': traffic,
                     'kernel': 'crafter_step_kernel', 'kernel_us': kern_us, 'reset_kernel_us': reset_us, 'algorithmic_bytes_per_launch': bytes_per_launch},
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(n, args.cpu_seconds)
      if args.area == 64 and not args.no_render:
        line['parity'] = parity_sample(tape_np, n, dev)
    print(json.dumps(line))
  if dist is not None:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
