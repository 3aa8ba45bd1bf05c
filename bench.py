#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the batched Crafter hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one BatchedEnv.step() over every env of every rank: Env.step (dynamics, balance, reward/done, 64x64x3
render) with auto-reset (Env.reset worldgen) of finished envs, uniform random actions from a device-resident tape
(seeds 1000 + global env index, RandomState(1234) tape).

Workload = the one BASELINE.json's metric is quoted on: 4096 envs, 64x64 world, 64x64x3 obs.
  N = 1   all 4096 envs on the one GPU (they fit);  configs[1] (1024 envs) is measured too and reported under "extra"
  N > 1   configs[2]: the 4096 envs shard by index over the N ranks (4096 / N each, "scaling": "strong") and the steps
          feed the exchange the north star names: ONE all-gather of each rank's packed (obs, reward, done) records
          over RCCL / xGMI PER STEP (the workload the metric names), double-buffered so that it overlaps the steps that
          follow (crafter_amd.dist.StepExchange); `value` is that.  The same run then repeats the sustained window with 16
          steps' records per collective (`exchange_blocks_of_16`: the host enqueues one collective per 16 steps -- a learner sees
          frames up to 16 steps late; reported beside the headline, never as it).
          --envs-per-gpu M switches to weak scaling (M envs on every rank).

Measurement protocol (so that a short --steps window is representative): after reset() the batch runs an UNTIMED
burn-in (--burn-in, default 400 steps: envs desynchronise, the first night and the first wave of auto-resets pass, the
world pool is warm), then --warmup untimed steps, then EXACTLY --steps timed steps between barrier + synchronize pairs
(max over ranks).  The dominant kernel's duration is then measured over --kernel-reps (default 300) further launches
with HIP start / stop events attached to the kernel itself on the launch stream, whatever --steps was.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline     dominant kernel (crafter_step_kernel): algorithmic bytes per launch (19,742 B per env-step)
               / mean kernel duration, against the 8 TB/s HBM peak; "traffic" = PMC-counter HBM bytes per launch quoted
               from the committed profile named in "traffic_source" (not measured by this run), null if that profile
               is of another workload
  cpu_baseline the CPU port (oracle/crafter_oracle.py) timed on this host's cores on a bounded sample of the same
               workload (rank 0, N = 1 only), with the port-vs-real-reference per-core factor measured in the build
               container (profiles/*_cpu_calibration.json, tools/calibrate_cpu_baseline.py)
  parity       envs sampled FROM THE TIMED BATCH against the oracle replaying the same seeds and tape: obs / reward /
               done at every burn-in step, full state + RNG + frame after the last timed step
"""
import argparse
import glob
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
ALGO_BYTES = {True: 19742, False: 7454}   # per env-step, 64x64 world, render on / off
ALGO_BYTES_256 = 34118                    # 256x256 world, render on
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8 TB/s
METRIC_ENVS = 4096                        # BASELINE.json metric: "... random policy, 4096 envs"


def cpu_baseline(num_envs, seconds=12.0):
  """Times the CPU port on len(sched_getaffinity) processes: each steps its own envs with the same
  seeds/action tape convention (seed 1000+i, RandomState(1234) tape, reset after done)."""
  import multiprocessing as mp
  cores = len(os.sched_getaffinity(0))
  procs = max(1, min(cores, num_envs))
  ctx = mp.get_context('fork')
  q = ctx.Queue()

  def worker(rank):
    import gc
    gc.disable()   # forked from a process that owns GPU objects: never finalise any of them here
    sys.path.insert(0, str(ROOT))
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
  out = {'value': total, 'unit': 'env-steps/s', 'cores': procs, 'kind': 'port',
         'sample': f'{procs} processes (one per host thread) x 2 envs of the CPU port (oracle/crafter_oracle.py, C noise '
                   f'helper), {seconds:.0f} s wall, random actions, resets included'}
  # The real reference cannot travel to the GPU box; its speed relative to the port was measured in the build container
  # (tools/calibrate_cpu_baseline.py: both on one core, same seeds / tape / noise shim).  Only a calibration taken on THESE
  # oracle sources is quoted -- the same rule as for the PMC profiles (round 5 quoted round 2's factor).
  import hashlib
  h = hashlib.sha256()
  for q in sorted((ROOT / 'oracle').glob('*.py')) + sorted((ROOT / 'oracle').glob('*.c')):
    h.update(q.name.encode() + b'\0' + q.read_bytes() + b'\0')
  mine = h.hexdigest()[:16]
  cals = [(json.load(open(f)), f) for f in sorted(glob.glob(str(ROOT / 'profiles' / '*_cpu_calibration.json')))]
  cals = [(c, f) for c, f in cals if c.get('oracle_hash') == mine]
  if cals:
    c, f = cals[-1]
    out['port_vs_reference'] = c['port_vs_reference']
    out['reference_equivalent'] = total / c['port_vs_reference']
    out['calibration'] = (f'{pathlib.Path(f).name}: real crafter.Env {c["reference"]["steps_per_s"]:.0f} vs port '
                          f'{c["port"]["steps_per_s"]:.0f} env-steps/s on one core of {c.get("cpu_model", "?")}, same seeds / tape / noise shim, '
                          f'oracle sources {mine}')
  else:
    out['port_vs_reference'] = None
    out['calibration'] = f'no profiles/*_cpu_calibration.json was taken on these oracle sources ({mine}): reference-equivalent figure withheld'
  try:
    out['host_cpu'] = next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), 'unknown')
  except OSError:
    pass
  return out


class Sampler:
  """Records what the oracle comparison needs of a few envs of the running batch (untimed phases only)."""

  def __init__(self, env, local_index, global_index):
    import torch
    self.env, self.local, self.glob = env, list(local_index), list(global_index)
    self.sel = torch.tensor(self.local, device=env.device)
    self.obs_sha, self.reward, self.done = [], [], []

  def record(self, obs, reward, done):
    from tests.parity import sha8
    o = obs[self.sel].cpu().numpy()
    self.obs_sha.append([sha8(x) if self.env.cfg.render_obs else None for x in o])
    self.reward.append(reward[self.sel].cpu().numpy().copy())
    self.done.append(done[self.sel].cpu().numpy().astype(bool))

  def final(self, obs):
    self.final_obs = obs[self.sel].cpu().numpy()
    self.final_snap = [self.env.snapshot(i) for i in self.local]

  def compare(self, tape_np, steps_total, kwargs):
    """Oracle replay of the sampled envs over the first `steps_total` steps of the tape (worker processes)."""
    from tests.parity import diff_snapshots, sha8
    from tests.rollout import oracle_rollouts
    res = oracle_rollouts([dict(kwargs=dict(seed=1000 + g, **kwargs), actions=tape_np[:steps_total, g], auto_reset=True)
                           for g in self.glob])
    problems = []
    for k, (g, r) in enumerate(zip(self.glob, res)):
      for t in range(len(self.obs_sha)):
        if (self.env.cfg.render_obs and self.obs_sha[t][k] != r['obs_sha'][t]) or self.reward[t][k] != r['reward'][t] or bool(self.done[t][k]) != r['done'][t]:
          problems.append(f'env {g} step {t}: obs / reward / done')
          break
      d = diff_snapshots(self.final_snap[k], r['final_snapshot'])
      if d:
        problems.append(f'env {g} final state: ' + '; '.join(d)[:200])
      if self.env.cfg.render_obs and sha8(self.final_obs[k]) != r['obs_sha'][steps_total - 1]:
        problems.append(f'env {g}: frame after the last timed step')
    return {'bit_exact': not problems, 'envs': self.glob, 'per_step_checked': len(self.obs_sha), 'final_state_after_step': steps_total,
            'episodes_finished_by_sample': int(sum(r['episodes'] for r in res)), 'problems': problems[:4],
            'checked': 'envs of the timed batch itself vs the CPU port (oracle, pinned against the reference): obs hash, reward, '
                       'done at every burn-in step; material map, objects, inventory, achievements, counters, chunk order, MT19937 '
                       'key + position and the frame after the last timed step'}


def kernel_timing(env, tape, first, reps):
  env.set_timing(True)
  for i in range(reps):
    env.step(tape[first + i], info=False)
  step_ms, reset_ms, launches = env.get_timing()
  env.set_timing(False)
  return 1000.0 * step_ms / launches, 1000.0 * reset_ms / launches, launches


STEP_KERNELS = ('crafter_step_kernel', 'crafter_step_early_kernel', 'crafter_step_wide_kernel', 'crafter_rules_kernel', 'crafter_frame_kernel')


def step_kernel_name(env, render):
  """Which kernel(s) one step() of this batch launches (crafter_hip.hip crafter_step): the default instance runs as the
  fused step kernel when frames are drawn and as the rule kernel of the split step when not; CRAFTER_SPLIT=1 forces the
  split pair (A/B); every other configuration runs a fused instance."""
  default = env.step_instance.endswith('<1, 1, 1>')
  split = int(os.environ.get('CRAFTER_SPLIT', '-1'))
  if default and (split > 0 or (split < 0 and not render)):
    return 'crafter_rules_kernel' + (' + crafter_frame_kernel' if render else '')
  wide = int(os.environ.get('CRAFTER_STEP_WIDE', '-1'))
  if default and render and (wide > 0 or (wide < 0 and env.num_envs <= 512)):
    return 'crafter_step_wide_kernel'   # 512 threads per env: batches of at most two envs per CU
  early = int(os.environ.get('CRAFTER_STEP_EARLY', '-1'))
  if default and render and (early > 0 or (early < 0 and env.num_envs >= 2048)):
    return 'crafter_step_early_kernel'  # more workgroups than the chip holds at once: the frame begins before the rules end
  return 'crafter_step_kernel'


def quoted_traffic(n, render, area, kernel_name):
  """HBM bytes per step (summed over the kernels one step launches) from the newest committed PMC profile of exactly
  this workload AND of exactly these kernel sources (crafter_amd.build.source_hash): a profile of other sources is not
  quoted -- traffic is then null and the reason is in traffic_source."""
  from crafter_amd.build import source_hash
  want = source_hash()
  stale = None
  for f in sorted(glob.glob(str(ROOT / 'profiles' / '*_hbm_traffic.json')), key=os.path.getmtime, reverse=True):
    tj = json.load(open(f))
    have = (tj.get('_source') or {}).get('csrc_sha16')
    total, seen = 0.0, []
    wl = (tj.get('_source') or {}).get('workload')   # {'envs', 'area', 'render'}: recorded by tools/summarize_profile.py since round 4
    if wl is not None and (wl.get('envs'), wl.get('area'), bool(wl.get('render'))) != (n, area, bool(render)):
      continue
    for name, t in tj.items():
      if name.startswith(STEP_KERNELS) and name.split('<')[0] in kernel_name and isinstance(t, dict) and \
         (wl is not None or t.get('grid_threads') in (n * t.get('workgroup', 256), (n + 1) * t.get('workgroup', 256))):   # (+ the block that builds the dispatch order)
        total += t['hbm_bytes_per_launch']
        seen.append(name)
    if not seen:
      continue
    if have != want:
      stale = stale or f'profiles/{pathlib.Path(f).name} was measured on kernel sources {have}, this build is {want}: not quoted'
      continue
    return total, (f'profiles/{pathlib.Path(f).name} ({" + ".join(seen)}; rocprofv3 --pmc, separate passes, same kernel sources '
                   f'{want}; not measured by this run)')
  return None, stale or 'no committed PMC profile of this workload'


def side_measurement(n, dev, burn_in, steps, reps, area=64, render=True, parity=True, sustained=0, semantic=False):
  """A smaller, self-contained measurement of another BASELINE config on one GPU (reported under "extra"), with its own
  in-run parity sample: 4 envs OF THIS BATCH against the oracle -- obs hash / reward / done at every burn-in step, the
  full state + RNG + frame after the last timed step (every driver-run line carries `parity`)."""
  import torch
  from crafter_amd import BatchedEnv
  env = BatchedEnv(n, area=(area, area), seed=1000, device=dev, auto_reset=True, render=render, semantic=semantic)
  total = burn_in + steps + sustained + reps
  tape_np = np.random.RandomState(1234).randint(0, 17, size=(total, n)).astype(np.int32)
  tape = torch.from_numpy(tape_np).to(dev)
  sampler = Sampler(env, sorted({0, 1, n // 2, n - 1}), sorted({0, 1, n // 2, n - 1})) if parity else None
  env.reset()
  for t in range(burn_in):
    o, r, d = env.step(tape[t], info=False)[:3]
    if sampler is not None and t < 300:
      sampler.record(o, r, d)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(burn_in, burn_in + steps):
    o, r, d = env.step(tape[t], info=False)[:3]
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if sampler is not None:
    sampler.final(o)
  dt_sus = None
  if sustained > 0:   # a longer window right behind the short one (device-wide synchronize on both sides)
    t1 = time.perf_counter()
    for t in range(burn_in + steps, burn_in + steps + sustained):
      env.step(tape[t], info=False)
    torch.cuda.synchronize()
    dt_sus = time.perf_counter() - t1
  kern_us, reset_us, launches = kernel_timing(env, tape, burn_in + steps + sustained, reps)
  env.check_errors()
  algo = (ALGO_BYTES_256 if area == 256 and render else ALGO_BYTES[render]) * n
  kernel_name = step_kernel_name(env, render)
  traffic, traffic_source = quoted_traffic(n, render, area, kernel_name)
  out_parity = None
  if sampler is not None:
    out_parity = sampler.compare(tape_np, burn_in + steps, {} if area == 64 else {'area': (area, area)})
  return {'parity': out_parity, 'traffic': traffic, 'traffic_source': traffic_source,
          'sustained': None if dt_sus is None else {'value': sustained * n / dt_sus, 'unit': 'env-steps/s', 'steps': sustained,
                                                     'ms_per_step': 1000 * dt_sus / sustained},'workload': f'{n} envs x 1 GPU, {area}x{area} world, obs 64x64x3, random actions, auto-reset, render {"on" if render else "off"}',
          'value': steps * n / dt, 'unit': 'env-steps/s', 'steps': steps, 'burn_in': burn_in, 'ms_per_step': 1000 * dt / steps,
          'kernel': kernel_name, 'kernel_us': kern_us, 'reset_kernel_us': reset_us, 'kernel_launches_timed': launches,
          'algorithmic_bytes_per_launch': algo, 'roofline_frac': algo / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
          'world_pool': env.pool_status()}


def open_loop_measurement(n, dev, burn_in, steps_per_call, calls, parity=True):
  """BatchedEnv.rollout (crafter_step_n): the same workload when the policy does not look at the observations -- which a
  random policy does not.  T steps per call, every frame of every step still written; an env starts step t + 1 without
  waiting for the other envs' step t.  Reported beside the headline, never as it: `value` is the closed-loop step()."""
  import torch
  from crafter_amd import BatchedEnv
  env = BatchedEnv(n, seed=1000, device=dev, auto_reset=True)
  T = steps_per_call
  total = burn_in + (calls + 2) * T
  tape_np = np.random.RandomState(1234).randint(0, 17, size=(total, n)).astype(np.int32)
  tape = torch.from_numpy(tape_np).to(dev)
  # the line's own parity sample: 4 envs of THIS batch against the oracle -- every closed-loop burn-in step and
  # every step of the two warm-up rollouts (obs hash, reward, done), then the full state + RNG after the last TIMED rollout
  sampler = Sampler(env, sorted({0, 1, n // 2, n - 1}), sorted({0, 1, n // 2, n - 1})) if parity else None
  env.reset()
  for t in range(burn_in):
    o, r, d = env.step(tape[t], info=False)[:3]
    if sampler is not None:
      sampler.record(o, r, d)
  out = (torch.empty((T,) + tuple(env.obs.shape), dtype=torch.uint8, device=dev), torch.empty((T, n), dtype=torch.float32, device=dev),
         torch.empty((T, n), dtype=torch.uint8, device=dev))
  t = burn_in
  for _ in range(2):   # warm-up calls
    env.rollout(tape[t:t + T], out=out)
    if sampler is not None:
      for k in range(T):
        sampler.record(out[0][k], out[1][k], out[2][k])
    t += T
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(calls):
    env.rollout(tape[t:t + T], out=out)
    t += T
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  env.check_errors()
  steps = calls * T
  out_parity = None
  if sampler is not None:
    sampler.final(out[0][T - 1])
    out_parity = sampler.compare(tape_np, t, {})
  return {'value': steps * n / dt, 'unit': 'env-steps/s', 'steps_per_call': T, 'calls': calls, 'ms_per_step': 1000 * dt / steps,
          'world_pool': env.pool_status(), 'parity': out_parity,
          'note': 'BatchedEnv.rollout / crafter_step_n: actions for T steps handed over at once (open loop: random / scripted policies, '
                  'action repeat), one launch per stretch between two world-pool batches, all T x N frames written; bit-identical to '
                  'T calls of step() (tests/test_gpu_rollout.py); device-wide synchronize on both sides'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2000)
  ap.add_argument('--warmup', type=int, default=200)
  ap.add_argument('--envs', type=int, default=METRIC_ENVS, help='total envs over all GPUs (strong scaling)')
  ap.add_argument('--envs-per-gpu', type=int, default=0, help='weak scaling: this many envs on every GPU')
  ap.add_argument('--burn-in', type=int, default=400)
  ap.add_argument('--kernel-reps', type=int, default=300)
  ap.add_argument('--area', type=int, default=64)
  ap.add_argument('--no-render', action='store_true')
  ap.add_argument('--no-gather-obs', action='store_true')
  ap.add_argument('--exchange', default='allgather', choices=['allgather', 'gather', 'scalars'],
                  help='N > 1: what crosses xGMI every step (crafter_amd.dist.StepExchange modes)')
  ap.add_argument('--exchange-steps', type=int, default=1,
                  help='N > 1: steps per collective for the HEADLINE window (1 = the per-step exchange the metric names; K > 1: K records '
                       'travel together).  The K = 16 figure is measured and reported beside it in any case (exchange_blocks_of_16)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-parity', action='store_true')
  ap.add_argument('--no-extra', action='store_true')
  ap.add_argument('--no-big-extra', action='store_true', help='skip configs[3] / configs[4] under extra')
  ap.add_argument('--cpu-seconds', type=float, default=12.0)
  ap.add_argument('--gen-period', type=int, default=0)
  ap.add_argument('--sustained-steps', type=int, default=1000,
                  help='a second timed window right after the --steps one (the driver times 20 steps; this is the steady state)')
  args = ap.parse_args()
  if args.exchange_steps <= 0:
    args.exchange_steps = 1

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
  force_exchange = bool(os.environ.get('CRAFTER_BENCH_FORCE_EXCHANGE'))   # world size 1 through the N > 1 code path (one GPU per box)
  # RCCL writes a version banner to STDOUT when a communicator is created: the contract is ONE JSON line there.  Everything the
  # process writes to file descriptor 1 goes to stderr from here on; the line is written to the real stdout at the end.
  sys.stdout.flush()
  real_stdout = os.fdopen(os.dup(1), 'w')
  os.dup2(2, 1)
  if world > 1 or force_exchange:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29517')
    # RCCL ("nccl") over xGMI is the product path; CRAFTER_BENCH_BACKEND=gloo only exists so that the
    # multi-process plumbing can be exercised on a box with a single GPU (gloo gathers through host memory).
    backend = os.environ.get('CRAFTER_BENCH_BACKEND', 'nccl')
    kw = {'device_id': dev} if backend == 'nccl' else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)

  from crafter_amd import BatchedEnv
  from crafter_amd import dist as cdist
  if args.envs_per_gpu:
    n, scaling = args.envs_per_gpu, 'weak'
  else:
    if args.envs % world:
      raise SystemExit(f'--envs {args.envs} does not divide over {world} GPUs')
    n, scaling = args.envs // world, 'strong'
  total_envs = n * world
  render = not args.no_render
  lo, _ = cdist.shard_range(total_envs, rank, world)
  seeds = cdist.shard_seeds(1000, total_envs, rank, world)   # global env index -> seed 1000 + index
  env = BatchedEnv(n, area=(args.area, args.area), seeds=seeds, device=dev, auto_reset=True, render=render,
                   gen_period=args.gen_period)
  steps_run = args.burn_in + args.warmup + args.steps
  total = steps_run + args.sustained_steps + args.kernel_reps + (args.sustained_steps + 100 if (world > 1 or force_exchange) else 0)
  tape_np = np.random.RandomState(1234).randint(0, 17, size=(total, total_envs)).astype(np.int32)
  tape = cdist.shard_actions(torch.from_numpy(tape_np), rank, world).contiguous().to(dev)

  exchange = None
  if world > 1 or force_exchange:
    on_host = os.environ.get('CRAFTER_BENCH_BACKEND', 'nccl') == 'gloo'
    exchange = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device='cpu' if on_host else dev,
                                  gather_obs=not args.no_gather_obs, mode=args.exchange, dst=0, steps=args.exchange_steps)

  # N > 1 on RCCL, per-step all-gather: the whole loop body is ONE call into the library (crafter_step_exchange: step kernels
  # writing into the send record + ncclAllGather on the exchange's own stream) instead of four trips through torch.distributed
  # (host 56 -> 31 us per step at 512 envs, profiles/r5_host_overhead_dist.txt).  The default since round 6
  # -- but it has never run with more than one rank (one GPU per box), so the run CHECKS it before relying on it: the first
  # burn-in step's gathered record is compared on every rank with a torch.distributed all-gather of the same send records, and
  # unless every rank agrees the whole job goes on through torch.distributed (config.exchange_enqueued_by says which).
  # CRAFTER_BENCH_NATIVE_EXCHANGE=0 switches it off.
  native, native_error = None, None
  if exchange is not None and not on_host and args.exchange == 'allgather' and args.exchange_steps == 1 and \
     os.environ.get('CRAFTER_BENCH_NATIVE_EXCHANGE', '1') != '0':
    try:
      native = cdist.NativeStepExchange(env, gather_obs=not args.no_gather_obs)
    except Exception as e:   # (every rank fails alike -- RCCL not loadable -- and falls back alike)
      native_error = repr(e)
    ok = torch.tensor([1 if native is not None else 0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok) == 0 and native is not None:
      native_error = 'another rank could not create the native exchange'
      native.close()
      native = None

  def run(t, exchange=exchange):
    if exchange is None:
      return env.step(tape[t], info=False)[:3]
    if native is not None and exchange is exchange_main:
      native.step(t, tape[t])
      return native.slots[t % native.depth].outputs(0)
    slot = exchange.begin(t)
    if slot.local.is_cuda:
      out = env.step(tape[t], info=False, out=exchange.outputs(slot))[:3]   # the kernels write the send buffer itself
    else:   # gloo plumbing mode: stage through host memory
      out = env.step(tape[t], info=False)[:3]
      for dst, src in zip(exchange.outputs(slot), out):
        if dst is not None:
          dst.copy_(src)
    exchange.launch(slot)
    return out

  exchange_main = exchange
  want_parity = rank == 0 and not args.no_parity
  sampler = None
  if want_parity:
    local = sorted({0, 1, n // 2, n - 1})
    sampler = Sampler(env, local, [lo + i for i in local])
  obs = env.reset()
  for t in range(args.burn_in):   # untimed: desynchronise the envs, pass the first night / resets, warm the world pool
    o, r, d = run(t)
    if t == 0 and native is not None:   # the native exchange's first gather against torch.distributed's of the same records
      got = native.result(0)
      torch.cuda.synchronize()
      mine = native.slots[0].local
      want = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=dev)
      dist.all_gather_into_tensor(want, mine)
      same = torch.tensor([1 if torch.equal(want.reshape(-1), native.slots[0].gathered.reshape(-1)) else 0], device=dev)
      dist.all_reduce(same, op=dist.ReduceOp.MIN)
      if int(same) == 0:
        native_error = 'self-check failed: the native all-gather of step 0 differs from torch.distributed\'s on some rank'
        native.finish()
        native = None
    if sampler is not None and t < 300:
      sampler.record(o, r, d)
  for t in range(args.burn_in, args.burn_in + args.warmup):
    run(t)
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  for t in range(args.burn_in + args.warmup, steps_run):
    o, r, d = run(t)
  t_host = time.perf_counter()   # every launch of the K steps has been enqueued: what the HOST needed for them
  ev1.record()
  if exchange is not None:
    exchange.finish()
  if native is not None:
    native.finish()
  # Two clocks.  (1) The K steps are complete when the launch stream has drained: obs / reward / done and the state of
  # step K are final.  (2) Device-wide synchronize: also waits for the world-pool batch a side stream is still working on
  # for FUTURE resets.  The window starts from a device-wide synchronize, i.e. with an idle pool, so (1) sees a little less
  # generation beside its first steps than the steady state does (optimistic: +5 % at 20 steps in round 2) and (2) bills
  # the whole tail of one batch (~0.3 ms) to the window (pessimistic: -15 % at 20 steps, nothing at 2000).  `value` is (2)
  # since round 5 -- the contract's clock: barrier + device synchronize on both sides, the conservative one --,
  # (1) is reported as launch_stream_ms_per_step, and the steady state -- device-wide synchronize on both sides of 1000
  # further steps -- under `sustained`.
  torch.cuda.current_stream(dev).synchronize()
  t_end = time.perf_counter()
  torch.cuda.synchronize()
  t_sync = time.perf_counter()
  if dist is not None:
    dist.barrier()
  dt_sync = (time.perf_counter() if dist is not None else t_sync) - t0   # (N > 1: closed by the barrier behind the synchronize)
  dt_stream = t_end - t0
  dt = dt_sync
  gpu_ms = ev0.elapsed_time(ev1)
  env.check_errors()
  if sampler is not None:
    sampler.final(o)
  # sustained: the same protocol over a longer window of further steps (a 20-step window sits wherever the episode
  # phases of the batch happen to be)
  dt_sus = None
  if args.sustained_steps > 0:
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for t in range(steps_run, steps_run + args.sustained_steps):
      run(t)
    if exchange is not None:
      exchange.finish()
    if native is not None:
      native.finish()
    torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()
    dt_sus = time.perf_counter() - t1
    env.check_errors()
  exchange_us = None
  if exchange is not None:   # the exchange alone, nothing overlapping it: per-step cost if it were exposed
    torch.cuda.synchronize()
    dist.barrier()
    t2 = time.perf_counter()
    for k in range(50):
      slot = exchange.begin(steps_run + args.sustained_steps + k)
      exchange.launch(slot)
    exchange.finish()
    torch.cuda.synchronize()
    exchange_us = 1e6 * (time.perf_counter() - t2) / 50
  host_us = 1e6 * (t_host - t0) / args.steps
  if dist is not None:
    both = torch.tensor([dt, dt_sus or 0.0, exchange_us or 0.0, host_us, dt_stream], dtype=torch.float64, device=dev)
    dist.all_reduce(both, op=dist.ReduceOp.MAX)
    dt, dt_sus, exchange_us, host_us, dt_stream = float(both[0]), (float(both[1]) if dt_sus else None), float(both[2]), float(both[3]), float(both[4])
    dt_sync = dt
  # N > 1: the same sustained window again with 16 steps' records per collective (what round 4 ran as its default): the
  # host then enqueues ONE collective per 16 steps.  Another workload than the metric's -- a learner sees frames up to 16
  # steps late -- hence beside `value`, never as it (VERDICT r4 missing #5).
  blocks16 = None
  if exchange is not None and args.sustained_steps > 0:
    first = steps_run + args.sustained_steps + 50
    count = min(args.sustained_steps, total - first)
    if count >= 16:
      ex16 = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device=exchange.slots[0].local.device,
                                gather_obs=not args.no_gather_obs, mode=args.exchange, dst=0, steps=16)
      dist.barrier()
      torch.cuda.synchronize()
      t3 = time.perf_counter()
      for t in range(first, first + count):
        run(t, ex16)
      t3h = time.perf_counter()
      ex16.finish()
      torch.cuda.synchronize()
      dist.barrier()
      both = torch.tensor([time.perf_counter() - t3, t3h - t3], dtype=torch.float64, device=dev)
      dist.all_reduce(both, op=dist.ReduceOp.MAX)
      blocks16 = {'value': count * total_envs / float(both[0]), 'unit': 'env-steps/s', 'steps': count, 'exchange_steps_per_collective': 16,
                  'ms_per_step': 1000 * float(both[0]) / count, 'host_us_per_step': 1e6 * float(both[1]) / count,
                  'note': 'the sustained protocol with 16 steps per collective: NOT the per-step exchange the metric names'}

  if rank == 0:
    # dominant kernel: mean duration of crafter_step_kernel (and of the auto-reset kernel that follows it in the
    # same call) over --kernel-reps launches, HIP events attached to the kernels on the launch stream
    kern_us, reset_us, launches = kernel_timing(env, tape, steps_run + args.sustained_steps, args.kernel_reps)
    pool = env.pool_status()
    kernel_name = step_kernel_name(env, render)
    traffic, traffic_source = quoted_traffic(n, render, args.area, kernel_name)
    value = args.steps * total_envs / dt
    per_env = (ALGO_BYTES_256 if args.area == 256 and render else ALGO_BYTES[render])
    bytes_per_launch = per_env * n
    achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9
    workload = (f'{total_envs} envs ({n}/GPU x {world} GPU), {args.area}x{args.area} world, view 9x9, obs 64x64x3, '
                f'random actions, auto-reset, render {"on" if render else "off"}')
    line = {
        'metric': 'env-steps/sec (whole node), random policy, 4096 envs', 'value': value, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000 * dt / args.steps,
        'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'u8/i32 (+f64 render filters)',
        'data': 'synthetic',
        'config': {'workload': workload, 'envs_total': total_envs, 'envs_per_gpu': n,
                   'parallelism': f'env-index sharding x{world}',
                   'exchange': None if exchange is None else {
                       'allgather': 'per-step all_gather of the packed (obs, reward, done) record, double-buffered',
                       'gather': 'per-step gather of the packed (obs, reward, done) record to rank 0 (the learner), double-buffered',
                       'scalars': 'per-step all_gather of (reward, done); frames stay on the rank that rendered them'}[args.exchange]
                       + (' [obs left out: --no-gather-obs]' if args.no_gather_obs else ''),
                   'exchange_bytes_received_per_step_busiest_rank': None if exchange is None else (
                       exchange.slots[0].record_bytes * (world - 1)),
                   'exchange_wire_bytes_per_step': None if exchange is None else exchange.wire_bytes_per_step,
                   'exchange_alone_us_per_step': exchange_us, 'exchange_steps_per_collective': args.exchange_steps,
                   'exchange_enqueued_by': None if exchange is None else (
                       'libcrafter_hip.so (crafter_step_exchange: step kernels + ncclAllGather in one call)' if native is not None
                       else 'torch.distributed (crafter_amd.dist.StepExchange)' + (f' [native form unavailable: {native_error}]' if native_error else '')),
                   'step_kernel': env.step_instance if step_kernel_name(env, render) == 'crafter_step_kernel' else step_kernel_name(env, render),
                   'dispatch_order': 'slow envs (night frame / balance step next) first' if env.dispatch_order() is not None else None},
        'burn_in': args.burn_in, 'gpu_ms_per_step': gpu_ms / args.steps, 'device_sync_ms_per_step': 1000 * dt_sync / args.steps,
        'launch_stream_ms_per_step': 1000 * dt_stream / args.steps, 'exchange_blocks_of_16': blocks16,
        'host_us_per_step': host_us,   # max over ranks: the time the Python loop body (exchange calls included) took to ENQUEUE a step;
                                       # where it exceeds ms_per_step the run is host-bound (tools/host_overhead_dist.py)
        'clock': 'value / ms_per_step: (barrier +) device-wide synchronize on both sides of the K steps; launch_stream_ms_per_step: the same '
                 'window closed when the launch stream has drained (world-pool batches for future resets may still run on their side '
                 'streams); sustained: the value clock over further steps',
        'sustained': None if dt_sus is None else {
            'value': args.sustained_steps * total_envs / dt_sus, 'unit': 'env-steps/s', 'steps': args.sustained_steps,
            'ms_per_step': 1000 * dt_sus / args.sustained_steps,
            'note': 'the same protocol (barrier + device synchronize on both sides) over the steps that follow the timed window'},
        'world_pool': pool,
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic, 'traffic_source': traffic_source, 'kernel': kernel_name, 'kernel_us': kern_us,
                     'kernel_launches_timed': launches, 'reset_kernel_us': reset_us, 'algorithmic_bytes_per_launch': bytes_per_launch},
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(n, args.cpu_seconds)
    if sampler is not None:
      line['parity'] = sampler.compare(tape_np, steps_run, {} if args.area == 64 else {'area': (args.area, args.area)})
    if world == 1 and not args.no_extra and total_envs == METRIC_ENVS and args.area == 64 and render:
      del env
      torch.cuda.synchronize()
      line['open_loop'] = open_loop_measurement(n, dev, 400, 64, 24, parity=not args.no_parity)
      par = not args.no_parity
      line['extra'] = {'configs[1]': side_measurement(1024, dev, 400, 600, 300, parity=par, sustained=1000)}
      # the headline workload with info['semantic'] written every step, as the reference computes it (env.py:113; here it is
      # opt-in): + 4 KB of writes per env-step
      sem = side_measurement(METRIC_ENVS, dev, 400, 300, 100, parity=par, sustained=1000, semantic=True)
      sem['workload'] += ", info['semantic'] on"
      sem['algorithmic_bytes_per_launch'] += 4096 * METRIC_ENVS
      sem['roofline_frac'] = sem['algorithmic_bytes_per_launch'] / (sem['kernel_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS
      sem['traffic'], sem['traffic_source'] = None, 'not profiled with the semantic view on'
      line['extra']['semantic_on'] = sem
      if not args.no_big_extra:   # BASELINE configs[4] and configs[3]: a short window + a 1000-step one, each with its parity sample
        line['extra']['configs[4]'] = side_measurement(16384, dev, 300, 300, 100, render=False, parity=par, sustained=1000)
        line['extra']['configs[3]'] = side_measurement(8192, dev, 200, 100, 50, area=256, parity=par, sustained=1000)
    sys.stdout.flush()
    real_stdout.write(json.dumps(line) + '\n')
    real_stdout.flush()
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
