"""-m gpu: first execution of the multi-GPU code path on the device (VERDICT r2 weak #1b): torch.distributed with the
`nccl` backend (= RCCL on ROCm) at world_size 1 -- a box has one GPU -- driving crafter_amd.dist.StepExchange with
device records, the step kernels writing their outputs straight into the send record (BatchedEnv.step(out=...)), the
collective launched asynchronously behind the step and its result consumed one step late.  Equality with a plain
BatchedEnv run of the same seeds and tape shows that the record's views are laid out the way the kernels write them
and that the RCCL stream is ordered behind the step that filled the record."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


@pytest.fixture(scope='module')
def rccl_group():
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1, device_id=dev)
  yield dev
  dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['allgather', 'gather', 'scalars'])
def test_step_writes_the_send_record_and_rccl_returns_it(rccl_group, mode):
  from crafter_amd import BatchedEnv
  from crafter_amd import dist as cdist
  dev = rccl_group
  n, T = 64, 50
  seeds = cdist.shard_seeds(1000, n, 0, 1)
  tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).to(dev)
  ref_env = BatchedEnv(n, seeds=seeds, device=dev, auto_reset=True, length=30)
  ref_env.reset()
  want = []
  for t in range(T):
    o, r, d, _ = ref_env.step(tape[t], info=False)
    want.append((o.clone(), r.clone(), d.clone()))
  env = BatchedEnv(n, seeds=seeds, device=dev, auto_reset=True, length=30)
  ex = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device=dev, mode=mode, dst=0)
  assert ex.slots[0].local.is_cuda and ex.receives
  env.reset()

  def consume(t):
    obs, rew, done = ex.result(t)
    o, r, d = want[t]
    if mode == 'scalars':
      assert obs.shape == o.shape and torch.equal(obs, o), f'{mode} step {t}: own frames'
    else:
      assert obs.shape == (1,) + tuple(o.shape) and torch.equal(obs[0], o), f'{mode} step {t}: gathered frames'
    assert torch.equal(rew[0], r) and torch.equal(done[0], d), f'{mode} step {t}: reward / done'

  for t in range(T):
    slot = ex.begin(t)
    out = ex.outputs(slot)
    assert all(x.is_cuda for x in out)
    env.step(tape[t], info=False, out=out)   # the kernels (auto-reset kernel included) write the record itself
    ex.launch(slot)
    if t >= 1:
      consume(t - 1)   # late: the collective of step t is in flight
  consume(T - 1)
  ex.finish()
  env.check_errors()
  assert int(torch.stack([w[2] for w in want]).sum()) >= n, 'length=30: every env finished (and was regenerated) at least once'


@pytest.mark.parametrize('mode', ['allgather', 'gather'])
def test_blocks_of_several_steps_per_collective_over_rccl(rccl_group, mode):
  """StepExchange(steps=K) on the nccl backend: K consecutive steps' records are written by the step kernels into the rows
  of ONE send buffer and travel in one collective (what `bench.py --gpus N` does by default for N > 1: the per-step
  exchange costs the host more than a 512-env step costs the GPU).  53 steps in blocks of 4 -- the last
  block closed short by finish() -- consumed a block late, against a plain BatchedEnv run."""
  from crafter_amd import BatchedEnv
  from crafter_amd import dist as cdist
  dev = rccl_group
  n, T, K = 64, 53, 4
  seeds = cdist.shard_seeds(1000, n, 0, 1)
  tape = torch.from_numpy(np.random.RandomState(4321).randint(0, 17, size=(T, n)).astype(np.int32)).to(dev)
  ref_env = BatchedEnv(n, seeds=seeds, device=dev, auto_reset=True, length=30)
  ref_env.reset()
  want = []
  for t in range(T):
    o, r, d, _ = ref_env.step(tape[t], info=False)
    want.append((o.clone(), r.clone(), d.clone()))
  env = BatchedEnv(n, seeds=seeds, device=dev, auto_reset=True, length=30)
  ex = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device=dev, mode=mode, dst=0, steps=K)
  env.reset()

  def consume(t):
    obs, rew, done = ex.result(t)
    o, r, d = want[t]
    assert torch.equal(obs[0], o) and torch.equal(rew[0], r) and torch.equal(done[0], d), f'{mode} step {t}'

  for t in range(T):
    slot = ex.begin(t)
    env.step(tape[t], info=False, out=ex.outputs(slot))
    ex.launch(slot)
    if t % K == K - 1 and t >= 2 * K - 1:   # the block before the one just launched
      for u in range(t - 2 * K + 1, t - K + 1):
        consume(u)
  ex.finish()
  for u in range(((T - 1) // K - 1) * K, T):   # the last full block and the short one
    consume(u)
  env.check_errors()


def test_out_tensors_are_validated(rccl_group):
  from crafter_amd import BatchedEnv
  env = BatchedEnv(4, seed=1, device=rccl_group)
  env.reset()
  a = torch.zeros(4, dtype=torch.int32, device=rccl_group)
  buf = torch.zeros(4 * 64 * 64 * 3 + 1, dtype=torch.uint8, device=rccl_group)
  with pytest.raises(ValueError):
    env.step(a, out=(buf[1:].view(4, 64, 64, 3), env.reward, env.done))   # misaligned
  with pytest.raises(ValueError):
    env.step(a, out=(None, env.reward.to(torch.float64), env.done))       # wrong dtype


def test_closed_loop_actions_over_rccl(rccl_group):
  """StepExchange.scatter_actions on the nccl backend (world size 1: the broadcast / scatter of the learner's actions is
  enqueued behind the exchange and ahead of the step that consumes them): a policy that reads the gathered reward / done
  of step t to choose the actions of step t + 1, against the same loop without any exchange."""
  from crafter_amd import BatchedEnv
  from crafter_amd import dist as cdist
  dev = rccl_group
  n, T = 64, 40

  def policy(t, rew, done):
    idx = torch.arange(n, device=dev)
    return ((t + idx + 3 * done.to(torch.int64) + (rew > 0).to(torch.int64)) % 17).to(torch.int32)

  ref = BatchedEnv(n, seed=1000, device=dev, auto_reset=True, length=30)
  ref.reset()
  want, rew, done = [], torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
  for t in range(T):
    o, rew, done, _ = ref.step(policy(t, rew, done), info=False)
    rew, done = rew.clone(), done.clone()
    want.append((o.clone(), rew, done))
  for how in ('broadcast', 'scatter'):
    env = BatchedEnv(n, seed=1000, device=dev, auto_reset=True, length=30)
    ex = cdist.StepExchange(n, obs_shape=tuple(env.obs.shape[1:]), device=dev, mode='allgather')
    env.reset()
    rew, done = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
    for t in range(T):
      mine = ex.scatter_actions(policy(t, rew, done), src=0, how=how)
      slot = ex.begin(t)
      env.step(mine, info=False, out=ex.outputs(slot))
      ex.launch(slot)
      obs, g_rew, g_done = ex.result(t)
      rew, done = g_rew[0].clone(), g_done[0].clone()
      assert torch.equal(obs[0], want[t][0]) and torch.equal(rew, want[t][1]) and torch.equal(done, want[t][2]), (how, t)
    ex.finish()
    env.check_errors()


def test_native_step_exchange_enqueued_from_c(rccl_group):
  """crafter_step_exchange (include/crafter_hip.h; crafter_amd.dist.NativeStepExchange): ONE call into the library enqueues
  the step kernels -- outputs straight into the packed send record -- and the RCCL all-gather of the record on the
  exchange's own stream (the library's own communicator, its id handed round through torch.distributed).  World size 1 --
  a box has one GPU --: the gathered record must equal a plain run's outputs, consumed one step late while the next
  exchange is in flight, over more steps than there are slots."""
  from crafter_amd import BatchedEnv
  from crafter_amd import dist as cdist
  dev = rccl_group
  n, T = 64, 60
  seeds = cdist.shard_seeds(1000, n, 0, 1)
  tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).to(dev)
  ref_env = BatchedEnv(n, seeds=seeds, device=dev, auto_reset=True, length=30)
  ref_env.reset()
  want = []
  for t in range(T):
    o, r, d, _ = ref_env.step(tape[t], info=False)
    want.append((o.clone(), r.clone(), d.clone()))
  env = BatchedEnv(n, seeds=seeds, device=dev, auto_reset=True, length=30)
  ex = cdist.NativeStepExchange(env)
  assert ex.world == 1 and ex.slots[0].local.is_cuda
  env.reset()

  def consume(t):
    obs, rew, done = ex.result(t)
    o, r, d = want[t]
    assert obs.shape == (1,) + tuple(o.shape) and torch.equal(obs[0], o), f'step {t}: gathered frames'
    assert torch.equal(rew[0], r) and torch.equal(done[0], d), f'step {t}: reward / done'

  for t in range(T):
    ex.step(t, tape[t])
    if t >= 1:
      consume(t - 1)
  consume(T - 1)
  with pytest.raises(RuntimeError, match='no longer buffered'):
    ex.result(T - 3)
  ex.finish()
  env.check_errors()
  for i in (0, n - 1):
    a, b = env.snapshot(i), ref_env.snapshot(i)
    assert a['step'] == b['step'] and np.array_equal(a['mat'], b['mat']) and a['objects'] == b['objects']
  ex.close()
