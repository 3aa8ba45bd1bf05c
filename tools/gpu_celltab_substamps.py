"""GPU box, probe of round 6: where the frame's cell-table phase goes (day steps, 4096 envs).  Needs a library built with the two
extra stamps of tools/patches/r6_celltab_substamps.patch: CRAFTER_HIP_LIB=gpurun_ab/substamp.so python tools/gpu_celltab_substamps.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from crafter_amd import BatchedEnv
n=4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
T=1000
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(T, n)).astype(np.int32)).cuda()
for t in range(400): env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
day = env.tables.daylight
acc=[]
for t in range(400, T):
  if t % 25 == 24:
    torch.cuda.synchronize(); prof.zero_(); sb = env.records()['step'].astype(np.int64)
  env.step(tape[t], info=False)
  if t % 25 == 24:
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.int64); rec = env.records()
    s = rec['step'].astype(np.int64)
    ok = (p[:,5]>0)&(p[:,6]==0)&(rec['step']==sb+1)&(day[np.clip(s,0,len(day)-1)]>=0.5)&(s%10!=0)
    acc.append(np.stack([p[:,14]-p[:,11], p[:,15]-p[:,14], p[:,12]-p[:,15], p[:,11]-p[:,3]],1)[ok])
a=np.concatenate(acc)
print('day steps: render entry -> wave-0 block %.0f | cell table + lists %.0f | sprite rows (fetch + place) %.0f | (stamp3 -> stamp11: %.0f)' % tuple(a.mean(0)))
