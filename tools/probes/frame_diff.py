"""GPU box debugging aid: which pixels of which envs differ from oracle frames saved beforehand (gpurun_ab/oracle37.npz)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crafter_amd import BatchedEnv
ref = np.load('gpurun_ab/oracle37.npz')
n = 4096
tapes = np.random.RandomState(1234).randint(0, 17, size=(300, n)).astype(np.int32)
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
dev = torch.from_numpy(tapes).cuda()
for t in range(len(dev)):
  obs, rew, done, _ = env.step(dev[t], info=False)
  for k in ref.files:
    i = int(k)
    if t >= len(ref[k]): continue
    a = obs[i].cpu().numpy(); b = ref[k][t]
    if not np.array_equal(a, b):
      ys, xs, cs = np.nonzero(a != b)
      print('env', i, 'step', t, 'differing bytes', len(ys), 'rows', sorted(set(ys.tolist()))[:20], 'cols', sorted(set(xs.tolist()))[:30])
      for y, x in list(dict.fromkeys(zip(ys.tolist(), xs.tolist())))[:8]:
        print('   (y %d, x %d) device %s oracle %s' % (y, x, a[y, x].tolist(), b[y, x].tolist()))
print('done')
