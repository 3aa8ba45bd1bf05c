#!/usr/bin/env python3
"""GPU box: phases of one world generation (crafter_reset_kernel, 1024 threads per world) from shader-clock stamps."""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = BatchedEnv(n, seed=1000, auto_reset=False)
prof = env.enable_phase_stamps(True)
env.reset()
torch.cuda.synchronize()
prof.zero_()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); env.reset(); t1.record(); torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.int64)
names = ['load+clear+mtseed', 'simplex perm', 'classify noise', 'material draws', 'creature draws', 'finalize+render', 'store']
d = np.diff(p[:, 8:16], axis=1)
print('reset of', n, 'envs:', round(t0.elapsed_time(t1) * 1000), 'us; per-world phases (ticks):')
for i, k in enumerate(names):
  print(f'  {k:20s} mean {d[:, i].mean():10.0f}  max {d[:, i].max():10.0f}')
print(f'  total                mean {(p[:, 15] - p[:, 8]).mean():10.0f}')
