#!/usr/bin/env python3
"""GPU box: on which SIMD of its CU does each wave of a step workgroup run?  Needs a PROBE BUILD of the library through
CRAFTER_HIP_LIB (tools/build_simd_probe.sh: after the last stamp of step_body every wave's first lane stores the low 16
bits of HW_ID -- WAVE_ID[3:0], SIMD_ID[5:4], PIPE_ID[7:6], CU_ID[11:8], SH_ID[12], SE_ID[15:13] -- into its quarter of
stamp slot 9).  Wave 0 of a step workgroup runs the whole rule phase: if the dispatcher puts every workgroup's wave 0 on
the same SIMD, that SIMD's vector issue bounds the launch.
usage: CRAFTER_HIP_LIB=gpurun_ab/simd_probe.so tools/gpu_simd_probe.py [envs]"""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from crafter_amd import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedEnv(n, seed=1000, auto_reset=True)
env.reset()
tape = torch.from_numpy(np.random.RandomState(1234).randint(0, 17, size=(310, n)).astype(np.int32)).cuda()
for t in range(300):
  env.step(tape[t], info=False)
prof = env.enable_phase_stamps(True)
for t in range(300, 303):
  torch.cuda.synchronize(); prof.zero_()
  env.step(tape[t], info=False)
  torch.cuda.synchronize()
  p = prof.cpu().numpy()
  hw = p[:, 9].copy().view(np.uint16).reshape(n, 4).astype(np.int64)   # [env][wave]
  simd = (hw >> 4) & 3
  slot = hw & 15
  cu = (hw[:, 0] >> 8) & 0xFF
  print(f'launch {t}: SIMD of wave 0 over envs:', np.bincount(simd[:, 0], minlength=4).tolist(),
        '| wave 1:', np.bincount(simd[:, 1], minlength=4).tolist(), '| wave 2:', np.bincount(simd[:, 2], minlength=4).tolist(),
        '| wave 3:', np.bincount(simd[:, 3], minlength=4).tolist())
  print('   (simd of wave k - simd of wave 0) mod 4, k = 1..3:', [np.bincount((simd[:, k] - simd[:, 0]) % 4, minlength=4).tolist() for k in (1, 2, 3)])
  print('   same CU for all four waves:', float(((hw >> 8) == (hw[:, :1] >> 8)).all(1).mean()), '| wave slots used by wave 0:', np.bincount(slot[:, 0], minlength=16).tolist())
  print('   first 12 envs [simd of waves 0..3]:', simd[:12].tolist())
