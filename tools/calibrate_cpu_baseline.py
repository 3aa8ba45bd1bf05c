#!/usr/bin/env python3
"""Build container only (needs /root/reference): times the REAL reference ``crafter.Env`` (imported untouched through
oracle/reference_harness.py: three import shims, slot-ordered chunk sets) and the CPU port (oracle/crafter_oracle.py) on
the same cores, the same seeds and the same action tape, and records their per-core ratio.

bench.py's ``cpu_baseline`` has to time the port on the GPU box (the reference tree does not exist there); this file
gives the factor that converts it into "the reference on those cores" -- written to profiles/<tag>_cpu_calibration.json,
which bench.py quotes as ``cpu_baseline.port_vs_reference``.

Also records BASELINE.json configs[0] -- the reference's own instrument ``python -m crafter.run_random --seed 0``
(run_random.py:28-43: reset time, step time / FPS, episode length) -- run through the same shims.

Noise shim: oracle/refshim/opensimplex.py -> oracle/noise.py (C helper oracle/osimplex.c when gcc is present); both
the reference and the port call the same noise implementation, so the ratio is not a noise artefact."""
import argparse
import io
import json
import os
import pathlib
import sys
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def oracle_hash():
  """sha256 (16 hex digits) over the CPU port's sources: what the calibration's factor was measured on."""
  import hashlib
  h = hashlib.sha256()
  for q in sorted((ROOT / 'oracle').glob('*.py')) + sorted((ROOT / 'oracle').glob('*.c')):
    h.update(q.name.encode() + b'\0' + q.read_bytes() + b'\0')
  return h.hexdigest()[:16]


def run(make, seeds, tape, seconds):
  envs = [make(s) for s in seeds]
  t0 = time.perf_counter()
  for e in envs:
    e.reset()
  reset_s = (time.perf_counter() - t0) / len(envs)
  steps, resets, t = 0, 0, 0
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    for k, e in enumerate(envs):
      _, _, done, _ = e.step(int(tape[t % len(tape), k]))
      steps += 1
      if done:
        e.reset()
        resets += 1
    t += 1
  dt = time.perf_counter() - t0
  return {'steps_per_s': steps / dt, 'steps': steps, 'resets': resets, 'seconds': dt, 'reset_s': reset_s}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--tag', default='r6')
  ap.add_argument('--seconds', type=float, default=20.0)
  ap.add_argument('--envs', type=int, default=4)
  args = ap.parse_args()
  from oracle import reference_harness as rh
  from oracle.crafter_oracle import OracleEnv
  from oracle import noise
  seeds = [1000 + i for i in range(args.envs)]
  tape = np.random.RandomState(1234).randint(0, 17, size=(100000, args.envs)).astype(np.int32)
  os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})   # both on the same single core, one after the other
  ref = run(lambda s: rh.make_env(seed=s), seeds, tape, args.seconds)
  port = run(lambda s: OracleEnv(seed=s), seeds, tape, args.seconds)
  # configs[0]: the reference's CLI, in-process (its module-level code is what `python -m crafter.run_random` runs)
  crafter = rh.load()
  import importlib
  rr = importlib.import_module('crafter.run_random')
  buf = io.StringIO()
  argv = sys.argv
  sys.argv = ['crafter.run_random', '--seed', '0']
  try:
    with redirect_stdout(buf):
      rr.main()
  finally:
    sys.argv = argv
  crafter.constants.items['health']['max'] = 9
  crafter.constants.items['health']['initial'] = 9
  cpu_model = next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), 'unknown')
  out = {
      'host': os.uname().nodename, 'cpu_model': cpu_model, 'host_cores': os.cpu_count(), 'cores_used': 1, 'envs': args.envs, 'seeds': seeds,
      'oracle_hash': oracle_hash(),   # bench.py only quotes a calibration taken on the oracle sources it is timing
      'tape': 'RandomState(1234).randint(0, 17), auto-reset on done, resets included',
      'noise': f'oracle/noise.py ({"C helper osimplex.c" if noise.have_c() else "pure Python"}) behind both',
      'reference': ref, 'port': port,
      'port_vs_reference': port['steps_per_s'] / ref['steps_per_s'],
      'config0_run_random_seed0': buf.getvalue().strip().splitlines(),
  }
  path = ROOT / 'profiles' / f'{args.tag}_cpu_calibration.json'
  path.write_text(json.dumps(out, indent=1) + '\n')
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
