#!/usr/bin/env python3
"""CPU (oracle only): sweeps (seed, episode) pairs for worlds in which worldgen.py:25-27's `start > 0.5` is a STRUCTURAL
tie -- one of the four cells at distance exactly 4 from the player, where start = 4 - 4 + 2 * simplex(x, y, 8, 3) is the
noise alone, with |start| below 1e-15 but not zero: there the flavour of exp() (SVML / libm / correctly rounded) decides
1 / (1 + exp(-start)) > 0.5, and with it a material and every later uniform() of the world.
Writes the first `want` triples to tests/golden/exp_ties.json with the measured rate, and -- over `min_worlds` worlds -- how
many such cells THIS host's np.exp decides differently from the correctly rounded exponential (oracle/exp_cr.py): the worlds
on which the untouched reference and this framework really part (the first of those are kept too).
usage: tools/sweep_exp_ties.py [want] [episodes per seed] [min worlds]"""
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import noise  # noqa: E402
from oracle.exp_cr import exp_cr  # noqa: E402

want = int(sys.argv[1]) if len(sys.argv) > 1 else 10
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W = H = 64
px, py = W // 2, H // 2
cells = [(px - 4, py), (px + 4, py), (px, py - 4), (px, py + 4)]
min_worlds = int(sys.argv[3]) if len(sys.argv) > 3 else 0
found, decisive, n_ties, n_decisive, worlds, t0 = [], [], 0, 0, 0, time.time()
seed = 0
while len(found) < want or worlds < min_worlds:
  for ep in range(1, episodes + 1):
    wseed = hash((seed, ep)) % (2 ** 31 - 1)                                 # env.py:74
    sseed = int(np.random.RandomState(wseed).randint(0, 2 ** 31 - 1))        # worldgen.py:11
    sx = noise.OpenSimplex(sseed)
    worlds += 1
    for (x, y) in cells:
      start = 4 - np.sqrt((x - px) ** 2 + (y - py) ** 2) + 2 * (sx.noise3(x / 3, y / 3, 8) / 1.0)   # worldgen.py:25,80-87 (one octave: weight 1, normalised)
      if start != 0 and abs(start) < 1e-15:
        rec = {'seed': seed, 'episode': ep, 'cell': [x, y], 'start': float(start)}
        flips = (1 / (1 + exp_cr(-float(start))) > 0.5) != (1 / (1 + np.exp(-start)) > 0.5)   # worldgen.py:27
        n_ties += 1
        n_decisive += flips
        if len(found) < want:
          found.append(rec)
        if flips and len(decisive) < want:
          decisive.append(rec)
        print(rec, 'np.exp DECIDES DIFFERENTLY here' if flips else '', f'({worlds} worlds, {time.time() - t0:.0f} s)', flush=True)
  seed += 1
out = {'worlds_swept': worlds, 'ties': found, 'ties_found': n_ties, 'rate': n_ties / worlds,
       'decided_differently_by_this_hosts_np_exp': {'count': int(n_decisive), 'rate': n_decisive / worlds, 'first': decisive,
                                                     'np_exp_of_the_known_tie': float(np.exp(np.float64(-1.638387376145862e-16))).hex()},
       'note': 'area 64x64; cells at distance exactly 4 from the player with 0 < |start| < 1e-15 (tools/sweep_exp_ties.py)'}
(ROOT / 'tests' / 'golden' / 'exp_ties.json').write_text(json.dumps(out, indent=1) + '\n')
print(f'{n_ties} ties in {worlds} worlds: 1 in {worlds / max(1, n_ties):.0f}; {n_decisive} of them decided differently by this host\'s np.exp: 1 world in {worlds / max(1, n_decisive):.0f}')
