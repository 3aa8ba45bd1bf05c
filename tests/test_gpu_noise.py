"""-m gpu: the world generator's arithmetic evaluated ON gfx950 (VERDICT r2 weak #1c) -- the kernels' noise3
(csrc/simplex.hpp), the square root of worldgen.py:25 and the pinned exponential of worldgen.py:27 (exp_cr) -- through
the C ABI (crafter_debug_eval) against the CPU oracle (oracle/noise.py, oracle/exp_cr.py, numpy's sqrt).  The CPU suite runs the same noise3 body through tests/hostsim with the host's libm; this is the
device's own code generation (v_mul_f64 / v_add_f64 without contraction, v_floor_f64, f64 <-> int conversions).

The 3-D noise itself stays "parity unpinned" against the real opensimplex package (tests/test_noise.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import noise
from oracle import opensimplex_ref as ref

pytestmark = pytest.mark.gpu

# (z, size) of every simplex() call in worldgen.py:21-61, dict order of the octaves
LOOKUPS = [(8, 3), (3, 15), (3, 5), (0, 15), (0, 5), (6, 7), (1, 8), (2, 6), (6, 5), (4, 9), (5, 7)]


def _worldgen_points(area):
  xs, ys = np.meshgrid(np.arange(float(area)), np.arange(float(area)), indexing='ij')
  x, y = xs.ravel(), ys.ravel()
  pts = [np.stack([x / size, y / size, np.full(x.size, float(z))], 1) for z, size in LOOKUPS]
  pts.append(np.stack([(2 * x) / 3, (y / 5) / 3, np.full(x.size, 7.0)], 1))   # horizontal tunnels: simplex(2 * x, y / 5, 7, 3)
  pts.append(np.stack([(x / 5) / 3, (2 * y) / 3, np.full(x.size, 7.0)], 1))   # vertical tunnels:   simplex(x / 5, 2 * y, 7, 3)
  return np.concatenate(pts)


def _eval(mode, perm, x, y=None, z=None):
  from crafter_amd import lib as libmod
  lib = libmod.load()
  dev = torch.device('cuda', 0)
  t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  tp, tx, ty, tz = t(perm), t(x), t(y), t(z)
  out = torch.empty(len(x), dtype=torch.float64, device=dev)
  ptr = lambda a: C.c_void_p(None if a is None else a.data_ptr())
  rc = lib.crafter_debug_eval(mode, ptr(tp), ptr(tx), ptr(ty), ptr(tz), ptr(out), len(x),
                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
  assert rc == 0, libmod.last_error(lib, None)
  torch.cuda.synchronize()
  return out.cpu().numpy()


def test_device_noise3_is_bit_identical_to_the_oracle():
  rs = np.random.RandomState(11)
  p = np.ascontiguousarray(np.concatenate([_worldgen_points(64), _worldgen_points(256)[::7], rs.uniform(-60, 60, size=(60000, 3))]))
  assert len(p) >= 100000
  for seed in (0, 1234, 2147483646):
    want = noise.OpenSimplex(seed).noise3_many(p[:, 0], p[:, 1], p[:, 2])
    perm8 = np.array(ref.make_perm(seed)[0], np.uint8)
    got = _eval(0, perm8, p[:, 0], p[:, 1], p[:, 2])
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert not len(bad), (seed, len(bad), p[bad[:3]], got[bad[:3]], want[bad[:3]])


def test_device_sqrt_and_pinned_sigmoid_of_worldgen():
  """worldgen.py:25-27: start = 4 - np.sqrt(dx ** 2 + dy ** 2) + 2 * simplex(x, y, 8, 3); start = 1 / (1 + np.exp(-start)).
  sqrt is correctly rounded everywhere: bit-exact.  The exponential is the pinned one (oracle/exp_cr.py, csrc/worldgen.hpp
  exp_cr: correctly rounded, the same IEEE operations on both sides -- np.exp itself is SVML on AVX512 hosts, libm
  elsewhere, and the first GPU run of this test measured ocml's exp a third flavour: 294 of 4096 sigmoid values of a
  world one to three ulps away from numpy's): bit-exact on every cell of a 64x64 and of a 256x256 world, three seeds,
  on a dense sweep of the argument range and on the tiny arguments where 1 + x is a rounding tie."""
  from oracle.exp_cr import exp_cr, sigmoid
  d2 = np.arange(0, 2 * 256 * 256 + 1, dtype=np.float64)
  got = _eval(2, None, d2)
  assert np.array_equal(got.view(np.uint64), (4 - np.sqrt(d2)).view(np.uint64))
  total = 0
  for seed in (0, 1234, 2147483646, 1922490873):
    o = noise.OpenSimplex(seed)
    for area in (64, 256):
      xs, ys = np.meshgrid(np.arange(float(area)), np.arange(float(area)), indexing='ij')
      x, y = xs.ravel(), ys.ravel()
      start = 4 - np.sqrt((x - area // 2) ** 2 + (y - area // 2) ** 2)
      start = start + 2 * o.noise3_many(x / 3, y / 3, np.full(x.size, 8.0))
      want, got = sigmoid(start), _eval(1, None, start)
      assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (seed, area, int((got != want).sum()))
      total += x.size
  rs = np.random.RandomState(2)
  sweep = np.concatenate([np.linspace(-190.0, 12.0, 400001), rs.uniform(-1e-6, 1e-6, 50000), [k * 2.0 ** -56 for k in range(-3000, 3000)], [0.0, -0.0]])
  got, want = _eval(3, None, sweep), exp_cr(sweep)
  assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), int((got.view(np.uint64) != want.view(np.uint64)).sum())
  print(f'[noise] pinned sigmoid bit-exact on {total} worldgen arguments, exp_cr on {len(sweep)} swept arguments')
