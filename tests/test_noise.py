"""OpenSimplex restatement (oracle/opensimplex_ref.py, oracle/osimplex.c).

Pinned: seeding/permutation and the 2-D path, by the two known answers of the upstream README.
NOT pinned ("parity unpinned"): the 3-D path -- no output of the real package is available in
this image; it is checked for self-consistency, against the survey's provisional vectors and
between the Python and C twins.  If ``opensimplex`` is ever importable the last test pins it.
"""
import importlib.util
import sys

import numpy as np
import pytest

from oracle import noise
from oracle import opensimplex_ref as ref


def test_perm_check_values():
  assert ref.make_perm(0)[0][:8] == [254, 50, 92, 24, 36, 10, 190, 16]
  assert ref.make_perm(1234)[0][:8] == [250, 28, 6, 27, 142, 36, 8, 124]
  for seed in (0, 1, 1234, 2 ** 31 - 2):
    perm, pg3 = ref.make_perm(seed)
    assert sorted(perm) == list(range(256))
    assert pg3 == [(p % 24) * 3 for p in perm]


def test_upstream_readme_known_answers_2d():
  # opensimplex >= 0.4 README: seed(1234); noise2(10, 10)
  assert ref.OpenSimplex(1234).noise2(10, 10) == 0.580279369186297
  # opensimplex <= 0.3 README: OpenSimplex() (seed 0) .noise2d(10, 10) -> 0.732051569572 (12 digits)
  assert abs(ref.OpenSimplex(0).noise2d(10, 10) - 0.732051569572) < 5e-13


PROVISIONAL_3D = [  # SURVEY.md App. B (surveyor's independent restatement)
    (0, (0.5, 0.25, 3), -0.036162379490133276),
    (0, (10 / 15, 20 / 15, 3), 0.16485074844045103),
    (0, (63 / 5, 2 / 5, 0), -0.3796337708840147),
    (0, (-1.25, 7.5, 8), -0.4065090414004033),
    (0, (31 / 7, 33 / 7, 6), -0.5727193528836819),
    (2147483646, (0.5, 0.25, 3), -0.49037267938015144),
    (2147483646, (126 / 3, (63 / 5) / 3, 7), -0.19026982645986668),
    (2147483646, (4.2, 4.2, 1), 0.2716172401614128),
]


@pytest.mark.parametrize('seed,args,want', PROVISIONAL_3D)
def test_provisional_3d_vectors(seed, args, want):
  assert ref.OpenSimplex(seed).noise3(*args) == want
  assert noise.OpenSimplex(seed).noise3(*args) == want


def test_3d_invariants():
  o = ref.OpenSimplex(1234)
  assert abs(o.noise3(0, 0, 0)) < 1e-60
  rs = np.random.RandomState(0)
  pts = rs.uniform(-40, 40, size=(3000, 3))
  vals = np.array([o.noise3(*p) for p in pts])
  assert -1 < vals.min() < -0.6 and 0.6 < vals.max() < 1
  # continuity along a line across many simplex cells
  t = np.linspace(0, 6, 6001)
  line = np.array([o.noise3(0.3 + 1.7 * s, -2.0 + 0.9 * s, 1.1 * s) for s in t])
  assert np.abs(np.diff(line)).max() < 3.0 * (t[1] - t[0]) * 2.2


def test_c_twin_is_bit_identical():
  if not noise.have_c():
    pytest.skip('gcc helper not built')
  rs = np.random.RandomState(1)
  for seed in (0, 77, 2147483646):
    c, p = noise.OpenSimplex(seed), noise.OpenSimplex(seed, force_python=True)
    for _ in range(1500):
      x, y = rs.randint(0, 64) / rs.choice([3, 5, 6, 7, 8, 9, 15]), rs.randint(0, 128) / rs.choice([3, 5, 15])
      z = float(rs.randint(0, 9))
      assert c.noise3(x, y, z) == p.noise3(x, y, z)


@pytest.mark.skipif(importlib.util.find_spec('opensimplex') is None or 'refshim' in str(
    getattr(importlib.util.find_spec('opensimplex'), 'origin', '')), reason='real opensimplex package not installed')
def test_against_real_package_if_present():
  import opensimplex  # the real one
  real = opensimplex.OpenSimplex(seed=1234)
  fn = real.noise3 if hasattr(real, 'noise3') else real.noise3d
  mine = ref.OpenSimplex(1234)
  rs = np.random.RandomState(2)
  for _ in range(2000):
    x, y, z = rs.uniform(-30, 30), rs.uniform(-30, 30), float(rs.randint(0, 9))
    assert fn(x, y, z) == mine.noise3(x, y, z)


def test_kernel_noise3_is_bit_identical_to_the_oracle():
  """csrc/simplex.hpp (the kernels' noise3, compiled for the CPU harness) against oracle/osimplex.c on random
  points and on the exact arguments worldgen passes (worldgen.py:21-61): every bit of every value."""
  import ctypes as C
  from tests.hostsim import driver
  lib = driver.lib()
  rs = np.random.RandomState(11)
  pts = [rs.uniform(-60, 60, size=(120000, 3))]
  xs, ys = np.meshgrid(np.arange(64.0), np.arange(64.0), indexing='ij')
  for sx, sy, z in ((15, 15, 3), (5, 5, 3), (15, 15, 0), (5, 5, 0), (8, 8, 3), (6, 6, 7), (1, 1, 8), (2, 2, 6), (4, 4, 9), (5, 5, 7)):
    pts.append(np.stack([xs.ravel() / sx, ys.ravel() / sy, np.full(xs.size, float(z))], 1))
  pts.append(np.stack([2 * xs.ravel(), ys.ravel() / 5, np.full(xs.size, 7.0)], 1))   # horizontal tunnels
  pts.append(np.stack([xs.ravel() / 5, 2 * ys.ravel(), np.full(xs.size, 7.0)], 1))   # vertical tunnels
  # ties: points whose in-cell coordinates are equal or sum to exactly 1 or 2 -- where the strictness of every one of the
  # region's comparisons (>= vs >, <= vs <) decides which extra vertices contribute (simplex.hpp simplex_extras)
  g = np.arange(-8, 9) / 4.0
  gx, gy, gz = np.meshgrid(g, g, g, indexing='ij')
  pts.append(np.stack([gx.ravel(), gy.ravel(), gz.ravel()], 1))
  h = np.arange(-9, 10) / 3.0
  hx, hy, hz = np.meshgrid(h, h, h, indexing='ij')
  pts.append(np.stack([hx.ravel(), hy.ravel(), hz.ravel()], 1))
  u = rs.uniform(0, 1, size=(20000, 2))
  base = rs.randint(-20, 20, size=(20000, 3)).astype(np.float64)
  pts.append(base + np.stack([u[:, 0], u[:, 0], u[:, 1]], 1))   # xins == yins before the skew
  pts.append(base + np.stack([u[:, 0], u[:, 1], u[:, 1]], 1))
  pts.append(base + np.stack([u[:, 1], u[:, 0], u[:, 1]], 1))
  p = np.ascontiguousarray(np.concatenate(pts))
  pd = C.POINTER(C.c_double)
  for seed in (0, 1234, 2147483646):
    o = noise.OpenSimplex(seed)
    want = o.noise3_many(p[:, 0], p[:, 1], p[:, 2])
    perm8 = np.array(ref.make_perm(seed)[0], np.uint8)
    got = np.empty(len(p), np.float64)
    a, b, c = (np.ascontiguousarray(p[:, k]) for k in range(3))
    lib.hostsim_noise3(perm8.ctypes.data_as(C.c_void_p), a.ctypes.data_as(pd), b.ctypes.data_as(pd), c.ctypes.data_as(pd),
                       got.ctypes.data_as(pd), len(p))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), seed
