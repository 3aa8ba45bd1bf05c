"""The pinned exponential of the world generator (oracle/exp_cr.py; csrc/worldgen.hpp performs the same operations):
correctly rounded -- checked against 100-digit arithmetic -- where numpy's own np.exp (SVML on AVX512 hosts, the C
library's exp elsewhere) is not, and the one world of the test suite in which that decides a material."""
import math
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np

from oracle.exp_cr import exp_cr, sigmoid


def _cr(x):
  getcontext().prec = 100
  return float(Fraction(Decimal(float(x)).exp()))   # Fraction -> float rounds to nearest, ties to even: exact


def test_exp_cr_is_correctly_rounded():
  rs = np.random.RandomState(1)
  tiny = [k * 2.0 ** -58 for k in range(-4000, 4000)] + [k * 2.0 ** -50 for k in range(-3000, 3000)]
  few = [rs.randint(1, 2 ** 12) * 2.0 ** (-rs.randint(20, 70)) * (1 if rs.rand() < .5 else -1) for _ in range(10000)]
  xs = np.array(tiny + few + list(rs.uniform(-180, 10, 40000)) + list(rs.uniform(-1e-7, 1e-7, 10000)) + [0.0, -0.0, -1.638387376145862e-16])
  got = exp_cr(xs)
  want = np.array([_cr(v) for v in xs])
  bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
  assert not len(bad), (len(bad), xs[bad[:5]], got[bad[:5]], want[bad[:5]])
  assert exp_cr(0.5) == _cr(0.5) and isinstance(exp_cr(0.5), np.floating)


def test_the_cell_on_which_the_flavours_of_exp_disagree():
  """Seed 7327, episode 10 (simplex seed 1922490873), cell (32, 28): distance exactly 4 from the player and the noise
  vanishes up to rounding dust, so start = 2 * 8.19e-17.  exp(-start) = 1 - 1.64e-16 lies between 1 - 2^-53 and 1 - 2^-52,
  nearer the former: the correctly rounded value is 0x1.fffffffffffffp-1, start becomes exactly 0.5, `start > 0.5` is
  false and the cell goes on to the tree branch (one uniform() draw).  An exp that returns 0x1.ffffffffffffep-1 (numpy on
  an AVX512 host does) makes it 0.5000000000000001 and the cell grass without a draw: every later draw of the world shifts."""
  from oracle import noise
  v = noise.OpenSimplex(1922490873).noise3(32 / 3, 28 / 3, 8)
  pre = 4 - np.sqrt(float((32 - 32) ** 2 + (28 - 32) ** 2)) + 2 * v
  assert 0 < pre < 1e-15
  assert exp_cr(-pre).hex() == '0x1.fffffffffffffp-1' == float(_cr(-pre)).hex() == math.exp(-pre).hex()
  assert sigmoid(pre) == 0.5


def test_kernel_exp_cr_is_the_oracle_exp_cr():
  """csrc/worldgen.hpp exp_cr compiled for the CPU harness (same operations, no fused multiply-add) against
  oracle/exp_cr.py, bit for bit -- the -m gpu twin is tests/test_gpu_noise.py."""
  import ctypes as C
  from tests.hostsim import driver
  lib = driver.lib()
  rs = np.random.RandomState(3)
  xs = np.concatenate([rs.uniform(-180, 10, 200000), rs.uniform(-1e-6, 1e-6, 20000), [k * 2.0 ** -56 for k in range(-3000, 3000)], [0.0, -0.0]])
  out = np.empty_like(xs)
  pd = C.POINTER(C.c_double)
  lib.hostsim_exp_cr(xs.ctypes.data_as(pd), out.ctypes.data_as(pd), len(xs))
  want = exp_cr(xs)
  assert np.array_equal(out.view(np.uint64), want.view(np.uint64))
