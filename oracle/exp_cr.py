"""ORACLE (test infrastructure).  The exponential of worldgen.py:27 -- ``start = 1 / (1 + np.exp(-start))`` -- pinned.

Why: ``np.exp`` is not one function.  On x86-64 Linux with AVX512 numpy evaluates float64 exp with Intel SVML, elsewhere
with the C library's ``exp``; the two differ in the last bit on ~4 % of worldgen's arguments (and the GPU's ocml ``exp``
from both).  Almost always that changes nothing: the value is compared with 0.5 and fed into ``water`` / ``mountain``,
which are compared with thresholds.  But the comparison ``start > 0.5`` has STRUCTURAL ties: on the four cells at distance
exactly 4 from the player the argument is ``2 * simplex(...)``, and the noise vanishes (up to ~1e-17 of rounding dust) on
about one such cell in 6000 -- there ``exp`` returns 1 - 1 ulp or 1 - 2 ulp depending on whose ``exp`` it is, ``start``
becomes 0.5 or 0.5000000000000001, the cell takes another branch of worldgen.py:36-61 and every later ``uniform()`` draw
of the world shifts (found by tests/test_gpu_pool.py: seed 7327, episode 10 -- this container's numpy (SVML) makes cell
(32, 28) grass, glibc's and the GPU's exp make it a tree candidate).  The reference's terrain is therefore a function of the
host CPU; the oracle and the device pin it to the one flavour that is defined without reference to an implementation:
the CORRECTLY ROUNDED exponential.

``exp_cr`` evaluates exp in double-double arithmetic (error < 2^-96, i.e. correctly rounded except for arguments whose
exponential lies within 2^-96 of a rounding boundary: arguments of magnitude <= 2^-52, where 1 + x can be an exact tie,
are rounded by an exact rule; tests/test_exp_cr.py checks 10^5 arguments, random and structured, against 100-digit
arithmetic) using only IEEE +, -, * (no fused multiply-add): csrc/worldgen.hpp ``exp_cr``
performs the very same operations in the same order, so the device reproduces it bit for bit (-m gpu:
tests/test_gpu_noise.py).  Vectorised over numpy arrays.

  x = k ln2 + r, ln2 = L1 + L2 + L3 (42 + 53 + 53 bits; k L1 is exact);  r' = r / 256
  exp(r') = sum_{n <= 8} r'^n / n!  (Horner, double-double coefficients; the tail is < 2^-104)
  exp(x) = 2^k exp(r')^256  (eight squarings)
"""
import numpy as np

SPLIT = 134217729.0                      # 2^27 + 1 (Veltkamp)
INV_LN2 = float.fromhex('0x1.71547652b82fep+0')
L1 = float.fromhex('0x1.62e42fefa3800p-1')
L2 = float.fromhex('0x1.ef35793c76730p-45')
L3 = float.fromhex('0x1.f97b57a079a19p-103')
# 1 / n! as double-double (hi, lo), n = 0 .. 8
COEF = [
    (1.0, 0.0),
    (1.0, 0.0),
    (0.5, 0.0),
    (float.fromhex('0x1.5555555555555p-3'), float.fromhex('0x1.5555555555555p-57')),
    (float.fromhex('0x1.5555555555555p-5'), float.fromhex('0x1.5555555555555p-59')),
    (float.fromhex('0x1.1111111111111p-7'), float.fromhex('0x1.1111111111111p-63')),
    (float.fromhex('0x1.6c16c16c16c17p-10'), float.fromhex('-0x1.f49f49f49f49fp-65')),
    (float.fromhex('0x1.a01a01a01a01ap-13'), float.fromhex('0x1.a01a01a01a01ap-73')),
    (float.fromhex('0x1.a01a01a01a01ap-16'), float.fromhex('0x1.a01a01a01a01ap-76')),
]


def _split(a):
  t = SPLIT * a
  hi = t - (t - a)
  return hi, a - hi


def _two_prod(a, b):
  p = a * b
  ah, al = _split(a)
  bh, bl = _split(b)
  e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
  return p, e


def _two_sum(a, b):
  s = a + b
  bb = s - a
  return s, (a - (s - bb)) + (b - bb)


def _quick_two_sum(a, b):
  s = a + b
  return s, b - (s - a)


def _dd_mul(ah, al, bh, bl):
  p, e = _two_prod(ah, bh)
  e = e + (ah * bl + al * bh)
  return _quick_two_sum(p, e)


def _dd_add(ah, al, bh, bl):
  s, e = _two_sum(ah, bh)
  e = e + (al + bl)
  return _quick_two_sum(s, e)


def exp_cr(x):
  """Correctly rounded exp(x) (see the module text) for float64 scalars or arrays, |x| < 700."""
  x = np.asarray(x, np.float64)
  with np.errstate(all='ignore'):
    k = np.rint(x * INV_LN2)
    a = x - k * L1
    p2, e2 = _two_prod(k, np.float64(L2))
    s, t = _two_sum(a, -p2)
    t = (t - e2) - k * L3
    rh, rl = _quick_two_sum(s, t)
    rh, rl = rh * 0.00390625, rl * 0.00390625
    ah, al = np.full_like(rh, COEF[8][0]), np.full_like(rh, COEF[8][1])
    for n in range(7, -1, -1):
      ah, al = _dd_mul(ah, al, rh, rl)
      ah, al = _dd_add(ah, al, np.float64(COEF[n][0]), np.float64(COEF[n][1]))
    for _ in range(8):
      ah, al = _dd_mul(ah, al, ah, al)
    out = np.ldexp(ah, k.astype(np.int64))
    # |x| <= 2^-52: exp(x) = 1 + x + d with 0 < d < 2^-105, beyond the reach of a double-double when 1 + x falls exactly
    # half way between two doubles -- d then decides, upwards.  (s, e) = 1 + x exactly; e == +half the gap above s is that
    # case, and the only one: e is a multiple of ulp(x) > d, so e + d crosses no other rounding boundary.
    tiny = np.abs(x) <= 2.220446049250313e-16
    s, e = _two_sum(np.float64(1.0), x)
    half_up = np.where(s < 1.0, 5.551115123125783e-17, 1.1102230246251565e-16)
    s = np.where(e == half_up, np.nextafter(s, np.inf), s)
    out = np.where(tiny, s, out)
  return out if out.ndim else np.float64(out)


def sigmoid(x):
  """worldgen.py:27 with the pinned exponential: 1 / (1 + exp(-x))."""
  return 1 / (1 + exp_cr(-np.asarray(x, np.float64)))
