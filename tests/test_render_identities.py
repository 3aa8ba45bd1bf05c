"""Arithmetic identities the renderer's kernels rely on (crafter_amd/csrc/render.hpp light(), mt19937.hpp
mt_uniform_32_127): each rewrites an expression of the reference (engine.py:189-209, Pillow's ImagingBlend) into a cheaper
one that yields the same bits.  Checked here exhaustively, or on a dense sample where the domain is 2^53."""
import numpy as np
from crafter_amd import tables


def test_pillow_desaturate_blend_is_an_integer_division():
  # ImageEnhance.Color(0.4): (u8)(L + 0.4f * (c - L)) in C float == floor((3 L + 2 c) / 5) == ((3 L + 2 c) * 13108) >> 16
  lum = np.arange(256, dtype=np.int32)[:, None]
  c = np.arange(256, dtype=np.int32)[None, :]
  ref = (lum.astype(np.float32) + np.float32(0.4) * (c - lum).astype(np.float32)).astype(np.int32)
  t = 3 * lum + 2 * c
  assert np.array_equal(ref, t // 5)
  assert np.array_equal(ref, (t * 13108) >> 16)
  hi = ((t.astype(np.uint64) << np.uint64(8)) * np.uint64(13108 << 8)) >> np.uint64(32)   # the kernel's v_mul_hi_u32_u24 form
  assert np.array_equal(ref, hi.astype(np.int32))
  assert (13108 << 8) < (1 << 24) and (int(t.max()) << 8) < (1 << 24)


def test_tint_and_daylight_mix_fold_into_one_scaling():
  # (1 - D) * (0.5 * e + 0.5 * tint) == ((1 - D) * 0.5) * (e + tint) for every step's daylight, every e, the three tints
  D = np.unique(tables.daylight_table(10000))
  iD = (1.0 - D)[:, None, None]
  e = np.arange(256, dtype=np.float64)[None, :, None]
  tint = np.array([0.0, 16.0, 64.0])[None, None, :]
  assert np.array_equal(iD * (0.5 * e + 0.5 * tint), (iD * 0.5) * (e + tint))


def test_uniform_32_127_with_one_multiply():
  # 32 + 95 * (X * 2^-53) == 32 + X * (95 * 2^-53), X = (a >> 5) * 2^26 + (b >> 6) exact in binary64
  rng = np.random.RandomState(5)
  a = rng.randint(0, 2 ** 32, size=2_000_000, dtype=np.uint64)
  b = rng.randint(0, 2 ** 32, size=2_000_000, dtype=np.uint64)
  edge = np.array([0, 1, 31, 32, 63, 64, 2 ** 32 - 1, 2 ** 31, 2 ** 31 - 1], dtype=np.uint64)
  a = np.concatenate([a, np.repeat(edge, len(edge))])
  b = np.concatenate([b, np.tile(edge, len(edge))])
  X = ((a >> np.uint64(5)) * np.uint64(2 ** 26) + (b >> np.uint64(6)))
  assert int(X.max()) < 2 ** 53
  Xf = (a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)
  assert np.array_equal(Xf, X.astype(np.float64))
  ref = 32.0 + 95.0 * (Xf / 9007199254740992.0)   # RandomState.uniform(32, 127): low + (high - low) * random_sample()
  assert np.array_equal(ref, 32.0 + Xf * (95.0 / 9007199254740992.0))


def test_probability_thresholds_in_the_integer_domain():
  """csrc/mt19937.hpp mt_prob53 / Env::uniform_below: `uniform() < p` (objects.py:277,298,299,333,336,338,339) as the
  integer comparison X < ceil(p * 2^53) on the 53-bit X of random_sample() -- on both sides of every threshold, through
  the very float expression numpy evaluates, and on random words."""
  from fractions import Fraction
  rs = np.random.RandomState(0)
  for p in (0.5, 0.9, 0.8, 0.6, 0.3, 0.2):
    t = p * 9007199254740992.0           # the constexpr's arithmetic: exact scaling ...
    f = int(t)
    below = f + 1 if float(f) < t else f
    assert below == -(-Fraction(p) * 2 ** 53 // 1), p      # ... = ceil(p * 2^53) in exact arithmetic
    xs = [below + d for d in range(-3, 4)] + [0, 2 ** 53 - 1] + [int(v) for v in rs.randint(0, 2 ** 53, size=2000, dtype=np.int64)]
    for x in xs:
      a, b = (x >> 26) << 5, (x & (2 ** 26 - 1)) << 6    # words whose (a >> 5, b >> 6) give X back
      u = (float(a >> 5) * 67108864.0 + float(b >> 6)) / 9007199254740992.0
      assert (u < p) == (x < below), (p, x)
