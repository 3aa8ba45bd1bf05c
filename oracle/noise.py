"""ORACLE (test infrastructure).  ``OpenSimplex`` front-end used by the oracle env and by
the reference shim: C build when oracle/_build/libosimplex.so exists, else pure Python
(oracle/opensimplex_ref.py).  Both restate the same published algorithm and are checked
bit-identical in tests/test_noise.py.  3-D path: parity unpinned (see opensimplex_ref.py)."""
import ctypes
import pathlib

import numpy as np

from . import opensimplex_ref

_LIB_PATH = pathlib.Path(__file__).parent / '_build' / 'libosimplex.so'
_lib = None


def build(force=False):
  """Compile the C helper with gcc (called by __graft_entry__.build and tests/conftest)."""
  import subprocess
  src = pathlib.Path(__file__).parent / 'osimplex.c'
  if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
    subprocess.run(['make', '-C', str(src.parent), '-B', '_build/libosimplex.so'],
                   check=True, capture_output=True)
  return _LIB_PATH


def _load():
  global _lib
  if _lib is None and _LIB_PATH.exists():
    lib = ctypes.CDLL(str(_LIB_PATH))
    p16 = ctypes.POINTER(ctypes.c_int16)
    pd = ctypes.POINTER(ctypes.c_double)
    lib.osn_make_perm.argtypes = [ctypes.c_int64, p16, p16]
    lib.osn_make_perm.restype = None
    lib.osn_noise3.argtypes = [p16, p16, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.osn_noise3.restype = ctypes.c_double
    lib.osn_noise3_many.argtypes = [p16, p16, pd, pd, pd, pd, ctypes.c_int64]
    lib.osn_noise3_many.restype = None
    _lib = lib
  return _lib


def have_c():
  return _load() is not None


class OpenSimplex:
  """OpenSimplex(seed).noise3(x, y, z) -- the surface worldgen.py:11,84-87 uses."""

  def __init__(self, seed=0, force_python=False):
    self._py = opensimplex_ref.OpenSimplex(seed)
    lib = None if force_python else _load()
    self._lib = lib
    if lib is not None:
      self._perm = np.zeros(256, np.int16)
      self._pg3 = np.zeros(256, np.int16)
      p16 = ctypes.POINTER(ctypes.c_int16)
      self._pp = self._perm.ctypes.data_as(p16)
      self._pg = self._pg3.ctypes.data_as(p16)
      lib.osn_make_perm(int(seed), self._pp, self._pg)
      assert self._perm.tolist() == self._py.perm

  def noise3(self, x, y, z):
    if self._lib is None:
      return self._py.noise3(float(x), float(y), float(z))
    return self._lib.osn_noise3(self._pp, self._pg, float(x), float(y), float(z))

  def noise3_many(self, xs, ys, zs):
    xs = np.ascontiguousarray(xs, np.float64)
    ys = np.ascontiguousarray(ys, np.float64)
    zs = np.ascontiguousarray(zs, np.float64)
    out = np.empty(xs.shape, np.float64)
    if self._lib is None:
      flat = [self._py.noise3(a, b, c) for a, b, c in zip(xs.ravel(), ys.ravel(), zs.ravel())]
      out[...] = np.array(flat).reshape(xs.shape)
      return out
    pd = ctypes.POINTER(ctypes.c_double)
    self._lib.osn_noise3_many(self._pp, self._pg, xs.ctypes.data_as(pd), ys.ctypes.data_as(pd),
                              zs.ctypes.data_as(pd), out.ctypes.data_as(pd), xs.size)
    return out

  def noise2(self, x, y):
    return self._py.noise2(x, y)

  noise3d = noise3
  noise2d = noise2
