"""ORACLE support (test infrastructure): import the UNTOUCHED reference from /root/reference.

Only usable in the build container (the GPU box has no /root/reference).  It is used to
(a) validate oracle/crafter_oracle.py against the real code and (b) generate the golden
fixtures under tests/golden/ (tools/make_golden.py).  Nothing in the product, bench.py,
smoke() or the ``-m gpu`` tests imports this module.

Three import shims (oracle/refshim/): ruamel.yaml -> PyYAML, imageio -> Pillow,
opensimplex -> oracle/noise.py (the one piece of arithmetic the reference gets from an
absent third-party package).

Canonicalisation (SURVEY.md section 0 trap 1, section 8c): the reference iterates a Python
``set`` of objects when it picks a creature to despawn (env.py:162,176 via engine.py:36),
so its trajectories depend on heap addresses.  We patch ``World.chunks`` -- in this
harness, never in the reference tree -- to return chunk members sorted by object slot
(a legal set order); dict (chunk-key) order is untouched.
"""
import os
import pathlib
import sys

REFERENCE_ROOT = pathlib.Path(os.environ.get('CRAFTER_REFERENCE', '/root/reference'))
_SHIMS = pathlib.Path(__file__).parent / 'refshim'


def available():
  return (REFERENCE_ROOT / 'crafter' / 'env.py').exists()


_crafter = None


def load():
  """Returns the reference ``crafter`` module (imported once, canonicalised)."""
  global _crafter
  if _crafter is not None:
    return _crafter
  if not available():
    raise RuntimeError(f'reference tree not found at {REFERENCE_ROOT}')
  sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree
  repo = str(pathlib.Path(__file__).resolve().parent.parent)
  for p in (str(REFERENCE_ROOT), str(_SHIMS), repo):
    if p not in sys.path:
      sys.path.insert(0, p)
  # shims must win over any real package of the same name, reference must win for 'crafter'
  sys.path.remove(str(_SHIMS))
  sys.path.insert(0, str(_SHIMS))
  import crafter  # noqa: E402  (the reference package)
  from crafter import engine

  def _slot(world, obj):
    return int(world._obj_map[tuple(obj.pos)])

  def chunks(self):
    return {key: sorted(members, key=lambda o: _slot(self, o))
            for key, members in self._chunks.items()}

  engine.World.chunks = property(chunks)
  _crafter = crafter
  return crafter


def make_env(**kwargs):
  return load().Env(**kwargs)
