"""Shim for the subset of imageio the reference imports (engine.py:5, recorder.py:5)."""
from . import v3  # noqa: F401


def mimsave(*args, **kwargs):
  raise NotImplementedError('imageio shim: video writing is out of scope')


def imsave(path, array):
  from PIL import Image
  Image.fromarray(array).save(path)
