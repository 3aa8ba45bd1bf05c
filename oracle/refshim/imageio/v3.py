"""Shim: ``imageio.v3.imread(bytes)`` -> Pillow decode (all reference assets are 8-bit RGB/RGBA)."""
import io

import numpy as np
from PIL import Image


def imread(data):
  if isinstance(data, (bytes, bytearray)):
    data = io.BytesIO(data)
  return np.array(Image.open(data))
