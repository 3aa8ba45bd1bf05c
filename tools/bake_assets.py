#!/usr/bin/env python3
"""Bake the reference's DATA (not source) into travel-safe files.

Reads   /root/reference/crafter/data.yaml      (rule tables, reference constants.py:6-8)
        /root/reference/crafter/assets/*.png   (56 16x16 textures, reference engine.py:122-129)
Writes  crafter_amd/data/rules.json            (same mapping, JSON, key order preserved)
        crafter_amd/data/textures.npz          (name -> uint8 [16,16,3|4] exactly as Pillow decodes it)

The GPU box has no /root/reference, so the product loads these two files.  The rule
constants are never typed from memory of upstream crafter: this script is the only
producer of rules.json and it copies whatever the mounted reference's data.yaml holds
(trap 3).  Run again whenever the reference changes.
"""
import json
import pathlib
import sys

import numpy as np
import yaml
from PIL import Image

REF = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else '/root/reference/crafter')
OUT = pathlib.Path(__file__).resolve().parent.parent / 'crafter_amd' / 'data'


def main():
  OUT.mkdir(parents=True, exist_ok=True)
  rules = yaml.safe_load((REF / 'data.yaml').read_text())
  (OUT / 'rules.json').write_text(json.dumps(rules, indent=1) + '\n')
  textures = {}
  for path in sorted((REF / 'assets').glob('*.png')):
    image = np.array(Image.open(path))
    assert image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] in (3, 4), path
    textures[path.stem] = image
  np.savez_compressed(OUT / 'textures.npz', **textures)
  print(f'rules: {list(rules)}; textures: {len(textures)} -> {OUT}')


if __name__ == '__main__':
  main()
