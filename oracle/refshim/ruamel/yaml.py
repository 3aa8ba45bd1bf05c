"""Shim: ``ruamel.yaml.YAML(typ='safe', pure=True).load(text)`` -> PyYAML safe_load.
Used only to import the untouched reference (constants.py:3-8) in this container."""
import yaml as _pyyaml


class YAML:

  def __init__(self, typ=None, pure=False):
    pass

  def load(self, text):
    return _pyyaml.safe_load(text)
