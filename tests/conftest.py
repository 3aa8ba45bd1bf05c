import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')


@pytest.fixture(scope='session', autouse=True)
def _oracle_c_helper():
  """The oracle's C noise helper is test infrastructure; build it if the box lacks it."""
  from oracle import noise
  try:
    noise.build()
  except Exception:  # gcc missing: the pure-Python twin is used instead
    pass
  yield
