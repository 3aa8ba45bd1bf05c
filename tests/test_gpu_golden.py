"""-m gpu: the HIP path, through crafter_amd.Env, against the reference's own outputs (tests/golden)."""
import pytest

from tests import adapters
from tests.parity import golden_cases, replay_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', golden_cases())
def test_hip_reproduces_reference_fixture(name):
  replay_golden(name, adapters.HipAdapter)
