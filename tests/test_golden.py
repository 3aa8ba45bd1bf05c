"""CPU: the oracle (and the kernel bodies run through tests/hostsim) against fixtures produced by
the real reference (tests/golden, tools/make_golden.py).  This is what pins the oracle on a machine
without /root/reference, e.g. the GPU box."""
import pytest

from tests import adapters
from tests.parity import golden_cases, replay_golden


@pytest.mark.parametrize('name', golden_cases())
def test_oracle_reproduces_reference_fixture(name):
  replay_golden(name, adapters.OracleAdapter)


@pytest.mark.parametrize('name', golden_cases())
def test_kernel_bodies_on_cpu_reproduce_reference_fixture(name):
  replay_golden(name, adapters.HostSimAdapter)
