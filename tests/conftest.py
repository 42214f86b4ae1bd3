import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port_oracle():
    from oracle.oracle import PortOracle
    return PortOracle()


@pytest.fixture(scope="session")
def devlog_oracle():
    """The port with the HIP kernels' table-driven log10 in its lookups: checker of the conditioning argument."""
    from oracle.oracle import PortOracle
    return PortOracle(log10="device")


@pytest.fixture(scope="session")
def ref_oracle():
    """The compiled reference Fortran (oracle/_ref/libc2ray_ref.so: built in the container, travels to the GPU
    box as a built file), or None where it was never built."""
    from oracle.oracle import REF_SO, RefOracle
    return RefOracle() if os.path.exists(REF_SO) else None


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
