import os
import sys

import pytest

# the oracle opens many tiny OpenMP regions; on a 256-thread host the fork/join cost dominates
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One jxlgpu context on device 0 through the C ABI.  Fails loudly without the HIP library."""
    from jxl_oxide_amd import runtime
    runtime.prime_gpu()
    ctx = runtime.Context(0)
    yield ctx
    ctx.close()
