import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gpu_engine():
    """The product library on a GPU box; fails loudly (no fallback) if it is missing."""
    from flac_amd import engine
    assert os.path.exists(engine.LIB_PATH), "libflacamd.so is not built"
    assert engine.device_count() > 0, "no HIP device visible"
    return engine
