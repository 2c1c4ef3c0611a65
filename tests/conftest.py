import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def fixtures_dir():
    return os.path.join(ROOT, "tests", "fixtures")


@pytest.fixture
def devopt():
    """devopt(name, value) sets a developer option of the library for this test (value None removes it); all options are removed
    afterwards. The library reads no algorithm switch from the environment (include/graphminer_amd.h gm_dev_option)."""
    from graphminer_amd._lib import dev_option

    yield dev_option
    dev_option(None)
