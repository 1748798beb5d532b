import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (test infrastructure), compiled on first use."""
    from oracle import pyoracle
    pyoracle.build_oracle()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def api():
    """The product binding (rtk_amd.api over librtk_amd.so). Builds the library first if a fresh
    checkout does not have it yet; never substitutes anything for it."""
    from rtk_amd import api as _api
    if not os.path.exists(_api.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    _api.lib()
    return _api
