import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_pure():
    return load_golden("pure.json")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with g++."""
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hip_lib():
    """Make sure libpcsaft_hip.so exists (cross-compiles without a GPU)."""
    from feos_torch_amd import build

    build.build()
    from feos_torch_amd import _lib

    return _lib.lib()


@pytest.fixture(scope="session", autouse=True)
def _host_extension():
    """The native gc row encoder (g++, host code) is built on demand like the oracle."""
    from feos_torch_amd import build

    build.build_host()
