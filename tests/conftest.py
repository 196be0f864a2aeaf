import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from zkmi_loader import load_pkg  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def zk(pkg):
    """The C-ABI library.  Missing .so is a hard failure, never a skip."""
    return pkg.Zkmi()


@pytest.fixture(scope="session")
def ctx(zk):
    if zk.device_count() <= 0:
        pytest.fail("gpu test selected but no HIP device is visible")
    c = zk.context(0)
    yield c
    c.close()
