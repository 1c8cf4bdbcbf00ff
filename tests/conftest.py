import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """tests never silently skip the native code: build (or find) the HIP library and the oracle first"""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def talos():
    import loik_amd
    return loik_amd.builtin_model("talos32")


@pytest.fixture(scope="session")
def panda7():
    import loik_amd
    return loik_amd.builtin_model("panda7")


@pytest.fixture(scope="session")
def panda9():
    import loik_amd
    return loik_amd.builtin_model("panda9")
