import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def golden_path(name: str) -> str:
    return os.path.join(ROOT, "tests", "golden", name.replace("+", "p") + ".npz")


@pytest.fixture(scope="session")
def amd_lib():
    from oracle import cases
    return cases.lib_namespace("amd")
