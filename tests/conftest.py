import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU cases")


@pytest.fixture(scope="session", autouse=True)
def _build_checker():
    """Builds the CPU checker (oracle/) once per session; test infrastructure only."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def root():
    return ROOT


def ref_tool(name):
    p = os.path.join(ROOT, "oracle", "_ref", name)
    return p if os.path.exists(p) else None


@pytest.fixture(scope="session")
def ref_index_and_search():
    p = ref_tool("index_and_search")
    if not p:
        pytest.skip("oracle/_ref/index_and_search not built (reference sources absent)")
    return p
