import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both native libraries are built before any test; the HIP one cross-compiles without a GPU."""
    import __graft_entry__ as g
    g.build()


def scene_path(name):
    return os.path.join(SCENES, name, "statex_00001.xml")
