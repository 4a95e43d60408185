import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One Engine for the whole GPU session (fails loudly if the HIP library / GPU is missing)."""
    from longsom_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()
