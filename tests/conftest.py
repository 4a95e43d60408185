import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library builds its per-load structures (tile index, tile-major store) at the FOURTH count of a load (lsg_set_layout_policy); most
# tests count a load once.  The suite runs with the eager policy so that every GPU test goes through the streaming count — the form
# the benchmark measures — while tests/test_paths_gpu.py pins all three forms of the count, and the default policy, against each other.
os.environ.setdefault("LSG_LAYOUT", "eager")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One Engine for the whole GPU session (fails loudly if the HIP library / GPU is missing)."""
    from longsom_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()
