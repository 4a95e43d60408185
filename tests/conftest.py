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


FORMS = {"store": {"LSG_LAYOUT": "eager"}, "index": {"LSG_LAYOUT": "eager", "LSG_NO_TM": "1"}, "scatter": {"LSG_LAYOUT": "never"}}


@pytest.fixture(params=sorted(FORMS))
def count_form(request, monkeypatch):
    """runs the test once per form of the count (tile-major store / tile index / scatter + sort): the library reads these variables
    at every lsg_pileup_count"""
    for k in ("LSG_LAYOUT", "LSG_NO_TM", "LSG_NO_INDEX"):
        monkeypatch.delenv(k, raising=False)
    for k, v in FORMS[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param
