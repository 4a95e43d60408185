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
    """One Engine for the whole GPU session (fails loudly if the HIP library / GPU is missing).  Product mode: the tile store is
    the only resident copy of the events; a test that wants the arrays back (reads_to_host) asks with the kept_reads fixture."""
    from longsom_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


@pytest.fixture
def kept_reads(engine):
    """loads made inside the test also keep the compact events beside the store (lsg_set_keep_reads), for reads_to_host()"""
    engine.set_keep_reads(True)
    yield engine
    engine.set_keep_reads(False)
