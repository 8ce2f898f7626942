import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine_factory():
    """Engine constructor for -m gpu tests.  Fails loudly (no fallback) when the HIP library or device is absent."""
    import pos_evolution_amd as pea

    made = []

    def make(**cfg):
        e = pea.Engine(**cfg)
        made.append(e)
        return e

    yield make
    for e in made:
        e.close()
