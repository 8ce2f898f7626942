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

    # torch (used by the sharded tests for exchange buffers and collectives) initialises its HIP context first: a lazy
    # init after dozens of engines had come and gone was seen to fail once with "No HIP GPUs are available".
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    made = []

    def make(**cfg):
        e = pea.Engine(**cfg)
        made.append(e)
        return e

    yield make
    for e in made:
        e.close()
