import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_context_first():
    """torch (used by some -m gpu tests for device buffers, exchange buffers and collectives) brings its own copy of the
    HIP runtime and must initialise it BEFORE the engine's library initialises /opt/rocm's: the other order leaves torch
    with "No HIP GPUs are available" (seen whenever the first GPU test of a session created an engine before anything
    touched torch.cuda).  Session-wide and automatic, so that no test order can get it wrong; nothing happens on a box
    without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield


@pytest.fixture(scope="session")
def engine_factory(_torch_hip_context_first):
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
