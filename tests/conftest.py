import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def cham():
    """The product's Chameleon-shaped binding, initialised on cuda:0 (GPU tests only)."""
    if not _has_gpu():
        pytest.skip("no GPU visible")
    from dense_linear_app_amd import chameleon as ch

    ch.CHAMELEON_Init(1, 1)
    yield ch


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.lib()
    return oracle
