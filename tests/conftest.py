import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture
def oracle_backend(monkeypatch):
    """Route pyro_amd.kernels to the numpy oracle (TEST-ONLY; see tests/oracle_backend.py)."""
    import pyro_amd
    from tests import oracle_backend as ob

    ob.install(monkeypatch)
    pyro_amd.clear_param_store()
    pyro_amd.set_rng_seed(0)
    yield
    pyro_amd.clear_param_store()
