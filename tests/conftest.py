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


@pytest.fixture(autouse=True, scope="module")
def _release_captured_steps():
    """Captured steps (hipGraph executables with their private memory pools, pinned mailboxes, gate
    words) die with the SVI objects that own them, and those sit in reference cycles: collect them at
    the end of every test module instead of whenever the cyclic collector gets to it -- the GPU suite
    creates several hundred captures in one process."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:      # noqa: BLE001
        pass


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
