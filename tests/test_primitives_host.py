"""tests/test_primitives.py of the reference restated (CPU; the fused families go through the
oracle backend): bare primitives outside inference, the observe warning, obs_mask."""
import warnings

import pytest
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import poutine


@pytest.fixture(autouse=True)
def _host(monkeypatch):
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    pyro.clear_param_store()


def test_sample_ok():
    x = pyro.sample("x", dist.Normal(torch.tensor(0.0), 1.0))
    assert isinstance(x, torch.Tensor) and x.shape == ()


def test_observe_warn():
    with pytest.warns(RuntimeWarning):
        pyro.sample("x", dist.Normal(torch.tensor(0.0), 1.0), obs=torch.tensor(0.0))


def test_param_ok():
    x = pyro.param("x", torch.tensor(0.0))
    assert isinstance(x, torch.Tensor) and x.shape == ()


def test_deterministic_ok():
    with warnings.catch_warnings():
        warnings.simplefilter("error")           # a deterministic site is not "observing"
        x = pyro.deterministic("x", torch.tensor(0.0))
    assert isinstance(x, torch.Tensor) and x.shape == ()


@pytest.mark.parametrize("mask", [None, torch.tensor(True), torch.tensor([True]),
                                  torch.tensor([True, False, True])])
def test_obs_mask_shape(mask):
    data = torch.randn(3, 2)

    def model():
        with pyro.plate("data", 3):
            pyro.sample("y", dist.MultivariateNormal(torch.zeros(2), scale_tril=torch.eye(2)),
                        obs=data, obs_mask=mask)

    trace = poutine.trace(model).get_trace()
    y_dist = trace.nodes["y"]["fn"]
    assert y_dist.batch_shape == (3,) and y_dist.event_shape == (2,)


def test_obs_mask_semantics():
    """Where the mask holds the observation is scored and kept; elsewhere a latent is drawn and
    scored (pyro/primitives.py:94-122)."""
    data = torch.tensor([1.0, 2.0, 3.0, 4.0])
    mask = torch.tensor([True, False, True, False])

    def model():
        with pyro.plate("data", 4):
            return pyro.sample("y", dist.Laplace(torch.zeros(4), 1.0), obs=data, obs_mask=mask)

    trace = poutine.trace(model).get_trace()
    y = trace.nodes["y"]["value"]
    assert torch.equal(y[mask], data[mask]) and not torch.equal(y[~mask], data[~mask])
    assert trace.nodes["y_observed"]["is_observed"] and not trace.nodes["y_unobserved"]["is_observed"]
    trace.compute_log_prob()
    lp_obs, lp_un = trace.nodes["y_observed"]["log_prob"], trace.nodes["y_unobserved"]["log_prob"]
    assert (lp_obs[~mask] == 0).all() and (lp_obs[mask] != 0).all()
    assert (lp_un[mask] == 0).all() and (lp_un[~mask] != 0).all()
    with pytest.raises(ValueError, match="Invalid obs_mask shape"):
        def bad():
            with pyro.plate("data", 4):
                pyro.sample("y", dist.Laplace(torch.zeros(4), 1.0), obs=data,
                            obs_mask=torch.tensor([True, False, True]))
        poutine.trace(bad).get_trace()


def test_philox_stream_follows_torchs_default_generator(oracle_backend):
    """torch.manual_seed alone restarts the kernels' stream (reference: every draw goes through torch's
    generator, pyro/util.py:37-45 seeds nothing else); get / set_rng_state carry its position."""
    import torch
    import pyro_amd as pyro
    from pyro_amd import rng
    dev = torch.device("cpu")
    torch.manual_seed(5)
    a = rng.normal((6,), torch.float64, dev)
    b = rng.normal((6,), torch.float64, dev)
    assert rng.current_seed() == 5 and not torch.equal(a, b)
    torch.manual_seed(5)
    assert torch.equal(rng.normal((6,), torch.float64, dev), a)
    state = pyro.util.get_rng_state()
    c = rng.normal((6,), torch.float64, dev)
    assert torch.equal(c, b)
    torch.manual_seed(99)
    assert not torch.equal(rng.normal((6,), torch.float64, dev), a)
    pyro.util.set_rng_state(state)
    assert rng.current_seed() == 5 and torch.equal(rng.normal((6,), torch.float64, dev), b)
    pyro.set_rng_seed(5)
    assert torch.equal(rng.normal((6,), torch.float64, dev), a)
    pyro.set_rng_seed(5)                               # the seed already set: starts over all the same
    assert torch.equal(rng.normal((6,), torch.float64, dev), a)
