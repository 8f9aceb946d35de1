"""poutine.reparam with LocScaleReparam / TransformReparam (reference: tests/infer/reparam/test_loc_scale.py,
test_transform.py restated): the reparameterised program has the same distribution over the original site, the
auxiliary site carries the density, and HMC runs on it."""
import math

import pytest
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import poutine
from pyro_amd.infer.reparam import LocScaleReparam, TransformReparam


@pytest.fixture(autouse=True)
def _host(monkeypatch):
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    pyro.clear_param_store()


@pytest.mark.parametrize("centered", [0.0, 0.6, 1.0, torch.tensor(0.4), None])
@pytest.mark.parametrize("shape", [(), (4,), (3, 2)], ids=str)
def test_loc_scale_moments_and_structure(shape, centered):
    loc = torch.empty(shape).uniform_(-1.0, 1.0)
    scale = torch.empty(shape).uniform_(0.5, 1.5)

    def model():
        with pyro.plate_stack("plates", shape), pyro.plate("particles", 20000):
            return pyro.sample("x", dist.Normal(loc, scale))

    pyro.set_rng_seed(0)
    expected = model()
    pyro.set_rng_seed(1)
    reparam_model = poutine.reparam(model, config={"x": LocScaleReparam(centered)})
    tr = poutine.trace(reparam_model).get_trace()
    value = tr.nodes["x"]["value"]
    if isinstance(centered, float) and centered == 1.0:
        assert "x_decentered" not in tr.nodes and not tr.nodes["x"]["is_observed"]
    else:
        assert tr.nodes["x"]["is_observed"] and not tr.nodes["x_decentered"]["is_observed"]
        assert type(tr.nodes["x"]["fn"]).__name__ == "MaskedDistribution"        # a Delta that scores nothing
        if centered is None:
            assert pyro.param("x_centered").shape == ()                          # learnable, starts at 0.5
    for moment in (lambda v: v.mean(0), lambda v: v.std(0)):
        assert torch.allclose(moment(value), moment(expected), atol=0.05)


def test_loc_scale_density_of_the_auxiliary_site():
    """Conditioning the auxiliary site: log p(decentered) = log p(value) + (1 - c) log(scale)."""
    loc, scale, c = torch.tensor(0.7), torch.tensor(2.5), 0.3

    def model():
        return pyro.sample("x", dist.Normal(loc, scale))

    d = torch.tensor(-0.4)
    tr = poutine.trace(poutine.condition(poutine.reparam(model, config={"x": LocScaleReparam(c)}),
                                         data={"x_decentered": d})).get_trace()
    value = tr.nodes["x"]["value"]
    assert torch.allclose(value, loc + scale ** (1 - c) * (d - c * loc))
    expected = dist.Normal(loc, scale).log_prob(value) + (1 - c) * math.log(2.5)
    assert abs(float(tr.log_prob_sum()) - float(expected)) < 1e-5


def test_observed_site_is_reparameterised_consistently():
    def model():
        return pyro.sample("x", dist.Normal(torch.tensor(1.0), torch.tensor(2.0)), obs=torch.tensor(3.0))

    tr = poutine.trace(poutine.reparam(model, config={"x": LocScaleReparam(0.0)})).get_trace()
    assert tr.nodes["x_decentered"]["is_observed"]
    assert torch.allclose(tr.nodes["x_decentered"]["value"], torch.tensor(1.0))      # (3 - 1) / 2
    assert torch.equal(tr.nodes["x"]["value"], torch.tensor(3.0))


def test_transform_reparam_of_a_log_normal():
    def model():
        with pyro.plate("particles", 20000):
            return pyro.sample("x", dist.LogNormal(torch.tensor(0.2), torch.tensor(0.5)))

    pyro.set_rng_seed(0)
    expected = model()
    tr = poutine.trace(poutine.reparam(model, config={"x": TransformReparam()})).get_trace()
    assert "x_base" in tr.nodes and tr.nodes["x"]["is_observed"]
    assert torch.allclose(tr.nodes["x"]["value"], tr.nodes["x_base"]["value"].exp())
    assert abs(float(tr.nodes["x"]["value"].log().mean()) - float(expected.log().mean())) < 0.02


def test_config_as_a_function_and_as_a_decorator_that_sees_the_arguments():
    seen = {}

    class Spy(LocScaleReparam):
        def apply(self, msg):
            seen["args_kwargs"] = self.args_kwargs
            return super().apply(msg)

    @poutine.reparam(config=lambda site: Spy(0.0) if site["name"] == "theta" else None)
    def model(n, flag=False):
        mu = pyro.sample("mu", dist.Normal(0.0, 5.0))
        with pyro.plate("J", n):
            return pyro.sample("theta", dist.Normal(mu, 2.0))

    tr = poutine.trace(model).get_trace(8, flag=True)
    assert "theta_decentered" in tr.nodes and "mu_decentered" not in tr.nodes
    assert seen["args_kwargs"] == ((8,), {"flag": True})


def test_hmc_runs_on_the_non_centred_funnel():
    from pyro_amd.infer import MCMC, NUTS
    y = torch.tensor([28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0])
    sigma = torch.tensor([15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0])

    @poutine.reparam(config={"theta": LocScaleReparam(centered=0.0)})
    def model():
        mu = pyro.sample("mu", dist.Normal(0.0, 5.0))
        tau = pyro.sample("tau", dist.HalfCauchy(5.0))
        with pyro.plate("J", 8):
            theta = pyro.sample("theta", dist.Normal(mu, tau))
            pyro.sample("obs", dist.Normal(theta, sigma), obs=y)

    pyro.set_rng_seed(0)
    mcmc = MCMC(NUTS(model, max_tree_depth=5), num_samples=40, warmup_steps=40)
    mcmc.run()
    samples = mcmc.get_samples()
    assert set(samples) == {"mu", "tau", "theta_decentered"}
    assert samples["theta_decentered"].shape == (40, 8) and torch.isfinite(samples["mu"]).all()
