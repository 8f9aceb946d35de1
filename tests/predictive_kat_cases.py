"""tests/infer/test_predictive.py of the reference restated against the drop-in API (SURVEY 8f
rank 3: Predictive either side of the SVI path): manual Beta guide, AutoDelta, AutoDiagonalNormal
(samples and vectorised trace), one-hot observations, shapes, pyro.deterministic sites, the
get_mask() optimisation.  Statistical tolerances are the reference's."""
import contextlib

import numpy as np
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import poutine
from pyro_amd.infer import SVI, Predictive, Trace_ELBO
from pyro_amd.infer.autoguide import AutoDelta, AutoDiagonalNormal


def _close(a, b, rtol):
    np.testing.assert_allclose(torch.as_tensor(a).detach().cpu().numpy(),
                               torch.as_tensor(b).detach().cpu().numpy(), rtol=rtol)


def model(num_trials):
    with pyro.plate("data", num_trials.size(0)):
        phi_prior = dist.Uniform(num_trials.new_tensor(0.0), num_trials.new_tensor(1.0))
        success_prob = pyro.sample("phi", phi_prior)
        return pyro.sample("obs", dist.Binomial(num_trials, success_prob))


def one_hot_model(pseudocounts, classes=None):
    probs = pyro.sample("probs", dist.Dirichlet(pseudocounts))
    with pyro.plate("classes", classes.size(0) if classes is not None else 1, dim=-1):
        return pyro.sample("obs", dist.OneHotCategorical(probs), obs=classes)


def beta_guide(num_trials):
    phi_c0 = pyro.param("phi_c0", num_trials.new_tensor(5.0).expand([num_trials.size(0)]))
    phi_c1 = pyro.param("phi_c1", num_trials.new_tensor(5.0).expand([num_trials.size(0)]))
    with pyro.plate("data", num_trials.size(0)):
        pyro.sample("phi", dist.Beta(concentration0=phi_c0, concentration1=phi_c1))


def run_manual_guide(device, parallel, num_svi_steps=2000):
    pyro.clear_param_store(); pyro.set_rng_seed(0); torch.manual_seed(0)
    true_probs = torch.ones(5, device=device) * 0.7
    num_trials = torch.ones(5, device=device) * 400
    num_success = dist.Binomial(num_trials, true_probs).sample()
    conditioned = poutine.condition(model, data={"obs": num_success})
    svi = SVI(conditioned, beta_guide, pyro.optim.Adam(dict(lr=3.0)),
              Trace_ELBO(num_particles=100, vectorize_particles=True))
    for _ in range(num_svi_steps):
        svi.step(num_trials)
    pred = Predictive(model, guide=beta_guide, num_samples=10000, parallel=parallel,
                      return_sites=["_RETURN"])
    vals = pred(num_trials)["_RETURN"]
    _close(vals.mean(dim=0), torch.ones(5) * 280, rtol=0.1)


def run_auto_delta(device, parallel):
    pyro.clear_param_store(); pyro.set_rng_seed(0); torch.manual_seed(0)
    num_trials = torch.ones(5, device=device) * 1000
    num_success = dist.Binomial(num_trials, torch.ones(5, device=device) * 0.7).sample()
    conditioned = poutine.condition(model, data={"obs": num_success})
    guide = AutoDelta(conditioned)
    svi = SVI(conditioned, guide, pyro.optim.Adam(dict(lr=1.0)), Trace_ELBO())
    for _ in range(1000):
        svi.step(num_trials)
    pred = Predictive(model, guide=guide, num_samples=10000, parallel=parallel)
    vals = pred.get_samples(num_trials)["obs"]
    _close(vals.mean(dim=0), torch.ones(5) * 700, rtol=0.05)


def run_auto_diag_normal(device, return_trace):
    pyro.clear_param_store(); pyro.set_rng_seed(0); torch.manual_seed(0)
    num_trials = torch.ones(5, device=device) * 1000
    num_success = dist.Binomial(num_trials, torch.ones(5, device=device) * 0.7).sample()
    conditioned = poutine.condition(model, data={"obs": num_success})
    guide = AutoDiagonalNormal(conditioned)
    svi = SVI(conditioned, guide, pyro.optim.Adam(dict(lr=0.1)), Trace_ELBO())
    for _ in range(1000):
        svi.step(num_trials)
    pred = Predictive(model, guide=guide, num_samples=10000, parallel=True)
    if return_trace:
        vals = pred.get_vectorized_trace(num_trials).nodes["obs"]["value"]
    else:
        vals = pred.get_samples(num_trials)["obs"]
    _close(vals.mean(dim=0), torch.ones(5) * 700, rtol=0.05)


def run_one_hot(device):
    pyro.clear_param_store(); pyro.set_rng_seed(0); torch.manual_seed(0)
    pseudocounts = torch.ones(3, device=device) * 0.1
    true_probs = torch.tensor([0.15, 0.6, 0.25], device=device)
    classes = dist.OneHotCategorical(true_probs).sample((10000,))
    guide = AutoDelta(one_hot_model)
    svi = SVI(one_hot_model, guide, pyro.optim.Adam(dict(lr=0.1)), Trace_ELBO())
    for _ in range(1000):
        svi.step(pseudocounts, classes=classes)
    posterior_samples = Predictive(guide, num_samples=10000).get_samples(pseudocounts)
    vals = Predictive(one_hot_model, posterior_samples).get_samples(pseudocounts)["obs"]
    _close(vals.mean(dim=0), true_probs.unsqueeze(0), rtol=0.1)


def run_shapes(device, parallel):
    pyro.clear_param_store(); pyro.set_rng_seed(0)
    num_samples = 10

    def m():
        x = pyro.sample("x", dist.Normal(torch.zeros((), device=device), 1.0).expand([2]).to_event(1))
        with pyro.plate("plate", 5):
            loc, log_scale = x.unbind(-1)
            y = pyro.sample("y", dist.Normal(loc, log_scale.exp()))
        return dict(x=x, y=y)

    guide = AutoDiagonalNormal(m)
    vectorize = pyro.plate("_vectorize", num_samples, dim=-2)
    trace = poutine.trace(vectorize(guide)).get_trace()
    expected = poutine.replay(vectorize(m), trace)()
    actual = Predictive(m, guide=guide, return_sites=["x", "y"], num_samples=num_samples,
                        parallel=parallel)()
    assert set(actual) == set(expected)
    assert actual["x"].shape == expected["x"].shape
    assert actual["y"].shape == expected["y"].shape


def run_deterministic(device, with_plate, event_shape):
    pyro.clear_param_store(); pyro.set_rng_seed(0)

    def m(y=None):
        with (pyro.plate("plate", 3) if with_plate else contextlib.nullcontext()):
            x = pyro.sample("x", dist.Normal(torch.zeros((), device=device), 1.0)
                            .expand(event_shape).to_event(len(event_shape)))
            x2 = pyro.deterministic("x2", x ** 2, event_dim=len(event_shape))
        pyro.deterministic("x3", x2)
        return pyro.sample("obs", dist.Normal(x2, 0.1).to_event(x2.dim()), obs=y)

    y = torch.tensor(4.0, device=device)
    guide = AutoDiagonalNormal(m)
    svi = SVI(m, guide, pyro.optim.Adam(dict(lr=0.1)), Trace_ELBO())
    for _ in range(100):
        svi.step(y)
    actual = Predictive(m, guide=guide, return_sites=["x2", "x3"], num_samples=1000)()
    x2_batch = (3,) if with_plate else ()
    assert actual["x2"].shape == (1000,) + x2_batch + event_shape
    x3_batch = (1, 3) if with_plate else ()          # prepended 1: Pyro's shape semantics
    assert actual["x3"].shape == (1000,) + x3_batch + event_shape
    _close(actual["x2"].mean(), y, rtol=0.1)
    _close(actual["x3"].mean(), y, rtol=0.1)


def run_get_mask_optimization(device):
    z = lambda v: torch.tensor(v, device=device)   # noqa: E731

    def m():
        x = pyro.sample("x", dist.Normal(z(0.0), 1.0))
        pyro.sample("y", dist.Normal(x, 1.0), obs=z(0.0))
        called.add("model-always")
        if poutine.get_mask() is not False:
            called.add("model-sometimes")
            pyro.factor("f", x + 1)

    def g():
        x = pyro.sample("x", dist.Normal(z(0.0), 1.0))
        called.add("guide-always")
        if poutine.get_mask() is not False:
            called.add("guide-sometimes")
            pyro.factor("g", 2 - x)

    called = set()
    trace = poutine.trace(g).get_trace()
    poutine.replay(m, trace)()
    assert called == {"model-always", "guide-always", "model-sometimes", "guide-sometimes"}
    called = set()
    with poutine.mask(mask=False):
        trace = poutine.trace(g).get_trace()
        poutine.replay(m, trace)()
    assert called == {"model-always", "guide-always"}
    called = set()
    Predictive(m, guide=g, num_samples=2, parallel=True)()
    assert called == {"model-always", "guide-always"}
