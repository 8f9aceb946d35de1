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


def binomial_model(num_trials):
    """Per-row success probability phi ~ U(0, 1), obs ~ Binomial(num_trials, phi); returns obs."""
    rows = num_trials.size(0)
    zero, one = num_trials.new_tensor(0.0), num_trials.new_tensor(1.0)
    with pyro.plate("data", rows):
        phi = pyro.sample("phi", dist.Uniform(zero, one))
        return pyro.sample("obs", dist.Binomial(num_trials, phi))


model = binomial_model


def beta_guide(num_trials):
    rows = num_trials.size(0)
    shapes = [pyro.param(name, num_trials.new_full((rows,), 5.0)) for name in ("phi_c0", "phi_c1")]
    with pyro.plate("data", rows):
        pyro.sample("phi", dist.Beta(concentration0=shapes[0], concentration1=shapes[1]))


def one_hot_model(pseudocounts, classes=None):
    probs = pyro.sample("probs", dist.Dirichlet(pseudocounts))
    rows = 1 if classes is None else classes.size(0)
    with pyro.plate("classes", rows, dim=-1):
        return pyro.sample("obs", dist.OneHotCategorical(probs), obs=classes)


def _fresh():
    pyro.clear_param_store()
    pyro.set_rng_seed(0)
    torch.manual_seed(0)


def _fit_binomial(device, trials, make_guide, lr, steps, elbo):
    """Five rows with true success probability 0.7 and ``trials`` trials each; the model conditioned on one
    draw of the counts is fitted with the guide ``make_guide(conditioned_model)`` builds."""
    _fresh()
    num_trials = torch.full((5,), float(trials), device=device)
    counts = dist.Binomial(num_trials, torch.full((5,), 0.7, device=device)).sample()
    conditioned = poutine.condition(binomial_model, data={"obs": counts})
    guide = make_guide(conditioned)
    svi = SVI(conditioned, guide, pyro.optim.Adam({"lr": lr}), elbo)
    for _ in range(steps):
        svi.step(num_trials)
    return guide, num_trials


def run_manual_guide(device, parallel, num_svi_steps=2000):
    guide, num_trials = _fit_binomial(device, 400, lambda _: beta_guide, 3.0, num_svi_steps,
                                      Trace_ELBO(num_particles=100, vectorize_particles=True))
    predictive = Predictive(binomial_model, guide=guide, num_samples=10000, parallel=parallel,
                            return_sites=["_RETURN"])
    _close(predictive(num_trials)["_RETURN"].mean(dim=0), torch.full((5,), 280.0), rtol=0.1)


def run_auto_delta(device, parallel):
    guide, num_trials = _fit_binomial(device, 1000, AutoDelta, 1.0, 1000, Trace_ELBO())
    predictive = Predictive(binomial_model, guide=guide, num_samples=10000, parallel=parallel)
    _close(predictive.get_samples(num_trials)["obs"].mean(dim=0), torch.full((5,), 700.0), rtol=0.05)


def run_auto_diag_normal(device, return_trace):
    guide, num_trials = _fit_binomial(device, 1000, AutoDiagonalNormal, 0.1, 1000, Trace_ELBO())
    predictive = Predictive(binomial_model, guide=guide, num_samples=10000, parallel=True)
    if return_trace:
        obs = predictive.get_vectorized_trace(num_trials).nodes["obs"]["value"]
    else:
        obs = predictive.get_samples(num_trials)["obs"]
    _close(obs.mean(dim=0), torch.full((5,), 700.0), rtol=0.05)


def run_one_hot(device):
    _fresh()
    pseudocounts = torch.full((3,), 0.1, device=device)
    true_probs = torch.tensor([0.15, 0.6, 0.25], device=device)
    classes = dist.OneHotCategorical(true_probs).sample((10000,))
    guide = AutoDelta(one_hot_model)
    svi = SVI(one_hot_model, guide, pyro.optim.Adam({"lr": 0.1}), Trace_ELBO())
    for _ in range(1000):
        svi.step(pseudocounts, classes=classes)
    # two stages: draws of the guide's sites, then the model's sites given those
    posterior = Predictive(guide, num_samples=10000).get_samples(pseudocounts)
    obs = Predictive(one_hot_model, posterior).get_samples(pseudocounts)["obs"]
    _close(obs.mean(dim=0), true_probs.unsqueeze(0), rtol=0.1)


def run_shapes(device, parallel):
    """Predictive's shapes equal those of a replay of the guide's vectorised trace."""
    _fresh()
    draws = 10
    zero = torch.zeros((), device=device)

    def program():
        x = pyro.sample("x", dist.Normal(zero, 1.0).expand([2]).to_event(1))
        loc, log_scale = x.unbind(-1)
        with pyro.plate("plate", 5):
            y = pyro.sample("y", dist.Normal(loc, log_scale.exp()))
        return {"x": x, "y": y}

    guide = AutoDiagonalNormal(program)
    batched = pyro.plate("_vectorize", draws, dim=-2)
    expected = poutine.replay(batched(program), poutine.trace(batched(guide)).get_trace())()
    actual = Predictive(program, guide=guide, return_sites=list(expected), num_samples=draws,
                        parallel=parallel)()
    assert {k: v.shape for k, v in actual.items()} == {k: v.shape for k, v in expected.items()}


def run_deterministic(device, with_plate, event_shape):
    """pyro.deterministic sites come back from Predictive with Pyro's shape semantics."""
    _fresh()
    event_dim = len(event_shape)
    zero = torch.zeros((), device=device)

    def program(y=None):
        scope = pyro.plate("plate", 3) if with_plate else contextlib.nullcontext()
        with scope:
            x = pyro.sample("x", dist.Normal(zero, 1.0).expand(event_shape).to_event(event_dim))
            squared = pyro.deterministic("x2", x ** 2, event_dim=event_dim)
        pyro.deterministic("x3", squared)              # outside the plate: the whole batch is the event
        return pyro.sample("obs", dist.Normal(squared, 0.1).to_event(squared.dim()), obs=y)

    y = torch.tensor(4.0, device=device)
    guide = AutoDiagonalNormal(program)
    svi = SVI(program, guide, pyro.optim.Adam({"lr": 0.1}), Trace_ELBO())
    for _ in range(100):
        svi.step(y)
    got = Predictive(program, guide=guide, return_sites=["x2", "x3"], num_samples=1000)()
    plate_shape = (3,) if with_plate else ()
    assert got["x2"].shape == (1000,) + plate_shape + event_shape
    assert got["x3"].shape == (1000,) + ((1,) + plate_shape if with_plate else ()) + event_shape
    for name in ("x2", "x3"):
        _close(got[name].mean(), y, rtol=0.1)


def run_get_mask_optimization(device):
    """Code guarded by ``poutine.get_mask() is not False`` runs under trace / replay, and is skipped both
    under an explicit mask(False) and inside Predictive (which masks the scoring away)."""
    visited = set()

    def at(v):
        return torch.tensor(v, device=device)

    def program():
        x = pyro.sample("x", dist.Normal(at(0.0), 1.0))
        pyro.sample("y", dist.Normal(x, 1.0), obs=at(0.0))
        visited.add("model")
        if poutine.get_mask() is not False:
            visited.add("model, guarded")
            pyro.factor("f", x + 1)

    def guide():
        x = pyro.sample("x", dist.Normal(at(0.0), 1.0))
        visited.add("guide")
        if poutine.get_mask() is not False:
            visited.add("guide, guarded")
            pyro.factor("g", 2 - x)

    def replay_once():
        visited.clear()
        poutine.replay(program, poutine.trace(guide).get_trace())()
        return set(visited)

    assert replay_once() == {"model", "guide", "model, guarded", "guide, guarded"}
    with poutine.mask(mask=False):
        assert replay_once() == {"model", "guide"}
    visited.clear()
    Predictive(program, guide=guide, num_samples=2, parallel=True)()
    assert visited == {"model", "guide"}
