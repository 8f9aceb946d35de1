"""Host logic (handlers, Trace_ELBO assembly, autoguide, SVI, flat Adam) against the golden
vectors of the unmodified reference, with the kernels answered by the numpy oracle
(tests/oracle_backend.py).  The same test bodies run on the GPU through the real HIP kernels in
tests/test_svi_gpu.py."""
import os

import numpy as np
import pytest
import torch

from tests import models

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


@pytest.fixture(autouse=True)
def _cpu_backend(oracle_backend):
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def test_eight_schools_loss_grads_and_trajectory(monkeypatch):
    models.run_eight_schools(load("eight_schools"), torch.device("cpu"), monkeypatch, rtol=1e-9)


@pytest.mark.parametrize("tag,fused", [("f64", False), ("p1", False), ("f64", True), ("p1", True)])
def test_logreg_loss_and_grads(monkeypatch, tag, fused):
    # fused=True exercises the LinearLogits -> glm kernel route (oracle GLM here, f64 arithmetic)
    models.run_logreg(load("logreg_" + tag), torch.device("cpu"), monkeypatch, fused=fused,
                      dtype=torch.float64, rtol=1e-9)


def test_scale_mask_subsample(monkeypatch):
    models.run_scale_mask(load("scale_mask"), torch.device("cpu"), monkeypatch, rtol=1e-9)


def test_score_function_guide(monkeypatch):
    models.run_score_function(load("score_function"), torch.device("cpu"), rtol=1e-9)


@pytest.mark.parametrize("fused", [False, True])
def test_hierarchical_logreg_loss_and_grads(monkeypatch, fused):
    # config 5 at toy size: per-group weights under plate("groups"), ragged groups (one empty)
    models.run_hier(load("hier"), torch.device("cpu"), monkeypatch, fused=fused, rtol=1e-9)


@pytest.mark.parametrize("lazy_on", [True, False])
def test_hierarchical_reference_text_with_unsorted_groups(monkeypatch, lazy_on):
    """SURVEY 8(d) config 5 as the reference writes it ((w[..., g, :] * X).sum(-1) + b, unsorted int64
    ids) against the reference's own loss and gradients: recognised lazily (DeferredGroupDot -> the
    grouped site, the oracle standing in for the kernel) and operator by operator."""
    from pyro_amd import kernels
    from pyro_amd.ops import lazy
    calls = []
    real = kernels.glm_bernoulli_grouped_fwd_bwd
    monkeypatch.setattr(kernels, "glm_bernoulli_grouped_fwd_bwd",
                        lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setitem(lazy.ENABLED, "on", lazy_on)
    if lazy_on:
        monkeypatch.setattr(lazy, "_DTYPES", (torch.float32, torch.float64))   # the oracle is float64
        from pyro_amd import distributions as dist
        monkeypatch.setattr(dist.families._BernoulliLinear, "_allow_f64", True)
    models.run_hier_unsorted(load("hier_unsorted"), torch.device("cpu"), monkeypatch, rtol=1e-9)
    assert len(calls) == (1 if lazy_on else 0)


@pytest.mark.parametrize("tag", ["p1", "p5"])
def test_trace_mean_field_elbo(monkeypatch, tag):
    models.run_meanfield(load("meanfield"), torch.device("cpu"), monkeypatch, tag, rtol=1e-9)


def test_predictive(monkeypatch):
    models.run_predictive(load("predictive"), torch.device("cpu"), monkeypatch, rtol=1e-10)


@pytest.mark.parametrize("which", ["diag", "mvn"])
@pytest.mark.parametrize("tag", ["p1", "p4"])
def test_autocontinuous_guides(monkeypatch, which, tag):
    models.run_autocont(load("autocont"), torch.device("cpu"), monkeypatch, which, tag, rtol=1e-9)


@pytest.mark.parametrize("which", ["diag", "mvn"])
@pytest.mark.parametrize("tag", ["p1", "p4"])
def test_autocontinuous_guides_under_trace_mean_field(monkeypatch, which, tag):
    models.run_autocont(load("autocont"), torch.device("cpu"), monkeypatch, which, tag, rtol=1e-9,
                        mean_field=True)


def test_tracegraph_baselines_match_reference(monkeypatch):
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    torch.set_default_dtype(torch.float64)
    try:
        models.run_tracegraph_baselines(load("tracegraph"), torch.device("cpu"), 1e-9)
    finally:
        torch.set_default_dtype(torch.float32)


def test_tracegraph_provenance_matches_reference(monkeypatch):
    """Downstream costs by data-flow provenance (tracegraph_elbo.py:178-236): the reference's
    gradients on a program where they differ from the plate-only Rao-Blackwellisation."""
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    torch.set_default_dtype(torch.float64)
    try:
        models.run_tracegraph_provenance(load("tracegraph_prov"), torch.device("cpu"), 1e-9)
    finally:
        torch.set_default_dtype(torch.float32)


def test_provenance_tensor_propagation():
    from pyro_amd.ops.provenance import (ProvenanceTensor, detach_provenance, get_provenance,
                                         site_provenance, track_provenance)
    a = track_provenance(torch.tensor([1.0, 2.0]), {"a"})
    b = track_provenance(torch.tensor(3.0), {"b"})
    k = track_provenance(torch.tensor(1), {"k"})
    assert isinstance(a, ProvenanceTensor) and get_provenance(a) == {"a"}
    assert get_provenance(a * 2 + 1) == {"a"} and get_provenance(a + b) == {"a", "b"}
    assert get_provenance(torch.stack([a, a * b])) == {"a", "b"}
    assert get_provenance(torch.tensor([0.1, 0.9])[k]) == {"k"}
    assert get_provenance(torch.ones(2)) == frozenset()
    assert get_provenance(track_provenance(a, {"c"})) == {"a", "c"}
    plain = detach_provenance(a + b)
    assert type(plain) is torch.Tensor and torch.equal(plain, torch.tensor([4.0, 5.0]))
    d = torch.distributions.Independent(torch.distributions.Normal(a, 1.0), 1)
    assert site_provenance({"value": torch.zeros(2), "fn": d, "mask": None, "scale": 1.0}) == {"a"}
    assert site_provenance({"value": b, "fn": torch.distributions.Normal(0.0, 1.0)}) == {"b"}
    # gradients flow through tagged tensors as through plain ones
    p = torch.tensor(2.0, requires_grad=True)
    (track_provenance(p * 1.0, {"z"}) * 3).backward()
    assert p.grad.item() == 3.0


def test_large_plated_site_takes_the_nd_route(monkeypatch):
    """A latent under a plate AND the particle plate scored against parameters that broadcast along
    the middle dim (config 5's w[P, G, D] ~ Normal(mu[P, 1, D], tau[P, 1, D])): the N-D site
    kernels + sum_to reduction against torch autograd of the reference formulation."""
    import numpy as np
    from pyro_amd.distributions import fused
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    monkeypatch.setattr(fused, "_ND_MIN_ELEMS", 64)
    torch.manual_seed(0)
    P, G, D = 3, 7, 5
    w = torch.randn(P, G, D, dtype=torch.float64, requires_grad=True)
    mu = torch.randn(P, 1, D, dtype=torch.float64, requires_grad=True)
    tau = (torch.rand(P, 1, D, dtype=torch.float64) + 0.5).requires_grad_(True)
    mask = torch.rand(G, 1) < 0.8
    calls = []
    import pyro_amd.kernels as k
    real = k.dist_log_prob_sum_nd
    monkeypatch.setattr(k, "dist_log_prob_sum_nd", lambda *a: (calls.append(a[1]), real(*a))[1])
    out = fused.log_prob_sum(0, w, mu, tau, mask=mask, scale=2.5)
    assert calls == [(P, G, D)]
    (out * 0.7).backward()
    got = [t.grad.clone() for t in (w, mu, tau)]
    for t in (w, mu, tau):
        t.grad = None
    ref = (torch.distributions.Normal(mu, tau).log_prob(w) * 2.5 * mask).sum()
    np.testing.assert_allclose(out.item(), ref.item(), rtol=1e-12)
    (ref * 0.7).backward()
    for g, t in zip(got, (w, mu, tau)):
        assert g.shape == t.shape
        torch.testing.assert_close(g, t.grad, rtol=1e-10, atol=1e-12)
    # leading broadcast (the guide's [G, D] parameters under the particle plate): merged to 2 dims
    loc = torch.randn(G, D, dtype=torch.float64, requires_grad=True)
    calls.clear()
    out2 = fused.log_prob_sum(0, w.detach(), loc, torch.ones((), dtype=torch.float64))
    out2.backward()
    ref2 = torch.distributions.Normal(loc.detach().requires_grad_(True), 1.0)
    l2 = ref2.log_prob(w.detach()).sum()
    l2.backward()
    np.testing.assert_allclose(out2.item(), l2.item(), rtol=1e-12)
    torch.testing.assert_close(loc.grad, ref2.loc.grad, rtol=1e-10, atol=1e-12)


def test_flat_adam_steps_only_what_it_is_given_and_resumes_from_a_checkpoint():
    """Partial steps, a parameter created at step 4 (own bias correction), get_state / set_state
    round trip into a fresh optimizer: equal to one torch.optim.Adam per parameter
    (pyro/optim/optim.py:117-155,157-200)."""
    from tests import optim_cases
    optim_cases.run_semantics(torch.device("cpu"))


def test_guide_gradients_reach_autograd_outside_loss_and_grads(monkeypatch):
    """The fused guide draws sink their gradients straight into .grad only inside
    ELBO.loss_and_grads; differentiable_loss + torch.autograd.grad (the torch-optimizer /
    ELBOModule route of pyro/infer/elbo.py:137-142) must see ordinary gradients, also after the
    flat optimizer has made every .grad a permanent view."""
    import pyro_amd as pyro
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    pyro.clear_param_store()
    pyro.set_rng_seed(1)
    dev = torch.device("cpu")
    g = np.random.default_rng(0)
    X = torch.as_tensor(g.standard_normal((50, 4)))
    y = torch.as_tensor((g.uniform(size=50) < 0.5).astype(np.float64))
    guide = AutoNormal(models.logreg_model_fused, init_scale=0.1)
    elbo = Trace_ELBO(num_particles=3, vectorize_particles=True, max_plate_nesting=1)
    svi = SVI(models.logreg_model_fused, guide, pyro.optim.Adam({"lr": 0.01}), elbo)
    svi.step(X, y)                                     # .grad is now a view into the flat buffer
    params = [p for _, p in sorted(pyro.get_param_store()._params.items())]
    assert all(p.grad is not None for p in params)
    pyro.set_rng_seed(5)
    loss = elbo.differentiable_loss(models.logreg_model_fused, guide, X, y)
    grads = torch.autograd.grad(loss, params, allow_unused=False)
    assert all(gr is not None and bool(torch.isfinite(gr).all()) for gr in grads)
    assert all(float(p.grad.abs().sum()) == 0.0 for p in params)     # nothing leaked into .grad
    pyro.set_rng_seed(5)
    ref = elbo.loss_and_grads(models.logreg_model_fused, guide, X, y)  # same draws, sink path
    for p, gr in zip(params, grads):
        np.testing.assert_allclose(p.grad.numpy(), gr.numpy(), rtol=1e-10, atol=1e-12)
    assert abs(ref - float(loss)) < 1e-9 * abs(ref)
    pyro.clear_param_store()
