"""Autoguide behaviour found by running tests/infer/test_autoguide.py of the reference against this package
(tests/refsuite): subsampled plates (the parameters cover the FULL plate, each step touches the rows of
its subsample), the error for discrete latent sites, ``guide.call``."""
import pytest
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import poutine
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import (AutoDelta, AutoDiagonalNormal, AutoGuideList,
                                      AutoLowRankMultivariateNormal, AutoMultivariateNormal, AutoNormal,
                                      init_to_feasible, init_to_median)
from pyro_amd.optim import Adam


@pytest.fixture(autouse=True)
def _host(monkeypatch):
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    pyro.clear_param_store()


@pytest.mark.parametrize("auto_class", [AutoDelta, AutoNormal, AutoGuideList])
def test_subsample_guide(auto_class):
    """The model of tutorial/source/easyguide.ipynb (tests/infer/test_autoguide.py:1146-1193): local
    latents in a plate whose subsample is passed in; two epochs over consecutive mini-batches."""
    def model(batch, subsample, full_size):
        steps = len(batch)
        result = [None] * steps
        drift = pyro.sample("drift", dist.LogNormal(-1.0, 0.5))
        data_plate = pyro.plate("data", full_size, subsample=subsample)
        assert data_plate.size == 50
        with data_plate:
            z = 0.0
            for t in range(steps):
                z = pyro.sample("state_{}".format(t), dist.Normal(z, drift))
                result[t] = pyro.sample("obs_{}".format(t), dist.Bernoulli(logits=z), obs=batch[t])
        return torch.stack(result)

    def create_plates(batch, subsample, full_size):
        return pyro.plate("data", full_size, subsample=subsample)

    if auto_class is AutoGuideList:
        guide = AutoGuideList(model, create_plates=create_plates)
        guide.append(AutoDelta(poutine.block(model, expose=["drift"])))
        guide.append(AutoNormal(poutine.block(model, hide=["drift"])))
    else:
        guide = auto_class(model, create_plates=create_plates)
    full_size, batch_size, steps = 50, 20, 4
    pyro.set_rng_seed(123456789)
    data = model([None] * steps, torch.arange(full_size), full_size)
    assert data.shape == (steps, full_size)
    pyro.clear_param_store()
    svi = SVI(model, guide, Adam({"lr": 0.02}), Trace_ELBO())
    for epoch in range(2):
        for beg in range(0, full_size, batch_size):
            end = min(full_size, beg + batch_size)
            svi.step(data[:, beg:end], torch.arange(beg, end), full_size=full_size)
    store = pyro.get_param_store()
    local = [name for name in store.keys() if "state_0" in name]
    assert local and all(store[name].shape == (full_size,) for name in local)
    assert all(store[name].shape == () for name in store.keys() if "drift" in name)


@pytest.mark.parametrize("independent", [True, False], ids=["independent", "dependent"])
@pytest.mark.parametrize("auto_class", [AutoDelta, AutoNormal])
def test_subsample_guide_with_create_plates_and_pyro_subsample(auto_class, independent):
    def model(data):
        size = data.shape[0]
        with pyro.plate("origin", size, dim=-2), pyro.plate("destin", size, dim=-1):
            batch = pyro.subsample(data, event_dim=0)
            assert batch.size(0) == batch.size(1), batch.shape
            pyro.sample("obs", dist.Normal(0.0, 1.0), obs=batch)

    def create_plates(data):
        size = data.shape[0]
        origin = pyro.plate("origin", size, subsample_size=5, dim=-2)
        if independent:
            return origin, pyro.plate("destin", size, subsample_size=5, dim=-1)
        with origin as subsample:
            pass
        return origin, pyro.plate("destin", size, subsample=subsample, dim=-1)

    svi = SVI(model, auto_class(model, create_plates=create_plates), Adam({"lr": 0.01}), Trace_ELBO())
    data = torch.randn(10, 10)
    for _ in range(2):
        svi.step(data)


@pytest.mark.parametrize("auto_class", [AutoDelta, AutoDiagonalNormal, AutoMultivariateNormal, AutoNormal])
@pytest.mark.parametrize("init_loc_fn", [init_to_feasible, init_to_median])
def test_discrete_site_gets_a_helpful_error(auto_class, init_loc_fn):
    def model():
        p = pyro.sample("p", dist.Beta(2.0, 2.0))
        x = pyro.sample("x", dist.Bernoulli(p))
        pyro.sample("obs", dist.Bernoulli(p * x + (1 - p) * (1 - x)), obs=torch.tensor([1.0, 0.0]))

    guide = auto_class(model, init_loc_fn=init_loc_fn)
    with pytest.raises(ValueError, match=".*enumeration.html.*"):
        guide()
    # ... and the advice works: hide the discrete site from the guide
    pyro.clear_param_store()
    auto_class(poutine.block(model, hide=["x"]), init_loc_fn=init_loc_fn)()


def test_call_returns_the_draws_ordered_by_name():
    def model():
        pyro.sample("b", dist.Normal(0.0, 1.0))
        pyro.sample("a", dist.LogNormal(0.0, 1.0))

    guide = AutoNormal(model)
    guide()                                   # the first call also draws the prototype
    pyro.set_rng_seed(0)
    as_dict = guide()
    pyro.set_rng_seed(0)
    as_tuple = guide.call()
    assert torch.equal(as_tuple[0], as_dict["a"]) and torch.equal(as_tuple[1], as_dict["b"])


@pytest.mark.parametrize("auto_class", [AutoLowRankMultivariateNormal, AutoGuideList])
def test_median_of_the_fitted_guide(auto_class):
    """tests/infer/test_autoguide.py:359-417 (test_median): Normal / LogNormal / Beta priors, no data."""
    def model():
        pyro.sample("x", dist.Normal(0.0, 1.0))
        pyro.sample("y", dist.LogNormal(0.0, 1.0))
        pyro.sample("z", dist.Beta(2.0, 2.0))

    if auto_class is AutoGuideList:
        guide = AutoGuideList(model)
        guide.append(AutoNormal(poutine.block(model, expose=["x"])))
        guide.append(AutoLowRankMultivariateNormal(poutine.block(model, hide=["x"])))
    else:
        guide = auto_class(model)
    pyro.set_rng_seed(0)
    svi = SVI(model, guide, Adam({"lr": 0.02, "betas": (0.8, 0.99)}),
              Trace_ELBO(num_particles=200, vectorize_particles=True))
    for _ in range(150):
        svi.step()
    median = guide.median()
    assert abs(float(median["x"])) < 0.15
    assert abs(float(median["y"]) - 1.0) < 0.15
    assert abs(float(median["z"]) - 0.5) < 0.1
    q = guide.quantiles([0.1, 0.5, 0.9])
    assert all(float(q[k][0]) < float(q[k][1]) < float(q[k][2]) for k in ("x", "y", "z"))
    names = set(pyro.get_param_store().keys())
    if auto_class is AutoGuideList:
        assert {"AutoGuideList.0.locs.x", "AutoGuideList.1.loc", "AutoGuideList.1.cov_factor"} <= names
    else:
        assert pyro.param("AutoLowRankMultivariateNormal.cov_factor").shape == (3, 2)


@pytest.mark.parametrize("support", ["positive", "tensor_bound", "event"])
def test_positive_support_site_takes_the_one_kernel_form_with_the_transform_s_numbers(monkeypatch, support):
    """AutoNormal.forward on a site with support (lower, inf): value and Jacobian term come from
    fused.exp_site (one kernel each way) and equal biject_to(support)'s transform + log|det J|
    (reference: guides.py:494-519), gradients included."""
    from pyro_amd import kernels
    from pyro_amd.distributions import fused

    def model():
        with pyro.plate("particles", 5, dim=-2):
            if support == "positive":
                with pyro.plate("g", 4, dim=-1):
                    pyro.sample("tau", dist.HalfNormal(torch.ones((), dtype=torch.float64)))
            elif support == "tensor_bound":
                with pyro.plate("g", 4, dim=-1):
                    pyro.sample("tau", dist.Pareto(torch.full((), 1.5, dtype=torch.float64),
                                                   torch.full((), 3.0, dtype=torch.float64)))
            else:
                pyro.sample("tau", dist.LogNormal(torch.zeros(3, dtype=torch.float64), 1.0).to_event(1))

    def run(fast):
        pyro.clear_param_store()
        pyro.set_rng_seed(0)
        calls = []
        if fast:
            real = fused.exp_site
            monkeypatch.setattr(fused, "exp_site", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        else:
            monkeypatch.setattr(kernels, "on_device", lambda t: False)
        guide = AutoNormal(model)
        tr = poutine.trace(guide).get_trace()
        tr.compute_log_prob()
        site = tr.nodes["tau"]
        u = tr.nodes["tau_unconstrained"]["value"]
        loss = (site["value"] * site["value"].cos()).sum() + 1.7 * site["log_prob"].sum()
        params = [pyro.get_param_store()._params[n] for n in sorted(pyro.get_param_store().keys())]
        grads = torch.autograd.grad(loss, params)
        return site["value"].detach(), site["log_prob"].detach(), [g.detach() for g in grads], calls, u.detach()

    v1, l1, g1, calls, u1 = run(True)
    # (a bound held in a tensor would cost a device read per step: that site keeps the transform)
    assert calls == ([] if support == "tensor_bound" else [1])
    v0, l0, g0, _, u0 = run(False)
    torch.testing.assert_close(u1, u0, rtol=0, atol=0)
    torch.testing.assert_close(v1, v0, rtol=1e-14, atol=0)
    torch.testing.assert_close(l1, l0, rtol=1e-14, atol=1e-14)
    assert l1.shape == l0.shape
    for a, b in zip(g1, g0):
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-14)


def test_exp_site_with_a_host_side_lower_bound():
    from torch.distributions import biject_to, constraints
    from pyro_amd.distributions import fused
    from pyro_amd.infer.autoguide.guides import _exp_lower
    assert _exp_lower(biject_to(constraints.greater_than(1.5))) == 1.5
    assert _exp_lower(biject_to(constraints.positive)) == 0.0
    assert _exp_lower(biject_to(constraints.independent(constraints.nonnegative, 2))) == 0.0
    assert _exp_lower(biject_to(constraints.unit_interval)) is None
    assert _exp_lower(biject_to(constraints.greater_than(torch.tensor(1.5)))) is None
    t = biject_to(constraints.greater_than(1.5))
    # (size-1 event dims: the event rank cannot be recovered from the number of event elements)
    for shape, ed in (((3, 2, 4), 0), ((3, 2, 4), 1), ((3, 2, 4), 2), ((3, 1), 1), ((3, 1, 5), 2), ((4, 1, 1), 2),
                      ((1,), 1), ((2, 5, 1), 1)):
        u = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
        value, ld = fused.exp_site(u, ed, 1.5)
        assert ld.shape == u.shape[:u.dim() - ed]
        ref_v = t(u)
        ref_ld = t.inv.log_abs_det_jacobian(ref_v, u)
        ref_ld = ref_ld.sum(tuple(range(-ed, 0))) if ed else ref_ld
        torch.testing.assert_close(value, ref_v, rtol=1e-14, atol=0)
        torch.testing.assert_close(ld, ref_ld, rtol=1e-14, atol=1e-14)
        w = torch.randn_like(ld)
        g, = torch.autograd.grad((value.sin()).sum() + (w * ld).sum(), u, retain_graph=True)
        rg, = torch.autograd.grad((ref_v.sin()).sum() + (w * ref_ld).sum(), u)
        torch.testing.assert_close(g, rg, rtol=1e-12, atol=1e-14)
        # one of the two outputs unused: its gradient arrives as None / zeros
        g, = torch.autograd.grad(value.sum(), u)
        torch.testing.assert_close(g, ref_v.detach() - 1.5, rtol=1e-14, atol=0)


def test_large_guide_site_scored_in_closed_form_matches_autograd_through_log_prob(monkeypatch):
    """A mean-field site too large for the many-small-sites launch, scored at the guide's own draw:
    Normal.fused_score_term hands over sum log q(z) with its TOTAL derivative (-1/scale per element, 0 for
    loc) behind a single autograd node.  Loss and parameter gradients equal the ones autograd derives through
    Normal.log_prob's three paths (reference: trace_elbo.py:142-160) to rounding."""
    from pyro_amd import _lib
    from pyro_amd.distributions.families import Normal

    G, D, P = 50, 8, 16
    g = torch.Generator().manual_seed(0)
    X = torch.randn(300, D, dtype=torch.float64, generator=g)
    y = (torch.rand(300, dtype=torch.float64, generator=g) < 0.5).double()
    grp = torch.randint(0, G, (300,), generator=g)

    def model():
        tau = pyro.sample("tau", dist.HalfNormal(torch.ones(D, dtype=torch.float64)).to_event(1))
        with pyro.plate("groups", G):
            w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=torch.float64), tau).to_event(1))
        with pyro.plate("data", 300):
            pyro.sample("obs", dist.Bernoulli(logits=(w[..., grp, :] * X).sum(-1)), obs=y)

    monkeypatch.setattr(_lib, "MULTI_MAX_ELEMS", 1000)       # w's draw: 16 x 400 elements, its scale: 400
    real = Normal.fused_score_term

    def run(closed_form):
        calls = []

        def spy(self, value, scale=1.0, mask=None):
            out = real(self, value, scale, mask) if closed_form else None
            if out is not None:
                calls.append(tuple(value.shape))
            return out

        monkeypatch.setattr(Normal, "fused_score_term", spy)
        pyro.clear_param_store()
        pyro.set_rng_seed(1)
        guide = AutoNormal(model, init_scale=0.3)
        elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
        loss = elbo.loss_and_grads(model, guide)
        store = pyro.get_param_store()
        return loss, {n: store._params[n].grad.clone() for n in sorted(store.keys())}, calls

    loss1, g1, calls = run(True)
    assert calls == [(P, G, D)]
    loss0, g0, _ = run(False)
    assert loss1 == pytest.approx(loss0, rel=1e-12)
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], rtol=1e-9, atol=1e-11)
