"""Model/guide definitions and golden-comparison drivers shared by the CPU (oracle-backed) and
GPU (HIP-backed) test suites.  Models are the reference's, written against the drop-in API."""
import numpy as np
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
import pyro_amd.poutine as poutine
from pyro_amd.distributions import constraints
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal


class EpsReplay:
    """Feeds the reference's recorded normal draws to pyro_amd.rng.normal, in order."""

    def __init__(self, eps_list, device, lenient=False):
        self.eps = list(eps_list)
        self.device = device
        self.i = 0
        # lenient: draws whose shape is not the next recorded one are "don't care" draws (prototype
        # runs, which the reference makes through torch.normal, outside the recorded stream)
        self.lenient = lenient

    def __call__(self, shape, dtype, device):
        if self.lenient and (self.i >= len(self.eps)
                             or tuple(self.eps[self.i].shape) != tuple(shape)):
            return torch.zeros(tuple(shape), dtype=dtype, device=self.device)
        e = self.eps[self.i]
        self.i += 1
        assert tuple(e.shape) == tuple(shape), (e.shape, tuple(shape))
        return torch.as_tensor(e, dtype=dtype, device=self.device)


def store_grads():
    return {name: (None if p.grad is None else p.grad.detach().cpu().numpy().copy())
            for name, p in pyro.get_param_store().named_parameters()}


def _eps_of(g, prefix):
    return [g[k] for k in sorted(k for k in g.files if k.startswith(prefix + "/"))]


def assert_grads(got, g, prefix, rtol):
    for name, val in got.items():
        ref = g[prefix + "/" + name]
        sc = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(val, ref, rtol=rtol, atol=rtol * sc, err_msg=name)


# ---- eight schools (examples/eight_schools/svi.py:20-64) ----------------------------------------
J = 8


def es_model(data):
    y = data[:, 0]
    sigma = data[:, 1]
    with pyro.plate("data", J):
        eta = pyro.sample("eta", dist.Normal(torch.zeros(J, device=data.device), torch.ones(J, device=data.device)))
        mu = pyro.sample("mu", dist.Normal(torch.zeros(1, device=data.device), 10 * torch.ones(1, device=data.device)))
        tau = pyro.sample("tau", dist.HalfCauchy(scale=25 * torch.ones(1, device=data.device)))
        theta = mu + tau * eta
        pyro.sample("obs", dist.Normal(theta, sigma), obs=y)


def es_guide_factory(inits):
    def guide(data):
        m_eta = pyro.param("loc_eta", inits["loc_eta"].clone())
        s_eta = pyro.param("scale_eta", inits["scale_eta"].clone(), constraint=constraints.positive)
        m_mu = pyro.param("loc_mu", inits["loc_mu"].clone())
        s_mu = pyro.param("scale_mu", inits["scale_mu"].clone(), constraint=constraints.positive)
        m_lt = pyro.param("loc_logtau", inits["loc_logtau"].clone())
        s_lt = pyro.param("scale_logtau", inits["scale_logtau"].clone(), constraint=constraints.positive)
        dist_tau = dist.LogNormal(m_lt, s_lt)  # == TransformedDistribution(Normal, ExpTransform)
        with pyro.plate("data", J):
            pyro.sample("eta", dist.Normal(m_eta, s_eta))
            pyro.sample("mu", dist.Normal(m_mu, s_mu))
            pyro.sample("tau", dist_tau)
    return guide


def run_eight_schools(g, device, monkeypatch, rtol, optim_factory=None):
    from pyro_amd import rng
    dtype = torch.get_default_dtype()
    y = torch.tensor([28.0, 8, -3, 7, -1, 1, 18, 12], dtype=dtype, device=device)
    sigma = torch.tensor([15.0, 10, 16, 11, 9, 11, 10, 18], dtype=dtype, device=device)
    data = torch.stack([y, sigma], dim=1)
    inits = {k.split("/")[1]: torch.as_tensor(g[k], dtype=dtype, device=device)
             for k in g.files if k.startswith("inits/")}
    pyro.clear_param_store()
    guide = es_guide_factory(inits)
    replay = EpsReplay(_eps_of(g, "eps"), device)
    monkeypatch.setattr(rng, "normal", replay)
    loss0 = Trace_ELBO().loss_and_grads(es_model, guide, data)
    np.testing.assert_allclose(loss0, float(g["loss0"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads0", rtol * 10)
    assert replay.i == int(g["n_eps_first"])
    for p in pyro.get_param_store()._params.values():
        p.grad = None
    optim = optim_factory() if optim_factory else pyro.optim.Adam({"lr": 0.01})
    # (the draws are injected through a replaced pyro_amd.rng.normal -- host code a captured step would
    #  run once: eager steps, said explicitly)
    svi = SVI(es_model, guide, optim, loss=Trace_ELBO(), hip_graph=False)
    losses = [svi.step(data) for _ in range(len(g["losses"]))]
    np.testing.assert_allclose(losses, g["losses"], rtol=rtol * 100)
    for name, value in pyro.get_param_store().items():
        np.testing.assert_allclose(value.detach().cpu().numpy(), g["final/" + name], rtol=rtol * 1000,
                                   atol=rtol * 1000)


# ---- Bayesian logistic regression ------------------------------------------
def logreg_model(X, y):
    """The reference formulation: materialised logits through torch ops."""
    N, D = X.shape
    w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype, device=X.device), 1.0).to_event(1))
    b = pyro.sample("b", dist.Normal(torch.zeros((), dtype=X.dtype, device=X.device), 1.0))
    with pyro.plate("data", N):
        logits = w @ X.t()
        logits = logits.squeeze(-2) if logits.dim() > 1 else logits
        pyro.sample("obs", dist.Bernoulli(logits=logits + b), obs=y)


def logreg_model_fused(X, y):
    """Same model; the logits stay lazy so the observed site runs the fused one-pass GLM kernel."""
    N, D = X.shape
    w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype, device=X.device), 1.0).to_event(1))
    b = pyro.sample("b", dist.Normal(torch.zeros((), dtype=X.dtype, device=X.device), 1.0))
    with pyro.plate("data", N):
        pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)


def run_logreg(g, device, monkeypatch, fused, dtype, rtol):
    from pyro_amd import rng
    from pyro_amd.distributions import families
    if fused and dtype == torch.float64:
        # the HIP GLM kernel is f32-only; in f64 host-logic tests the oracle GLM answers instead
        monkeypatch.setattr(families._BernoulliLinear, "_allow_f64", True, raising=False)
    X = torch.as_tensor(g["X"], dtype=dtype, device=device)
    y = torch.as_tensor(g["y"], dtype=dtype, device=device)
    P = int(g["P"])
    model = logreg_model_fused if fused else logreg_model
    if not fused:
        # "unfused" means the logits are materialised: switch the lazy recognition of
        # w @ X.t() (pyro_amd/ops/lazy.py) off for this run
        from pyro_amd.ops import lazy
        monkeypatch.setitem(lazy.ENABLED, "on", False)
    pyro.clear_param_store()
    guide = AutoNormal(model, init_scale=0.1)
    elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
    guide._setup_prototype(X, y)  # initialisation draws happen before the recorded ones
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps"), device))
    loss = elbo.loss_and_grads(model, guide, X, y)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads", rtol * 10)
    # second evaluation at the moved parameters of the golden run
    store = pyro.get_param_store()
    with torch.no_grad():
        for name in list(store.keys()):
            store[name] = torch.as_tensor(g["params2/" + name], dtype=dtype, device=device)
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps2"), device, lenient=True))
    loss2 = elbo.loss_and_grads(model, guide, X, y)
    np.testing.assert_allclose(loss2, float(g["loss2"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads2", rtol * 10)


# ---- scale / mask / subsample ---------------------------------------------------------------------
def run_scale_mask(g, device, monkeypatch, rtol):
    from pyro_amd import rng
    dtype = torch.get_default_dtype()
    data = torch.as_tensor(g["data"], dtype=dtype, device=device)
    mask = torch.as_tensor(g["mask"], device=device)
    idx = torch.as_tensor(g["idx"], device=device)
    N = data.shape[0]

    def model(data, mask, idx):
        loc = pyro.sample("loc", dist.Normal(torch.tensor(0.0, device=device), 2.0))
        with poutine.scale(scale=0.5):
            s = pyro.sample("s", dist.LogNormal(torch.tensor(0.0, device=device), 0.3))
        with pyro.plate("data", N, subsample=idx) as ind:
            with poutine.mask(mask=mask[ind]):
                pyro.sample("obs", dist.Normal(loc, s), obs=data[ind])

    def guide(data, mask, idx):
        ql = pyro.param("ql", torch.tensor(0.3, device=device))
        qs = pyro.param("qs", torch.tensor(0.2, device=device), constraint=constraints.positive)
        sl = pyro.param("sl", torch.tensor(-0.1, device=device))
        ss = pyro.param("ss", torch.tensor(0.15, device=device), constraint=constraints.positive)
        pyro.sample("loc", dist.Normal(ql, qs))
        with poutine.scale(scale=0.5):
            pyro.sample("s", dist.LogNormal(sl, ss))
        with pyro.plate("data", N, subsample=idx):
            pass

    pyro.clear_param_store()
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps"), device))
    loss = Trace_ELBO().loss_and_grads(model, guide, data, mask, idx)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads", rtol * 10)


def run_tracegraph_provenance(g, device, rtol):
    """TraceGraph_ELBO on a program whose downstream costs differ site by site (independent and
    chained non-reparameterised sites, a local one in a plate, a discrete index): loss, gradients
    and the stored average of a's downstream cost against the reference (tracegraph_prov.npz)."""
    from pyro_amd.infer import TraceGraph_ELBO
    dtype = torch.get_default_dtype()

    def t(v, dt=dtype):
        return torch.as_tensor(np.asarray(v), dtype=dt, device=device)

    x, y, w, table = t(g["x"]), t(g["y"]), t(g["w"]), t(g["table"])
    fixed_vals = {"a": t(g["fixed/a"]), "b": t(g["fixed/b"]), "k": t(g["fixed/k"], torch.int64),
                  "c": t(g["fixed/c"])}

    def model(x, y, w):
        a = pyro.sample("a", dist.Normal(t(0.0), 1.0))
        b = pyro.sample("b", dist.Normal(t(0.0), 1.0))
        k = pyro.sample("k", dist.Categorical(t([0.3, 0.3, 0.4])))
        with pyro.plate("d", 4):
            c = pyro.sample("c", dist.Normal(a, 1.0))
            pyro.sample("x", dist.Normal(c, 0.5), obs=x)
        pyro.sample("y", dist.Normal(b * b, 0.7), obs=y)
        pyro.sample("w", dist.Bernoulli(table[k]), obs=w)

    def guide(x, y, w):
        qa = pyro.param("qa", t(0.2))
        qb = pyro.param("qb", t(-0.3))
        qc = pyro.param("qc", t([0.1, -0.1, 0.3, 0.0]))
        qk = pyro.param("qk", t([0.2, 0.3, 0.5]), constraint=constraints.simplex)
        a = pyro.sample("a", NonreparameterizedNormal(qa, 0.9),
                        infer={"baseline": {"use_decaying_avg_baseline": True, "baseline_beta": 0.7}})
        pyro.sample("b", NonreparameterizedNormal(qb, 1.1))
        pyro.sample("k", dist.Categorical(qk))
        with pyro.plate("d", 4):
            pyro.sample("c", NonreparameterizedNormal(qc + 0.5 * a, 0.8))

    pyro.clear_param_store()
    for k_ in range(2):
        fixed = poutine.trace(poutine.condition(guide, data=fixed_vals)).get_trace(x, y, w)
        for name in fixed_vals:
            fixed.nodes[name]["is_observed"] = False
        for p_ in pyro.get_param_store()._params.values():
            p_.grad = None
        loss = TraceGraph_ELBO().loss_and_grads(model, poutine.replay(guide, trace=fixed), x, y, w)
        np.testing.assert_allclose(loss, float(g["loss%d" % k_]), rtol=rtol)
        grads = {n: v for n, v in store_grads().items() if not n.startswith("__baseline")}
        assert_grads(grads, g, "grads%d" % k_, rtol * 10)
        avg = pyro.get_param_store()["__baseline_avg_downstream_cost_a"].detach().cpu().numpy()
        np.testing.assert_allclose(avg, g["avg%d" % k_], rtol=rtol * 10)


# ---- Gamma-function families (Gamma / Beta latents, Poisson / Binomial likelihoods) ---------------
def run_expfam(g, device, rtol, dtype=None):
    dtype = dtype or torch.get_default_dtype()
    t = lambda k: torch.as_tensor(g[k], dtype=dtype, device=device)   # noqa: E731
    c = lambda x: torch.tensor(float(x), dtype=dtype, device=device)  # noqa: E731
    counts, trials, succ, expo = t("counts"), t("trials"), t("succ"), t("expo")
    N = counts.shape[0]

    def model(counts, trials, succ, expo):
        rate = pyro.sample("rate", dist.Gamma(c(2.0), 0.5))
        p = pyro.sample("p", dist.Beta(c(1.5), 2.5))
        with pyro.plate("data", N):
            pyro.sample("c", dist.Poisson(rate * expo), obs=counts)
            pyro.sample("k", dist.Binomial(trials, probs=p), obs=succ)

    def guide(counts, trials, succ, expo):
        qc = pyro.param("qc", c(4.0), constraint=constraints.positive)
        qr = pyro.param("qr", c(1.3), constraint=constraints.positive)
        qa = pyro.param("qa", c(2.2), constraint=constraints.positive)
        qb = pyro.param("qb", c(5.1), constraint=constraints.positive)
        pyro.sample("rate", dist.Gamma(qc, qr))
        pyro.sample("p", dist.Beta(qa, qb))

    pyro.clear_param_store()
    z = {"rate": c(g["rate"]), "p": c(g["p"])}
    args = (counts, trials, succ, expo)
    fixed = poutine.trace(poutine.condition(guide, data=z)).get_trace(*args)
    for name in z:
        fixed.nodes[name]["is_observed"] = False
    loss = Trace_ELBO().loss_and_grads(model, poutine.replay(guide, trace=fixed), *args)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads", rtol * 10)


# ---- score-function guide site (non-reparameterised) ----------------------------------------------
class NonreparameterizedNormal(dist.Normal):
    has_rsample = False


def run_score_function(g, device, rtol):
    dtype = torch.get_default_dtype()
    data = torch.as_tensor(g["data"], dtype=dtype, device=device)
    zval = torch.as_tensor(g["z"], dtype=dtype, device=device)

    def model2(data):
        with pyro.plate("p", 3):
            z = pyro.sample("z", dist.Normal(torch.zeros(3, device=device), 1.0))
            with pyro.plate("d", 4):
                pyro.sample("x", dist.Normal(z, 0.7), obs=data)

    def guide2(data):
        loc = pyro.param("loc", torch.tensor([0.1, -0.2, 0.4], device=device))
        sc = pyro.param("sc", torch.tensor([0.9, 1.1, 0.8], device=device), constraint=constraints.positive)
        with pyro.plate("p", 3):
            pyro.sample("z", NonreparameterizedNormal(loc, sc))

    pyro.clear_param_store()
    fixed = poutine.trace(poutine.condition(guide2, data={"z": zval})).get_trace(data)
    fixed.nodes["z"]["is_observed"] = False
    loss = Trace_ELBO().loss_and_grads(model2, poutine.replay(guide2, trace=fixed), data)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads", rtol * 10)


# ---- hierarchical logistic regression (config 5) -------------------------------------------------
def run_hier(g, device, monkeypatch, fused, dtype=torch.float64, rtol=1e-9):
    """Loss and gradients of the unmodified reference (tests/golden/hier.npz) through the grouped
    GLM route (fused) or the gather formulation (unfused)."""
    from pyro_amd import examples, kernels, rng
    X = torch.tensor(g["X"], dtype=dtype, device=device)
    y = torch.tensor(g["y"], dtype=dtype, device=device)
    segs = kernels.GroupSegments(g["offsets"], device, target_segments=7)
    model = examples.hier_logreg_model if fused else examples.hier_logreg_model_unfused
    P = int(g["P"])
    pyro.clear_param_store()
    guide = AutoNormal(model, init_scale=0.1)
    guide._setup_prototype(X, y, segs)
    guide(X, y, segs)      # creates the parameters
    store = pyro.get_param_store()
    with torch.no_grad():
        for name in list(store.keys()):
            target = torch.tensor(g["params/" + name], dtype=dtype, device=device)
            unconstrained = store._params[name]
            from torch.distributions import transform_to
            unconstrained.copy_(transform_to(store._constraints[name]).inv(target))
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps"), device))
    elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
    if fused and dtype == torch.float64:
        monkeypatch.setattr(dist.families._BernoulliLinear, "_allow_f64", True)
    loss = elbo.loss_and_grads(model, guide, X, y, segs)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads", rtol * 10)


def run_hier_unsorted(g, device, monkeypatch, dtype=torch.float64, rtol=1e-9):
    """tests/golden/hier_unsorted.npz: the reference's loss and gradients for the gather formulation
    with unsorted group ids, through examples.hier_logreg_model_reference (the same text)."""
    from pyro_amd import examples, rng
    X = torch.tensor(g["X"], dtype=dtype, device=device)
    y = torch.tensor(g["y"], dtype=dtype, device=device)
    ids = torch.tensor(g["g"], dtype=torch.int64, device=device)
    G, P = int(g["G"]), int(g["P"])
    model = examples.hier_logreg_model_reference
    pyro.clear_param_store()
    guide = AutoNormal(model, init_scale=0.1)
    guide._setup_prototype(X, y, ids, G)
    guide(X, y, ids, G)      # creates the parameters
    store = pyro.get_param_store()
    with torch.no_grad():
        for name in list(store.keys()):
            target = torch.tensor(g["params/" + name], dtype=dtype, device=device)
            from torch.distributions import transform_to
            store._params[name].copy_(transform_to(store._constraints[name]).inv(target))
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps"), device))
    elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
    loss = elbo.loss_and_grads(model, guide, X, y, ids, G)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads", rtol * 10)
    pyro.clear_param_store()


# ---- TraceMeanField_ELBO and Predictive (golden: tests/golden/make_golden.py g_meanfield) ---------
def run_meanfield(g, device, monkeypatch, tag, rtol):
    """Loss and gradients of the reference's TraceMeanField_ELBO: analytic KL for the Normal and
    LogNormal pairs, sampled fall-back for the Gamma prior / LogNormal guide pair."""
    from pyro_amd import rng
    from pyro_amd.infer import TraceMeanField_ELBO
    dtype = torch.get_default_dtype()
    data = torch.as_tensor(g["data"], dtype=dtype, device=device)
    N = data.shape[0]

    def t(x):
        return torch.tensor(x, dtype=dtype, device=device)

    def model(data):
        loc = pyro.sample("loc", dist.Normal(torch.zeros(3, dtype=dtype, device=device), 2.0).to_event(1))
        sc = pyro.sample("sc", dist.LogNormal(t(0.0), 0.5))
        gg = pyro.sample("g", dist.Gamma(t(2.0), t(3.0)))
        with pyro.plate("d", N):
            pyro.sample("x", dist.Normal(loc.sum(-1) * gg, sc), obs=data)

    def guide(data):
        ql = pyro.param("ql", t([0.3, -0.2, 0.1]))
        qs = pyro.param("qs", t([0.5, 0.7, 0.9]), constraint=constraints.positive)
        sl = pyro.param("sl", t(-0.1))
        ss = pyro.param("ss", t(0.3), constraint=constraints.positive)
        gl = pyro.param("gl", t(-0.4))
        gs = pyro.param("gs", t(0.2), constraint=constraints.positive)
        pyro.sample("loc", dist.Normal(ql, qs).to_event(1))
        pyro.sample("sc", dist.LogNormal(sl, ss))
        pyro.sample("g", dist.LogNormal(gl, gs))

    P = 1 if tag == "p1" else 5
    pyro.clear_param_store()
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps_" + tag), device))
    elbo = TraceMeanField_ELBO(num_particles=P, vectorize_particles=P > 1, max_plate_nesting=1)
    loss = elbo.loss_and_grads(model, guide, data)
    np.testing.assert_allclose(loss, float(g["loss_" + tag]), rtol=rtol)
    assert_grads(store_grads(), g, "grads_" + tag, rtol * 10)


def run_predictive(g, device, monkeypatch, rtol):
    from pyro_amd import rng
    from pyro_amd.infer import Predictive
    dtype = torch.get_default_dtype()
    N = g["x"].shape[1]

    def model_p(data):
        m = pyro.sample("m", dist.Normal(torch.zeros((), dtype=dtype, device=device), 2.0))
        s = pyro.sample("s", dist.LogNormal(torch.zeros((), dtype=dtype, device=device), 0.5))
        with pyro.plate("d", N):
            pyro.sample("x", dist.Normal(m, s), obs=data)

    post = {k: torch.as_tensor(g["post/" + k], dtype=dtype, device=device) for k in ("m", "s")}
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps"), device, lenient=True))
    pred = Predictive(model_p, posterior_samples=post, parallel=True)(None)
    assert set(pred) == {"x"}
    np.testing.assert_allclose(pred["x"].cpu().numpy(), g["x"], rtol=rtol)
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps2"), device, lenient=True))
    pred = Predictive(model_p, posterior_samples=post, parallel=True, return_sites=["x", "m"])(None)
    np.testing.assert_allclose(pred["x"].cpu().numpy(), g["x2"], rtol=rtol)
    assert tuple(pred["m"].shape) == g["m2"].shape
    np.testing.assert_allclose(pred["m"].cpu().numpy(), g["m2"], rtol=rtol)
    # sequential: the recorded draws are the six [N] vectors of x; the two prototype runs come first
    replay = EpsReplay(_eps_of(g, "eps3"), device, lenient=True)
    skip = {"n": 2}

    def seq_normal(shape, dtype, dev):
        if tuple(shape) == (N,) and skip["n"] > 0:
            skip["n"] -= 1
            return torch.zeros(tuple(shape), dtype=dtype, device=device)
        return replay(shape, dtype, dev)

    monkeypatch.setattr(rng, "normal", seq_normal)
    pred = Predictive(model_p, posterior_samples=post, parallel=False)(None)
    np.testing.assert_allclose(pred["x"].cpu().numpy(), g["x3"], rtol=rtol)


# ---- AutoContinuous guides (golden: tests/golden/make_golden.py g_autocont) -----------------------
def run_autocont(g, device, monkeypatch, which, tag, rtol, mean_field=False):
    """AutoDiagonalNormal / AutoMultivariateNormal: loss and gradients of the reference for a model
    with a vector site, a positive site (exp transform, Jacobian term) and a plated site."""
    from torch.distributions import transform_to
    from pyro_amd import rng
    from pyro_amd.infer.autoguide import (AutoDiagonalNormal, AutoMultivariateNormal,
                                          init_to_feasible)
    dtype = torch.get_default_dtype()
    X = torch.as_tensor(g["X"], dtype=dtype, device=device)
    y = torch.as_tensor(g["y"], dtype=dtype, device=device)
    N, D = X.shape

    def t(x):
        return torch.tensor(x, dtype=dtype, device=device)

    def model(X, y):
        w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=dtype, device=device), 1.0).to_event(1))
        s = pyro.sample("s", dist.LogNormal(t(0.0), 1.0))
        with pyro.plate("g", 2):
            u = pyro.sample("u", dist.Normal(t(0.0), 1.0))
        with pyro.plate("data", N):
            mean = (X * w.unsqueeze(-2)).sum(-1) + u.sum(-1, keepdim=True)
            pyro.sample("obs", dist.Normal(mean, s.unsqueeze(-1)), obs=y)

    cls = AutoDiagonalNormal if which == "diag" else AutoMultivariateNormal
    key = which + "_" + tag
    P = 1 if tag == "p1" else 4
    pyro.clear_param_store()
    guide = cls(model, init_loc_fn=init_to_feasible, init_scale=0.1)
    guide(X, y)       # creates the parameters
    store = pyro.get_param_store()
    with torch.no_grad():
        for name in list(store.keys()):
            target = torch.tensor(g["params_" + key + "/" + name], dtype=dtype, device=device)
            store._params[name].copy_(transform_to(store._constraints[name]).inv(target))
    monkeypatch.setattr(rng, "normal", EpsReplay(_eps_of(g, "eps_" + key), device))
    if mean_field:
        # the guide's Delta sites take kl_divergence(Delta, prior) = -prior.log_prob(value): the
        # reference's loss differs from Trace_ELBO's by the Jacobian term of the positive site
        from pyro_amd.infer import TraceMeanField_ELBO
        elbo = TraceMeanField_ELBO(num_particles=P, vectorize_particles=P > 1, max_plate_nesting=1)
        loss = elbo.loss_and_grads(model, guide, X, y)
        np.testing.assert_allclose(loss, float(g["mf_loss_" + key]), rtol=rtol)
        assert abs(float(g["mf_loss_" + key]) - float(g["loss_" + key])) > 1e-3
        assert_grads(store_grads(), g, "mf_grads_" + key, rtol * 10)
        return
    elbo = Trace_ELBO(num_particles=P, vectorize_particles=P > 1, max_plate_nesting=1)
    loss = elbo.loss_and_grads(model, guide, X, y)
    np.testing.assert_allclose(loss, float(g["loss_" + key]), rtol=rtol)
    assert_grads(store_grads(), g, "grads_" + key, rtol * 10)


# ---- TraceGraph_ELBO baselines against the reference (tests/golden/tracegraph.npz) ----------------
def run_tracegraph_baselines(g, device, rtol):
    """Loss, parameter gradients (incl. the trainable baseline_value) and the stored decaying
    average of two consecutive evaluations with the latent fixed through replay."""
    from pyro_amd.infer import TraceGraph_ELBO
    dtype = torch.get_default_dtype()
    data = torch.as_tensor(g["data"], dtype=dtype, device=device)
    zvals = [torch.as_tensor(g["z%d" % k], dtype=dtype, device=device) for k in range(2)]

    def t(v):
        return torch.tensor(v, dtype=dtype, device=device)

    for tag in ("avg", "value", "both"):
        pyro.clear_param_store()

        def model(data):
            with pyro.plate("p", 3):
                z = pyro.sample("z", dist.Normal(torch.zeros(3, dtype=dtype, device=device), 1.0))
                with pyro.plate("d", 4):
                    pyro.sample("x", dist.Normal(z, 0.7), obs=data)

        def guide(data):
            loc = pyro.param("loc", t([0.1, -0.2, 0.4]))
            sc = pyro.param("sc", t([0.9, 1.1, 0.8]), constraint=constraints.positive)
            b = {}
            if tag in ("value", "both"):
                b["baseline_value"] = pyro.param("bv", t([-3.0, -6.0, -9.0]))
            if tag in ("avg", "both"):
                b.update({"use_decaying_avg_baseline": True, "baseline_beta": 0.8})
            with pyro.plate("p", 3):
                pyro.sample("z", NonreparameterizedNormal(loc, sc), infer={"baseline": b})

        for k, zval in enumerate(zvals):
            fixed = poutine.trace(poutine.condition(guide, data={"z": zval})).get_trace(data)
            fixed.nodes["z"]["is_observed"] = False
            for p_ in pyro.get_param_store()._params.values():
                p_.grad = None
            loss = TraceGraph_ELBO().loss_and_grads(model, poutine.replay(guide, trace=fixed), data)
            np.testing.assert_allclose(loss, float(g["%s/loss%d" % (tag, k)]), rtol=rtol)
            grads = {n: v for n, v in store_grads().items() if not n.startswith("__baseline")}
            assert_grads(grads, g, "%s/grads%d" % (tag, k), rtol * 10)
            key = "%s/avg%d" % (tag, k)
            if key in g.files:
                avg = pyro.get_param_store()["__baseline_avg_downstream_cost_z"].detach().cpu().numpy()
                np.testing.assert_allclose(avg, g[key], rtol=rtol * 10)


# ---- JitTrace_ELBO (pyro/infer/trace_elbo.py:162-257 over pyro/ops/jit.py:48-163) -----------------
def run_logreg_jit(g, device, monkeypatch, fused, dtype, rtol, expect_ops=()):
    """Trace ``differentiable_loss`` of the config-2 model at the golden file's FIRST parameter set with
    its SECOND noise, then move the parameters to the second set and replay the recorded graph: loss
    and gradients must be the reference's second evaluation (params2, eps2) -- a traced graph that had
    frozen a kernel's output as a constant would reproduce the first parameter set's numbers."""
    from pyro_amd import rng
    from pyro_amd.distributions import families
    from pyro_amd.infer import JitTrace_ELBO
    if fused and dtype == torch.float64:
        monkeypatch.setattr(families._BernoulliLinear, "_allow_f64", True, raising=False)
    X = torch.as_tensor(g["X"], dtype=dtype, device=device)
    y = torch.as_tensor(g["y"], dtype=dtype, device=device)
    P = int(g["P"])
    model = logreg_model_fused if fused else logreg_model
    pyro.clear_param_store()
    guide = AutoNormal(model, init_scale=0.1)
    guide._setup_prototype(X, y)
    eps2 = _eps_of(g, "eps2")
    # the first call runs the function once eagerly (parameter discovery) and once under the tracer
    monkeypatch.setattr(rng, "normal", EpsReplay(eps2 + eps2, device, lenient=True))
    elbo = JitTrace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1,
                         ignore_jit_warnings=True)
    loss1 = elbo.loss_and_grads(model, guide, X, y)
    grads1 = store_grads()
    # eager Trace_ELBO at the same parameters and noise
    for p in pyro.get_param_store()._params.values():
        p.grad = None
    monkeypatch.setattr(rng, "normal", EpsReplay(eps2, device, lenient=True))
    ref1 = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1).loss_and_grads(
        model, guide, X, y)
    np.testing.assert_allclose(loss1, ref1, rtol=1e-6 if dtype == torch.float32 else 1e-12)
    for name, val in store_grads().items():
        sc = max(1.0, float(np.abs(val).max()))
        np.testing.assert_allclose(grads1[name], val, rtol=rtol, atol=rtol * sc, err_msg=name)
    (compiled,) = [c for c, _, _ in elbo._jit_cache.values()]
    (traced,) = compiled.compiled.values()
    graph = str(traced.graph)
    for op in expect_ops:
        assert op in graph, (op, sorted({ln.split("= ")[1].split("(")[0] for ln in graph.split("\n")
                                         if "= pyro_amd::" in ln}))
    # move the parameters; the recorded graph (noise eps2 frozen in it) answers at the new values
    store = pyro.get_param_store()
    with torch.no_grad():
        for name in list(store.keys()):
            store[name] = torch.as_tensor(g["params2/" + name], dtype=dtype, device=device)
    for p in store._params.values():
        p.grad = None

    def no_draws(shape, dtype, device):
        raise AssertionError("a replay of the traced loss must not re-run the Python model")
    monkeypatch.setattr(rng, "normal", no_draws)
    loss2 = elbo.loss_and_grads(model, guide, X, y)
    np.testing.assert_allclose(loss2, float(g["loss2"]), rtol=rtol)
    assert_grads(store_grads(), g, "grads2", rtol * 10)
    return graph
