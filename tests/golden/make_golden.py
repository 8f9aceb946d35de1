"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference
(pyro-ppl/pyro 1.9.1 at /root/reference) in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is pure Python, so it cannot travel to the GPU box; the vectors it produces do.
`opt_einsum` (absent third-party dependency) is replaced by the stand-in in oracle/refshim.
Randomness of the reference is pinned by intercepting its normal draws
(torch.distributions.utils._standard_normal) / its pyro.sample calls inside NUTS, so that the
same numbers can be injected into the oracle and into the HIP path.
All reference computations run in float64 on CPU unless a case says float32.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))  # after `import torch` on purpose
sys.path.insert(0, "/root/reference")

import pyro  # noqa: E402
import pyro.distributions as dist  # noqa: E402
import pyro.poutine as poutine  # noqa: E402
from pyro.infer import SVI, Trace_ELBO  # noqa: E402
from pyro.infer.autoguide import AutoNormal  # noqa: E402
from pyro.distributions import constraints  # noqa: E402

assert pyro.__version__ == "1.9.1"


class EpsBank:
    """Replaces torch's standard-normal draw by a pre-generated bank, recording what was used."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.used = []

    def __call__(self, shape, dtype, device):
        e = self.rng.standard_normal(tuple(shape))
        self.used.append(e)
        return torch.as_tensor(e, dtype=dtype, device=device)

    def __enter__(self):
        import torch.distributions.multivariate_normal as tm
        import torch.distributions.normal as tn
        self._old = tn._standard_normal
        tn._standard_normal = self
        tm._standard_normal = self
        return self

    def __exit__(self, *a):
        import torch.distributions.multivariate_normal as tm
        import torch.distributions.normal as tn
        tn._standard_normal = self._old
        tm._standard_normal = self._old


def grads_of_store():
    out = {}
    for name, p in pyro.get_param_store().named_parameters():
        out[name] = None if p.grad is None else p.grad.detach().clone().numpy()
    return out


def save(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[k + "/" + kk] = np.asarray(vv)
        elif isinstance(v, (list, tuple)) and len(v) and isinstance(v[0], np.ndarray):
            for i, vv in enumerate(v):
                flat["%s/%03d" % (k, i)] = vv
        else:
            flat[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, {k: np.asarray(v).shape for k, v in flat.items()})


# ---------------------------------------------------------------------------------------------
# G9: log_prob known answers for every fused family (reference distributions, float64)
# ---------------------------------------------------------------------------------------------
def g_dists():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(0)
    out = {}
    v = rng.standard_normal((4, 7)); a = rng.standard_normal((1, 7)); b = rng.uniform(0.5, 2, (4, 1))
    out["normal"] = dict(v=v, a=a, b=b, lp=dist.Normal(torch.tensor(a), torch.tensor(b)).log_prob(torch.tensor(v)).numpy())
    y = (rng.uniform(size=(4, 7)) < 0.5).astype(float); l = 5 * rng.standard_normal((4, 7))
    out["bernoulli_logits"] = dict(v=y, a=l, lp=dist.Bernoulli(logits=torch.tensor(l)).log_prob(torch.tensor(y)).numpy())
    v = np.abs(rng.standard_cauchy((4, 7))); s = rng.uniform(0.5, 30, (1, 7))
    out["half_cauchy"] = dict(v=v, a=s, lp=dist.HalfCauchy(torch.tensor(s)).log_prob(torch.tensor(v)).numpy())
    v = np.exp(rng.standard_normal((4, 7)))
    out["log_normal"] = dict(v=v, a=a, b=b, lp=dist.LogNormal(torch.tensor(a), torch.tensor(b)).log_prob(torch.tensor(v)).numpy())
    v = rng.exponential(size=(4, 7)); r = rng.uniform(0.5, 2, (4, 1))
    out["exponential"] = dict(v=v, a=r, lp=dist.Exponential(torch.tensor(r)).log_prob(torch.tensor(v)).numpy())
    v = np.abs(rng.standard_normal((4, 7)))
    out["half_normal"] = dict(v=v, a=b, lp=dist.HalfNormal(torch.tensor(b)).log_prob(torch.tensor(v)).numpy())
    flat = {}
    for fam, d in out.items():
        for k, x in d.items():
            flat[fam + "/" + k] = x
    # scale_and_mask (pyro/distributions/util.py:311-328)
    from pyro.distributions.util import scale_and_mask
    x = rng.standard_normal((4, 7)); m = rng.uniform(size=(4, 7)) < 0.6
    flat["scale_and_mask/x"] = x
    flat["scale_and_mask/mask"] = m
    flat["scale_and_mask/out"] = scale_and_mask(torch.tensor(x), 2.5, torch.tensor(m)).numpy()
    # Gamma-function families (drawn AFTER everything above so the older arrays keep their values);
    # autograd gradients of the reference's log_prob are stored with them
    def with_grads(fam, make, v, a, b=None, value_grad=True):
        tv = torch.tensor(v, requires_grad=value_grad)
        ta = torch.tensor(a, requires_grad=True)
        tb = torch.tensor(b, requires_grad=True) if b is not None else None
        lp = make(ta, tb).log_prob(tv)
        ins = [t for t in (tv if value_grad else None, ta, tb) if t is not None]
        gs = list(torch.autograd.grad(lp.sum(), ins))
        flat[fam + "/v"], flat[fam + "/a"], flat[fam + "/lp"] = v, a, lp.detach().numpy()
        if b is not None:
            flat[fam + "/b"] = b
        if value_grad:
            flat[fam + "/dv"] = gs.pop(0).numpy()
        flat[fam + "/da"] = gs.pop(0).numpy()
        if b is not None:
            flat[fam + "/db"] = gs.pop(0).numpy()

    c = rng.uniform(0.2, 12, (1, 7)); r = rng.uniform(0.3, 4, (4, 1))
    with_grads("gamma", lambda a, b: dist.Gamma(a, b), rng.gamma(np.broadcast_to(c, (4, 7))) / r, c, r)
    c1 = rng.uniform(0.3, 9, (4, 7)); c0 = rng.uniform(0.3, 9, (1, 7))
    with_grads("beta", lambda a, b: dist.Beta(a, b), rng.beta(c1, np.broadcast_to(c0, (4, 7))), c1, c0)
    lam = rng.uniform(0.2, 40, (4, 1))
    with_grads("poisson", lambda a, b: dist.Poisson(a), rng.poisson(np.broadcast_to(lam, (4, 7))).astype(float),
               lam, value_grad=False)
    n = rng.integers(1, 60, (1, 7)).astype(float); lg = 3 * rng.standard_normal((4, 7))
    k = rng.binomial(np.broadcast_to(n, (4, 7)).astype(int), 1 / (1 + np.exp(-lg))).astype(float)
    tk, tl = torch.tensor(k), torch.tensor(lg, requires_grad=True)
    lp = dist.Binomial(torch.tensor(n), logits=tl).log_prob(tk)
    flat["binomial_logits/v"], flat["binomial_logits/a"], flat["binomial_logits/b"] = k, lg, n
    flat["binomial_logits/lp"] = lp.detach().numpy()
    flat["binomial_logits/da"] = torch.autograd.grad(lp.sum(), tl)[0].numpy()
    # analytic KL of the Normal / Normal pair (trace_mean_field_elbo.py:121-137 uses kl_divergence)
    lq, sq = rng.standard_normal((4, 7)), rng.uniform(0.2, 2, (4, 7))
    lp_, sp = rng.standard_normal((1, 7)), rng.uniform(0.5, 3, (4, 1))
    ts = [torch.tensor(x, requires_grad=True) for x in (lq, sq, lp_, sp)]
    kl = torch.distributions.kl_divergence(dist.Normal(ts[0], ts[1]), dist.Normal(ts[2], ts[3]))
    gs = torch.autograd.grad(kl.sum(), ts)
    for k, x in zip(("lq", "sq", "lp", "sp"), (lq, sq, lp_, sp)):
        flat["kl_normal/" + k] = x
    flat["kl_normal/kl"] = kl.detach().numpy()
    for k, x in zip(("dlq", "dsq", "dlp", "dsp"), gs):
        flat["kl_normal/" + k] = x.numpy()
    # Dirichlet rows (examples/lda.py:45-60): a shared concentration vector and per-row ones
    for tag, cshape in (("dirichlet_shared", (5,)), ("dirichlet_rows", (6, 5))):
        c = rng.uniform(0.2, 6, cshape)
        xx = rng.dirichlet(np.ones(5), 6)
        tx = torch.tensor(xx, requires_grad=True)
        tc = torch.tensor(c, requires_grad=True)
        lp = dist.Dirichlet(tc).log_prob(tx)
        w = torch.tensor(rng.standard_normal(6))
        gx, gc = torch.autograd.grad((lp * w).sum(), [tx, tc])
        flat[tag + "/x"], flat[tag + "/c"], flat[tag + "/w"] = xx, c, w.numpy()
        flat[tag + "/lp"], flat[tag + "/dx"], flat[tag + "/dc"] = lp.detach().numpy(), gx.numpy(), gc.numpy()
    save("dists", **flat)


# ---------------------------------------------------------------------------------------------
# Gamma-function families inside an ELBO: Gamma / Beta latents (model AND guide sites), Poisson /
# Binomial likelihoods under a plate; the guide is replayed at fixed values so the estimate is a
# deterministic function of the parameters
# ---------------------------------------------------------------------------------------------
def g_expfam():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(23)
    N = 37
    counts = torch.tensor(rng.poisson(3.0, N).astype(float))
    trials = torch.tensor(rng.integers(1, 30, N).astype(float))
    succ = torch.tensor(rng.binomial(trials.numpy().astype(int), 0.3).astype(float))
    expo = torch.tensor(rng.uniform(0.5, 2.0, N))

    def model(counts, trials, succ, expo):
        rate = pyro.sample("rate", dist.Gamma(2.0, 0.5))
        p = pyro.sample("p", dist.Beta(1.5, 2.5))
        with pyro.plate("data", N):
            pyro.sample("c", dist.Poisson(rate * expo), obs=counts)
            pyro.sample("k", dist.Binomial(trials, probs=p), obs=succ)

    def guide(counts, trials, succ, expo):
        qc = pyro.param("qc", torch.tensor(4.0), constraint=constraints.positive)
        qr = pyro.param("qr", torch.tensor(1.3), constraint=constraints.positive)
        qa = pyro.param("qa", torch.tensor(2.2), constraint=constraints.positive)
        qb = pyro.param("qb", torch.tensor(5.1), constraint=constraints.positive)
        pyro.sample("rate", dist.Gamma(qc, qr))
        pyro.sample("p", dist.Beta(qa, qb))

    pyro.clear_param_store()
    z = {"rate": torch.tensor(2.7), "p": torch.tensor(0.31)}
    args = (counts, trials, succ, expo)
    fixed = poutine.trace(poutine.condition(guide, data=z)).get_trace(*args)
    for name in z:
        fixed.nodes[name]["is_observed"] = False
    loss = Trace_ELBO().loss_and_grads(model, poutine.replay(guide, trace=fixed), *args)
    save("expfam", counts=counts.numpy(), trials=trials.numpy(), succ=succ.numpy(), expo=expo.numpy(),
         rate=2.7, p=0.31, loss=loss, grads=grads_of_store())


# ---------------------------------------------------------------------------------------------
# G1: eight schools (examples/eight_schools/svi.py:20-64 verbatim model/guide; guide inits fixed)
# ---------------------------------------------------------------------------------------------
J = 8
Y = torch.tensor([28.0, 8, -3, 7, -1, 1, 18, 12])
SIGMA = torch.tensor([15.0, 10, 16, 11, 9, 11, 10, 18])


def es_model(data):
    y = data[:, 0]
    sigma = data[:, 1]
    with pyro.plate("data", J):
        eta = pyro.sample("eta", dist.Normal(torch.zeros(J), torch.ones(J)))
        mu = pyro.sample("mu", dist.Normal(torch.zeros(1), 10 * torch.ones(1)))
        tau = pyro.sample("tau", dist.HalfCauchy(scale=25 * torch.ones(1)))
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
        dist_tau = dist.TransformedDistribution(dist.Normal(m_lt, s_lt),
                                                transforms=dist.transforms.ExpTransform())
        with pyro.plate("data", J):
            pyro.sample("eta", dist.Normal(m_eta, s_eta))
            pyro.sample("mu", dist.Normal(m_mu, s_mu))
            pyro.sample("tau", dist_tau)
    return guide


def g_eight_schools():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(1)
    inits = {"loc_eta": torch.tensor(rng.standard_normal(J)),
             "scale_eta": torch.tensor(0.1 * rng.uniform(0.2, 1, J)),
             "loc_mu": torch.tensor(rng.standard_normal(1)),
             "scale_mu": torch.tensor(0.1 * rng.uniform(0.2, 1, 1)),
             "loc_logtau": torch.tensor(rng.standard_normal(1)),
             "scale_logtau": torch.tensor(0.1 * rng.uniform(0.2, 1, 1))}
    data = torch.stack([Y.double(), SIGMA.double()], dim=1)
    pyro.clear_param_store()
    guide = es_guide_factory(inits)
    svi = SVI(es_model, guide, pyro.optim.Adam({"lr": 0.01}), loss=Trace_ELBO())
    losses = []
    with EpsBank(11) as bank:
        # step 0: loss and grads at the initial parameters
        loss0 = Trace_ELBO().loss_and_grads(es_model, guide, data)
        g0 = grads_of_store()
        for p in pyro.get_param_store()._params.values():
            p.grad = None
        n_first = len(bank.used)
        for _ in range(30):
            losses.append(svi.step(data))
    final = {k: v.detach().numpy() for k, v in pyro.get_param_store().items()}
    save("eight_schools", inits={k: v.numpy() for k, v in inits.items()}, loss0=loss0, grads0=g0,
         eps=[e for e in bank.used], n_eps_first=n_first, losses=np.array(losses), final=final)


# ---------------------------------------------------------------------------------------------
# G2: Bayesian logistic regression with AutoNormal, vectorised particles
# ---------------------------------------------------------------------------------------------
def logreg_model(X, y):
    N, D = X.shape
    w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype), 1.0).to_event(1))
    b = pyro.sample("b", dist.Normal(torch.zeros((), dtype=X.dtype), 1.0))
    with pyro.plate("data", N):
        logits = w @ X.t()
        logits = logits.squeeze(-2) if logits.dim() > 1 else logits
        pyro.sample("obs", dist.Bernoulli(logits=logits + b), obs=y)


def g_logreg():
    for tag, dtype, N, D, P in [("f64", torch.float64, 1000, 32, 64), ("f32", torch.float32, 4096, 32, 64),
                                ("p1", torch.float64, 257, 5, 1)]:
        torch.set_default_dtype(dtype)
        rng = np.random.default_rng(5)
        X = rng.standard_normal((N, D))
        w_true = rng.standard_normal(D)
        y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ w_true))).astype(float)
        Xt, yt = torch.tensor(X, dtype=dtype), torch.tensor(y, dtype=dtype)
        pyro.clear_param_store()
        guide = AutoNormal(logreg_model, init_scale=0.1)
        elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
        with EpsBank(3) as bank:
            loss = elbo.loss_and_grads(logreg_model, guide, Xt, yt)
        grads = grads_of_store()
        params = {k: v.detach().clone().numpy() for k, v in pyro.get_param_store().items()}
        # second evaluation after moving the parameters (non-trivial locs)
        with torch.no_grad():
            for name, p in pyro.get_param_store().named_parameters():
                p.add_(torch.tensor(np.random.default_rng(9).standard_normal(p.shape) * 0.3, dtype=dtype))
                p.grad = None
        params2 = {k: v.detach().clone().numpy() for k, v in pyro.get_param_store().items()}
        with EpsBank(4) as bank2:
            loss2 = elbo.loss_and_grads(logreg_model, guide, Xt, yt)
        grads2 = grads_of_store()
        save("logreg_" + tag, X=X, y=y, P=P, loss=loss, grads=grads, params=params,
             eps=list(bank.used), loss2=loss2, grads2=grads2, params2=params2, eps2=list(bank2.used))


# ---------------------------------------------------------------------------------------------
# G3: subsampling scale + mask + poutine.scale semantics (trace_struct.py:264-278,
#     subsample_messenger.py:159-174) on a small plated model; G4: score-function guide site
# ---------------------------------------------------------------------------------------------
def g_scale_mask():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(7)
    N, B = 50, 20
    data = torch.tensor(rng.standard_normal(N) + 1.0)
    mask = torch.tensor(rng.uniform(size=N) < 0.7)
    idx = torch.tensor(rng.permutation(N)[:B])

    def model(data, mask, idx):
        loc = pyro.sample("loc", dist.Normal(0.0, 2.0))
        with poutine.scale(scale=0.5):
            s = pyro.sample("s", dist.LogNormal(0.0, 0.3))
        with pyro.plate("data", N, subsample=idx) as ind:
            with poutine.mask(mask=mask[ind]):
                pyro.sample("obs", dist.Normal(loc, s), obs=data[ind])

    def guide(data, mask, idx):
        ql = pyro.param("ql", torch.tensor(0.3))
        qs = pyro.param("qs", torch.tensor(0.2), constraint=constraints.positive)
        sl = pyro.param("sl", torch.tensor(-0.1))
        ss = pyro.param("ss", torch.tensor(0.15), constraint=constraints.positive)
        pyro.sample("loc", dist.Normal(ql, qs))
        with poutine.scale(scale=0.5):
            pyro.sample("s", dist.LogNormal(sl, ss))
        with pyro.plate("data", N, subsample=idx):
            pass

    pyro.clear_param_store()
    with EpsBank(21) as bank:
        loss = Trace_ELBO().loss_and_grads(model, guide, data, mask, idx)
    save("scale_mask", data=data.numpy(), mask=mask.numpy(), idx=idx.numpy(), loss=loss,
         grads=grads_of_store(), eps=list(bank.used))

    # score-function (non-reparameterised) guide site with a plate: test_gradient.py:38-127 style
    from pyro.distributions.testing import fakes

    def model2(data):
        with pyro.plate("p", 3):
            z = pyro.sample("z", dist.Normal(torch.zeros(3), 1.0))
            with pyro.plate("d", 4):
                pyro.sample("x", dist.Normal(z, 0.7), obs=data)

    def guide2(data):
        loc = pyro.param("loc", torch.tensor([0.1, -0.2, 0.4]))
        sc = pyro.param("sc", torch.tensor([0.9, 1.1, 0.8]), constraint=constraints.positive)
        with pyro.plate("p", 3):
            pyro.sample("z", fakes.NonreparameterizedNormal(loc, sc))

    data2 = torch.tensor(rng.standard_normal((4, 3)))
    zval = torch.tensor(rng.standard_normal(3))
    pyro.clear_param_store()
    fixed = poutine.trace(poutine.condition(guide2, data={"z": zval})).get_trace(data2)
    # make the conditioned site a latent again so replay accepts it
    fixed.nodes["z"]["is_observed"] = False
    loss_sf = Trace_ELBO().loss_and_grads(model2, poutine.replay(guide2, trace=fixed), data2)
    save("score_function", data=data2.numpy(), z=zval.numpy(), loss=loss_sf, grads=grads_of_store())


# ---------------------------------------------------------------------------------------------
# G5: integrator known answers (tests/ops/test_integrator.py:40-180 systems) through the
#     reference's velocity_verlet
# ---------------------------------------------------------------------------------------------
def g_integrator():
    torch.set_default_dtype(torch.float64)
    from pyro.ops.integrator import velocity_verlet
    out = {}
    # harmonic oscillator: U = 0.5 q^2, unit mass
    cases = {"harmonic": (lambda q: 0.5 * q["x"] ** 2, 0.0, 1.0, 0.01, 628),
             "quartic": (lambda q: 0.25 * q["x"].pow(4), 0.02, 0.0, 0.1, 810)}
    for name, (pot, q0, p0, eps, n) in cases.items():
        z = {"x": torch.tensor(q0)}
        r = {"x": torch.tensor(p0)}
        zf, rf, gf, pe = velocity_verlet(z, r, lambda q: pot(q).sum(), lambda p: {"x": p["x"]}, eps, n)
        out[name] = dict(q0=q0, p0=p0, eps=eps, n=n, qf=zf["x"].item(), pf=rf["x"].item(), pe=pe.item())
    # 100-dim correlated Gaussian, diagonal inverse mass, a few steps
    rng = np.random.default_rng(2)
    D = 100
    A = rng.standard_normal((D, D))
    Lam = np.linalg.inv(A @ A.T / D + 0.1 * np.eye(D))
    Lam = 0.5 * (Lam + Lam.T)
    Lt = torch.tensor(Lam)
    im = torch.tensor(rng.uniform(0.5, 2.0, D))
    z = {"x": torch.tensor(rng.standard_normal(D))}
    r = {"x": torch.tensor(rng.standard_normal(D))}
    zf, rf, gf, pe = velocity_verlet({k: v.clone() for k, v in z.items()}, {k: v.clone() for k, v in r.items()},
                                     lambda q: 0.5 * q["x"] @ Lt @ q["x"],
                                     lambda p: {"x": im * p["x"]}, 0.05, 7)
    flat = {}
    for k, d in out.items():
        for kk, vv in d.items():
            flat[k + "/" + kk] = vv
    flat.update({"gauss/Lambda": Lam, "gauss/inv_mass": im.numpy(), "gauss/z0": z["x"].numpy(),
                 "gauss/r0": r["x"].numpy(), "gauss/zf": zf["x"].numpy(), "gauss/rf": rf["x"].numpy(),
                 "gauss/gf": gf["x"].numpy(), "gauss/pe": pe.item(), "gauss/eps": 0.05, "gauss/n": 7})
    save("integrator", **flat)


# ---------------------------------------------------------------------------------------------
# G6: reference NUTS transitions on the Gaussian potential with intercepted pyro.sample calls
# ---------------------------------------------------------------------------------------------
def g_nuts():
    torch.set_default_dtype(torch.float64)
    from pyro.infer.mcmc import NUTS
    rng = np.random.default_rng(3)
    flat = {}
    for case, (D, multinomial, step, n_trans) in {"d10_multi": (10, True, 0.35, 6),
                                                  "d100_multi": (100, True, 0.12, 4),
                                                  "d10_slice": (10, False, 0.3, 6),
                                                  "d10_dense": (10, True, 0.5, 6),
                                                  "d12_dense_slice": (12, False, 0.45, 5)}.items():
        A = rng.standard_normal((D, D))
        Lam = np.linalg.inv(A @ A.T / D + 0.1 * np.eye(D))
        Lam = 0.5 * (Lam + Lam.T)
        Lt = torch.tensor(Lam)
        dense = "dense" in case
        if dense:      # full_mass=True: a dense SPD inverse mass matrix (not the exact covariance)
            B = rng.standard_normal((D, D))
            inv_mass = 0.6 * np.linalg.inv(Lam) + 0.2 * (B @ B.T / D) + 0.1 * np.eye(D)
            inv_mass = 0.5 * (inv_mass + inv_mass.T)
        else:
            inv_mass = rng.uniform(0.5, 1.5, D)

        def potential_fn(z):
            return 0.5 * z["x"] @ Lt @ z["x"]

        kernel = NUTS(potential_fn=potential_fn, step_size=step, adapt_step_size=False,
                      adapt_mass_matrix=False, use_multinomial_sampling=multinomial, max_tree_depth=6,
                      full_mass=dense)
        z0 = torch.tensor(rng.standard_normal(D) * 0.5)
        kernel.initial_params = {"x": z0}
        kernel.setup(0)
        kernel.mass_matrix_adapter.inverse_mass_matrix = {("x",): torch.tensor(inv_mass)}
        drawn = {"u": [], "slice": [], "mom": []}
        real_sample = pyro.sample

        def fake_sample(name, fn, *a, **k):
            if name.startswith("r_"):
                e = rng.standard_normal(D)
                drawn["mom"].append(e)
                return torch.tensor(e)
            if name.startswith("slicevar"):
                e = float(rng.exponential())
                drawn["slice"].append(e)
                return torch.tensor(e)
            u = float(rng.uniform())
            drawn["u"].append(u)
            if name.startswith("rand"):
                return torch.tensor(u)
            # Bernoulli sites: direction / is_other_half_tree; sample() == (rand < probs)
            p = fn.probs
            return (torch.tensor(u) < p).to(p.dtype)

        pyro.sample = fake_sample
        try:
            params = {"x": z0}
            zs, marks = [], []
            for t in range(n_trans):
                params = kernel.sample(params)
                zs.append(params["x"].numpy().copy())
                marks.append((len(drawn["u"]), len(drawn["slice"])))
        finally:
            pyro.sample = real_sample
        flat.update({case + "/Lambda": Lam, case + "/inv_mass": inv_mass, case + "/z0": z0.numpy(),
                     case + "/step": step, case + "/multinomial": multinomial,
                     case + "/zs": np.array(zs), case + "/mom": np.array(drawn["mom"]),
                     case + "/u": np.array(drawn["u"]), case + "/slice": np.array(drawn["slice"]),
                     case + "/marks": np.array(marks), case + "/max_tree_depth": 6})
    save("nuts_reference", **flat)


# ---------------------------------------------------------------------------------------------
# G8: warm-up adaptation: schedule, dual averaging, Welford (adaptation.py:65-202,
#     ops/dual_averaging.py:55-78, ops/welford.py:27-52)
# ---------------------------------------------------------------------------------------------
def g_adaptation():
    torch.set_default_dtype(torch.float64)
    from pyro.infer.mcmc.adaptation import WarmupAdapter
    from pyro.ops.dual_averaging import DualAveraging
    from pyro.ops.welford import WelfordCovariance
    flat = {}
    for w in (10, 19, 20, 100, 150, 200, 1000):
        ad = WarmupAdapter(adapt_step_size=True, adapt_mass_matrix=True)
        ad._warmup_steps = w
        sched = ad._build_adaptation_schedule()
        flat["schedule/%d" % w] = np.array([[s.start, s.end] for s in sched])
    rng = np.random.default_rng(4)
    da = DualAveraging(prox_center=math.log(10 * 0.3))
    gs = rng.uniform(-0.5, 0.5, 50)
    xs = []
    for g in gs:
        da.step(float(g))
        xs.append(da.get_state())
    flat["dual/g"] = gs
    flat["dual/x"] = np.array(xs)
    flat["dual/prox_center"] = math.log(10 * 0.3)
    wc = WelfordCovariance(diagonal=True)
    samples = rng.standard_normal((40, 6)) * np.arange(1, 7)
    for s in samples:
        wc.update(torch.tensor(s))
    flat["welford/samples"] = samples
    flat["welford/cov_reg"] = wc.get_covariance(regularize=True).numpy()
    flat["welford/cov"] = wc.get_covariance(regularize=False).numpy()
    # dense estimator + the derived matrices of BlockMassMatrix (adaptation.py:270-392), one dense
    # block over two sites and one diagonal block
    wd = WelfordCovariance(diagonal=False)
    mix = rng.standard_normal((6, 6))
    dsamples = rng.standard_normal((40, 6)) @ mix
    for s_ in dsamples:
        wd.update(torch.tensor(s_))
    flat["welford_dense/samples"] = dsamples
    flat["welford_dense/cov_reg"] = wd.get_covariance(regularize=True).numpy()
    flat["welford_dense/cov"] = wd.get_covariance(regularize=False).numpy()
    from pyro.infer.mcmc.adaptation import BlockMassMatrix
    bm = BlockMassMatrix()
    bm.configure({("a", "b"): (4, 4), ("c",): (2,)}, adapt_mass_matrix=True,
                 options={"dtype": torch.float64})
    zs = rng.standard_normal((30, 6)) @ mix
    for z_ in zs:
        zt = torch.tensor(z_)
        bm.update({"a": zt[:3], "b": zt[3:4], "c": zt[4:]}, None)
    bm.end_adaptation()
    r = torch.tensor(rng.standard_normal(6))
    rd = {"a": r[:3], "b": r[3:4], "c": r[4:]}
    kg = bm.kinetic_grad(rd)
    flat["block/zs"] = zs
    flat["block/r"] = r.numpy()
    flat["block/inv_ab"] = bm.inverse_mass_matrix[("a", "b")].numpy()
    flat["block/inv_c"] = bm.inverse_mass_matrix[("c",)].numpy()
    flat["block/kinetic_grad"] = torch.cat([kg["a"], kg["b"], kg["c"]]).numpy()
    ru = {("a", "b"): r[:4], ("c",): r[4:]}
    sc = bm.scale(ru, rd)
    flat["block/scale"] = torch.cat([sc["a"], sc["b"], sc["c"]]).numpy()
    us = bm.unscale(rd)
    flat["block/unscale"] = torch.cat([us[("a", "b")], us[("c",)]]).numpy()
    save("adaptation", **flat)


# ---------------------------------------------------------------------------------------------
# G10: TraceEnum_ELBO (model-side parallel enumeration, reparameterised guide): LDA of
#      examples/lda.py:42-70 at toy size and a plated Gaussian mixture; values of the continuous
#      latents are pinned through the eps bank / Delta guides, loss and grads from the reference.
#      NOTE: the reference's contraction goes through the opt_einsum stand-in (oracle/refshim),
#      whose pairwise log-space steps are the reference's own pyro/ops/einsum/torch_log.py.
# ---------------------------------------------------------------------------------------------
def g_enum():
    torch.set_default_dtype(torch.float64)
    from pyro.infer import TraceEnum_ELBO
    rng = np.random.default_rng(12)
    T, V, W, D = 3, 7, 4, 5
    data = torch.tensor(rng.integers(0, V, (W, D)))
    tw0 = rng.uniform(0.5, 2.0, T)
    twd0 = rng.uniform(0.6, 2.0, (T, V))
    dt0 = rng.dirichlet(np.ones(T), D)

    def lda_model(data):
        with pyro.plate("topics", T):
            topic_weights = pyro.sample("topic_weights", dist.Gamma(1.0 / T, 1.0))
            topic_words = pyro.sample("topic_words", dist.Dirichlet(torch.ones(V) / V))
        with pyro.plate("documents", D):
            doc_topics = pyro.sample("doc_topics", dist.Dirichlet(topic_weights))
            with pyro.plate("words", W):
                word_topics = pyro.sample("word_topics", dist.Categorical(doc_topics),
                                          infer={"enumerate": "parallel"})
                pyro.sample("doc_words", dist.Categorical(topic_words[word_topics]), obs=data)

    def lda_guide(data):
        a = pyro.param("tw", torch.tensor(tw0), constraint=constraints.positive)
        b = pyro.param("twd", torch.tensor(twd0), constraint=constraints.positive)
        c = pyro.param("dt", torch.tensor(dt0), constraint=constraints.simplex)
        with pyro.plate("topics", T):
            pyro.sample("topic_weights", dist.Delta(a))
            pyro.sample("topic_words", dist.Delta(b / b.sum(-1, keepdim=True), event_dim=1))
        with pyro.plate("documents", D):
            pyro.sample("doc_topics", dist.Delta(c, event_dim=1))

    pyro.clear_param_store()
    loss = TraceEnum_ELBO(max_plate_nesting=2).loss_and_grads(lda_model, lda_guide, data)
    flat = {"lda/data": data.numpy(), "lda/tw0": tw0, "lda/twd0": twd0, "lda/dt0": dt0,
            "lda/loss": loss}
    for k, v in grads_of_store().items():
        flat["lda/grad/" + k] = v

    # plated Gaussian mixture: global component weights/locs, one assignment per datum
    K, N = 3, 11
    x = torch.tensor(rng.standard_normal(N) * 2)
    locs0 = rng.standard_normal(K)
    w0 = rng.dirichlet(np.ones(K))

    def gmm_model(x):
        w = pyro.sample("w", dist.Dirichlet(torch.ones(K)))
        with pyro.plate("comp", K):
            locs = pyro.sample("locs", dist.Normal(0.0, 3.0))
        with pyro.plate("data", N):
            z = pyro.sample("z", dist.Categorical(w), infer={"enumerate": "parallel"})
            pyro.sample("x", dist.Normal(locs[z], 0.7), obs=x)

    def gmm_guide(x):
        ql = pyro.param("ql", torch.tensor(locs0))
        qs = pyro.param("qs", torch.tensor(0.3), constraint=constraints.positive)
        qw = pyro.param("qw", torch.tensor(w0), constraint=constraints.simplex)
        pyro.sample("w", dist.Delta(qw, event_dim=1))
        with pyro.plate("comp", K):
            pyro.sample("locs", dist.Normal(ql, qs))

    pyro.clear_param_store()
    with EpsBank(31) as bank:
        loss2 = TraceEnum_ELBO(max_plate_nesting=1).loss_and_grads(gmm_model, gmm_guide, x)
    flat.update({"gmm/x": x.numpy(), "gmm/locs0": locs0, "gmm/w0": w0, "gmm/loss": loss2})
    for i, e in enumerate(bank.used):
        flat["gmm/eps/%03d" % i] = e
    for k, v in grads_of_store().items():
        flat["gmm/grad/" + k] = v
    # vectorised particles + subsample scale on the data plate
    idx = torch.tensor(rng.permutation(N)[:6])

    def gmm_model_sub(x, idx):
        w = pyro.sample("w", dist.Dirichlet(torch.ones(K)))
        with pyro.plate("comp", K):
            locs = pyro.sample("locs", dist.Normal(0.0, 3.0))
        with pyro.plate("data", N, subsample=idx):
            z = pyro.sample("z", dist.Categorical(w), infer={"enumerate": "parallel"})
            pyro.sample("x", dist.Normal(locs[z], 0.7), obs=x[idx])

    def gmm_guide_sub(x, idx):
        gmm_guide(x)
        with pyro.plate("data", N, subsample=idx):
            pass

    pyro.clear_param_store()
    with EpsBank(32) as bank:
        loss3 = TraceEnum_ELBO(max_plate_nesting=1).loss_and_grads(gmm_model_sub, gmm_guide_sub, x, idx)
    flat.update({"gmmsub/idx": idx.numpy(), "gmmsub/loss": loss3})
    for i, e in enumerate(bank.used):
        flat["gmmsub/eps/%03d" % i] = e
    for k, v in grads_of_store().items():
        flat["gmmsub/grad/" + k] = v
    save("enum", **flat)


# ---------------------------------------------------------------------------------------------
# G11b: the same model with UNSORTED int64 group ids, as SURVEY 8d config 5 writes it
#       (g = randint(0, G, (N,)); logits = (w[..., g, :] * X).sum(-1) + b): what the lazy recognition
#       of ops/lazy.py::DeferredGroupDot must reproduce through the grouped plane-image kernel.
# ---------------------------------------------------------------------------------------------
def g_hier_unsorted():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(22)
    N, D, G, P = 700, 6, 9, 8
    gid = rng.integers(0, G, size=N)
    gid[gid == 4] = 5                       # an empty group
    X = rng.standard_normal((N, D))
    y = (rng.uniform(size=N) < 0.5).astype(float)
    Xt, yt, gt = torch.tensor(X), torch.tensor(y), torch.tensor(gid)

    def model(X, y, g):
        z = torch.zeros(D)
        mu = pyro.sample("mu", dist.Normal(z, 1.0).to_event(1))
        tau = pyro.sample("tau", dist.HalfNormal(torch.ones(D)).to_event(1))
        b = pyro.sample("b", dist.Normal(torch.zeros(()), 1.0))
        with pyro.plate("groups", G):
            w = pyro.sample("w", dist.Normal(mu, tau).to_event(1))
        with pyro.plate("data", N):
            logits = (w[..., g, :] * X).sum(-1) + b
            pyro.sample("obs", dist.Bernoulli(logits=logits), obs=y)

    pyro.clear_param_store()
    guide = AutoNormal(model, init_scale=0.1)
    elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
    guide(Xt, yt, gt)   # prototype + params
    with torch.no_grad():
        for name, p in pyro.get_param_store().named_parameters():
            p.add_(torch.tensor(np.random.default_rng(5).standard_normal(p.shape) * 0.2))
    params = {k: v.detach().clone().numpy() for k, v in pyro.get_param_store().items()}
    with EpsBank(6) as bank:
        loss = elbo.loss_and_grads(model, guide, Xt, yt, gt)
    save("hier_unsorted", X=X, y=y, g=gid, G=G, P=P, loss=loss, grads=grads_of_store(), params=params,
         eps=list(bank.used))


# ---------------------------------------------------------------------------------------------
# G11: hierarchical logistic regression at toy size, AutoNormal, vectorised
#      particles: loss and gradients of the unmodified reference with banked eps.
# ---------------------------------------------------------------------------------------------
def g_hier():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(21)
    N, D, G, P = 300, 6, 5, 8
    sizes = np.array([80, 0, 120, 37, 63])
    off = np.concatenate([[0], np.cumsum(sizes)])
    g_of = torch.tensor(np.repeat(np.arange(G), sizes))
    X = rng.standard_normal((N, D))
    y = (rng.uniform(size=N) < 0.5).astype(float)
    Xt, yt = torch.tensor(X), torch.tensor(y)

    def model(X, y):
        z = torch.zeros(D)
        mu = pyro.sample("mu", dist.Normal(z, 1.0).to_event(1))
        tau = pyro.sample("tau", dist.HalfNormal(torch.ones(D)).to_event(1))
        b = pyro.sample("b", dist.Normal(torch.zeros(()), 1.0))
        with pyro.plate("groups", G):
            w = pyro.sample("w", dist.Normal(mu, tau).to_event(1))
        with pyro.plate("data", N):
            logits = (w[..., g_of, :] * X).sum(-1) + b
            pyro.sample("obs", dist.Bernoulli(logits=logits), obs=y)

    pyro.clear_param_store()
    guide = AutoNormal(model, init_scale=0.1)
    elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
    guide(Xt, yt)   # prototype + params
    with torch.no_grad():
        for name, p in pyro.get_param_store().named_parameters():
            p.add_(torch.tensor(np.random.default_rng(5).standard_normal(p.shape) * 0.2))
    params = {k: v.detach().clone().numpy() for k, v in pyro.get_param_store().items()}
    with EpsBank(6) as bank:
        loss = elbo.loss_and_grads(model, guide, Xt, yt)
    save("hier", X=X, y=y, offsets=off, P=P, loss=loss, grads=grads_of_store(), params=params,
         eps=list(bank.used))


# ---------------------------------------------------------------------------------------------
# G10: TraceMeanField_ELBO (analytic KL where registered, sampled fall-back otherwise) and
#      Predictive (vectorised posterior-predictive draws) through the reference
# ---------------------------------------------------------------------------------------------
def g_meanfield():
    torch.set_default_dtype(torch.float64)
    from pyro.infer import Predictive, TraceMeanField_ELBO
    rng = np.random.default_rng(11)
    N = 7
    data = torch.tensor(rng.standard_normal(N) + 1.0)

    def model(data):
        loc = pyro.sample("loc", dist.Normal(torch.zeros(3), 2.0).to_event(1))      # analytic KL
        sc = pyro.sample("sc", dist.LogNormal(0.0, 0.5))                            # analytic KL
        g = pyro.sample("g", dist.Gamma(2.0, 3.0))                                  # no KL vs LogNormal
        with pyro.plate("d", N):
            pyro.sample("x", dist.Normal(loc.sum(-1) * g, sc), obs=data)

    def guide(data):
        ql = pyro.param("ql", torch.tensor([0.3, -0.2, 0.1]))
        qs = pyro.param("qs", torch.tensor([0.5, 0.7, 0.9]), constraint=constraints.positive)
        sl = pyro.param("sl", torch.tensor(-0.1))
        ss = pyro.param("ss", torch.tensor(0.3), constraint=constraints.positive)
        gl = pyro.param("gl", torch.tensor(-0.4))
        gs = pyro.param("gs", torch.tensor(0.2), constraint=constraints.positive)
        pyro.sample("loc", dist.Normal(ql, qs).to_event(1))
        pyro.sample("sc", dist.LogNormal(sl, ss))
        pyro.sample("g", dist.LogNormal(gl, gs))

    out = {}
    for tag, P in (("p1", 1), ("p5", 5)):
        pyro.clear_param_store()
        elbo = TraceMeanField_ELBO(num_particles=P, vectorize_particles=P > 1, max_plate_nesting=1)
        with EpsBank(31) as bank:
            loss = elbo.loss_and_grads(model, guide, data)
        out[tag] = dict(loss=loss, grads=grads_of_store(), eps=list(bank.used))
    save("meanfield", data=data.numpy(), loss_p1=out["p1"]["loss"], grads_p1=out["p1"]["grads"],
         eps_p1=out["p1"]["eps"], loss_p5=out["p5"]["loss"], grads_p5=out["p5"]["grads"],
         eps_p5=out["p5"]["eps"])

    # Predictive: posterior draws of the latents given, x sampled afresh, vectorised
    def model_p(data):
        m = pyro.sample("m", dist.Normal(0.0, 2.0))
        s = pyro.sample("s", dist.LogNormal(0.0, 0.5))
        with pyro.plate("d", N):
            pyro.sample("x", dist.Normal(m, s), obs=data)

    S = 6
    post = {"m": torch.tensor(rng.standard_normal(S)),
            "s": torch.tensor(np.exp(0.3 * rng.standard_normal(S)))}
    with EpsBank(32) as bank:
        pred = Predictive(model_p, posterior_samples=post, parallel=True)(None)      # x sampled
    with EpsBank(33) as bank2:
        pred_all = Predictive(model_p, posterior_samples=post, parallel=True,
                              return_sites=["x", "m"])(None)
    with EpsBank(34) as bank3:
        pred_seq = Predictive(model_p, posterior_samples=post, parallel=False)(None)
    save("predictive", post=post_np(post), x=pred["x"].numpy(), eps=list(bank.used),
         x2=pred_all["x"].numpy(), m2=pred_all["m"].numpy(), eps2=list(bank2.used),
         x3=pred_seq["x"].numpy(), eps3=list(bank3.used))


# ---------------------------------------------------------------------------------------------
# G11: AutoContinuous guides (AutoDiagonalNormal, AutoMultivariateNormal): loss and gradients
# ---------------------------------------------------------------------------------------------
def g_autocont():
    torch.set_default_dtype(torch.float64)
    from pyro.infer.autoguide import AutoDiagonalNormal, AutoMultivariateNormal, init_to_feasible
    rng = np.random.default_rng(17)
    N, D = 12, 3
    X = torch.tensor(rng.standard_normal((N, D)))
    y = torch.tensor(rng.standard_normal(N))

    def model(X, y):
        w = pyro.sample("w", dist.Normal(torch.zeros(D), 1.0).to_event(1))
        s = pyro.sample("s", dist.LogNormal(0.0, 1.0))                    # positive support
        with pyro.plate("g", 2):
            u = pyro.sample("u", dist.Normal(0.0, 1.0))                   # plated latent
        with pyro.plate("data", N):
            mean = (X * w.unsqueeze(-2)).sum(-1) + u.sum(-1, keepdim=True)
            pyro.sample("obs", dist.Normal(mean, s.unsqueeze(-1)), obs=y)

    out = {}
    for gname, cls in (("diag", AutoDiagonalNormal), ("mvn", AutoMultivariateNormal)):
        for tag, P in (("p1", 1), ("p4", 4)):
            pyro.clear_param_store()
            guide = cls(model, init_loc_fn=init_to_feasible, init_scale=0.1)
            guide(X, y)       # create parameters
            store = pyro.get_param_store()
            g2 = np.random.default_rng(23)
            from torch.distributions import transform_to
            with torch.no_grad():
                for name in sorted(store.keys()):
                    p_ = store[name]
                    if name.endswith("scale_tril"):
                        target = torch.tensor(np.tril(0.1 * g2.standard_normal(tuple(p_.shape)), -1)
                                              + np.eye(p_.shape[0]))
                    elif name.endswith("scale"):
                        target = torch.tensor(np.exp(0.2 * g2.standard_normal(tuple(p_.shape)) - 1.5))
                    else:
                        target = torch.tensor(0.3 * g2.standard_normal(tuple(p_.shape)))
                    # in place on the unconstrained leaf (the guide module holds the same tensor)
                    store._params[name].data.copy_(transform_to(store._constraints[name]).inv(target))
            params = {k: v.detach().clone().numpy() for k, v in store.items()}
            elbo = Trace_ELBO(num_particles=P, vectorize_particles=P > 1, max_plate_nesting=1)
            with EpsBank(41) as bank:
                loss = elbo.loss_and_grads(model, guide, X, y)
            key = gname + "_" + tag
            out["loss_" + key] = loss
            out["grads_" + key] = grads_of_store()
            out["params_" + key] = params
            out["eps_" + key] = list(bank.used)
            # the same guide under TraceMeanField_ELBO: its Delta sites take the analytic route
            # kl_divergence(Delta, prior) = -prior.log_prob(value) (pyro/distributions/kl.py:19-21)
            from pyro.infer import TraceMeanField_ELBO
            for p_ in store._params.values():
                p_.grad = None
            mf = TraceMeanField_ELBO(num_particles=P, vectorize_particles=P > 1, max_plate_nesting=1)
            with EpsBank(41) as bank2:
                mf_loss = mf.loss_and_grads(model, guide, X, y)
            assert len(bank2.used) == len(bank.used) and all(np.array_equal(a, b) for a, b in zip(bank2.used, bank.used))
            out["mf_loss_" + key] = mf_loss
            out["mf_grads_" + key] = grads_of_store()
    save("autocont", X=X.numpy(), y=y.numpy(), **out)


def post_np(d):
    return {k: v.numpy() for k, v in d.items()}


# ---------------------------------------------------------------------------------------------
# G11: hidden Markov models (examples/hmm.py:97-137 model_1, :192-222 model_3 shapes at toy size):
#      pyro.markov recycles the enumeration dims, sequences of ragged lengths are masked, the
#      emissions sit in a nested plate.  Parameters are pyro.params so that the loss is the exact
#      negative log marginal likelihood and its gradient is deterministic.
# ---------------------------------------------------------------------------------------------
def g_hmm():
    torch.set_default_dtype(torch.float64)
    from pyro.infer import TraceEnum_ELBO
    rng = np.random.default_rng(21)
    S, L, K, D = 4, 6, 3, 5
    lengths = np.array([6, 3, 5, 1])
    seqs = (rng.uniform(size=(S, L, D)) < 0.4).astype(np.float64)
    px0 = rng.dirichlet(np.ones(K), K)
    py0 = rng.uniform(0.1, 0.9, (K, D))
    pw0 = rng.dirichlet(np.ones(K), K)
    pyw0 = rng.uniform(0.1, 0.9, (K, K, D))
    sequences, lens = torch.tensor(seqs), torch.tensor(lengths)

    def model_1(sequences, lengths):
        probs_x = pyro.param("probs_x", torch.tensor(px0), constraint=constraints.simplex)
        probs_y = pyro.param("probs_y", torch.tensor(py0), constraint=constraints.unit_interval)
        tones_plate = pyro.plate("tones", D, dim=-1)
        with pyro.plate("sequences", S, dim=-2):
            x = 0
            for t in pyro.markov(range(int(lengths.max()))):
                with poutine.mask(mask=(t < lengths).unsqueeze(-1)):
                    x = pyro.sample("x_{}".format(t), dist.Categorical(probs_x[x]),
                                    infer={"enumerate": "parallel"})
                    with tones_plate:
                        pyro.sample("y_{}".format(t), dist.Bernoulli(probs_y[x.squeeze(-1)]),
                                    obs=sequences[:, t])

    def model_3(sequences, lengths):
        # two hidden chains w, x; the emission depends on both (factorial HMM)
        probs_w = pyro.param("probs_w", torch.tensor(pw0), constraint=constraints.simplex)
        probs_x = pyro.param("probs_x", torch.tensor(px0), constraint=constraints.simplex)
        probs_y = pyro.param("probs_yw", torch.tensor(pyw0), constraint=constraints.unit_interval)
        tones_plate = pyro.plate("tones", D, dim=-1)
        with pyro.plate("sequences", S, dim=-2):
            w, x = 0, 0
            for t in pyro.markov(range(int(lengths.max()))):
                with poutine.mask(mask=(t < lengths).unsqueeze(-1)):
                    w = pyro.sample("w_{}".format(t), dist.Categorical(probs_w[w]),
                                    infer={"enumerate": "parallel"})
                    x = pyro.sample("x_{}".format(t), dist.Categorical(probs_x[x]),
                                    infer={"enumerate": "parallel"})
                    with tones_plate as tones:
                        pyro.sample("y_{}".format(t), dist.Bernoulli(probs_y[w, x, tones]),
                                    obs=sequences[:, t])

    def guide(sequences, lengths):
        pass

    flat = {"sequences": seqs, "lengths": lengths, "probs_x": px0, "probs_y": py0, "probs_w": pw0,
            "probs_yw": pyw0}
    for tag, model in (("m1", model_1), ("m3", model_3)):
        pyro.clear_param_store()
        elbo = TraceEnum_ELBO(max_plate_nesting=2)
        loss = elbo.differentiable_loss(model, guide, sequences, lens)
        names = sorted(pyro.get_param_store().keys())
        params = [pyro.param(n).unconstrained() for n in names]
        grads = torch.autograd.grad(loss, params)
        flat[tag + "/loss"] = loss.item()
        for n, g in zip(names, grads):
            flat[tag + "/grad/" + n] = g.numpy()
    # brute-force check of model_1's loss for the shortest sequences (forward algorithm in numpy)
    def fwd(seq, T):
        alpha = px0[0].copy()
        for t in range(T):
            if t > 0:
                alpha = alpha @ px0
            em = np.prod(np.where(seq[t] > 0, py0, 1 - py0), axis=1)
            alpha = alpha * em
        return np.log(alpha.sum())
    flat["m1/loss_forward_algorithm"] = -sum(fwd(seqs[i], lengths[i]) for i in range(S))
    assert abs(flat["m1/loss_forward_algorithm"] - flat["m1/loss"]) < 1e-9, (flat["m1/loss"],
                                                                           flat["m1/loss_forward_algorithm"])
    save("hmm", **flat)


# ---------------------------------------------------------------------------------------------
# G12: pyro.distributions.DiscreteHMM.log_prob (pyro/distributions/hmm.py:243-369) and its
#      gradients: heterogeneous (per-step, per-batch) and homogeneous (shared) parameters,
#      Normal and multi-dimensional Bernoulli observations, data longer than the parameters' time
#      axis in the homogeneous case.
# ---------------------------------------------------------------------------------------------
def g_discrete_hmm():
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(31)
    flat = {}
    K, T, B, D = 4, 7, 3, 5
    cases = {
        "hetero": dict(init=rng.standard_normal((B, K)), trans=rng.standard_normal((B, T, K, K)),
                       loc=rng.standard_normal((B, T, K)), value=rng.standard_normal((B, T))),
        "homog": dict(init=rng.standard_normal((K,)), trans=rng.standard_normal((1, K, K)),
                      loc=rng.standard_normal((1, K)), value=rng.standard_normal((B, T))),
        "steps": dict(init=rng.standard_normal((K,)), trans=rng.standard_normal((T, K, K)),
                      loc=rng.standard_normal((T, K)), value=rng.standard_normal((B, T))),
    }
    for tag, c in cases.items():
        init, trans, loc = (torch.tensor(c[k], requires_grad=True) for k in ("init", "trans", "loc"))
        d = dist.DiscreteHMM(init, trans, dist.Normal(loc, 0.7))
        lp = d.log_prob(torch.tensor(c["value"]))
        lp.sum().backward()
        flat.update({tag + "/" + k: v for k, v in c.items()})
        flat[tag + "/log_prob"] = lp.detach().numpy()
        flat[tag + "/g_init"], flat[tag + "/g_trans"], flat[tag + "/g_loc"] = (
            init.grad.numpy(), trans.grad.numpy(), loc.grad.numpy())
    # Bernoulli emissions with an event dim (the hmm.py example: tones)
    init = torch.tensor(rng.standard_normal((K,)), requires_grad=True)
    trans = torch.tensor(rng.standard_normal((1, K, K)), requires_grad=True)
    py = torch.tensor(rng.uniform(0.1, 0.9, (K, D)), requires_grad=True)
    value = torch.tensor((rng.uniform(size=(B, T, D)) < 0.4).astype(np.float64))
    d = dist.DiscreteHMM(init, trans, dist.Bernoulli(py).to_event(1))
    lp = d.log_prob(value)
    lp.sum().backward()
    flat.update({"bern/init": init.detach().numpy(), "bern/trans": trans.detach().numpy(),
                 "bern/probs": py.detach().numpy(), "bern/value": value.numpy(),
                 "bern/log_prob": lp.detach().numpy(), "bern/g_init": init.grad.numpy(),
                 "bern/g_trans": trans.grad.numpy(), "bern/g_probs": py.grad.numpy()})
    save("discrete_hmm", **flat)


# ---------------------------------------------------------------------------------------------
# G13: TraceGraph_ELBO with baselines (pyro/infer/tracegraph_elbo.py:28-236): a score-function
#      site in a plate with a decaying-average baseline and a trainable baseline_value, the
#      latent fixed through replay, two consecutive evaluations (the average moves in between).
# ---------------------------------------------------------------------------------------------
def g_tracegraph():
    torch.set_default_dtype(torch.float64)
    from pyro.distributions.testing import fakes
    from pyro.infer import TraceGraph_ELBO
    rng = np.random.default_rng(41)
    data = torch.tensor(rng.standard_normal((4, 3)))
    zvals = [torch.tensor(rng.standard_normal(3)) for _ in range(2)]
    flat = {"data": data.numpy(), "z0": zvals[0].numpy(), "z1": zvals[1].numpy()}
    for tag, opts in (("avg", {"use_decaying_avg_baseline": True, "baseline_beta": 0.8}),
                      ("value", "value"), ("both", "both")):
        pyro.clear_param_store()

        def model(data):
            with pyro.plate("p", 3):
                z = pyro.sample("z", dist.Normal(torch.zeros(3), 1.0))
                with pyro.plate("d", 4):
                    pyro.sample("x", dist.Normal(z, 0.7), obs=data)

        def guide(data):
            loc = pyro.param("loc", torch.tensor([0.1, -0.2, 0.4]))
            sc = pyro.param("sc", torch.tensor([0.9, 1.1, 0.8]), constraint=constraints.positive)
            if opts == "value":
                b = {"baseline_value": pyro.param("bv", torch.tensor([-3.0, -6.0, -9.0]))}
            elif opts == "both":
                b = {"baseline_value": pyro.param("bv", torch.tensor([-3.0, -6.0, -9.0])),
                     "use_decaying_avg_baseline": True, "baseline_beta": 0.8}
            else:
                b = dict(opts)
            with pyro.plate("p", 3):
                pyro.sample("z", fakes.NonreparameterizedNormal(loc, sc), infer={"baseline": b})

        for k, zval in enumerate(zvals):
            fixed = poutine.trace(poutine.condition(guide, data={"z": zval})).get_trace(data)
            fixed.nodes["z"]["is_observed"] = False
            for p_ in pyro.get_param_store()._params.values():
                p_.grad = None
            loss = TraceGraph_ELBO().loss_and_grads(model, poutine.replay(guide, trace=fixed), data)
            flat["%s/loss%d" % (tag, k)] = loss
            for name, g_ in grads_of_store().items():
                if not name.startswith("__baseline"):
                    flat["%s/grads%d/%s" % (tag, k, name)] = g_
            store = pyro.get_param_store()
            if "__baseline_avg_downstream_cost_z" in store:
                flat["%s/avg%d" % (tag, k)] = store["__baseline_avg_downstream_cost_z"].detach().numpy()
    save("tracegraph", **flat)


# ---------------------------------------------------------------------------------------------
# TraceGraph_ELBO with data-flow provenance (tracegraph_elbo.py:178-236): independent and chained
# non-reparameterised sites, a local one inside a plate, a discrete one used as an index; the
# downstream cost of a site holds only the terms that depend on it.  Latents fixed through replay.
# ---------------------------------------------------------------------------------------------
def g_tracegraph_prov():
    torch.set_default_dtype(torch.float64)
    from pyro.distributions.testing import fakes
    from pyro.infer import TraceGraph_ELBO
    rng = np.random.default_rng(43)
    x = torch.tensor(rng.standard_normal(4))
    y = torch.tensor(0.7)
    w = torch.tensor(1.0)
    table = torch.tensor([0.2, 0.5, 0.9])
    fixed_vals = {"a": torch.tensor(0.4), "b": torch.tensor(-0.8), "k": torch.tensor(2),
                  "c": torch.tensor(rng.standard_normal(4))}

    def model(x, y, w):
        a = pyro.sample("a", dist.Normal(0.0, 1.0))
        b = pyro.sample("b", dist.Normal(0.0, 1.0))
        k = pyro.sample("k", dist.Categorical(torch.tensor([0.3, 0.3, 0.4])))
        with pyro.plate("d", 4):
            c = pyro.sample("c", dist.Normal(a, 1.0))
            pyro.sample("x", dist.Normal(c, 0.5), obs=x)
        pyro.sample("y", dist.Normal(b * b, 0.7), obs=y)
        pyro.sample("w", dist.Bernoulli(table[k]), obs=w)

    def guide(x, y, w):
        qa = pyro.param("qa", torch.tensor(0.2))
        qb = pyro.param("qb", torch.tensor(-0.3))
        qc = pyro.param("qc", torch.tensor([0.1, -0.1, 0.3, 0.0]))
        qk = pyro.param("qk", torch.tensor([0.2, 0.3, 0.5]), constraint=constraints.simplex)
        a = pyro.sample("a", fakes.NonreparameterizedNormal(qa, 0.9),
                        infer={"baseline": {"use_decaying_avg_baseline": True, "baseline_beta": 0.7}})
        pyro.sample("b", fakes.NonreparameterizedNormal(qb, 1.1))
        pyro.sample("k", dist.Categorical(qk))
        with pyro.plate("d", 4):
            pyro.sample("c", fakes.NonreparameterizedNormal(qc + 0.5 * a, 0.8))

    pyro.clear_param_store()
    flat = {"x": x.numpy(), "y": y.numpy(), "w": w.numpy(), "table": table.numpy()}
    for name, v in fixed_vals.items():
        flat["fixed/" + name] = v.numpy()
    for k_ in range(2):
        fixed = poutine.trace(poutine.condition(guide, data=fixed_vals)).get_trace(x, y, w)
        for name in fixed_vals:
            fixed.nodes[name]["is_observed"] = False
        for p_ in pyro.get_param_store()._params.values():
            p_.grad = None
        loss = TraceGraph_ELBO().loss_and_grads(model, poutine.replay(guide, trace=fixed), x, y, w)
        flat["loss%d" % k_] = loss
        for name, g_ in grads_of_store().items():
            if not name.startswith("__baseline"):
                flat["grads%d/%s" % (k_, name)] = g_
        flat["avg%d" % k_] = pyro.get_param_store()["__baseline_avg_downstream_cost_a"].detach().numpy()
    save("tracegraph_prov", **flat)


# ---------------------------------------------------------------------------------------------
# ArrowheadMassMatrix (adaptation.py:395-580, ops/arrowhead.py, ops/welford.py:55-101): the mass
# matrix adapted from gradient samples, its inverse, and the three products, from the reference.
# ---------------------------------------------------------------------------------------------
def g_arrowhead():
    torch.set_default_dtype(torch.float64)
    from pyro.infer.mcmc.adaptation import ArrowheadMassMatrix
    from pyro.ops.arrowhead import SymmArrowhead
    rng = np.random.default_rng(47)
    D, h = 6, 4
    flat = {}
    A = rng.standard_normal((D, 2 * D))
    prec = A @ A.T * 0.1
    grads = rng.multivariate_normal(np.zeros(D), prec, size=40)
    for tag in ("adapted", "not_pd"):
        mm = ArrowheadMassMatrix()
        shapes = {("a", "b"): (h, h), ("c",): (D - h,)}
        mm.configure(shapes, adapt_mass_matrix=True, options={"dtype": torch.float64})
        names = ("a", "b", "c")
        if tag == "adapted":
            for g_ in grads:
                t = torch.tensor(g_)
                mm.update(None, {"a": t[:2], "b": t[2:4], "c": t[4:]})
            mm.end_adaptation()
        else:     # a head-tail block too large for positive definiteness: the sqrt halves it
            top = torch.eye(h, D)
            top[:, h:] = 0.9
            mm.mass_matrix = {names: SymmArrowhead(top, torch.ones(D - h))}
        M = mm.mass_matrix[names]
        flat[tag + "/top"], flat[tag + "/bottom_diag"] = M.top.numpy(), M.bottom_diag.numpy()
        flat[tag + "/inverse_mass"] = mm.inverse_mass_matrix[names].numpy()
        r = torch.tensor(rng.standard_normal(D))
        rd = {"a": r[:2], "b": r[2:4], "c": r[4:]}
        v = mm.kinetic_grad(rd)
        flat[tag + "/r"] = r.numpy()
        flat[tag + "/kinetic_grad"] = torch.cat([v[n] for n in names]).numpy()
        u = mm.unscale(rd)[names]
        flat[tag + "/unscale"] = u.numpy()
        sc = mm.scale({names: r}, rd)
        flat[tag + "/scale"] = torch.cat([sc[n] for n in names]).numpy()
    flat["grads"] = grads
    save("arrowhead", **flat)


# ---------------------------------------------------------------------------------------------
# G14: guide-side parallel enumeration with DiCE (traceenum_elbo.py:112-214, infer/util.py:196-326):
#      the "auto" programs of tests/infer/test_enum.py:2121-2208 (everything inside one masked
#      plate, x enumerated in the guide, y in the model) and :1823-1866 (no plate), plus a
#      score-function site upstream of an enumerated one; loss and gradients from the reference.
# ---------------------------------------------------------------------------------------------
def g_guide_enum():
    torch.set_default_dtype(torch.float64)
    from pyro.distributions.testing import fakes
    from pyro.infer import TraceEnum_ELBO, config_enumerate
    flat = {}
    data = torch.tensor([0, 1, 1])
    mask = torch.tensor([True, True, False])

    def params():
        pyro.clear_param_store()
        pyro.param("guide_probs_x", torch.tensor([0.1, 0.9]), constraint=constraints.simplex)
        pyro.param("model_probs_x", torch.tensor([0.4, 0.6]), constraint=constraints.simplex)
        pyro.param("model_probs_y", torch.tensor([[0.75, 0.25], [0.55, 0.45]]), constraint=constraints.simplex)
        pyro.param("model_probs_z", torch.tensor([[0.3, 0.7], [0.2, 0.8]]), constraint=constraints.simplex)

    def record(tag, loss):
        names = sorted(pyro.get_param_store().keys())
        ps = [pyro.param(n).unconstrained() for n in names]
        gs = torch.autograd.grad(loss, ps, allow_unused=True)
        flat[tag + "/loss"] = loss.item()
        for n, g_ in zip(names, gs):
            if g_ is not None:
                flat[tag + "/grad/" + n] = g_.numpy()

    # (1) one masked plate around x (guide-enumerated), y (model-enumerated), z (observed)
    @poutine.scale(scale=10.0)
    def model1(data):
        px, py, pz = (pyro.param("model_probs_" + k) for k in "xyz")
        with pyro.plate("data", 3), poutine.mask(mask=mask):
            x = pyro.sample("x", dist.Categorical(px))
            y = pyro.sample("y", dist.Categorical(py[x]), infer={"enumerate": "parallel"})
            pyro.sample("z", dist.Categorical(pz[y]), obs=data)

    @poutine.scale(scale=10.0)
    @config_enumerate
    def guide1(data):
        pq = pyro.param("guide_probs_x")
        with pyro.plate("data", 3), poutine.mask(mask=mask):
            pyro.sample("x", dist.Categorical(pq))

    params()
    record("plate", TraceEnum_ELBO(max_plate_nesting=1, strict_enumeration_warning=False)
           .differentiable_loss(model1, guide1, data))

    # (2) a score-function Normal site upstream of a guide-enumerated Categorical
    zfix = torch.tensor(0.37)

    def model2():
        s = pyro.sample("s", dist.Normal(0.0, 1.0))
        px = pyro.param("model_probs_x")
        x = pyro.sample("x", dist.Categorical(px))
        pz = pyro.param("model_probs_z")
        pyro.sample("obs", dist.Normal(s + x.to(s.dtype), 0.8), obs=torch.tensor(0.9))
        pyro.sample("z", dist.Categorical(pz[x]), obs=torch.tensor(1))

    @config_enumerate
    def guide2():
        loc = pyro.param("s_loc", torch.tensor(0.2))
        pyro.sample("s", fakes.NonreparameterizedNormal(loc, 0.9))
        pyro.sample("x", dist.Categorical(pyro.param("guide_probs_x")))

    params()
    fixed = poutine.trace(poutine.condition(guide2, data={"s": zfix})).get_trace()
    fixed.nodes["s"]["is_observed"] = False

    def guide2_fixed():
        # replay only the continuous draw; the discrete site is enumerated afresh
        tr = poutine.Trace()
        tr.add_node("s", **fixed.nodes["s"])
        return poutine.replay(guide2, trace=tr)()

    record("score", TraceEnum_ELBO(max_plate_nesting=0, strict_enumeration_warning=False)
           .differentiable_loss(model2, guide2_fixed))
    flat["score/s"] = zfix.numpy()
    save("guide_enum", **flat)


# ---------------------------------------------------------------------------------------------
# G15: discrete latents summed out of the HMC/NUTS potential (pyro/infer/mcmc/util.py:162-286
#      TraceEinsumEvaluator + _PEMaker): the GMM and Bernoulli-latent models of
#      tests/infer/mcmc/test_nuts.py:273-328 and a 5-step Gaussian HMM (:331-392) -- potential
#      energy and its gradient at fixed unconstrained points from the reference's potential_fn.
# ---------------------------------------------------------------------------------------------
def g_mcmc_enum():
    torch.set_default_dtype(torch.float64)
    from pyro.infer.mcmc.util import initialize_model
    flat = {}
    K, N = 3, 40

    def gmm(data):
        phi = pyro.sample("phi", dist.Dirichlet(torch.ones(K)))
        with pyro.plate("num_clusters", K):
            means = pyro.sample("cluster_means", dist.Normal(torch.arange(float(K)), 1.0))
        with pyro.plate("data", data.shape[0]):
            a = pyro.sample("assignments", dist.Categorical(phi))
            pyro.sample("obs", dist.Normal(means[a], 1.0), obs=data)

    @poutine.broadcast
    def bern(data):
        y_prob = pyro.sample("y_prob", dist.Beta(1.0, 1.0))
        with pyro.plate("data", data.shape[0]):
            y = pyro.sample("y", dist.Bernoulli(y_prob))
            z = pyro.sample("z", dist.Bernoulli(0.65 * y + 0.1))
            pyro.sample("obs", dist.Normal(2.0 * z, 1.0), obs=data)

    dim = 3

    def hmm(data):
        initialize = pyro.sample("initialize", dist.Dirichlet(torch.ones(dim)))
        with pyro.plate("states", dim):
            transition = pyro.sample("transition", dist.Dirichlet(torch.ones(dim, dim)))
            loc = pyro.sample("emission_loc", dist.Normal(torch.zeros(dim), torch.ones(dim)))
            scale = pyro.sample("emission_scale", dist.LogNormal(torch.zeros(dim), torch.ones(dim)))
        x = None
        for t, y in pyro.markov(enumerate(data)):
            x = pyro.sample("x_{}".format(t),
                            dist.Categorical(initialize if x is None else transition[x]),
                            infer={"enumerate": "parallel"})
            pyro.sample("y_{}".format(t), dist.Normal(loc[x], scale[x]), obs=y)

    gen = torch.Generator().manual_seed(11)
    datasets = {
        "gmm": torch.tensor([1.0, 5.0, 10.0])[torch.randint(0, 3, (N,), generator=gen)]
        + torch.randn(N, generator=gen),
        "bern": 2.0 * (torch.rand(N, generator=gen) < 0.3).double() + torch.randn(N, generator=gen),
        "hmm": torch.randn(6, generator=gen) + torch.arange(6.0) % 3,
    }
    for tag, model in (("gmm", gmm), ("bern", bern), ("hmm", hmm)):
        data = datasets[tag]
        pyro.set_rng_seed(0)
        init, potential_fn, transforms, _ = initialize_model(model, (data,), max_plate_nesting=1)
        flat[tag + "/data"] = data.numpy()
        for k in range(3):                          # three points per model
            z = {n: (torch.randn(v.shape, generator=gen) * 0.7).requires_grad_(True)
                 for n, v in sorted(init.items())}
            pe = potential_fn(z)
            grads = torch.autograd.grad(pe, list(z.values()))
            flat["%s/pe%d" % (tag, k)] = pe.item()
            for (n, v), g_ in zip(z.items(), grads):
                flat["%s/z%d/%s" % (tag, k, n)] = v.detach().numpy()
                flat["%s/g%d/%s" % (tag, k, n)] = g_.numpy()
    save("mcmc_enum", **flat)


# ---------------------------------------------------------------------------------------------
# G16: potentials of models with constrained supports (pyro/infer/mcmc/util.py:264-286 _PEMaker,
#      :370-482 initialize_model: biject_to(support).inv per site, log|det J| correction) -- the
#      conjugate programs of tests/infer/mcmc/test_nuts.py:184-270,394-462 and test_hmc.py:216-275;
#      potential energy and gradient at fixed unconstrained points from the reference.
# ---------------------------------------------------------------------------------------------
def g_mcmc_potential():
    torch.set_default_dtype(torch.float64)
    from pyro.infer.mcmc.util import initialize_model
    flat = {}
    gen = torch.Generator().manual_seed(5)

    def beta_bernoulli(data):
        p = pyro.sample("p_latent", dist.Beta(torch.tensor([1.1, 1.1]), torch.tensor([1.1, 1.1])))
        pyro.sample("obs", dist.Bernoulli(p), obs=data)

    def gamma_normal(data):
        s = pyro.sample("p_latent", dist.Gamma(torch.tensor([1.0, 1.0]), torch.tensor([1.0, 1.0])))
        pyro.sample("obs", dist.Normal(3.0, s), obs=data)

    def dirichlet_categorical(data):
        p = pyro.sample("p_latent", dist.Dirichlet(torch.tensor([1.0, 1.0, 1.0])))
        pyro.sample("obs", dist.Categorical(p), obs=data)

    def gamma_beta(data):
        a = pyro.sample("alpha", dist.Gamma(concentration=1.0, rate=1.0))
        b = pyro.sample("beta", dist.Gamma(concentration=1.0, rate=1.0))
        pyro.sample("x", dist.Beta(concentration1=a, concentration0=b), obs=data)

    def beta_binomial(data):
        a = pyro.sample("alpha", dist.HalfCauchy(1.0))
        b = pyro.sample("beta", dist.HalfCauchy(1.0))
        with pyro.plate("plate_0", data.shape[-1]):
            probs = pyro.sample("probs", dist.Beta(a, b))
            with pyro.plate("data", data.shape[0]):
                pyro.sample("binomial", dist.Binomial(probs=probs, total_count=1000), obs=data)

    def gamma_poisson(data):
        a = pyro.sample("alpha", dist.HalfCauchy(1.0))
        b = pyro.sample("beta", dist.HalfCauchy(1.0))
        with pyro.plate("plate_0", data.shape[-1]):
            rate = pyro.sample("rate", dist.Gamma(a, b))
            with pyro.plate("data", data.shape[0]):
                pyro.sample("obs", dist.Poisson(rate), obs=data)

    datasets = {
        "beta_bernoulli": (torch.rand(50, 2, generator=gen) < torch.tensor([0.9, 0.1])).double(),
        "gamma_normal": 3.0 + torch.randn(40, 2, generator=gen) * torch.tensor([0.5, 2.0]),
        "dirichlet_categorical": torch.multinomial(torch.tensor([0.1, 0.6, 0.3]), 60, True, generator=gen),
        "gamma_beta": torch.rand(30, generator=gen) * 0.8 + 0.1,
        "beta_binomial": torch.round(1000 * (torch.rand(6, 3, generator=gen) * 0.2
                                             + torch.tensor([0.1, 0.4, 0.7]))),
        "gamma_poisson": torch.poisson(torch.tensor([3.0, 10.0]).expand(8, 2), generator=gen),
    }
    models = dict(beta_bernoulli=beta_bernoulli, gamma_normal=gamma_normal,
                  dirichlet_categorical=dirichlet_categorical, gamma_beta=gamma_beta,
                  beta_binomial=beta_binomial, gamma_poisson=gamma_poisson)
    for tag, model in models.items():
        data = datasets[tag]
        pyro.set_rng_seed(0)
        init, potential_fn, transforms, _ = initialize_model(model, (data,))
        flat[tag + "/data"] = data.numpy()
        for k in range(3):
            z = {n: (torch.randn(v.shape, generator=gen) * 0.7).requires_grad_(True)
                 for n, v in sorted(init.items())}
            pe = potential_fn(z)
            grads = torch.autograd.grad(pe, list(z.values()))
            flat["%s/pe%d" % (tag, k)] = pe.item()
            for (n, v), g_ in zip(z.items(), grads):
                flat["%s/z%d/%s" % (tag, k, n)] = v.detach().numpy()
                flat["%s/g%d/%s" % (tag, k, n)] = g_.numpy()
    save("mcmc_potential", **flat)


# ---------------------------------------------------------------------------------------------
# FLAT models under MCMC: the reference's potential_fn (pyro/infer/mcmc/util.py:264-286, built by
# initialize_model :370-482) and its autograd gradient (pyro/ops/integrator.py:68-94) at fixed unconstrained
# points -- what pa_nuts_tree_run_advance_direct / pa_nuts_direct_potential assemble themselves (GLM kernel +
# the latent sites' six fused families through the identity / exp transforms).
# ---------------------------------------------------------------------------------------------
def g_mcmc_direct_potential():
    torch.set_default_dtype(torch.float64)
    from pyro.infer.mcmc.util import initialize_model
    gen = torch.Generator().manual_seed(11)
    N, D = 300, 8
    X = torch.randn(N, D, generator=gen)
    w_true = torch.randn(D, generator=gen)
    y = (torch.rand(N, generator=gen) < torch.sigmoid(X @ w_true + 0.3)).double()

    def logreg(X, y):
        w = pyro.sample("w", dist.Normal(X.new_zeros(D), 1.0).to_event(1))
        b = pyro.sample("b", dist.Normal(X.new_zeros(()), 1.0))
        with pyro.plate("data", N):
            pyro.sample("obs", dist.Bernoulli(logits=X @ w + b), obs=y)

    def positive_site(X, y):
        pyro.sample("tau", dist.HalfNormal(X.new_ones(())))
        logreg(X, y)

    def all_families(X, y):
        # every family the direct program scores, vector- and scalar-valued, tensor and broadcast parameters
        pyro.sample("a_hc", dist.HalfCauchy(X.new_full((3,), 0.7)).to_event(1))
        pyro.sample("c_ln", dist.LogNormal(X.new_tensor([0.2, -0.4]), X.new_tensor([0.5, 1.5])).to_event(1))
        pyro.sample("d_ex", dist.Exponential(X.new_tensor(1.3)))
        pyro.sample("e_hn", dist.HalfNormal(X.new_tensor([0.8, 2.0])).to_event(1))
        pyro.sample("f_ga", dist.Gamma(X.new_tensor([2.5, 0.6]), X.new_tensor([1.5, 0.9])).to_event(1))
        w = pyro.sample("w", dist.Normal(X.new_full((D,), 0.1), X.new_full((D,), 2.0)).to_event(1))
        with pyro.plate("data", N):
            pyro.sample("obs", dist.Bernoulli(logits=X @ w), obs=y)

    def hier_scale(X, y):
        # a HIERARCHICAL prior: the scale of w is another latent (direct.py encodes it as a parent-valued parameter)
        tau = pyro.sample("tau", dist.HalfNormal(X.new_ones(())))
        w = pyro.sample("w", dist.Normal(X.new_zeros(D), tau.unsqueeze(-1)).to_event(1))   # (broadcast-safe text)
        with pyro.plate("data", N):
            pyro.sample("obs", dist.Bernoulli(logits=X @ w), obs=y)

    def hier_loc_scale(X, y):
        # BASELINE configs[4]'s prior structure at a size the tree kernel's direct form holds: vector parents
        # mu[D], tau[D]; a plated block theta[3, D] and the regression weights w[D] both drawn around them
        mu = pyro.sample("mu", dist.Normal(X.new_zeros(D), 1.0).to_event(1))
        tau = pyro.sample("tau", dist.HalfNormal(X.new_ones(D)).to_event(1))
        b = pyro.sample("b", dist.Normal(X.new_zeros(()), 1.0))
        with pyro.plate("groups", 3):
            pyro.sample("theta", dist.Normal(mu, tau).to_event(1))
        w = pyro.sample("w", dist.Normal(mu, tau).to_event(1))
        with pyro.plate("data", N):
            pyro.sample("obs", dist.Bernoulli(logits=X @ w + b), obs=y)

    flat = {"X": X.numpy(), "y": y.numpy()}
    for tag, model in (("logreg", logreg), ("positive_site", positive_site), ("all_families", all_families),
                       ("hier_scale", hier_scale), ("hier_loc_scale", hier_loc_scale)):
        pyro.set_rng_seed(0)
        init, potential_fn, transforms, _ = initialize_model(model, (X, y))
        flat[tag + "/sites"] = np.array(sorted(init), dtype="U16")
        for k in range(5):
            z = {n: (torch.randn(v.shape, generator=gen) * (0.25 if n == "w" else 0.7)).requires_grad_(True)
                 for n, v in sorted(init.items())}
            pe = potential_fn(z)
            grads = torch.autograd.grad(pe, list(z.values()))
            flat["%s/pe%d" % (tag, k)] = pe.item()
            for (n, v), g_ in zip(z.items(), grads):
                flat["%s/z%d/%s" % (tag, k, n)] = v.detach().numpy()
                flat["%s/g%d/%s" % (tag, k, n)] = g_.numpy()
    save("mcmc_direct_potential", **flat)


# ---------------------------------------------------------------------------------------------
# posterior marginals of model-enumerated sites (traceenum_elbo.py:224-251, 473-493): a global
# Bernoulli and a plated Categorical sharing a Normal likelihood
# ---------------------------------------------------------------------------------------------
def g_marginals():
    from pyro.infer import TraceEnum_ELBO, config_enumerate
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(21)
    data = rng.standard_normal(7) * 1.5
    pi = rng.dirichlet(np.ones(3))
    locs = np.array([-1.0, 0.5, 2.0])
    shift = 0.8
    td, tpi, tl = torch.tensor(data), torch.tensor(pi), torch.tensor(locs)

    @config_enumerate
    def model():
        s = pyro.sample("s", dist.Bernoulli(torch.tensor(0.3)))
        with pyro.plate("data", len(td)):
            z = pyro.sample("z", dist.Categorical(tpi))
            pyro.sample("obs", dist.Normal(tl[z] + shift * s, 1.0), obs=td)

    def guide():
        pass

    pyro.clear_param_store()
    m = TraceEnum_ELBO(max_plate_nesting=1).compute_marginals(model, guide)
    save("marginals", data=data, pi=pi, locs=locs, shift=np.array(shift),
         s_probs=m["s"].probs.detach().numpy(), z_probs=m["z"].probs.detach().numpy())


# ---------------------------------------------------------------------------------------------
# Reparameterised Gamma / Beta / Dirichlet draws of the reference (pyro.distributions.Gamma.rsample ->
# torch._standard_gamma + torch._standard_gamma_grad, torch/distributions/gamma.py:80-88): the implicit
# gradient d sample / d concentration at fixed (concentration, sample), quantile by quantile, and the
# pathwise gradients of a Gamma site's draw w.r.t. both parameters through pyro's own class.
# ---------------------------------------------------------------------------------------------
def g_gamma_grad():
    from scipy import stats
    conc = np.array([0.05, 0.3, 0.9, 1.0, 1.7, 2.5, 7.9, 8.1, 30.0, 200.0, 1500.0])
    quant = np.array([0.005, 0.05, 0.2, 0.5, 0.8, 0.95, 0.995])
    a = np.broadcast_to(conc[:, None], (conc.size, quant.size)).copy()
    x = stats.gamma.ppf(quant[None, :], a)
    g = torch._standard_gamma_grad(torch.tensor(a), torch.tensor(x)).numpy()
    # through the reference's class: value = standard_gamma(c) / r with value fixed by the draw
    c = torch.tensor([0.7, 2.0, 11.0], requires_grad=True)
    r = torch.tensor([0.5, 3.0, 1.5], requires_grad=True)
    torch.manual_seed(3)
    v = dist.Gamma(c, r).rsample()
    (v * torch.tensor([1.0, 2.0, 3.0])).sum().backward()
    # large concentrations (a learned guide concentration): values within a +- 3 sqrt(a), where the
    # series / continued fraction need ~ sqrt(a) terms (round-3 ADVICE: a fixed budget truncated them)
    conc_l = np.array([1e4, 3e4, 1e5, 1e6, 1e7, 3e8])
    zs = np.array([-3.0, -2.0, -1.0, -0.3, 0.0, 0.3, 1.0, 2.0, 3.0])
    a_l = np.broadcast_to(conc_l[:, None], (conc_l.size, zs.size)).copy()
    x_l = a_l + zs[None, :] * np.sqrt(a_l)
    g_l = torch._standard_gamma_grad(torch.tensor(a_l), torch.tensor(x_l)).numpy()
    save("gamma_grad", conc=a, value=x, grad=g, site_conc=c.detach().numpy(), site_rate=r.detach().numpy(),
         site_value=v.detach().numpy(), site_dconc=c.grad.numpy(), site_drate=r.grad.numpy(),
         conc_large=a_l, value_large=x_l, grad_large=g_l)


if __name__ == "__main__":
    which = sys.argv[1:] or ["dists", "eight_schools", "logreg", "scale_mask", "integrator", "nuts",
                             "adaptation", "enum", "hier", "hier_unsorted", "meanfield", "autocont", "hmm", "discrete_hmm", "tracegraph", "guide_enum", "mcmc_enum", "mcmc_potential", "mcmc_direct_potential", "marginals", "expfam", "tracegraph_prov", "arrowhead", "gamma_grad"]
    for w in which:
        globals()["g_" + w]()

