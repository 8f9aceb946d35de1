"""The reference's known-answer tests for the posterior of enumerated model sites
(tests/infer/test_enum.py:3725-4010: compute_marginals, sample_posterior) restated against the
drop-in API; shared by the CPU host-logic tests and the MI355X tests.  Quantitative answers come
from brute-force enumeration in float64 (and from the reference itself, tests/golden/marginals.npz)."""
import itertools

import numpy as np
import pytest
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd.infer import TraceEnum_ELBO, config_enumerate


def _t(v, device, dtype=None):
    return torch.tensor(v, device=device, dtype=dtype)


def run_marginals_single(device, which, prior):
    """test_compute_marginals_single: one global enumerated site over a plate of observations;
    the marginal equals Bayes' rule and an enumerating guide built from it has zero gradient."""
    Dist = {"bernoulli": dist.Bernoulli, "categorical": dist.Categorical,
            "onehot": dist.OneHotCategorical}[which]
    prior_t = _t(prior, device, torch.get_default_dtype())
    data = _t([0.0, 0.1, 0.2, 0.9, 1.0, 1.1], device, torch.get_default_dtype())
    locs = _t([-1.0, 0.0, 1.0, 2.0], device, torch.get_default_dtype())

    @config_enumerate
    def model():
        x = pyro.sample("x", Dist(prior_t))
        if Dist is dist.Bernoulli:
            x = x.long()
        elif Dist is dist.OneHotCategorical:
            x = x.max(-1)[1]
        with pyro.plate("data", len(data)):
            pyro.sample("obs", dist.Normal(locs[x], 1.0), obs=data)

    def empty_guide():
        pass

    pyro.clear_param_store()
    elbo = TraceEnum_ELBO(max_plate_nesting=1)
    marginals = elbo.compute_marginals(model, empty_guide)
    assert list(marginals) == ["x"]
    assert type(marginals["x"]).__name__ == Dist.__name__
    probs = marginals["x"].probs
    assert probs.shape == prior_t.shape
    # Bayes' rule by hand (float64)
    p = np.array(prior if which != "bernoulli" else [1 - prior, prior], dtype=np.float64)
    d = data.cpu().double().numpy()
    lo = locs.cpu().double().numpy()[:len(p)]
    like = np.exp(-0.5 * (d[None, :] - lo[:, None]) ** 2).prod(1)
    post = p * like / (p * like).sum()
    want = post if which != "bernoulli" else post[1]
    tol = 1e-9 if probs.dtype == torch.float64 else 2e-5
    np.testing.assert_allclose(probs.cpu().double().numpy(), want, rtol=tol, atol=tol)
    # the marginal is the optimum of an enumerating guide: zero gradient there
    pyro.param("probs", probs.detach().clone())

    @config_enumerate
    def exact_guide():
        pyro.sample("x", Dist(pyro.param("probs")))

    loss = elbo.differentiable_loss(model, exact_guide)
    leaf = pyro.get_param_store()._params["probs"]
    (g,) = torch.autograd.grad(loss, [leaf])
    if which != "bernoulli":
        g = g - g.mean()            # on the simplex only differences of the gradient matter
    assert float(g.abs().max()) < (1e-8 if probs.dtype == torch.float64 else 2e-4), g
    pyro.clear_param_store()


def run_marginals_restrictions(device, ok, enumerate_guide, num_particles, vectorize_particles, what):
    """test_compute_marginals_restrictions / test_backwardsample_posterior_restrictions."""
    f = lambda v: _t(v, device, torch.get_default_dtype())  # noqa: E731

    @config_enumerate
    def model():
        w = pyro.sample("w", dist.Bernoulli(f(0.1)))
        x = pyro.sample("x", dist.Bernoulli(f(0.2)))
        y = pyro.sample("y", dist.Bernoulli(f(0.3)))
        z = pyro.sample("z", dist.Bernoulli(f(0.4)))
        pyro.sample("obs", dist.Normal(f(0.0), f(1.0)), obs=w + x + y + z)
        return w, x, y, z

    @config_enumerate(default=enumerate_guide)
    def guide():
        pyro.sample("w", dist.Bernoulli(f(0.4)))
        pyro.sample("y", dist.Bernoulli(f(0.7)))

    elbo = TraceEnum_ELBO(max_plate_nesting=0, num_particles=num_particles,
                          vectorize_particles=vectorize_particles)
    assert np.isfinite(elbo.loss(model, guide))
    call = elbo.compute_marginals if what == "marginals" else elbo.sample_posterior
    if ok:
        out = call(model, guide)
        if what == "marginals":
            assert set(out.keys()) == {"x", "z"}
        else:
            assert all(v.shape == () for v in out)
    else:
        with pytest.raises(NotImplementedError, match="compute_marginals" if what == "marginals"
                           else "sample_posterior"):
            call(model, guide)


def _hmm_brute_force(size):
    tp = np.array([[0.75, 0.25], [0.25, 0.75]])
    ep = np.array([[0.75, 0.25], [0.25, 0.75]])
    post = np.zeros((size, 2))
    for xs in itertools.product(range(2), repeat=size):
        p, prev = 1.0, 0
        for i, x in enumerate(xs):
            p *= tp[prev, x] * ep[x, 0]
            prev = x
        p *= tp[prev, 1]                      # x_size observed == 1
        for i, x in enumerate(xs):
            post[i, x] += p
    return post / post.sum(1, keepdims=True)


def run_marginals_hmm(device, size):
    """test_compute_marginals_hmm (pyro.markov chain, last state observed) + exact values."""
    tp = _t([[0.75, 0.25], [0.25, 0.75]], device, torch.get_default_dtype())
    ep = _t([[0.75, 0.25], [0.25, 0.75]], device, torch.get_default_dtype())

    @config_enumerate
    def model(data):
        x = _t(0, device)
        for i in pyro.markov(range(len(data) + 1)):
            if i < len(data):
                x = pyro.sample("x_{}".format(i), dist.Categorical(tp[x]))
                pyro.sample("y_{}".format(i), dist.Categorical(ep[x]), obs=data[i])
            else:
                pyro.sample("x_{}".format(i), dist.Categorical(tp[x]), obs=_t(1, device))

    def guide(data):
        pass

    data = torch.zeros(size, dtype=torch.long, device=device)
    elbo = TraceEnum_ELBO(max_plate_nesting=0)
    marginals = elbo.compute_marginals(model, guide, data)
    assert set(marginals.keys()) == {"x_{}".format(i) for i in range(size)}
    for i in range(size):
        assert marginals["x_{}".format(i)].batch_shape == ()
    for i in range(size - 1):
        d1, d2 = marginals["x_{}".format(i)], marginals["x_{}".format(i + 1)]
        assert d1.probs[0] > d2.probs[0] and d1.probs[1] < d2.probs[1]
    if size <= 10:
        want = _hmm_brute_force(size)
        got = np.stack([marginals["x_{}".format(i)].probs.cpu().double().numpy() for i in range(size)])
        tol = 1e-9 if tp.dtype == torch.float64 else 2e-5
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol)


def run_marginals_2678(device, observed):
    f = lambda v: _t(v, device, torch.get_default_dtype())  # noqa: E731

    @config_enumerate
    def model(a=None, b=None):
        a = pyro.sample("a", dist.Bernoulli(f(0.75)), obs=a)
        pyro.sample("b", dist.Bernoulli(1 - 0.25 * a), obs=b)

    def guide(a=None, b=None):
        pass

    kwargs = {name: f(1.0) for name in observed}
    TraceEnum_ELBO(strict_enumeration_warning=False).compute_marginals(model, guide, **kwargs)


def run_marginals_plated_golden(device, g, rtol):
    """A plated mixture (global + local enumerated sites, masked plate slice) against the
    reference's own compute_marginals (tests/golden/marginals.npz)."""
    dt = torch.get_default_dtype()
    data = torch.as_tensor(g["data"], device=device, dtype=dt)
    pi = torch.as_tensor(g["pi"], device=device, dtype=dt)
    locs = torch.as_tensor(g["locs"], device=device, dtype=dt)
    shift = torch.as_tensor(g["shift"], device=device, dtype=dt)

    @config_enumerate
    def model():
        s = pyro.sample("s", dist.Bernoulli(torch.as_tensor(0.3, device=device, dtype=dt)))
        with pyro.plate("data", len(data)):
            z = pyro.sample("z", dist.Categorical(pi))
            pyro.sample("obs", dist.Normal(locs[z] + shift * s, 1.0), obs=data)

    def guide():
        pass

    m = TraceEnum_ELBO(max_plate_nesting=1).compute_marginals(model, guide)
    np.testing.assert_allclose(m["s"].probs.cpu().double().numpy(), g["s_probs"], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(m["z"].probs.cpu().double().numpy(), g["z_probs"], rtol=rtol, atol=rtol)


def run_backwardsample_smoke(device, data):
    dt = torch.get_default_dtype()
    data = [None if d is None else torch.as_tensor(d, device=device) for d in data]
    if data[0] is not None:
        data[0] = data[0].to(dt)
    g = torch.Generator().manual_seed(0)
    loc0 = torch.randn(2, generator=g).to(device=device, dtype=dt)
    logits0 = torch.randn(3, 2, generator=g).to(device=device, dtype=dt)

    @config_enumerate
    def model(data):
        xs = list(data)
        zs = []
        for i in range(2):
            K = i + 2
            zs.append(pyro.sample("z_{}".format(i), dist.Categorical(torch.ones(K, device=device, dtype=dt))))
            if i == 0:
                loc = pyro.param("loc", loc0)[zs[i]]
                xs[i] = pyro.sample("x_{}".format(i), dist.Normal(loc, 1.0), obs=data[i])
            elif i == 1:
                logits = pyro.param("logits", logits0)[zs[i]]
                xs[i] = pyro.sample("x_{}".format(i), dist.Categorical(logits=logits), obs=data[i])
        z12 = zs[0] + 2 * zs[1]
        pyro.sample("z_12", dist.Categorical(torch.arange(6.0, device=device, dtype=dt)), obs=z12)
        return xs, zs

    def guide(data):
        pass

    pyro.clear_param_store()
    xs, zs = TraceEnum_ELBO(max_plate_nesting=1).sample_posterior(model, guide, data)
    for x, datum in zip(xs, data):
        assert datum is None or datum is x
    for z in zs:
        assert z.shape == ()
    pyro.clear_param_store()


def run_backwardsample_2(device, n=10000):
    dt = torch.get_default_dtype()

    @config_enumerate
    def model(data):
        with pyro.plate("particles", n):
            p_z = torch.tensor([0.1, 0.9], device=device, dtype=dt)
            x = pyro.sample("x", dist.Categorical(torch.tensor([0.5, 0.5], device=device, dtype=dt)))
            z = pyro.sample("z", dist.Bernoulli(p_z[x]), obs=data)
        return x, z

    def guide(data):
        pass

    x, z = TraceEnum_ELBO(max_plate_nesting=1).sample_posterior(
        model, guide, data=torch.zeros(n, device=device, dtype=dt))
    assert x.shape == (n,)
    assert abs(0.9 - (x.type_as(z) == z).float().mean().item()) < 0.05


def run_backwardsample_3(device, n=10000):
    dt = torch.get_default_dtype()

    @config_enumerate
    def model(data):
        with pyro.plate("particles", n):
            p_z = torch.tensor([[0.9, 0.1], [0.1, 0.9]], device=device, dtype=dt)
            x = pyro.sample("x", dist.Categorical(torch.tensor([0.5, 0.5], device=device, dtype=dt)))
            y = pyro.sample("y", dist.Categorical(torch.tensor([0.5, 0.5], device=device, dtype=dt)))
            z = pyro.sample("z", dist.Bernoulli(p_z[x, y]), obs=data)
        return x, y, z

    def guide(data):
        pass

    elbo = TraceEnum_ELBO(max_plate_nesting=1)
    x, y, z = elbo.sample_posterior(model, guide, data=torch.ones(n, device=device, dtype=dt))
    assert abs(0.9 - (x == y).float().mean().item()) < 0.05
    x, y, z = elbo.sample_posterior(model, guide, data=torch.zeros(n, device=device, dtype=dt))
    assert abs(0.1 - (x == y).float().mean().item()) < 0.05


def run_backwardsample_hmm_joint(device, size=4, n=4000):
    """The JOINT law of a backward-sampled chain (not only its marginals): pairwise posteriors of
    consecutive states against brute force."""
    dt = torch.get_default_dtype()
    tp = _t([[0.75, 0.25], [0.25, 0.75]], device, dt)
    ep = _t([[0.75, 0.25], [0.25, 0.75]], device, dt)

    @config_enumerate
    def model(data):
        with pyro.plate("draws", n):
            x = torch.zeros(n, dtype=torch.long, device=device)
            xs = []
            for i in pyro.markov(range(size)):
                x = pyro.sample("x_{}".format(i), dist.Categorical(tp[x]))
                pyro.sample("y_{}".format(i), dist.Categorical(ep[x]), obs=data[i].expand(n))
                xs.append(x)
        return xs            # (under enumeration the entries have different shapes: no stack here)

    def guide(data):
        pass

    data = torch.tensor([0, 1, 1, 0][:size], device=device)
    pyro.set_rng_seed(3)
    xs = TraceEnum_ELBO(max_plate_nesting=1).sample_posterior(model, guide, data)
    xs = torch.stack([x.reshape(n) for x in xs]).cpu().numpy()
    tpn, epn, d = tp.cpu().double().numpy(), ep.cpu().double().numpy(), data.cpu().numpy()
    joint = np.zeros((size - 1, 2, 2))
    for seq in itertools.product(range(2), repeat=size):
        p, prev = 1.0, 0
        for i, x in enumerate(seq):
            p *= tpn[prev, x] * epn[x, d[i]]
            prev = x
        for i in range(size - 1):
            joint[i, seq[i], seq[i + 1]] += p
    joint /= joint.sum((1, 2), keepdims=True)
    for i in range(size - 1):
        emp = np.zeros((2, 2))
        for a in range(2):
            for b in range(2):
                emp[a, b] = np.mean((xs[i] == a) & (xs[i + 1] == b))
        assert np.abs(emp - joint[i]).max() < 0.04, (i, emp, joint[i])
