"""TraceEnum_ELBO test bodies shared by the CPU (oracle-backed kernels) and GPU suites; the
expected values come from the unmodified reference (tests/golden/enum.npz, make_golden.py G10)."""
import numpy as np
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
import pyro_amd.poutine as poutine
from pyro_amd.distributions import constraints
from pyro_amd.infer import TraceEnum_ELBO
from tests.models import EpsReplay, assert_grads, store_grads


def _t(x, device, dtype=torch.float64):
    return torch.tensor(np.asarray(x), dtype=dtype, device=device)


def run_lda(g, device, monkeypatch=None, rtol=1e-9, dtype=torch.float64, expect_fused=None):
    T, V = g["lda/twd0"].shape
    W, D = g["lda/data"].shape
    data = torch.tensor(g["lda/data"], device=device)

    def model(data):
        with pyro.plate("topics", T):
            topic_weights = pyro.sample("topic_weights", dist.Gamma(
                torch.full((), 1.0 / T, dtype=dtype, device=device), 1.0))
            topic_words = pyro.sample("topic_words", dist.Dirichlet(
                torch.ones(V, dtype=dtype, device=device) / V))
        with pyro.plate("documents", D):
            doc_topics = pyro.sample("doc_topics", dist.Dirichlet(topic_weights))
            with pyro.plate("words", W):
                word_topics = pyro.sample("word_topics", dist.Categorical(doc_topics),
                                          infer={"enumerate": "parallel"})
                pyro.sample("doc_words", dist.Categorical(topic_words[word_topics]), obs=data)

    def guide(data):
        a = pyro.param("tw", _t(g["lda/tw0"], device, dtype), constraint=constraints.positive)
        b = pyro.param("twd", _t(g["lda/twd0"], device, dtype), constraint=constraints.positive)
        c = pyro.param("dt", _t(g["lda/dt0"], device, dtype), constraint=constraints.simplex)
        with pyro.plate("topics", T):
            pyro.sample("topic_weights", dist.Delta(a))
            pyro.sample("topic_words", dist.Delta(b / b.sum(-1, keepdim=True), event_dim=1))
        with pyro.plate("documents", D):
            pyro.sample("doc_topics", dist.Delta(c, event_dim=1))

    pyro.clear_param_store()
    calls = []
    if expect_fused is not None:
        import pyro_amd.kernels as k
        orig = k.lda_factor_fwd_bwd

        def spy(*a, **kw):
            calls.append(1)
            return orig(*a, **kw)
        monkeypatch.setattr(k, "lda_factor_fwd_bwd", spy)
    loss = TraceEnum_ELBO(max_plate_nesting=2).loss_and_grads(model, guide, data)
    np.testing.assert_allclose(loss, float(g["lda/loss"]), rtol=rtol)
    assert_grads(store_grads(), g, "lda/grad", rtol)
    if expect_fused is not None:
        assert bool(calls) == expect_fused
    # differentiable_loss and loss agree with loss_and_grads
    elbo = TraceEnum_ELBO(max_plate_nesting=2)
    np.testing.assert_allclose(elbo.loss(model, guide, data), float(g["lda/loss"]), rtol=rtol)
    np.testing.assert_allclose(elbo.differentiable_loss(model, guide, data).item(),
                               float(g["lda/loss"]), rtol=rtol)


def run_gmm(g, device, monkeypatch, sub, rtol=1e-9, dtype=torch.float64, expect_fused=None):
    from pyro_amd import kernels, rng
    calls = []
    if expect_fused is not None:
        import pyro_amd.ops.contract as c
        monkeypatch.setattr(c, "FUSED_MIXTURE", bool(expect_fused))
        real = kernels.mixture_fwd_bwd
        monkeypatch.setattr(kernels, "mixture_fwd_bwd", lambda *a: calls.append(1) or real(*a))
    K = g["gmm/locs0"].shape[0]
    x = _t(g["gmm/x"], device, dtype)
    N = x.shape[0]
    tag = "gmmsub" if sub else "gmm"
    idx = torch.tensor(g["gmmsub/idx"], device=device) if sub else None

    def model(x, idx=None):
        w = pyro.sample("w", dist.Dirichlet(torch.ones(K, dtype=dtype, device=device)))
        with pyro.plate("comp", K):
            locs = pyro.sample("locs", dist.Normal(torch.zeros((), dtype=dtype, device=device), 3.0))
        with pyro.plate("data", N, subsample=idx):
            z = pyro.sample("z", dist.Categorical(w), infer={"enumerate": "parallel"})
            pyro.sample("x", dist.Normal(locs[z], 0.7), obs=x if idx is None else x[idx])

    def guide(x, idx=None):
        ql = pyro.param("ql", _t(g["gmm/locs0"], device, dtype))
        qs = pyro.param("qs", torch.tensor(0.3, dtype=dtype, device=device),
                        constraint=constraints.positive)
        qw = pyro.param("qw", _t(g["gmm/w0"], device, dtype), constraint=constraints.simplex)
        pyro.sample("w", dist.Delta(qw, event_dim=1))
        with pyro.plate("comp", K):
            pyro.sample("locs", dist.Normal(ql, qs))
        if idx is not None:
            with pyro.plate("data", N, subsample=idx):
                pass

    pyro.clear_param_store()
    eps = [g[k] for k in sorted(k for k in g.files if k.startswith(tag + "/eps/"))]
    monkeypatch.setattr(rng, "normal", EpsReplay(eps, device))
    args = (x, idx) if sub else (x,)
    loss = TraceEnum_ELBO(max_plate_nesting=1).loss_and_grads(model, guide, *args)
    np.testing.assert_allclose(loss, float(g[tag + "/loss"]), rtol=rtol)
    assert_grads(store_grads(), g, tag + "/grad", rtol)
    if expect_fused is not None:
        assert len(calls) == (1 if expect_fused else 0), "mixture leaf: %d kernel calls" % len(calls)


def lda_brute_force_loss(g):
    """Independent check of the golden LDA loss: exact marginal by explicit summation over the
    topic of every word (torch CPU, float64)."""
    import torch.distributions as td
    tw, twd, dt = torch.tensor(g["lda/tw0"]), torch.tensor(g["lda/twd0"]), torch.tensor(g["lda/dt0"])
    T, V = twd.shape
    data = torch.tensor(g["lda/data"])
    phi = twd / twd.sum(-1, keepdim=True)
    lp = td.Gamma(1.0 / T, 1.0).log_prob(tw).sum()
    lp = lp + td.Dirichlet(torch.ones(V, dtype=torch.float64) / V).log_prob(phi).sum()
    lp = lp + td.Dirichlet(tw).log_prob(dt).sum()
    W, D = data.shape
    for w in range(W):
        for d in range(D):
            lp = lp + torch.log((dt[d] * phi[:, data[w, d]]).sum())
    return -lp.item()


# ---- hidden Markov models under pyro.markov (tests/golden/hmm.npz from the reference) -------------
def run_hmm(g, device, which, rtol=1e-9, dtype=torch.float64, fused_chain=True):
    """examples/hmm.py model_1 (one hidden chain, emissions in a nested plate, ragged masked
    sequences) and model_3 (two hidden chains, factorial emission) with pyro.params: the loss is
    the exact negative log marginal likelihood; value and gradients against the reference."""
    from pyro_amd.distributions import constraints
    seqs, lengths = _t(g["sequences"], device, dtype), torch.tensor(g["lengths"], device=device)
    S, L, D = seqs.shape

    def model_1(sequences, lengths):
        probs_x = pyro.param("probs_x", _t(g["probs_x"], device, dtype), constraint=constraints.simplex)
        probs_y = pyro.param("probs_y", _t(g["probs_y"], device, dtype),
                             constraint=constraints.unit_interval)
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
        probs_w = pyro.param("probs_w", _t(g["probs_w"], device, dtype), constraint=constraints.simplex)
        probs_x = pyro.param("probs_x", _t(g["probs_x"], device, dtype), constraint=constraints.simplex)
        probs_y = pyro.param("probs_yw", _t(g["probs_yw"], device, dtype),
                             constraint=constraints.unit_interval)
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

    model, tag = (model_1, "m1") if which == 1 else (model_3, "m3")
    pyro.clear_param_store()
    elbo = TraceEnum_ELBO(max_plate_nesting=2)
    import pyro_amd.kernels as k
    import pyro_amd.ops.contract as contract
    calls = []
    real = k.logchain_fwd_bwd
    k.logchain_fwd_bwd = lambda u, p: (calls.append(tuple(u.shape)), real(u, p))[1]
    try:
        contract.FUSED_CHAIN = fused_chain
        loss = elbo.differentiable_loss(model, guide, seqs, lengths)
    finally:
        k.logchain_fwd_bwd = real
        contract.FUSED_CHAIN = True
    # model_1 is one chain per sequence: the whole elimination is ONE fused launch over [S, T, K]
    assert calls == ([(S, int(lengths.max()), g["probs_x"].shape[0])] if which == 1 and fused_chain
                     else []), calls
    np.testing.assert_allclose(loss.item(), float(g[tag + "/loss"]), rtol=rtol)
    names = sorted(pyro.get_param_store().keys())
    params = [pyro.param(n).unconstrained() for n in names]
    grads = torch.autograd.grad(loss, params)
    for n, gr in zip(names, grads):
        ref = g[tag + "/grad/" + n]
        np.testing.assert_allclose(gr.cpu().numpy(), ref, rtol=rtol * 100,
                                   atol=rtol * 100 * float(np.abs(ref).max()), err_msg=n)
    # the trace needs only history + 1 enumeration dims per chain, however long the sequence
    tr = poutine.trace(poutine.enum(model, first_available_dim=-3)).get_trace(seqs, lengths)
    dims = {s["infer"]["_enumerate_dim"] for s in tr.nodes.values()
            if s["type"] == "sample" and s["infer"].get("_enumerate_dim") is not None}
    assert len(dims) <= (2 if which == 1 else 4), dims


# ---- DiscreteHMM (tests/golden/discrete_hmm.npz from pyro/distributions/hmm.py) -------------------
def run_discrete_hmm(g, device, dtype=torch.float64, rtol=1e-9):
    """log_prob and gradients w.r.t. initial / transition logits and emission parameters against the
    reference's parallel-scan implementation, for per-batch, per-step and shared parameters."""
    for tag in ("hetero", "homog", "steps"):
        init, trans, loc = (_t(g[tag + "/" + k], device, dtype).requires_grad_(True)
                            for k in ("init", "trans", "loc"))
        d = dist.DiscreteHMM(init, trans, dist.Normal(loc, torch.tensor(0.7, dtype=dtype, device=device)))
        lp = d.log_prob(_t(g[tag + "/value"], device, dtype))
        np.testing.assert_allclose(lp.detach().cpu().numpy(), g[tag + "/log_prob"], rtol=rtol)
        lp.sum().backward()
        for got, name in ((init.grad, "g_init"), (trans.grad, "g_trans"), (loc.grad, "g_loc")):
            ref = g[tag + "/" + name]
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=rtol * 100,
                                       atol=rtol * 100 * float(np.abs(ref).max()), err_msg=tag + name)
    init, trans, py = (_t(g["bern/" + k], device, dtype).requires_grad_(True)
                       for k in ("init", "trans", "probs"))
    d = dist.DiscreteHMM(init, trans, dist.Bernoulli(py).to_event(1))
    value = _t(g["bern/value"], device, dtype)
    lp = d.log_prob(value)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), g["bern/log_prob"], rtol=rtol)
    lp.sum().backward()
    for got, name in ((init.grad, "g_init"), (trans.grad, "g_trans"), (py.grad, "g_probs")):
        ref = g["bern/" + name]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=rtol * 100,
                                   atol=rtol * 100 * float(np.abs(ref).max()), err_msg=name)
    # homogeneous parameters score data of any length; expand adds batch dims without copying
    assert d.expand([2, 3]).batch_shape == (2, 3) and d.event_shape == (1, value.shape[-1])
    lp2 = d.log_prob(value[:, :4])
    assert lp2.shape == (3,) and bool((lp2 > lp.detach()).all())


def run_hmm_vectorised_equals_markov(device, dtype=torch.float64, rtol=1e-9):
    """examples/hmm.py model_7's construction (time inside one DiscreteHMM site) has the likelihood
    of model_1 (one enumerated state per step under pyro.markov): same loss, same gradients."""
    from pyro_amd import examples
    torch.manual_seed(0)
    S, L, D, K = 5, 7, 4, 3
    seqs = (torch.rand(S, L, D) < 0.4).to(dtype).to(device)
    lengths = torch.tensor([7, 3, 5, 1, 6], device=device)
    out = []
    for model in (examples.hmm_model_1, examples.hmm_model_vectorised):
        pyro.clear_param_store()
        g = torch.Generator().manual_seed(1)
        pyro.param("probs_x", torch.softmax(torch.randn(K, K, generator=g, dtype=dtype), -1).to(device),
                   constraint=constraints.simplex)
        pyro.param("probs_y", (torch.rand(K, D, generator=g, dtype=dtype) * 0.8 + 0.1).to(device),
                   constraint=constraints.unit_interval)
        elbo = TraceEnum_ELBO(max_plate_nesting=2 if model is examples.hmm_model_1 else 1)
        loss = elbo.differentiable_loss(lambda s, l: model(s, l, K), lambda s, l: None, seqs, lengths)
        params = [pyro.param(n).unconstrained() for n in ("probs_x", "probs_y")]
        out.append((loss.item(), [x.cpu().numpy() for x in torch.autograd.grad(loss, params)]))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=rtol)
    for a, b in zip(out[1][1], out[0][1]):
        np.testing.assert_allclose(a, b, rtol=rtol * 100, atol=rtol * 100 * float(np.abs(b).max()))


# ---- guide-side enumeration + DiCE (tests/golden/guide_enum.npz from traceenum_elbo.py) -----------
class _NonreparameterizedNormal(dist.Normal):
    has_rsample = False


def run_guide_enum_vs_reference(g, device, dtype=torch.float64, rtol=1e-9):
    """Loss and gradients of the reference for (1) a masked, scaled plate with x enumerated in the
    guide and y in the model, (2) a score-function Normal upstream of a guide-enumerated site."""
    from pyro_amd import poutine
    from pyro_amd.infer import config_enumerate

    def t(v):
        return torch.tensor(v, dtype=dtype, device=device)

    def params():
        pyro.clear_param_store()
        pyro.param("guide_probs_x", t([0.1, 0.9]), constraint=constraints.simplex)
        pyro.param("model_probs_x", t([0.4, 0.6]), constraint=constraints.simplex)
        pyro.param("model_probs_y", t([[0.75, 0.25], [0.55, 0.45]]), constraint=constraints.simplex)
        pyro.param("model_probs_z", t([[0.3, 0.7], [0.2, 0.8]]), constraint=constraints.simplex)

    def check(tag, loss):
        names = sorted(n for n in pyro.get_param_store().keys() if tag + "/grad/" + n in g.files)
        ps = [pyro.param(n).unconstrained() for n in names]
        gs = torch.autograd.grad(loss, ps)
        np.testing.assert_allclose(loss.item(), float(g[tag + "/loss"]), rtol=rtol)
        for n, got in zip(names, gs):
            ref = g[tag + "/grad/" + n]
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=rtol * 100,
                                       atol=rtol * 100 * float(np.abs(ref).max()), err_msg=tag + n)

    data = torch.tensor([0, 1, 1], device=device)
    mask = torch.tensor([True, True, False], device=device)

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
    check("plate", TraceEnum_ELBO(max_plate_nesting=1, strict_enumeration_warning=False)
          .differentiable_loss(model1, guide1, data))

    zfix = torch.as_tensor(g["score/s"], dtype=dtype, device=device)

    def model2():
        s = pyro.sample("s", dist.Normal(t(0.0), t(1.0)))
        x = pyro.sample("x", dist.Categorical(pyro.param("model_probs_x")))
        pz = pyro.param("model_probs_z")
        pyro.sample("obs", dist.Normal(s + x.to(dtype), t(0.8)), obs=t(0.9))
        pyro.sample("z", dist.Categorical(pz[x]), obs=torch.tensor(1, device=device))

    @config_enumerate
    def guide2():
        loc = pyro.param("s_loc", t(0.2))
        pyro.sample("s", _NonreparameterizedNormal(loc, t(0.9)))
        pyro.sample("x", dist.Categorical(pyro.param("guide_probs_x")))

    params()
    fixed = poutine.trace(poutine.condition(guide2, data={"s": zfix})).get_trace()
    fixed.nodes["s"]["is_observed"] = False

    def guide2_fixed():
        tr = poutine.Trace()
        tr.add_node("s", **fixed.nodes["s"])
        return poutine.replay(guide2, trace=tr)()

    check("score", TraceEnum_ELBO(max_plate_nesting=0, strict_enumeration_warning=False)
          .differentiable_loss(model2, guide2_fixed))
