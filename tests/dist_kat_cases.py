"""The reference's known-answer tests for the distribution wrappers that sit on the ELBO path --
Delta (autoguides), MaskedDistribution / ``.mask`` (poutine.mask, rows a4 / a5), Categorical
enumeration support (row a14) -- restated against the drop-in API
(tests/distributions/test_delta.py, test_mask.py, test_categorical.py).  ``device`` = CPU for the
torch-wrapped classes, the MI355X for the fused ones."""
import numpy as np
import torch
from torch.distributions import kl_divergence

import pyro_amd.distributions as dist
from pyro_amd.distributions.util import broadcast_shape, scale_and_mask


def _close(a, b, prec=1e-6):
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        a, b = torch.as_tensor(a), torch.as_tensor(b)
        assert a.shape == b.shape, (a.shape, b.shape)
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=prec, atol=prec)
    else:
        assert abs(a - b) <= prec


def checker_mask(shape, device):
    mask = torch.tensor(0, device=device)
    for size in shape:
        mask = mask.unsqueeze(-1) + torch.arange(float(size), device=device).long()
    return mask.fmod(2).bool()


# ---- test_delta.py ------------------------------------------------------------------------------
def run_delta(device):
    v = torch.tensor([3.0], device=device)
    vs = torch.tensor([[0.0], [1.0], [2.0], [3.0]], device=device)
    vs_expanded = vs.expand(4, 3)
    test_data = torch.tensor([[3.0], [3.0], [3.0]], device=device)
    b1 = torch.arange(0.0, 4.0, device=device).unsqueeze(1).expand(4, 3)
    b2 = torch.arange(4.0, 8.0, device=device).unsqueeze(1).expand(4, 3)
    b3 = torch.tensor([[3.0], [3.0], [3.0], [3.0]], device=device)
    assert dist.Delta(v).log_prob(test_data).sum().item() == 0
    assert dist.Delta(vs_expanded).log_prob(b1).sum().item() == 0
    assert dist.Delta(vs_expanded).log_prob(b2).sum().item() == float("-inf")
    assert dist.Delta(vs).log_prob(b3).size() == (4, 1)
    assert dist.Delta(v).log_prob(b3).size() == (4, 1)
    draws = [dist.Delta(v).sample().item() for _ in range(10)]
    assert np.mean(draws) == 3.0 and np.var(draws) == 0.0


def run_delta_shapes(device, batch_dim, event_dim, has_log_density):
    shape = tuple(range(2, 2 + batch_dim + event_dim))
    batch_shape = shape[:batch_dim]
    v = torch.randn(shape, device=device)
    log_density = torch.randn(batch_shape, device=device) if has_log_density else 0
    d = dist.Delta(v, log_density=log_density, event_dim=event_dim)
    x = d.rsample()
    assert (x == v).all()
    assert (d.log_prob(x) == log_density).all()


def run_delta_expand(device, batch_shape):
    d2 = dist.Delta(torch.tensor(1.234, device=device)).expand(batch_shape)
    assert d2.batch_shape == torch.Size(batch_shape)


# ---- test_mask.py -------------------------------------------------------------------------------
def run_mask(device, make_base, batch_dim, event_dim, mask_dim):
    shape = torch.Size([2, 3, 4, 5, 6][: batch_dim + event_dim])
    batch_shape = shape[:batch_dim]
    mask_shape = batch_shape[batch_dim - mask_dim:]
    base_dist = make_base(shape).to_event(event_dim)
    mask = checker_mask(mask_shape, device)
    d = base_dist.mask(mask)
    sample = base_dist.sample()
    assert d.batch_shape == base_dist.batch_shape and d.event_shape == base_dist.event_shape
    assert d.log_prob(sample).shape == base_dist.log_prob(sample).shape
    _close(d.mean, base_dist.mean)
    _close(d.variance, base_dist.variance)
    _close(d.log_prob(sample), scale_and_mask(base_dist.log_prob(sample), mask=mask))
    got, want = d.score_parts(sample), base_dist.score_parts(sample).scale_and_mask(mask=mask)
    for a, e in zip(got, want):
        _close(a, e, prec=0)
    if not d.event_shape and getattr(base_dist, "has_enumerate_support", False):
        for kw in ({}, {"expand": True}, {"expand": False}):
            _close(d.enumerate_support(**kw), base_dist.enumerate_support(**kw))


def run_mask_type(device, Normal, mask):
    p = Normal(torch.randn(2, 2, device=device), torch.randn(2, 2, device=device).exp())
    mask = mask if isinstance(mask, bool) else mask.to(device)
    p_masked = p.mask(mask)
    m = torch.tensor(mask, device=device) if isinstance(mask, bool) else mask
    x = p.sample()
    _close(p_masked.log_prob(x), p.log_prob(x) * m.float())
    for a, e in zip(p_masked.score_parts(x), p.score_parts(x)):
        if isinstance(e, torch.Tensor):
            e = e * m.float()
        _close(a, e)


def run_mask_broadcast(device, Normal, event_shape, dist_shape, mask_shape):
    mask = torch.empty(torch.Size(mask_shape), device=device).bernoulli_(0.5).bool()
    base = Normal(torch.zeros(dist_shape + event_shape, device=device), 1.0).to_event(len(event_shape))
    assert base.batch_shape == dist_shape and base.event_shape == event_shape
    d = base.mask(mask)
    assert d.batch_shape == broadcast_shape(mask.shape, base.batch_shape)
    assert d.event_shape == event_shape


def run_mask_kl(device, Normal):
    mask = torch.tensor([[0, 1], [1, 1]], device=device).bool()
    mk = lambda: Normal(torch.randn(2, 2, device=device), torch.randn(2, 2, device=device).exp())  # noqa: E731
    p, q = mk(), mk()
    expected = kl_divergence(p.to_event(2), q.to_event(2))
    actual = kl_divergence(p.mask(mask).to_event(2), q.mask(mask).to_event(2)) + \
        kl_divergence(p.mask(~mask).to_event(2), q.mask(~mask).to_event(2))
    _close(actual, expected, prec=1e-5)


def run_mask_kl_type(device, Normal, p_mask, q_mask):
    mk = lambda: Normal(torch.randn(2, 2, device=device), torch.randn(2, 2, device=device).exp())  # noqa: E731
    p, q = mk(), mk()
    as_t = lambda m: torch.tensor(m, device=device) if isinstance(m, bool) else m.to(device)  # noqa: E731
    mask = (as_t(p_mask) & as_t(q_mask)).expand(2, 2)
    expected = kl_divergence(p, q)
    expected[~mask] = 0
    pm = p_mask if isinstance(p_mask, bool) else p_mask.to(device)
    qm = q_mask if isinstance(q_mask, bool) else q_mask.to(device)
    actual = kl_divergence(p.mask(pm), q.mask(qm))
    if p_mask is False or q_mask is False:
        assert isinstance(actual, float) and actual == 0.0
    else:
        _close(actual, expected, prec=1e-5)


def run_mask_noop(device, Normal, shape):
    class NormalBomb(Normal):
        def log_prob(self, value):
            raise ValueError("Should not be called")

        def score_parts(self, value):
            raise ValueError("Should not be called")

        def expand(self, batch_shape, _instance=None):
            new = NormalBomb(self.loc.expand(batch_shape), self.scale.expand(batch_shape))
            return new

    d = NormalBomb(torch.zeros((), device=device), torch.ones((), device=device)).mask(False)
    if shape is not None:
        d = d.expand(shape)
    x = d.sample()
    zeros = torch.zeros(shape if shape else (), device=device)
    _close(d.log_prob(x), zeros)
    _close(d.score_parts(x).log_prob, zeros)


# ---- test_categorical.py ------------------------------------------------------------------------
def _wrap(x, dim):
    return x if dim < 1 else _wrap([x], dim - 1)


def run_categorical(device):
    probs = torch.tensor([0.1, 0.6, 0.3], device=device)
    batch_probs = torch.tensor([[0.1, 0.6, 0.3], [0.2, 0.4, 0.4]], device=device)
    lp = dist.Categorical(probs).log_prob(torch.tensor([2], device=device)).sum().item()
    _close(lp, float(np.log(0.3)))
    s = dist.Categorical(probs).enumerate_support()
    _close(s.float(), torch.tensor([0.0, 1.0, 2.0], device=device))
    s = dist.Categorical(batch_probs).enumerate_support()
    _close(s.float(), torch.tensor([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0]], device=device))
    for dim in (1, 2, 3):
        p = torch.tensor(_wrap([0.1, 0.6, 0.3], dim - 1), device=device)
        c = dist.Categorical(p)
        support = c.enumerate_support()
        assert support.size() == torch.Size((p.size(-1),) + p.size()[:-1])
        assert c.sample().size() == c.shape()
        assert c.log_prob(support).size() == torch.Size((3,) + c.batch_shape)
    logits = torch.randn((1, 2, 1, 3, 1, 2), device=device)      # test_view_reshape_bug
    dist.Categorical(logits=logits).sample((4,))


def run_second_order_gradients(device):
    """tests/infer/test_enum.py:307-335 takes a Newton step inside a guide: gradients of gradients of
    log_prob (``create_graph=True``) through the fused families, against torch's own classes."""
    import torch.distributions as td
    from torch.autograd import grad
    import pyro_amd.distributions as dist

    def t(x, rg=False):
        return torch.tensor(x, dtype=torch.float64, device=device, requires_grad=rg)

    cases = [
        (dist.Normal, td.Normal, [0.3, 1.3], t([0.0, 1.0, 3.0])),
        (dist.LogNormal, td.LogNormal, [0.3, 0.8], t([0.5, 1.0, 3.0])),
        (dist.Gamma, td.Gamma, [2.0, 1.5], t([0.5, 1.0, 3.0])),
        (dist.Beta, td.Beta, [2.0, 3.0], t([0.2, 0.5, 0.9])),
        (dist.HalfCauchy, td.HalfCauchy, [1.2], t([0.5, 1.0, 3.0])),
        (dist.Exponential, td.Exponential, [0.7], t([0.5, 1.0, 3.0])),
        (dist.Poisson, td.Poisson, [2.5], t([0.0, 1.0, 3.0])),
    ]
    for ours, theirs, params, data in cases:
        out = []
        for cls in (ours, theirs):
            ps = [t(p, True) for p in params]
            lp = cls(*ps).log_prob(data).sum()
            g = grad(lp, ps[:1], create_graph=True)[0]
            assert g.requires_grad
            H = grad(g, ps, allow_unused=True)
            out.append((g.detach(), [h.detach() for h in H if h is not None]))
        torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-9, atol=1e-11)
        for a, b in zip(out[0][1], out[1][1]):
            torch.testing.assert_close(a, b, rtol=1e-8, atol=1e-10)
    # the value as the differentiated input, and a first-order backward afterwards (the Newton step)
    x = t(0.2, True)
    scale = t(1.3, True)
    data = t([0.0, 1.0, 3.0])
    results = []
    for N in (dist.Normal, td.Normal):
        loss = -(N(t(0.0), t(10.0)).log_prob(x) + N(x, scale).log_prob(data).sum())
        g = grad(loss, [x], create_graph=True)[0]
        H = grad(g, [x], create_graph=True)[0]
        newton = x.detach() - g / H
        results.append((newton.detach(), grad(newton, [scale])[0]))
    torch.testing.assert_close(results[0][0], results[1][0], rtol=1e-9, atol=0)
    torch.testing.assert_close(results[0][1], results[1][1], rtol=1e-8, atol=1e-12)
