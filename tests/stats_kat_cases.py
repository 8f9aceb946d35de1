"""Known-answer tests of the chain diagnostics, restated from the reference's own suite
(tests/ops/test_stats.py:50-240) for pyro_amd.ops.stats; run on CPU and on the MI355X."""
import numpy as np
import torch

from pyro_amd.ops import stats


def _close(a, b, prec):
    a = a if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = torch.as_tensor(b, dtype=a.dtype, device=a.device)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert float((a - b).abs().max()) <= prec, (a, b)


def run_quantile_pi_hpdi(device):
    g = torch.Generator().manual_seed(3)
    x = torch.tensor([0.0, 1.0, 2.0], device=device)
    _close(stats.quantile(x, probs=[0.0, 0.4, 0.5, 1.0]), [0.0, 0.8, 1.0, 2.0], 1e-6)
    y = torch.rand(2000, generator=g).to(device)
    z = torch.randn(2000, generator=g).to(device)
    _close(stats.quantile(y, probs=0.2), 0.2, 0.02)
    _close(stats.quantile(z, probs=0.8413), 1.0, 0.04)
    e = torch.randn(1000, generator=g).exp().to(device)
    _close(stats.pi(e, prob=0.8), stats.quantile(e, probs=[0.1, 0.9]), 1e-6)
    n = torch.randn(20000, generator=g).to(device)
    _close(stats.hpdi(n, prob=0.8), stats.pi(n, prob=0.8), 0.03)
    ex = torch.empty(20000).exponential_(1, generator=g).to(device)
    _close(stats.hpdi(ex, prob=0.2), [0.0, 0.22], 0.01)


def run_interval_statistics_batch(device):
    g = torch.Generator().manual_seed(0)
    fns = [lambda x, dim=0: stats.quantile(x, probs=[0.1, 0.6], dim=dim),
           lambda x, dim=0: stats.pi(x, prob=0.8, dim=dim),
           lambda x, dim=0: stats.hpdi(x, prob=0.8, dim=dim)]
    for fn in fns:
        for sample_shape in ((), (3,), (2, 3)):
            xs = torch.rand((10,) + sample_shape, generator=g).to(device)
            y = fn(xs)
            assert y.shape == (2,) + xs.shape[1:]
            cols = [fn(x) for x in xs.reshape(10, -1).split(1, dim=1)]
            _close(torch.cat(cols, dim=1).reshape(y.shape), y, 1e-6)
            a = xs.transpose(0, -1)
            _close(fn(a, dim=-1), y.transpose(0, -1), 1e-6)


def run_autocorrelation(device):
    x = torch.arange(10.0, device=device)
    _close(stats.autocorrelation(x),
           [1, 0.78, 0.52, 0.21, -0.13, -0.52, -0.94, -1.4, -1.91, -2.45], 0.01)
    _close(stats.autocovariance(x),
           [8.25, 6.42, 4.25, 1.75, -1.08, -4.25, -7.75, -11.58, -15.75, -20.25], 0.01)
    _close(stats.autocorrelation(torch.zeros(10, device=device)), torch.ones(10), 0.01)
    g = torch.Generator().manual_seed(1)
    v = torch.randn(3, 4, 5, generator=g)
    v[1, 2] = 0
    v[2, 3] = 1
    v = v.to(device)
    actual = stats.autocorrelation(v, dim=-1)
    expected = torch.stack([torch.stack([stats.autocorrelation(xij) for xij in xi]) for xi in v])
    _close(actual, expected, 1e-5)
    assert bool((actual[1, 2] == 1).all()) and bool((actual[2, 3] == 1).all())
    for fn in (stats.autocorrelation, stats.autocovariance):
        for sample_shape in ((), (3,), (2, 3)):
            xs = torch.rand((10,) + sample_shape, generator=g).to(device)
            y = fn(xs)
            assert y.shape == xs.shape
            cols = [fn(x) for x in xs.reshape(10, -1).split(1, dim=1)]
            _close(torch.cat(cols, dim=1).reshape(xs.shape), y, 1e-5)
            _close(fn(xs.transpose(0, -1), dim=-1), y.transpose(0, -1), 1e-5)


def run_chain_diagnostics(device):
    x = torch.empty(2, 10)
    x[0] = torch.arange(10.0)
    x[1] = torch.arange(10.0) + 1
    _close(stats.gelman_rubin(x.to(device)), 0.98, 0.01)
    g = torch.Generator().manual_seed(2)
    u = torch.rand(2, 10, generator=g).to(device)
    _close(stats.split_gelman_rubin(u), stats.gelman_rubin(u.reshape(2, 2, 5).reshape(4, 5)), 1e-6)
    # the reference pins this value against arviz (tests/ops/test_stats.py:222-227)
    ess = stats.effective_sample_size(torch.arange(1000.0, device=device).reshape(100, 10))
    np.testing.assert_allclose(ess.item(), 52.64, atol=0.01)
    for fn in (stats.gelman_rubin, stats.split_gelman_rubin, stats.effective_sample_size):
        for sample_shape in ((), (3,), (2, 3)):
            xs = torch.rand((4, 100) + sample_shape, generator=g).to(device)
            y = fn(xs)
            assert y.shape == sample_shape
            cols = [fn(c) for c in xs.reshape(4, 100, -1).split(1, dim=2)]
            _close(torch.cat(cols, dim=0).reshape(sample_shape), y, 2e-4)
            a = xs.transpose(0, 1)
            b = xs.unsqueeze(-1).transpose(0, -1).squeeze(0)
            c = xs.unsqueeze(-1).transpose(1, -1).squeeze(1)
            _close(fn(a, chain_dim=1, sample_dim=0), y, 2e-4)
            _close(fn(b, chain_dim=-1, sample_dim=0), y, 2e-4)
            _close(fn(c, sample_dim=-1), y, 2e-4)
