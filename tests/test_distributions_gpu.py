"""Reference distribution KATs (tests/distributions/test_delta.py, test_mask.py, test_categorical.py)
on the MI355X: Delta, .mask over the FUSED Bernoulli / Normal families, Categorical."""
import os

import numpy as np
import pytest
import torch

import pyro_amd.distributions as dist
from tests import dist_kat_cases as dk

pytestmark = pytest.mark.gpu


def test_delta(gpu):
    dk.run_delta(gpu)


@pytest.mark.parametrize("batch_dim,event_dim", [(b, e) for b in range(4) for e in range(1 + b)])
@pytest.mark.parametrize("has_log_density", [False, True])
def test_delta_shapes(gpu, batch_dim, event_dim, has_log_density):
    dk.run_delta_shapes(gpu, batch_dim, event_dim, has_log_density)


@pytest.mark.parametrize("batch_dim,mask_dim", [(b, m) for b in range(3) for m in range(1 + b)])
@pytest.mark.parametrize("event_dim", [0, 1, 2])
def test_mask_over_fused_bernoulli(gpu, batch_dim, event_dim, mask_dim):
    dk.run_mask(gpu, lambda shape: dist.Bernoulli(torch.tensor(0.1, device=gpu)).expand_by(shape),
                batch_dim, event_dim, mask_dim)


@pytest.mark.parametrize("mask", [False, True, torch.tensor(False), torch.tensor(True)])
def test_mask_type_fused_normal(gpu, mask):
    dk.run_mask_type(gpu, dist.Normal, mask)


@pytest.mark.parametrize("event_shape", [(), (4,)])
@pytest.mark.parametrize("dist_shape", [(), (3,), (2, 1), (2, 3)])
@pytest.mark.parametrize("mask_shape", [(), (3,), (2, 1), (2, 3)])
def test_mask_broadcast(gpu, event_shape, dist_shape, mask_shape):
    dk.run_mask_broadcast(gpu, dist.Normal, event_shape, dist_shape, mask_shape)


def test_mask_kl_divergence(gpu):
    dk.run_mask_kl(gpu, dist.Normal)


@pytest.mark.parametrize("p_mask", [False, True, torch.tensor(False), torch.tensor(True)])
@pytest.mark.parametrize("q_mask", [False, True, torch.tensor(False), torch.tensor(True)])
def test_mask_kl_divergence_type(gpu, p_mask, q_mask):
    dk.run_mask_kl_type(gpu, dist.Normal, p_mask, q_mask)


@pytest.mark.parametrize("shape", [None, (), (4,), (3, 2)], ids=str)
def test_mask_noop(gpu, shape):
    dk.run_mask_noop(gpu, dist.Normal, shape)


def test_categorical(gpu):
    dk.run_categorical(gpu)


def test_second_order_gradients_through_the_fused_families(gpu):
    dk.run_second_order_gradients(gpu)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_gamma_rsample_kernel_equals_the_oracle(gpu, dtype):
    """pa_gamma_rsample: the same Philox blocks and the same arithmetic as oracle/gamma.py -- draws
    and implicit gradients element for element (fp64 evaluation: 1e-12; the f32 tensor holds the
    rounded values), any launch geometry, device-side offset."""
    from oracle import gamma as o_gamma
    from pyro_amd import kernels as k
    rng = np.random.default_rng(0)
    alpha = np.exp(rng.uniform(np.log(0.05), np.log(300.0), size=(37, 211)))
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    ta = torch.as_tensor(alpha.astype(np_dt), device=gpu)
    off_dev = torch.tensor([1000], dtype=torch.int64, device=gpu)
    out, dal = k.gamma_rsample(ta, 37, 211, 42, 234, off_dev)
    a64 = alpha.astype(np_dt).astype(np.float64)
    ref = np.maximum(o_gamma.standard_gamma(a64, 42, 1234), np.finfo(np_dt).tiny)   # (clamped like torch's)
    # (f32: the fp64 draw rounded to f32; device and numpy log / cos / pow differ in the last fp64
    #  bits, which moves a rounding now and then: 4 ulp)
    tol = 1e-12 if dtype == torch.float64 else 5e-7
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=tol)
    refg = o_gamma.implicit_grad(a64, out.cpu().numpy().astype(np.float64))
    np.testing.assert_allclose(dal.cpu().numpy(), refg, rtol=1e-9 if dtype == torch.float64 else 5e-6)
    # a broadcast concentration (stride-0 view) reads the same blocks as the materialised one
    row = ta[:1]
    o2, _ = k.gamma_rsample(row, 37, 211, 42, 1234)
    o3, _ = k.gamma_rsample(row.expand(37, 211).contiguous(), 37, 211, 42, 1234)
    assert torch.equal(o2, o3)
    # the gradient alone, against the reference's own function on the golden grid
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gamma_grad.npz"))
    got = k.gamma_implicit_grad(torch.as_tensor(g["conc"], device=gpu), torch.as_tensor(g["value"], device=gpu))
    np.testing.assert_allclose(got.cpu().numpy(), g["grad"], rtol=2e-3)
    # large concentrations (1e4 .. 3e8, values within a +- 3 sqrt(a)): the iteration budget grows with
    # sqrt(a); against torch's function and, statement for statement, against the oracle
    al, xl = g["conc_large"], g["value_large"]
    got = k.gamma_implicit_grad(torch.as_tensor(al, device=gpu), torch.as_tensor(xl, device=gpu)).cpu().numpy()
    np.testing.assert_allclose(got, g["grad_large"], rtol=2e-5)
    np.testing.assert_allclose(got, o_gamma.implicit_grad(al, xl), rtol=1e-9)


def test_gamma_beta_dirichlet_rsample_distribution_and_pathwise_gradients(gpu):
    """The distribution classes draw through the kernel: Kolmogorov-Smirnov against scipy, and the
    pathwise gradients are unbiased: E[d x / d theta] = d E[x] / d theta for the means
    (Gamma: a / r; Beta: a / (a + b); Dirichlet: a_k / sum a), 4e5 draws, 5 standard errors."""
    from scipy import stats
    import pyro_amd as pyro
    from pyro_amd import distributions as dist
    pyro.set_rng_seed(9)
    n = 400_000
    a = torch.tensor(2.3, device=gpu, requires_grad=True)
    r = torch.tensor(1.7, device=gpu, requires_grad=True)
    x = dist.Gamma(a, r).rsample((n,))
    assert stats.kstest(x.detach().cpu().numpy()[:50000], stats.gamma(2.3, scale=1 / 1.7).cdf).pvalue > 1e-3
    x.mean().backward()
    se = float(x.std()) / n ** 0.5
    assert abs(float(a.grad) - 1 / 1.7) < 5 * se + 2e-3 and abs(float(r.grad) + 2.3 / 1.7 ** 2) < 5 * se + 2e-3
    a = torch.tensor(0.6, device=gpu, requires_grad=True)
    b = torch.tensor(3.1, device=gpu, requires_grad=True)
    x = dist.Beta(a, b).rsample((n,))
    assert stats.kstest(x.detach().cpu().numpy()[:50000], stats.beta(0.6, 3.1).cdf).pvalue > 1e-3
    x.mean().backward()
    assert abs(float(a.grad) - 3.1 / 3.7 ** 2) < 3e-3 and abs(float(b.grad) + 0.6 / 3.7 ** 2) < 3e-3
    c = torch.tensor([0.4, 1.5, 6.0], device=gpu, requires_grad=True)
    x = dist.Dirichlet(c).rsample((n,))
    assert torch.allclose(x.sum(-1), torch.ones(n, device=gpu), atol=1e-5)
    np.testing.assert_allclose(x.mean(0).detach().cpu().numpy(), np.array([0.4, 1.5, 6.0]) / 7.9, atol=3e-3)
    x[:, 0].mean().backward()
    np.testing.assert_allclose(c.grad.cpu().numpy(), np.array([7.5 / 7.9 ** 2, -0.4 / 7.9 ** 2, -0.4 / 7.9 ** 2]),
                               atol=3e-3)
    # replay-safe: the same (seed, offset) gives the same draws whatever else ran in between
    pyro.set_rng_seed(9)
    y = dist.Gamma(torch.tensor(2.3, device=gpu), torch.tensor(1.7, device=gpu)).rsample((n,))
    pyro.set_rng_seed(9)
    assert torch.equal(y, dist.Gamma(torch.tensor(2.3, device=gpu), torch.tensor(1.7, device=gpu)).rsample((n,)))

