"""The many-small-sites kernels (pa_multi_log_prob_sum / _grad, pa_meanfield_normal_sample / _bwd)
through the C-ABI against the numpy restatements in tests/oracle_backend.py, and the fused
ELBO-assembly / mean-field guide host paths against the per-site paths they replace.

Tolerances: float64 rtol 1e-11 (same arithmetic, different summation order); float32 vs the float64
restatement rtol 2e-5; Philox draws as in test_kernels_gpu.py.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import philox as o_philox
from tests import oracle_backend as ob


def _entries(rng, dtype, dev):
    """A table mixing every family, every broadcast pattern, masks and an identity term."""
    def t(a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)

    R, C = 48, 20
    ents = []
    # Normal: value full, loc row-broadcast ([1,C] -> summed over rows), scale scalar
    ents.append(dict(dist=0, rows=R, cols=C, value=t(rng.standard_normal((R, C))),
                     p0=t(rng.standard_normal((1, C))), p1=t(rng.uniform(0.5, 2, (1, 1))), mask=None,
                     coef=-1.0, need=(True, True, True)))
    # Normal: value full, loc col-broadcast ([R,1]), scale full, with a mask
    ents.append(dict(dist=0, rows=R, cols=C, value=t(rng.standard_normal((R, C))),
                     p0=t(rng.standard_normal((R, 1))), p1=t(rng.uniform(0.5, 2, (R, C))),
                     mask=torch.as_tensor(rng.uniform(size=(R, C)) < 0.7, device=dev), coef=2.5,
                     need=(True, True, True)))
    # Bernoulli-logits
    ents.append(dict(dist=1, rows=7, cols=300, value=t((rng.uniform(size=(7, 300)) < 0.4) * 1.0),
                     p0=t(rng.standard_normal((7, 300))), p1=None, mask=None, coef=1.0,
                     need=(False, True, False)))
    # HalfCauchy, LogNormal, Exponential, HalfNormal on positive values; scalar / row params
    pos = rng.uniform(0.1, 3, (5, 33))
    ents.append(dict(dist=2, rows=5, cols=33, value=t(pos), p0=t(rng.uniform(0.5, 2, (1, 1))), p1=None,
                     mask=None, coef=1.0, need=(True, True, False)))
    ents.append(dict(dist=3, rows=5, cols=33, value=t(pos), p0=t(rng.standard_normal((1, 33))),
                     p1=t(rng.uniform(0.5, 2, (5, 1))), mask=None, coef=-0.5, need=(True, True, True)))
    ents.append(dict(dist=4, rows=5, cols=33, value=t(pos), p0=t(rng.uniform(0.5, 2, (5, 33))), p1=None,
                     mask=None, coef=1.0, need=(True, True, False)))
    ents.append(dict(dist=5, rows=1, cols=33, value=t(pos[:1]), p0=t(rng.uniform(0.5, 2, (1, 33))),
                     p1=None, mask=None, coef=3.0, need=(True, True, False)))
    # identity: an already-reduced per-particle term
    ents.append(dict(dist=100, rows=1, cols=64, value=t(rng.standard_normal((1, 64))), p0=None, p1=None,
                     mask=None, coef=1.0, need=(True, False, False)))
    # an empty site
    ents.append(dict(dist=0, rows=0, cols=4, value=t(np.zeros((0, 4))), p0=t(np.zeros((1, 4))),
                     p1=t(np.ones((1, 1))), mask=None, coef=1.0, need=(False, False, False)))
    return ents


def _cpu(e):
    out = dict(e)
    for k in ("value", "p0", "p1", "mask"):
        if out[k] is not None:
            # keep the broadcast pattern visible to the restatement: size-1 dims stay size 1
            out[k] = out[k].cpu()
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_multi_log_prob_sum_and_grad_vs_restatement(gpu, dtype):
    from pyro_amd import kernels as k

    rng = np.random.default_rng(3)
    ents = _entries(rng, dtype, gpu)
    rtol = 2e-5 if dtype == torch.float32 else 1e-11
    tot = k.multi_log_prob_sum(ents, -0.25, dtype, gpu)
    cpu_ents = [_cpu(e) for e in ents]
    ref = ob.multi_log_prob_sum([dict(e, value=e["value"].double(),
                                      p0=None if e["p0"] is None else e["p0"].double(),
                                      p1=None if e["p1"] is None else e["p1"].double())
                                 for e in cpu_ents], -0.25, torch.float64, None)
    np.testing.assert_allclose(float(tot), float(ref), rtol=rtol)
    g = torch.tensor(1.7, dtype=dtype, device=gpu)
    grads = k.multi_log_prob_grad(g, ents, -0.25, dtype, gpu)
    ref_g = ob.multi_log_prob_grad(torch.tensor(1.7, dtype=torch.float64), cpu_ents, -0.25,
                                   torch.float64, None)
    for e, gs, rs in zip(ents, grads, ref_g):
        for a, b in zip(gs, rs):
            assert (a is None) == (b is None)
            if a is not None:
                assert tuple(a.shape) == tuple(b.shape)
                sc = max(1.0, float(b.abs().max())) if b.numel() else 1.0
                np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=rtol * 5, atol=rtol * 5 * sc)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_multi_sum_grad_in_one_launch_equals_the_two_launches(gpu, dtype):
    """pa_multi_log_prob_sum_grad (unit upstream gradient) == pa_multi_log_prob_sum followed by
    pa_multi_log_prob_grad(g = 1): the gradients bit for bit (same code, same launch geometry per
    entry), the total to summation-order tolerance (4 waves instead of 16)."""
    from pyro_amd import kernels as k

    rng = np.random.default_rng(8)
    ents = _entries(rng, dtype, gpu)
    tot2, grads2 = k.multi_log_prob_sum_grad(ents, -0.25, dtype, gpu)
    tot = k.multi_log_prob_sum(ents, -0.25, dtype, gpu)
    grads = k.multi_log_prob_grad(torch.ones((), dtype=dtype, device=gpu), ents, -0.25, dtype, gpu)
    np.testing.assert_allclose(float(tot2), float(tot), rtol=2e-6 if dtype == torch.float32 else 1e-13)
    for gs, hs in zip(grads, grads2):
        for a, b in zip(gs, hs):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b)


def test_multi_more_entries_than_one_launch(gpu):
    """> PA_MULTI_MAX_ENTRIES entries: chained launches accumulate into the same total."""
    from pyro_amd import kernels as k

    rng = np.random.default_rng(4)
    ents = []
    for i in range(37):
        v = torch.as_tensor(rng.standard_normal((3, 5)), device=gpu)
        ents.append(dict(dist=0, rows=3, cols=5, value=v, p0=torch.zeros((1, 1), dtype=v.dtype, device=gpu),
                         p1=torch.ones((1, 1), dtype=v.dtype, device=gpu), mask=None, coef=(-1.0) ** i,
                         need=(True, False, False)))
    tot = k.multi_log_prob_sum(ents, 1.0, torch.float64, gpu)
    ref = ob.multi_log_prob_sum([_cpu(e) for e in ents], 1.0, torch.float64, None)
    np.testing.assert_allclose(float(tot), float(ref), rtol=1e-11)
    grads = k.multi_log_prob_grad(torch.ones((), dtype=torch.float64, device=gpu), ents, 1.0,
                                  torch.float64, gpu)
    assert len(grads) == 37
    for i, (e, gs) in enumerate(zip(ents, grads)):
        np.testing.assert_allclose(gs[0].cpu().numpy(), (-1.0) ** i * -e["value"].cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_meanfield_sample_and_backward(gpu, dtype):
    from pyro_amd import kernels as k

    rng = np.random.default_rng(5)
    P, sizes = 64, [32, 1, 1000]
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    locs = [torch.as_tensor(rng.standard_normal(n), dtype=dtype, device=gpu) for n in sizes]
    rhos = [torch.as_tensor(rng.uniform(-3, 25, n), dtype=dtype, device=gpu) for n in sizes]
    offsets = [10, 700, 900]
    zs, scales, louts, epss = k.meanfield_normal_sample(locs, rhos, P, 11, offsets)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for n, loc, rho, off, z, sc, lo, eps in zip(sizes, locs, rhos, offsets, zs, scales, louts, epss):
        ref_eps = o_philox.normal(P * n, np_dt, 11, off).reshape(P, n)
        np.testing.assert_allclose(eps.cpu().numpy(), ref_eps, rtol=tol, atol=10 * tol)
        r = rho.double().cpu().numpy()
        ref_sc = np.where(r > 20, r, np.log1p(np.exp(np.minimum(r, 20))))
        np.testing.assert_allclose(sc.cpu().numpy(), ref_sc, rtol=20 * tol)
        assert torch.equal(lo, loc)
        np.testing.assert_allclose(z.cpu().numpy(), loc.cpu().numpy()[None] + ref_sc[None] * ref_eps,
                                   rtol=50 * tol, atol=50 * tol)
    d_zs = [torch.as_tensor(rng.standard_normal((P, n)), dtype=dtype, device=gpu) for n in sizes]
    d_zs[1] = None
    d_scs = [torch.as_tensor(rng.standard_normal(n), dtype=dtype, device=gpu) for n in sizes]
    d_scs[2] = None
    d_los = [None, torch.as_tensor(rng.standard_normal(1), dtype=dtype, device=gpu), None]
    d_locs, d_rhos = k.meanfield_normal_sample_bwd(rhos, epss, d_zs, d_scs, d_los, P)
    cpu = lambda xs: [None if x is None else x.double().cpu() for x in xs]   # noqa: E731
    r_locs, r_rhos = ob.meanfield_normal_sample_bwd(cpu(rhos), cpu(epss), cpu(d_zs), cpu(d_scs),
                                                    cpu(d_los), P)
    rt = 3e-5 if dtype == torch.float32 else 1e-11
    for a, b in zip(d_locs + d_rhos, r_locs + r_rhos):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=rt, atol=rt * 10)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("P", [64, 5])
def test_meanfield_backward_of_a_plated_site(gpu, dtype, P):
    """A site of thousands of columns next to small ones (config 5's w beside mu, tau, b): 64-column tiles
    x 4 row groups when P >= 16, the 256-column form otherwise; column sums against the float64
    restatement (float32 rtol 3e-5 of sums of P terms)."""
    from pyro_amd import kernels as k

    rng = np.random.default_rng(15)
    sizes = [32, 1, 4999]
    rhos = [torch.as_tensor(rng.uniform(-3, 25, n), dtype=dtype, device=gpu) for n in sizes]
    epss = [torch.as_tensor(rng.standard_normal((P, n)), dtype=dtype, device=gpu) for n in sizes]
    d_zs = [torch.as_tensor(rng.standard_normal((P, n)), dtype=dtype, device=gpu) for n in sizes]
    d_scs = [torch.as_tensor(rng.standard_normal(n), dtype=dtype, device=gpu) for n in sizes]
    d_los = [None, None, torch.as_tensor(rng.standard_normal(4999), dtype=dtype, device=gpu)]
    d_locs, d_rhos = k.meanfield_normal_sample_bwd(rhos, epss, d_zs, d_scs, d_los, P)
    again = k.meanfield_normal_sample_bwd(rhos, epss, d_zs, d_scs, d_los, P)
    for a, b in zip(d_locs + d_rhos, again[0] + again[1]):
        assert torch.equal(a, b)
    cpu = lambda xs: [None if x is None else x.double().cpu() for x in xs]   # noqa: E731
    r_locs, r_rhos = ob.meanfield_normal_sample_bwd(cpu(rhos), cpu(epss), cpu(d_zs), cpu(d_scs),
                                                    cpu(d_los), P)
    rt = 3e-5 if dtype == torch.float32 else 1e-11
    for a, b in zip(d_locs + d_rhos, r_locs + r_rhos):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=rt, atol=rt * 10)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_meanfield_sample_block_kernel_draws_the_same_numbers(gpu, dtype):
    """Launches whose largest site has >= 64 K elements take the kernel that draws one Philox block (4
    f32 / 2 f64 normals) per thread and trip instead of one element per thread: bit for bit the same
    outputs for every site, whatever element count / block remainder (sizes not multiples of 4)."""
    from pyro_amd import kernels as k

    rng = np.random.default_rng(6)
    P = 64
    small, big = [33, 1, 1001], [33, 1, 1001, 1027]           # 64 x 1027 = 65 728 elements
    locs = [torch.as_tensor(rng.standard_normal(n), dtype=dtype, device=gpu) for n in big]
    rhos = [torch.as_tensor(rng.uniform(-3, 25, n), dtype=dtype, device=gpu) for n in big]
    offsets = [10, 700, 900, 30000]
    a = k.meanfield_normal_sample(locs[:3], rhos[:3], P, 11, offsets[:3])
    b = k.meanfield_normal_sample(locs, rhos, P, 11, offsets)
    for group_a, group_b in zip(a, b):
        for ta, tb in zip(group_a, group_b[:3]):
            assert torch.equal(ta, tb)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    ref_eps = o_philox.normal(P * 1027, np_dt, 11, 30000).reshape(P, 1027)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    np.testing.assert_allclose(b[3][3].cpu().numpy(), ref_eps, rtol=tol, atol=10 * tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_meanfield_sample_whole_block_rows_draw_the_same_numbers(gpu, dtype):
    """Sites whose rows are whole Philox blocks (n % 4 == 0 in f32, % 2 in f64) take the block kernel's
    column-block form (softplus once per thread, 16-byte stores): bit for bit the per-element kernel's
    outputs; and a site with more column blocks than the grid has threads against the oracle's stream."""
    from pyro_amd import kernels as k

    rng = np.random.default_rng(8)
    P = 64
    small, big = [32, 4, 1000], [32, 4, 1000, 1028]           # 64 x 1028 = 65 792 elements
    locs = [torch.as_tensor(rng.standard_normal(n), dtype=dtype, device=gpu) for n in big]
    rhos = [torch.as_tensor(rng.uniform(-3, 25, n), dtype=dtype, device=gpu) for n in big]
    offsets = [10, 700, 900, 30000]
    a = k.meanfield_normal_sample(locs[:3], rhos[:3], P, 11, offsets[:3])      # per-element kernel
    b = k.meanfield_normal_sample(locs, rhos, P, 11, offsets)                  # block kernel
    for group_a, group_b in zip(a, b):
        for ta, tb in zip(group_a, group_b[:3]):
            assert torch.equal(ta, tb)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    ref_eps = o_philox.normal(P * 1028, np_dt, 11, 30000).reshape(P, 1028)
    np.testing.assert_allclose(b[3][3].cpu().numpy(), ref_eps, rtol=tol, atol=10 * tol)
    # more column blocks than threads in the grid (each thread walks several), 3 particles
    n, P3 = 1_200_000, 3
    loc = torch.as_tensor(rng.standard_normal(n), dtype=dtype, device=gpu)
    rho = torch.as_tensor(rng.uniform(-3, 3, n), dtype=dtype, device=gpu)
    zs, scales, louts, epss = k.meanfield_normal_sample([loc], [rho], P3, 5, [123])
    ref = o_philox.normal(P3 * n, np_dt, 5, 123).reshape(P3, n)
    np.testing.assert_allclose(epss[0].cpu().numpy(), ref, rtol=tol, atol=10 * tol)
    sp = torch.nn.functional.softplus(rho.double())
    torch.testing.assert_close(scales[0].double(), sp, rtol=tol * 4, atol=0)
    assert torch.equal(louts[0], loc)
    torch.testing.assert_close(zs[0].double(), loc.double() + scales[0].double() * epss[0].double(),
                               rtol=tol * 4, atol=tol * 4)


def _logreg_loss_and_grads(gpu, batched, fused_guide, monkeypatch):
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    from pyro_amd.distributions import fused

    N, D, P = 3000, 32, 64
    X, y = examples.synthetic_logreg_data(N, D, gpu, seed=0)
    pyro.clear_param_store()
    pyro.set_rng_seed(7)
    if not batched:
        monkeypatch.setattr(fused.SiteBatch, "add_site", lambda self, *a, **k: False)
    if not fused_guide:
        monkeypatch.setattr(AutoNormal, "_fused_draw", lambda self: None)
    guide = AutoNormal(examples.logreg_model, init_scale=0.1)
    elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
    loss = elbo.loss_and_grads(examples.logreg_model, guide, X, y)
    grads = {n: p.grad.detach().clone() for n, p in pyro.get_param_store().named_parameters()}
    monkeypatch.undo()
    return loss, grads


def test_batched_elbo_and_fused_guide_equal_per_site_paths(gpu, monkeypatch):
    """One ELBO-gradient evaluation of the config-2 model with (a) every site reduced on its own
    and the guide drawn site by site, (b) the batched ELBO assembly, (c) batched assembly + fused
    mean-field draw: same Philox draws, same loss and gradients up to float32 summation order."""
    import pyro_amd as pyro
    pyro.enable_validation(False)
    try:
        ref = _logreg_loss_and_grads(gpu, False, False, monkeypatch)
        for batched, fused_guide in ((True, False), (True, True)):
            out = _logreg_loss_and_grads(gpu, batched, fused_guide, monkeypatch)
            assert abs(out[0] - ref[0]) <= 2e-6 * abs(ref[0]), (out[0], ref[0])
            assert set(out[1]) == set(ref[1])
            for name in ref[1]:
                torch.testing.assert_close(out[1][name], ref[1][name], rtol=2e-4, atol=2e-4)
    finally:
        pyro.enable_validation(True)


# ---- full-covariance Normal guide (pa_mvn_tril_sample / _bwd) -------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n,P", [(1, 1), (6, 4), (33, 64), (200, 7), (700, 3)])
def test_mvn_tril_sample_kernels(gpu, dtype, n, P):
    """Forward against torch's own MultivariateNormal (rsample formula + log_prob through the
    triangular solve) in float64, backward against autograd of that formulation; the in-kernel
    Philox draws are the ones the stream's fill kernel produces."""
    from pyro_amd import kernels as k
    g = np.random.default_rng(n * 100 + P)
    loc = torch.tensor(g.standard_normal(n), dtype=dtype, device=gpu)
    rho = torch.tensor(g.uniform(-2.0, 1.5, n), dtype=dtype, device=gpu)
    if n > 3:
        rho[0] = 25.0                         # beyond the softplus threshold
    A = torch.tensor(g.standard_normal((n, n)) / np.sqrt(n), dtype=dtype, device=gpu)
    z, logq, eps = k.mvn_tril_sample(loc, rho, A, P, seed=11, offset=5)
    want_eps = k.philox_normal((P, n), dtype, gpu, 11, 5)
    assert torch.equal(eps, want_eps)
    z2, logq2, _ = k.mvn_tril_sample(loc, rho, A, P, eps=eps.clone())
    assert torch.equal(z, z2) and torch.equal(logq, logq2)

    l64, r64, A64 = (t.double().clone().requires_grad_(True) for t in (loc, rho, A))
    S = torch.nn.functional.softplus(r64)
    T = S[:, None] * (A64.tril(-1) + torch.eye(n, dtype=torch.float64, device=gpu))
    mvn = torch.distributions.MultivariateNormal(l64, scale_tril=T)
    z_ref = l64 + eps.double() @ T.T
    lq_ref = mvn.log_prob(z_ref)
    tol = 1e-11 if dtype == torch.float64 else 3e-5
    sc = float(z_ref.detach().abs().max())
    torch.testing.assert_close(z.double(), z_ref.detach(), rtol=tol, atol=tol * sc)
    torch.testing.assert_close(logq.double(), lq_ref.detach(), rtol=tol, atol=tol * n)

    d_z = torch.tensor(g.standard_normal((P, n)), dtype=dtype, device=gpu)
    d_q = torch.tensor(g.standard_normal(P), dtype=dtype, device=gpu)
    ((z_ref * d_z.double()).sum() + (lq_ref * d_q.double()).sum()).backward()
    d_loc, d_rho, d_A = k.mvn_tril_sample_bwd(loc, rho, eps, z, d_z, d_q)
    for got, ref in ((d_loc, l64.grad), (d_rho, r64.grad), (d_A, A64.grad)):
        torch.testing.assert_close(got.double(), ref, rtol=tol * 10, atol=tol * 10 * float(ref.abs().max() + 1))
    assert float(d_A.triu().abs().max()) == 0.0
    # accumulating form (the optimizer's flat gradient views)
    sinks = (torch.ones(n, dtype=dtype, device=gpu), torch.ones(n, dtype=dtype, device=gpu),
             torch.ones((n, n), dtype=dtype, device=gpu))
    k.mvn_tril_sample_bwd(loc, rho, eps, z, d_z, d_q, sinks=sinks)
    torch.testing.assert_close(sinks[0], d_loc + 1)
    torch.testing.assert_close(sinks[2], d_A + 1)


def test_fused_mvn_guide_equals_plain_posterior(gpu):
    """AutoMultivariateNormal through the fused draw == through MultivariateNormal (same Philox
    stream position, so the same standard normals): loss and every parameter gradient."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer import Trace_ELBO
    from pyro_amd.infer.autoguide import AutoMultivariateNormal, guides

    data = torch.tensor([0.3, -1.2, 2.0, 0.7, 0.1], dtype=torch.float64, device=gpu)

    def model(data):
        mu = pyro.sample("mu", dist.Normal(torch.zeros(3, dtype=torch.float64, device=gpu), 2.0).to_event(1))
        s = pyro.sample("s", dist.LogNormal(torch.zeros((), dtype=torch.float64, device=gpu), 0.5))
        with pyro.plate("d", 5):
            pyro.sample("x", dist.Normal(mu.sum(-1, keepdim=True), s.unsqueeze(-1)), obs=data)

    out = {}
    for fused in (True, False):
        pyro.clear_param_store()
        pyro.set_rng_seed(3)
        guide = AutoMultivariateNormal(model, init_scale=0.3)
        if not fused:
            guide.get_posterior = lambda *a, **k: guide._plain_posterior()
        elbo = Trace_ELBO(num_particles=8, vectorize_particles=True, max_plate_nesting=1)
        guide(data)                                       # create the parameters
        st = pyro.get_param_store()
        with torch.no_grad():
            g = torch.Generator().manual_seed(0)
            st._params["AutoMultivariateNormal.scale_tril"].copy_(
                (torch.randn(4, 4, generator=g, dtype=torch.float64) * 0.3).tril(-1).to(gpu))
        pyro.set_rng_seed(5)
        loss = elbo.loss_and_grads(model, guide, data)
        out[fused] = (loss, {n: p.grad.clone() for n, p in st._params.items()})
    assert isinstance(guide.get_posterior(), guides._GuideMVN)
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-10)
    for name, gr in out[False][1].items():
        torch.testing.assert_close(out[True][1][name], gr, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("shape,ed", [((64, 1, 32), 0), ((7, 3, 130), 1), ((5, 6, 7), 2), ((3, 100000), 1),
                                      ((0, 4), 1)])
def test_exp_site_kernels_against_the_restatement(dtype, shape, ed):
    """pa_exp_site_fwd / _bwd: value = lower + exp(u), log_density = -sum_event u, d u = d value exp(u)
    - d log_density.  float64 rtol 1e-13 (row sums in a different order); float32 vs the float64
    restatement rtol 2e-6 on the values, atol 2e-6 * sqrt(cols) * max|u| on the row sums."""
    from pyro_amd import kernels as k
    dev = torch.device("cuda:0")
    g = np.random.default_rng(5)
    u = torch.as_tensor(g.standard_normal(shape) * 1.5, dtype=dtype, device=dev)
    cols = int(np.prod(shape[len(shape) - ed:])) if ed else 1
    rt = 1e-13 if dtype == torch.float64 else 2e-6
    value, ld = k.exp_site_fwd(u, cols, 0.75)
    rv, rld = ob.exp_site_fwd(u.double().cpu(), cols, 0.75)
    torch.testing.assert_close(value.double().cpu(), rv, rtol=rt, atol=0)
    at = 1e-12 if dtype == torch.float64 else 2e-6 * np.sqrt(cols) * 6
    torch.testing.assert_close(ld.double().cpu(), rld, rtol=rt, atol=at)
    assert ld.shape == (u.numel() // cols if u.numel() else 0,)
    gv = torch.as_tensor(g.standard_normal(shape), dtype=dtype, device=dev)
    gl = torch.as_tensor(g.standard_normal(tuple(ld.shape)), dtype=dtype, device=dev)
    for a, b in ((gv, gl), (gv, None), (None, gl)):
        got = k.exp_site_bwd(value, a, b, cols, 0.75)
        ref = ob.exp_site_bwd(value.double().cpu(), None if a is None else a.double().cpu(),
                              None if b is None else b.double().cpu(), cols, 0.75)
        torch.testing.assert_close(got.double().cpu(), ref, rtol=rt * 4, atol=1e-13 if dtype == torch.float64 else 1e-5)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_exp_constrained_parameter_on_the_device(dtype):
    """pyro.param under greater_than(0.5) on the device: pa_exp_site_fwd without the Jacobian term and
    pa_exp_site_bwd against transform_to(constraint) (float32 rtol 2e-6)."""
    import pyro_amd as pyro
    from torch.distributions import constraints as C, transform_to
    dev = torch.device("cuda:0")
    pyro.clear_param_store()
    g = torch.Generator().manual_seed(1)
    init = (torch.rand(8, 1024, generator=g, dtype=torch.float64) * 3 + 0.6).to(dtype).to(dev)
    value = pyro.param("tw", init, constraint=C.greater_than(0.5))
    assert value.grad_fn is not None and "ExpLower" in type(value.grad_fn).__name__
    u = pyro.get_param_store()._params["tw"]
    ref = transform_to(C.greater_than(0.5))(u)
    rt = 1e-13 if dtype == torch.float64 else 2e-6
    torch.testing.assert_close(value, ref, rtol=rt, atol=0)
    w = torch.randn(8, 1024, generator=g, dtype=torch.float64).to(dtype).to(dev)
    got, = torch.autograd.grad((w * value).sum(), u)
    want, = torch.autograd.grad((w * ref).sum(), u)
    torch.testing.assert_close(got, want, rtol=rt * 4, atol=0)
    pyro.clear_param_store()


def test_positive_site_through_the_guide_equals_the_transform_path():
    """AutoNormal on a HalfNormal site on the device: value, log-density and parameter gradients from
    exp_site_fwd / _bwd equal those of biject_to(support)'s transform chain (float32: rtol 2e-6 on the
    values, 2e-5 on the gradients, which sum 64 particles)."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd import kernels as k, poutine
    from pyro_amd.infer.autoguide import AutoNormal
    dev = torch.device("cuda:0")

    def model():
        with pyro.plate("particles", 64, dim=-2):
            with pyro.plate("g", 32, dim=-1):
                pyro.sample("tau", dist.HalfNormal(torch.ones((), device=dev)))

    outs = []
    for fast in (True, False):
        pyro.clear_param_store()
        pyro.set_rng_seed(3)
        if not fast:
            import pyro_amd.infer.autoguide.guides as G
            real = G._exp_lower
            G._exp_lower = lambda t: None
        try:
            guide = AutoNormal(model)
            guide()                                    # (prototype + parameters)
            pyro.set_rng_seed(4)
            tr = poutine.trace(guide).get_trace()
            tr.compute_log_prob()
            loss = tr.nodes["tau"]["value"].sum() + 0.3 * tr.nodes["tau"]["log_prob"].sum()
            params = [pyro.get_param_store()._params[n] for n in sorted(pyro.get_param_store().keys())]
            grads = torch.autograd.grad(loss, params)
            outs.append((tr.nodes["tau"]["value"].detach().clone(), tr.nodes["tau"]["log_prob"].detach().clone(),
                         [x.clone() for x in grads]))
        finally:
            if not fast:
                G._exp_lower = real
    (v1, l1, g1), (v0, l0, g0) = outs
    torch.testing.assert_close(v1, v0, rtol=2e-6, atol=0)
    torch.testing.assert_close(l1, l0, rtol=2e-6, atol=2e-6)
    for a, b in zip(g1, g0):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_large_guide_site_closed_form_score_equals_autograd_through_log_prob(dtype, monkeypatch):
    """Config 5's guide site w [64, 1000, 32] (too large for the many-small-sites launch) scored by
    Normal.fused_score_term (pa_meanfield_score: value in one pass, total derivative -1/scale in closed
    form) against the same Philox draw scored by Normal.log_prob + autograd (log_prob_sum / grad / sum_to
    kernels).  float64: loss rel 1e-12, gradients rtol 1e-9 (the eps terms that cancel analytically cancel
    to rounding in autograd); float32: loss rel 2e-6, gradients rtol 2e-4 of each gradient's max."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.distributions.families import Normal
    from pyro_amd.infer import Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    dev = torch.device("cuda:0")
    G, D, P, N = 1000, 32, 64, 20000
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, D, dtype=dtype, generator=g).to(dev)
    y = (torch.rand(N, generator=g) < 0.5).to(dtype).to(dev)
    grp = torch.randint(0, G, (N,), generator=g).to(dev)

    def model():
        mu = pyro.sample("mu", dist.Normal(torch.zeros(D, dtype=dtype, device=dev), 1.0).to_event(1))
        tau = pyro.sample("tau", dist.HalfNormal(torch.ones(D, dtype=dtype, device=dev)).to_event(1))
        with pyro.plate("groups", G):
            w = pyro.sample("w", dist.Normal(mu, tau).to_event(1))
        with pyro.plate("data", N):
            pyro.sample("obs", dist.Bernoulli(logits=(w[..., grp, :] * X).sum(-1)), obs=y)

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
        guide = AutoNormal(model, init_scale=0.1)
        elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
        loss = elbo.loss_and_grads(model, guide)
        store = pyro.get_param_store()
        return loss, {n: store._params[n].grad.clone() for n in sorted(store.keys())}, calls

    loss1, g1, calls = run(True)
    assert calls == [(P, G, D)]
    loss0, g0, _ = run(False)
    f64 = dtype == torch.float64
    assert loss1 == pytest.approx(loss0, rel=1e-12 if f64 else 2e-6)
    for n in g0:
        scale = float(g0[n].abs().max())
        torch.testing.assert_close(g1[n], g0[n], rtol=1e-9 if f64 else 2e-4,
                                   atol=(1e-11 if f64 else 2e-4) * scale)
    pyro.clear_param_store()
