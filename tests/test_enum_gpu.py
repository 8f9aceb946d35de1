"""TraceEnum_ELBO on the MI355X: reference golden (f64) through the fused LDA kernel and the
generic device contraction, fused == generic at a larger size, and SVI on examples/lda.py."""
import os

import numpy as np
import pytest
import torch

import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO
from tests import enum_cases as ec

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("fused", [True, False])
def test_lda_matches_reference(gpu, monkeypatch, fused):
    if not fused:
        import pyro_amd.ops.contract as c
        monkeypatch.setattr(c, "_try_fused_lda", lambda *a: None)
    ec.run_lda(load("enum"), gpu, monkeypatch, rtol=1e-9, expect_fused=fused)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("sub", [False, True])
def test_gmm_matches_reference(gpu, monkeypatch, sub, fused):
    """The unmodified reference's TraceEnum_ELBO loss and gradients of a plated Gaussian mixture (whole plate and
    subsampled): through the mixture leaf kernel (csrc/mixture.hip) and through the generic contraction."""
    ec.run_gmm(load("enum"), gpu, monkeypatch, sub, rtol=1e-9, expect_fused=fused)


def _lda_loss_and_grads(args, data, fused, monkeypatch, seed=0):
    import pyro_amd.ops.contract as c
    if not fused:
        monkeypatch.setattr(c, "_try_fused_lda", lambda *a: None)
    pyro.clear_param_store()
    pyro.set_rng_seed(seed)
    torch.manual_seed(seed)
    predictor = examples.lda_make_predictor(args, data.device)
    guide = lambda data, args: examples.lda_guide(predictor, data, args)  # noqa: E731
    elbo = TraceEnum_ELBO(max_plate_nesting=2)
    loss = elbo.loss_and_grads(examples.lda_model, guide, data, args)
    grads = {n: p.grad.detach().clone() for n, p in pyro.get_param_store().named_parameters()
             if p.grad is not None}
    monkeypatch.undo()
    return loss, grads


def test_lda_fused_equals_generic_f32(gpu, monkeypatch):
    """examples/lda.py at 2000 docs x 64 words, T=8, V=1024 (f32): the fused kernel against the
    generic log-space contraction on the same draws (torch RNG for Gamma/Dirichlet)."""
    args = examples.LdaArgs(num_docs=2000)
    data = examples.synthetic_lda_data(args, gpu)
    assert data.dtype == torch.int64 and int(data.min()) >= 0 and int(data.max()) < args.num_words
    l1, g1 = _lda_loss_and_grads(args, data, True, monkeypatch)
    l2, g2 = _lda_loss_and_grads(args, data, False, monkeypatch)
    assert abs(l1 - l2) <= 2e-5 * abs(l2), (l1, l2)
    assert set(g1) == set(g2) and len(g1) >= 4
    for k in g1:
        scale = float(g2[k].abs().max()) + 1e-12
        assert float((g1[k] - g2[k]).abs().max()) <= 2e-4 * scale, k


def test_lda_svi_improves(gpu):
    args = examples.LdaArgs(num_docs=1000)
    data = examples.synthetic_lda_data(args, gpu)
    pyro.clear_param_store()
    pyro.set_rng_seed(0)
    predictor = examples.lda_make_predictor(args, gpu)
    guide = lambda data, args: examples.lda_guide(predictor, data, args)  # noqa: E731
    svi = SVI(examples.lda_model, guide, pyro.optim.TorchAdam({"lr": 0.01}),
              TraceEnum_ELBO(max_plate_nesting=2))
    losses = [svi.step(data, args) for _ in range(30)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < np.mean(losses[:5])


@pytest.mark.parametrize("dtype,rtol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
@pytest.mark.parametrize("fused_chain", [True, False], ids=["fused_chain", "generic"])
@pytest.mark.parametrize("which", [1, 3])
def test_hmm_under_markov_matches_reference(gpu, which, fused_chain, dtype, rtol):
    ec.run_hmm(load("hmm"), gpu, which, fused_chain=fused_chain, dtype=dtype, rtol=rtol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,T,K,shared", [(1, 1, 1, False), (5, 2, 3, False), (7, 9, 16, True),
                                          (3, 130, 64, False), (229, 129, 16, True), (229, 129, 16, False),
                                          (4, 23, 5, False), (6, 8, 17, False), (3, 40, 24, True),
                                          (2, 35, 32, False), (2, 9, 33, False)])
def test_logchain_kernel(gpu, dtype, B, T, K, shared):
    """pa_logchain_fwd_bwd against the numpy forward-backward restatement and torch autograd of a
    plain log-space forward recursion; -inf potentials (forbidden transitions) included."""
    from pyro_amd import kernels as k
    from tests import oracle_backend as ob
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + K)
    U = torch.randn(B, T, K, generator=g, dtype=torch.float64)
    P = torch.randn((T - 1, K, K) if shared else (B, max(T - 1, 0), K, K), generator=g,
                    dtype=torch.float64)
    if K > 2 and T > 1:
        P[..., 0, 1] = float("-inf")
    Ud, Pd = U.to(gpu, dtype).contiguous(), P.to(gpu, dtype).contiguous()
    lz, gu, gp = k.logchain_fwd_bwd(Ud, Pd)
    rlz, rgu, rgp = ob.logchain_fwd_bwd(U, P)
    tol = 1e-10 if dtype == torch.float64 else 3e-5
    torch.testing.assert_close(lz.double().cpu(), rlz, rtol=tol, atol=tol * T)
    torch.testing.assert_close(gu.double().cpu(), rgu, rtol=tol * 10, atol=tol * 10)
    torch.testing.assert_close(gp.double().cpu(), rgp, rtol=tol * 10, atol=tol * 10)
    # posteriors are distributions
    torch.testing.assert_close(gu.sum(-1), torch.ones_like(gu.sum(-1)), rtol=tol * 100, atol=tol * 100)
    if T <= 9:                                   # autograd of the plain recursion
        Ua = U.clone().requires_grad_(True)
        Pa = (P if not shared else P.expand(B, T - 1, K, K)).clone().requires_grad_(True)
        a = Ua[:, 0]
        for t in range(1, T):
            a = Ua[:, t] + torch.logsumexp(a[:, :, None] + Pa[:, t - 1], dim=1)
        torch.logsumexp(a, dim=1).sum().backward()
        torch.testing.assert_close(gu.double().cpu(), Ua.grad, rtol=tol * 10, atol=tol * 10)
        if T > 1:
            torch.testing.assert_close(gp.double().cpu(), Pa.grad, rtol=tol * 10, atol=tol * 10)


# ---- the reference's hand-vs-auto enumeration KATs (tests/enum_kat_cases.py) on the device -------
from tests import enum_kat_cases as ekc   # noqa: E402


@pytest.mark.parametrize("scale", [1, 10])
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_elbo_enumerate_chain(gpu, variant, scale):
    ekc.run_enumerate_chain(gpu, variant, scale)


@pytest.mark.parametrize("scale", [1, 10])
@pytest.mark.parametrize("num_samples,num_masked", [(1, 1), (2, 2), (3, 2)],
                         ids=["single", "batch", "masked"])
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_elbo_enumerate_plate(gpu, variant, num_samples, num_masked, scale):
    ekc.run_enumerate_plate(gpu, variant, num_samples, num_masked, scale)


@pytest.mark.parametrize("scale", [1, 10])
@pytest.mark.parametrize("variant", [1, 2])
def test_elbo_enumerate_plates(gpu, variant, scale):
    ekc.run_enumerate_plates(gpu, variant, scale)


def test_guide_enumeration_is_the_exact_expectation(gpu):
    ekc.run_guide_enumeration_closed_form(gpu)


def test_sequential_guide_enumeration_equals_parallel(gpu):
    ekc.run_sequential_equals_parallel(gpu)


@pytest.mark.parametrize("enums", [("sequential",) * 4, ("parallel",) * 4,
                                   ("parallel", "sequential", "parallel", "sequential"),
                                   ("sequential", "sequential", "parallel", "parallel")], ids="-".join)
def test_elbo_plate_plate(gpu, enums):
    ekc.run_elbo_plate_plate(gpu, enums)


@pytest.mark.parametrize("tmc,expand", [("diagonal", False), ("mixture", False), ("diagonal", True)])
def test_local_monte_carlo_sampling(gpu, tmc, expand):
    ekc.run_local_sampling(gpu, tmc=tmc, expand=expand)


def test_local_sampling_of_a_reparameterised_site(gpu):
    ekc.run_local_sampling_of_a_reparameterised_site(gpu)


@pytest.mark.parametrize("dtype,rtol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
def test_discrete_hmm_matches_reference(gpu, dtype, rtol):
    ec.run_discrete_hmm(load("discrete_hmm"), gpu, dtype=dtype, rtol=rtol)


def test_hmm_vectorised_over_time_equals_markov_model(gpu):
    ec.run_hmm_vectorised_equals_markov(gpu)


def test_guide_enumeration_and_dice_match_reference(gpu):
    ec.run_guide_enum_vs_reference(load("guide_enum"), gpu)


# ---- posterior of the enumerated sites (compute_marginals / sample_posterior) on the MI355X -------
from tests import posterior_kat_cases as pkc   # noqa: E402


@pytest.mark.parametrize("which,prior", [("bernoulli", 0.2), ("categorical", [0.2, 0.3, 0.5]),
                                         ("onehot", [0.2, 0.3, 0.3, 0.2])])
def test_compute_marginals_single(gpu, which, prior):
    pkc.run_marginals_single(gpu, which, prior)


@pytest.mark.parametrize("size", [3, 10, 20])
def test_compute_marginals_hmm(gpu, size):
    pkc.run_marginals_hmm(gpu, size)


def test_compute_marginals_matches_reference_on_a_plated_mixture(gpu):
    torch.set_default_dtype(torch.float64)
    try:
        pkc.run_marginals_plated_golden(gpu, np.load(os.path.join(os.path.dirname(__file__), "golden",
                                                                  "marginals.npz")), 1e-9)
    finally:
        torch.set_default_dtype(torch.float32)


def test_backwardsample_posterior(gpu):
    pkc.run_backwardsample_smoke(gpu, [0.0, None])
    pkc.run_backwardsample_2(gpu)
    pkc.run_backwardsample_3(gpu)
    pkc.run_backwardsample_hmm_joint(gpu)


def test_lda_minibatch_graphed_steps_draw_the_eager_subsamples(gpu):
    """examples/lda.py with batch_size (a fresh sub-sampled word matrix per step) under
    SVI(hip_graph=True): every replay draws a NEW subsample -- the same one the eager step draws --
    so the loss sequences agree step for step (learning rate 0: the parameters stay put)."""
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, TraceEnum_ELBO
    args = examples.LdaArgs(num_docs=3000)
    data = examples.synthetic_lda_data(args, gpu)
    seqs = []
    for graph in (False, True):
        pyro.clear_param_store(); pyro.set_rng_seed(0); torch.manual_seed(0)
        predictor = examples.lda_make_predictor(args, gpu)
        guide = lambda data, args: examples.lda_guide(predictor, data, args, 64)  # noqa: E731
        svi = SVI(examples.lda_model, guide, pyro.optim.ClippedAdam({"lr": 0.0}),
                  TraceEnum_ELBO(max_plate_nesting=2), hip_graph=graph, graph_warmup=3)
        seqs.append([svi.step(data, args) for _ in range(9)])
        if graph:
            assert len(svi._graphs) == 1
    assert len(set(seqs[0])) == 9                       # nine different subsamples
    np.testing.assert_allclose(seqs[0], seqs[1], rtol=1e-6)


@pytest.mark.parametrize("dist_id,name", [(0, "Normal"), (3, "LogNormal"), (4, "Exponential"), (1, "Bernoulli"),
                                           (8, "Poisson"), (6, "Gamma")])
@pytest.mark.parametrize("dtype,K,N", [(torch.float64, 5, 1237), (torch.float32, 16, 100_003), (torch.float64, 64, 777),
                                       (torch.float32, 1, 5000), (torch.float64, 3, 1)])
def test_mixture_kernel_against_the_oracle(gpu, dist_id, name, dtype, K, N):
    """pa_mixture_fwd_bwd (csrc/mixture.hip) against oracle/mixture.py: S and the three gradient sums for every
    family it scores, K not a power of two / 1 / 64, ragged N, one component switched off (a = -inf), a shared
    second parameter (stride 0).  float64: 1e-11; float32 kernels against float64 values: 2e-5 of S, 2e-4 of the
    largest gradient entry."""
    from oracle import mixture
    from pyro_amd import kernels

    rng = np.random.default_rng(11 * dist_id + K)
    a = np.log(rng.dirichlet(np.ones(K)))
    if K > 2:
        a[1] = -np.inf
    if name in ("Normal", "LogNormal"):
        p0, p1 = rng.standard_normal(K), rng.uniform(0.5, 2.0, 1)
        x = rng.standard_normal(N) * 2 if name == "Normal" else np.exp(rng.standard_normal(N))
    elif name == "Exponential":
        p0, p1, x = rng.uniform(0.3, 3.0, K), None, rng.exponential(1.0, N)
    elif name == "Bernoulli":
        p0, p1, x = rng.standard_normal(K) * 2, None, (rng.uniform(size=N) < 0.4).astype(np.float64)
    elif name == "Poisson":
        p0, p1, x = rng.uniform(0.5, 6.0, K), None, rng.poisson(3.0, N).astype(np.float64)
    else:
        p0, p1, x = rng.uniform(0.5, 4.0, K), rng.uniform(0.5, 2.0, K), rng.gamma(2.0, 1.0, N)
    to = lambda v: None if v is None else torch.tensor(v, dtype=dtype, device=gpu)  # noqa: E731
    tx, ta, t0, t1 = to(x), to(a), to(p0), to(p1)
    # (the oracle sees the values the kernel sees)
    back = lambda t: None if t is None else t.double().cpu().numpy()  # noqa: E731
    S, da, d0, d1 = mixture.mixture_fwd_bwd(dist_id, back(tx), back(ta), back(t0), back(t1))
    s1 = 0 if (t1 is None or t1.numel() == 1) else 1
    out = kernels.mixture_fwd_bwd(dist_id, tx, ta, t0, 1 if K > 1 else 0, t1, s1).cpu().numpy()
    again = kernels.mixture_fwd_bwd(dist_id, tx, ta, t0, 1 if K > 1 else 0, t1, s1).cpu().numpy()
    assert np.array_equal(out, again)                     # bit-reproducible
    tol_s, tol_g = (1e-11, 1e-10) if dtype == torch.float64 else (2e-5, 2e-4)
    np.testing.assert_allclose(out[0], S, rtol=tol_s)
    for got, want in ((out[1:1 + K], da), (out[1 + K:1 + 2 * K], d0), (out[1 + 2 * K:], d1)):
        if t1 is None and want is d1:
            continue
        scale = max(np.abs(want).max(), 1e-30)
        np.testing.assert_allclose(got / scale, want / scale, rtol=0, atol=tol_g)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_mixture_leaf_is_fused_and_equals_the_generic_contraction(gpu, monkeypatch, dtype):
    """TraceEnum_ELBO on a plated Gaussian mixture (the text of tests/golden/make_golden.py's gmm_model at
    N = 20 000, K = 7): the likelihood is recognised lazily (infer/traceenum_elbo.py::_lazy_family), the leaf runs
    pa_mixture_fwd_bwd -- no [K, N] tensor -- and loss and gradients equal the generic contraction's (which
    tests/test_enum_gpu.py::test_gmm_matches_reference pins on the reference)."""
    import pyro_amd.distributions as dist
    import pyro_amd.ops.contract as c
    from pyro_amd import kernels
    from torch.distributions import constraints

    K, N = 7, 20_000
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(N, generator=g, dtype=torch.float64) + 3.0 * torch.randint(0, K, (N,), generator=g)).to(gpu, dtype)
    locs0 = 3.0 * torch.arange(K, dtype=dtype, device=gpu) + 0.1

    def model(x):
        w = pyro.sample("w", dist.Dirichlet(torch.ones(K, dtype=dtype, device=gpu)))
        with pyro.plate("comp", K):
            locs = pyro.sample("locs", dist.Normal(torch.zeros((), dtype=dtype, device=gpu), 10.0))
        with pyro.plate("data", N):
            z = pyro.sample("z", dist.Categorical(w), infer={"enumerate": "parallel"})
            pyro.sample("x", dist.Normal(locs[z], 0.7), obs=x)

    def guide(x):
        ql = pyro.param("ql", locs0.clone())
        qs = pyro.param("qs", torch.tensor(0.3, dtype=dtype, device=gpu), constraint=constraints.positive)
        qw = pyro.param("qw", torch.full((K,), 1.0 / K, dtype=dtype, device=gpu), constraint=constraints.simplex)
        pyro.sample("w", dist.Delta(qw, event_dim=1))
        with pyro.plate("comp", K):
            pyro.sample("locs", dist.Normal(ql, qs))

    calls = []
    real = kernels.mixture_fwd_bwd
    monkeypatch.setattr(kernels, "mixture_fwd_bwd", lambda *a: calls.append(1) or real(*a))

    def run(fused):
        monkeypatch.setattr(c, "FUSED_MIXTURE", fused)
        pyro.clear_param_store(); pyro.set_rng_seed(1)
        loss = TraceEnum_ELBO(max_plate_nesting=1).loss_and_grads(model, guide, x)
        return loss, {n: p.grad.detach().clone() for n, p in pyro.get_param_store().named_parameters()}

    la, ga = run(True)
    assert len(calls) == 1, "the mixture leaf was not recognised"
    lb, gb = run(False)
    assert len(calls) == 1
    rtol = 1e-10 if dtype == torch.float64 else 2e-5
    assert abs(la - lb) <= rtol * abs(lb), (la, lb)
    for n in gb:
        scale = float(gb[n].abs().max()) + 1e-30
        assert float((ga[n] - gb[n]).abs().max()) <= (1e-9 if dtype == torch.float64 else 2e-4) * scale, n


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_mixture_kernel_with_a_batch_of_parameter_sets(gpu, dtype):
    """pa_mixture_fwd_bwd with B parameter sets over the same data (vectorised chains / particles): every set equals
    its own single evaluation bit for bit, and the oracle; a scale shared by the components AND the sets (strides
    0, 0), weights and locations per set."""
    from oracle import mixture
    from pyro_amd import kernels

    rng = np.random.default_rng(3)
    B, K, N = 5, 6, 3001
    x = torch.tensor(rng.standard_normal(N) * 2, dtype=dtype, device=gpu)
    a = torch.tensor(np.log(rng.dirichlet(np.ones(K), size=B)), dtype=dtype, device=gpu)
    p0 = torch.tensor(rng.standard_normal((B, K)), dtype=dtype, device=gpu)
    p1 = torch.tensor([0.8], dtype=dtype, device=gpu)
    out = kernels.mixture_fwd_bwd(0, x, a, p0.reshape(-1), 1, p1, 0, K, 0).cpu().numpy()
    assert out.shape == (B, 1 + 3 * K)
    for b in range(B):
        one = kernels.mixture_fwd_bwd(0, x, a[b].contiguous(), p0[b].contiguous(), 1, p1, 0).cpu().numpy()
        if b == 0:      # (the grid of a single set is larger: another summation order -- equal to rounding)
            np.testing.assert_allclose(out[b], one, rtol=1e-12 if dtype == torch.float64 else 1e-5)
        S, da, d0, d1 = mixture.mixture_fwd_bwd(0, x.double().cpu().numpy(), a[b].double().cpu().numpy(),
                                                p0[b].double().cpu().numpy(), p1.double().cpu().numpy())
        tol = 1e-11 if dtype == torch.float64 else 2e-5
        np.testing.assert_allclose(out[b, 0], S, rtol=tol)
        for got, want in ((out[b, 1:1 + K], da), (out[b, 1 + K:1 + 2 * K], d0), (out[b, 1 + 2 * K:], d1)):
            scale = np.abs(want).max()
            np.testing.assert_allclose(got / scale, want / scale, rtol=0, atol=10 * tol)


def test_mixture_leaf_under_vectorised_particles(gpu, monkeypatch):
    """TraceEnum_ELBO(num_particles = 4, vectorize_particles = True) on the plated Gaussian mixture: the particle
    plate is the leaf kernel's batch of parameter sets -- ONE launch for all particles -- and loss and gradients
    equal the generic contraction's on the same draws (float64, 1e-9)."""
    import pyro_amd.distributions as dist
    import pyro_amd.ops.contract as c
    from pyro_amd import kernels
    from pyro_amd.ops.indexing import Vindex
    from torch.distributions import constraints

    dtype = torch.float64
    K, N, P = 4, 5000, 4
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(N, generator=g, dtype=dtype) + 3.0 * torch.randint(0, K, (N,), generator=g)).to(gpu)

    def model(x):
        w = pyro.sample("w", dist.Dirichlet(torch.ones(K, dtype=dtype, device=gpu)))
        with pyro.plate("comp", K):
            locs = pyro.sample("locs", dist.Normal(torch.zeros((), dtype=dtype, device=gpu), 10.0))
        with pyro.plate("data", N):
            z = pyro.sample("z", dist.Categorical(w), infer={"enumerate": "parallel"})
            # (broadcast-safe under a particle dim: the component dim moves off the data plate's dim first)
            pyro.sample("x", dist.Normal(Vindex(locs.unsqueeze(-2))[..., z], 0.7), obs=x)

    def guide(x):
        ql = pyro.param("ql", 3.0 * torch.arange(K, dtype=dtype, device=gpu) + 0.1)
        qs = pyro.param("qs", torch.tensor(0.3, dtype=dtype, device=gpu), constraint=constraints.positive)
        qw = pyro.param("qw", torch.full((K,), 1.0 / K, dtype=dtype, device=gpu), constraint=constraints.simplex)
        pyro.sample("w", dist.Delta(qw, event_dim=1))
        with pyro.plate("comp", K):
            pyro.sample("locs", dist.Normal(ql, qs))

    calls = []
    real = kernels.mixture_fwd_bwd
    monkeypatch.setattr(kernels, "mixture_fwd_bwd", lambda *a: calls.append(tuple(a[2].shape)) or real(*a))

    def run(fused):
        monkeypatch.setattr(c, "FUSED_MIXTURE", fused)
        pyro.clear_param_store(); pyro.set_rng_seed(1)
        elbo = TraceEnum_ELBO(max_plate_nesting=1, num_particles=P, vectorize_particles=True)
        loss = elbo.loss_and_grads(model, guide, x)
        return loss, {n: p.grad.detach().clone() for n, p in pyro.get_param_store().named_parameters()}

    la, ga = run(True)
    assert calls == [(P, K)], calls
    lb, gb = run(False)
    assert abs(la - lb) <= 1e-10 * abs(lb), (la, lb)
    for n in gb:
        scale = float(gb[n].abs().max()) + 1e-30
        assert float((ga[n] - gb[n]).abs().max()) <= 1e-9 * scale, n


@pytest.mark.parametrize("dtype,B,K,D,N", [(torch.float64, 1, 5, 3, 1237), (torch.float32, 1, 16, 2, 100_003),
                                           (torch.float64, 3, 6, 8, 777), (torch.float32, 2, 1, 5, 4000),
                                           (torch.float64, 1, 64, 1, 300)])
def test_mixture_diag_normal_kernel_against_the_oracle(gpu, dtype, B, K, D, N):
    """pa_mixture_diag_normal_fwd_bwd (csrc/mixture.hip): the leaf for a diagonal Normal over D features -- S and the
    gradient sums per (set, component, feature) against oracle/mixture.py; a scale shared by the components, a
    component switched off, D not a power of two, D = 1 / 8, batches of parameter sets.  Bit-reproducible."""
    from oracle import mixture
    from pyro_amd import kernels

    rng = np.random.default_rng(100 * K + D)
    a = np.log(rng.dirichlet(np.ones(K), size=B))
    if K > 2:
        a[:, 1] = -np.inf
    loc = rng.standard_normal((B, K, D))
    scale = rng.uniform(0.5, 2.0, (B, 1, D))
    x = rng.standard_normal((N, D)) * 2
    to = lambda v: torch.tensor(v, dtype=dtype, device=gpu)  # noqa: E731
    tx, ta, tl, ts = to(x), to(a), to(loc), to(scale)
    S, da, dl, dc = (t.cpu().numpy() for t in kernels.mixture_diag_normal_fwd_bwd(tx, ta, tl, ts))
    S2, da2, dl2, dc2 = (t.cpu().numpy() for t in kernels.mixture_diag_normal_fwd_bwd(tx, ta, tl, ts))
    assert np.array_equal(S, S2) and np.array_equal(dl, dl2) and np.array_equal(dc, dc2)
    tol_s, tol_g = (1e-11, 1e-10) if dtype == torch.float64 else (2e-5, 3e-4)
    for b in range(B):
        wS, wa, wl, wc = mixture.mixture_diag_normal_fwd_bwd(tx.double().cpu().numpy(), ta[b].double().cpu().numpy(),
                                                             tl[b].double().cpu().numpy(),
                                                             np.broadcast_to(ts[b].double().cpu().numpy(), (K, D)))
        np.testing.assert_allclose(S[b], wS, rtol=tol_s)
        for got, want in ((da[b], wa), (dl[b], wl), (dc[b], wc)):
            sc = max(np.abs(want).max(), 1e-30)
            np.testing.assert_allclose(got / sc, want / sc, rtol=0, atol=tol_g)


@pytest.mark.parametrize("particles", [None, 3])
def test_multidimensional_mixture_leaf_equals_the_generic_contraction(gpu, monkeypatch, particles):
    """TraceEnum_ELBO on a Gaussian mixture over 3-dimensional data, `Normal(locs[z], scale).to_event(1)`: ONE launch
    of pa_mixture_diag_normal_fwd_bwd (all vectorised particles in it), loss and gradients equal to the generic
    contraction's on the same draws (float64, 1e-9)."""
    import pyro_amd.distributions as dist
    import pyro_amd.ops.contract as c
    from pyro_amd import kernels
    from pyro_amd.ops.indexing import Vindex
    from torch.distributions import constraints

    dtype = torch.float64
    K, N, D = 4, 3000, 3
    g = torch.Generator().manual_seed(9)
    centres = 3.0 * torch.randn(K, D, generator=g, dtype=dtype)
    x = (centres[torch.randint(0, K, (N,), generator=g)] + torch.randn(N, D, generator=g, dtype=dtype)).to(gpu)

    def model(x):
        w = pyro.sample("w", dist.Dirichlet(torch.ones(K, dtype=dtype, device=gpu)))
        sigma = pyro.sample("sigma", dist.LogNormal(torch.zeros(D, dtype=dtype, device=gpu), 0.5).to_event(1))
        with pyro.plate("comp", K):
            locs = pyro.sample("locs", dist.Normal(torch.zeros(D, dtype=dtype, device=gpu), 10.0).to_event(1))
        with pyro.plate("data", N):
            z = pyro.sample("z", dist.Categorical(w), infer={"enumerate": "parallel"})
            # (broadcast-safe under a particle dim: index the component dim, keep the feature dim last)
            m = Vindex(locs.unsqueeze(-3))[..., z, :]
            pyro.sample("x", dist.Normal(m, sigma).to_event(1), obs=x)

    def guide(x):
        ql = pyro.param("ql", centres.to(gpu) + 0.1)
        qs = pyro.param("qs", torch.tensor(0.3, dtype=dtype, device=gpu), constraint=constraints.positive)
        qw = pyro.param("qw", torch.full((K,), 1.0 / K, dtype=dtype, device=gpu), constraint=constraints.simplex)
        qsig = pyro.param("qsig", torch.ones(D, dtype=dtype, device=gpu), constraint=constraints.positive)
        pyro.sample("w", dist.Delta(qw, event_dim=1))
        pyro.sample("sigma", dist.LogNormal(qsig.log(), 0.05).to_event(1))
        with pyro.plate("comp", K):
            pyro.sample("locs", dist.Normal(ql, qs).to_event(1))

    calls = []
    real = kernels.mixture_diag_normal_fwd_bwd
    monkeypatch.setattr(kernels, "mixture_diag_normal_fwd_bwd",
                        lambda *a: calls.append((tuple(a[1].shape), tuple(a[2].shape))) or real(*a))

    def run(fused):
        monkeypatch.setattr(c, "FUSED_MIXTURE", fused)
        pyro.clear_param_store(); pyro.set_rng_seed(2)
        kw = {} if particles is None else {"num_particles": particles, "vectorize_particles": True}
        loss = TraceEnum_ELBO(max_plate_nesting=1, **kw).loss_and_grads(model, guide, x)
        return loss, {n: p.grad.detach().clone() for n, p in pyro.get_param_store().named_parameters()}

    la, ga = run(True)
    P = particles or 1
    assert calls == [((P, K), (P, K, D))], calls
    lb, gb = run(False)
    assert abs(la - lb) <= 1e-10 * abs(lb), (la, lb)
    for n in gb:
        scale = float(gb[n].abs().max()) + 1e-30
        assert float((ga[n] - gb[n]).abs().max()) <= 1e-9 * scale, n
