"""End-to-end parity of the drop-in SVI path on the GPU (HIP kernels through the C-ABI) against
the golden vectors of the unmodified reference.  Tolerances: float64 element-wise / reduction
kernels 1e-9; float32 (including the fused f32-MFMA GLM kernel) vs the reference's own float32
run 2e-4 on the loss and 2e-3 (relative to the largest gradient entry) on gradients."""
import os
import warnings

import numpy as np
import pytest
import torch

from tests import models

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


@pytest.fixture(autouse=True)
def _clean():
    import pyro_amd
    pyro_amd.clear_param_store()
    yield
    torch.set_default_dtype(torch.float32)
    pyro_amd.clear_param_store()


def test_native_library_is_loaded(gpu):
    from pyro_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libpyro_amd.so" in maps


def test_eight_schools_f64(gpu, monkeypatch):
    torch.set_default_dtype(torch.float64)
    models.run_eight_schools(load("eight_schools"), gpu, monkeypatch, rtol=1e-9)


@pytest.mark.parametrize("tag", ["f64", "p1"])
def test_logreg_unfused_f64(gpu, monkeypatch, tag):
    torch.set_default_dtype(torch.float64)
    models.run_logreg(load("logreg_" + tag), gpu, monkeypatch, fused=False, dtype=torch.float64, rtol=1e-9)


@pytest.mark.parametrize("fused", [False, True])
def test_logreg_f32(gpu, monkeypatch, fused):
    models.run_logreg(load("logreg_f32"), gpu, monkeypatch, fused=fused, dtype=torch.float32, rtol=2e-4)


@pytest.mark.parametrize("tag", ["f64", "p1"])
def test_logreg_fused_f32_kernel_vs_f64_reference(gpu, monkeypatch, tag):
    """The f32 MFMA kernel against the reference's float64 numbers."""
    models.run_logreg(load("logreg_" + tag), gpu, monkeypatch, fused=True, dtype=torch.float32, rtol=2e-4)


def test_scale_mask_subsample(gpu, monkeypatch):
    torch.set_default_dtype(torch.float64)
    models.run_scale_mask(load("scale_mask"), gpu, monkeypatch, rtol=1e-9)


@pytest.mark.parametrize("dtype,rtol", [(torch.float64, 1e-9), (torch.float32, 3e-5)])
def test_gamma_function_families_in_an_elbo(gpu, dtype, rtol):
    """Gamma / Beta model and guide sites, Poisson / Binomial likelihoods: estimate and gradients
    of the reference (fixture expfam) through the fused site kernels."""
    torch.set_default_dtype(dtype)
    models.run_expfam(load("expfam"), gpu, rtol, dtype=dtype)


def test_score_function_guide(gpu):
    torch.set_default_dtype(torch.float64)
    models.run_score_function(load("score_function"), gpu, rtol=1e-9)


def test_tracegraph_baselines_match_reference(gpu):
    torch.set_default_dtype(torch.float64)
    models.run_tracegraph_baselines(load("tracegraph"), gpu, rtol=1e-9)


def test_tracegraph_provenance_matches_reference(gpu):
    torch.set_default_dtype(torch.float64)
    models.run_tracegraph_provenance(load("tracegraph_prov"), gpu, rtol=1e-9)


@pytest.mark.parametrize("tag", ["p1", "p5"])
def test_trace_mean_field_elbo(gpu, monkeypatch, tag):
    """TraceMeanField_ELBO (analytic KL + sampled fall-back) against the reference's loss / grads."""
    torch.set_default_dtype(torch.float64)
    models.run_meanfield(load("meanfield"), gpu, monkeypatch, tag, rtol=1e-9)


@pytest.mark.parametrize("dtype,rtol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
@pytest.mark.parametrize("P", [1, 6])
def test_fused_normal_kl_equals_kl_divergence(gpu, monkeypatch, dtype, rtol, P):
    """The Normal/Normal (and LogNormal/LogNormal) KL terms of TraceMeanField_ELBO ride in the
    multi-site launch as two entries each; with the route switched off the same terms come from
    torch.distributions.kl_divergence.  Same loss, same gradients -- on sites with event dims,
    under a masked + subsample-scaled plate, with broadcast prior parameters and particles."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd import poutine, rng
    from pyro_amd.infer import TraceMeanField_ELBO, trace_mean_field_elbo as tmf
    from torch.distributions import constraints
    torch.set_default_dtype(dtype)
    G, N = 5, 9
    gen = torch.Generator().manual_seed(3)
    data = torch.randn(N, G, generator=gen).to(gpu)
    mask = (torch.rand(G, generator=gen) < 0.7).to(gpu)
    c = lambda *x: torch.tensor(x if len(x) > 1 else x[0], device=gpu)    # noqa: E731

    def model(data):
        w = pyro.sample("w", dist.Normal(torch.zeros(3, device=gpu), c(1.0, 2.0, 0.5)).to_event(1))
        s = pyro.sample("s", dist.LogNormal(c(0.0), 0.4))
        with pyro.plate("g", 2 * G, subsample_size=G, dim=-1), poutine.mask(mask=mask):
            m = pyro.sample("m", dist.Normal(w.sum(-1), 1.5))
            with pyro.plate("n", N, dim=-2):
                pyro.sample("x", dist.Normal(m, s), obs=data)

    def guide(data):
        wl = pyro.param("wl", c(0.2, -0.1, 0.3))
        ws = pyro.param("ws", c(0.4, 0.6, 0.8), constraint=constraints.positive)
        sl = pyro.param("sl", c(-0.2))
        ss = pyro.param("ss", c(0.3), constraint=constraints.positive)
        ml = pyro.param("ml", torch.linspace(-1, 1, 2 * G, device=gpu))
        ms = pyro.param("ms", torch.linspace(0.3, 0.9, 2 * G, device=gpu), constraint=constraints.positive)
        pyro.sample("w", dist.Normal(wl, ws).to_event(1))
        pyro.sample("s", dist.LogNormal(sl, ss))
        with pyro.plate("g", 2 * G, subsample_size=G, dim=-1) as idx, poutine.mask(mask=mask):
            pyro.sample("m", dist.Normal(ml[idx], ms[idx]))

    def run(fused_route):
        pyro.clear_param_store()
        pyro.set_rng_seed(11)
        if not fused_route:
            monkeypatch.setattr(tmf, "_add_normal_kl", lambda *a: False)
        taken = []
        real = tmf._add_normal_kl
        monkeypatch.setattr(tmf, "_add_normal_kl", lambda *a: taken.append(real(*a)) or taken[-1])
        elbo = TraceMeanField_ELBO(num_particles=P, vectorize_particles=P > 1, max_plate_nesting=2)
        loss = elbo.loss_and_grads(model, guide, data)
        grads = {k: v.grad.clone() for k, v in pyro.get_param_store().named_parameters()}
        monkeypatch.undo()
        return loss, grads, taken

    loss_f, grads_f, taken = run(True)
    assert taken == [True, True, True]
    loss_t, grads_t, taken = run(False)
    assert taken == [False, False, False]
    np.testing.assert_allclose(loss_f, loss_t, rtol=rtol)
    assert set(grads_f) == set(grads_t) == {"wl", "ws", "sl", "ss", "ml", "ms"}
    for k in grads_f:
        np.testing.assert_allclose(grads_f[k].cpu().numpy(), grads_t[k].cpu().numpy(), rtol=rtol * 10,
                                   atol=rtol * 10 * float(grads_t[k].abs().max()))


@pytest.mark.parametrize("which", ["diag", "mvn"])
@pytest.mark.parametrize("tag", ["p1", "p4"])
def test_autocontinuous_guides(gpu, monkeypatch, which, tag):
    """AutoDiagonalNormal / AutoMultivariateNormal against the reference's loss and gradients."""
    torch.set_default_dtype(torch.float64)
    models.run_autocont(load("autocont"), gpu, monkeypatch, which, tag, rtol=1e-9)


@pytest.mark.parametrize("which", ["diag", "mvn"])
@pytest.mark.parametrize("tag", ["p1", "p4"])
def test_autocontinuous_guides_under_trace_mean_field(gpu, monkeypatch, which, tag):
    """The guides' Delta sites under TraceMeanField_ELBO: kl_divergence(Delta, prior), as the
    reference registers it -- loss and gradients of the reference."""
    torch.set_default_dtype(torch.float64)
    models.run_autocont(load("autocont"), gpu, monkeypatch, which, tag, rtol=1e-9, mean_field=True)


def test_predictive(gpu, monkeypatch):
    """Predictive (vectorised and sequential) against the reference's draws (recorded normals)."""
    torch.set_default_dtype(torch.float64)
    models.run_predictive(load("predictive"), gpu, monkeypatch, rtol=1e-10)


def test_svi_converges_to_reference_posterior(gpu):
    """Posterior means after optimisation match the reference's (north-star criterion): the
    deterministic analytic optimum of the Gaussian family is the comparison point -- both
    implementations are run with the SAME injected noise bank on a small problem."""
    import pyro_amd as pyro
    from pyro_amd import rng
    from oracle.ref_port_torch import LogRegAutoNormalPort
    N, D, P = 2000, 8, 16
    g = np.random.default_rng(0)
    X = g.standard_normal((N, D)).astype(np.float32)
    w_true = g.standard_normal(D)
    y = (g.uniform(size=N) < 1 / (1 + np.exp(-X @ w_true))).astype(np.float32)
    steps = 150
    bank = [(g.standard_normal((P, 1, D)), g.standard_normal((P, 1))) for _ in range(steps)]
    # reference arithmetic (torch CPU port of the reference step, pinned against golden)
    port = LogRegAutoNormalPort(torch.tensor(X), torch.tensor(y), P, lr=0.05)
    for ew, eb in bank:
        port.loss_and_grads(torch.tensor(ew, dtype=torch.float32), torch.tensor(eb, dtype=torch.float32))
        for o in port.optims:
            o.step()
        for p in port.params:
            p.grad = None
    # HIP path
    Xt, yt = torch.as_tensor(X, device=gpu), torch.as_tensor(y, device=gpu)
    pyro.clear_param_store()
    guide = pyro.infer.autoguide.AutoNormal(models.logreg_model_fused, init_scale=0.1)
    svi = pyro.infer.SVI(models.logreg_model_fused, guide, pyro.optim.Adam({"lr": 0.05}),
                         pyro.infer.Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1),
                         hip_graph=False)        # (injected draws: host code of every step)
    guide._setup_prototype(Xt, yt)
    flat = [e for pair in bank for e in pair]
    orig = rng.normal
    rng.normal = models.EpsReplay(flat, gpu)
    try:
        for _ in range(steps):
            svi.step(Xt, yt)
    finally:
        rng.normal = orig
    mean = pyro.get_param_store()["AutoNormal.locs.w"].detach().cpu().numpy()
    ref = port.loc_w.detach().numpy()
    rel = np.abs(mean - ref).max() / np.abs(ref).max()
    assert rel < 1e-4, rel


def test_hip_graph_step_equals_eager_step(gpu):
    """SVI(hip_graph=True): after the eager warm-up steps the captured step is replayed; the
    Philox stream advances on the device, so losses and parameters follow EXACTLY the eager
    trajectory (deterministic kernels => bit-identical)."""
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=3)
    runs = []
    for use_graph in (False, True, "split", "auto"):
        pyro.clear_param_store()
        pyro.set_rng_seed(7)
        pyro.enable_validation(False)
        try:
            guide = AutoNormal(examples.logreg_model, init_scale=0.1)
            optim = pyro.optim.Adam({"lr": 0.02})
            if use_graph == "split":
                # the multi-GPU shape of the step on one GPU: [loss+backward] graph, eager
                # (here: no-op) gradient all-reduce, [optimizer] graph
                optim = pyro.optim.RcclOptimizer(optim)
            elbo = Trace_ELBO(num_particles=16, vectorize_particles=True, max_plate_nesting=1)
            if use_graph == "auto":
                # the reference's constructor and nothing else: device arguments => captured by itself
                svi = SVI(examples.logreg_model, guide, optim, elbo)
                assert svi._auto_graph
            else:
                svi = SVI(examples.logreg_model, guide, optim, elbo, hip_graph=bool(use_graph), graph_warmup=3)
            svi._force_split = use_graph == "split"
            losses = [svi.step(X, y) for _ in range(12)]
            if use_graph:
                assert svi.hip_graph and len(svi._graphs) == 1   # captured, not fallen back
                entry = next(iter(svi._graphs.values()))
                assert (entry.graph2 is not None) == (use_graph == "split")
            params = {k: v.detach().clone() for k, v in pyro.get_param_store().items()}
        finally:
            pyro.enable_validation(True)
        runs.append((losses, params))
    for other in runs[1:]:
        assert runs[0][0] == other[0], (runs[0][0], other[0])
        for k in runs[0][1]:
            assert torch.equal(runs[0][1][k], other[1][k]), k
    assert runs[0][0][-1] < runs[0][0][0]


def test_default_svi_captures_only_what_can_be_a_graph(gpu):
    """SVI(model, guide, optim, loss) decides by itself: device tensors as arguments => captured step;
    host tensors, no arguments at all (a model that closes over its data), a wrapped torch optimizer, a
    capture that fails (the model synchronises with the host) => eager steps, silently; the
    svi_capture_steps setting switches the decision off."""
    import warnings

    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(4096, 8, gpu, seed=1)
    elbo = lambda: Trace_ELBO(num_particles=4, vectorize_particles=True, max_plate_nesting=1)  # noqa: E731

    def build(model, optim=None, **kw):
        pyro.clear_param_store()
        pyro.set_rng_seed(1)
        return SVI(model, AutoNormal(model, init_scale=0.1), optim or pyro.optim.Adam({"lr": 0.01}), elbo(), **kw)

    pyro.enable_validation(False)
    try:
        from pyro_amd.infer.svi import CapturedStepWarning
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            warnings.simplefilter("ignore", CapturedStepWarning)    # (said once per process: its own test below)
            svi = build(examples.logreg_model)
            for _ in range(6):
                svi.step(X, y)
            assert svi.hip_graph and len(svi._graphs) == 1
            closed = lambda: examples.logreg_model(X, y)  # noqa: E731
            svi = build(closed)
            for _ in range(6):
                svi.step()
            assert not svi._graphs

            def syncing(X, y):
                w = pyro.sample("w", dist.Normal(X.new_zeros(X.shape[1]), 1.0).to_event(1))
                float(w.detach().sum())             # a host read: not capturable
                with pyro.plate("data", X.shape[0]):
                    logits = w @ X.t()
                    pyro.sample("obs", dist.Bernoulli(logits=logits.squeeze(-2) if logits.dim() > 1 else logits),
                                obs=y)
            svi = build(syncing)
            losses = [svi.step(X, y) for _ in range(6)]
            assert not svi.hip_graph and not svi._graphs and all(np.isfinite(losses))
            # a wrapped torch optimizer: the loss and the gradients are captured, the update runs eagerly behind
            # every replay (test_torch_optimizer_behind_a_captured_loss_follows_the_eager_trajectory)
            svi = build(examples.logreg_model, optim=pyro.optim.PyroOptim(torch.optim.Adam, {"lr": 0.01}))
            for _ in range(5):
                svi.step(X, y)
            assert svi.hip_graph and len(svi._graphs) == 1 and svi._eager_update
            assert next(iter(svi._graphs.values())).eager_params is not None
            with pyro.settings.context(svi_capture_steps=False):
                svi = build(examples.logreg_model)
            for _ in range(5):
                svi.step(X, y)
            assert not svi.hip_graph and not svi._graphs
            svi = build(examples.logreg_model, hip_graph=False)
            assert not svi.hip_graph and not svi._auto_graph
    finally:
        pyro.enable_validation(True)
        pyro.clear_param_store()


@pytest.mark.parametrize("fused", [False, True])
def test_hierarchical_logreg_matches_reference(gpu, monkeypatch, fused):
    """Config 5 (toy size) against the reference's loss/gradients: the grouped GLM kernel runs in
    f32 on f32 copies of the f64 golden inputs (rtol 2e-4), the gather formulation in f64."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hier.npz"))
    if fused:
        models.run_hier(g, gpu, monkeypatch, fused=True, dtype=torch.float32, rtol=2e-4)
    else:
        models.run_hier(g, gpu, monkeypatch, fused=False, dtype=torch.float64, rtol=1e-9)


def test_hierarchical_logreg_full_size_properties(gpu):
    """Config 5 shape per GPU at reduced N (2e6 rows, G=1000, D=32, P=64): run-to-run bitwise
    determinism of the grouped kernel and agreement with the per-group flat kernel on a few
    groups."""
    from pyro_amd import examples, kernels as k
    N, D, G, P = 2_000_000, 32, 1000, 64
    X, y, off = examples.synthetic_hier_logreg_data(N, D, G, gpu)
    segs = k.GroupSegments(off, gpu)
    w = torch.randn((P, G, D), device=gpu) * 0.2
    b = torch.randn((P,), device=gpu)
    # first sighting: the kernel that splits X on the fly; from the second on the plane image
    a0 = k.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, None, 1.0, segs)
    a1 = k.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, None, 1.0, segs)
    a2 = k.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, None, 1.0, segs)
    assert segs._planes[4] is not None
    for u, v in zip(a1, a2):
        assert torch.equal(u, v)
    torch.testing.assert_close(a0[0], a1[0], rtol=2e-5, atol=1e-2)
    torch.testing.assert_close(a0[1], a1[1], rtol=1e-4, atol=1e-2)
    k.glm_set_planes_mode(k.GLM_PLANES_OFF)
    try:
        b1 = k.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, None, 1.0, segs)
        b2 = k.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, None, 1.0, segs)
    finally:
        k.glm_set_planes_mode(k.GLM_PLANES_AUTO)
    for u, v in zip(b1, b2):
        assert torch.equal(u, v)
    ll_sum = torch.zeros(P, device=gpu, dtype=torch.float64)
    for grp in (0, 17, 999):
        lo, hi = int(off[grp]), int(off[grp + 1])
        l, gw, gb = k.glm_bernoulli_fwd_bwd(X[lo:hi].contiguous(), y[lo:hi].contiguous(),
                                            w[:, grp].contiguous(), b, None, 1.0)
        torch.testing.assert_close(a1[1][:, grp], gw, rtol=1e-4, atol=1e-2)
    # ll is additive over groups
    for grp in range(0, G, 1):
        pass
    lo_all = torch.zeros(P, device=gpu, dtype=torch.float64)
    step = 100
    for g0 in range(0, G, step):
        lo, hi = int(off[g0]), int(off[min(g0 + step, G)])
        sub_off = off[g0:g0 + step + 1] - off[g0]
        s2 = k.GroupSegments(sub_off, gpu)
        l, _, _ = k.glm_bernoulli_grouped_fwd_bwd(X[lo:hi].contiguous(), y[lo:hi].contiguous(),
                                                  w[:, g0:g0 + step].contiguous(), b, None, 1.0, s2)
        lo_all += l.double()
    torch.testing.assert_close(a1[0].double(), lo_all, rtol=2e-5, atol=1e-2)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-6)])
def test_flat_adam_semantics_and_checkpoint(gpu, dtype, tol):
    """tests/optim_cases.py on the HIP update kernel: parameters not passed do not move, a late
    parameter counts its own steps, get_state / set_state resumes exactly."""
    from tests import optim_cases
    optim_cases.run_semantics(gpu, dtype, tol)


def test_reference_model_text_reaches_the_fused_kernel(gpu, monkeypatch):
    """SURVEY 8(d)'s model verbatim (``logits = w @ X.t(); logits.squeeze(-2); Bernoulli(logits=logits
    + b)``): the latent's matmul with the constant design matrix is deferred (ops/lazy.py) and the
    observed site runs pa_glm_bernoulli*_fwd_bwd; loss and gradients equal the materialised route."""
    import pyro_amd as pyro
    from pyro_amd import kernels
    from pyro_amd.infer import Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    from pyro_amd.ops import lazy

    g = np.random.default_rng(0)
    N, D, P = 6000, 32, 40
    X = torch.as_tensor(g.standard_normal((N, D)).astype(np.float32), device=gpu)
    y = torch.as_tensor((g.uniform(size=N) < 0.5).astype(np.float32), device=gpu)
    calls = []
    real = kernels.glm_bernoulli_fwd_bwd
    monkeypatch.setattr(kernels, "glm_bernoulli_fwd_bwd",
                        lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    out = {}
    for mode in (True, False):
        monkeypatch.setitem(lazy.ENABLED, "on", mode)
        pyro.clear_param_store()
        pyro.set_rng_seed(11)
        guide = AutoNormal(models.logreg_model, init_scale=0.1)
        elbo = Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
        n0 = len(calls)
        loss = elbo.loss_and_grads(models.logreg_model, guide, X, y)
        out[mode] = (loss, models.store_grads(), len(calls) - n0)
    assert out[True][2] == 1 and out[False][2] == 0          # fused kernel only on the lazy route
    assert abs(out[True][0] - out[False][0]) < 2e-5 * abs(out[False][0])
    for k in out[False][1]:
        np.testing.assert_allclose(out[True][1][k], out[False][1][k], rtol=2e-3, atol=2e-3)
    pyro.clear_param_store()


def test_deferred_matmul_behaves_as_the_product_everywhere_else(gpu):
    """Anything but squeeze / + bias / Bernoulli(logits=...) evaluates the deferred product: model
    text that uses the logits differently runs exactly as written."""
    from pyro_amd.ops import lazy
    g = torch.Generator(device="cpu").manual_seed(0)
    X = torch.randn((500, 8), generator=g).to(gpu)
    w = torch.randn((3, 1, 8), generator=g).to(gpu)
    b = torch.randn((3, 1), generator=g).to(gpu)
    wl = lazy.as_latent(w)
    assert isinstance(wl, lazy.LatentTensor) and type(wl * 2.0) is torch.Tensor
    d = wl @ X.t()
    ref = w @ X.t()
    assert isinstance(d, lazy.DeferredMatmul) and d.shape == ref.shape and d.dim() == 3
    torch.testing.assert_close(torch.sigmoid(d), torch.sigmoid(ref))
    torch.testing.assert_close(d * 2.0, ref * 2.0)
    torch.testing.assert_close(d - 1.0, ref - 1.0)
    torch.testing.assert_close(d[1], ref[1])
    torch.testing.assert_close(d.sum(-1), ref.sum(-1))
    torch.testing.assert_close(d.squeeze(0), ref.squeeze(0))            # not the plate dim: evaluated
    s = d.squeeze(-2)
    assert isinstance(s, lazy.DeferredMatmul) and s.shape == (3, 500)
    sb = s + lazy.as_latent(b)
    assert isinstance(sb, lazy.DeferredMatmul) and sb.as_linear_logits() is not None
    torch.testing.assert_close(sb.materialize(), ref.squeeze(-2) + b)
    torch.testing.assert_close(s + torch.ones((500,), device=gpu), ref.squeeze(-2) + 1.0)   # not a bias
    assert type(wl @ torch.randn((8, 8), device=gpu)) is torch.Tensor    # small matrix: not deferred
    x1 = lazy.as_latent(torch.randn((8,), generator=g).to(gpu))
    d1 = X @ x1
    assert isinstance(d1, lazy.DeferredMatmul) and d1.shape == (500,)
    torch.testing.assert_close(d1.materialize(), X @ x1.as_subclass(torch.Tensor))


def test_north_star_acceptance_at_the_north_star_config(gpu):
    """BASELINE.json's acceptance criterion at ITS config: Bayesian logistic regression, plate
    N = 1e6, D = 32, 64 vectorised particles -- 100 Adam steps of the HIP path (SURVEY 8d's model
    text verbatim, lazy matmul -> plane-image GLM kernel from the second step on) against the
    torch-CPU port of the reference step (oracle/ref_port_torch.py, pinned on the reference's
    golden vectors) fed the SAME standard-normal bank: posterior means within 1e-4 relative."""
    import pyro_amd as pyro
    from pyro_amd import examples, rng
    from oracle.ref_port_torch import LogRegAutoNormalPort
    N, D, P, steps, lr = 1_000_000, 32, 64, 100, 0.05
    g = np.random.default_rng(0)
    X = g.standard_normal((N, D), dtype=np.float32)
    w_true = g.standard_normal(D)
    y = (g.uniform(size=N) < 1 / (1 + np.exp(-X @ w_true))).astype(np.float32)
    bank = [(g.standard_normal((P, 1, D)), g.standard_normal((P, 1))) for _ in range(steps)]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # the port's best count (bench.py sweep)
    try:
        port = LogRegAutoNormalPort(torch.tensor(X), torch.tensor(y), P, lr=lr)
        for ew, eb in bank:
            port.loss_and_grads(torch.tensor(ew, dtype=torch.float32), torch.tensor(eb, dtype=torch.float32))
            for o in port.optims:
                o.step()
            for p in port.params:
                p.grad = None
    finally:
        torch.set_num_threads(threads)
    Xt, yt = torch.as_tensor(X, device=gpu), torch.as_tensor(y, device=gpu)
    pyro.clear_param_store()
    pyro.enable_validation(False)
    try:
        guide = pyro.infer.autoguide.AutoNormal(examples.logreg_model, init_scale=0.1)
        svi = pyro.infer.SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": lr}),
                             pyro.infer.Trace_ELBO(num_particles=P, vectorize_particles=True,
                                                   max_plate_nesting=1),
                             hip_graph=False)    # (injected draws: host code of every step)
        guide._setup_prototype(Xt, yt)
        flat = [e for pair in bank for e in pair]
        orig = rng.normal
        rng.normal = models.EpsReplay(flat, gpu)
        try:
            for _ in range(steps):
                svi.step(Xt, yt)
        finally:
            rng.normal = orig
    finally:
        pyro.enable_validation(True)
    from pyro_amd import kernels
    assert kernels.glm_planes_of(Xt.t().t()) is not None          # the plane-image kernel ran
    store = pyro.get_param_store()
    for name, ref in (("AutoNormal.locs.w", port.loc_w), ("AutoNormal.locs.b", port.loc_b)):
        mean = store[name].detach().cpu().numpy().reshape(-1)
        ref = ref.detach().numpy().reshape(-1)
        rel = np.abs(mean - ref).max() / np.abs(ref).max()
        assert rel < 1e-4, (name, rel)
    # posterior scales too (same criterion)
    sw = store["AutoNormal.scales.w"].detach().cpu().numpy()
    ref_sw = torch.nn.functional.softplus(port.rho_w).detach().numpy()
    assert np.abs(sw - ref_sw).max() / np.abs(ref_sw).max() < 1e-3
    pyro.clear_param_store()


def test_rccl_all_reduce_eager_and_captured_in_the_step_graph(gpu):
    """RCCL itself, at world size 1 (all a one-GPU box offers): the flat-gradient all-reduce as an
    eager launch between two graphs (the default multi-rank step) and captured inside ONE hipGraph
    (PYRO_AMD_GRAPH_COLLECTIVE=1) give the losses of the plain single-process step.  Runs in a
    subprocess: a process group is global state."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RCCL_ONE_RANK_ROWS="200000", RCCL_ONE_RANK_STEPS="20",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_one_rank.py")], cwd=root,
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "RCCL one-rank OK" in out.stdout
    assert "two graphs + eager collective" in out.stdout and out.stdout.count("one graph") >= 2


def test_torch_manual_seed_alone_reproduces_a_run(gpu):
    """The Philox stream of the kernels is owned by torch's default generator (the reference's draws all go
    through it, pyro/util.py:37-45): torch.manual_seed(s) -- without pyro.set_rng_seed -- restarts it, for
    eager steps and for a captured step (whose draws hold the seed as a launch constant: re-captured)."""
    import pyro_amd as pyro
    from pyro_amd import examples, rng
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(4096, 8, gpu, seed=1)

    def run(seed_fn, graph):
        pyro.clear_param_store()
        seed_fn()
        guide = AutoNormal(examples.logreg_model, init_scale=0.1)
        svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
                  Trace_ELBO(num_particles=4, vectorize_particles=True, max_plate_nesting=1), hip_graph=graph)
        a = [svi.step(X, y) for _ in range(7)]
        # the same parameters, the stream restarted in the middle of the run
        seed_fn()
        b = [svi.step(X, y) for _ in range(3)]
        seed_fn()
        c = [svi.step(X, y) for _ in range(3)]
        return a, b, c, svi

    pyro.enable_validation(False)
    try:
        for graph in (False, True):
            a1, b1, c1, _ = run(lambda: pyro.set_rng_seed(11), graph)
            a2, b2, c2, svi = run(lambda: torch.manual_seed(11), graph)
            assert a1 == a2 and b1 == b2 and c1 == c2
            assert rng.current_seed() == 11
            a3 = run(lambda: torch.manual_seed(12), graph)[0]
            assert a3 != a1
        torch.manual_seed(3)
        u = rng.normal((8,), torch.float32, gpu)
        v = rng.normal((8,), torch.float32, gpu)
        torch.manual_seed(3)
        assert torch.equal(rng.normal((8,), torch.float32, gpu), u) and not torch.equal(u, v)
    finally:
        pyro.enable_validation(True)
        pyro.clear_param_store()


def test_replay_enqueued_ahead_notices_tensors_the_model_closes_over(gpu):
    """SVI(prearm=True) enqueues its next replay ahead of the host.  It may only run if nothing it reads has
    changed since: not just step()'s arguments and the parameters, every tensor the captured step reads from
    outside itself -- here the prior scale the model closes over, rewritten in place between two steps.  The
    losses equal the eager run's bit for bit, before and after the change."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=2)
    prior_scale = torch.ones((), device=gpu)
    zeros = torch.zeros(32, device=gpu)

    def model(X, y):
        w = pyro.sample("w", dist.Normal(zeros, prior_scale).to_event(1))
        b = pyro.sample("b", dist.Normal(zeros[0], prior_scale))
        with pyro.plate("data", X.shape[0]):
            pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)

    runs = []
    pyro.enable_validation(False)
    try:
        for graph in (False, None):
            pyro.clear_param_store()
            pyro.set_rng_seed(3)
            prior_scale.fill_(1.0)
            svi = SVI(model, AutoNormal(model, init_scale=0.1), pyro.optim.Adam({"lr": 0.02}),
                      Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=graph,
                      prearm=graph is None)
            losses = [svi.step(X, y) for _ in range(9)]
            if graph is None:
                entry = next(iter(svi._graphs.values()))
                assert entry.gate is not None and entry.armed, "the step is not pre-armed here"
                assert any(t is prior_scale for t in entry.reads)
            prior_scale.fill_(0.05)                 # (a tight prior: the loss jumps)
            losses += [svi.step(X, y) for _ in range(5)]
            runs.append((losses, {k: v.detach().clone() for k, v in pyro.get_param_store().items()}))
    finally:
        pyro.enable_validation(True)
        pyro.clear_param_store()
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert abs(runs[0][0][9] - runs[0][0][8]) > 1.0
    for k in runs[0][1]:      # (the chained tail rounds this model's prior gradient once differently: 1 ulp)
        torch.testing.assert_close(runs[0][1][k], runs[1][1][k], rtol=1e-6, atol=1e-7)


# ---- what a captured step freezes on the host -------------------------------------------------
def _guard_run(gpu, graph, between=None, steps=14, model=None, seed=5, **kw):
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    model = model or examples.logreg_model
    X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=2)
    pyro.clear_param_store()
    pyro.set_rng_seed(seed)
    svi = SVI(model, AutoNormal(model, init_scale=0.1), pyro.optim.Adam({"lr": 0.02}),
              Trace_ELBO(num_particles=16, vectorize_particles=True, max_plate_nesting=1), hip_graph=graph, **kw)
    losses, seen = [], []
    for k in range(steps):
        losses.append(svi.step(X, y))
        if between is not None:
            seen.append(between(k, svi))
    return losses, seen, svi


def test_default_svi_is_not_prearmed_and_reads_between_steps_see_the_step_just_made(gpu):
    """the bare constructor must not enqueue step k+1 ahead of the caller.  A parameter
    cloned between two steps (asynchronously, on the step's stream) is the parameter after step k -- the
    eager run's, bit for bit; and no captured step of the default SVI has a gate."""
    import pyro_amd as pyro
    pyro.enable_validation(False)
    try:
        def snap(k, svi):
            return pyro.param("AutoNormal.locs.w").detach().clone()       # enqueued, not synchronised
        l0, s0, _ = _guard_run(gpu, False, snap)
        l1, s1, svi = _guard_run(gpu, None, snap)
        assert svi._graphs and not svi.prearm
        assert all(e.gate is None and not e.armed for e in svi._graphs.values())
        assert l0 == l1
        for a, b in zip(s0, s1):
            assert torch.equal(a, b)
    finally:
        pyro.enable_validation(True)


def test_captured_step_is_remade_when_a_host_scalar_of_the_model_changes(gpu):
    """a model that reads `self.beta` -- a Python float the training loop updates every
    few steps (KL annealing) -- trains with the CURRENT value: the capture is dropped when the scalar
    moves, a few eager steps follow, then it is captured anew.  Losses equal the eager run's bit for bit."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist

    class Annealed:
        def __init__(self):
            self.beta = 1.0

        def __call__(self, X, y):
            w = pyro.sample("w", dist.Normal(X.new_zeros(X.shape[1]), 1.0).to_event(1))
            b = pyro.sample("b", dist.Normal(X.new_zeros(()), 1.0))
            with pyro.plate("data", X.shape[0]), pyro.poutine.scale(scale=self.beta):
                pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)

    pyro.enable_validation(False)
    try:
        runs = []
        for graph in (False, None):
            m = Annealed()
            captured = []

            def anneal(k, svi, m=m, captured=captured):
                captured.append(len(svi._graphs))
                if k == 7:
                    m.beta = 0.25
            losses, _, svi = _guard_run(gpu, graph, anneal, steps=16, model=m)
            runs.append(losses)
            if graph is None:
                # captured before the change, dropped by it, captured again after the eager warm-up
                assert captured[6] == 1 and captured[8] == 0 and captured[-1] == 1, captured
        assert runs[0] == runs[1], runs
        assert abs(runs[0][8] - runs[0][7]) > 100.0      # (the likelihood's weight changed: the loss jumps)
    finally:
        pyro.enable_validation(True)


def test_model_that_mutates_its_own_host_state_is_never_captured(gpu):
    """A scalar that moves while the model RUNS (a step counter driving a schedule) cannot be frozen: the
    bare constructor stays eager, silently, and follows the eager trajectory trivially."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist

    class Counting:
        calls = 0

        def __init__(self):
            self.calls = 0

        def __call__(self, X, y):
            self.calls += 1
            w = pyro.sample("w", dist.Normal(X.new_zeros(X.shape[1]), 1.0).to_event(1))
            with pyro.plate("data", X.shape[0]), pyro.poutine.scale(scale=min(1.0, self.calls / 8.0)):
                pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, None)), obs=y)

    pyro.enable_validation(False)
    try:
        l0, _, _ = _guard_run(gpu, False, model=Counting(), steps=10)
        l1, _, svi = _guard_run(gpu, None, model=Counting(), steps=10)
        assert not svi._graphs and not svi.hip_graph and svi._self_mutating
        assert l0 == l1
    finally:
        pyro.enable_validation(True)


def test_captured_step_is_dropped_when_the_param_store_is_cleared(gpu):
    """pyro.clear_param_store() followed by re-initialisation on the SAME SVI object: the captured step holds
    the old leaves by address; the store's generation moved, so it is dropped and the new parameters train
    (equal to a fresh eager run from the same seed)."""
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=2)
    pyro.enable_validation(False)
    try:
        def run(graph, restart):
            pyro.clear_param_store()
            pyro.set_rng_seed(5)
            guide = AutoNormal(examples.logreg_model, init_scale=0.1)
            optim = pyro.optim.Adam({"lr": 0.02})
            svi = SVI(examples.logreg_model, guide, optim,
                      Trace_ELBO(num_particles=16, vectorize_particles=True, max_plate_nesting=1), hip_graph=graph)
            out = [svi.step(X, y) for _ in range(8)]
            if restart:
                pyro.clear_param_store()
                pyro.set_rng_seed(5)
                guide = AutoNormal(examples.logreg_model, init_scale=0.1)
                svi.guide, svi.optim = guide, pyro.optim.Adam({"lr": 0.02})
                out = [svi.step(X, y) for _ in range(8)]
            return out, {k: v.detach().clone() for k, v in pyro.get_param_store().items()}
        l0, p0 = run(False, False)
        l1, p1 = run(None, True)
        assert l0 == l1
        for k in p0:
            assert torch.equal(p0[k], p1[k]), k
    finally:
        pyro.enable_validation(True)


def test_self_capturing_svi_says_so_once(gpu):
    import warnings

    from pyro_amd.infer import svi as svi_mod
    svi_mod._WARNED_CAPTURE[0] = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _guard_run(gpu, None, steps=6)
        _guard_run(gpu, None, steps=6)
        _guard_run(gpu, True, steps=6)
    said = [x for x in w if issubclass(x.category, svi_mod.CapturedStepWarning)]
    assert len(said) == 1 and "hip_graph=False" in str(said[0].message)


@pytest.mark.parametrize("which", ["sgd_momentum", "torch_adam", "rmsprop", "callable"])
def test_torch_optimizer_behind_a_captured_loss_follows_the_eager_trajectory(gpu, which):
    """An optimizer whose update cannot sit in a graph -- `PyroOptim(torch.optim.X)`: host-side state per call; a
    plain callable -- no longer sends the whole step back to the handlers: the step's loss and gradients are ONE
    graph replay (the reference's `loss_and_grads`, pyro/infer/svi.py:144-150), the update and the gradient
    zeroing (:153-156) run eagerly behind it.  Losses equal the eager run's bit for bit, parameters to a gradient's last bit."""
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=4)
    seen = []

    def make():
        if which == "sgd_momentum":
            return pyro.optim.PyroOptim(torch.optim.SGD, {"lr": 1e-5, "momentum": 0.9})
        if which == "torch_adam":
            return pyro.optim.PyroOptim(torch.optim.Adam, {"lr": 0.02})
        if which == "rmsprop":
            return pyro.optim.RMSprop({"lr": 0.01})

        def plain(params, *a, **k):                    # a user's own update: p -= 1e-6 g
            with torch.no_grad():
                for p in sorted(params, key=lambda t: t.numel()):
                    seen.append(float(p.grad.abs().sum()))
                    p.add_(p.grad, alpha=-1e-6)
        return plain

    pyro.enable_validation(False)
    try:
        runs = []
        for graph in (False, None):
            pyro.clear_param_store()
            pyro.set_rng_seed(9)
            del seen[:]
            svi = SVI(examples.logreg_model, AutoNormal(examples.logreg_model, init_scale=0.1), make(),
                      Trace_ELBO(num_particles=16, vectorize_particles=True, max_plate_nesting=1), hip_graph=graph)
            losses = [svi.step(X, y) for _ in range(12)]
            if graph is None:
                assert svi.hip_graph and len(svi._graphs) == 1 and svi._eager_update
            # gradients are zero between steps, as after the reference's zero_grads (pyro/infer/util.py:85-91)
            for name, p in pyro.get_param_store().named_parameters():
                assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name
            runs.append((losses, {k: v.detach().clone() for k, v in pyro.get_param_store().items()}, list(seen)))
        assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
        for k in runs[0][1]:      # (the chained tail rounds the prior's gradient once differently: 1 ulp of a gradient)
            torch.testing.assert_close(runs[0][1][k], runs[1][1][k], rtol=1e-6, atol=1e-7)
        assert len(runs[0][2]) == len(runs[1][2])
        for a, b in zip(sorted(runs[0][2]), sorted(runs[1][2])):
            assert abs(a - b) <= 1e-5 * max(abs(a), 1.0)
        assert runs[0][0][-1] < runs[0][0][0] or which in ("sgd_momentum", "callable")
    finally:
        pyro.enable_validation(True)
        pyro.clear_param_store()


def test_short_captured_step_is_replayed_as_its_kernels(gpu):
    """csrc/replay.hip (opt-in: kernels.DIRECT_REPLAY): the captured config-2 step is a chain of two kernel nodes;
    switched on, it is launched as those two kernels (pa_graph_direct_launch) instead of through hipGraphLaunch.
    Same device work: the losses and the parameters of 15 steps equal, bit for bit, those of the run that
    replays the graph."""
    import pyro_amd as pyro
    from pyro_amd import examples, kernels
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(50_000, 32, gpu, seed=0)

    def run(direct):
        prev = kernels.DIRECT_REPLAY["on"]
        kernels.DIRECT_REPLAY["on"] = direct
        try:
            pyro.clear_param_store(); pyro.set_rng_seed(3)
            guide = AutoNormal(examples.logreg_model, init_scale=0.1)
            svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
                      Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                losses = [svi.step(X, y) for _ in range(15)]
            entries = list(svi._graphs.values())
            assert len(entries) == 1
            plan = entries[0].direct
            params = {k: v.detach().clone() for k, v in pyro.get_param_store().items()}
            svi.release()
            return losses, params, plan
        finally:
            kernels.DIRECT_REPLAY["on"] = prev

    run(False)        # (the first step of a process on new data builds the plane image of X: a different launch)
    la, pa_, plan = run(True)
    assert plan is not None and plan.n_nodes == 2, "the headline step is expected to be two kernel nodes"
    lb, pb, none = run(False)
    assert none is None
    assert la == lb
    for k in pa_:
        assert torch.equal(pa_[k], pb[k]), k


def test_direct_replay_plan_only_for_short_kernel_chains(gpu):
    """pa_graph_direct_plan: a chain of three kernels gets a plan whose launch does what graph.replay() does; a
    graph of more nodes than DIRECT_REPLAY["max_nodes"], and one holding a memset node, get none."""
    from pyro_amd import kernels

    prev = kernels.DIRECT_REPLAY["on"]
    kernels.DIRECT_REPLAY["on"] = True

    def capture(n_ops, memset=False):
        x = torch.zeros(1024, device=gpu)
        g = kernels.new_graph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            if memset:
                x.zero_()
            for _ in range(n_ops):
                x.add_(1.0)
        return g, x

    g, x = capture(3)
    plan = kernels.graph_direct_plan(g)
    assert plan is not None and plan.n_nodes == 3
    plan.launch(); plan.launch()
    g.replay()
    torch.cuda.synchronize()
    assert float(x[0]) == 9.0 and float(x[-1]) == 9.0
    g2, _ = capture(kernels.DIRECT_REPLAY["max_nodes"] + 2)
    assert kernels.graph_direct_plan(g2) is None
    g3, x3 = capture(1, memset=True)
    p3 = kernels.graph_direct_plan(g3)
    if p3 is not None:                       # (a runtime that captures zero_() as a kernel: still a chain)
        p3.launch()
    else:
        g3.replay()
    torch.cuda.synchronize()
    kernels.DIRECT_REPLAY["on"] = prev
    assert float(x3[0]) == 1.0


def test_no_update_optimizer_leaves_parameters_and_zeroes_gradients(gpu):
    """pyro_amd.optim.NoUpdate: SVI.step computes the loss and the gradients, hands the loss over, zeroes the
    gradients; the parameters stay bit for bit -- eagerly and as a captured step (which ends in the fused tail of
    the full step: two graph nodes), and both give the losses of the full step's FIRST evaluation sequence with
    frozen parameters (the same draws: the Philox stream advances as in any step)."""
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(20_000, 32, gpu, seed=0)

    def run(graph):
        pyro.clear_param_store(); pyro.set_rng_seed(7)
        guide = AutoNormal(examples.logreg_model, init_scale=0.1)
        svi = SVI(examples.logreg_model, guide, pyro.optim.NoUpdate(),
                  Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=graph)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            first = svi.step(X, y)
            before = {k: v.detach().clone() for k, v in pyro.get_param_store().items()}
            losses = [first] + [svi.step(X, y) for _ in range(9)]
        after = {k: v.detach().clone() for k, v in pyro.get_param_store().items()}
        grads = [p.grad for p in pyro.get_param_store()._params.values() if p.grad is not None]
        captured = len(svi._graphs)
        svi.release()
        return losses, before, after, grads, captured

    le, be, ae, ge, ce = run(False)
    lg, bg, ag, gg, cg = run(True)
    assert ce == 0 and cg == 1
    for before, after in ((be, ae), (bg, ag)):
        for k in before:
            assert torch.equal(before[k], after[k]), k
    for g in ge + gg:
        assert float(g.abs().max()) == 0.0
    assert len(set(le)) > 1                      # new draws every step
    np.testing.assert_allclose(lg, le, rtol=2e-6)
