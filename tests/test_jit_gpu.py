"""JitTrace_ELBO on the device (reference: pyro/infer/trace_elbo.py:162-257 over pyro/ops/jit.py:48-163):
the traced ``differentiable_loss`` holds the step's kernels as ``pyro_amd::*`` dispatcher ops
(pyro_amd/ops/torch_library.py), replays at new parameter values with the reference's numbers
(tests/golden/logreg_f32.npz, written by the unmodified reference), and its Philox draws continue the
stream the eager estimator would have consumed."""
import os

import numpy as np
import pytest
import torch

from tests import models

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(autouse=True)
def _clean():
    import pyro_amd
    pyro_amd.clear_param_store()
    yield
    torch.set_default_dtype(torch.float32)
    pyro_amd.clear_param_store()


@pytest.mark.parametrize("fused", [True, False])
def test_traced_config2_loss_replays_with_the_reference_numbers(gpu, monkeypatch, fused):
    g = np.load(os.path.join(G, "logreg_f32.npz"))
    # (the guide draw is not the fused one here: the recorded noise of the fixture is fed through
    # rng.normal; test_traced_draws_continue_the_philox_stream covers pyro_amd::fn_meanfield_normal_sample)
    ops = ("pyro_amd::glm_bernoulli", "pyro_amd::fn_multi_log_prob_sum") if fused else ()
    graph = models.run_logreg_jit(g, gpu, monkeypatch, fused=fused, dtype=torch.float32, rtol=2e-4,
                                  expect_ops=ops)
    recorded = {ln.split("= ")[1].split("(")[0] for ln in graph.split("\n") if "= pyro_amd::" in ln}
    assert recorded, "no kernel of this package was recorded as a graph node"
    # (what is not pyro_amd:: is ATen glue; nothing may be a Python call-back)
    assert "PythonOp" not in graph


def test_traced_draws_continue_the_philox_stream(gpu):
    """Replays draw fresh noise: the k-th traced evaluation equals the eager estimator's evaluation at
    the same position of the Philox stream (the first call consumes two evaluations' worth: the eager
    parameter-discovery run and the run under the tracer)."""
    import pyro_amd as pyro
    from pyro_amd.infer import JitTrace_ELBO, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    N, D, P = 4096, 16, 64
    gen = torch.Generator().manual_seed(0)
    X = torch.randn((N, D), generator=gen).to(gpu)
    y = (torch.rand((N,), generator=gen) < 0.5).float().to(gpu)

    def run(cls, n):
        pyro.clear_param_store()
        pyro.set_rng_seed(7)
        guide = AutoNormal(models.logreg_model_fused, init_scale=0.1)
        guide._setup_prototype(X, y)
        pyro.set_rng_seed(7)
        kw = dict(ignore_jit_warnings=True) if cls is JitTrace_ELBO else {}
        elbo = cls(num_particles=P, vectorize_particles=True, max_plate_nesting=1, **kw)
        out = [elbo.loss(models.logreg_model_fused, guide, X, y) for _ in range(n)]
        if cls is JitTrace_ELBO:
            (compiled,) = [c for c, _, _ in elbo._jit_cache.values()]
            (traced,) = compiled.compiled.values()
            graph = str(traced.graph)
            for op in ("pyro_amd::fn_meanfield_normal_sample", "pyro_amd::glm_bernoulli",
                       "pyro_amd::fn_multi_log_prob_sum"):
                assert op in graph, (op, graph)
        return out

    eager = run(Trace_ELBO, 6)
    assert len(set(eager)) == 6
    traced = run(JitTrace_ELBO, 4)
    np.testing.assert_allclose(traced, eager[2:], rtol=1e-6)


def test_svi_with_a_traced_loss_follows_the_eager_trajectory(gpu):
    """SVI over JitTrace_ELBO (the way the reference's examples use it, examples/svi_horovod.py:101 /
    --jit flags): same parameters after a few Adam steps as SVI over Trace_ELBO started two draws
    later in the stream."""
    import pyro_amd as pyro
    from pyro_amd import rng
    from pyro_amd.infer import SVI, JitTrace_ELBO, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    N, D, P = 2048, 8, 64
    gen = torch.Generator().manual_seed(1)
    X = torch.randn((N, D), generator=gen).to(gpu)
    y = (torch.rand((N,), generator=gen) < 0.5).float().to(gpu)
    model = models.logreg_model_fused

    def run(cls, skip):
        pyro.clear_param_store()
        pyro.set_rng_seed(3)
        guide = AutoNormal(model, init_scale=0.1)
        guide._setup_prototype(X, y)
        pyro.set_rng_seed(3)
        kw = dict(ignore_jit_warnings=True) if cls is JitTrace_ELBO else {}
        elbo = cls(num_particles=P, vectorize_particles=True, max_plate_nesting=1, **kw)
        if skip:
            with torch.no_grad():
                for _ in range(skip):
                    Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1).loss(
                        model, guide, X, y)
        svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.05}), loss=elbo, hip_graph=False)
        losses = [svi.step(X, y) for _ in range(5)]
        return losses, {k: v.detach().cpu().numpy().copy() for k, v in pyro.get_param_store().items()}

    l_eager, p_eager = run(Trace_ELBO, 2)
    l_jit, p_jit = run(JitTrace_ELBO, 0)
    np.testing.assert_allclose(l_jit, l_eager, rtol=2e-6)
    for name in p_eager:
        np.testing.assert_allclose(p_jit[name], p_eager[name], rtol=1e-5, atol=1e-6, err_msg=name)


def test_torch_compile_of_the_config2_loss_over_the_registered_ops(gpu):
    """torch.compile(fullgraph=True) of config 2's differentiable loss written against the registered
    operators (the guide draw, the GLM site, the one-launch sums of the small sites): dynamo records them as
    pyro_amd::* graph nodes, their shape functions answer without a prior eager call, AOT autograd
    differentiates through the registered formulas; loss and gradients equal the eager evaluation with the
    same draws.  (The Python effect handlers themselves are not dynamo's to trace; ``ops.jit.trace`` /
    ``JitTrace_ELBO`` record the full loss by executing it, see the tests above.)"""
    import pyro_amd as pyro
    from pyro_amd import _lib, examples, rng
    from pyro_amd.distributions import fused
    from pyro_amd.ops import torch_library as tl

    N, D, P = 4096, 32, 64
    X, y = examples.synthetic_logreg_data(N, D, gpu, seed=5)
    from pyro_amd import kernels
    kernels.glm_bernoulli_fwd_bwd(X, y, torch.zeros(P, D, device=gpu), None, None, 1.0)   # (second sighting:
    kernels.glm_bernoulli_fwd_bwd(X, y, torch.zeros(P, D, device=gpu), None, None, 1.0)   #  the plane image)
    loc_w = torch.zeros(D, device=gpu, requires_grad=True)
    rho_w = torch.full((D,), -2.0, device=gpu, requires_grad=True)
    loc_b = torch.zeros(1, device=gpu, requires_grad=True)
    rho_b = torch.full((1,), -2.0, device=gpu, requires_grad=True)
    zero = torch.zeros((), device=gpu)
    one = torch.ones((), device=gpu)
    # the plane image of X and the label moments are facts about the DATA, looked up on the host (by tensor
    # identity): taken outside the compiled function, they enter its graph as constants
    planes, moments = kernels.glm_planes_of(X), kernels.glm_label_moments_of(X, y)
    fmt = kernels._format_of(planes)
    assert planes is not None and tl.available()

    def loss_fn(loc_w, rho_w, loc_b, rho_b, X, y):
        (zw, sw, lw), (zb, sb, lb) = fused.meanfield_sample([loc_w, loc_b], [rho_w, rho_b], P)
        ll = torch.ops.pyro_amd.glm_bernoulli_planes(planes, y, zw, zb.reshape(P), 1.0, N, D, fmt, moments)[0]
        prior = fused.log_prob_sum(_lib.DIST_NORMAL, zw, zero, one) + fused.log_prob_sum(_lib.DIST_NORMAL, zb, zero, one)
        entropy = fused.log_prob_sum(_lib.DIST_NORMAL, zw, lw, sw) + fused.log_prob_sum(_lib.DIST_NORMAL, zb, lb, sb)
        return -(ll.sum() + prior - entropy) / P

    from pyro_amd.ops.jit import _ReplaySafeDraws
    leaves = (loc_w, rho_w, loc_b, rho_b)
    with tl.routing():
        # the draws are addressed relative to a device word the wrapper sets before each call (what
        # ops.jit.trace does for a traced loss): a compiled call draws what the eager code would draw then.
        # One eager call first: it enters the call signatures in the table (a compiler replays such side
        # effects only after it has finished recording)
        draws = _ReplaySafeDraws(gpu)
        pyro.set_rng_seed(21)
        ref = draws.run(lambda: loss_fn(*leaves, X, y), first=True)
        ref_g = torch.autograd.grad(ref, leaves)
        pyro.set_rng_seed(22)
        ref2 = draws.run(lambda: loss_fn(*leaves, X, y), first=False)
        n_specs = len(tl._SPECS)
        compiled = torch.compile(loss_fn, fullgraph=True, backend="aot_eager")
        pyro.set_rng_seed(21)
        got = draws.run(lambda: compiled(*leaves, X, y), first=False)
        got_g = torch.autograd.grad(got, leaves)
        pyro.set_rng_seed(22)
        got2 = draws.run(lambda: compiled(*leaves, X, y), first=False)
        torch.testing.assert_close(got2, ref2, rtol=1e-6, atol=1e-6)
        assert abs(float(ref2) - float(ref)) > 1e-4
        # the Philox seed / offsets are not part of a call signature: more draws, no more signatures
        before = len(tl._SPECS)
        for _ in range(3):
            draws.run(lambda: loss_fn(*leaves, X, y), first=False)
        assert len(tl._SPECS) == before == n_specs
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
    for a, b in zip(got_g, ref_g):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    assert torch.isfinite(ref) and float(ref) > 0
