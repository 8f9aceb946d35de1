"""pyro_amd.ops.jit.trace / JitTrace_ELBO on the host (kernels answered by the numpy oracle through
tests/oracle_backend.py): the reference's contract for traced functions (pyro/ops/jit.py:48-163,
tests/ops/test_jit.py in the reference) and the traced ELBO against the reference-generated fixture."""
import os

import numpy as np
import pytest
import torch

import pyro_amd as pyro
from tests import models

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(autouse=True)
def _cpu_backend(oracle_backend):
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def test_trace_reads_params_as_graph_inputs():
    """A traced function of pyro.param values follows the parameters (they are inputs of the recorded
    graph, not constants) and is differentiable w.r.t. them."""
    from pyro_amd.ops import jit

    calls = []

    @jit.trace
    def f(x):
        calls.append(1)
        a = pyro.param("a", torch.tensor(2.0))
        b = pyro.param("b", torch.tensor([1.0, 3.0]))
        return (a * x * b).sum()

    x = torch.tensor([1.0, 2.0])
    assert float(f(x)) == pytest.approx(2.0 * (1.0 + 6.0))
    n_python_runs = len(calls)
    with torch.no_grad():
        pyro.get_param_store()._params["a"].fill_(5.0)
    out = f(x)
    assert float(out) == pytest.approx(5.0 * 7.0)
    assert len(calls) == n_python_runs                       # replayed, not re-run
    out.backward()
    np.testing.assert_allclose(pyro.get_param_store()._params["b"].grad.numpy(), [5.0, 10.0])
    # keyword arguments are part of the signature: one trace per distinct set
    assert len(f.compiled) == 1


def test_trace_keyword_arguments_select_the_compiled_graph():
    from pyro_amd.ops import jit

    @jit.trace
    def f(x, scale=1.0):
        return pyro.param("s", torch.tensor(3.0)) * x.sum() * scale

    x = torch.ones(4)
    assert float(f(x, scale=2.0)) == pytest.approx(24.0)
    assert float(f(x, scale=0.5)) == pytest.approx(6.0)
    assert float(f(x, scale=2.0)) == pytest.approx(24.0)
    assert len(f.compiled) == 2


def test_trace_accepts_unhashable_keyword_arguments():
    """(lists / dicts / sets in kwargs are frozen into the signature key, pyro/ops/jit.py:68-77)"""
    from pyro_amd.ops import jit

    @jit.trace
    def f(x, dims=None, opts=None):
        return pyro.param("t", torch.tensor(2.0)) * x.sum(dims) .sum() * opts["k"]

    x = torch.ones(2, 3)
    assert float(f(x, dims=[0], opts={"k": 2.0})) == pytest.approx(24.0)
    assert float(f(x, dims=[0], opts={"k": 2.0})) == pytest.approx(24.0)
    assert len(f.compiled) == 1


@pytest.mark.parametrize("fused", [False, True])
def test_jit_trace_elbo_replays_at_new_parameter_values(monkeypatch, fused):
    g = np.load(os.path.join(G, "logreg_f64.npz"))
    models.run_logreg_jit(g, torch.device("cpu"), monkeypatch, fused=fused, dtype=torch.float64, rtol=1e-9)


@pytest.mark.parametrize("fused", [False, True])
def test_svi_over_a_traced_loss_updates_the_parameters(monkeypatch, fused):
    """SVI collects the parameters of a step from the "param" messages it sees: a replayed graph runs
    no Python, so CompiledFunction reads its leaves through the param primitive on every call.  With
    the same (frozen) noise the traced and the eager estimator give the same Adam trajectory."""
    from pyro_amd import rng
    from pyro_amd.distributions import families
    from pyro_amd.infer import SVI, JitTrace_ELBO, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    if fused:
        monkeypatch.setattr(families._BernoulliLinear, "_allow_f64", True, raising=False)
    N, D, P = 300, 6, 5
    gen = torch.Generator().manual_seed(1)
    X = torch.randn((N, D), generator=gen)
    y = (torch.rand((N,), generator=gen) < 0.5).double()
    model = models.logreg_model_fused if fused else models.logreg_model

    def frozen_noise(shape, dtype, device):
        n = int(np.prod(shape))
        return torch.sin(torch.arange(1, n + 1, dtype=dtype) * 0.7).reshape(tuple(shape))
    monkeypatch.setattr(rng, "normal", frozen_noise)

    def run(cls):
        pyro.clear_param_store()
        guide = AutoNormal(model, init_scale=0.1)
        guide._setup_prototype(X, y)
        kw = dict(ignore_jit_warnings=True) if cls is JitTrace_ELBO else {}
        elbo = cls(num_particles=P, vectorize_particles=True, max_plate_nesting=1, **kw)
        svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.05}), loss=elbo)
        losses = [svi.step(X, y) for _ in range(6)]
        return losses, {k: v.detach().numpy().copy() for k, v in pyro.get_param_store().items()}

    l_eager, p_eager = run(Trace_ELBO)
    l_jit, p_jit = run(JitTrace_ELBO)
    assert len(set(l_jit)) == 6
    np.testing.assert_allclose(l_jit, l_eager, rtol=1e-10)
    for name in p_eager:
        np.testing.assert_allclose(p_jit[name], p_eager[name], rtol=1e-9, atol=1e-12, err_msg=name)


def test_a_registered_function_compiles_with_fullgraph_on_the_host():
    """ops/torch_library.dispatcher_op under torch.compile(fullgraph=True, backend="aot_eager"): the call site's
    ``invoke`` is inlined by dynamo and hands the call to the dispatcher op; the shape function answers for
    a signature it has never run (one evaluation on zeros), the backward's shape function too; a volatile
    integer argument (a Philox seed) is not part of the signature."""
    import torch
    from pyro_amd.ops import torch_library as tl

    @tl.dispatcher_op("host_probe_scale")
    class _Probe(torch.autograd.Function):
        volatile_args = (1,)

        @staticmethod
        def forward(ctx, x, seed, k):
            ctx.k = k
            ctx.save_for_backward(x)
            return x * k + (seed % 7), x.sum()

        @staticmethod
        def backward(ctx, g, gs):
            (x,) = ctx.saved_tensors
            return g * ctx.k + gs, None, None

    def f(x, seed):
        a, s = _Probe.invoke(x, seed, 3.0)
        return (a * a).sum() + s

    x = torch.randn(5, requires_grad=True)
    with tl.routing():
        n0 = len(tl._SPECS)
        compiled = torch.compile(f, fullgraph=True, backend="aot_eager")
        got = compiled(x, 12345678901234)                     # (no eager call before: nothing knows the shapes)
        g_got, = torch.autograd.grad(got, x)
        ref = f(x, 12345678901234)
        g_ref, = torch.autograd.grad(ref, x)
        assert len(tl._SPECS) == n0 + 1
        f(x, 99)                                              # another seed: the same signature
        f(x, (1 << 64) - 3)                                   # (unsigned 64-bit values survive the int64 tensor)
        assert len(tl._SPECS) == n0 + 1
    torch.testing.assert_close(got, ref)
    torch.testing.assert_close(g_got, g_ref)
    assert float(f(x, 8)) != float(f(x, 9))                   # eager apply, not routed: seed % 7 differs
