"""The run-time generated element-wise kernels (pyro_amd/ops/fuser.py over pa_rtc_compile / pa_rtc_launch)
against the ATen operators they replace: same programs eagerly and under the fuser, forward values and
gradients; memory hazards (views, in-place writes); inside a captured graph; and what they buy a captured
step (launch counts of BASELINE configs 1 and 4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _program(dtype, dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(7, 5, dtype=dtype, generator=g).to(dev).requires_grad_(True)
    w = torch.randn(5, dtype=dtype, generator=g).to(dev).requires_grad_(True)
    m = (torch.rand(7, 5, generator=g) > 0.5).to(dev)
    y = (x * w + 2.0).exp().clamp(min=1e-3, max=50.0)
    p = y / y.sum(-1, keepdim=True)
    z = torch.where(m, p.log(), torch.zeros((), dtype=dtype, device=dev)) * 3.0 - torch.sigmoid(x) ** 2
    q = z.sum(0) + w.abs().sqrt().sum()
    loss = (q * torch.ones(5, dtype=dtype, device=dev)).sum() + (x.t().contiguous() ** 3).sum()
    loss.backward()
    acc = torch.zeros(7, 5, dtype=dtype, device=dev)
    acc.add_(x.detach(), alpha=0.5).mul_(2.0).clamp_(min=-1.0)
    acc[2:4].zero_()
    acc[:, 1].fill_(3.0)
    b = (acc > 0) & (acc < 2.0) | torch.isnan(acc)
    return [loss.detach(), x.grad, w.grad, acc, b, b.to(dtype).sum(), p.detach(), z.detach()]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fused_program_equals_the_operators_it_replaces(gpu, dtype):
    """Forward values and gradients (the duals run on the autograd thread: recorded there too); everything
    that does not pass through a reduction is BIT-identical (same libm calls, no contraction), sums agree to
    rounding of a different summation order."""
    from pyro_amd.ops import fuser
    ref = _program(dtype, gpu)
    before = dict(fuser.STATS)
    with fuser.Fuser():
        got = _program(dtype, gpu)
    torch.cuda.synchronize()
    d = {k: fuser.STATS[k] - before[k] for k in fuser.STATS}
    assert d["recorded"] >= 60 and 0 < d["kernels"] < d["recorded"] // 2, d
    tol = dict(rtol=2e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-13, atol=1e-13)
    for a, b in zip(got, ref):
        assert a.dtype == b.dtype and a.shape == b.shape
        if a.dtype == torch.bool:
            assert torch.equal(a, b)
        else:
            torch.testing.assert_close(a, b, **tol)
    assert torch.equal(got[3], ref[3]) and torch.equal(got[4], ref[4])        # element-wise only: bitwise


def test_elementwise_chain_is_bitwise_and_one_kernel(gpu):
    from pyro_amd.ops import fuser

    def run(x, y):
        t = torch.sigmoid(x * y + 0.25) / (y.abs() + 1.5)
        u = torch.log1p(t.clamp(min=1e-6)) - torch.tanh(x).pow(2)
        return torch.where(u > 0.1, u * 2.0, -u), u.neg().exp().reciprocal()

    g = torch.Generator().manual_seed(1)
    x = torch.randn(300, 17, generator=g).to(gpu)
    y = torch.randn(17, generator=g).to(gpu)
    ref = run(x, y)
    before = fuser.STATS["kernels"]
    with fuser.Fuser():
        got = run(x, y)
    assert fuser.STATS["kernels"] - before == 1
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_views_and_in_place_writes_keep_their_order(gpu):
    """Reads through a DIFFERENT view of memory a recorded operator writes, writes under a recorded read, a
    transposed in-place target, an expanded operand, a reduction between element-wise runs."""
    from pyro_amd.ops import fuser

    def run(dev):
        a = torch.arange(24.0, device=dev).reshape(4, 6)
        b = a * 2.0                       # recorded
        c = b[:, ::2] + 1.0               # reads a strided VIEW of a recorded output
        b.t().mul_(0.5)                   # in-place through a transposed view, after c read it
        d = b.sum(1, keepdim=True)        # reduction of the updated values
        e = (b - d) / (c.sum() + 1.0)     # full reduction, 0-dim operand
        a.add_(e)                         # writes the tensor the first operator read
        f = a.unsqueeze(0).expand(3, 4, 6) * torch.ones(3, 1, 1, device=dev)
        b[1].copy_(f[2, 3])
        return a, b, c, d, e, f.sum((0, 2))

    ref = run(gpu)
    with fuser.Fuser():
        got = run(gpu)
    torch.cuda.synchronize()
    for x, y in zip(got, ref):
        torch.testing.assert_close(x, y, rtol=1e-6, atol=1e-6)


def test_short_reductions_inside_elementwise_kernels(gpu):
    """A reduction over <= 32 elements is a loop of the thread that owns the output element, in one kernel
    with what follows it -- but never in one kernel with the operator that PRODUCES what it sums (it reads
    a range, not its own element) nor with one that overwrites it."""
    from pyro_amd.ops import fuser

    def run(x, w):
        pe = -(x * w).sum(-1)                       # [C]: product, short row sum, neg
        total = pe.sum()                            # reads ALL of pe
        g = (pe - total / 6.0).unsqueeze(-1) * x    # back to [C, D]
        col = g.sum(0, keepdim=True)                # short column sum, keepdim
        x2 = x.clone()
        s = x2.sum(-1, keepdim=True)
        x2.sub_(s)                                  # overwrites what the row sum read
        return pe, total, g, col, x2, (g / (col + 7.0)).sum((0, 1))

    gen = torch.Generator().manual_seed(4)
    x = torch.randn(6, 8, generator=gen, dtype=torch.float64).to(gpu)
    w = torch.randn(8, generator=gen, dtype=torch.float64).to(gpu)
    ref = run(x, w)
    before = dict(fuser.STATS)
    with fuser.Fuser():
        got = run(x, w)
    torch.cuda.synchronize()
    assert fuser.STATS["kernels"] - before["kernels"] <= 8
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-13, atol=1e-13)


def test_fused_kernels_inside_a_captured_graph_follow_their_inputs(gpu):
    from pyro_amd.ops import fuser
    x = torch.randn(64, 8, device=gpu)
    w = torch.randn(8, device=gpu, requires_grad=True)

    def step():
        w.grad = None
        loss = ((x * w).sigmoid().log().sum(1) * 0.5).sum() + (w ** 2).sum()
        loss.backward()
        return loss.detach(), w.grad

    # warm-up and capture on ONE side stream (torch's recipe for a backward inside a capture: the
    # AccumulateGrad node of ``w`` must not have been created on another stream)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with fuser.Fuser():
            step()                               # (generates and compiles the kernels)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        with fuser.Fuser():
            loss, grad = step()
    for seed in (1, 2):
        x.copy_(torch.randn(64, 8, generator=torch.Generator().manual_seed(seed)).to(gpu))
        graph.replay()
        torch.cuda.synchronize()
        got = (loss.clone(), grad.clone())
        ref = step()
        torch.testing.assert_close(got[0], ref[0], rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(got[1], ref[1], rtol=2e-5, atol=1e-6)


def test_unknown_operators_and_host_tensors_pass_through(gpu):
    from pyro_amd.ops import fuser
    x = torch.randn(16, 16, device=gpu)
    with fuser.Fuser():
        a = (x + 1.0) @ (x * 2.0)                   # matmul: not fused, its inputs materialised first
        b = torch.softmax(a.exp().clamp(max=10.0), -1)
        c = (torch.ones(3) * 2.0 + 1.0).sum()       # host tensors: untouched
        i = (x > 0).long().sum()                    # integer results: not fused
        v = float((b.sum() + 1.0).item())           # a host read materialises what it needs
    ref = torch.softmax(((x + 1.0) @ (x * 2.0)).exp().clamp(max=10.0), -1)
    torch.testing.assert_close(b, ref, rtol=1e-6, atol=1e-7)
    assert float(c) == 9.0 and int(i) == int((x > 0).sum()) and abs(v - 17.0) < 1e-4


def test_captured_steps_launch_fewer_kernels(gpu):
    """BASELINE configs[0] (eight schools) and configs[3] (LDA, toy size): the captured step with the fuser
    records the same trajectory (to rounding of the sums) as with it switched off, from fewer launches."""
    import pyro_amd as pyro
    from pyro_amd.ops import fuser
    from tools import bench_configs as bc

    out = {}
    for on in (False, True):
        fuser.ENABLED["on"] = on
        try:
            before = dict(fuser.STATS)
            pyro.set_rng_seed(5)
            r1 = bc.config1(gpu, steps=30)
            pyro.set_rng_seed(5)
            r4 = bc.config4(gpu, docs=2000, steps=6)
            out[on] = (r1, r4, {k: fuser.STATS[k] - before[k] for k in fuser.STATS})
        finally:
            fuser.ENABLED["on"] = True
    assert out[False][2]["recorded"] == 0 and out[True][2]["recorded"] > 50
    assert out[True][2]["kernels"] < out[True][2]["recorded"]
    for j in (0, 1):
        a, b = out[True][j], out[False][j]
        assert a["graphed"] and b["graphed"]
        np.testing.assert_allclose(a["last_loss"], b["last_loss"], rtol=2e-4)
