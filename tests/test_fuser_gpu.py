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


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_enumeration_indexing_and_its_backward(gpu, dtype):
    """``table[enumerated_values]`` (one leading int64 index) and the accumulate=True index_put that is its
    dual -- duplicates, negative entries, values that broadcast -- element-wise only: bitwise gathers, sums in
    the operator's own order."""
    from pyro_amd.ops import fuser
    g = torch.Generator().manual_seed(3)
    t0 = torch.randn(5, 3, 4, dtype=dtype, generator=g).to(gpu)
    idx = torch.tensor([[4], [0], [-1], [2], [0], [0]], device=gpu)               # [6, 1]: duplicates + negative
    w0 = torch.randn(6, 1, 3, 4, dtype=dtype, generator=g).to(gpu)

    def run():
        table = t0.clone().requires_grad_(True)
        out = table[idx] * 1.5
        (out * w0).sum().backward()
        base = torch.zeros(5, 4, dtype=dtype, device=gpu)
        base.index_put_((torch.tensor([1, 1, 3], device=gpu),), torch.ones(4, dtype=dtype, device=gpu) * 0.25,
                        accumulate=True)
        fn = torch.index_put(t0[:, 0], (torch.tensor([[2, 2], [0, 4]], device=gpu),), w0[:4, 0, 0].reshape(2, 2, 4),
                             accumulate=True)
        return out.detach(), table.grad, base, fn

    ref = run()
    fuser.UNFUSED.clear()
    before = dict(fuser.STATS)
    with fuser.Fuser():
        got = run()
    torch.cuda.synchronize()
    assert not any(k.startswith("aten::index") or k.startswith("aten::_index") for k in fuser.UNFUSED), fuser.UNFUSED
    assert fuser.STATS["kernels"] - before["kernels"] <= 6
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
    tol = dict(rtol=2e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-13, atol=1e-13)
    torch.testing.assert_close(got[1], ref[1], **tol)
    torch.testing.assert_close(got[3], ref[3], **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("dim", [-1, 0, 1])
def test_softmax_family_over_a_short_dim(gpu, dtype, dim):
    from pyro_amd.ops import fuser
    g = torch.Generator().manual_seed(4)
    x0 = (torch.randn(9, 7, 8, dtype=dtype, generator=g) * 3).to(gpu)
    w0 = torch.randn(9, 7, 8, dtype=dtype, generator=g).to(gpu)

    def run():
        x = x0.clone().requires_grad_(True)
        y = x.transpose(0, 1)                                  # (strided operand)
        a = torch.softmax(y, dim)
        b = torch.log_softmax(x * 0.5, dim)
        ((a.transpose(0, 1) * w0).sum() + (b * w0 * w0).sum()).backward()
        return a.detach(), b.detach(), x.grad

    ref = run()
    fuser.UNFUSED.clear()
    with fuser.Fuser():
        got = run()
    torch.cuda.synchronize()
    assert not any("softmax" in k for k in fuser.UNFUSED), fuser.UNFUSED
    tol = dict(rtol=3e-6, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-13, atol=1e-13)
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, **tol)


def test_independent_kernels_share_a_launch(gpu):
    """Kernels of different iteration domains without a hazard between them are ONE launch (guarded bodies);
    dependent ones stay ordered.  Same numbers with the merge switched off."""
    from pyro_amd.ops import fuser
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(n, generator=g).to(gpu) for n in (5, 300, 7000, 64)]

    def run():
        ys = [(x * 2.0 + 1.0).exp() for x in xs]              # four domains, independent
        z = ys[1].sum() + ys[0].sum()                         # reductions: the next level
        return ys + [z * xs[3]]

    outs = {}
    for on in (False, True):
        fuser.MERGE_LEVELS["on"] = on
        try:
            before = fuser.STATS["kernels"]
            with fuser.Fuser():
                outs[on] = run()
            torch.cuda.synchronize()
            outs[on, "n"] = fuser.STATS["kernels"] - before
        finally:
            fuser.MERGE_LEVELS["on"] = True
    assert outs[True, "n"] < outs[False, "n"] and outs[True, "n"] <= 4, (outs[True, "n"], outs[False, "n"])
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)
    ref = run()
    for a, b in zip(outs[True][:4], ref[:4]):
        assert torch.equal(a, b)
    torch.testing.assert_close(outs[True][4], ref[4], rtol=1e-5, atol=1e-5)


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


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_family_nodes_are_the_site_kernels_arithmetic(gpu, dtype):
    """A family's log-density / partial derivatives recorded under a scope (csrc/dist_fam.h compiled into the
    generated source) against pa_dist_log_prob / pa_dist_log_prob_grad: the same expressions (the library
    contracts a*b+c into fused multiply-adds, the generated source does not: agreement to an ulp or two)."""
    from pyro_amd import _lib as L
    from pyro_amd.distributions import fused
    from pyro_amd.ops import fuser
    g = torch.Generator().manual_seed(7)
    cases = [(L.DIST_NORMAL, torch.randn(6, 40, 9, generator=g), torch.randn(40, 1, generator=g),
              torch.rand(9, generator=g) + 0.5),
             (L.DIST_BERNOULLI_LOGITS, (torch.rand(6, 40, 9, generator=g) > 0.5).float(),
              torch.randn(6, 1, 9, generator=g) * 3, None),
             (L.DIST_GAMMA, torch.rand(6, 40, 9, generator=g) + 0.1, torch.rand(40, 9, generator=g) + 0.5,
              torch.rand(1, generator=g) + 0.5),
             (L.DIST_BETA, torch.rand(6, 40, 9, generator=g) * 0.9 + 0.05, torch.rand(9, generator=g) + 0.5,
              torch.rand(40, 1, generator=g) + 0.5),
             (L.DIST_HALF_CAUCHY, torch.rand(6, 40, 9, generator=g) * 4, torch.rand(9, generator=g) + 0.5, None)]
    for dist_id, v, a, b in cases:
        v, a = v.to(dtype).to(gpu), a.to(dtype).to(gpu)
        b = None if b is None else b.to(dtype).to(gpu)

        def run():
            ps = [a.clone().requires_grad_(True)] + ([] if b is None else [b.clone().requires_grad_(True)])
            lp = fused.log_prob(dist_id, v, ps[0], ps[1] if len(ps) > 1 else None)
            (lp * 0.5).sum().backward()
            return [lp.detach()] + [p.grad for p in ps]

        ref = run()
        before = fuser.STATS["recorded"]
        with fuser.Fuser():
            got = run()
        torch.cuda.synchronize()
        assert fuser.STATS["recorded"] - before >= 3
        tight = dict(rtol=2e-6, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-14, atol=1e-14)
        torch.testing.assert_close(got[0], ref[0], **tight)
        tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)
        for x, y in zip(got[1:], ref[1:]):
            torch.testing.assert_close(x, y, **tol)


def test_joins_integer_comparisons_and_layouts(gpu):
    """stack / cat as copies into slices (the producers' own stores are dropped), dot of short vectors,
    comparisons of integer tensors with scalars that travel as launch arguments, transposed results."""
    from pyro_amd.ops import fuser
    g = torch.Generator().manual_seed(8)
    x = torch.randn(7, 5, generator=g).to(gpu)
    w = torch.randn(5, generator=g).to(gpu)
    lengths = torch.randint(1, 9, (7,), generator=g).to(gpu)

    def run():
        parts = [torch.where((t < lengths).unsqueeze(-1), x * float(t + 3), x.new_zeros(())) for t in range(5)]
        st = torch.stack(parts).permute(1, 0, 2).contiguous() + torch.cat(parts, -1).sum()
        d = torch.dot(w, w * 2.0) + (x.t() / (x.t().abs() + 1.0)).sum()
        tr = x.t() * 3.0 - x.t().exp()                      # (a transposed result, as the operators give it)
        sel = x.clone().requires_grad_(True)                # the duals of x[:, k] / x[k]: one node over the full shape
        ((sel[:, 2] * 2.0).sum() + sel[1].exp().sum() + sel.sum().sum()).backward()
        return st, d, tr, (lengths >= 4) | (lengths == 1), sel.grad

    ref = run()
    fuser.UNFUSED.clear()
    before = fuser.STATS["kernels"]
    with fuser.Fuser():
        got = run()
    torch.cuda.synchronize()
    assert not fuser.UNFUSED, fuser.UNFUSED
    assert fuser.STATS["kernels"] - before <= 16
    assert got[2].stride() == ref[2].stride() and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
    torch.testing.assert_close(got[0], ref[0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[1], ref[1], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[4], ref[4], rtol=1e-6, atol=1e-6)


def test_markov_loop_is_batched_across_its_time_steps(gpu):
    """examples/hmm.py model_1 (toy size) under pyro.markov: the sites of ALL time steps are recorded before
    anything is launched -- the captured step holds a few dozen launches, not a few per time step -- and the
    trajectory is the one of the step with the fuser switched off."""
    import pyro_amd as pyro
    from pyro_amd.ops import fuser
    from tools import bench_configs as bc

    out = {}
    for on in (False, True):
        fuser.ENABLED["on"] = on
        try:
            before = dict(fuser.STATS)
            pyro.set_rng_seed(3)
            out[on] = (bc.config_hmm(gpu, S=20, L=24, K=5, D=11, steps=4, graph=True),
                       {k: fuser.STATS[k] - before[k] for k in fuser.STATS})
        finally:
            fuser.ENABLED["on"] = True
    a, b = out[True][0], out[False][0]
    assert a["graphed"] and b["graphed"]
    np.testing.assert_allclose(a["loss_first"], b["loss_first"], rtol=2e-5)
    np.testing.assert_allclose(a["loss_last"], b["loss_last"], rtol=2e-4)
    d = out[True][1]
    # two passes (the eager pre-pass and the capture) of 24 time steps: far fewer launches than time steps x sites
    assert d["recorded"] > 24 * 40 and d["kernels"] <= 2 * 60, d


def test_long_sums_in_two_recorded_stages(gpu):
    """Sums longer than a lane group takes (1e5 documents down to a scalar / to a parameter's shape): a view
    [.., S, C, ..] of the contiguous operand, C elements per group, then the S partial sums -- recorded, so
    that they share launches with their neighbours."""
    from pyro_amd.ops import fuser
    g = torch.Generator().manual_seed(9)
    x = torch.randn(100000, 8, generator=g).to(gpu)
    y = torch.randn(100000, generator=g).to(gpu)

    def run():
        return (x * 2.0).sum(0), y.sum(), y.exp().sum(), x.sum(), x.view(10, 10000, 8).sum(1)

    ref = run()
    fuser.UNFUSED.clear()
    with fuser.Fuser():
        got = run()
    torch.cuda.synchronize()
    assert not fuser.UNFUSED, fuser.UNFUSED
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-3)


_CACHE_PROBE = r"""
import json, sys
import torch
sys.path.insert(0, %r)
import pyro_amd as pyro
from pyro_amd.ops import fuser
from tools import bench_configs
dev = torch.device("cuda", 0)
r = bench_configs.config1(dev, steps=5)          # eight schools: 34 operators per step -> generated kernels
print("STATS " + json.dumps({"stats": fuser.STATS, "loss": r["last_loss"], "graphed": r["graphed"]}))
"""


def test_second_process_compiles_nothing(gpu, tmp_path):
    """Persistent cache of generated kernels (~/.cache/pyro_amd/rtc/<sha256>.hsaco, here a temporary
    directory): the first process compiles every kernel of a captured eight-schools step with hiprtc and
    writes the code objects; a second process loads them (hipModuleLoadData) and runs the same step --
    same loss, captured -- without a single hiprtc call."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYRO_AMD_RTC_CACHE=str(tmp_path / "rtc"))
    outs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, "-c", _CACHE_PROBE % root], env=env, capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("STATS ")][-1]
        outs.append(json.loads(line[6:]))
    first, second = outs
    files = [f for f in os.listdir(tmp_path / "rtc") if f.endswith(".hsaco")]
    assert first["stats"]["compiled"] >= 2 and first["stats"]["from_disk"] == 0
    assert len(files) == first["stats"]["compiled"] and not [f for f in os.listdir(tmp_path / "rtc") if ".tmp." in f]
    assert second["stats"]["compiled"] == 0, second
    assert second["stats"]["from_disk"] == second["stats"]["loaded"] == first["stats"]["loaded"]
    assert second["graphed"] and first["graphed"] and second["loss"] == first["loss"]


def test_capture_parameter_blocks_are_freed_with_the_capture(gpu):
    """a generated-kernel launch made during a capture keeps a ~3 KB parameter block for the
    graph; the blocks belong to the captured step and go when it goes (SVI.release, eviction, re-capture)."""
    import gc

    from pyro_amd.ops import fuser
    from tools import bench_configs
    freed = []
    orig = fuser.RtcBlocks.free

    def counting_free(self):
        if self._scope is not None:
            freed.append(self.count)
        orig(self)
    fuser.RtcBlocks.free = counting_free
    try:
        bench_configs.config1(gpu, steps=3)
        gc.collect()
    finally:
        fuser.RtcBlocks.free = orig
    assert freed and max(freed) >= 2, freed
