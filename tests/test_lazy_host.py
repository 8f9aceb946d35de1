"""Host logic of the lazy recognition (ops/lazy.py) on CPU: the predicates that say "device tensor" are
answered with yes (tests/oracle_backend.py), so the recognition LOGIC -- which expressions stay
deferred, what they evaluate to when used any other way -- is tested without a GPU; the kernels behind
the recognised sites are tested in the -m gpu files."""
import numpy as np
import pytest
import torch

from oracle import glm as o_glm


@pytest.fixture(autouse=True)
def _cpu_backend(oracle_backend):
    yield


def test_oracle_group_rows_is_the_stable_sort():
    rng = np.random.default_rng(0)
    for N, G in [(0, 3), (1, 1), (1000, 7), (5000, 1000)]:
        g = rng.integers(0, G, size=N)
        off, rows = o_glm.group_rows(g, G)
        assert off[0] == 0 and off[-1] == N and (np.diff(off) >= 0).all() and off.shape == (G + 1,)
        assert sorted(rows.tolist()) == list(range(N))
        for k in range(G):
            seg = rows[off[k]:off[k + 1]]
            assert (g[seg] == k).all() and (np.diff(seg) > 0).all()       # ascending inside a group
    with pytest.raises(IndexError):
        o_glm.group_rows(np.array([0, 5]), 5)
    with pytest.raises(IndexError):
        o_glm.group_rows(np.array([0, -6]), 5)
    off, rows = o_glm.group_rows(np.array([0, -1, 2, -5, 4]), 5)            # ids in [-G, 0) count from the end
    assert off.tolist() == [0, 2, 2, 3, 3, 5] and rows.tolist() == [0, 3, 2, 1, 4]


def test_group_gather_recognition_and_fallbacks():
    from pyro_amd.ops import lazy
    gen = torch.Generator().manual_seed(0)
    N, D, G, P = 500, 8, 7, 3
    X = torch.randn((N, D), generator=gen)
    ids = torch.randint(0, G, (N,), generator=gen)
    w = torch.randn((P, G, D), generator=gen)
    b = torch.randn((P, 1), generator=gen)
    wl, bl = lazy.as_latent(w), lazy.as_latent(b)
    ref = w[..., ids, :]
    d = wl[..., ids, :]
    assert isinstance(d, lazy.DeferredGroupDot) and d.stage == "gather" and d.shape == ref.shape
    assert isinstance(wl[:, ids], lazy.DeferredGroupDot) and isinstance(wl[:, ids, :], lazy.DeferredGroupDot)
    assert type(wl[..., ids]) is torch.Tensor and type(wl[0]) is torch.Tensor
    assert type(wl[:, ids[:10]]) is torch.Tensor and type(wl[:, ids.to(torch.int32)]) is torch.Tensor
    w2 = lazy.as_latent(torch.randn((G, D), generator=gen))
    assert isinstance(w2[ids], lazy.DeferredGroupDot) and w2[ids].shape == (N, D)
    torch.testing.assert_close(d * 2.0, ref * 2.0)
    torch.testing.assert_close(d - 1.0, ref - 1.0)
    torch.testing.assert_close(torch.tanh(d), torch.tanh(ref))
    p = d * X
    assert isinstance(p, lazy.DeferredGroupDot) and p.stage == "product"
    assert isinstance(X * d, lazy.DeferredGroupDot) and isinstance(torch.mul(d, X), lazy.DeferredGroupDot)
    torch.testing.assert_close(d * X[:, :1], ref * X[:, :1])                 # not the design matrix
    torch.testing.assert_close(p.sum(), (ref * X).sum())
    torch.testing.assert_close(p.sum(-1, keepdim=True), (ref * X).sum(-1, keepdim=True))
    torch.testing.assert_close(p.sum(0), (ref * X).sum(0))
    for s in (p.sum(-1), torch.sum(p, -1), p.sum(dim=-1), p.sum(2)):
        assert isinstance(s, lazy.DeferredGroupDot) and s.stage == "logits" and s.shape == (P, N)
    s = p.sum(-1)
    torch.testing.assert_close(torch.sigmoid(s), torch.sigmoid((ref * X).sum(-1)))
    for sb in (s + bl, bl + s, s + 0.5):
        assert isinstance(sb, lazy.DeferredGroupDot) and sb.stage == "logits"
    torch.testing.assert_close((bl + s).materialize(), (ref * X).sum(-1) + b)
    assert (s + bl).bias is b                                               # the site's own tensor
    torch.testing.assert_close(s + torch.ones(N), (ref * X).sum(-1) + 1.0)   # not a bias: evaluated
    torch.testing.assert_close((s + bl) + 1.0, (ref * X).sum(-1) + b + 1.0)  # a second addend: evaluated
    lz = (s + bl).as_grouped_linear_logits()
    assert lz is not None and lz.shape == (P, N) and lz.segments.ids is ids
    torch.testing.assert_close(lz.materialize(), (ref * X).sum(-1) + b)
    assert d.as_grouped_linear_logits() is None and p.as_grouped_linear_logits() is None


def test_bias_on_the_left_keeps_the_plated_glm_deferred():
    from pyro_amd.ops import lazy
    gen = torch.Generator().manual_seed(1)
    X = torch.randn((400, 8), generator=gen)
    w = lazy.as_latent(torch.randn((3, 1, 8), generator=gen))
    b = lazy.as_latent(torch.randn((3, 1), generator=gen))
    s = (w @ X.t()).squeeze(-2)
    for sb in (s + b, b + s):
        assert isinstance(sb, lazy.DeferredMatmul) and sb.as_linear_logits() is not None


def test_deferred_linear_behaves_like_its_result_everywhere_but_under_a_sigmoid():
    """ops/lazy.py::DeferredLinear (a tall Linear layer not launched yet): every use that is not a sigmoid
    goes on with the plain layer's result, launched once -- operators, comparisons, torch functions, methods,
    attributes (host stand-in for the launch; the fused route is a GPU test)."""
    from pyro_amd.ops import lazy
    d = lazy.DeferredLinear("tall", (), (3, 2), torch.zeros(1))
    ref = torch.arange(6.0).reshape(3, 2)
    d._plain = ref.clone()                       # (what materialize() would have launched)
    assert d.shape == (3, 2) and d.dim() == 2 and d.size(1) == 2 and len(d) == 3 and d.dtype == torch.float32
    assert torch.equal(d == ref, torch.ones(3, 2, dtype=torch.bool)) and hash(d) == hash(d)
    for got, want in ((d + 1, ref + 1), (2 - d, 2 - ref), (d * d, ref * ref), (-d, -ref), (abs(-d), ref),
                      (d / 2, ref / 2), (2 ** d, 2 ** ref), (d[1:], ref[1:]), (d @ ref.t(), ref @ ref.t()),
                      (torch.tanh(d), torch.tanh(ref)), (torch.cat([d, d]), torch.cat([ref, ref])),
                      (d.t(), ref.t()), (d.clamp(min=2.0), ref.clamp(min=2.0)), (d > 2, ref > 2)):
        assert torch.equal(got, want)
    assert float(d[0, 1]) == 1.0 and [row.tolist() for row in d] == ref.tolist()


def test_amortised_guide_layers_take_the_fused_routes(oracle_backend, monkeypatch):
    """examples/lda.py's guide text (histogram by scatter_add, then nn.Sequential(Linear, Sigmoid, Linear,
    Sigmoid, Linear, Sigmoid, Softmax)) on the host with the kernels answered by the oracle: from the second
    sighting of the corpus the first layer is the bag-of-words route, every layer is deferred until its Sigmoid
    arrives and launched WITH it (forward: sigmoid_out, backward: the gradient through it in the operand loads,
    the first layer's bias gradient from its partial sums); values and all parameter gradients equal the dense
    torch route."""
    import torch.nn as nn
    from pyro_amd import kernels as k
    from pyro_amd.ops import lazy
    monkeypatch.setattr(lazy, "TALL_MIN_ROWS", 16)
    gen = torch.Generator().manual_seed(0)
    V, B, Wd = 256, 70, 12
    data = torch.randint(0, V, (Wd, B), generator=gen)
    torch.manual_seed(0)
    predictor = nn.Sequential(nn.Linear(V, 20), nn.Sigmoid(), nn.Linear(20, 9), nn.Sigmoid(), nn.Linear(9, 4),
                              nn.Sigmoid(), nn.Softmax(dim=-1))

    def guide_body():
        counts = torch.zeros(V, B).scatter_add(0, data, torch.ones(data.shape))
        return predictor(counts.transpose(0, 1))

    log = []
    for name in ("bow_linear_fwd", "bow_linear_bwd", "tall_linear", "tall_wgrad"):
        real = getattr(k, name)
        monkeypatch.setattr(k, name, lambda *a, _r=real, _n=name, **kw: (log.append((_n, kw.get("sigmoid", False),
                                                                                  kw.get("y_mul") is not None)), _r(*a, **kw))[1])
    outs, grads = [], []
    for rep in range(3):
        for p in predictor.parameters():
            p.grad = None
        monkeypatch.setitem(lazy.ENABLED, "on", rep > 0)          # (the first run: torch's dense route)
        with lazy.watch_histograms():
            y = guide_body()
        (y * torch.arange(4.0)).sum().backward()
        outs.append(y.detach().clone())
        grads.append([p.grad.clone() for p in predictor.parameters()])
    for o, gr in zip(outs[1:], grads[1:]):
        torch.testing.assert_close(torch.as_tensor(o), torch.as_tensor(outs[0]), rtol=1e-5, atol=1e-6)
        for a, b in zip(gr, grads[0]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-9)
    names = [n for n, _, _ in log]
    assert names.count("bow_linear_fwd") == 2 and names.count("bow_linear_bwd") == 2 and \
        names.count("tall_linear") == 2 * (2 + 2) and names.count("tall_wgrad") == 2 * 2
    assert all(sig for n, sig, _ in log if n in ("bow_linear_fwd",))                      # launched with the Sigmoid
    assert all(sig or ym for n, sig, ym in log if n == "tall_linear")
    assert all(ym for n, _, ym in log if n in ("tall_wgrad", "bow_linear_bwd"))
