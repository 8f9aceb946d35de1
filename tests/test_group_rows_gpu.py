"""SURVEY 8(d) config 5 in the reference's own formulation: UNSORTED int64 group ids, logits by an
advanced-index gather.  Parity of (i) the sort permutation (pa_group_rows_build: integer work, bit for
bit against oracle/glm.py::group_rows = numpy's stable argsort), (ii) the plane image packed through
it (bit for bit the oracle's image of X[rows]), (iii) the grouped kernel on that image against the
float64 restatement that gathers with the unsorted ids, (iv) the model text verbatim against the
unmodified reference's loss and gradients (tests/golden/hier_unsorted.npz), and (v) at the BASELINE
size N = 1e7: float64 oracle comparisons on slices covering the first / last tiles and group seams."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import glm as o_glm
from tests import models


def _k():
    from pyro_amd import kernels
    return kernels


def tt(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


@pytest.mark.parametrize("N,G", [(0, 3), (1, 1), (63, 5), (64, 5), (65, 2), (1000, 1000), (100_003, 17),
                                 (1_000_000, 1000), (300_000, 16384)])
def test_group_rows_bit_exact(gpu, N, G):
    k = _k()
    rng = np.random.default_rng(N + G)
    g = rng.integers(0, G, size=N)
    if G > 2:
        g[g == 1] = 2                         # an empty group
    off, rows = k.group_rows_build(tt(g.astype(np.int64), gpu), G)
    r_off, r_rows = o_glm.group_rows(g, G)
    assert np.array_equal(off.cpu().numpy(), r_off)
    assert np.array_equal(rows.cpu().numpy(), r_rows)


def test_group_rows_rejects_ids_out_of_range_like_torch(gpu):
    k = _k()
    g = torch.tensor([0, 3, 1, 7, 2], device=gpu)
    with pytest.raises(IndexError):
        k.group_rows_build(g, 5)
    with pytest.raises(IndexError):
        k.group_rows_build(torch.tensor([0, -6, 2], device=gpu), 5)
    # ids in [-G, 0) count from the end, as w[..., g, :] reads them
    off, rows = k.group_rows_build(torch.tensor([0, -1, 2, -5, 4], device=gpu), 5)
    assert off.tolist() == [0, 2, 2, 3, 3, 5] and rows.tolist() == [0, 3, 2, 1, 4]
    with pytest.raises(k.Unsupported):
        k.group_rows_build(torch.zeros(10, dtype=torch.int64, device=gpu), 16385)


@pytest.fixture(params=["f16x2", "bf16x3"])
def planes_fmt(request):
    k = _k()
    before = k.glm_planes_format()
    k.glm_set_planes_format(k.GLM_PLANES_F16X2 if request.param == "f16x2" else k.GLM_PLANES_BF16X3)
    yield request.param
    k.glm_set_planes_format(before)


@pytest.mark.parametrize("N,D,P,G", [(5000, 32, 64, 7), (70_000, 17, 40, 50), (700, 8, 130, 3), (300, 32, 2, 9)])
def test_grouped_image_and_kernel_from_unsorted_rows(gpu, planes_fmt, N, D, P, G):
    k = _k()
    rng = np.random.default_rng(N + D + P + G)
    g = rng.integers(0, G, size=N)
    g[g == G - 1] = 0                         # an empty group
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, G, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32)
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    tX, ty, tg = tt(X, gpu), tt(y, gpu), tt(g.astype(np.int64), gpu)
    segs = k.grouped_rows_of(tg, G)
    assert segs is k.grouped_rows_of(tg, G) and segs.rows is not None and segs.ids is tg
    r_off, r_rows = o_glm.group_rows(g, G)
    assert np.array_equal(segs.rows.cpu().numpy(), r_rows) and np.array_equal(segs.group_offsets, r_off)
    planes = k.glm_pack_planes_grouped(tX, ty, segs)
    got = planes.cpu().numpy()
    seg = segs.seg.cpu().numpy()
    if planes_fmt == "f16x2":
        img, ypad, kx = o_glm.glm_grouped_plane_image_f16(X[r_rows], y[r_rows], seg)
    else:
        img, ypad = o_glm.glm_grouped_plane_image(X[r_rows], y[r_rows], seg)
    nb = img.size * 2
    assert np.array_equal(got[:nb].view(np.uint16).reshape(img.shape), img)          # bit-exact
    assert np.array_equal(got[nb:nb + ypad.size * 4].view(np.float32), ypad)
    # first sighting already takes the image (no other kernel serves rows in their original order)
    ll, gw, gb = k.glm_bernoulli_grouped_fwd_bwd(tX, ty, tt(w, gpu), tt(b, gpu), None, 2.0, segs)
    assert segs._planes[4] is not None
    rll, rgw, rgb = o_glm.glm_bernoulli_grouped_fwd_bwd(X, y, w, g, b, None, 2.0)     # gathers with g
    sc = max(1.0, float(np.abs(rll).max()))
    np.testing.assert_allclose(ll.cpu().numpy(), rll, rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(gb.cpu().numpy(), rgb, rtol=2e-5, atol=2e-5 * N ** 0.5)
    np.testing.assert_allclose(gw.cpu().numpy(), rgw, rtol=2e-5, atol=2e-5 * N ** 0.5)
    # a masked call cannot be served: reported, not mis-served
    assert not k.glm_grouped_rows_servable(tX, ty, torch.ones(N, dtype=torch.bool, device=gpu), segs)
    with pytest.raises(k.Unsupported):
        k.glm_bernoulli_grouped_fwd_bwd(tX, ty, tt(w, gpu), tt(b, gpu),
                                        torch.ones(N, dtype=torch.bool, device=gpu), 2.0, segs)
    # new ids in the same tensor object: a new partition
    tg.copy_(torch.flip(tg, [0]))
    segs2 = k.grouped_rows_of(tg, G)
    assert segs2 is not segs
    assert np.array_equal(segs2.rows.cpu().numpy(), o_glm.group_rows(g[::-1], G)[1])


@pytest.mark.parametrize("fused", [True, False])
def test_hierarchical_reference_text_matches_the_reference(gpu, monkeypatch, fused):
    """examples.hier_logreg_model_reference (the gather formulation, nothing backend-specific in the
    text) against the unmodified reference's loss and gradients: through the grouped plane-image kernel
    (float32) and with the recognition off (float64, operator by operator)."""
    from pyro_amd import kernels
    from pyro_amd.ops import lazy
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hier_unsorted.npz"))
    calls = []
    real = kernels.glm_bernoulli_grouped_planes_fwd_bwd
    monkeypatch.setattr(kernels, "glm_bernoulli_grouped_planes_fwd_bwd",
                        lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    if fused:
        models.run_hier_unsorted(g, gpu, monkeypatch, dtype=torch.float32, rtol=2e-4)
        assert len(calls) == 1
    else:
        monkeypatch.setitem(lazy.ENABLED, "on", False)
        models.run_hier_unsorted(g, gpu, monkeypatch, dtype=torch.float64, rtol=1e-9)
        assert len(calls) == 0


def test_deferred_group_dot_behaves_as_the_expression_everywhere_else(gpu):
    from pyro_amd.ops import lazy
    gen = torch.Generator(device="cpu").manual_seed(0)
    N, D, G, P = 500, 8, 7, 3
    X = torch.randn((N, D), generator=gen).to(gpu)
    ids = torch.randint(0, G, (N,), generator=gen).to(gpu)
    w = torch.randn((P, G, D), generator=gen).to(gpu)
    b = torch.randn((P, 1), generator=gen).to(gpu)
    wl, bl = lazy.as_latent(w), lazy.as_latent(b)
    d = wl[..., ids, :]
    ref = w[..., ids, :]
    assert isinstance(d, lazy.DeferredGroupDot) and d.stage == "gather" and d.shape == ref.shape
    assert isinstance(wl[:, ids], lazy.DeferredGroupDot) and type(wl[..., ids]) is torch.Tensor
    assert type(wl[0]) is torch.Tensor and type(wl[:, ids[:10]]) is torch.Tensor        # short: evaluated
    torch.testing.assert_close(d * 2.0, ref * 2.0)
    torch.testing.assert_close(d - 1.0, ref - 1.0)
    torch.testing.assert_close(torch.tanh(d), torch.tanh(ref))
    p = d * X
    assert isinstance(p, lazy.DeferredGroupDot) and p.stage == "product"
    assert isinstance(X * d, lazy.DeferredGroupDot)
    torch.testing.assert_close(p.sum(), (ref * X).sum())
    torch.testing.assert_close(p.sum(-1, keepdim=True), (ref * X).sum(-1, keepdim=True))
    torch.testing.assert_close(p.sum(0), (ref * X).sum(0))
    s = p.sum(-1)
    assert isinstance(s, lazy.DeferredGroupDot) and s.stage == "logits" and s.shape == (P, N)
    assert isinstance(torch.sum(p, -1), lazy.DeferredGroupDot) and isinstance(p.sum(dim=-1), lazy.DeferredGroupDot)
    torch.testing.assert_close(torch.sigmoid(s), torch.sigmoid((ref * X).sum(-1)))
    for sb in (s + bl, bl + s):
        assert isinstance(sb, lazy.DeferredGroupDot) and sb.bias is b
        torch.testing.assert_close(sb.materialize(), (ref * X).sum(-1) + b)
        assert sb.as_grouped_linear_logits() is not None
    torch.testing.assert_close(s + torch.ones((N,), device=gpu), (ref * X).sum(-1) + 1.0)   # not a bias
    # ids out of range: torch's IndexError, from the recognised route too
    bad = ids.clone()
    bad[3] = G
    with pytest.raises(IndexError):
        ((wl[..., bad, :] * X).sum(-1) + bl).as_grouped_linear_logits()


def test_config5_full_size_against_the_oracle_on_slices(gpu):
    """BASELINE config 5 on one GPU's share (N = 1e7 rows, D = 32, G = 1000 unsorted groups, 64
    particles), the reference formulation end to end: integer work bit-exact at the full size
    (permutation, offsets), and the kernel's per-group gradients / log-likelihood against the float64
    oracle on the groups holding the data's first and last rows, the image's first and last tiles
    and two interior seams -- the oracle gathers those groups' rows with the UNSORTED ids."""
    from pyro_amd import examples
    k = _k()
    N, D, G, P = 10_000_000, 32, 1000, 64
    X, y, g = examples.synthetic_hier_logreg_data_unsorted(N, D, G, gpu)
    segs = k.grouped_rows_of(g, G)
    gh = g.cpu().numpy()
    r_off, r_rows = o_glm.group_rows(gh, G)
    assert np.array_equal(segs.group_offsets, r_off)
    assert np.array_equal(segs.rows.cpu().numpy(), r_rows)
    gen = torch.Generator(device=gpu).manual_seed(1)
    w = 0.3 * torch.randn((P, G, D), device=gpu, generator=gen)
    b = 0.1 * torch.randn((P,), device=gpu, generator=gen)
    ll, gw, gb = k.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, None, 1.0, segs)
    ll2, gw2, gb2 = k.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, None, 1.0, segs)
    assert torch.equal(ll, ll2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)     # fixed-order sums
    groups = sorted({0, G - 1, int(gh[0]), int(gh[-1]), 333, 334})
    wh, bh = w.cpu().numpy(), b.cpu().numpy()
    for grp in groups:
        rows = r_rows[r_off[grp]:r_off[grp + 1]]
        Xs, ys = X[torch.as_tensor(rows, device=gpu)].cpu().numpy(), y[torch.as_tensor(rows, device=gpu)].cpu().numpy()
        _, rgw, _ = o_glm.glm_bernoulli_grouped_fwd_bwd(Xs, ys, wh[:, grp:grp + 1], np.zeros(len(rows), dtype=np.int64),
                                                        bh, None, 1.0)
        np.testing.assert_allclose(gw[:, grp].cpu().numpy(), rgw[:, 0], rtol=2e-4, atol=2e-4 * len(rows) ** 0.5)
    # the log-likelihood and bias gradient over ALL rows: float64 on the device in chunks (the oracle's
    # formula, torch float64 as the calculator: 6.4e8 logits)
    ll_ref = torch.zeros(P, dtype=torch.float64, device=gpu)
    gb_ref = torch.zeros(P, dtype=torch.float64, device=gpu)
    step = 1 << 18
    w64 = w.double()
    for lo in range(0, N, step):
        hi = min(N, lo + step)
        lg = torch.einsum("pnd,nd->pn", w64[:, g[lo:hi]], X[lo:hi].double()) + b.double()[:, None]
        yy = y[lo:hi].double()[None]
        ll_ref += (yy * lg - torch.nn.functional.softplus(lg)).sum(1)
        gb_ref += (yy - torch.sigmoid(lg)).sum(1)
    np.testing.assert_allclose(ll.cpu().numpy(), ll_ref.cpu().numpy(), rtol=1e-4)       # f32 1e-4
    np.testing.assert_allclose(gb.cpu().numpy(), gb_ref.cpu().numpy(), rtol=1e-4, atol=1e-4 * N ** 0.5)
