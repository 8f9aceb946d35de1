"""Parity of every HIP kernel (called through the C-ABI) against the numpy oracle.

Tolerances: integer work (Philox bits, enumerated indices) bit-exact; float32 kernels vs the
float64 oracle rtol 2e-5 on reduced sums / 1e-5 element-wise; float64 kernels rtol 1e-11.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import adam as o_adam
from oracle import dists as o_dists
from oracle import glm as o_glm
from oracle import integrator as o_int
from oracle import lda as o_lda
from oracle import nuts as o_nuts
from oracle import philox as o_philox


def _k():
    from pyro_amd import kernels
    return kernels


def tt(a, dev, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=dev)
    return t.to(dtype) if dtype is not None else t


# ---------------------------------------------------------------------------------------------
# RNG
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 3, 4, 5, 1000, 100003])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_philox_uniform_bit_exact(gpu, n, dtype):
    k = _k()
    out = k.philox_uniform((n,), dtype, gpu, seed=0x1234567890ABCDEF, offset=77).cpu().numpy()
    ref = o_philox.uniform(n, out.dtype, 0x1234567890ABCDEF, 77)
    assert np.array_equal(out, ref)  # uniforms are exact functions of the integer stream


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float64, 1e-13)])
def test_philox_normal(gpu, dtype, tol):
    k = _k()
    n = 200001
    out = k.philox_normal((n,), dtype, gpu, seed=42, offset=5).cpu().numpy()
    ref = o_philox.normal(n, out.dtype, 42, 5)
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * 10)
    assert abs(out.mean()) < 0.01 and abs(out.std() - 1) < 0.01


# ---------------------------------------------------------------------------------------------
# element-wise site kernels
# ---------------------------------------------------------------------------------------------
def _dist_inputs(dist_id, rows, cols, rng, bcast):
    shape_v = (rows, cols)
    shape_a = (1, cols) if bcast == "row" else ((rows, 1) if bcast == "col" else (rows, cols))
    if dist_id == 0:
        return rng.standard_normal(shape_v), rng.standard_normal(shape_a), rng.uniform(0.5, 2, shape_a)
    if dist_id == 1:
        return (rng.uniform(size=shape_v) < 0.4).astype(float), 4 * rng.standard_normal(shape_a), None
    if dist_id == 2:
        return np.abs(rng.standard_cauchy(shape_v)), rng.uniform(0.5, 30, shape_a), None
    if dist_id == 3:
        return np.exp(rng.standard_normal(shape_v)), rng.standard_normal(shape_a), rng.uniform(0.5, 2, shape_a)
    if dist_id == 4:
        return rng.exponential(size=shape_v), rng.uniform(0.5, 2, shape_a), None
    if dist_id == 5:
        return np.abs(rng.standard_normal(shape_v)), rng.uniform(0.5, 2, shape_a), None
    if dist_id == 6:      # Gamma(concentration, rate): small and large shapes (recurrence + series)
        return (rng.gamma(2.0, size=shape_v) + 1e-3, rng.uniform(0.1, 30, shape_a),
                rng.uniform(0.3, 4, shape_a))
    if dist_id == 7:      # Beta(c1, c0)
        return (rng.uniform(0.01, 0.99, shape_v), rng.uniform(0.2, 20, shape_a),
                rng.uniform(0.2, 20, shape_a))
    if dist_id == 8:      # Poisson(rate)
        return rng.poisson(6.0, shape_v).astype(float), rng.uniform(0.2, 40, shape_a), None
    if dist_id == 9:      # Binomial(logits, total_count)
        n = rng.integers(1, 80, shape_a).astype(float)
        k = np.floor(rng.uniform(size=shape_v) * (np.broadcast_to(n, shape_v) + 1))
        return np.minimum(k, np.broadcast_to(n, shape_v)), 3 * rng.standard_normal(shape_a), n
    if dist_id == 10:     # KL_NORMAL_LOC(lq; lp, sp)
        return rng.standard_normal(shape_v), rng.standard_normal(shape_a), rng.uniform(0.5, 2, shape_a)
    if dist_id == 11:     # KL_NORMAL_SCALE(sq; sp)
        return rng.uniform(0.2, 2, shape_v), rng.uniform(0.5, 2, shape_a), None


@pytest.mark.parametrize("dist_id", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("rows,cols,bcast", [(1, 1, "none"), (3, 7, "none"), (5, 1031, "row"),
                                             (7, 2500, "col"), (64, 4099, "none"),
                                             (300, 100, "none"), (2, 70001, "row")])
def test_dist_log_prob_sum_grad(gpu, dist_id, dtype, rows, cols, bcast):
    k = _k()
    rng = np.random.default_rng(dist_id * 100 + rows)
    v, a, b = _dist_inputs(dist_id, rows, cols, rng, bcast)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    v, a = v.astype(np_dt), a.astype(np_dt)
    b = b.astype(np_dt) if b is not None else None
    tv, ta = tt(v, gpu), tt(a, gpu)
    tb = tt(b, gpu) if b is not None else None
    rtol = 3e-5 if dtype == torch.float32 else 1e-11
    if dist_id >= 6 and dtype == torch.float32:
        rtol = 2e-4      # differences of lgamma values of size O(100) evaluated in f32
    v64, a64 = v.astype(np.float64), a.astype(np.float64)
    b64 = b.astype(np.float64) if b is not None else None

    lp = k.dist_log_prob(dist_id, tv, ta, tb, rows, cols).cpu().numpy()
    ref = np.broadcast_to(o_dists.LOG_PROB[dist_id](v64, a64, b64), (rows, cols))
    np.testing.assert_allclose(lp, ref, rtol=rtol, atol=rtol)

    mask = rng.uniform(size=(rows, cols)) < 0.7
    for m, scale in [(None, 1.0), (mask, 2.5)]:
        tm = tt(m, gpu) if m is not None else None
        s, tot = k.dist_log_prob_sum(dist_id, tv, ta, tb, tm, scale, rows, cols, want_total=True)
        s = s.cpu().numpy()
        ref_s = o_dists.log_prob_sum(dist_id, v64, a64, b64, m, scale)
        np.testing.assert_allclose(s, ref_s, rtol=rtol, atol=rtol * max(1.0, np.abs(ref_s).max()))
        np.testing.assert_allclose(tot.item(), ref_s.sum(), rtol=rtol,
                                   atol=rtol * max(1.0, np.abs(ref_s).sum()))
        g_row = rng.standard_normal(rows).astype(np_dt)
        tg = tt(g_row.reshape(rows, 1), gpu)
        outs = k.dist_log_prob_grad(dist_id, tg, tv, ta, tb, tm, scale, rows, cols,
                                    (True, True, b is not None))
        refs = o_dists.log_prob_sum_grad(dist_id, g_row.astype(np.float64), v64, a64, b64, m, scale)
        for o, r in zip(outs, refs):
            if o is not None:
                r = np.broadcast_to(r, (rows, cols))
                np.testing.assert_allclose(o.cpu().numpy(), r, rtol=rtol * 3,
                                           atol=rtol * 3 * max(1.0, np.abs(r).max()))


@pytest.mark.parametrize("fam,dist_id", [("gamma", 6), ("beta", 7), ("poisson", 8), ("binomial_logits", 9)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_gamma_function_family_classes(gpu, fam, dist_id, dtype):
    """pyro_amd.distributions.{Gamma,Beta,Poisson,Binomial}: log_prob and autograd gradients of the
    reference (fixture dists.npz, made by tests/golden/make_golden.py) through the class interface,
    element-wise and through the fused log_prob -> sum route.  f32 is checked against the oracle at
    the ROUNDED inputs (a Beta draw next to 1 moves by more than the tolerance when rounded)."""
    import os
    import pyro_amd.distributions as d
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dists.npz"))
    cont = fam in ("gamma", "beta")
    np_dt = np.float64 if dtype == torch.float64 else np.float32
    v, a = g[fam + "/v"].astype(np_dt), g[fam + "/a"].astype(np_dt)
    b = g[fam + "/b"].astype(np_dt) if fam + "/b" in g.files else None
    tv = torch.tensor(v, device=gpu, requires_grad=cont)
    ta = torch.tensor(a, device=gpu, requires_grad=True)
    tb = None if b is None else torch.tensor(b, device=gpu, requires_grad=cont)
    dd = {"gamma": lambda: d.Gamma(ta, tb), "beta": lambda: d.Beta(ta, tb),
          "poisson": lambda: d.Poisson(ta), "binomial_logits": lambda: d.Binomial(tb, logits=ta)}[fam]()
    lp = dd.log_prob(tv)
    v64, a64 = v.astype(np.float64), a.astype(np.float64)
    b64 = None if b is None else b.astype(np.float64)
    if dtype == torch.float64:
        tol, ref_lp = 1e-11, g[fam + "/lp"]
        refs = {k: g[fam + "/" + k] for k in ("dv", "da", "db") if fam + "/" + k in g.files}
    else:
        tol, ref_lp = 2e-4, o_dists.LOG_PROB[dist_id](v64, a64, b64)
        full = dict(zip(("dv", "da", "db"), o_dists.log_prob_grad(dist_id, v64, a64, b64)))
        refs = {}
        for k, like in (("dv", v), ("da", a), ("db", b)):
            if like is not None:
                f = np.broadcast_to(full[k], ref_lp.shape)
                axes = tuple(i for i, n in enumerate(like.shape) if n == 1 and f.shape[i] != 1)
                refs[k] = f.sum(axes, keepdims=True)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), ref_lp, rtol=tol, atol=tol)
    ins = [(k, t) for k, t in (("dv", tv), ("da", ta), ("db", tb)) if t is not None and t.requires_grad]
    for out in (lp.sum(), dd.fused_log_prob_sum(tv)):
        np.testing.assert_allclose(out.item(), ref_lp.sum(), rtol=tol, atol=tol)
        gs = torch.autograd.grad(out, [t for _, t in ins], retain_graph=True)
        for (k, _), got in zip(ins, gs):
            np.testing.assert_allclose(got.cpu().numpy(), refs[k], rtol=tol * 10,
                                       atol=tol * 10 * np.abs(refs[k]).max())
    # expand keeps the kernels' operands un-expanded and the values unchanged
    e = dd.expand((3,) + tuple(dd.batch_shape))
    np.testing.assert_allclose(e.log_prob(tv).detach().cpu().numpy()[1], ref_lp, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("rows,K,shared", [(1000, 8, True), (1000, 8, False), (8, 1024, False),
                                           (3, 70, True), (1, 2, False), (40000, 5, True), (5, 300, True),
                                           (2, 5000, False), (2000, 300, False)])
def test_dirichlet_log_prob_and_grad(gpu, dtype, rows, K, shared):
    """pa_dirichlet_log_prob / _grad (thread-per-row for K <= 32, wave-per-row above, a 1024-thread workgroup
    per row for few rows of K >= 256) against the
    oracle, with a concentration vector shared by all rows read through row stride 0."""
    k = _k()
    rng = np.random.default_rng(rows + K)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    x = rng.dirichlet(np.ones(K) * 0.7, rows).astype(np_dt)
    x = np.maximum(x, 1e-30).astype(np_dt)
    c = rng.uniform(0.1, 8, (K,) if shared else (rows, K)).astype(np_dt)
    w = rng.standard_normal(rows).astype(np_dt)
    tx, tc, tw = tt(x, gpu), tt(c, gpu), tt(w, gpu)
    tol = 1e-4 if dtype == torch.float32 else 1e-11
    lp = k.dirichlet_log_prob(tx, tc)
    ref = o_dists.dirichlet_log_prob(x.astype(np.float64), c.astype(np.float64))
    np.testing.assert_allclose(lp.cpu().numpy(), ref, rtol=tol, atol=tol * max(1.0, np.abs(ref).max()))
    dx, dc = k.dirichlet_log_prob_grad(tw, tx, tc, True, True)
    rdx, rdc = o_dists.dirichlet_log_prob_grad(w, x.astype(np.float64), c.astype(np.float64))
    np.testing.assert_allclose(dx.cpu().numpy(), rdx, rtol=tol * 3, atol=tol * 3 * np.abs(rdx).max())
    np.testing.assert_allclose(dc.cpu().numpy(), rdc, rtol=tol * 3, atol=tol * 3 * np.abs(rdc).max())


@pytest.mark.parametrize("tag", ["dirichlet_shared", "dirichlet_rows"])
def test_dirichlet_class_vs_reference_fixture(gpu, tag):
    """pyro_amd.distributions.Dirichlet: the reference's log_prob and autograd gradients (f64),
    also through expand over a plate."""
    import os
    import pyro_amd.distributions as d
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dists.npz"))
    tx = torch.tensor(g[tag + "/x"], device=gpu, requires_grad=True)
    tc = torch.tensor(g[tag + "/c"], device=gpu, requires_grad=True)
    w = torch.tensor(g[tag + "/w"], device=gpu)
    dd = d.Dirichlet(tc)
    if tc.dim() == 1:
        dd = dd.expand((6,))
    lp = dd.log_prob(tx)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), g[tag + "/lp"], rtol=1e-11)
    gx, gc = torch.autograd.grad((lp * w).sum(), [tx, tc])
    np.testing.assert_allclose(gx.cpu().numpy(), g[tag + "/dx"], rtol=1e-10)
    np.testing.assert_allclose(gc.cpu().numpy(), g[tag + "/dc"], rtol=1e-10, atol=1e-11)


def test_dist_empty_and_errors(gpu):
    k = _k()
    v = torch.zeros((4, 0), device=gpu)
    s = k.dist_log_prob_sum(0, v, v, v, None, 1.0, 4, 0)
    assert s.shape == (4,) and float(s.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        k.dist_log_prob(99, torch.zeros((1, 1), device=gpu), torch.zeros((1, 1), device=gpu), None, 1, 1)
    with pytest.raises(RuntimeError):
        k.dist_log_prob(0, torch.zeros((1, 1)), torch.zeros((1, 1)), torch.ones((1, 1)), 1, 1)


def test_half_cauchy_negative_support(gpu):
    k = _k()
    v = torch.tensor([[-1.0, 0.0, 2.0]], device=gpu)
    a = torch.tensor([[3.0]], device=gpu)
    lp = k.dist_log_prob(2, v, a, None, 1, 3).cpu().numpy()[0]
    assert lp[0] == -np.inf and np.isfinite(lp[1:]).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_normal_rsample(gpu, dtype):
    k = _k()
    rows, cols = 17, 33
    rng = np.random.default_rng(1)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    loc = rng.standard_normal((1, cols)).astype(np_dt)
    scale = rng.uniform(0.1, 2, (1, cols)).astype(np_dt)
    out, eps = k.normal_rsample(tt(loc, gpu), tt(scale, gpu), rows, cols, seed=9, offset=1000)
    ref_eps = o_philox.normal(rows * cols, np_dt, 9, 1000).reshape(rows, cols)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    np.testing.assert_allclose(eps.cpu().numpy(), ref_eps, rtol=tol, atol=10 * tol)
    np.testing.assert_allclose(out.cpu().numpy(), loc + scale * ref_eps, rtol=10 * tol, atol=10 * tol)


# ---------------------------------------------------------------------------------------------
# fused Bernoulli GLM -- all kernel choices (automatic = few-particle VALU kernel / bf16x3 split
# on the bf16 matrix cores, exact-f32 MFMA, bf16x3 at every P) are held to the SAME tolerances
# against the float64 oracle
# ---------------------------------------------------------------------------------------------
@pytest.fixture(params=[0, 1, 2], ids=["auto", "exact_f32", "bf16x3"])
def glm_variant(request):
    k = _k()
    k.glm_set_variant(request.param)
    yield request.param
    k.glm_set_variant(k.GLM_AUTO)


@pytest.mark.parametrize("N,D,P", [(1, 1, 1), (31, 3, 2), (32, 32, 64), (33, 32, 64), (1000, 32, 64),
                                   (4099, 8, 5), (2048, 32, 33), (5000, 20, 100), (3000, 64, 40),
                                   (1500, 48, 7), (1200, 128, 12), (700, 100, 3), (0, 4, 2),
                                   # few particles: the vector-ALU streaming kernel (every lanes-
                                   # per-row width, ragged tails, P = 1..4)
                                   (5000, 32, 1), (4097, 8, 2), (3001, 12, 3), (2500, 16, 4),
                                   (7777, 20, 1), (999, 64, 2), (1234, 128, 4), (65, 36, 3),
                                   (100000, 32, 4)])
@pytest.mark.parametrize("use_mask,use_bias", [(False, True), (True, False)])
def test_glm_bernoulli(gpu, glm_variant, N, D, P, use_mask, use_bias):
    k = _k()
    rng = np.random.default_rng(N + D + P)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32) if use_bias else None
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    mask = (rng.uniform(size=N) < 0.8) if use_mask else None
    scale = 3.0
    ll, gw, gb = k.glm_bernoulli_fwd_bwd(tt(X, gpu), tt(y, gpu), tt(w, gpu),
                                         tt(b, gpu) if b is not None else None,
                                         tt(mask, gpu) if mask is not None else None, scale)
    rll, rgw, rgb = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, mask, scale)
    sc = max(1.0, float(np.abs(rll).max()) if N else 1.0)
    np.testing.assert_allclose(ll.cpu().numpy(), rll, rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(gb.cpu().numpy(), rgb, rtol=2e-5, atol=2e-5 * max(1.0, N ** 0.5))
    np.testing.assert_allclose(gw.cpu().numpy(), rgw, rtol=2e-5, atol=2e-5 * max(1.0, N ** 0.5))


@pytest.mark.parametrize("N,D,P,G", [(1000, 32, 64, 7), (5000, 32, 64, 50), (777, 8, 5, 3),
                                     (3000, 64, 20, 11), (400, 32, 33, 40), (2000, 100, 3, 4)])
@pytest.mark.parametrize("use_mask", [False, True])
def test_glm_bernoulli_grouped(gpu, glm_variant, N, D, P, G, use_mask):
    """Hierarchical GLM (config 5): per-group weights, rows sorted by group, ragged groups
    (some empty), against the numpy restatement."""
    k = _k()
    rng = np.random.default_rng(N + D + P + G)
    sizes = rng.multinomial(N, rng.dirichlet(np.ones(G) * 0.7))
    sizes[rng.integers(0, G)] = 0                      # an empty group
    sizes[-1] += N - sizes.sum()
    off = np.concatenate([[0], np.cumsum(sizes)])
    g_of = np.repeat(np.arange(G), sizes)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, G, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32)
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    mask = (rng.uniform(size=N) < 0.8) if use_mask else None
    segs = k.GroupSegments(off, gpu, target_segments=37)
    assert segs.nseg >= int((sizes > 0).sum()) and int(segs.group_seg_off[-1]) == segs.nseg
    ll, gw, gb = k.glm_bernoulli_grouped_fwd_bwd(tt(X, gpu), tt(y, gpu), tt(w, gpu), tt(b, gpu),
                                                 tt(mask, gpu) if mask is not None else None, 2.0,
                                                 segs)
    rll, rgw, rgb = o_glm.glm_bernoulli_grouped_fwd_bwd(X, y, w, g_of, b, mask, 2.0)
    sc = max(1.0, float(np.abs(rll).max()))
    np.testing.assert_allclose(ll.cpu().numpy(), rll, rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(gb.cpu().numpy(), rgb, rtol=2e-5, atol=2e-5 * N ** 0.5)
    np.testing.assert_allclose(gw.cpu().numpy(), rgw, rtol=2e-5, atol=2e-5 * N ** 0.5)
    # one group == the flat kernel, bit for bit in ll/gb ordering-independent tolerance
    segs1 = k.GroupSegments(np.array([0, N]), gpu)
    l1, g1, b1 = k.glm_bernoulli_grouped_fwd_bwd(tt(X, gpu), tt(y, gpu), tt(w[:, :1].copy(), gpu),
                                                 tt(b, gpu), None, 1.0, segs1)
    l0, g0, b0 = k.glm_bernoulli_fwd_bwd(tt(X, gpu), tt(y, gpu), tt(w[:, 0].copy(), gpu), tt(b, gpu),
                                         None, 1.0)
    torch.testing.assert_close(l1, l0, rtol=1e-5, atol=1e-4 * sc)
    torch.testing.assert_close(g1[:, 0], g0, rtol=1e-5, atol=1e-4 * N ** 0.5)


@pytest.fixture(params=["f16x2", "bf16x3"])
def planes_fmt(request):
    """Both plane-image formats (include/pyro_amd.h PA_GLM_PLANES_*); the process default is restored."""
    k = _k()
    before = k.glm_planes_format()
    k.glm_set_planes_format(k.GLM_PLANES_F16X2 if request.param == "f16x2" else k.GLM_PLANES_BF16X3)
    yield request.param
    k.glm_set_planes_format(before)


@pytest.mark.parametrize("N,D,P,G", [(5000, 32, 64, 7), (70_000, 17, 40, 50), (300, 8, 130, 3)])
def test_glm_grouped_plane_image_kernel(gpu, planes_fmt, N, D, P, G):
    """The hierarchical GLM site on the plane image (pa_glm_pack_planes_grouped +
    pa_glm_bernoulli_grouped_planes_fwd_bwd): the packer bit-exact against the oracle's image, the
    kernel against the numpy restatement (same tolerances as the kernel that splits X on the fly)
    and against that kernel itself, ragged groups incl. an empty one, P > 64 (two particle passes)."""
    k = _k()
    rng = np.random.default_rng(N + D + P + G)
    sizes = rng.multinomial(N, rng.dirichlet(np.ones(G) * 0.7))
    sizes[rng.integers(0, G)] = 0
    sizes[-1] += N - sizes.sum()
    off = np.concatenate([[0], np.cumsum(sizes)])
    g_of = np.repeat(np.arange(G), sizes)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, G, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32)
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    segs = k.GroupSegments(off, gpu, target_segments=23)
    tX, ty = tt(X, gpu), tt(y, gpu)
    planes = k.glm_pack_planes_grouped(tX, ty, segs)
    got = planes.cpu().numpy()
    if planes_fmt == "f16x2":
        img, ypad, kx = o_glm.glm_grouped_plane_image_f16(X, y, segs.seg.cpu().numpy())
        trailer = got[img.size * 2 + ypad.size * 4:][:256].view(np.int32)
        assert np.array_equal(trailer[32:64], kx)                  # one exponent per column
    else:
        img, ypad = o_glm.glm_grouped_plane_image(X, y, segs.seg.cpu().numpy())
    nb_img = img.size * 2
    assert np.array_equal(got[:nb_img].view(np.uint16).reshape(img.shape), img)      # bit-exact
    assert np.array_equal(got[nb_img:nb_img + ypad.size * 4].view(np.float32), ypad)
    ll, gw, gb = k.glm_bernoulli_grouped_planes_fwd_bwd(planes, tt(w, gpu), tt(b, gpu), 2.0, N, D, segs)
    rll, rgw, rgb = o_glm.glm_bernoulli_grouped_fwd_bwd(X, y, w, g_of, b, None, 2.0)
    sc = max(1.0, float(np.abs(rll).max()))
    np.testing.assert_allclose(ll.cpu().numpy(), rll, rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(gb.cpu().numpy(), rgb, rtol=2e-5, atol=2e-5 * N ** 0.5)
    np.testing.assert_allclose(gw.cpu().numpy(), rgw, rtol=2e-5, atol=2e-5 * N ** 0.5)
    k.glm_set_planes_mode(k.GLM_PLANES_OFF)
    try:
        l0, g0, b0 = k.glm_bernoulli_grouped_fwd_bwd(tX, ty, tt(w, gpu), tt(b, gpu), None, 2.0, segs)
    finally:
        k.glm_set_planes_mode(k.GLM_PLANES_AUTO)
    torch.testing.assert_close(ll, l0, rtol=1e-5, atol=1e-4 * sc)
    torch.testing.assert_close(gw, g0, rtol=1e-5, atol=1e-4 * N ** 0.5)
    # the cached route: second sighting packs, an in-place change of y re-packs into the same buffer
    segs2 = k.GroupSegments(off, gpu, target_segments=23)
    for _ in range(2):
        l1, g1, b1 = k.glm_bernoulli_grouped_fwd_bwd(tX, ty, tt(w, gpu), tt(b, gpu), None, 2.0, segs2)
    assert segs2._planes[4] is not None and torch.equal(l1, ll) and torch.equal(g1, gw)
    buf = segs2._planes[4].data_ptr()
    ty.copy_(1.0 - ty)
    l2, g2, b2 = k.glm_bernoulli_grouped_fwd_bwd(tX, ty, tt(w, gpu), tt(b, gpu), None, 2.0, segs2)
    assert segs2._planes[4].data_ptr() == buf
    rll2, _, _ = o_glm.glm_bernoulli_grouped_fwd_bwd(X, 1.0 - y, w, g_of, b, None, 2.0)
    np.testing.assert_allclose(l2.cpu().numpy(), rll2, rtol=2e-5, atol=2e-5 * sc)


def test_glm_bernoulli_transpose_detecting(gpu, glm_variant):
    """Asymmetric inputs: a swapped (p,d) or (n,p) mapping cannot pass."""
    k = _k()
    N, D, P = 96, 32, 64
    X = np.zeros((N, D), np.float32)
    for n in range(N):
        X[n, (7 * n) % D] = 1.0 + n / N
    w = np.arange(P * D, dtype=np.float32).reshape(P, D) / (P * D) - 0.3
    y = (np.arange(N) % 3 == 0).astype(np.float32)
    ll, gw, gb = k.glm_bernoulli_fwd_bwd(tt(X, gpu), tt(y, gpu), tt(w, gpu), None, None, 1.0)
    rll, rgw, rgb = o_glm.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)
    np.testing.assert_allclose(ll.cpu().numpy(), rll, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gw.cpu().numpy(), rgw, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gb.cpu().numpy(), rgb, rtol=1e-5, atol=1e-5)


def test_glm_bernoulli_deterministic_and_linear(gpu, glm_variant):
    """Full-size properties (BASELINE N=1e6, D=32, P=64): bitwise run-to-run determinism, and
    additivity over a split of the plate (sum of two half-plates == whole plate)."""
    k = _k()
    N, D, P = 1_000_000, 32, 64
    g = torch.Generator(device=gpu).manual_seed(0)
    X = torch.randn((N, D), device=gpu, generator=g)
    w = torch.randn((P, D), device=gpu, generator=g) * 0.2
    b = torch.randn((P,), device=gpu, generator=g)
    y = (torch.rand((N,), device=gpu, generator=g) < 0.5).float()
    a0 = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    # automatic mode: the first call sees X for the first time and splits it on the fly; from the
    # second sight on X is kept as its bf16 plane image and the plane-image kernel runs (same
    # arithmetic class, different summation order): deterministic from there on
    a1 = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    a2 = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    for u, v in zip(a1, a2):
        assert torch.equal(u, v)
    for u, v in zip(a0, a1):
        torch.testing.assert_close(u, v, rtol=2e-5, atol=2e-2)
    if glm_variant == 0:
        assert k.glm_planes_of(X) is not None
    h = N // 2 + 13
    p1 = k.glm_bernoulli_fwd_bwd(X[:h].contiguous(), y[:h].contiguous(), w, b, None, 1.0)
    p2 = k.glm_bernoulli_fwd_bwd(X[h:].contiguous(), y[h:].contiguous(), w, b, None, 1.0)
    for whole, u, v in zip(a1, p1, p2):
        torch.testing.assert_close(whole, u + v, rtol=2e-5, atol=2e-2)
    # against torch fp64 on the same device (independent arithmetic) at full size
    logits = (w.double() @ X.double().T) + b.double()[:, None]
    ll_ref = (y.double() * logits - torch.nn.functional.softplus(logits)).sum(1)
    torch.testing.assert_close(a1[0].double(), ll_ref, rtol=2e-5, atol=1e-3)
    # the gradients at full size, all 64 particles, against torch fp64
    gfac = y.double() - torch.sigmoid(logits)
    for out in (a0, a1):
        torch.testing.assert_close(out[1].double(), gfac @ X.double(), rtol=1e-4, atol=2e-2)
        torch.testing.assert_close(out[2].double(), gfac.sum(1), rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("variant", [0, 1, 2], ids=["auto", "exact_f32", "bf16x3"])
def test_glm_masked_rows_with_huge_garbage_contribute_nothing(gpu, variant):
    """where(mask, x, 0) semantics on the matrix-core kernels (P = 64): masked rows holding 1e30
    (|x . w| far beyond the -1e30 logit offset that silences ordinary masked rows) and 3e38 leave
    every output equal to the run on the unmasked rows alone."""
    k = _k()
    N, D, P = 20000, 32, 64
    g = torch.Generator(device=gpu).manual_seed(4)
    X = torch.randn((N, D), device=gpu, generator=g)
    w = torch.randn((P, D), device=gpu, generator=g)
    b = torch.randn((P,), device=gpu, generator=g)
    y = (torch.rand((N,), device=gpu, generator=g) < 0.5).float()
    mask = torch.rand((N,), device=gpu, generator=g) < 0.7
    try:
        k.glm_set_variant(variant)
        m0 = k.glm_bernoulli_fwd_bwd(X[mask].contiguous(), y[mask].contiguous(), w, b, None, 1.0)
        for junk in (1e30, -3e38):
            Xn = X.clone()
            Xn[~mask] = junk
            m1 = k.glm_bernoulli_fwd_bwd(Xn, y, w, b, mask, 1.0)
            for u, v in zip(m1, m0):
                assert bool(torch.isfinite(u).all())
                torch.testing.assert_close(u, v, rtol=2e-5, atol=2e-2)
    finally:
        k.glm_set_variant(k.GLM_AUTO)


@pytest.mark.parametrize("P", [1, 4])
def test_glm_few_particles_full_size(gpu, P):
    """N = 1e6 with the reference's default num_particles = 1 (and 4): the streaming kernel against
    the exact-f32 matrix-core kernel and torch fp64, bitwise run-to-run determinism, masked rows
    holding huge (finite) garbage contribute exactly nothing (where(mask, x, 0) semantics; NaN
    data under the mask poison the gradient in the reference too: 0 * NaN in the matmul backward)."""
    k = _k()
    N, D = 1_000_000, 32
    g = torch.Generator(device=gpu).manual_seed(1)
    X = torch.randn((N, D), device=gpu, generator=g)
    w = torch.randn((P, D), device=gpu, generator=g) * 0.2
    b = torch.randn((P,), device=gpu, generator=g)
    y = (torch.rand((N,), device=gpu, generator=g) < 0.5).float()
    a1 = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    a2 = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    for u, v in zip(a1, a2):
        assert torch.equal(u, v)
    try:
        k.glm_set_variant(k.GLM_EXACT_F32)
        e = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    finally:
        k.glm_set_variant(k.GLM_AUTO)
    for u, v in zip(a1, e):
        torch.testing.assert_close(u, v, rtol=2e-5, atol=2e-2)
    logits = (w.double() @ X.double().T) + b.double()[:, None]
    ll_ref = (y.double() * logits - torch.nn.functional.softplus(logits)).sum(1)
    torch.testing.assert_close(a1[0].double(), ll_ref, rtol=2e-5, atol=1e-3)
    gw_ref = (y.double() - torch.sigmoid(logits)) @ X.double()
    torch.testing.assert_close(a1[1].double(), gw_ref, rtol=1e-4, atol=2e-2)
    mask = torch.rand((N,), device=gpu, generator=g) < 0.7
    Xn = X.clone()
    Xn[~mask] = 1e30
    m1 = k.glm_bernoulli_fwd_bwd(Xn, y, w, b, mask, 1.0)
    m0 = k.glm_bernoulli_fwd_bwd(X[mask].contiguous(), y[mask].contiguous(), w, b, None, 1.0)
    for u, v in zip(m1, m0):
        assert bool(torch.isfinite(u).all())
        torch.testing.assert_close(u, v, rtol=2e-5, atol=2e-2)


def test_glm_bf16x3_is_f32_class(gpu):
    """The split-precision variant is an f32-equivalent: its error against the float64 oracle on
    per-row logit-sensitive outputs is of the size of the exact-f32 kernel's own rounding error
    (a plain bf16 GEMM would be ~1e-2 relative here)."""
    k = _k()
    N, D, P = 4096, 32, 64
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((N, D)) * np.exp(rng.uniform(-3, 3, (N, 1)))).astype(np.float32)
    w = rng.standard_normal((P, D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32)
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    ref = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    errs = {}
    try:
        for v in (k.GLM_BF16X3, k.GLM_EXACT_F32):
            k.glm_set_variant(v)
            out = k.glm_bernoulli_fwd_bwd(tt(X, gpu), tt(y, gpu), tt(w, gpu), tt(b, gpu), None, 1.0)
            errs[v] = [float(np.abs(o.cpu().numpy() - r).max() / np.abs(r).max())
                       for o, r in zip(out, ref)]
    finally:
        k.glm_set_variant(k.GLM_AUTO)
    for e_split, e_exact in zip(errs[k.GLM_BF16X3], errs[k.GLM_EXACT_F32]):
        assert e_split < 3e-6, errs                     # f32-class absolute bound
        assert e_split < 8 * e_exact + 5e-7, errs       # and comparable to the exact kernel's


# ---------------------------------------------------------------------------------------------
# fused Bernoulli GLM on the cached bf16 plane image of X (pa_glm_pack_planes /
# pa_glm_bernoulli_planes_fwd_bwd): same tolerances against the float64 oracle as the on-the-fly
# kernels; the image itself is integer work and must match the numpy restatement bit for bit
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,D", [(1, 1), (31, 3), (32, 32), (33, 32), (127, 20), (128, 32), (129, 7),
                                 (1000, 32), (4099, 8), (0, 4)])
def test_glm_plane_image_bit_exact(gpu, planes_fmt, N, D):
    k = _k()
    rng = np.random.default_rng(7 * N + D)
    X = (rng.standard_normal((N, D)) * np.exp(rng.uniform(-6, 6, (N, 1)))).astype(np.float32)
    img = k.glm_pack_planes(tt(X, gpu)).cpu().numpy()
    if planes_fmt == "f16x2":
        ref, kx = o_glm.glm_plane_image_f16(X)
        assert np.array_equal(img[ref.size * 2:][:256].view(np.int32)[32:64], kx)   # the trailer's exponents
    else:
        ref = o_glm.glm_plane_image(X)
    got = img[:ref.size * 2].view(np.uint16).reshape(ref.shape)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("N,D", [(1, 33), (100, 64), (4099, 100), (1000, 128), (0, 70)])
def test_glm_plane_image_with_feature_tiles_bit_exact(gpu, N, D):
    """32 < D <= 128 (csrc/glm_planes16d.h): DT = 2 / 4 sub-tiles of 32 columns per row tile, 128 column
    exponents in a 1-KiB trailer; bf16x3 images stop at 32 columns."""
    k = _k()
    rng = np.random.default_rng(7 * N + D)
    X = (rng.standard_normal((N, D)) * np.exp(rng.uniform(-6, 6, (N, 1)))
         * np.exp(rng.uniform(-8, 8, (1, D)))).astype(np.float32)
    img = k.glm_pack_planes(tt(X, gpu), fmt=k.GLM_PLANES_F16X2).cpu().numpy()
    ref, kx = o_glm.glm_plane_image_f16(X)
    assert np.array_equal(img[ref.size * 2:][:1024].view(np.int32)[128:256], kx)
    assert np.array_equal(img[:ref.size * 2].view(np.uint16).reshape(ref.shape), ref)
    with pytest.raises(Exception):
        k.glm_pack_planes(tt(X, gpu), fmt=k.GLM_PLANES_BF16X3)


@pytest.mark.parametrize("N,D,P", [(1, 33, 64), (63, 40, 33), (4099, 64, 64), (1000, 100, 70), (70000, 128, 64),
                                   (5000, 65, 130)])
@pytest.mark.parametrize("use_bias", [True, False])
def test_glm_planes_with_feature_tiles(gpu, N, D, P, use_bias):
    """The plane-image kernel for 32 < D <= 128 against the float64 oracle (the tolerances of the D <= 32
    kernel) and against the kernel that splits X on the fly."""
    k = _k()
    rng = np.random.default_rng(N + D + P)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32) if use_bias else None
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    tX, ty, tw = tt(X, gpu), tt(y, gpu), tt(w, gpu)
    tb = tt(b, gpu) if use_bias else None
    planes = k.glm_pack_planes(tX, fmt=k.GLM_PLANES_F16X2)
    ll, gw, gb = k.glm_bernoulli_planes_fwd_bwd(planes, ty, tw, tb, 3.0, N, D)
    rll, rgw, rgb = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 3.0)
    sc = max(1.0, float(np.abs(rll).max()))
    np.testing.assert_allclose(ll.cpu().numpy(), rll, rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(gb.cpu().numpy(), rgb, rtol=2e-5, atol=2e-5 * max(1.0, N ** 0.5))
    np.testing.assert_allclose(gw.cpu().numpy(), rgw, rtol=2e-5, atol=2e-5 * max(1.0, N ** 0.5))
    k.glm_set_planes_mode(k.GLM_PLANES_OFF)
    try:
        l0, g0, b0 = k.glm_bernoulli_fwd_bwd(tX, ty, tw, tb, None, 3.0)
    finally:
        k.glm_set_planes_mode(k.GLM_PLANES_AUTO)
    torch.testing.assert_close(ll, l0, rtol=1e-5, atol=1e-4 * sc)
    torch.testing.assert_close(gw, g0, rtol=1e-5, atol=1e-4 * max(1.0, N ** 0.5))


@pytest.mark.parametrize("scale", [1e-30, 3e-5, 1.0, 77.0, 1e20, 0.0])
def test_glm_plane_image_f16_exponent(gpu, scale):
    """The f16 image's power-of-two scale follows max |X| over the whole f32 range (and is 0 for an
    all-zero matrix); pieces bit-exact against the numpy restatement."""
    k = _k()
    rng = np.random.default_rng(3)
    X = (rng.standard_normal((200, 9)) * scale).astype(np.float32)
    img = k.glm_pack_planes(tt(X, gpu), fmt=k.GLM_PLANES_F16X2).cpu().numpy()
    ref, kx = o_glm.glm_plane_image_f16(X)
    assert np.array_equal(img[ref.size * 2:][:256].view(np.int32)[32:64], kx)
    assert np.array_equal(img[:ref.size * 2].view(np.uint16).reshape(ref.shape), ref)


# (9: one wave per 32-row tile and 64 particles, csrc/glm_planes16w.h -- f16 image only; the bf16x3
#  image reads the code as its ring depth 4)
@pytest.fixture(params=[3, 4, 9], ids=["ring3", "ring4", "wide"])
def planes_ring(request):
    k = _k()
    k.glm_planes_tune(request.param, 0)
    yield request.param
    k.glm_planes_tune(0, 0)


@pytest.mark.parametrize("N,D,P", [(1, 1, 33), (31, 3, 40), (32, 32, 64), (33, 32, 64), (63, 32, 64),
                                   (64, 32, 64), (65, 32, 64), (1000, 32, 64), (4099, 8, 37),
                                   (2048, 32, 33), (5000, 20, 100), (70000, 32, 64),
                                   (3001, 12, 130), (129, 32, 5)])
@pytest.mark.parametrize("use_bias", [True, False])
def test_glm_planes(gpu, planes_fmt, planes_ring, N, D, P, use_bias):
    k = _k()
    rng = np.random.default_rng(N + D + P)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32) if use_bias else None
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    scale = 3.0
    planes = k.glm_pack_planes(tt(X, gpu))
    ll, gw, gb = k.glm_bernoulli_planes_fwd_bwd(planes, tt(y, gpu), tt(w, gpu),
                                                tt(b, gpu) if b is not None else None, scale, N, D)
    rll, rgw, rgb = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, scale)
    sc = max(1.0, float(np.abs(rll).max()))
    np.testing.assert_allclose(ll.cpu().numpy(), rll, rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(gb.cpu().numpy(), rgb, rtol=2e-5, atol=2e-5 * max(1.0, N ** 0.5))
    np.testing.assert_allclose(gw.cpu().numpy(), rgw, rtol=2e-5, atol=2e-5 * max(1.0, N ** 0.5))


def test_glm_planes_transpose_detecting_and_f32_class(gpu, planes_fmt):
    """Asymmetric operands with a wide dynamic range (any row/column or K-slot permutation error in
    the DMA image, the row reads or the transpose reads shows up), and the f32-class error bound
    of test_glm_bf16x3_is_f32_class on the plane-image kernel."""
    k = _k()
    N, D, P = 4096 + 37, 32, 64
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((N, D)) * np.exp(rng.uniform(-3, 3, (N, 1)))).astype(np.float32)
    X *= (1.0 + 0.1 * np.arange(D, dtype=np.float32))[None, :]
    w = (rng.standard_normal((P, D)) * (1.0 + 0.05 * np.arange(P))[:, None]).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32)
    y = (rng.uniform(size=N) < 0.3).astype(np.float32)
    ref = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    out = k.glm_bernoulli_planes_fwd_bwd(k.glm_pack_planes(tt(X, gpu)), tt(y, gpu), tt(w, gpu),
                                         tt(b, gpu), 1.0, N, D)
    for o, r in zip(out, ref):
        assert float(np.abs(o.cpu().numpy() - r).max() / np.abs(r).max()) < 3e-6


@pytest.mark.parametrize("case", ["plain", "bias_dominated", "tiny_w", "huge_w", "zero_w", "wide_x",
                                  "mixed_particles", "x_1e-30", "raw_columns", "wide_columns"])
def test_glm_planes_f16_is_f32_class(gpu, case):
    """The two-plane f16 image (csrc/glm_planes16.h): its error against the float64 oracle is of the
    size of an f32 evaluation's own error -- measured next to torch's f32 matmul + softplus on the
    same inputs -- whatever the magnitudes of X, w and b (the power-of-two scales are per COLUMN of X
    and per particle).  ``raw_columns``: an un-standardised design matrix, columns of order 1e6 beside
    0/1 indicators, with weights that re-balance them (products of order one in every column);
    ``wide_columns``: column scales 18 decades apart, weights inverse to them."""
    k = _k()
    N, D, P = 20000, 32, 64
    rng = np.random.default_rng(17)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32)
    if case == "bias_dominated":
        w *= np.float32(1e-3)
        b *= np.float32(8.0)
    elif case == "tiny_w":
        w *= np.float32(1e-12)
        b *= np.float32(1e-12)
    elif case == "huge_w":
        X *= np.float32(1e-9)
        w *= np.float32(1e9)
    elif case == "zero_w":
        w[:] = 0
        b[:] = 0
    elif case == "wide_x":
        X *= np.exp(rng.uniform(-9, 9, (N, 1))).astype(np.float32)
        w *= np.float32(0.01)
    elif case == "x_1e-30":
        X *= np.float32(1e-30)
        w *= np.float32(1e30)
    elif case == "mixed_particles":
        w *= np.exp(rng.uniform(-12, 3, (P, 1))).astype(np.float32)
    elif case == "raw_columns":
        X[:, 16:] = (rng.uniform(size=(N, 16)) < 0.3).astype(np.float32)          # indicators
        X[:, :16] *= np.float32(1e6)                                              # raw measurements
        w[:, :16] *= np.float32(1e-6)
    elif case == "wide_columns":
        cs = np.exp(rng.uniform(-20, 20, (1, D))).astype(np.float32)
        X *= cs
        w /= cs
    y = (rng.uniform(size=N) < 0.4).astype(np.float32)
    ref = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    tX, ty, tw, tb = tt(X, gpu), tt(y, gpu), tt(w, gpu), tt(b, gpu)
    out = k.glm_bernoulli_planes_fwd_bwd(k.glm_pack_planes(tX, fmt=k.GLM_PLANES_F16X2), ty, tw, tb,
                                         1.0, N, D)
    # torch's own f32 evaluation of the same quantities (CPU, the reference's arithmetic)
    cX, cy, cw, cb = (torch.from_numpy(a) for a in (X, y, w, b))
    lg = cw @ cX.t() + cb[:, None]
    t_ll = (cy * lg - torch.nn.functional.softplus(lg)).sum(1)
    g = cy - torch.sigmoid(lg)
    f32 = (t_ll.numpy(), (g @ cX).numpy(), g.sum(1).numpy())
    for name, o, r, f in zip(("ll", "gw", "gb"), out, ref, f32):
        denom = max(float(np.abs(r).max()), 1e-30)
        e_ours = float(np.abs(o.cpu().numpy() - r).max()) / denom
        e_f32 = float(np.abs(f - r).max()) / denom
        assert e_ours < 3e-6, (case, name, e_ours, e_f32)
        assert e_ours < 4 * e_f32 + 3e-7, (case, name, e_ours, e_f32)


@pytest.mark.parametrize("variant", ["planes", "planes_bf16x3", "bf16", "exact"])
def test_glm_extreme_logits(gpu, variant):
    """Saturated logits (|x.w + b| up to ~200): exp2(-|l|) underflows to zero, 1 + e = 1, the
    sigmoid is exactly 0 or 1; log-likelihood and gradients stay finite and equal the float64 oracle
    on every kernel variant."""
    k = _k()
    N, D, P = 3000, 32, 64
    rng = np.random.default_rng(11)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (6.0 * rng.standard_normal((P, D))).astype(np.float32)          # logits ~ N(0, 34^2)
    b = (20.0 * rng.standard_normal(P)).astype(np.float32)
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    ref = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    tX, ty, tw, tb = tt(X, gpu), tt(y, gpu), tt(w, gpu), tt(b, gpu)
    try:
        if variant.startswith("planes"):
            fmt = k.GLM_PLANES_BF16X3 if variant.endswith("bf16x3") else k.GLM_PLANES_F16X2
            out = k.glm_bernoulli_planes_fwd_bwd(k.glm_pack_planes(tX, fmt=fmt), ty, tw, tb, 1.0, N, D)
        else:
            k.glm_set_planes_mode(k.GLM_PLANES_OFF)
            k.glm_set_variant(k.GLM_EXACT_F32 if variant == "exact" else k.GLM_AUTO)
            out = k.glm_bernoulli_fwd_bwd(tX, ty, tw, tb, None, 1.0)
    finally:
        k.glm_set_planes_mode(k.GLM_PLANES_AUTO)
        k.glm_set_variant(k.GLM_AUTO)
    for o, r in zip(out, ref):
        got = o.cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, r, rtol=3e-5, atol=3e-5 * np.abs(r).max())


def test_glm_planes_cache_follows_the_tensor(gpu):
    """The image is cached per tensor object: first sight -> on-the-fly kernel, second sight ->
    packed; an in-place update of X re-packs into the same buffer (the pointer a captured graph
    holds stays valid); results always equal the on-the-fly kernel's within f32 roundoff."""
    k = _k()
    N, D, P = 5000, 32, 64
    g = torch.Generator(device="cpu").manual_seed(3)
    X = torch.randn((N, D), generator=g).to(gpu)
    y = (torch.rand((N,), generator=g) < 0.5).float().to(gpu)
    w = (torch.randn((P, D), generator=g) * 0.2).to(gpu)
    try:
        k.glm_set_planes_mode(k.GLM_PLANES_OFF)
        base = k.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)
        k.glm_set_planes_mode(k.GLM_PLANES_AUTO)
        assert k.glm_planes_of(X) is None                  # first sight
        img = k.glm_planes_of(X)                           # second sight: packed
        assert img is not None and k.glm_planes_of(X) is img
        out = k.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)
        for u, v in zip(out, base):
            torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-3)
        X.mul_(-0.5)                                       # in place: version counter moves
        k.glm_planes_revalidate()
        assert k.glm_planes_of(X).data_ptr() == img.data_ptr()
        out2 = k.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)
        k.glm_set_planes_mode(k.GLM_PLANES_OFF)
        base2 = k.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)
        for u, v in zip(out2, base2):
            torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-3)
        assert not torch.allclose(out2[1], out[1])
    finally:
        k.glm_set_planes_mode(k.GLM_PLANES_AUTO)


# ---------------------------------------------------------------------------------------------
# leapfrog
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_leapfrog_matches_oracle(gpu, dtype):
    k = _k()
    C, D = 37, 100
    rng = np.random.default_rng(3)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    A = rng.standard_normal((D, D))
    Lam = np.linalg.inv(A @ A.T / D + 0.1 * np.eye(D))
    z = rng.standard_normal((C, D)).astype(np_dt)
    r = rng.standard_normal((C, D)).astype(np_dt)
    im = rng.uniform(0.5, 2, (C, D)).astype(np_dt)
    step = rng.uniform(0.01, 0.2, C).astype(np_dt)
    tz, tr, tim, tstep = tt(z, gpu), tt(r, gpu), tt(im, gpu), tt(step, gpu)
    tL = tt(Lam.astype(np_dt), gpu)
    tg = tz @ tL
    nsteps = 5
    for _ in range(nsteps):
        k.leapfrog_kick_drift(tz, tr, tg, tim, tstep)
        tg = tz @ tL
        k.leapfrog_kick(tr, tg, tstep)
    pg = o_int.gaussian_potential(Lam)
    tol = 2e-4 if dtype == torch.float32 else 1e-10
    for c in range(0, C, 9):
        zz, rr, _, _ = o_int.velocity_verlet(z[c].astype(np.float64), r[c].astype(np.float64), pg,
                                             im[c].astype(np.float64), float(step[c]), nsteps)
        np.testing.assert_allclose(tz[c].cpu().numpy(), zz, rtol=tol, atol=tol)
        np.testing.assert_allclose(tr[c].cpu().numpy(), rr, rtol=tol, atol=tol)


# ---------------------------------------------------------------------------------------------
# fused NUTS transition on the Gaussian potential
# ---------------------------------------------------------------------------------------------
def _gauss_problem(D, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((D, D))
    Sigma = A @ A.T / D + 0.1 * np.eye(D)
    Lam = np.linalg.inv(Sigma)
    return Sigma, 0.5 * (Lam + Lam.T)


@pytest.mark.parametrize("D,multinomial", [(5, True), (100, True), (70, False), (120, True)])
def test_nuts_gaussian_f64_matches_recursive_oracle(gpu, D, multinomial):
    """float64: the iterative one-wave-per-chain kernel must reproduce the recursive reference
    formulation chain by chain (identical tree sizes / accept decisions, positions to 1e-9)."""
    k = _k()
    C, T_ = 24, 4
    _, Lam = _gauss_problem(D)
    rng = np.random.default_rng(5)
    z0 = rng.standard_normal((C, D)) * 0.5
    im = rng.uniform(0.5, 1.5, (C, D))
    step = rng.uniform(0.05, 0.4, C)
    seed = 2024
    tz = tt(z0, gpu)
    tL = tt(Lam, gpu)
    tg = (tz @ tL).contiguous()
    tpe = (0.5 * (tz * tg).sum(1)).contiguous()
    tim, tstep = tt(im, gpu), tt(step, gpu)
    pg = o_int.gaussian_potential(Lam)
    zs = [z0[c].copy() for c in range(C)]
    st = [pg(zs[c]) for c in range(C)]
    for t in range(T_):
        out = k.nuts_gaussian_transition(tz, tpe, tg, tL, tim, tstep, 8, multinomial, seed, t)
        nl = out["n_leapfrog"].cpu().numpy()
        dp = out["depth"].cpu().numpy()
        ac = out["accepted"].cpu().numpy()
        dv = out["diverging"].cpu().numpy()
        ap = out["accept_prob"].cpu().numpy()
        for c in range(C):
            ref = o_nuts.nuts_transition(zs[c], st[c][0], st[c][1], pg, im[c], step[c],
                                         o_nuts.KeyedDraws(seed, c, t, np.float64), 8, multinomial)
            assert nl[c] == ref["n_leapfrog"], (t, c)
            assert dp[c] == ref["depth"] and bool(ac[c]) == ref["accepted"]
            assert bool(dv[c]) == ref["diverging"]
            np.testing.assert_allclose(ap[c], ref["accept_prob"], rtol=1e-9)
            np.testing.assert_allclose(tz[c].cpu().numpy(), ref["z"], rtol=1e-9, atol=1e-9)
            zs[c] = ref["z"]
            st[c] = (ref["pe"], ref["grad"])
        np.testing.assert_allclose(tpe.cpu().numpy(), [s[0] for s in st], rtol=1e-9, atol=1e-9)


def test_nuts_gaussian_f32_statistics(gpu):
    """float32, config-3 size (1024 chains x 100 dims): after a short run from N(0,I) starts with a
    fixed step size the pooled sample covariance matches Sigma (statistical parity)."""
    k = _k()
    C, D = 1024, 100
    Sigma, Lam = _gauss_problem(D)
    tz = torch.zeros((C, D), device=gpu)
    tL = tt(Lam.astype(np.float32), gpu)
    tg = (tz @ tL).contiguous()
    tpe = torch.zeros((C,), device=gpu)
    tim = torch.ones((C, D), device=gpu)
    tstep = torch.full((C,), 0.12, device=gpu)
    acc, tot = [], 0
    for t in range(60):
        out = k.nuts_gaussian_transition(tz, tpe, tg, tL, tim, tstep, 10, True, 11, t)
        tot += int(out["n_leapfrog"].sum())
        if t >= 30:
            acc.append(tz.cpu().numpy().copy())
    samples = np.concatenate(acc, 0)
    cov = np.cov(samples.T)
    assert float(out["accept_prob"].mean()) > 0.6
    assert np.abs(samples.mean(0)).max() < 0.1 * np.sqrt(np.diag(Sigma)).max() + 0.05
    rel = np.abs(np.diag(cov) - np.diag(Sigma)) / np.diag(Sigma)
    assert rel.max() < 0.15, rel.max()
    assert tot > 0


# ---------------------------------------------------------------------------------------------
# LDA enumerated factor
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("Wd,B,T,V", [(8, 100, 8, 100), (64, 1000, 8, 1024), (3, 1, 5, 7),
                                      (16, 300, 20, 50), (0, 10, 4, 6)])
def test_lda_factor(gpu, dtype, Wd, B, T, V):
    k = _k()
    rng = np.random.default_rng(Wd + B)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    words = rng.integers(0, V, (Wd, B))
    theta = rng.dirichlet(np.ones(T) * 0.5, B)
    phi = rng.dirichlet(np.ones(V) * 0.1, T)
    lt = np.log(theta).astype(np_dt)
    lp = np.log(np.maximum(phi, 1e-30)).astype(np_dt)
    out, gt, gp = k.lda_factor_fwd_bwd(tt(words, gpu), tt(lt, gpu), tt(lp, gpu))
    r_out, r_gt, r_gp = o_lda.lda_factor(words, lt, lp)
    rtol = 3e-5 if dtype == torch.float32 else 1e-11
    np.testing.assert_allclose(out.cpu().numpy(), r_out, rtol=rtol, atol=rtol * 10)
    np.testing.assert_allclose(gt.cpu().numpy(), r_gt, rtol=rtol, atol=rtol * max(Wd, 1))
    np.testing.assert_allclose(gp.cpu().numpy(), r_gp, rtol=rtol * 4, atol=rtol * max(Wd * B / V, 1) * 4)
    # posterior responsibilities sum to one per word: sum of g_theta rows == Wd (size-free property)
    np.testing.assert_allclose(gt.sum(1).cpu().numpy(), np.full(B, Wd), rtol=1e-5)


def _lda_words(rng, Wd, B, V, zipf):
    if not zipf:
        return rng.integers(0, V, (Wd, B))
    p = 1.0 / np.arange(1, V + 1) ** 1.3          # a few words hold most of the corpus
    return rng.choice(V, size=(Wd, B), p=p / p.sum())


@pytest.mark.parametrize("Wd,B,V,zipf", [(8, 100, 100, False), (64, 3000, 1024, False),
                                         (64, 3000, 1024, True), (3, 1, 7, False), (0, 10, 6, False),
                                         (5, 70001, 33, True)])
def test_lda_word_index_bit_exact(gpu, Wd, B, V, zipf):
    """pa_lda_build_index against the oracle's stable sort: integer work, bit for bit."""
    k = _k()
    rng = np.random.default_rng(Wd * 7 + B)
    words = _lda_words(rng, Wd, B, V, zipf)
    if Wd * B > 10:
        words.reshape(-1)[3] = V + 5          # a support violation is filed under word 0
        words.reshape(-1)[7] = -2
    img = k.lda_build_index(tt(words, gpu), V).cpu().numpy()
    off, first_task, task_v, task_start, task_len, docs = o_lda.lda_word_index(words, V)
    n, nt = Wd * B, len(task_v)
    cap = n // o_lda.LDA_SEG + V + 1
    assert tuple(img[1:7]) == (Wd, B, V, nt, cap, n)
    p = 8
    assert np.array_equal(img[p:p + V + 1], off); p += V + 1
    assert np.array_equal(img[p:p + V + 1], first_task); p += V + 1
    assert np.array_equal(img[p:p + nt], task_v); p += cap
    assert np.array_equal(img[p:p + nt], task_start); p += cap
    assert np.array_equal(img[p:p + nt], task_len); p += cap
    assert np.array_equal(img[p:p + n], docs) and p + n == img.size


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("Wd,B,T,V,zipf", [(8, 100, 8, 100, False), (64, 1000, 8, 1024, False),
                                           (64, 4000, 8, 1024, True), (3, 1, 5, 7, False),
                                           (16, 300, 20, 50, True), (0, 10, 4, 6, False)])
def test_lda_factor_indexed(gpu, dtype, Wd, B, T, V, zipf):
    """The atomic-free (inverted index) route: oracle parity, bitwise reproducible results, and
    agreement with the atomic route."""
    k = _k()
    rng = np.random.default_rng(Wd + B)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    words = _lda_words(rng, Wd, B, V, zipf)
    theta = rng.dirichlet(np.ones(T) * 0.5, B)
    phi = rng.dirichlet(np.ones(V) * 0.1, T)
    lt = np.log(theta).astype(np_dt)
    lp = np.log(np.maximum(phi, 1e-30)).astype(np_dt)
    tw, tlt, tlp = tt(words, gpu), tt(lt, gpu), tt(lp, gpu)
    index = k.lda_build_index(tw, V)
    out, gt, gp = k.lda_factor_fwd_bwd(tw, tlt, tlp, index=index)
    r_out, r_gt, r_gp = o_lda.lda_factor(words, lt, lp)
    rtol = 3e-5 if dtype == torch.float32 else 1e-11
    np.testing.assert_allclose(out.cpu().numpy(), r_out, rtol=rtol, atol=rtol * 10)
    np.testing.assert_allclose(gt.cpu().numpy(), r_gt, rtol=rtol, atol=rtol * max(Wd, 1))
    np.testing.assert_allclose(gp.cpu().numpy(), r_gp, rtol=rtol * 4, atol=rtol * max(Wd * B / V, 1) * 4)
    out2, gt2, gp2 = k.lda_factor_fwd_bwd(tw, tlt, tlp, index=k.lda_build_index(tw, V))
    assert torch.equal(gp, gp2) and torch.equal(out, out2) and torch.equal(gt, gt2)
    k.lda_set_index_mode(k.LDA_INDEX_OFF)
    try:
        out3, gt3, gp3 = k.lda_factor_fwd_bwd(tw, tlt, tlp)
    finally:
        k.lda_set_index_mode(k.LDA_INDEX_AUTO)
    # (the two routes split a document's words over a different number of waves: same sums, other order)
    np.testing.assert_allclose(out.cpu().numpy(), out3.cpu().numpy(), rtol=rtol, atol=rtol * 10)
    np.testing.assert_allclose(gt.cpu().numpy(), gt3.cpu().numpy(), rtol=rtol, atol=rtol * max(Wd, 1))
    np.testing.assert_allclose(gp.cpu().numpy(), gp3.cpu().numpy(), rtol=rtol * 4,
                               atol=rtol * max(Wd * B / V, 1) * 4)


def test_lda_index_cache_policy(gpu):
    """AUTO: the first call on a tensor takes the atomic route, the second builds the index; an
    in-place change of the corpus re-builds it into the same buffer."""
    k = _k()
    rng = np.random.default_rng(5)
    Wd, B, T, V = 16, 500, 8, 64
    words = tt(rng.integers(0, V, (Wd, B)), gpu)
    lt = torch.log_softmax(torch.randn(B, T, device=gpu), -1)
    lp = torch.log_softmax(torch.randn(T, V, device=gpu), -1)
    k._lda_index_cache.clear()
    a = k.lda_factor_fwd_bwd(words, lt, lp)
    ent = next(iter(k._lda_index_cache.values()))
    assert ent[2] is None and ent[3] == 1
    b = k.lda_factor_fwd_bwd(words, lt, lp)
    assert ent[2] is not None and ent[3] == 2
    buf = ent[2].data_ptr()
    for x, y in zip(a, b):
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-5, atol=2e-5)
    words[0, :10] = (words[0, :10] + 1) % V
    c = k.lda_factor_fwd_bwd(words, lt, lp)
    assert ent[2].data_ptr() == buf and ent[1] == words._version
    r = o_lda.lda_factor(words.cpu().numpy(), lt.cpu().numpy(), lp.cpu().numpy())
    np.testing.assert_allclose(c[2].cpu().numpy(), r[2], rtol=2e-4, atol=2e-3)
    # a mini-batch is a fresh tensor every step: first sighting, atomic route, no index built
    sub = words[:, :100].contiguous()
    k.lda_factor_fwd_bwd(sub, lt[:100].contiguous(), lp)
    fresh = [e for e in k._lda_index_cache.values() if e[0]() is sub]
    assert len(fresh) == 1 and fresh[0][2] is None


def test_lda_factor_indexed_full_size_properties(gpu):
    """BASELINE config 4 size (1e5 documents x 64 words, T=8, V=1024), f32: responsibilities sum
    to one per pair on both sides of the factor, both routes agree, g_phi is reproducible."""
    k = _k()
    g = torch.Generator(device=gpu).manual_seed(0)
    Wd, B, T, V = 64, 100_000, 8, 1024
    words = torch.randint(0, V, (Wd, B), device=gpu, generator=g)
    lt = torch.log_softmax(torch.randn(B, T, device=gpu, generator=g), -1)
    lp = torch.log_softmax(2 * torch.randn(T, V, device=gpu, generator=g), -1)
    index = k.lda_build_index(words, V)
    out, gt, gp = k.lda_factor_fwd_bwd(words, lt, lp, index=index)
    assert torch.allclose(gt.sum(1), torch.full((B,), float(Wd), device=gpu), rtol=1e-5)
    counts = torch.bincount(words.reshape(-1), minlength=V).double()
    np.testing.assert_allclose(gp.double().sum(0).cpu().numpy(), counts.cpu().numpy(), rtol=2e-5)
    k.lda_set_index_mode(k.LDA_INDEX_OFF)
    try:
        out_a, gt_a, gp_a = k.lda_factor_fwd_bwd(words, lt, lp)
    finally:
        k.lda_set_index_mode(k.LDA_INDEX_AUTO)
    np.testing.assert_allclose(out.cpu().numpy(), out_a.cpu().numpy(), rtol=3e-5)
    np.testing.assert_allclose(gt.cpu().numpy(), gt_a.cpu().numpy(), rtol=3e-5, atol=2e-3)
    np.testing.assert_allclose(gp.cpu().numpy(), gp_a.cpu().numpy(), rtol=2e-4, atol=1e-2)
    assert torch.equal(gp, k.lda_factor_fwd_bwd(words, lt, lp, index=index)[2])


def test_lda_factor_indexed_full_size_against_the_oracle(gpu):
    """BASELINE config 4 at ITS size (1e5 documents x 64 words, T = 8, V = 1024): the
    inverted index bit for bit against oracle/lda.py::lda_word_index at 6.4 M pairs (int32 ranks, the
    task cut of frequent words), and pa_lda_factor_indexed_fwd_bwd against oracle/lda.py::lda_factor
    (float64 numpy, the reference's max-shift / exp / sum / log chain) at f32 1e-4 -- a Zipf-like
    corpus, so that word lists span many tasks and the last documents sit in ragged tiles."""
    k = _k()
    Wd, B, T, V = 64, 100_000, 8, 1024
    rng = np.random.default_rng(4)
    words = _lda_words(rng, Wd, B, V, True)
    lt = np.log(rng.dirichlet(np.ones(T) * 0.5, size=B)).astype(np.float32)
    lp = np.log(rng.dirichlet(np.ones(V) * 0.1, size=T) + 1e-30).astype(np.float32)
    tw, tlt, tlp = tt(words, gpu), tt(lt, gpu), tt(lp, gpu)
    img = k.lda_build_index(tw, V).cpu().numpy()
    off, first_task, task_v, task_start, task_len, docs = o_lda.lda_word_index(words, V)
    n, nt = Wd * B, len(task_v)
    cap = n // o_lda.LDA_SEG + V + 1
    assert tuple(img[1:7]) == (Wd, B, V, nt, cap, n)
    p = 8 + 2 * (V + 1)
    assert np.array_equal(img[8:8 + V + 1], off) and np.array_equal(img[p:p + nt], task_v)
    assert np.array_equal(img[p + 3 * cap:p + 3 * cap + n], docs)                     # bit-exact
    out, gt, gp = k.lda_factor_fwd_bwd(tw, tlt, tlp, index=k.lda_build_index(tw, V))
    r_out, r_gt, r_gp = o_lda.lda_factor(words, lt, lp)
    np.testing.assert_allclose(out.cpu().numpy(), r_out, rtol=1e-4)
    np.testing.assert_allclose(gt.cpu().numpy(), r_gt, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gp.cpu().numpy(), r_gp, rtol=1e-4, atol=1e-4 * float(np.sqrt(r_gp.max())))


def test_lda_factor_minus_inf_column(gpu):
    k = _k()
    lt = torch.log(torch.tensor([[1.0, 0.0], [0.5, 0.5]], device=gpu, dtype=torch.float64))
    lp = torch.log(torch.tensor([[0.5, 0.5, 0.0], [0.2, 0.3, 0.5]], device=gpu, dtype=torch.float64))
    words = torch.tensor([[0, 2], [2, 1]], device=gpu)
    out, gt, gp = k.lda_factor_fwd_bwd(words, lt, lp)
    r_out, r_gt, r_gp = o_lda.lda_factor(words.cpu().numpy(), lt.cpu().numpy(), lp.cpu().numpy())
    np.testing.assert_allclose(out.cpu().numpy(), r_out)   # includes a -inf document
    assert not torch.isnan(gt).any() and not torch.isnan(gp).any()


# ---------------------------------------------------------------------------------------------
# one elimination step of the plated sum-product (logsumexp of a sum of broadcast terms)
# ---------------------------------------------------------------------------------------------
_LSE_CASES = [
    # frame, reduced dim, term shapes (broadcastable to the frame)
    ((16, 5000), 0, [(16, 1), (16, 5000)]),                         # mixture: weights + per-datum terms
    ((7, 33), 0, [(7, 33)]),
    ((4, 6, 50), 1, [(4, 6, 1), (1, 6, 50), (4, 1, 50)]),           # middle dim, three factors
    ((3, 5, 2, 129), 1, [(3, 5, 1, 1), (1, 5, 2, 129), (3, 1, 2, 1), (1, 5, 1, 129)]),
    ((8, 1, 1000), 0, [(8, 1, 1), (8, 1, 1000)]),
    ((5,), 0, [(5,), (5,)]),
    ((2, 3, 4, 5, 6, 7), 3, [(2, 1, 4, 5, 1, 7), (1, 3, 1, 5, 6, 1)]),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("case", range(len(_LSE_CASES)))
def test_logsumexp_terms(gpu, dtype, case):
    from oracle import logsumexp as o_lse
    k = _k()
    frame, rdim, shapes = _LSE_CASES[case]
    rng = np.random.default_rng(case)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    terms = [(3 * rng.standard_normal(sh)).astype(np_dt) for sh in shapes]
    terms[-1].reshape(-1)[::7] = -np.inf                      # impossible assignments
    if case == 1:
        terms[0][:, 3] = -np.inf                              # a whole column: out = -inf, G = 0
    tts = [tt(t, gpu) for t in terms]
    out = k.logsumexp_terms(tts, frame, rdim)
    ref, _ = o_lse.logsumexp_terms(terms, frame, rdim)
    tol = 2e-6 if dtype == torch.float32 else 1e-12
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=tol, atol=tol * 10)
    g_out = rng.standard_normal(ref.shape).astype(np_dt)
    G = k.logsumexp_terms_grad(tts, frame, rdim, out, tt(g_out, gpu))
    refG = o_lse.logsumexp_terms_grad(terms, frame, rdim, g_out)
    assert not torch.isnan(G).any()
    np.testing.assert_allclose(G.cpu().numpy(), refG, rtol=tol * 20, atol=tol * 20)
    # posterior weights sum to the upstream gradient over the eliminated variable
    np.testing.assert_allclose(G.sum(rdim).cpu().numpy()[fin], g_out[fin], rtol=tol * 50, atol=tol * 50)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_logsumexp_terms_nan_and_inf_follow_torch(gpu, dtype):
    """A NaN log-factor must poison the result (not drop out), two +inf terms give +inf (not NaN),
    forward and backward as autograd through torch.logsumexp does."""
    k = _k()
    x = torch.randn((6, 5), dtype=dtype, device=gpu)
    y = torch.randn((1, 5), dtype=dtype, device=gpu)
    x[0, 1] = float("nan")
    x[1, 2] = float("inf")
    x[2, 2] = float("inf")
    x[3, 3] = float("inf")
    x[:, 4] = -float("inf")
    out = k.logsumexp_terms([x, y], (6, 5), 0)
    xr, yr = x.clone().requires_grad_(True), y.clone()
    ref = torch.logsumexp(xr + yr, 0)
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    assert torch.equal(out[~torch.isnan(ref)], ref[~torch.isnan(ref)].detach()) or \
        torch.allclose(out[~torch.isnan(ref)], ref[~torch.isnan(ref)].detach(), rtol=1e-5)
    g = torch.ones(5, dtype=dtype, device=gpu)
    G = k.logsumexp_terms_grad([x, y], (6, 5), 0, out, g)
    ref.backward(g)
    rg = xr.grad
    # column 4 (all -inf): torch's own backward gives NaN there (exp(-inf + inf)); the kernel's 0 is
    # the convention of the sum-product (an impossible assignment has no posterior weight)
    cols = [0, 1, 2, 3]
    assert torch.equal(torch.isnan(G[:, cols]), torch.isnan(rg[:, cols]))
    ok = ~torch.isnan(rg[:, cols])
    assert torch.allclose(G[:, cols][ok], rg[:, cols][ok], rtol=1e-5, atol=1e-7)
    assert (G[:, 4] == 0).all()


def test_digamma_of_garbage_terminates_with_nan(gpu):
    """Gamma / Beta / Poisson gradients are computed before masking: a -inf or -1e30 sentinel in a
    masked-out row must give NaN, not an endless recurrence."""
    k = _k()
    for dtype in (torch.float32, torch.float64):
        conc = torch.tensor([[2.0, -float("inf"), -1e30, 0.0]], dtype=dtype, device=gpu)
        rate = torch.ones_like(conc)
        v = torch.full_like(conc, 0.5)
        g = torch.ones_like(conc)
        dv, da, db = k.dist_log_prob_grad(6, g, v, conc, rate, None, 1.0, 1, 4, (True, True, True))
        torch.cuda.synchronize()
        assert torch.isfinite(da[0, 0]) and torch.isnan(da[0, 1:]).all()


def test_fused_sumproduct_equals_torch_route(gpu):
    """ops.contract._sumproduct through pa_logsumexp_terms against its own torch route (aligned adds
    + torch.logsumexp), values and the gradients of every term (reduced to the term's shape)."""
    from pyro_amd.ops import contract as c
    rng = np.random.default_rng(4)
    ord_ = frozenset()
    shapes = {("a", "b"): (3, 4, 1, 50), ("b",): (4, 2, 50), ("a",): (3, 2, 1), ("b", "a"): (4, 3, 1, 1)}
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 3e-6)):
        res = []
        for fused in (True, False):
            c.FUSED_SUMPRODUCT = fused
            try:
                ts = [torch.tensor(rng0, dtype=dtype, device=gpu, requires_grad=True)
                      for rng0 in [np.random.default_rng(7 + i).standard_normal(sh)
                                   for i, sh in enumerate(shapes.values())]]
                terms = [c.Term(t, ids, ord_) for t, ids in zip(ts, shapes)]
                out, ids = c._sumproduct(terms, {"b"})
                assert ids == ["a"]
                grads = torch.autograd.grad(out.sum() * 1.5, ts)
                res.append((out.detach(), grads))
            finally:
                c.FUSED_SUMPRODUCT = True
        (o1, g1), (o2, g2) = res
        np.testing.assert_allclose(o1.cpu().numpy(), o2.cpu().numpy(), rtol=tol, atol=tol)
        for x, y in zip(g1, g2):
            assert x.shape == y.shape
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=tol * 20, atol=tol * 20)


# ---------------------------------------------------------------------------------------------
# Adam
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("clipped", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_adam_step(gpu, clipped, dtype):
    k = _k()
    n = 100_003      # several workgroups: the last one to finish advances the step counter
    rng = np.random.default_rng(0)
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    p = rng.standard_normal(n).astype(np_dt)
    m = np.zeros(n)
    v = np.zeros(n)
    tp, tm, tv = tt(p, gpu), torch.zeros(n, device=gpu, dtype=dtype), torch.zeros(n, device=gpu, dtype=dtype)
    step_dev = torch.zeros(2, dtype=torch.int64, device=gpu)      # [step, ticket]
    pr = p.astype(np.float64)
    for step in range(1, 6):
        g = (rng.standard_normal(n) * 20).astype(np_dt)
        tg = tt(g, gpu)
        k.adam_step(tp, tg, tm, tv, step_dev, lr=0.01, weight_decay=0.01 if clipped else 0.0,
                    clip_norm=10.0, lrd=0.999, clipped=clipped, zero_grad=True)
        pr, m, v = o_adam.adam_step(pr, g, m, v, step, 0.01, weight_decay=0.01 if clipped else 0.0,
                                    clip_norm=10.0, lrd=0.999, clipped=clipped)
        assert float(tg.abs().sum()) == 0.0
    assert step_dev.tolist() == [5, 0]
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    np.testing.assert_allclose(tp.cpu().numpy(), pr, rtol=tol, atol=tol)


def test_adam_step_publish_hands_over_the_scalar(gpu):
    """pa_adam_step_publish == pa_adam_step + pa_publish_scalar: same update, and the pinned
    mailbox / device counter advance exactly once per launch."""
    k = _k()
    n = 70_001
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal(n).astype(np.float32)
    bufs = []
    for _ in range(2):
        bufs.append((tt(p0, gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu),
                     torch.zeros(2, dtype=torch.int64, device=gpu)))
    hv = torch.zeros(1, dtype=torch.float64).pin_memory()
    hs = torch.zeros(1, dtype=torch.int64).pin_memory()
    counter = torch.tensor([40, 100], dtype=torch.int64, device=gpu)   # {Philox base, publish sequence}
    for step in range(1, 4):
        g = (rng.standard_normal(n) * 3).astype(np.float32)
        loss = torch.full((), 1.5 * step, device=gpu, dtype=torch.float32)
        (pa, ma, va, sa), (pb, mb, vb, sb) = bufs
        k.adam_step(pa, tt(g, gpu), ma, va, sa, lr=0.01)
        k.adam_step(pb, tt(g, gpu), mb, vb, sb, lr=0.01, publish=(loss, hv, hs, counter, 7))
        torch.cuda.synchronize()
        assert hs.item() == 100 + step and hv.item() == 1.5 * step      # the device-side sequence
        assert counter.tolist() == [40 + 7 * step, 100 + step]
    assert torch.equal(bufs[0][0], bufs[1][0]) and bufs[1][3].tolist() == [3, 0]


# ---------------------------------------------------------------------------------------------
# N-D site kernels (operands that broadcast along a middle dim) and the sum_to reduction
# ---------------------------------------------------------------------------------------------
def _nd_cases(rng):
    # (dist id, frame, value shape, p0 shape, p1 shape or None, positive value?)
    return [(0, (7, 50, 33), (7, 50, 33), (7, 1, 33), (7, 1, 33), False),      # mu[P,1,D] vs w[P,G,D]
            (0, (5, 40, 16), (5, 40, 16), (40, 16), (1,), False),               # leading broadcast
            (0, (3, 4, 30, 8), (3, 4, 30, 8), (3, 1, 30, 1), (4, 1, 8), False), # 4 dims, mixed
            (1, (6, 70, 9), (70, 9), (6, 70, 9), None, False),                  # Bernoulli logits
            (2, (4, 33, 5), (4, 33, 5), (4, 1, 5), None, True),                 # HalfCauchy
            (3, (4, 33, 5), (4, 33, 5), (1, 33, 1), (4, 1, 5), True),           # LogNormal
            (5, (2, 300, 3), (2, 300, 3), (2, 1, 3), None, True)]               # HalfNormal


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("use_mask", [False, True])
def test_dist_log_prob_nd(gpu, dtype, use_mask):
    from tests import oracle_backend as ob
    k = _k()
    rng = np.random.default_rng(17)
    for dist_id, shape, vs, as_, bs, positive in _nd_cases(rng):
        if dist_id == 1:
            v = (rng.uniform(size=vs) < 0.4).astype(np.float64)
            a = rng.standard_normal(as_)
        else:
            v = rng.uniform(0.2, 3.0, vs) if positive else rng.standard_normal(vs)
            a = rng.uniform(0.5, 2.0, as_) if dist_id in (2, 5) else rng.standard_normal(as_)
        b = None if bs is None else rng.uniform(0.5, 2.0, bs)
        m = (rng.uniform(size=shape[-2:]) < 0.7) if use_mask else None
        tv, ta = (torch.tensor(x, dtype=dtype, device=gpu) for x in (v, a))
        tb = None if b is None else torch.tensor(b, dtype=dtype, device=gpu)
        tm = None if m is None else torch.tensor(m, device=gpu)
        cv, ca = tv.cpu(), ta.cpu()
        cb, cm = (None if t is None else t.cpu() for t in (tb, tm))
        tol = 1e-11 if dtype == torch.float64 else 3e-5
        got = k.dist_log_prob_sum_nd(dist_id, shape, tv, ta, tb, tm, 1.7)
        ref = ob.dist_log_prob_sum_nd(dist_id, shape, cv.double(), ca.double(),
                                      None if cb is None else cb.double(), cm, 1.7)
        np.testing.assert_allclose(got.item(), ref.item(), rtol=tol, atol=tol * abs(ref.item()))
        g = torch.tensor([0.6], dtype=dtype, device=gpu)
        need = (True, True, b is not None)
        outs = k.dist_log_prob_grad_nd(dist_id, shape, g, tv, ta, tb, tm, 1.7, need)
        refs = ob.dist_log_prob_grad_nd(dist_id, shape, g.cpu().double(), cv.double(), ca.double(),
                                        None if cb is None else cb.double(), cm, 1.7, need)
        for o, r in zip(outs, refs):
            if r is None:
                assert o is None
                continue
            assert o.shape == tuple(shape) and o.is_contiguous()
            np.testing.assert_allclose(o.cpu().numpy(), r.numpy(), rtol=tol * 5,
                                       atol=tol * 5 * float(r.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("A,R,B", [(1, 64, 32000), (64, 1000, 32), (6400, 32, 1), (3, 5, 7),
                                   (1, 1, 1), (2, 100003, 1), (5, 1, 9), (1, 300000, 3),
                                   (17, 129, 257)])
def test_sum_to_nd(gpu, dtype, A, R, B):
    k = _k()
    rng = np.random.default_rng(A + R + B)
    x = rng.standard_normal((A, R, B))
    tx = torch.tensor(x, dtype=dtype, device=gpu)
    got = k.sum_to_nd(tx, A, R, B)
    again = k.sum_to_nd(tx, A, R, B)
    assert torch.equal(got, again)                       # deterministic
    ref = tx.double().sum(1)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    torch.testing.assert_close(got.double(), ref, rtol=tol, atol=tol * np.sqrt(R) * 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("A,R,B", [(64, 1000, 32), (1, 64, 32000), (3, 5, 7), (1, 300000, 3), (5, 1, 9)])
def test_sum_to_nd_pair_is_bitwise_two_single_reductions(gpu, dtype, A, R, B):
    """pa_sum_to_nd_pair: the two parameter gradients of one site through the same launches; each result
    bitwise the one pa_sum_to_nd gives on its own (same partition, same summation order)."""
    k = _k()
    rng = np.random.default_rng(A * 7 + R + B)
    x0 = torch.tensor(rng.standard_normal((A, R, B)), dtype=dtype, device=gpu)
    x1 = torch.tensor(rng.standard_normal((A, R, B)) * 3 + 1, dtype=dtype, device=gpu)
    o0, o1 = k.sum_to_nd_pair(x0, x1, A, R, B)
    assert o0.shape == o1.shape == (A, B)
    assert torch.equal(o0, k.sum_to_nd(x0, A, R, B)) and torch.equal(o1, k.sum_to_nd(x1, A, R, B))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("P,n", [(64, 32000), (1, 1), (3, 7), (512, 33), (5, 70001)])
def test_meanfield_score_against_the_restatement(gpu, dtype, P, n):
    """pa_meanfield_score: sum of the partial sums = coef * sum log Normal(z; loc, scale), gscale =
    -coef P / scale.  float64 rtol 1e-12; float32 against the float64 restatement of the SAME float32
    inputs rtol 2e-6 (the partial sums are rounded to float32 once each)."""
    from tests import oracle_backend as ob
    k = _k()
    rng = np.random.default_rng(P + n)
    loc = torch.tensor(rng.standard_normal(n), dtype=dtype, device=gpu)
    scale = torch.tensor(rng.uniform(0.05, 2.0, n), dtype=dtype, device=gpu)
    z = (loc + scale * torch.tensor(rng.standard_normal((P, n)), dtype=dtype, device=gpu)).contiguous()
    partial, gscale = k.meanfield_score(z, loc, scale, P, -0.25)
    again, _ = k.meanfield_score(z, loc, scale, P, -0.25)
    assert torch.equal(partial, again)                   # deterministic
    rp, rg = ob.meanfield_score(z.double().cpu(), loc.double().cpu(), scale.double().cpu(), P, -0.25)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert float(partial.double().sum()) == pytest.approx(float(rp.sum()), rel=tol)
    torch.testing.assert_close(gscale.double().cpu(), rg, rtol=tol, atol=0)


@pytest.mark.parametrize("Wd,B,V,H", [(64, 3000, 1024, 100), (7, 65, 128, 128), (30, 32, 256, 1)])
def test_bag_of_words_linear(gpu, Wd, B, V, H):
    """The first layer of examples/lda.py's amortised guide without the per-step histogram
    (pa_bow_linear_fwd / _bwd): the histogram images bit-exact (integer work) against the oracle's
    layout, forward and gradients against the float64 dense restatement at f32-roundoff class, and
    against torch's own f32 route."""
    from oracle import lda as o_lda
    k = _k()
    rng = np.random.default_rng(Wd + B + V + H)
    words = rng.integers(0, V, size=(Wd, B))
    words[:, 0] = 5                                              # a document of ONE word: count Wd
    W = (rng.standard_normal((H, V)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(H).astype(np.float32)
    d = rng.standard_normal((B, H)).astype(np.float32)
    tw = torch.as_tensor(words, device=gpu)
    ia, ib = k.bow_images(tw, V)
    ra, rb = o_lda.bow_images(words, V)
    assert np.array_equal(ia.float().cpu().numpy().reshape(ra.shape), ra)
    assert np.array_equal(ib.float().cpu().numpy().reshape(rb.shape), rb)
    out = k.bow_linear_fwd(ia, tt(W, gpu), tt(bias, gpu), B)
    ref = o_lda.bow_linear(words, V, W, bias)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=2e-6 * scale)
    dW = k.bow_linear_bwd(ib, tt(d, gpu), V)
    rdW, _ = o_lda.bow_linear_grad(words, V, d)
    np.testing.assert_allclose(dW.cpu().numpy(), rdW, rtol=2e-6, atol=2e-6 * np.abs(rdW).max())
    # no bias; the same through torch's f32 operators on the dense histogram
    out0 = k.bow_linear_fwd(ia, tt(W, gpu), None, B)
    counts = torch.zeros(V, B, device=gpu).scatter_add(0, tw, torch.ones(tw.shape, device=gpu))
    torch.testing.assert_close(out0, counts.t() @ tt(W, gpu).t(), rtol=1e-5, atol=1e-5 * float(scale))
    # the Sigmoid behind the layer, fused (pa_bow_linear_fwd_act / _bwd_act): y from the accumulators, the
    # gradient through it d * (1 - y) * y in the operand-split pass, which also leaves the bias gradient's sums
    y = k.bow_linear_fwd(ia, tt(W, gpu), tt(bias, gpu), B, sigmoid=True)
    ry = 1.0 / (1.0 + np.exp(-ref.astype(np.float64)))
    np.testing.assert_allclose(y.cpu().numpy(), ry, rtol=3e-6, atol=3e-6)
    yn = y.cpu().numpy().astype(np.float64)
    dh = d.astype(np.float64) * (1.0 - yn) * yn
    dW2, part = k.bow_linear_bwd(ib, tt(d, gpu), V, y_mul=y, want_bias=True)
    rdW2, _ = o_lda.bow_linear_grad(words, V, dh.astype(np.float32))
    np.testing.assert_allclose(dW2.cpu().numpy(), rdW2, rtol=3e-6, atol=3e-6 * np.abs(rdW2).max())
    db = part.sum(1).reshape(-1)[:H]
    np.testing.assert_allclose(db.cpu().numpy(), dh.sum(0), rtol=0, atol=2e-6 * np.abs(dh).sum(0).max() + 1e-6)
    dW3, part3 = k.bow_linear_bwd(ib, tt(d, gpu), V, y_mul=y, want_bias=True)
    assert torch.equal(dW2, dW3) and torch.equal(part, part3)


@pytest.mark.parametrize("B,n_in,n_out", [(20000, 100, 100), (5000, 100, 8), (4100, 8, 100), (33, 16, 32),
                                          (1000, 7, 5), (70000, 128, 128), (31, 1, 1)])
def test_tall_linear_layer(gpu, B, n_in, n_out):
    """pa_tall_linear / pa_tall_wgrad (csrc/tall.hip): forward, dx, dW and db of F.linear over a tall
    batch against the float64 restatement at f32-roundoff class, against torch's own f32 operators, and
    bitwise reproducible (fixed-order reduction of the batch dimension)."""
    from oracle import lda as o_lda
    k = _k()
    rng = np.random.default_rng(B + n_in + n_out)
    x = rng.standard_normal((B, n_in)).astype(np.float32)
    W = (rng.standard_normal((n_out, n_in)) / np.sqrt(n_in)).astype(np.float32)
    bias = rng.standard_normal(n_out).astype(np.float32)
    g = rng.standard_normal((B, n_out)).astype(np.float32)
    tx, tW, tb, tg = tt(x, gpu), tt(W, gpu), tt(bias, gpu), tt(g, gpu)
    out = k.tall_linear(tx, tW, 1, n_in, n_out, tb)                  # Wm = W^T
    ref = o_lda.tall_linear(x, W, bias)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=2e-6 * np.abs(ref).max())
    out0 = k.tall_linear(tx, tW, 1, n_in, n_out, None)
    torch.testing.assert_close(out0, tx @ tW.t(), rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()))
    dx = k.tall_linear(tg, tW, n_in, 1, n_in)                        # Wm = W
    dW, db = k.tall_wgrad(tg, tx)
    rdx, rdW, rdb = o_lda.tall_linear_grads(x, W, g)
    np.testing.assert_allclose(dx.cpu().numpy(), rdx, rtol=2e-6, atol=2e-6 * np.abs(rdx).max())
    scale = (np.abs(g).astype(np.float64).T @ np.abs(x).astype(np.float64)).max()
    np.testing.assert_allclose(dW.cpu().numpy(), rdW, rtol=0, atol=1e-6 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), rdb, rtol=0, atol=1e-6 * np.abs(g).sum(0).max())
    dW2, db2 = k.tall_wgrad(tg, tx)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    assert k.tall_wgrad(tg, tx, want_bias=False)[1] is None
    # the Sigmoid behind the layer, fused (pa_tall_linear_act / pa_tall_wgrad_act): y from the accumulators;
    # dx / dW / db of the layer BEHIND a sigmoid output y read g * (1 - y) * y in their operand loads
    y = k.tall_linear(tx, tW, 1, n_in, n_out, tb, sigmoid=True)
    np.testing.assert_allclose(y.cpu().numpy(), 1.0 / (1.0 + np.exp(-ref.astype(np.float64))), rtol=3e-6, atol=3e-6)
    yn = y.cpu().numpy().astype(np.float64)
    gh = (g.astype(np.float64) * (1.0 - yn) * yn).astype(np.float32)
    dxs = k.tall_linear(tg, tW, n_in, 1, n_in, y_mul=y)
    dWs, dbs = k.tall_wgrad(tg, tx, y_mul=y)
    sdx, sdW, sdb = o_lda.tall_linear_grads(x, W, gh)
    np.testing.assert_allclose(dxs.cpu().numpy(), sdx, rtol=3e-6, atol=3e-6 * np.abs(sdx).max())
    sscale = (np.abs(gh).astype(np.float64).T @ np.abs(x).astype(np.float64)).max()
    np.testing.assert_allclose(dWs.cpu().numpy(), sdW, rtol=0, atol=2e-6 * sscale)
    np.testing.assert_allclose(dbs.cpu().numpy(), sdb, rtol=0, atol=2e-6 * np.abs(gh).sum(0).max() + 1e-6)


def test_tall_linear_autograd_route(gpu):
    """The lazy route of ops/lazy.py (_TallLinear): an nn.Linear applied to a TallActivation gives the
    values and the parameter / input gradients of the plain torch route."""
    from pyro_amd.ops import lazy
    torch.manual_seed(0)
    B = 6000
    x = torch.randn((B, 100), device=gpu, requires_grad=True)
    lin = torch.nn.Linear(100, 8).to(gpu)
    ref = torch.sigmoid(lin(x))
    ref.square().sum().backward()
    want = [x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = lin.weight.grad = lin.bias.grad = None
    out = torch.sigmoid(lin(x.as_subclass(lazy.TallActivation)))
    assert isinstance(out, lazy.TallActivation)
    torch.testing.assert_close(out.as_subclass(torch.Tensor), ref, rtol=1e-5, atol=1e-6)
    out.square().sum().backward()
    for got, w in zip([x.grad, lin.weight.grad, lin.bias.grad], want):
        torch.testing.assert_close(got, w, rtol=1e-4, atol=1e-4 * float(w.abs().max()))


def test_sigmoid_behind_a_tall_linear_layer_rides_in_its_kernels(gpu):
    """nn.Sequential(Linear, Sigmoid, Linear, Sigmoid) over a tall batch (examples/lda.py:84-87): the layer is
    deferred until its consumer is known -- torch.sigmoid gets the fused kernels (values and every gradient as
    torch's own route gives them), anything else the plain layer, computed once."""
    import torch.nn as nn
    from pyro_amd.ops import lazy
    torch.manual_seed(1)
    B = 5000
    x0 = torch.randn((B, 100), device=gpu)
    net = nn.Sequential(nn.Linear(100, 100), nn.Sigmoid(), nn.Linear(100, 8), nn.Sigmoid()).to(gpu)

    def run(tall):
        for p in net.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        out = net(x.as_subclass(lazy.TallActivation) if tall else x)
        (out.as_subclass(torch.Tensor) ** 2).sum().backward()
        return [out.detach().as_subclass(torch.Tensor), x.grad] + [p.grad.clone() for p in net.parameters()]

    want = run(False)
    launched = []
    k = _k()
    real = k.tall_linear
    k.tall_linear = lambda *a, **kw: (launched.append(kw.get("sigmoid", False) or kw.get("y_mul") is not None),
                                      real(*a, **kw))[1]
    try:
        got = run(True)
    finally:
        k.tall_linear = real
    assert launched and all(launched)                      # every launch carried the activation
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()))
    # a deferred layer that meets anything but a sigmoid is the plain layer, launched once
    lin = net[0]
    d = lazy.TallActivation.__torch_function__(torch.nn.functional.linear, (),
                                               (x0.as_subclass(lazy.TallActivation), lin.weight, lin.bias))
    assert isinstance(d, lazy.DeferredLinear) and d.shape == (B, 100) and d.dim() == 2
    r1, r2 = torch.tanh(d), d + 1.0
    ref = torch.nn.functional.linear(x0, lin.weight, lin.bias)
    torch.testing.assert_close(r1.as_subclass(torch.Tensor), torch.tanh(ref), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(r2.as_subclass(torch.Tensor), ref + 1.0, rtol=1e-4, atol=1e-5)
    assert d._plain is not None
    lazy.FUSE_ACTIVATION["on"] = False
    try:
        plain = run(True)
    finally:
        lazy.FUSE_ACTIVATION["on"] = True
    for a, b in zip(plain, want):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()))


def test_word_histogram_is_recognised_in_guide_text(gpu):
    """examples/lda.py's guide text verbatim -- zeros(V, B).scatter_add(0, data, ones) then
    predictor(counts.transpose(0, 1)) -- under watch_histograms(): from the second sighting of the
    corpus the first Linear runs on the bag-of-words kernels; values and parameter gradients equal the
    dense route; every other use of the histogram materialises it."""
    import torch.nn as nn
    from pyro_amd import kernels as k
    from pyro_amd.ops import lazy
    g = torch.Generator(device=gpu).manual_seed(0)
    V, B, Wd = 256, 500, 40
    data = torch.randint(0, V, (Wd, B), device=gpu, generator=g)
    torch.manual_seed(0)
    predictor = nn.Sequential(nn.Linear(V, 30), nn.Sigmoid(), nn.Linear(30, 4), nn.Softmax(dim=-1)).to(gpu)

    def guide_body():
        counts = torch.zeros(V, data.shape[1], device=gpu).scatter_add(0, data, torch.ones(data.shape, device=gpu))
        return counts, predictor(counts.transpose(0, 1))

    calls = []
    orig = k.bow_linear_fwd
    k.bow_linear_fwd = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
    try:
        outs, grads = [], []
        for rep in range(3):
            for p in predictor.parameters():
                p.grad = None
            with lazy.watch_histograms():
                counts, y = guide_body()
            assert isinstance(counts, lazy.DeferredCounts) and tuple(counts.shape) == (V, B)
            (y * torch.arange(4, device=gpu)).sum().backward()
            outs.append(y.detach().clone())
            grads.append([p.grad.clone() for p in predictor.parameters()])
        assert len(calls) == 2                    # first sighting: the dense route; then the kernels
        for o, gr in zip(outs[1:], grads[1:]):
            torch.testing.assert_close(o, outs[0], rtol=1e-5, atol=1e-6)
            for a, b in zip(gr, grads[0]):
                torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-6 * float(b.abs().max()) + 1e-9)
        # another use: behaves as the dense tensor
        with lazy.watch_histograms():
            counts, _ = guide_body()
            assert float(counts.sum()) == Wd * B and torch.equal(counts.t().sum(1), torch.full((B,), float(Wd), device=gpu))
            # an in-place update before the scatter: not the pattern, runs as written
            z = torch.zeros(V, B, device=gpu)
            z += 1
            assert isinstance(z.scatter_add(0, data, torch.ones(data.shape, device=gpu)), torch.Tensor)
    finally:
        k.bow_linear_fwd = orig


@pytest.mark.parametrize("B,M,N", [(100_000, 100, 100), (5000, 8, 100), (33, 128, 1), (4097, 37, 128)])
def test_tall_skinny_weight_gradient_product(gpu, B, M, N):
    """pa_tsgemm_tn: a[B, M].T @ x[B, N] against float64 (f32-roundoff class: exact bf16x3 splits,
    f32 accumulation over the chunks, fp64 over the partials) and bitwise reproducible."""
    k = _k()
    rng = np.random.default_rng(B + M + N)
    a = rng.standard_normal((B, M)).astype(np.float32)
    x = (rng.standard_normal((B, N)) * np.exp(rng.uniform(-3, 3, size=(1, N)))).astype(np.float32)
    out = k.tsgemm_tn(tt(a, gpu), tt(x, gpu))
    ref = a.astype(np.float64).T @ x.astype(np.float64)
    bound = (np.abs(a).astype(np.float64).T @ np.abs(x).astype(np.float64))
    assert np.all(np.abs(out.cpu().numpy() - ref) <= 4e-7 * bound * max(1.0, np.sqrt(B) / 30) + 1e-30)
    assert torch.equal(out, k.tsgemm_tn(tt(a, gpu), tt(x, gpu)))


def test_layers_after_the_bag_of_words_layer_take_the_tall_weight_gradient(gpu):
    """nn.Sequential(Linear, Sigmoid, Linear, Sigmoid, Linear, Sigmoid, Softmax) on a large batch of
    histograms (examples/lda.py:76-92): parameter gradients equal the dense torch route; the second and
    third Linear take the tall-batch kernels of csrc/tall.hip (pa_tall_linear / pa_tall_wgrad)."""
    import torch.nn as nn
    from pyro_amd import kernels as k
    from pyro_amd.ops import lazy
    g = torch.Generator(device=gpu).manual_seed(1)
    V, B, Wd = 256, 6000, 20
    data = torch.randint(0, V, (Wd, B), device=gpu, generator=g)
    torch.manual_seed(1)
    predictor = nn.Sequential(nn.Linear(V, 100), nn.Sigmoid(), nn.Linear(100, 100), nn.Sigmoid(),
                              nn.Linear(100, 8), nn.Sigmoid(), nn.Softmax(dim=-1)).to(gpu)
    wts = torch.randn(B, 8, device=gpu, generator=g)
    calls = []
    orig = k.tall_wgrad
    k.tall_wgrad = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
    try:
        res = []
        for on in (False, True, True):
            lazy.ENABLED["on"] = on
            for p in predictor.parameters():
                p.grad = None
            with lazy.watch_histograms():
                counts = torch.zeros(V, B, device=gpu).scatter_add(0, data, torch.ones(data.shape, device=gpu))
                y = predictor(counts.transpose(0, 1))
            (y * wts).sum().backward()
            res.append((y.detach().clone(), [p.grad.clone() for p in predictor.parameters()]))
    finally:
        lazy.ENABLED["on"] = True
        k.tall_wgrad = orig
    assert len(calls) == 2                    # (third run: both later Linear layers)
    torch.testing.assert_close(res[2][0], res[0][0], rtol=1e-5, atol=1e-7)
    for a, b in zip(res[2][1], res[0][1]):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-6 * float(b.abs().max()) + 1e-10)



# ---- label moments: the (y - 1/2) . l part of the log-likelihood as a dot product with data moments ----
@pytest.mark.parametrize("N,D", [(1, 1), (63, 7), (5000, 32), (100_003, 20)])
def test_glm_label_moments_equal_the_oracle(gpu, N, D):
    k = _k()
    rng = np.random.default_rng(N + D)
    X = (rng.standard_normal((N, D)) * np.exp(rng.uniform(-3, 3, (1, D)))).astype(np.float32)
    y = (rng.uniform(size=N) < 0.3).astype(np.float32)
    got = k.glm_label_moments(tt(X, gpu), tt(y, gpu)).cpu().numpy()
    ref = o_glm.label_moments(X, y)
    scale = np.abs(np.asarray(X, np.float64)).sum(0).max() + 1.0
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13 * scale)
    again = k.glm_label_moments(tt(X, gpu), tt(y, gpu)).cpu().numpy()
    assert np.array_equal(got, again)                        # fixed summation order


@pytest.mark.parametrize("N,D,P,use_bias", [(4099, 32, 64, True), (70000, 17, 100, True), (2048, 32, 33, False)])
def test_glm_planes_with_label_moments(gpu, N, D, P, use_bias):
    """pa_glm_bernoulli_planes_fwd_bwd with the moments of (X, y): the gradient outputs are bit for bit
    those of the kernel that sums the label-linear term itself (the gradient never takes the moments
    route), the log-likelihood agrees with it and with the float64 oracle."""
    k = _k()
    rng = np.random.default_rng(N + P)
    X = rng.standard_normal((N, D)).astype(np.float32)
    w = (rng.standard_normal((P, D)) / np.sqrt(D)).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32) if use_bias else None
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    tX, ty, tw = tt(X, gpu), tt(y, gpu), tt(w, gpu)
    tb = tt(b, gpu) if use_bias else None
    planes = k.glm_pack_planes(tX, fmt=k.GLM_PLANES_F16X2)
    plain = k.glm_bernoulli_planes_fwd_bwd(planes, ty, tw, tb, 1.5, N, D)
    mom = k.glm_label_moments(tX, ty)
    lin = k.glm_bernoulli_planes_fwd_bwd(planes, ty, tw, tb, 1.5, N, D, moments=mom)
    assert torch.equal(plain[1], lin[1]) and torch.equal(plain[2], lin[2])
    ref = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.5)
    sc = max(1.0, float(np.abs(ref[0]).max()))
    np.testing.assert_allclose(lin[0].cpu().numpy(), ref[0], rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(lin[0].cpu().numpy(), plain[0].cpu().numpy(), rtol=2e-6, atol=2e-6 * sc)
    e_lin = float(np.abs(lin[0].cpu().numpy() - ref[0]).max())
    e_plain = float(np.abs(plain[0].cpu().numpy() - ref[0]).max())
    assert e_lin <= 2 * e_plain + 1e-6 * sc, (e_lin, e_plain)


@pytest.mark.parametrize("N,D,P", [(1, 3, 65), (33, 32, 128), (5000, 32, 129), (4099, 20, 192), (70001, 32, 256),
                                   (3001, 12, 300), (2500, 32, 513), (31, 1, 257), (400000, 32, 200)])
@pytest.mark.parametrize("use_bias,with_moments", [(True, False), (False, True), (True, True)])
def test_glm_planes_many_particles_in_one_pass(gpu, N, D, P, use_bias, with_moments):
    """More than 64 particles / chains (NUTS on a model: P = the number of chains): 65..128 run as 2 x 4 waves
    (128 per pass over the image), more as 1 x 8 (256 per pass) instead of ceil(P / 64) passes of the 2 x 2
    geometry.  Against the float64 oracle at the kernel's usual tolerance, and against the 2 x 2 passes of the
    same launch (pa_glm_planes_tune(11, 0)) to f32 rounding of the per-workgroup partial sums."""
    k = _k()
    rng = np.random.default_rng(N + 7 * D + P)
    X = (rng.standard_normal((N, D)) * np.exp(rng.uniform(-1, 1, (1, D)))).astype(np.float32)
    w = (rng.standard_normal((P, D)) / np.sqrt(D) * np.exp(rng.uniform(-2, 2, (P, 1)))).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32) if use_bias else None
    y = (rng.uniform(size=N) < 0.4).astype(np.float32)
    tX, ty, tw = tt(X, gpu), tt(y, gpu), tt(w, gpu)
    tb = tt(b, gpu) if use_bias else None
    planes = k.glm_pack_planes(tX, fmt=k.GLM_PLANES_F16X2)
    mom = k.glm_label_moments(tX, ty) if with_moments else None
    k.glm_planes_tune(13, 0)             # (the wide geometries at every N: by default only from N ~ 4e5 on)
    try:
        wide = k.glm_bernoulli_planes_fwd_bwd(planes, ty, tw, tb, 2.0, N, D, moments=mom)
        again = k.glm_bernoulli_planes_fwd_bwd(planes, ty, tw, tb, 2.0, N, D, moments=mom)
    finally:
        k.glm_planes_tune(0, 0)
    k.glm_planes_tune(11, 0)
    try:
        narrow = k.glm_bernoulli_planes_fwd_bwd(planes, ty, tw, tb, 2.0, N, D, moments=mom)
    finally:
        k.glm_planes_tune(0, 0)
    ref = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 2.0)
    sc = max(1.0, float(np.abs(ref[0]).max()))
    gs = max(1.0, float(np.abs(ref[1]).max()))
    for got in (wide, narrow):
        np.testing.assert_allclose(got[0].cpu().numpy(), ref[0], rtol=2e-5, atol=2e-5 * sc)
        np.testing.assert_allclose(got[1].cpu().numpy(), ref[1], rtol=2e-5, atol=2e-5 * max(gs, N ** 0.5))
        np.testing.assert_allclose(got[2].cpu().numpy(), ref[2], rtol=2e-5, atol=2e-5 * max(1.0, N ** 0.5))
    np.testing.assert_allclose(wide[0].cpu().numpy(), narrow[0].cpu().numpy(), rtol=3e-6, atol=3e-6 * sc)
    np.testing.assert_allclose(wide[1].cpu().numpy(), narrow[1].cpu().numpy(), rtol=3e-6, atol=3e-6 * max(gs, N ** 0.5))
    assert all(torch.equal(a, c) for a, c in zip(wide, again))               # fixed summation order
    if N >= 48 * 256 * 32:               # (>= 48 row tiles per workgroup on 256 CUs: the default's own choice)
        default = k.glm_bernoulli_planes_fwd_bwd(planes, ty, tw, tb, 2.0, N, D, moments=mom)
        assert all(torch.equal(a, c) for a, c in zip(wide, default))


def test_glm_label_moments_follow_the_tensors(gpu):
    """kernels.glm_label_moments_of: cached per (image of X, y); labels written in place are picked up
    by the revalidate hook (the buffer a captured step reads is refreshed, not replaced)."""
    k = _k()
    rng = np.random.default_rng(3)
    N, D, P = 5000, 32, 64
    X = tt(rng.standard_normal((N, D)).astype(np.float32), gpu)
    y = tt((rng.uniform(size=N) < 0.5).astype(np.float32), gpu)
    w = tt((rng.standard_normal((P, D)) * 0.2).astype(np.float32), gpu)
    assert k.glm_label_moments_of(X, y) is None                    # no image yet
    k.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)
    k.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)              # second sighting: image + moments
    m = k.glm_label_moments_of(X, y)
    assert m is not None and m is k.glm_label_moments_of(X, y)
    ref = o_glm.label_moments(X.cpu().numpy(), y.cpu().numpy())
    np.testing.assert_allclose(m.cpu().numpy(), ref, rtol=1e-12, atol=1e-9)
    y.copy_(1.0 - y)                                               # new labels in the same tensor
    assert k.revalidate_pending()
    k.glm_planes_revalidate()
    assert not k.revalidate_pending()
    assert k.glm_label_moments_of(X, y) is m                       # same buffer, new content
    np.testing.assert_allclose(m.cpu().numpy(), -ref, rtol=1e-12, atol=1e-9)
    got = k.glm_bernoulli_fwd_bwd(X, y, w, None, None, 1.0)
    want = o_glm.glm_bernoulli_fwd_bwd(X.cpu().numpy(), y.cpu().numpy(), w.cpu().numpy(), None, None, 1.0)
    np.testing.assert_allclose(got[0].cpu().numpy(), want[0], rtol=2e-5)
    # several label vectors over one X (one-vs-rest): the moments handed out for the first one stay the
    # first one's -- a captured step has their ADDRESS baked in -- however many others follow
    addr, keep = m.data_ptr(), []
    for j in range(6):
        yj = tt((rng.uniform(size=N) < 0.3 + 0.05 * j).astype(np.float32), gpu)
        keep.append(yj)
        mj = k.glm_label_moments_of(X, yj)
        assert mj is not None and mj.data_ptr() != addr
    again = k.glm_label_moments_of(X, y)
    assert again is m and again.data_ptr() == addr
    np.testing.assert_allclose(m.cpu().numpy(), -ref, rtol=1e-12, atol=1e-9)
