"""The GLM site as torch dispatcher ops (csrc/torch_ops.cpp + pyro_amd/ops/torch_library.py): same
numbers as the ctypes binding, autograd through pyro_amd::glm_chain, and torch.jit.trace records the
site as a graph node and replays it at other parameter values (pyro/ops/jit.py:104-109 is the
reference-side use: a traced loss over the unconstrained parameters)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _setup():
    from pyro_amd import kernels
    from pyro_amd.ops import torch_library
    assert torch_library.available(), "libpyro_amd_torch.so missing: python -m pyro_amd.csrc.build"
    return kernels, torch_library


@pytest.mark.parametrize("fmt", ["f16x2", "bf16x3"])
def test_ops_equal_the_ctypes_binding(gpu, fmt):
    k, _ = _setup()
    N, D, P = 5000, 32, 64
    g = torch.Generator(device="cpu").manual_seed(1)
    X = torch.randn((N, D), generator=g).to(gpu)
    y = (torch.rand((N,), generator=g) < 0.5).float().to(gpu)
    w = (0.3 * torch.randn((P, D), generator=g)).to(gpu)
    b = torch.randn((P,), generator=g).to(gpu)
    f = k.GLM_PLANES_F16X2 if fmt == "f16x2" else k.GLM_PLANES_BF16X3
    planes = torch.ops.pyro_amd.glm_pack_planes(X, f)
    ref_planes = k.glm_pack_planes(X, fmt=f)
    assert torch.equal(planes[:ref_planes.numel()], ref_planes)
    out = torch.ops.pyro_amd.glm_bernoulli_planes(planes, y, w, b, 2.0, N, D, f)
    ref = k.glm_bernoulli_planes_fwd_bwd(ref_planes, y, w, b, 2.0, N, D)
    for o, r in zip(out, ref):
        assert torch.equal(o, r)
    k.glm_set_planes_mode(k.GLM_PLANES_OFF)
    try:
        ref2 = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 2.0)
    finally:
        k.glm_set_planes_mode(k.GLM_PLANES_AUTO)
    out2 = torch.ops.pyro_amd.glm_bernoulli(X, y, w, b, None, 2.0)
    for o, r in zip(out2, ref2):
        assert torch.equal(o, r)
    gout = torch.randn((P,), generator=g).to(gpu)
    dw, db = torch.ops.pyro_amd.glm_chain(gout, out[1], out[2])
    torch.testing.assert_close(dw, gout[:, None] * out[1])
    torch.testing.assert_close(db, gout * out[2])


def test_autograd_through_the_ops(gpu):
    """d sum_p c_p ll_p / d (w, b) through register_autograd equals the float64 oracle's."""
    from oracle import glm as o_glm
    k, tl = _setup()
    N, D, P = 3000, 20, 40
    rng = np.random.default_rng(2)
    X = rng.standard_normal((N, D)).astype(np.float32)
    y = (rng.uniform(size=N) < 0.5).astype(np.float32)
    w = (rng.standard_normal((P, D)) * 0.3).astype(np.float32)
    b = rng.standard_normal(P).astype(np.float32)
    c = rng.standard_normal(P).astype(np.float32)
    tX, ty = torch.as_tensor(X, device=gpu), torch.as_tensor(y, device=gpu)
    tw = torch.as_tensor(w, device=gpu).requires_grad_()
    tb = torch.as_tensor(b, device=gpu).requires_grad_()
    tc = torch.as_tensor(c, device=gpu)
    for _ in range(2):                     # second call: the plane image of tX exists
        tw.grad = tb.grad = None
        ll = tl.glm_bernoulli_ll(tX, ty, tw, tb, None, 1.5)
        (ll * tc).sum().backward()
        rll, rgw, rgb = o_glm.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.5)
        np.testing.assert_allclose(ll.detach().cpu().numpy(), rll, rtol=2e-5)
        np.testing.assert_allclose(tw.grad.cpu().numpy(), c[:, None] * rgw, rtol=2e-4, atol=2e-5 * N ** 0.5)
        np.testing.assert_allclose(tb.grad.cpu().numpy(), c * rgb, rtol=2e-4, atol=2e-5 * N ** 0.5)


def test_autograd_without_a_bias(gpu):
    """b = None (a model without an intercept, e.g. the NUTS logistic regression of
    tests/test_mcmc_gpu.py): the backward returns no gradient for the absent input."""
    k, tl = _setup()
    N, D, P = 2000, 3, 32
    g = torch.Generator(device="cpu").manual_seed(5)
    X = torch.randn((N, D), generator=g).to(gpu)
    y = (torch.rand((N,), generator=g) < 0.5).float().to(gpu)
    w = torch.randn((P, D), generator=g).to(gpu).requires_grad_()
    ll = tl.glm_bernoulli_ll(X, y, w, None)
    (gw,) = torch.autograd.grad(ll.sum(), [w])
    lg = w.detach().double() @ X.double().t()
    ref = ((y.double() - torch.sigmoid(lg)) @ X.double())
    torch.testing.assert_close(gw.double(), ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


def _neg_elbo(X, y, eps_w, eps_b):
    """-ELBO of SURVEY 8(d)'s model under a mean-field Normal guide as a function of the guide's four
    unconstrained tensors (AutoNormal: scale = softplus-free 'scales' parameter is constrained positive,
    the golden file stores the constrained values); the observed site through the dispatcher op."""
    from pyro_amd.distributions import fused
    P = eps_w.shape[0]

    def loss(loc_w, scale_w, loc_b, scale_b):
        w = loc_w + scale_w * eps_w                       # [P, D]
        b = loc_b + scale_b * eps_b                       # [P]
        ll = fused.glm_bernoulli_ll(X, y, w, b)
        c = -0.5 * math.log(2 * math.pi)
        lp = (c - 0.5 * w * w).sum(-1) + (c - 0.5 * b * b)
        lq = (c - torch.log(scale_w) - 0.5 * eps_w * eps_w).sum(-1) + (c - torch.log(scale_b) - 0.5 * eps_b * eps_b)
        return -(ll + lp - lq).sum() / P
    return loss


def test_jit_trace_records_the_site_and_replays(gpu):
    """torch.jit.trace of the loss: the graph holds pyro_amd::glm_bernoulli_planes (the image of X
    exists after the first eager call), the traced function equals the golden loss of
    tests/golden/logreg_f32.npz (the reference's Trace_ELBO on the same inputs and noise) at both
    recorded parameter sets -- i.e. it replays at NEW parameter values -- and its gradient with
    respect to the guide's locations equals the golden gradient."""
    k, _ = _setup()
    g = np.load(os.path.join(GOLD, "logreg_f32.npz"))
    f32 = dict(dtype=torch.float32, device=gpu)
    X = torch.as_tensor(g["X"], **f32)
    y = torch.as_tensor(g["y"], **f32)
    P = int(g["P"])
    assert P >= 33

    def params(tag):
        return [torch.as_tensor(g["%s/AutoNormal.%s" % (tag, n)], **f32).clone()
                for n in ("locs.w", "scales.w", "locs.b", "scales.b")]

    def eps(tag):
        return (torch.as_tensor(g[tag + "/000"], **f32).reshape(P, -1),
                torch.as_tensor(g[tag + "/001"], **f32).reshape(P))

    loss = _neg_elbo(X, y, *eps("eps"))
    p1 = params("params")
    eager = loss(*p1)
    eager2 = loss(*p1)                                   # second sighting of X: plane image from now on
    np.testing.assert_allclose(float(eager), float(g["loss"]), rtol=2e-4)
    np.testing.assert_allclose(float(eager2), float(g["loss"]), rtol=2e-4)
    traced = torch.jit.trace(loss, tuple(p1), check_trace=False)
    graph = str(traced.graph)
    assert "pyro_amd::glm_bernoulli_planes" in graph, [ln[ln.index("= pyro_amd::"):][:160] for ln in graph.split("\n") if "= pyro_amd::" in ln]
    np.testing.assert_allclose(float(traced(*p1)), float(g["loss"]), rtol=2e-4)
    # gradient of the traced function (the backward node is pyro_amd::glm_chain)
    q = [t.clone().requires_grad_() for t in p1]
    traced(*q).backward()
    np.testing.assert_allclose(q[0].grad.cpu().numpy(), g["grads/AutoNormal.locs.w"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(float(q[2].grad), float(g["grads/AutoNormal.locs.b"]), rtol=2e-3, atol=1e-3)
    # new parameter values, same noise: against the eager evaluation at those values
    p2 = params("params2")
    np.testing.assert_allclose(float(traced(*p2)), float(loss(*p2)), rtol=1e-6)
    # the golden second evaluation used other noise: a second trace with it
    loss_b = _neg_elbo(X, y, *eps("eps2"))
    np.testing.assert_allclose(float(torch.jit.trace(loss_b, tuple(p2), check_trace=False)(*p2)),
                               float(g["loss2"]), rtol=2e-4)


def test_op_workspace_stays_alive_while_its_finalize_phase_is_pending(gpu):
    """Inside a chained tail the launcher only RECORDS the finalize phase, which reads the partial
    records in the op's workspace when the chain is flushed.  The workspace is an op OUTPUT and the
    Python wrapper parks it (and the outputs the phase writes) in the chain's keep-list: the caching
    allocator cannot hand the block to anyone else before the flush."""
    from pyro_amd import _lib
    k, tl = _setup()
    N, D, P = 4096, 32, 64
    g = torch.Generator(device="cpu").manual_seed(9)
    X = torch.randn((N, D), generator=g).to(gpu)
    y = (torch.rand((N,), generator=g) < 0.5).float().to(gpu)
    w = (0.3 * torch.randn((P, D), generator=g)).to(gpu).requires_grad_()
    b = torch.randn((P,), generator=g).to(gpu).requires_grad_()
    for _ in range(2):
        ref = tl.glm_bernoulli_ll(X, y, w, b)           # (second call: plane image)
    (rw, rb) = torch.autograd.grad(ref.sum(), [w, b])
    with k.chain_recording(gpu) as rec:
        ll = tl.glm_bernoulli_ll(X, y, w, b)
        assert _lib.load().pa_chain_pending() == 1
        kept = [t for t in k._CHAIN["keep"] if t.dtype == torch.uint8]
        assert kept, "the op's workspace is not in the chain's keep-list"
        ws_ptr = kept[-1].data_ptr()
        # blocks of the workspace's size requested now come from elsewhere
        others = [torch.empty((kept[-1].numel(),), dtype=torch.uint8, device=gpu) for _ in range(8)]
        assert all(o.data_ptr() != ws_ptr for o in others)
        assert _lib.load().pa_chain_pending() == 1      # (allocations do not flush)
    assert rec.stats == (1, 1)
    assert torch.equal(ll, ref)
    (gw, gb) = torch.autograd.grad(ll.sum(), [w, b])
    assert torch.equal(gw, rw) and torch.equal(gb, rb)


# ---- the typed C++ ops of round 6 (csrc/torch_ops.cpp): each against a plain torch float64 statement of the
#      operation it names (tolerances: f32 kernels 2e-5 relative unless said otherwise) -------------------------------
def _ops():
    from pyro_amd.ops import torch_library
    assert torch_library.available()
    return torch.ops.pyro_amd


def test_typed_dist_and_multi_log_prob_sum(gpu):
    ops = _ops()
    g = torch.Generator(device=gpu).manual_seed(0)
    v = torch.randn((6, 50), device=gpu, generator=g)
    loc = torch.randn((50,), device=gpu, generator=g)
    scale = torch.rand((1,), device=gpu, generator=g) + 0.5
    mask = torch.rand((6, 50), device=gpu, generator=g) < 0.7
    rs, tot = ops.dist_log_prob_sum(0, v, loc, scale, mask, 2.5)
    want = (torch.distributions.Normal(loc.double(), scale.double()).log_prob(v.double()) * 2.5 * mask).sum(-1)
    torch.testing.assert_close(rs.double(), want, rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(tot.double(), want.sum(), rtol=2e-5, atol=1e-4)
    # the ELBO assembly: three sites, one total (Normal, HalfCauchy = 2, Exponential = 4: include/pyro_amd.h PA_DIST_*)
    hc = torch.rand((4, 3), device=gpu, generator=g) + 0.1
    ex = torch.rand((7,), device=gpu, generator=g) + 0.1
    s_hc, r_ex = torch.tensor([0.7], device=gpu), torch.tensor([1.3], device=gpu)
    total = ops.multi_log_prob_sum([0, 2, 4], [v, hc, ex], [loc, s_hc, r_ex], [scale, None, None], [1.0, -1.0, 0.5], 2.0)
    want = 2.0 * (torch.distributions.Normal(loc.double(), scale.double()).log_prob(v.double()).sum()
                  - torch.distributions.HalfCauchy(s_hc.double()).log_prob(hc.double()).sum()
                  + 0.5 * torch.distributions.Exponential(r_ex.double()).log_prob(ex.double()).sum())
    torch.testing.assert_close(total.double(), want, rtol=2e-5, atol=1e-4)


def test_typed_meanfield_and_mvn_draws(gpu):
    """The guide draws: z = loc + softplus(rho) eps with eps the keyed Philox numbers -- the SAME numbers the
    package's own path (kernels.philox_normal at the same seed / offset) hands out; AutoMultivariateNormal's z and
    log q against torch.distributions.MultivariateNormal at the kernel's own eps."""
    from pyro_amd import kernels
    ops = _ops()
    P, seed = 16, 1234
    loc = [torch.linspace(-1, 1, 5, device=gpu), torch.tensor([0.3], device=gpu)]
    rho = [torch.linspace(-2, 0.5, 5, device=gpu), torch.tensor([-1.0], device=gpu)]
    z, scale, eps = ops.meanfield_normal_sample(loc, rho, P, seed, [0, 64], None)
    for i, off in enumerate((0, 64)):
        n = loc[i].numel()
        assert z[i].shape == (P, n) and eps[i].shape == (P, n)
        torch.testing.assert_close(scale[i], torch.nn.functional.softplus(rho[i]), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(z[i], loc[i] + scale[i] * eps[i], rtol=1e-6, atol=1e-7)
        want = kernels.philox_normal((P, n), torch.float32, gpu, seed, off)
        assert torch.equal(eps[i], want)
    z2, _, _ = ops.meanfield_normal_sample(loc, rho, P, seed, [0, 64], None)
    assert torch.equal(z[0], z2[0])
    n = 6
    g = torch.Generator(device=gpu).manual_seed(1)
    mloc, mrho = torch.randn((n,), device=gpu, generator=g), torch.randn((n,), device=gpu, generator=g) * 0.3
    A = torch.randn((n, n), device=gpu, generator=g) * 0.4
    e, zz, logq = ops.mvn_tril_sample(mloc, mrho, A, P, seed, 128, None)
    L = (torch.nn.functional.softplus(mrho)[:, None] * (torch.tril(A, -1) + torch.eye(n, device=gpu))).double()
    torch.testing.assert_close(zz.double(), mloc.double() + e.double() @ L.T, rtol=1e-5, atol=1e-5)
    mvn = torch.distributions.MultivariateNormal(mloc.double(), scale_tril=L)
    torch.testing.assert_close(logq.double(), mvn.log_prob(zz.double()), rtol=1e-5, atol=1e-4)


def test_typed_exp_site_and_its_backward(gpu):
    ops = _ops()
    u = torch.linspace(-3, 2, 24, device=gpu).reshape(4, 6).contiguous()
    value, ld = ops.exp_site(u, 0.5)
    torch.testing.assert_close(value, 0.5 + u.exp(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ld, -u.sum(-1), rtol=1e-6, atol=1e-6)      # the Delta site's density: -log|dv/du|
    gv, gl = torch.randn_like(value), torch.randn((4,), device=gpu)
    gu = ops.exp_site_bwd(value, gv, gl, 0.5)
    torch.testing.assert_close(gu, gv * u.exp() - gl[:, None], rtol=1e-5, atol=1e-6)


def test_typed_logsumexp_terms_and_logchain(gpu):
    ops = _ops()
    g = torch.Generator(device=gpu).manual_seed(2)
    a = torch.randn((5, 1, 7), device=gpu, generator=g)
    b = torch.randn((1, 4, 7), device=gpu, generator=g)
    c = torch.randn((5, 4, 1), device=gpu, generator=g)
    out = ops.logsumexp_terms([a, b, c], [5, 4, 7], 2)
    torch.testing.assert_close(out.double(), torch.logsumexp((a + b + c).double(), -1), rtol=1e-5, atol=1e-5)
    out0 = ops.logsumexp_terms([a, b], [5, 4, 7], 0)
    torch.testing.assert_close(out0.double(), torch.logsumexp((a + b).double(), 0), rtol=1e-5, atol=1e-5)
    # a chain of T variables with K states: log Z by the forward recursion, its gradient by autograd
    B, T, K = 9, 6, 4
    un = torch.randn((B, T, K), device=gpu, generator=g)
    pw = torch.randn((1, T - 1, K, K), device=gpu, generator=g)
    log_z, gu, gp = ops.logchain(un, pw)
    und, pwd = un.double().requires_grad_(True), pw.double().requires_grad_(True)
    alpha = und[:, 0]
    for t in range(1, T):
        alpha = torch.logsumexp(alpha[:, :, None] + pwd[0, t - 1][None], 1) + und[:, t]
    want = torch.logsumexp(alpha, -1)
    torch.testing.assert_close(log_z.double(), want.detach(), rtol=1e-5, atol=1e-5)
    want.sum().backward()
    torch.testing.assert_close(gu.double(), und.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gp.double().sum(0, keepdim=True), pwd.grad, rtol=1e-4, atol=1e-5)


def test_typed_lda_factor_and_tall_linear(gpu):
    from pyro_amd import kernels
    ops = _ops()
    g = torch.Generator(device=gpu).manual_seed(3)
    Wd, B, T, V = 12, 40, 8, 64
    words = torch.randint(0, V, (Wd, B), device=gpu, generator=g)
    lt = torch.log_softmax(torch.randn((B, T), device=gpu, generator=g), -1)
    lp = torch.log_softmax(torch.randn((T, V), device=gpu, generator=g), -1)
    index = kernels.lda_build_index(words, V)
    assert index is not None
    out_doc, g_theta, g_phi = ops.lda_factor_indexed(words, index.view(torch.uint8).reshape(-1), lt, lp)
    ltd, lpd = lt.double().requires_grad_(True), lp.double().requires_grad_(True)
    # out[d] = sum_w logsumexp_t(log_theta[d, t] + log_phi[t, words[w, d]])
    terms = ltd[None, :, :] + lpd.t()[words]              # [Wd, B, T]
    want = torch.logsumexp(terms, -1).sum(0)
    torch.testing.assert_close(out_doc.double(), want.detach(), rtol=1e-5, atol=1e-4)
    want.sum().backward()
    torch.testing.assert_close(g_theta.double(), ltd.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(g_phi.double(), lpd.grad, rtol=1e-4, atol=1e-4)
    # F.linear + Sigmoid over a tall batch, and the input gradient through the previous Sigmoid
    Bt, R, C = 5000, 24, 40
    G = torch.randn((Bt, R), device=gpu, generator=g)
    W = torch.randn((C, R), device=gpu, generator=g) * 0.3
    bias = torch.randn((C,), device=gpu, generator=g)
    Y = ops.tall_linear_act(G, W, bias, None, True, True)
    want = torch.sigmoid(G.double() @ W.double().t() + bias.double())
    torch.testing.assert_close(Y.double(), want, rtol=2e-5, atol=2e-6)
    ymul = torch.rand((Bt, C), device=gpu, generator=g)
    Gy = torch.randn((Bt, C), device=gpu, generator=g)
    dX = ops.tall_linear_act(Gy, W, None, ymul, False, False)          # dx = (g (1 - y) y) W
    want = (Gy.double() * (1 - ymul.double()) * ymul.double()) @ W.double()
    torch.testing.assert_close(dX.double(), want, rtol=2e-5, atol=2e-5)


def test_typed_nuts_round_is_the_ctypes_round(gpu):
    """pyro_amd::nuts_tree_run_advance on a copy of a span's state == kernels.NutsTree.run_advance (ctypes) on the
    original, bit for bit, forty rounds of a Gaussian potential (chains finish trees and start their next ones)."""
    import copy

    from pyro_amd import kernels
    ops = _ops()
    C, D = 8, 5
    g = torch.Generator(device=gpu).manual_seed(4)

    def fresh():
        gg = torch.Generator(device=gpu).manual_seed(4)
        z = torch.randn((C, D), device=gpu, generator=gg) * 0.3
        pe, grad = 0.5 * (z * z).sum(-1), z.clone()
        step = torch.full((C,), 0.2, device=gpu)
        tree = kernels.NutsTree(z, pe, grad, torch.ones((D,), device=gpu), step, 5, True, 7, 0)
        st = dict(da=torch.zeros((C, 5), device=gpu), wf=torch.zeros((C, 2, D), device=gpu),
                  mean=torch.zeros((C,), device=gpu), counters=torch.zeros((3, C), dtype=torch.int64, device=gpu))
        tree.set_span(0, 3, flags=kernels.NutsTree.RUN_COUNT_ACCEPTS)
        tree.run_begin()
        return tree, st
    ta, sa = fresh()
    tb, sb = fresh()
    for _ in range(40):
        for tree, st, typed in ((ta, sa, False), (tb, sb, True)):
            peq, gq = 0.5 * (tree.zq * tree.zq).sum(-1), tree.zq.clone()
            if typed:
                ops.nuts_tree_run_advance(tree.z, tree.pe, tree.grad, tree.zq, tree.rq, gq, peq, tree.inv_mass, tree.step,
                                          tree.max_tree_depth, True, tree.seed, tree.chain_offset, tree.ctl, st["da"], 0.8,
                                          st["wf"], st["mean"], st["counters"], tree.tc, tree.n_done, None, None, None,
                                          tree.accept_prob, tree.ints, tree.ws)
            else:
                tree.run_advance(peq, gq, st["da"], 0.8, st["wf"], st["mean"], st["counters"])
    for name in ("z", "pe", "grad", "zq", "rq", "accept_prob", "ints", "tc"):
        assert torch.equal(getattr(ta, name), getattr(tb, name)), name
    assert torch.equal(sa["counters"], sb["counters"]) and int(sa["counters"][0].sum()) > 0


def test_torch_compile_over_the_typed_ops(gpu):
    """torch.compile(fullgraph=True) of a function made of typed ops: they are graph nodes with shape functions
    (no graph break, no Python call-back), and the compiled function returns the eager numbers."""
    ops = _ops()
    g = torch.Generator(device=gpu).manual_seed(5)
    u = torch.randn((3, 8), device=gpu, generator=g)
    a = torch.randn((6, 1, 5), device=gpu, generator=g)
    b = torch.randn((1, 4, 5), device=gpu, generator=g)
    un = torch.randn((7, 5, 3), device=gpu, generator=g)
    pw = torch.randn((1, 4, 3, 3), device=gpu, generator=g)

    def f(u, a, b, un, pw):
        value, ld = ops.exp_site(u, 0.0)
        rs, tot = ops.dist_log_prob_sum(4, value, torch.ones((1,), device=u.device), None, None, 1.0)
        lse = ops.logsumexp_terms([a, b], [6, 4, 5], 2)
        log_z, _, _ = ops.logchain(un, pw)
        return tot + ld.sum() + lse.sum() + log_z.sum()

    want = f(u, a, b, un, pw)
    got = torch.compile(f, fullgraph=True, backend="aot_eager")(u, a, b, un, pw)
    torch.testing.assert_close(got, want, rtol=0, atol=0)


def test_typed_mixture_fwd_bwd(gpu):
    """pyro_amd::mixture_fwd_bwd(int dist, Tensor x, Tensor a, Tensor p0, Tensor? p1) -> Tensor: the leaf of a plated
    mixture for B parameter sets, against its float64 torch statement (the [K, N] log-prob frame, logsumexp, plate
    sum, autograd) -- the operators of the reference's route."""
    ops = _ops()
    g = torch.Generator(device=gpu).manual_seed(4)
    B, K, N = 3, 5, 4001
    x = torch.randn((N,), device=gpu, generator=g) * 2
    a = torch.log_softmax(torch.randn((B, K), device=gpu, generator=g), -1)
    p0 = torch.randn((B, K), device=gpu, generator=g)
    p1 = torch.rand((1, K), device=gpu, generator=g) + 0.5
    out = ops.mixture_fwd_bwd(0, x, a, p0, p1)
    assert out.shape == (B, 1 + 3 * K) and out.dtype == torch.float64
    ad, p0d = a.double().requires_grad_(True), p0.double().requires_grad_(True)
    p1d = p1.double().expand(B, K).clone().requires_grad_(True)
    lp = torch.distributions.Normal(p0d[:, :, None], p1d[:, :, None]).log_prob(x.double())
    S = torch.logsumexp(ad[:, :, None] + lp, 1).sum(-1)
    torch.testing.assert_close(out[:, 0], S.detach(), rtol=2e-5, atol=1e-3)
    S.sum().backward()
    for got, want in ((out[:, 1:1 + K], ad.grad), (out[:, 1 + K:1 + 2 * K], p0d.grad), (out[:, 1 + 2 * K:], p1d.grad)):
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4 * float(want.abs().max()))
    # ... and it composes under torch.compile(fullgraph=True)
    f = torch.compile(lambda x, a, p0, p1: ops.mixture_fwd_bwd(0, x, a, p0, p1)[:, 0].sum(), fullgraph=True,
                      backend="aot_eager")
    torch.testing.assert_close(f(x, a, p0, p1), out[:, 0].sum())
