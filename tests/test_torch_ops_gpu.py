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
