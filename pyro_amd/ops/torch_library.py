"""The fused GLM site as torch dispatcher ops (``torch.ops.pyro_amd.*``).

``csrc/torch_ops.cpp`` registers the schemas and the GPU implementations with ``TORCH_LIBRARY`` over
the same extern-"C" launchers the ctypes binding calls; this module loads that library and adds what
is Python-side by nature: the shape functions (``register_fake``: torch.compile / meta tensors) and
the autograd formula (``register_autograd``): the backward of ``glm_bernoulli[_planes]`` is ONE more
op, ``pyro_amd::glm_chain``.  With the ops in the dispatcher ``torch.jit.trace`` of a loss function
records the site as a graph node (pyro/ops/jit.py:104-109 traces the reference's loss the same way;
its graph is made of ATen nodes) and replays it on new parameter values.

Only this site is registered: the other kernels (guide sampling, the multi-site ELBO assembly, the
optimizer) stay ctypes launches, which a tracer does not see -- a whole ``differentiable_loss`` is
therefore NOT traceable into a reusable graph yet and no JitTrace_ELBO is offered.
"""
import os

import torch

from .. import _lib, kernels

_LIB = os.path.join(os.path.dirname(_lib.__file__), "lib", "libpyro_amd_torch.so")
_state = {"loaded": None}


def available():
    """True once libpyro_amd_torch.so is loaded (built by csrc/build.py next to libpyro_amd.so)."""
    if _state["loaded"] is None:
        _state["loaded"] = False
        if os.path.exists(_LIB) and os.environ.get("PYRO_AMD_TORCH_OPS", "1") != "0":
            _lib.load()                      # libpyro_amd.so first: the shim links against it
            torch.ops.load_library(_LIB)
            _register()
            _state["loaded"] = True
    return _state["loaded"]


def _register():
    lib = torch.library

    @lib.register_fake("pyro_amd::glm_pack_planes")
    def _(X, format):
        n = _lib.load().pa_glm_planes_bytes(int(format), X.shape[0], X.shape[1])
        return X.new_empty((max(n, 16),), dtype=torch.uint8)

    def _three(w):
        P = w.shape[0]
        return w.new_empty((P,)), w.new_empty(tuple(w.shape)), w.new_empty((P,))

    @lib.register_fake("pyro_amd::glm_bernoulli_planes")
    def _(planes, y, w, b, scale, N, D, format):
        return _three(w)

    @lib.register_fake("pyro_amd::glm_bernoulli")
    def _(X, y, w, b, mask, scale):
        return _three(w)

    @lib.register_fake("pyro_amd::glm_chain")
    def _(g, gw, gb):
        return torch.empty_like(gw), torch.empty_like(gb)

    def setup(ctx, inputs, output):
        _, gw, gb = output
        ctx.save_for_backward(gw, gb)
        ctx.has_b = inputs[3] is not None            # (.., .., w, b, ...) in both schemas

    def _grads(ctx, g_ll):
        gw, gb = ctx.saved_tensors
        dw, db = torch.ops.pyro_amd.glm_chain(g_ll, gw, gb)
        return (dw if ctx.needs_input_grad[2] else None,
                db if (ctx.has_b and ctx.needs_input_grad[3]) else None)

    def backward_planes(ctx, g_ll, g_gw, g_gb):
        dw, db = _grads(ctx, g_ll)
        # (planes, y, w, b, scale, N, D, format)
        return None, None, dw, db, None, None, None, None

    def backward_plain(ctx, g_ll, g_gw, g_gb):
        dw, db = _grads(ctx, g_ll)
        # (X, y, w, b, mask, scale)
        return None, None, dw, db, None, None

    lib.register_autograd("pyro_amd::glm_bernoulli_planes", backward_planes, setup_context=setup)
    lib.register_autograd("pyro_amd::glm_bernoulli", backward_plain, setup_context=setup)


def glm_bernoulli_ll(X, y, w, b=None, mask=None, scale=1.0):
    """Per-particle log-likelihood ll[P] of the Bernoulli-logits GLM site through the dispatcher ops
    (same kernel choice as kernels.glm_bernoulli_fwd_bwd: the cached plane image of X once it
    exists, the on-the-fly kernels otherwise); differentiable w.r.t. w[P,D] and b[P]."""
    # sizes and the image lookup are host-side facts about the DATA: taken with the tracer switched
    # off (under torch.jit.trace sizes are traced values, which would also miss the image cache)
    state = torch._C._get_tracing_state()
    torch._C._set_tracing_state(None)
    try:
        N, D = int(X.shape[0]), int(X.shape[1])
        P = int(w.shape[0])
        planes = None
        if (mask is None and D <= kernels._PLANES_MAX_D and P >= kernels._PLANES_MIN_P and N > 0
                and kernels._glm_variant == kernels.GLM_AUTO):
            planes = kernels.glm_planes_of(X)
    finally:
        torch._C._set_tracing_state(state)
    y = y.contiguous()
    w = w.contiguous()
    b = b.contiguous() if b is not None else None
    if planes is not None:
        return torch.ops.pyro_amd.glm_bernoulli_planes(planes, y, w, b, float(scale), N, D,
                                                       kernels._format_of(planes))[0]
    if mask is not None:
        mask = mask.contiguous()
    return torch.ops.pyro_amd.glm_bernoulli(X.contiguous(), y, w, b, mask, float(scale))[0]
