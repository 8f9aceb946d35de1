"""Autograd-aware launchers of the fused site kernels.

``log_prob`` / ``log_prob_sum`` of an element-wise family are torch.autograd.Functions whose
forward and backward are single HIP kernel launches on a collapsed 2-D broadcast frame
(pa_dist_log_prob, pa_dist_log_prob_sum, pa_dist_log_prob_grad).  Broadcast operands are
passed as stride-0 views and never materialised.
"""
import torch

from .. import kernels


def _collapse(t, shape):
    """Try to express ``t.expand(shape)`` as (split k -> rows/cols strides).
    Returns a list over k of (stride_row, stride_col) or None when not collapsible at k."""
    e = t.expand(shape)
    sizes, strides = list(e.shape), list(e.stride())
    n = len(sizes)
    out = []
    for k in range(n + 1):
        groups = []
        ok = True
        for lo, hi in ((0, k), (k, n)):
            dims = [(sizes[i], strides[i]) for i in range(lo, hi) if sizes[i] != 1]
            if not dims:
                groups.append(0)
                continue
            if all(st == 0 for _, st in dims):
                groups.append(0)
                continue
            good = True
            for (s0, st0), (s1, st1) in zip(dims[:-1], dims[1:]):
                if st0 != st1 * s1:
                    good = False
                    break
            if not good or dims[-1][1] == 0 and any(st != 0 for _, st in dims):
                ok = False
                break
            groups.append(dims[-1][1])
        out.append(tuple(groups) if ok else None)
    return e, out


def frame(operands, shape):
    """Find a [rows, cols] factorisation of ``shape`` under which every operand is a 2-D
    strided view.  Returns (rows, cols, [2-D as_strided tensors])."""
    shape = tuple(int(s) for s in shape)
    n = len(shape)
    if n == 0:
        return 1, 1, [None if o is None else o.reshape(1, 1) for o in operands]
    infos = [None if o is None else _collapse(o, shape) for o in operands]
    # prefer the split with the most columns (long coalesced rows) that everybody supports
    for k in range(0, n + 1):
        if all(i is None or i[1][k] is not None for i in infos):
            rows = 1
            for s in shape[:k]:
                rows *= s
            cols = 1
            for s in shape[k:]:
                cols *= s
            views = []
            for i in infos:
                if i is None:
                    views.append(None)
                    continue
                e, table = i
                sr, sc = table[k]
                views.append(torch.as_strided(e, (rows, cols), (sr, sc), e.storage_offset()))
            return rows, cols, views
    # not collapsible: materialise the broadcasts (rare; still on the GPU)
    rows = 1
    for s in shape[:-1]:
        rows *= s
    cols = shape[-1]
    return rows, cols, [None if o is None else o.expand(shape).contiguous().reshape(rows, cols)
                        for o in operands]


def _sum_to(g, like):
    if g is None:
        return None
    if g.shape == like.shape:
        return g
    return g.sum_to_size(like.shape) if like.dim() > 0 or g.dim() > 0 else g


class _LogProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dist_id, value, p0, p1):
        shape = torch.broadcast_shapes(value.shape, p0.shape, p1.shape if p1 is not None else ())
        rows, cols, (v2, a2, b2) = frame([value, p0, p1], shape)
        out = kernels.dist_log_prob(dist_id, v2, a2, b2, rows, cols).reshape(shape)
        ctx.dist_id, ctx.shape = dist_id, shape
        ctx.save_for_backward(value, p0, p1)
        return out

    @staticmethod
    def backward(ctx, g):
        value, p0, p1 = ctx.saved_tensors
        shape = ctx.shape
        rows, cols, (g2, v2, a2, b2) = frame([g, value, p0, p1], shape)
        need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2],
                p1 is not None and ctx.needs_input_grad[3])
        dv, da, db = kernels.dist_log_prob_grad(ctx.dist_id, g2, v2, a2, b2, None, 1.0, rows, cols,
                                                need)
        outs = [None if d is None else _sum_to(d.reshape(shape), like)
                for d, like in ((dv, value), (da, p0), (db, p1))]
        return (None,) + tuple(outs)


class _LogProbSum(torch.autograd.Function):
    """scalar = sum(scale_and_mask(log_prob(value), scale, mask))  -- one fused kernel."""

    @staticmethod
    def forward(ctx, dist_id, value, p0, p1, mask, scale):
        shapes = [value.shape, p0.shape]
        if p1 is not None:
            shapes.append(p1.shape)
        if mask is not None:
            shapes.append(mask.shape)
        shape = torch.broadcast_shapes(*shapes)
        rows, cols, (v2, a2, b2, m2) = frame([value, p0, p1, mask], shape)
        _, total = kernels.dist_log_prob_sum(dist_id, v2, a2, b2, m2, scale, rows, cols,
                                             want_total=True)
        ctx.dist_id, ctx.shape, ctx.scale = dist_id, shape, scale
        ctx.save_for_backward(value, p0, p1, mask)
        return total

    @staticmethod
    def backward(ctx, g):
        value, p0, p1, mask = ctx.saved_tensors
        shape = ctx.shape
        rows, cols, (v2, a2, b2, m2) = frame([value, p0, p1, mask], shape)
        g2 = g.reshape(1, 1)
        need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2],
                p1 is not None and ctx.needs_input_grad[3])
        dv, da, db = kernels.dist_log_prob_grad(ctx.dist_id, g2, v2, a2, b2, m2, ctx.scale, rows,
                                                cols, need)
        outs = [None if d is None else _sum_to(d.reshape(shape), like)
                for d, like in ((dv, value), (da, p0), (db, p1))]
        return (None,) + tuple(outs) + (None, None)


class _NormalRsample(torch.autograd.Function):
    """value = loc + scale * eps with eps from the Philox stream, ONE launch (pa_normal_rsample);
    backward: d loc = g, d scale = g * eps (torch: normal.py:83-86)."""

    @staticmethod
    def forward(ctx, loc, scale, shape, seed, offset, offset_dev):
        rows, cols, (l2, s2) = frame([loc, scale], shape)
        out, eps = kernels.normal_rsample(l2, s2, rows, cols, seed, offset, True, offset_dev)
        ctx.save_for_backward(eps.reshape(shape))
        ctx.like = (loc.shape, scale.shape)
        return out.reshape(shape)

    @staticmethod
    def backward(ctx, g):
        (eps,) = ctx.saved_tensors
        ls, ss = ctx.like
        d_loc = g.sum_to_size(ls) if ctx.needs_input_grad[0] else None
        d_scale = (g * eps).sum_to_size(ss) if ctx.needs_input_grad[1] else None
        return d_loc, d_scale, None, None, None, None


def normal_rsample(loc, scale, shape):
    """Reparameterised Normal draw of ``shape`` (>= broadcast of loc/scale shapes) from the
    process-wide Philox stream."""
    from .. import rng
    shape = torch.Size(shape)
    if rng.normal is not rng._default_normal or not loc.is_cuda:
        # someone replaced the eps source (the golden tests replay the reference's draws) or the
        # tensors are on the host (host-logic tests): keep the draw and the affine map separate
        eps = rng.normal(shape, loc.dtype, loc.device)
        return loc + eps * scale
    n = shape.numel()
    seed, off, off_dev = rng.reserve(n, loc.dtype)
    return _NormalRsample.apply(loc, scale, shape, seed, off, off_dev)


def log_prob(dist_id, value, p0, p1=None):
    return _LogProb.apply(dist_id, value, p0, p1)


def log_prob_sum(dist_id, value, p0, p1=None, mask=None, scale=1.0):
    if mask is not None and mask.dtype != torch.bool:
        mask = mask.bool()
    return _LogProbSum.apply(dist_id, value, p0, p1, mask, float(scale))


class _GlmBernoulliSum(torch.autograd.Function):
    """sum_p scale * sum_n mask_n log Bernoulli(y_n | logits = w_p.x_n + b_p): forward and
    backward of the whole observed site from ONE pass over X (pa_glm_bernoulli_fwd_bwd)."""

    @staticmethod
    def forward(ctx, X, y, w, b, mask, scale):
        ll, gw, gb = kernels.glm_bernoulli_fwd_bwd(X, y, w, b, mask, scale)
        ctx.save_for_backward(gw, gb)
        ctx.has_b = b is not None
        return ll

    @staticmethod
    def backward(ctx, g):
        gw, gb = ctx.saved_tensors
        dw = g[:, None] * gw if ctx.needs_input_grad[2] else None
        db = g * gb if (ctx.has_b and ctx.needs_input_grad[3]) else None
        return None, None, dw, db, None, None


def glm_bernoulli_ll(X, y, w, b=None, mask=None, scale=1.0):
    """Per-particle log-likelihood ll[P] (differentiable w.r.t. w[P,D], b[P])."""
    return _GlmBernoulliSum.apply(X, y, w, b, mask, float(scale))


class _GlmBernoulliGroupedSum(torch.autograd.Function):
    """Hierarchical GLM site: ll[P] and d ll / d (w[P,G,D], b[P]) from one pass over the
    group-sorted rows (pa_glm_bernoulli_grouped_fwd_bwd)."""

    @staticmethod
    def forward(ctx, X, y, w, b, mask, scale, segs):
        ll, gw, gb = kernels.glm_bernoulli_grouped_fwd_bwd(X, y, w, b, mask, scale, segs)
        ctx.save_for_backward(gw, gb)
        ctx.has_b = b is not None
        return ll

    @staticmethod
    def backward(ctx, g):
        gw, gb = ctx.saved_tensors
        dw = g[:, None, None] * gw if ctx.needs_input_grad[2] else None
        db = g * gb if (ctx.has_b and ctx.needs_input_grad[3]) else None
        return None, None, dw, db, None, None, None


def glm_bernoulli_grouped_ll(X, y, w, b, mask, scale, segs):
    return _GlmBernoulliGroupedSum.apply(X, y, w, b, mask, float(scale), segs)
