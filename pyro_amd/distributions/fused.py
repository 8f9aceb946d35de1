"""Autograd-aware launchers of the fused site kernels.

``log_prob`` / ``log_prob_sum`` of an element-wise family are torch.autograd.Functions whose
forward and backward are single HIP kernel launches on a collapsed 2-D broadcast frame
(pa_dist_log_prob, pa_dist_log_prob_sum, pa_dist_log_prob_grad).  Broadcast operands are
passed as stride-0 views and never materialised.
"""
import torch

from .. import kernels
from ..ops import fuser as _fuser
from ..ops.torch_library import dispatcher_op as _dispatcher_op


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


_ND_MIN_ELEMS = 65536      # below this the 2-D / torch paths are launch-bound anyway


def _frame_lazy(operands, shape):
    """frame(), except that a site the N-D kernels should take (see _use_nd) is reported as
    (rows, cols, [None, ...]) BEFORE any broadcast is materialised."""
    shape = tuple(int(s) for s in shape)
    n = 1
    for s_ in shape:
        n *= s_
    if _ND_MIN_ELEMS <= n < 2 ** 31 and kernels.on_device(operands[0]) \
            and operands[0].dtype in (torch.float32, torch.float64) and len(shape) > 0:
        infos = [None if o is None else _collapse(o, shape) for o in operands]
        best = None
        for k in range(0, len(shape) + 1):
            if all(i is None or i[1][k] is not None for i in infos):
                best = k
                break
        cols = 1
        if best is not None:
            for s_ in shape[best:]:
                cols *= s_
        if (best is None or cols < 128) and _nd_frame(operands, shape) is not None:
            return 0, 0, [None] * len(operands)
    return frame(operands, shape)


def _sum_plan(shape, like_shape):
    """The pa_sum_to_nd passes [(A, R, B), ...] that bring a tensor of ``shape`` down to the broadcast
    operand's ``like_shape`` (one pass per run of adjacent reduced dims), or None when a pass falls
    outside the kernel's grid."""
    shape = list(shape)
    lead = len(shape) - len(like_shape)
    keep = [False] * lead + [ls == s for ls, s in zip(like_shape, shape[lead:])]   # False = reduce
    plan = []
    d = len(shape) - 1
    while d >= 0:
        if keep[d] or shape[d] == 1:
            d -= 1
            continue
        hi = d
        while d - 1 >= 0 and (not keep[d - 1] or shape[d - 1] == 1):
            d -= 1
        A = 1
        for s_ in shape[:d]:
            A *= s_
        R = 1
        for s_ in shape[d:hi + 1]:
            R *= s_
        B = 1
        for s_ in shape[hi + 1:]:
            B *= s_
        if A >= 65536:                      # outside the kernel's grid: let torch do this one
            return None
        plan.append((A, R, B))
        for j in range(d, hi + 1):
            shape[j] = 1
        d -= 1
    return plan


def _sum_to_eligible(g):
    return g.numel() >= _ND_MIN_ELEMS and kernels.on_device(g) and g.dtype in (torch.float32, torch.float64)


def _sum_to(g, like):
    """``g`` (the frame's shape) summed down to the shape of the broadcast operand ``like``: large
    device tensors go through pa_sum_to_nd, one pass per run of adjacent broadcast dims."""
    if g is None:
        return None
    if g.shape == like.shape:
        return g
    if not _sum_to_eligible(g):
        return g.sum_to_size(like.shape) if like.dim() > 0 or g.dim() > 0 else g
    plan = _sum_plan(g.shape, like.shape)
    if plan is None:
        return g.sum_to_size(like.shape)
    if _fuser.active() is not None and len(plan) == 1 and \
            (plan[0][1] <= _fuser.MAX_REDUCE or (g.is_contiguous() and plan[0][1] <= (1 << 26))):
        return g.sum_to_size(like.shape)        # (a recorded reduction: no launch of its own, see ops/fuser.py)
    x = g.contiguous()
    for A, R, B in plan:
        x = kernels.sum_to_nd(x, A, R, B)
    return x.reshape(like.shape)


def _sum_to_pair(g0, like0, g1, like1):
    """(_sum_to(g0, like0), _sum_to(g1, like1)); two gradients of one shape going down to one shape (a
    site's two parameters broadcast alike) share their launches (pa_sum_to_nd_pair)."""
    if (g0 is None or g1 is None or g0.shape != g1.shape or like0.shape != like1.shape
            or g0.shape == like0.shape or g0.dtype != g1.dtype or not _sum_to_eligible(g0)
            or not kernels.on_device(g1)):
        return _sum_to(g0, like0), _sum_to(g1, like1)
    plan = _sum_plan(g0.shape, like0.shape)
    if plan is None:
        return _sum_to(g0, like0), _sum_to(g1, like1)
    x0, x1 = g0.contiguous(), g1.contiguous()
    for A, R, B in plan:
        x0, x1 = kernels.sum_to_nd_pair(x0, x1, A, R, B)
    return x0.reshape(like0.shape), x1.reshape(like1.shape)


def _nd_frame(operands, shape):
    """(merged shape, operands viewed on it) with at most 4 dims for the N-D site kernels, or None:
    size-1 dims are dropped and adjacent dims merged wherever every operand is laid out as one."""
    shape = tuple(int(s) for s in shape)
    exp = [None if o is None else o.expand(shape) for o in operands]
    dims = [i for i, n in enumerate(shape) if n != 1]
    if not dims:
        return None
    groups = [[dims[0]]]
    for i in dims[1:]:
        j = groups[-1][-1]
        if all(e is None or e.stride(j) == e.stride(i) * shape[i] for e in exp):
            groups[-1].append(i)
        else:
            groups.append([i])
    if len(groups) > 4:
        return None
    mshape = []
    for grp in groups:
        n = 1
        for i in grp:
            n *= shape[i]
        mshape.append(n)
    views = []
    for e in exp:
        if e is None:
            views.append(None)
            continue
        views.append(torch.as_strided(e, tuple(mshape), tuple(e.stride(grp[-1]) for grp in groups),
                                      e.storage_offset()))
    return tuple(mshape), views


def _differentiable_log_prob(dist_id, value, p0, p1):
    """The same densities written with torch operators.  Used ONLY when a gradient of a gradient is
    asked for (``torch.autograd.grad(..., create_graph=True)``: Newton steps inside a guide, Laplace
    approximations): the HIP backward kernels return plain tensors, so the first-order gradient is then
    re-derived by autograd from this expression, on the same device, and stays differentiable."""
    from .. import _lib as L
    td = torch.distributions
    if dist_id == L.DIST_NORMAL:
        return td.Normal(p0, p1, validate_args=False).log_prob(value)
    if dist_id == L.DIST_BERNOULLI_LOGITS:
        return -torch.nn.functional.binary_cross_entropy_with_logits(
            p0.expand(torch.broadcast_shapes(p0.shape, value.shape)),
            value.expand(torch.broadcast_shapes(p0.shape, value.shape)), reduction="none")
    if dist_id == L.DIST_HALF_CAUCHY:
        return td.HalfCauchy(p0, validate_args=False).log_prob(value)
    if dist_id == L.DIST_LOG_NORMAL:
        return td.LogNormal(p0, p1, validate_args=False).log_prob(value)
    if dist_id == L.DIST_EXPONENTIAL:
        return td.Exponential(p0, validate_args=False).log_prob(value)
    if dist_id == L.DIST_HALF_NORMAL:
        return td.HalfNormal(p0, validate_args=False).log_prob(value)
    if dist_id == L.DIST_GAMMA:
        return td.Gamma(p0, p1, validate_args=False).log_prob(value)
    if dist_id == L.DIST_BETA:
        return td.Beta(p0, p1, validate_args=False).log_prob(value)
    if dist_id == L.DIST_POISSON:
        return td.Poisson(p0, validate_args=False).log_prob(value)
    if dist_id == L.DIST_BINOMIAL_LOGITS:
        n, k = p1, value
        log_comb = torch.lgamma(n + 1) - torch.lgamma(k + 1) - torch.lgamma(n - k + 1)
        return k * p0 - n * torch.nn.functional.softplus(p0) + log_comb
    if dist_id == L.DIST_KL_NORMAL_LOC:
        return -((value - p0) ** 2) / (2 * p1 ** 2) - torch.log(p1)
    if dist_id == L.DIST_KL_NORMAL_SCALE:
        return torch.log(value) + 0.5 - value ** 2 / (2 * p0 ** 2)
    raise NotImplementedError("second-order gradients of distribution id {}".format(dist_id))


def _differentiable_grads(dist_id, g, value, p0, p1, mask, scale, needs):
    """(dv, da, db) as differentiable functions of the inputs and of ``g`` (see above)."""
    from .util import scale_and_mask
    with torch.enable_grad():
        lp = scale_and_mask(_differentiable_log_prob(dist_id, value, p0, p1), scale, mask)
        inputs = [t for t, need in zip((value, p0, p1), needs) if need]
        got = iter(torch.autograd.grad(lp, inputs, g.expand_as(lp) if g.dim() else g * torch.ones_like(lp),
                                       create_graph=True, allow_unused=True)) if inputs else iter(())
    return tuple(next(got) if need else None for need in needs)


@_dispatcher_op("dist_log_prob")
class _LogProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dist_id, value, p0, p1):
        shape = torch.broadcast_shapes(value.shape, p0.shape, p1.shape if p1 is not None else ())
        f = _fuser.active()
        out = f.family_log_prob(dist_id, value, p0, p1, shape) if f is not None else None
        if out is None:
            rows, cols, (v2, a2, b2) = frame([value, p0, p1], shape)
            out = kernels.dist_log_prob(dist_id, v2, a2, b2, rows, cols).reshape(shape)
        ctx.dist_id, ctx.shape = dist_id, shape
        ctx.save_for_backward(value, p0, p1)
        return out

    @staticmethod
    def backward(ctx, g):
        value, p0, p1 = ctx.saved_tensors
        shape = ctx.shape
        need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2],
                p1 is not None and ctx.needs_input_grad[3])
        if torch.is_grad_enabled():                 # create_graph=True
            return (None,) + _differentiable_grads(ctx.dist_id, g, value, p0, p1, None, 1.0, need)
        f = _fuser.active()
        got = f.family_grads(ctx.dist_id, g, value, p0, p1, shape, need) if f is not None else None
        if got is None:
            rows, cols, (g2, v2, a2, b2) = frame([g, value, p0, p1], shape)
            got = kernels.dist_log_prob_grad(ctx.dist_id, g2, v2, a2, b2, None, 1.0, rows, cols, need)
        dv, da, db = got
        outs = [None if d is None else _sum_to(d.reshape(shape), like)
                for d, like in ((dv, value), (da, p0), (db, p1))]
        return (None,) + tuple(outs)


@_dispatcher_op("dist_log_prob_sum")
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
        ctx.dist_id, ctx.shape, ctx.scale = dist_id, shape, scale
        ctx.save_for_backward(value, p0, p1, mask)
        ctx.nd = None
        rows, cols, (v2, a2, b2, m2) = _frame_lazy([value, p0, p1, mask], shape)
        if v2 is None:                       # N-D route chosen before anything was materialised
            ctx.nd = True
            mshape, (vn, an, bn, mn) = _nd_frame([value, p0, p1, mask], shape)
            return kernels.dist_log_prob_sum_nd(dist_id, mshape, vn, an, bn, mn, scale)
        _, total = kernels.dist_log_prob_sum(dist_id, v2, a2, b2, m2, scale, rows, cols,
                                             want_total=True)
        return total

    @staticmethod
    def backward(ctx, g):
        value, p0, p1, mask = ctx.saved_tensors
        shape = ctx.shape
        need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2],
                p1 is not None and ctx.needs_input_grad[3])
        if torch.is_grad_enabled():                 # create_graph=True
            return (None,) + _differentiable_grads(ctx.dist_id, g, value, p0, p1, mask, ctx.scale,
                                                   need) + (None, None)
        if ctx.nd:
            mshape, (vn, an, bn, mn) = _nd_frame([value, p0, p1, mask], shape)
            dv, da, db = kernels.dist_log_prob_grad_nd(ctx.dist_id, mshape, g, vn, an, bn, mn,
                                                       ctx.scale, need)
        else:
            rows, cols, (v2, a2, b2, m2) = frame([value, p0, p1, mask], shape)
            dv, da, db = kernels.dist_log_prob_grad(ctx.dist_id, g.reshape(1, 1), v2, a2, b2, m2,
                                                    ctx.scale, rows, cols, need)
        dv = None if dv is None else _sum_to(dv.reshape(shape), value)
        if da is not None and db is not None:
            da, db = _sum_to_pair(da.reshape(shape), p0, db.reshape(shape), p1)
        else:
            da = None if da is None else _sum_to(da.reshape(shape), p0)
            db = None if db is None else _sum_to(db.reshape(shape), p1)
        return (None, dv, da, db, None, None)


@_dispatcher_op("dirichlet_log_prob")
class _DirichletLogProb(torch.autograd.Function):
    """Dirichlet.log_prob in one launch each way (pa_dirichlet_log_prob / _grad)."""

    @staticmethod
    def forward(ctx, value, concentration):
        ctx.save_for_backward(value, concentration)
        return kernels.dirichlet_log_prob(value, concentration)

    @staticmethod
    def backward(ctx, g):
        value, conc = ctx.saved_tensors
        if torch.is_grad_enabled():                 # create_graph=True: see _differentiable_grads
            with torch.enable_grad():
                lp = torch.distributions.Dirichlet(conc, validate_args=False).log_prob(value)
                inputs = [t for t, need in zip((value, conc), ctx.needs_input_grad) if need]
                got = iter(torch.autograd.grad(lp, inputs, g, create_graph=True, allow_unused=True))
            return tuple(next(got) if need else None for need in ctx.needs_input_grad)
        dv, dc = kernels.dirichlet_log_prob_grad(g, value, conc, ctx.needs_input_grad[0],
                                                 ctx.needs_input_grad[1])
        return (None if dv is None else _sum_to(dv, value),
                None if dc is None else _sum_to(dc, conc))


def dirichlet_log_prob(value, concentration):
    return _DirichletLogProb.invoke(value, concentration)


@_dispatcher_op("normal_rsample")
class _NormalRsample(torch.autograd.Function):
    """value = loc + scale * eps with eps from the Philox stream, ONE launch (pa_normal_rsample);
    backward: d loc = g, d scale = g * eps (torch: normal.py:83-86)."""
    volatile_args = (3, 4)      # Philox seed / block offsets: new on every call (ops/torch_library.py)


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


@_dispatcher_op("standard_gamma")
class _StandardGamma(torch.autograd.Function):
    """g ~ Gamma(concentration, 1) of ``shape`` from the keyed Philox stream, ONE launch that also
    produces d g / d concentration (pa_gamma_rsample: Marsaglia-Tsang + the implicit
    reparameterisation gradient); backward = one product and the un-broadcast.  torch:
    _standard_gamma / _standard_gamma_grad behind Gamma.rsample (gamma.py:80-88)."""
    volatile_args = (2, 3)      # Philox seed / block offsets: new on every call (ops/torch_library.py)


    @staticmethod
    def forward(ctx, concentration, shape, seed, offset, offset_dev):
        rows, cols, (a2,) = frame([concentration], shape)
        need = ctx.needs_input_grad[0]
        out, dal = kernels.gamma_rsample(a2, rows, cols, seed, offset, offset_dev, want_grad=need)
        ctx.like = concentration
        if need:
            ctx.save_for_backward(dal.reshape(shape))
        return out.reshape(shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (dal,) = ctx.saved_tensors
        return _sum_to(g * dal, ctx.like), None, None, None, None


def standard_gamma(concentration, shape):
    """Reparameterised Gamma(concentration, 1) draw of ``shape`` from the process-wide Philox stream
    (one block per element is reserved, whatever the rejection loop consumes)."""
    from .. import rng
    shape = torch.Size(shape)
    seed, off, off_dev = rng.reserve_blocks(shape.numel())
    return _StandardGamma.invoke(concentration, shape, seed, off, off_dev)


@_dispatcher_op("exp_site")
class _ExpSite(torch.autograd.Function):
    """A latent with support (lower, inf) under a Normal guide: value = lower + exp(u) and the Delta
    site's log-density -sum_event u from ONE launch, their gradient from ONE (pa_exp_site_fwd / _bwd;
    reference: AutoNormal.forward, guides.py:494-519 -- eight torch operators and their duals)."""

    @staticmethod
    def forward(ctx, u, cols, lower, event_rank):
        value, ld = kernels.exp_site_fwd(u.detach().contiguous(), cols, lower)
        ctx.cols, ctx.lower = cols, lower
        ctx.set_materialize_grads(False)        # an unused output's gradient arrives as None, not zeros
        ctx.save_for_backward(value)
        # the site's event rank is the caller's knowledge: size-1 event dims cannot be told from ``cols``
        ld = ld.reshape(u.shape[:u.dim() - event_rank])
        return value, ld

    @staticmethod
    def backward(ctx, g_value, g_ld):
        (value,) = ctx.saved_tensors
        if g_value is None and g_ld is None:
            return None, None, None, None
        gv = None if g_value is None else g_value.contiguous()
        gl = None if g_ld is None else g_ld.contiguous().reshape(-1)
        return kernels.exp_site_bwd(value, gv, gl, ctx.cols, ctx.lower), None, None, None


@_dispatcher_op("exp_lower")
class _ExpLower(torch.autograd.Function):
    """lower + exp(u): a parameter read through a positive / greater-than constraint, one launch each way
    (pa_exp_site_fwd without the Jacobian term / pa_exp_site_bwd)."""

    @staticmethod
    def forward(ctx, u, lower):
        value, _ = kernels.exp_site_fwd(u.detach().contiguous(), 1, lower, want_ld=False)
        ctx.lower = lower
        ctx.save_for_backward(value)
        return value

    @staticmethod
    def backward(ctx, g):
        (value,) = ctx.saved_tensors
        if torch.is_grad_enabled():                  # create_graph=True: stay differentiable
            return g * (value - ctx.lower), None
        return kernels.exp_site_bwd(value, g.contiguous(), None, 1, ctx.lower), None


def exp_lower(u, lower=0.0):
    if _recording(u):
        return torch.exp(u) + float(lower) if lower else torch.exp(u)
    return _ExpLower.invoke(u, float(lower))


def exp_lower_bound_of(transform):
    """The lower bound L when ``transform`` is u -> L + exp(u) with a host-side scalar L (what ``biject_to`` /
    ``transform_to`` give for positive / greater_than / nonnegative supports), else None."""
    from torch.distributions import transforms as T
    while type(transform) is T.IndependentTransform:      # (.to_event(k) sites: the sum over the event
        transform = transform.base_transform              # dims is the caller's, by the site's event_dim)
    if type(transform) is T.ExpTransform:
        return 0.0
    if type(transform) is T.ComposeTransform and len(transform.parts) == 2:
        e, a = transform.parts
        if (type(e) is T.ExpTransform and type(a) is T.AffineTransform and a.event_dim == 0
                and isinstance(a.loc, (int, float)) and isinstance(a.scale, (int, float)) and a.scale == 1):
            return float(a.loc)
    return None


def exp_site(u, event_rank, lower=0.0):
    """(value = lower + exp(u), log_density = -(u summed over its ``event_rank`` rightmost dims))."""
    if _recording(u):
        value = torch.exp(u) + float(lower) if lower else torch.exp(u)
        return value, -(u.sum(tuple(range(u.dim() - event_rank, u.dim()))) if event_rank else u)
    cols = 1
    for d in u.shape[u.dim() - event_rank:] if event_rank else ():
        cols *= int(d)
    return _ExpSite.invoke(u, cols, float(lower), int(event_rank))


@_dispatcher_op("drawn_score")
class _DrawnScore(torch.autograd.Function):
    """coef * sum log Normal(z; loc, scale) at the guide's own draw z = loc + scale * eps, as a vector of
    partial sums (pa_meanfield_score); differentiable in ``scale`` only, with the TOTAL derivative
    -coef * P / scale (the paths through z and loc are accounted for: they cancel the eps terms)."""

    @staticmethod
    def forward(ctx, scale, z, loc, P, coef):
        partial, gscale = kernels.meanfield_score(z.detach(), loc.detach(), scale.detach(), P, coef)
        ctx.save_for_backward(gscale)
        return partial

    @staticmethod
    def backward(ctx, g):
        (gscale,) = ctx.saved_tensors
        # every partial sum has the same upstream coefficient (they are summed by the caller)
        return gscale * g.reshape(-1)[0], None, None, None, None


def drawn_score(z, loc, scale, P, coef):
    return _DrawnScore.invoke(scale, z, loc, P, coef)


@_dispatcher_op("meanfield_normal_sample")
class _MeanFieldSample(torch.autograd.Function):
    """All mean-field Normal sites of a guide: per site  scale = softplus(rho),
    z = loc + scale * eps  for P vectorised particles, ONE launch forward
    (pa_meanfield_normal_sample) and ONE backward (pa_meanfield_normal_sample_bwd).  Inputs are the
    unconstrained parameter leaves (loc_0, rho_0, loc_1, rho_1, ...); outputs per site
    (z [P, n], scale [n], loc_out [n])."""
    volatile_args = (1, 2)      # Philox seed / block offsets: new on every call (ops/torch_library.py)


    @staticmethod
    def forward(ctx, P, seed, offsets, offset_dev, *params):
        locs = [p.detach().reshape(-1) for p in params[0::2]]
        rhos = [p.detach().reshape(-1) for p in params[1::2]]
        zs, scales, louts, epss = kernels.meanfield_normal_sample(locs, rhos, P, seed, offsets,
                                                                  offset_dev)
        ctx.P, ctx.nsites = P, len(locs)
        ctx.shapes = [p.shape for p in params]
        ctx.params = params if GRAD_SINK["on"] else None
        ctx.save_for_backward(*rhos, *epss)
        out = []
        for z, sc, lo in zip(zs, scales, louts):
            out += [z, sc, lo]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        k = ctx.nsites
        saved = ctx.saved_tensors
        rhos, epss = saved[:k], saved[k:]
        # gradient sink: when both leaves of a site already own a dense .grad (the views into the
        # optimizer's flat gradient buffer), the kernel adds into it directly and autograd is told
        # "no gradient" -- one AccumulateGrad add launch per parameter less
        sinks = None
        if ctx.params is not None:
            sinks = []
            for i in range(k):
                pl, pr = ctx.params[2 * i], ctx.params[2 * i + 1]
                ok = all(p.is_leaf and p.grad is not None and p.grad.is_contiguous()
                         and p.grad.dtype == p.dtype and p.grad.device == p.device
                         and p.grad.grad_fn is None and not p._backward_hooks for p in (pl, pr))
                sinks.append((pl.grad.reshape(-1), pr.grad.reshape(-1)) if ok else None)
        d_locs, d_rhos = kernels.meanfield_normal_sample_bwd(
            rhos, epss, grads[0::3], grads[1::3], grads[2::3], ctx.P, sinks)
        out = []
        for i in range(k):
            out += [None if d_locs[i] is None else d_locs[i].reshape(ctx.shapes[2 * i]),
                    None if d_rhos[i] is None else d_rhos[i].reshape(ctx.shapes[2 * i + 1])]
        return (None, None, None, None) + tuple(out)


# _MeanFieldSample.backward / _MvnTrilSample.backward may add straight into existing parameter
# .grad buffers and tell autograd "no gradient" (see there).  That is only sound when the caller IS
# the accumulation into .grad: ELBO.loss_and_grads_device switches it on around its own
# guide run + backward (grad_sink() below).  Everywhere else -- differentiable_loss, ELBOModule,
# torch.autograd.grad, backward(inputs=...), create_graph -- it stays off and the Functions hand
# ordinary gradients to autograd.
GRAD_SINK = {"on": False}


class grad_sink:
    """Context manager / decorator: guide draws made inside may sink their parameter gradients
    directly into ``.grad`` when their backward runs (the draw remembers the setting)."""

    def __enter__(self):
        self._prev = GRAD_SINK["on"]
        GRAD_SINK["on"] = True
        return self

    def __exit__(self, *exc):
        GRAD_SINK["on"] = self._prev
        return False

    def __call__(self, fn):
        import functools

        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            with grad_sink():
                return fn(*args, **kwargs)
        return wrapped


def meanfield_sample(locs, rhos, P):
    """Draw every site of a mean-field Normal guide at once.  locs / rhos: the unconstrained
    parameter leaves per site (rho -> scale through softplus).  The Philox blocks are reserved
    site by site, exactly as a sequence of normal_rsample calls would, so the draws are the ones
    the unfused path produces.  Returns per site (z [P, n], scale [n], loc_out [n])."""
    from .. import rng
    seed, offsets, off_dev = None, [], None
    for loc in locs:
        seed, off, off_dev = rng.reserve(P * loc.numel(), loc.dtype)
        offsets.append(off)
    params = []
    for loc, rho in zip(locs, rhos):
        params += [loc, rho]
    out = _MeanFieldSample.invoke(P, seed, tuple(offsets), off_dev, *params)
    return [tuple(out[3 * i:3 * i + 3]) for i in range(len(locs))]


@_dispatcher_op("mvn_tril_sample")
class _MvnTrilSample(torch.autograd.Function):
    """Full-covariance Normal guide draw: z [P, n] and log q(z) [P] from the unconstrained leaves
    (loc, rho, A) in ONE launch (pa_mvn_tril_sample), their gradients in ONE (.._bwd), added
    straight into the parameters' dense .grad buffers when those exist (see _MeanFieldSample)."""
    volatile_args = (1, 2)      # Philox seed / block offsets: new on every call (ops/torch_library.py)


    @staticmethod
    def forward(ctx, P, seed, offset, offset_dev, eps, loc, rho, A):
        l, r, a = loc.detach().reshape(-1), rho.detach().reshape(-1), A.detach()
        z, logq, eps = kernels.mvn_tril_sample(l, r, a, P, seed, offset, offset_dev, eps)
        ctx.params = (loc, rho, A) if GRAD_SINK["on"] else None
        ctx.shapes = (loc.shape, rho.shape, A.shape)
        ctx.save_for_backward(l, r, eps, z)
        return z, logq

    @staticmethod
    def backward(ctx, d_z, d_logq):
        l, r, eps, z = ctx.saved_tensors
        sinks = None
        if ctx.params is not None and all(
                p.is_leaf and p.grad is not None and p.grad.is_contiguous()
                and p.grad.dtype == p.dtype and p.grad.device == p.device
                and p.grad.grad_fn is None and not p._backward_hooks for p in ctx.params):
            sinks = tuple(p.grad.reshape(-1) for p in ctx.params[:2]) + (ctx.params[2].grad,)
        d_loc, d_rho, d_A = kernels.mvn_tril_sample_bwd(l, r, eps, z, d_z, d_logq, sinks)
        if sinks is not None:
            return (None,) * 8
        return (None, None, None, None, None, d_loc.reshape(ctx.shapes[0]),
                d_rho.reshape(ctx.shapes[1]), d_A.reshape(ctx.shapes[2]))


def mvn_tril_sample(loc, rho, A, shape):
    """(z [P, n], log q(z) [P]) of Normal(loc, scale_tril = softplus(rho)[:, None] * (tril(A, -1)
    + I)) for the P = prod(shape[:-1]) particles of a draw of ``shape``; the standard normals are
    the ones ``rng.normal(shape)`` would return at this point of the stream (or whatever a replaced
    ``rng.normal`` returns)."""
    from .. import rng
    n = loc.numel()
    P = torch.Size(shape).numel() // n
    if rng.normal is not rng._default_normal:
        eps = rng.normal(tuple(shape), loc.dtype, loc.device).reshape(P, n).contiguous()
        return _MvnTrilSample.invoke(P, 0, 0, None, eps, loc, rho, A)
    seed, off, off_dev = rng.reserve(P * n, loc.dtype)
    return _MvnTrilSample.invoke(P, seed, off, off_dev, None, loc, rho, A)


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
    return _NormalRsample.invoke(loc, scale, shape, seed, off, off_dev)


# ---- many small sites in one launch ---------------------------------------------------------------
# layout signature of an entry -> its 2-D factorisation: (rows, cols, per-operand (stride_row,
# stride_col) or None) -- or False for "not an entry".  The factorisation is a pure function of the
# operands' shapes, strides, dtypes and requires_grad flags, and a model presents the same layouts
# every step, so the stride analysis (a few dozen microseconds of Python per entry) runs once.
_FRAME_CACHE = {}


def _entry_frame(dist_id, value, p0, p1, mask):
    """2-D strided views of one entry's operands over their broadcast frame, or None when the entry
    cannot go through the multi-site kernel (too large, operands not expressible as strided views,
    mixed dtypes)."""
    key = tuple(None if t is None else (t.shape, t.stride(), t.dtype, t.requires_grad)
                for t in (value, p0, p1, mask))
    hit = _FRAME_CACHE.get(key)
    if hit is not None:
        if hit is False:
            return None
        rows, cols, strides = hit
        views = [None if t is None else torch.as_strided(t, (rows, cols), st, t.storage_offset())
                 for t, st in zip((value, p0, p1, mask), strides)]
        return (rows, cols) + tuple(views)
    out = _entry_frame_uncached(dist_id, value, p0, p1, mask)
    if len(_FRAME_CACHE) < 4096:
        if out is None:
            _FRAME_CACHE[key] = False
        else:
            rows, cols = out[0], out[1]
            ok = all(v is None or v.untyped_storage().data_ptr() == t.untyped_storage().data_ptr()
                     for t, v in zip((value, p0, p1, mask), out[2:]))
            if ok:       # (a materialised broadcast is a fresh tensor: not a layout property)
                _FRAME_CACHE[key] = (rows, cols, [None if v is None else v.stride() for v in out[2:]])
    return out


def _entry_frame_uncached(dist_id, value, p0, p1, mask):
    from .. import _lib
    ops = [t for t in (value, p0, p1) if t is not None]
    if any(not t.is_floating_point() or t.dtype != value.dtype for t in ops):
        return None
    shapes = [t.shape for t in ops]
    if mask is not None:
        shapes.append(mask.shape)
    shape = torch.broadcast_shapes(*shapes)
    n = 1
    for d in shape:
        n *= int(d)
    if n > _lib.MULTI_MAX_ELEMS:
        return None
    rows, cols, (v2, a2, b2, m2) = frame([value, p0, p1, mask], shape)
    for t, t2 in ((value, v2), (p0, a2), (p1, b2)):
        if t is None or not t.requires_grad:
            continue
        # the reduced gradient [rows or 1, cols or 1] must be reshapeable to the operand itself
        rr = 1 if (t2.shape[0] == 1 or t2.stride(0) == 0) and rows > 1 else rows
        cc = 1 if (t2.shape[1] == 1 or t2.stride(1) == 0) and cols > 1 else cols
        if rr * cc != t.numel() or t2.untyped_storage().data_ptr() != t.untyped_storage().data_ptr():
            return None
    return rows, cols, v2, a2, b2, m2


@_dispatcher_op("multi_log_prob_sum")
class _MultiLogProbSum(torch.autograd.Function):
    """total = coef_all * sum_e coef_e * sum(mask_e ? log_prob_e(value_e; p0_e, p1_e) : 0) over a
    table of small entries: ONE launch forward (pa_multi_log_prob_sum), ONE launch backward that
    writes every operand gradient already reduced to the operand's shape
    (pa_multi_log_prob_grad).  Entries that score the same value tensor (a latent's prior and its
    guide density) are chained: their value gradients arrive summed, in one buffer.  ``extras`` are
    terms whose gradient w.r.t. one of the value tensors is already known (the fused GLM site)."""

    @staticmethod
    def forward(ctx, meta, extras, coef_all, *tensors):
        proto = tensors[0]
        ctx.meta, ctx.extras, ctx.coef_all = meta, extras, coef_all
        ctx.save_for_backward(*tensors)
        ctx.eager = None
        needs = ctx.needs_input_grad[3:]
        if EAGER_GRAD["on"] and any(needs) and len(meta) <= _lib_const("MULTI_MAX_ENTRIES"):
            # the caller differentiates the total right away with a unit upstream gradient
            # (Trace_ELBO.loss_and_grads): gradients come out of the forward launch
            entries = _MultiLogProbSum._grad_entries(meta, extras, tensors, needs)
            total, grads = kernels.multi_log_prob_sum_grad(entries, coef_all, proto.dtype,
                                                           proto.device)
            ctx.eager = grads
            return total
        entries = _MultiLogProbSum._entries(meta, tensors, None)
        return kernels.multi_log_prob_sum(entries, coef_all, proto.dtype, proto.device)

    @staticmethod
    def _entries(meta, tensors, needs):
        entries, i = [], 0
        for dist_id, nops, mask, coef, fr in meta:
            ops = list(tensors[i:i + nops]) + [None] * (3 - nops)
            if fr is None:       # not framed when the entry was added (a carrier entry)
                fr = _entry_frame(dist_id, ops[0], ops[1], ops[2], mask)
            assert fr is not None
            rows, cols, v2, a2, b2, m2 = fr
            e = dict(dist=dist_id, rows=rows, cols=cols, value=v2, p0=a2, p1=b2, mask=m2, coef=coef)
            if needs is not None:
                e["need"] = tuple(list(needs[i:i + nops]) + [False] * (3 - nops))
            entries.append(e)
            i += nops
        return entries

    @staticmethod
    def _grad_entries(meta, extras, tensors, needs):
        from .. import _lib
        entries = _MultiLogProbSum._entries(meta, tensors, needs)
        # position of every entry's value among the inputs
        pos, i = [], 0
        for dist_id, nops, mask, coef, _ in meta:
            pos.append(i)
            i += nops

        def unreduced(e):
            v = e["value"]
            return not ((v.shape[0] == 1 or v.stride(0) == 0) and e["rows"] > 1) and \
                not ((v.shape[1] == 1 or v.stride(1) == 0) and e["cols"] > 1)

        # chain entries scoring the same tensor (within one launch of PA_MULTI_MAX_ENTRIES entries)
        heads = {}
        for k, e in enumerate(entries):
            if not e["need"][0] or not unreduced(e):
                continue
            key = (id(tensors[pos[k]]), k // _lib.MULTI_MAX_ENTRIES, e["rows"], e["cols"])
            if key in heads:
                tail = heads[key]
                while entries[tail].get("chain_next", -1) >= 0:
                    tail = entries[tail]["chain_next"]
                entries[tail]["chain_next"] = k
                e["by_chain"] = True
            else:
                heads[key] = k
        for target_pos, xg, xcoef in extras:
            for (tid, _, _, _), k in heads.items():
                if tid == id(tensors[target_pos]) and "extra_grad" not in entries[k]:
                    entries[k]["extra_grad"], entries[k]["extra_coef"] = xg, xcoef
                    break
            else:
                raise RuntimeError("pyro_amd: no value-gradient carrier for an extra term")
        return entries

    @staticmethod
    def backward(ctx, g):
        tensors = ctx.saved_tensors
        needs = ctx.needs_input_grad[3:]
        proto = tensors[0]
        if ctx.eager is not None:
            # the forward assumed an upstream gradient of 1: true when g IS the cached unit tensor
            grads, unit = ctx.eager, g.data_ptr() == EAGER_GRAD.get("unit_ptr")
        else:
            entries = _MultiLogProbSum._grad_entries(ctx.meta, ctx.extras, tensors, needs)
            grads, unit = kernels.multi_log_prob_grad(g, entries, ctx.coef_all, proto.dtype,
                                                      proto.device), True
        out, i = [], 0
        for (dist_id, nops, mask, coef, _), gs in zip(ctx.meta, grads):
            for j in range(nops):
                t = tensors[i + j]
                d = None if gs[j] is None else gs[j].reshape(t.shape)
                out.append(d if d is None or unit else d * g)
            i += nops
        return (None, None, None) + tuple(out)


# Set by Trace_ELBO.loss_and_grads around the ELBO assembly: the total is differentiated at once
# with a unit upstream gradient, so _MultiLogProbSum may produce the gradients in its forward launch
EAGER_GRAD = {"on": False}


def _lib_const(name):
    from .. import _lib
    return getattr(_lib, name)


class SiteBatch:
    """Collects the small element-wise sites (and already-reduced terms) of an ELBO estimate and
    evaluates their signed sum with one fused launch."""

    def __init__(self):
        self.meta, self.tensors, self.const = [], [], 0.0
        self.extras = []          # (target tensor, known gradient, coef)

    def _compatible(self, x):
        return not self.tensors or (x.dtype == self.tensors[0].dtype
                                    and x.device == self.tensors[0].device)

    def add_site(self, dist_id, value, p0, p1, mask, scale, sign):
        """True if the site was taken into the batch."""
        if isinstance(scale, torch.Tensor) or not (mask is None or isinstance(mask, torch.Tensor)):
            return False
        if mask is not None and mask.dtype != torch.bool:
            mask = mask.bool()
        if not self._compatible(value):
            return False
        with torch.no_grad():        # the 2-D views are only read for their pointers / strides
            fr = _entry_frame(dist_id, value, p0, p1, mask)
        if fr is None:
            return False
        ops = [t for t in (value, p0, p1) if t is not None]
        # the frame travels with the entry: forward and backward reuse it instead of re-deriving
        # it (three derivations per entry were a third of the eager step's host time)
        self.meta.append((dist_id, len(ops), mask, float(sign) * float(scale), fr))
        self.tensors.extend(ops)
        return True

    def add_kl_normal(self, q_loc, q_scale, p_loc, p_scale, shape, mask, scale, sign):
        """sign * scale * sum over ``shape`` of mask ? KL(Normal(q_loc, q_scale) || Normal(p_loc,
        p_scale)) : 0 as TWO entries of the launch (the KL_NORMAL_LOC / KL_NORMAL_SCALE halves,
        include/pyro_amd.h).  The operands are the distributions' un-expanded parameters: dims of
        ``shape`` that no operand (nor the mask) spans contribute a constant factor, folded into
        the entry's coefficient.  True if taken."""
        from .. import _lib
        if isinstance(scale, torch.Tensor):
            return False
        total = 1
        for d in shape:
            total *= int(d)
        n_meta, n_tensors = len(self.meta), len(self.tensors)
        for dist_id, ops in ((_lib.DIST_KL_NORMAL_LOC, (q_loc, p_loc, p_scale)),
                             (_lib.DIST_KL_NORMAL_SCALE, (q_scale, p_scale, None))):
            shapes = [t.shape for t in ops if t is not None]
            if mask is not None and isinstance(mask, torch.Tensor):
                shapes.append(mask.shape)
            try:
                spanned = torch.broadcast_shapes(*shapes)
                ok = torch.broadcast_shapes(spanned, shape) == torch.Size(shape)
            except RuntimeError:
                ok = False
            n = 1
            for d in (spanned if ok else ()):
                n *= int(d)
            ok = ok and n > 0 and self.add_site(dist_id, ops[0], ops[1], ops[2], mask,
                                                float(scale) * (total // n), -sign)
            if not ok:
                del self.meta[n_meta:], self.tensors[n_tensors:]
                return False
        return True

    def add_term(self, x, sign):
        """An already computed term (tensor of any small shape, or a Python number): sign * x.sum()."""
        from .. import _lib
        if not isinstance(x, torch.Tensor):
            self.const += float(sign) * float(x)
            return True
        if not self._compatible(x):
            return False
        if not x.is_floating_point():
            return False
        with torch.no_grad():
            fr = _entry_frame(_lib.SITE_IDENTITY, x, None, None, None)
        if fr is None:
            return False
        self.meta.append((_lib.SITE_IDENTITY, 1, None, float(sign), fr))
        self.tensors.append(x)
        return True

    def add_linear_term(self, x, targets, sign):
        """sign * x.sum() where ``x`` carries no autograd history but its gradient w.r.t. each
        tensor of ``targets`` = [(tensor, d x.sum() / d tensor), ...] is known: the gradients ride
        in the batch's backward launch (added to the value gradient of the entries that score the
        same tensor) instead of an autograd node of their own."""
        from .. import _lib
        assert not x.requires_grad
        targets = [(t, gt) for t, gt in targets if t is not None and t.requires_grad]
        for t, gt in targets:
            if not self._compatible(t) or gt.dtype != t.dtype or gt.numel() != t.numel() \
                    or t.numel() > _lib.MULTI_MAX_ELEMS:
                return False
        if not self.add_term(x, sign):
            return False
        for t, gt in targets:
            self.extras.append((t, gt.contiguous(), float(sign)))
        return True

    def total(self, coef_all=1.0):
        """coef_all * (sum of everything added); a 0-dim tensor (or a float if nothing but
        constants was added)."""
        from .. import _lib
        if not self.tensors:
            return coef_all * self.const
        meta, tensors, extras = list(self.meta), list(self.tensors), []
        for t, gt, coef in self.extras:
            # the known gradient needs an entry that produces d/dt un-reduced: one scoring t itself
            # (found by identity among the value operands) or a carrier entry added for it
            pos, i = None, 0
            for dist_id, nops, mask, c, _ in meta:
                if tensors[i] is t:
                    pos = i
                    break
                i += nops
            if pos is None:
                pos = len(tensors)
                meta.append((_lib.SITE_NONE, 1, None, 0.0, None))
                tensors.append(t)
            extras.append((pos, gt, coef))
        out = _MultiLogProbSum.invoke(tuple(meta), tuple(extras), float(coef_all), *tensors)
        if self.const != 0.0:
            out = out + coef_all * self.const
        return out


def log_prob(dist_id, value, p0, p1=None):
    return _LogProb.invoke(dist_id, value, p0, p1)


_RECORD_MAX = 65536        # up to here a site scored under a recorder scope is recorded, not launched


def _recording(*tensors):
    """Is a recorder scope (ops/fuser.py) active and are these small device tensors?  Then the site's few
    operators are written with torch operators -- the recorder turns them into nodes of its generated kernels,
    next to whatever else runs on that level -- instead of taking a launch of their own each way."""
    if _fuser.active() is None or not _fuser.ENABLED["on"]:
        return False
    return all(t is None or (isinstance(t, torch.Tensor) and _fuser._dev(t) and t.numel() <= _RECORD_MAX
                             and t.dtype in (torch.float32, torch.float64, torch.bool)) for t in tensors)


def log_prob_sum(dist_id, value, p0, p1=None, mask=None, scale=1.0):
    if mask is not None and mask.dtype != torch.bool:
        mask = mask.bool()
    if _recording(value, p0, p1, mask) and torch.is_floating_point(value):
        from .util import scale_and_mask
        return scale_and_mask(log_prob(dist_id, value, p0, p1), float(scale), mask).sum()
    return _LogProbSum.invoke(dist_id, value, p0, p1, mask, float(scale))


@_dispatcher_op("glm_bernoulli_ll")
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
        dw, db = kernels.glm_chain(g, gw, gb, ctx.needs_input_grad[2],
                                   ctx.has_b and ctx.needs_input_grad[3])
        return None, None, dw, db, None, None


def glm_bernoulli_ll(X, y, w, b=None, mask=None, scale=1.0):
    """Per-particle log-likelihood ll[P] (differentiable w.r.t. w[P,D], b[P]).  On the GPU the site
    goes through the dispatcher ops of ops/torch_library.py (pyro_amd::glm_bernoulli[_planes] and, in
    the backward, pyro_amd::glm_chain: visible to torch.jit.trace / torch.compile) once
    libpyro_amd_torch.so is present; the autograd.Function over the ctypes binding otherwise."""
    if kernels.on_device(X) and X.dtype == torch.float32:
        from ..ops import torch_library
        if torch_library.available():
            return torch_library.glm_bernoulli_ll(X, y, w, b, mask, scale)
    return _GlmBernoulliSum.invoke(X, y, w, b, mask, float(scale))


@_dispatcher_op("glm_bernoulli_grouped_ll")
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
        dw, db = kernels.glm_chain(g, gw, gb, ctx.needs_input_grad[2],
                                   ctx.has_b and ctx.needs_input_grad[3])
        return None, None, dw, db, None, None, None


def glm_bernoulli_grouped_ll(X, y, w, b, mask, scale, segs):
    return _GlmBernoulliGroupedSum.invoke(X, y, w, b, mask, float(scale), segs)
