"""The potential of a FLAT model evaluated without the handlers and without autograd.

The reference evaluates the potential of every leapfrog step by running the conditioned model under its effect
handlers, summing the sites' log-densities and differentiating the sum (pyro/infer/mcmc/util.py:264-286,
pyro/ops/integrator.py:68-94).  For a model whose latent sites are scored by fused families at parameters that
do not depend on other latents -- or ARE the value of another latent site (a hierarchical prior: w ~
Normal(mu, tau) with mu / tau latent; round 6) -- and whose observed site is the Bernoulli-logits GLM over latent
weights / bias -- Bayesian logistic regression, BASELINE configs[1]'s model under NUTS -- the whole evaluation is

    GLM kernel (log-likelihood per chain + its gradient w.r.t. weights and bias, one pass over X)
    -> its finalize
    -> the tree kernel, which adds the latent sites' own log-densities / gradients itself
       (pa_nuts_tree_run_advance_direct)

three launches per round instead of ~25 operators.  ``recognise`` inspects ONE execution of the model (through
the handlers, as the generic path runs it) and returns a ``DirectProgram`` or None; nothing is assumed that the
inspection does not establish, and a model it does not cover keeps the generic potential.
"""
import ctypes

import torch
from torch.distributions import transforms as T

from ... import _lib, kernels, poutine
from ...primitives import plate

_FAMILIES = (_lib.DIST_NORMAL, _lib.DIST_HALF_CAUCHY, _lib.DIST_LOG_NORMAL, _lib.DIST_EXPONENTIAL,
             _lib.DIST_HALF_NORMAL, _lib.DIST_GAMMA)
ENABLED = {"on": True}


def _transform_kind(t):
    """(kind, lower) of ``t`` = biject_to(support).inv: 0 identity, 1 v = lower + exp(u); None otherwise."""
    from ...distributions import fused
    if t is T.identity_transform or (type(t) is T.ComposeTransform and not t.parts):
        return 0, 0.0
    inv = getattr(t, "inv", None)
    while type(inv) is T.IndependentTransform:
        inv = inv.base_transform
    if inv is T.identity_transform or (type(inv) is T.ComposeTransform and not inv.parts):
        return 0, 0.0
    lower = fused.exp_lower_bound_of(t.inv) if inv is not None else None
    if lower is not None:
        return 1, float(lower)
    return None


def _param_view(p, n, shape=None):
    """(tensor, stride) reading element j of a site of n elements at p[j * stride]; (None, 0) for an absent
    parameter; False for one this form cannot hold.  ``shape``: the site's chain-batched value shape."""
    if p is None:
        return None, 0
    if not isinstance(p, torch.Tensor) or p.requires_grad or p.dtype != torch.float32 or not p.is_cuda:
        return False
    if p.numel() == 1:
        return p.reshape(1).contiguous(), 0
    if p.numel() == n:
        return p.reshape(n).contiguous(), 1
    if shape is not None and p.dim() == len(shape) and tuple(p.shape) == tuple(shape) and p.stride(0) == 0:
        # broadcast against a chain-valued sibling parameter (Normal(zeros(D), tau)): the same n values for every chain
        return p[0].reshape(n).contiguous(), 1
    return False


def _parent_of(p, value, cond, C, n):
    """The latent site whose chain value the parameter tensor ``p`` IS: a pure view of it (same storage, no
    arithmetic in between) that, broadcast against the site's value [C, ..., event], reads element j % len_q of
    chain c's parent for flat element j of chain c.  -> name or None."""
    if not isinstance(p, torch.Tensor) or not p.is_cuda or p.dtype != torch.float32:
        return None
    try:
        pe = p.expand(value.shape)
    except RuntimeError:
        return None
    for qname, qv in cond.items():
        if qv is value or p.untyped_storage().data_ptr() != qv.untyped_storage().data_ptr():
            continue
        numel = qv.untyped_storage().nbytes() // 4
        if numel > (1 << 22) or qv.numel() % C != 0:
            return None
        lq = qv.numel() // C
        if n % lq != 0:
            return None
        base = torch.arange(numel)
        got = base.as_strided(tuple(pe.shape), tuple(pe.stride()), pe.storage_offset()).reshape(C, n)
        par = base.as_strided(tuple(qv.shape), tuple(qv.stride()), qv.storage_offset()).reshape(C, lq)
        want = par[:, torch.arange(n) % lq]
        return qname if torch.equal(got, want) else None
    return None


class DirectProgram:
    def __init__(self, layout, sites, glm):
        self.layout, self.sites, self.glm = layout, sites, glm
        n = len(sites)
        self.n = n
        self.off = (ctypes.c_int32 * n)(*[s["off"] for s in sites])
        self.len = (ctypes.c_int32 * n)(*[s["len"] for s in sites])
        self.dist = (ctypes.c_int32 * n)(*[s["dist"] for s in sites])
        self.transform = (ctypes.c_int32 * n)(*[s["transform"] for s in sites])
        self.lower = (ctypes.c_double * n)(*[s["lower"] for s in sites])
        # (a parent-valued parameter: no tensor, stride -(parent site + 1): include/pyro_amd.h)
        self.s0 = (ctypes.c_int64 * n)(*[s["s0"] if s.get("par0") is None else -(s["par0"] + 1) for s in sites])
        self.s1 = (ctypes.c_int64 * n)(*[s["s1"] if s.get("par1") is None else -(s["par1"] + 1) for s in sites])
        self.keep = [s["p0"] for s in sites] + [s["p1"] for s in sites]

    def pointers(self, g_ext):
        """(p0, p1, g_ext) pointer arrays of one launch; g_ext: site index -> tensor."""
        n = self.n
        p0 = (ctypes.c_void_p * n)(*[None if s["p0"] is None else kernels._ptr(s["p0"]).value for s in self.sites])
        p1 = (ctypes.c_void_p * n)(*[None if s["p1"] is None else kernels._ptr(s["p1"]).value for s in self.sites])
        ge = (ctypes.c_void_p * n)(*[None if g_ext.get(k) is None else kernels._ptr(g_ext[k]).value
                                     for k in range(n)])
        return p0, p1, ge

    def glm_round(self, pack, n_slots):
        """The observed site at the cursors of ``pack`` (site-major): (ll [n_slots], {site: d ll / d value})."""
        g = self.glm
        wk, bk = g["w_site"], g["b_site"]
        sw = self.sites[wk]
        w = pack[n_slots * sw["off"]: n_slots * (sw["off"] + sw["len"])].view(n_slots, sw["len"])
        b = None
        if bk is not None:
            sb = self.sites[bk]
            b = pack[n_slots * sb["off"]: n_slots * (sb["off"] + 1)]
        ll, gw, gb = kernels.glm_bernoulli_fwd_bwd(g["X"], g["y"], w, b, None, 1.0)
        ext = {wk: gw}
        if bk is not None:
            ext[bk] = gb
        return ll, ext


    def pack(self, z):
        """Row-major unconstrained points z [n, D] -> the site-major pack the round's kernels read."""
        n = z.shape[0]
        return torch.cat([z[:, s["off"]: s["off"] + s["len"]].reshape(-1) for s in self.sites]).contiguous(), n

    def potential(self, z):
        """(U [n], dU/du [n, D]) at unconstrained points z [n, D]: the observed site's kernel, then
        pa_nuts_direct_potential -- the arithmetic the tree kernel runs in registers, written out."""
        pack, n = self.pack(z.detach().to(torch.float32))
        ll, ext = self.glm_round(pack, n)
        p0, p1, ge = self.pointers(ext)
        pe = torch.empty((n,), dtype=torch.float32, device=z.device)
        grad = torch.empty((n, self.layout.D), dtype=torch.float32, device=z.device)
        kernels.check(_lib.load().pa_nuts_direct_potential(
            kernels._ptr(pack), n, self.layout.D, self.n, self.off, self.len, self.dist, self.transform,
            self.lower, p0, self.s0, p1, self.s1, ge, kernels._ptr(ll), kernels._ptr(pe), kernels._ptr(grad),
            kernels._stream()))
        return pe, grad


def recognise(pe_maker, layout, transforms, init_params, num_chains):
    """A DirectProgram for the model behind ``pe_maker`` (infer/mcmc/util._PEMaker), or None."""
    if not ENABLED["on"] or pe_maker.enum or num_chains < 2:
        return None
    try:
        return _recognise(pe_maker, layout, transforms, init_params, num_chains)
    except Exception:      # noqa: BLE001  (anything unexpected in the inspection: the generic potential stays)
        return None


def _recognise(pe_maker, layout, transforms, init_params, C):
    first = next(iter(init_params.values()))
    if not first.is_cuda or first.dtype != torch.float32 or len(layout.names) > 8 or layout.D > 512:
        return None
    # one execution as the generic potential makes it, the unconstrained values as leaves
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in init_params.items()}
    constrained = {k: transforms[k].inv(v) for k, v in leaves.items()}
    cond = {k: pe_maker._chain_value(k, v) for k, v in constrained.items()}

    def chained(*a, **kw):
        with plate("_num_chains", C, dim=-1 - pe_maker.mpn):
            return pe_maker.model(*a, **kw)

    trace = poutine.trace(poutine.condition(chained, cond)).get_trace(*pe_maker.args, **pe_maker.kwargs)
    sites, glm, seen = [], None, set()
    by_name = {}
    for name, site in trace.nodes.items():
        if site["type"] != "sample" or type(site["fn"]).__name__ == "_Subsample":
            continue
        if name == "_num_chains":
            continue
        fn, mask, scale = site["fn"], site["mask"], site["scale"]
        if mask is not None and mask is not True:
            return None
        if isinstance(scale, torch.Tensor) or float(scale) != 1.0:
            return None
        if name in layout.slices:                                     # a latent site
            if site["is_observed"] is False and name not in cond:
                return None
            entry = getattr(fn, "fused_site_entry", None)
            entry = entry(site["value"], 1.0, None) if entry is not None else None
            if entry is None or entry[0] not in _FAMILIES:
                return None
            a, b = layout.slices[name]
            kind = _transform_kind(transforms[name])
            if kind is None:
                return None
            p0 = _param_view(entry[2], b - a, site["value"].shape)
            p1 = _param_view(entry[3], b - a, site["value"].shape)
            par = [None, None]
            if p0 is False or p1 is False:
                # a hierarchical prior: the parameter is the value of another latent site (pa_nuts_*_direct take
                # it as a parent-valued parameter: the tree kernel reads it from the chain's own cursor)
                for t, (view, raw) in enumerate(((p0, entry[2]), (p1, entry[3]))):
                    if view is False:
                        par[t] = _parent_of(raw, site["value"], cond, C, b - a)
                        if par[t] is None:
                            return None
                p0 = (None, 0) if p0 is False else p0
                p1 = (None, 0) if p1 is False else p1
            by_name[name] = dict(name=name, off=a, len=b - a, dist=int(entry[0]), transform=kind[0], lower=kind[1],
                                 p0=p0[0], s0=p0[1], p1=p1[0], s1=p1[1], par0=par[0], par1=par[1])
            seen.add(name)
        else:                                                         # an observed site: the GLM, once
            lz = getattr(fn, "lazy", None)
            if glm is not None or lz is None or type(lz).__name__ != "LinearLogits" or not site["is_observed"]:
                return None
            X, y = lz.X, site["value"]
            if not (X.is_cuda and X.dtype == torch.float32 and X.is_contiguous() and y.dim() == 1
                    and y.shape[0] == X.shape[0] and X.shape[1] <= kernels.planes_max_d() and X.shape[0] > 0):
                return None
            glm = dict(X=X, y=y.to(torch.float32).contiguous(), w=lz.w, b=lz.b)
    if glm is None or seen != set(layout.names):
        return None
    ordered = [by_name[n] for n in layout.names]                      # ascending offsets: the flat layout's order
    index_of = {s_["name"]: k for k, s_ in enumerate(ordered)}
    for s_ in ordered:                                                # parents by site index; whole copies only
        for key in ("par0", "par1"):
            if s_[key] is not None:
                q = index_of[s_[key]]
                if ordered[q]["len"] == 0 or s_["len"] % ordered[q]["len"] != 0:
                    return None
                s_[key] = q

    def site_of(t):
        """The latent site whose (identity-transformed) value ``t`` is a view of."""
        if t is None:
            return None
        for k, s in enumerate(ordered):
            v = cond[s["name"]]
            if s["transform"] == 0 and t.numel() == v.numel() and \
                    t.untyped_storage().data_ptr() == v.untyped_storage().data_ptr():
                return k
        return False

    from ...ops.lazy import _plain
    wk, bk = site_of(_plain(glm["w"])), site_of(_plain(glm["b"]) if glm["b"] is not None else None)
    if wk is False or wk is None or bk is False:
        return None
    if ordered[wk]["len"] != glm["X"].shape[1] or (bk is not None and ordered[bk]["len"] != 1):
        return None
    glm.update(w_site=wk, b_site=bk)
    return DirectProgram(layout, ordered, glm)
