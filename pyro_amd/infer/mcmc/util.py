"""Model -> (initial params, potential function, transforms) for HMC/NUTS
(reference: pyro/infer/mcmc/util.py:264-286 _PEMaker._potential_fn, :370-482 initialize_model),
plus the flat chain-batched view of the unconstrained state the HIP kernels work on.

Chain batching.  The reference runs one Python process per chain (api.py:239-351); here all C
chains are one tensor batch.  For a model, the conditioned model is executed ONCE per leapfrog
step for all chains under an extra outermost ``plate("_num_chains", C, dim=-1-max_plate_nesting)``
(the same device the ELBO uses for vectorised particles, pyro/infer/elbo.py:186-216) and the
log-joint is reduced per chain; U(z)[c] = -log p(T^-1(z_c), data) + sum log|det J|.

Discrete latent sites are summed out of the potential exactly (reference: TraceEinsumEvaluator,
pyro/infer/mcmc/util.py:162-241): they are enumerated in parallel on the tensor dims to the left of
the plates (and of the chain dim) and the log-factors contracted by the plated sum-product of
``ops.contract`` -- per chain, because the chain plate is part of every factor's plate context.
"""
from collections import OrderedDict

import torch
from torch.distributions import biject_to

from ... import poutine
from ...distributions.util import scale_and_mask
from ...ops.integrator import potential_grad
from ...primitives import plate
from ...util import capture_scope
from ..autoguide.initialization import InitMessenger, init_to_uniform


class Layout:
    """Flat layout of the unconstrained latent sites: sorted by name (the reference sorts the
    site names of a mass-matrix block the same way, hmc.py:307-313), each site flattened."""

    def __init__(self, shapes):
        self.names = sorted(shapes)
        self.shapes = OrderedDict((n, torch.Size(shapes[n])) for n in self.names)
        self.slices = OrderedDict()
        off = 0
        for n in self.names:
            k = self.shapes[n].numel()
            self.slices[n] = (off, off + k)
            off += k
        self.D = off

    def flatten(self, z, C, batched):
        parts = []
        for n in self.names:
            v = z[n]
            parts.append(v.reshape(C, -1) if batched else v.reshape(1, -1))
        return torch.cat(parts, dim=1).contiguous() if parts else None

    def unflatten(self, z_flat, batched):
        out = {}
        C = z_flat.shape[0]
        for n in self.names:
            a, b = self.slices[n]
            v = z_flat[:, a:b]
            out[n] = v.reshape((C,) + tuple(self.shapes[n])) if batched else \
                v.reshape(self.shapes[n])
        return out


class FlatPotential:
    """(pe[C], grad[C, D]) of a dict-style potential at a flat state z[C, D]."""

    def __init__(self, potential_fn, layout, batched):
        self.fn, self.layout, self.batched = potential_fn, layout, batched
        self._direct = getattr(potential_fn, "potential_and_grad", None) \
            if len(layout.names) == 1 else None

    def __call__(self, z_flat):
        if self._direct is not None:
            return self._direct(z_flat)
        z = {k: v.detach().clone() for k, v in self.layout.unflatten(z_flat, self.batched).items()}
        grads, pe = potential_grad(self.fn, z)
        C = z_flat.shape[0]
        g = torch.cat([grads[n].reshape(C, -1) for n in self.layout.names], dim=1)
        pe = pe.reshape(-1)
        if pe.numel() != C:
            raise ValueError(
                "potential_fn must return one energy per chain (shape [{}]) when the parameters "
                "carry a leading chain dim; got shape {}".format(C, tuple(pe.shape)))
        return pe, g.contiguous()


class GraphedPotential:
    """``jit_compile=True`` of HMC / NUTS on this backend: after a few eager evaluations the whole
    potential evaluation -- the model executed for all chains under the handlers, its autograd
    backward, the mass-matrix products of a dense mass -- is captured into a hipGraph and every
    later leapfrog step replays it (the reference traces the same computation with torch.jit,
    pyro/infer/mcmc/util.py:300-338 + pyro/ops/jit.py).  One model execution is ~1 ms of Python
    and a few dozen launches; a replay is two stream operations.

    The captured kernels read ``z`` from a buffer owned by this object and write (pe, grad) into
    buffers owned by it; callers get copies.  A potential that cannot be captured (it synchronises
    with the host, allocates pinned memory, ...) falls back to eager evaluation with a warning."""

    def __init__(self, base, warmup=3):
        self.base, self.warmup = base, warmup
        self.calls, self.graph, self.failed = 0, None, False
        self.replays = 0           # graph launches so far (survives release of the graph)
        self.z_static = self.pe_static = self.grad_static = None

    def __call__(self, z_flat):
        if self.failed or not z_flat.is_cuda:
            return self.base(z_flat)
        if self.graph is None or z_flat.shape != self.z_static.shape:
            self.calls += 1
            if self.calls <= self.warmup or self.graph is not None:
                return self.base(z_flat)
            try:
                self._capture(z_flat)
            except Exception as e:  # noqa: BLE001  (anything that synchronises inside the capture)
                import os
                import warnings
                if os.environ.get("PYRO_AMD_DEBUG_GRAPH"):
                    raise
                warnings.warn("pyro_amd: hipGraph capture of the potential failed ({}: {}); "
                              "continuing with eager evaluations".format(type(e).__name__, e))
                self.failed = True
                return self.base(z_flat)
        self.z_static.copy_(z_flat)
        self.graph.replay()
        self.replays += 1
        return self.pe_static.clone(), self.grad_static.clone()

    def _capture(self, z_flat):
        self.z_static = z_flat.detach().clone()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with capture_scope(), torch.cuda.graph(graph):
            pe, grad = self.base(self.z_static)
            pe, grad = pe.detach().contiguous(), grad.detach().contiguous()
        self.graph, self.pe_static, self.grad_static = graph, pe, grad


def _guess_max_plate_nesting(model, args, kwargs):
    with poutine.block():
        trace = poutine.trace(model).get_trace(*args, **kwargs)
    dims = [f.dim for site in trace.nodes.values() if site["type"] == "sample"
            for f in site["cond_indep_stack"] if f.vectorized]
    return -min(dims) if dims else 0


def enum_log_joint(trace, nplates, C=None):
    """log of the joint density of ``trace`` with every enumerated site summed out: one value per
    chain (``C`` given; the chain plate is the outermost of the ``nplates`` plate dims) or a scalar.
    As in the reference every factor enters scaled and masked (trace_struct.py:248-288)."""
    from ...ops.contract import Term, contract_tensor_tree, pack
    from ..traceenum_elbo import _enum_log_prob, _lazy_family

    terms, enum_ids = [], set()
    for name, site in trace.nodes.items():
        if site["type"] != "sample" or type(site["fn"]).__name__ == "_Subsample":
            continue
        mask = site["mask"]
        if mask is False:
            continue
        if site["is_observed"] and not isinstance(site["scale"], torch.Tensor) \
                and float(site["scale"]) == 1.0:
            # a plated mixture's likelihood: kept lazy for the leaf kernel (csrc/mixture.hip; with vectorised
            # chains the chain plate is the kernel's batch of parameter sets over the shared data)
            lazy = _lazy_family(site, -1 - nplates)
            if lazy is not None:
                ordinal = frozenset(f for f in site["cond_indep_stack"] if f.vectorized)
                terms.append(Term(None, (lazy[0],), ordinal, lazy=lazy[1]))
                continue
        if site["infer"].get("_enumerate_dim") is not None and (mask is None or mask is True) \
                and not isinstance(site["scale"], torch.Tensor) and float(site["scale"]) == 1.0:
            # an enumerated site at its own support: the un-expanded table, constant along the data plate
            # (the plate product multiplies it by the plate's size), not a gather out of its plate-expanded copy
            lp = _enum_log_prob(site)
        else:
            lp = scale_and_mask(site["fn"].log_prob(site["value"], *site["args"], **site["kwargs"]),
                                site["scale"], None if mask is True else mask)
        ordinal = frozenset(f for f in site["cond_indep_stack"] if f.vectorized)
        terms.append(pack(lp, site["infer"].get("_dim_to_id", {}), nplates, ordinal))
        edim = site["infer"].get("_enumerate_dim")
        if edim is not None:
            enum_ids.add(site["infer"]["_dim_to_id"][edim])

    def reduce(t):
        if C is None:
            return t.sum()
        if t.dim() == nplates and t.shape[0] == C:
            return t.reshape(C, -1).sum(-1)
        return t.sum()         # the same for every chain

    total = 0.0
    factors = OrderedDict()
    for term in terms:
        if term.dims & enum_ids:
            factors.setdefault(term.ordinal, []).append(term)
        else:
            total = total + reduce(term.dense())
    if factors:
        # (every factor is summed over all its plates but the chain plate below: a leaf kernel may cover them)
        for out in contract_tensor_tree(factors, enum_ids, reduce_all=True,
                                        keep_dims=() if C is None else (-nplates,)).values():
            for term in out:
                total = total + reduce(term.tensor)
    return total


class TraceEinsumEvaluator:
    """``log_prob(model_trace)``: the log joint of a trace whose discrete latent sites were enumerated
    in parallel, with those sites summed out by plated variable elimination -- what HMC/NUTS
    differentiate for models with discrete latents (interface of pyro/infer/mcmc/util.py:161-241;
    the reference also has a TraceTreeEvaluator that walks the plate tree instead of calling einsum --
    same answers, and the same class here)."""

    def __init__(self, model_trace, has_enumerable_sites=False, max_plate_nesting=None):
        self.has_enumerable_sites = has_enumerable_sites
        self.max_plate_nesting = max_plate_nesting
        if has_enumerable_sites and max_plate_nesting is None:
            raise ValueError("Finite value required for `max_plate_nesting` when model has discrete "
                             "(enumerable) sites.")

    def log_prob(self, model_trace):
        if not self.has_enumerable_sites:
            return model_trace.log_prob_sum()
        from ...util import check_site_shape
        from ..util import is_validation_enabled
        if is_validation_enabled():
            for site in model_trace.nodes.values():
                if site["type"] == "sample" and type(site["fn"]).__name__ != "_Subsample":
                    check_site_shape(site, self.max_plate_nesting)
        return enum_log_joint(model_trace, self.max_plate_nesting)


TraceTreeEvaluator = TraceEinsumEvaluator


class _PEMaker:
    def __init__(self, model, model_args, model_kwargs, transforms, max_plate_nesting, num_chains,
                 batch_ndims):
        self.model, self.args, self.kwargs = model, model_args, model_kwargs
        self.transforms = transforms
        self.mpn = max_plate_nesting
        self.C = self._rows = num_chains
        self.batch_ndims = batch_ndims   # site -> number of batch dims of its distribution
        self.enum = False                # the model has discrete latent sites to sum out

    # torch drops the strong reference an inverse transform holds to its transform when pickled
    # (Transform.__getstate__ sets _inv = None): potential functions are torch.save'd by users
    # (the reference patches torch for this, distributions/torch_patch.py:45-53), so the
    # inverse transforms travel as their base transform
    def __getstate__(self):
        from torch.distributions.transforms import _InverseTransform
        state = self.__dict__.copy()
        state["transforms"] = {k: ("inv", t._inv) if isinstance(t, _InverseTransform) else ("fwd", t)
                               for k, t in self.transforms.items()}
        return state

    def __setstate__(self, state):
        state = dict(state)
        state["transforms"] = {k: (t.inv if kind == "inv" else t)
                               for k, (kind, t) in state["transforms"].items()}
        self.__dict__.update(state)

    def _chain_value(self, name, v):
        """[C, *site_shape] -> [C, 1 ... 1, *site_shape] so the chain dim sits at
        -1 - max_plate_nesting relative to the site's batch dims."""
        pad = self.mpn - self.batch_ndims[name]
        return v.reshape((v.shape[0],) + (1,) * pad + tuple(v.shape[1:]))

    def _chain_sum(self, site):
        fn, value, scale, mask = site["fn"], site["value"], site["scale"], site["mask"]
        C = self._rows
        if mask is False:
            return 0.0
        if mask is True:
            mask = None
        batch = getattr(fn, "fused_log_prob_batch", None)
        if batch is not None and not isinstance(scale, torch.Tensor):
            out = batch(value, scale, mask)
            if out is not None and out.numel() == C:
                return out.reshape(C)
        lp = scale_and_mask(fn.log_prob(value, *site["args"], **site["kwargs"]), scale, mask)
        if lp.dim() > self.mpn + 1:
            raise NotImplementedError("enumerated sites are not supported by the vectorised "
                                      "HMC/NUTS potential")
        if lp.dim() == self.mpn + 1 and lp.shape[0] == C:
            return lp.reshape(C, -1).sum(-1)
        return lp.sum()   # does not depend on the chain: a constant shared by all chains

    def _enum_log_joint(self, trace, nplates, C):
        return enum_log_joint(trace, nplates, C)

    def _enumerated(self, fn, nplates):
        from ..enum import config_enumerate
        return poutine.enum(config_enumerate(fn), first_available_dim=-1 - nplates)

    def potential_fn(self, params):
        if self.C == 1 and not self._batched(params):
            constrained = {k: self.transforms[k].inv(v) for k, v in params.items()}
            model = self._enumerated(self.model, self.mpn) if self.enum else self.model
            trace = poutine.trace(poutine.condition(model, constrained)).get_trace(
                *self.args, **self.kwargs)
            log_joint = self._enum_log_joint(trace, self.mpn, None) if self.enum else \
                trace.log_prob_sum()
            for name, t in self.transforms.items():
                log_joint = log_joint - torch.sum(
                    t.log_abs_det_jacobian(constrained[name], params[name]))
            return -log_joint
        # the number of chain rows of THIS evaluation: num_chains, or fewer in a compacted round of NUTS
        # (the cursors of the chains still building trees only)
        C = self._rows = int(next(iter(params.values())).shape[0]) if params else self.C
        constrained = {k: self.transforms[k].inv(v) for k, v in params.items()}
        cond = {k: self._chain_value(k, v) for k, v in constrained.items()}

        def chained(*a, **kw):
            with plate("_num_chains", C, dim=-1 - self.mpn):
                return self.model(*a, **kw)

        if self.enum:
            chained = self._enumerated(chained, self.mpn + 1)
        trace = poutine.trace(poutine.condition(chained, cond)).get_trace(*self.args,
                                                                         **self.kwargs)
        if self.enum:
            log_joint = self._enum_log_joint(trace, self.mpn + 1, C)
        else:
            log_joint = 0.0
            for name, site in trace.nodes.items():
                if site["type"] == "sample" and type(site["fn"]).__name__ != "_Subsample":
                    log_joint = log_joint + self._chain_sum(site)
        for name, t in self.transforms.items():
            ladj = t.log_abs_det_jacobian(constrained[name], params[name])
            log_joint = log_joint - ladj.reshape(C, -1).sum(-1)
        if not isinstance(log_joint, torch.Tensor) or log_joint.dim() == 0:
            log_joint = torch.as_tensor(log_joint).expand(C)
        return -log_joint

    def _batched(self, params):
        for k, v in params.items():
            return v.dim() == len(self._site_shape[k]) + 1
        return False


def initialize_model(model, model_args=(), model_kwargs=None, transforms=None,
                     max_plate_nesting=None, jit_compile=False, jit_options=None,
                     skip_jit_warnings=False, num_chains=1, init_strategy=init_to_uniform,
                     initial_params=None):
    """Returns (initial_params, potential_fn, transforms, prototype_trace).

    With ``num_chains > 1`` the initial parameters carry a leading chain dim and the potential
    maps them to one energy per chain."""
    model_kwargs = {} if model_kwargs is None else model_kwargs
    # jit_compile / jit_options / skip_jit_warnings: accepted for scripts written for the reference; there is
    # no tracing compiler here (kernels capture their rounds in HIP graphs, see HMC(jit_compile=True))
    automatic = transforms is None
    transforms = {} if transforms is None else dict(transforms)
    if max_plate_nesting is None:
        max_plate_nesting = _guess_max_plate_nesting(model, model_args, model_kwargs)

    def draw():
        return poutine.trace(InitMessenger(init_strategy)(model)).get_trace(*model_args,
                                                                            **model_kwargs)

    trace = draw()
    from ...util import check_site_shape
    from ..util import is_validation_enabled
    discrete = [name for name, node in trace.iter_stochastic_nodes()
                if getattr(node["fn"], "has_enumerate_support", False)]
    if discrete and is_validation_enabled():
        # summing out discrete sites relies on the plate structure: every batch dim must be declared
        # (the reference checks this in TraceEinsumEvaluator, mcmc/util.py:196-203)
        for node in trace.nodes.values():
            if node["type"] == "sample" and type(node["fn"]).__name__ != "_Subsample":
                check_site_shape(node, max_plate_nesting)
    # batch dims that are not declared through a plate count too: the chain plate must sit to the
    # left of every batch dim of every site
    for node in trace.nodes.values():
        if node["type"] == "sample" and type(node["fn"]).__name__ != "_Subsample":
            max_plate_nesting = max(max_plate_nesting, len(node["fn"].batch_shape))
    batch_ndims, site_shape = {}, {}
    enumerated = set()

    def collect(tr):
        out = {}
        for name, node in tr.iter_stochastic_nodes():
            fn = node["fn"]
            if type(fn).__name__ == "_Subsample":
                if fn.subsample_size is not None and fn.subsample_size < fn.size:
                    raise NotImplementedError("HMC/NUTS does not support models with subsample "
                                              "sites")
                continue
            if getattr(fn, "has_enumerate_support", False):
                enumerated.add(name)       # summed out of the potential, not a coordinate
                continue
            out[name] = node["value"].detach()
            batch_ndims[name] = len(fn.batch_shape)
            if automatic:
                transforms[name] = biject_to(fn.support).inv
        return out

    samples = collect(trace)
    pe_maker = _PEMaker(model, model_args, model_kwargs, transforms, max_plate_nesting,
                        num_chains, batch_ndims)
    draws_made = initial_params is None
    if initial_params is None:
        draws = [samples] + [collect(draw()) for _ in range(num_chains - 1)]
        initial_params = {}
        for k in samples:
            vals = [transforms[k](d[k]) for d in draws]
            initial_params[k] = torch.stack(vals) if num_chains > 1 else vals[0]
    drawn = draws_made
    for k, v in initial_params.items():
        site_shape[k] = tuple(v.shape[1:]) if num_chains > 1 else tuple(v.shape)
    pe_maker._site_shape = site_shape
    pe_maker.enum = bool(enumerated)
    if drawn:
        # A chain that starts where the potential or its gradient is not finite rejects every
        # proposal (and drives the step-size search to its floor): such starting points are drawn
        # again, as the reference does (pyro/infer/mcmc/util.py:430-470, up to 100 attempts).
        for attempt in range(100):
            bad = ~_finite_start(pe_maker.potential_fn, initial_params, num_chains)
            if not bool(bad.any()):
                break
            for c in torch.nonzero(bad).reshape(-1).tolist():
                d = collect(draw())
                for k in initial_params:
                    z = transforms[k](d[k])
                    if num_chains > 1:
                        initial_params[k][c] = z
                    else:
                        initial_params[k] = z
        else:
            raise ValueError("Model specification seems incorrect - cannot find valid initial "
                             "params (the potential energy or its gradient is not finite after "
                             "100 draws).")
    return initial_params, pe_maker.potential_fn, transforms, trace


def _finite_start(potential_fn, params, num_chains):
    """bool [num_chains]: potential and gradient finite at the given unconstrained point(s)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    with torch.enable_grad():
        pe = potential_fn(leaves)
        pe = pe if isinstance(pe, torch.Tensor) else torch.as_tensor(pe)
        grads = torch.autograd.grad(pe.sum(), list(leaves.values()), allow_unused=True) \
            if pe.requires_grad else [None] * len(leaves)
    ok = torch.isfinite(pe.detach()).reshape(-1)
    if ok.numel() != num_chains:
        ok = ok.all().expand(num_chains).clone()
    for g in grads:
        if g is not None:
            ok = ok & torch.isfinite(g.reshape(num_chains, -1)).all(1)
    return ok


def __getattr__(name):
    # pyro.infer.mcmc.util is where the reference keeps these (mcmc/util.py:620-806); they live next
    # to the MCMC driver here (api.py imports this module, hence the late lookup)
    if name in ("select_samples", "diagnostics", "print_summary"):
        from . import api
        return getattr(api, name)
    raise AttributeError("module {!r} has no attribute {!r}".format(__name__, name))
