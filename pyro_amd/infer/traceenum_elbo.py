"""TraceEnum_ELBO: ELBO with exact marginalisation of discrete model variables enumerated in
parallel (reference: pyro/infer/traceenum_elbo.py:112-214 _compute_model_factors /
_compute_dice_elbo, :334-394 _get_trace/_get_traces, :415-470 loss_and_grads).

Scope of this plugin (what BASELINE config 4, examples/lda.py, needs):
  * discrete sites enumerated in the MODEL (``infer={"enumerate": "parallel"}`` or
    ``config_enumerate``) and absent from the guide are summed out exactly by plated
    sum-product message passing (pyro_amd/ops/contract.py);
  * with a fully reparameterised guide every DiCE weight is 1 and the surrogate equals the ELBO
    estimate (pyro/infer/util.py:264-326 reduces to a plain sum);
  * guide sites enumerated in parallel (``config_enumerate(guide)``) and score-function
    (non-reparameterised) sampled guide sites enter through their DiCE factors: every cost at
    plate context o is weighted by exp(sum of the factors of the guide sites whose context is
    contained in o) -- q itself for an enumerated site, exp(log q - stop_gradient(log q)) for a
    sampled one -- and summed over plates and enumeration dims (Dice.compute_expectation,
    pyro/infer/util.py:196-326; the reference contracts the same products with an einsum and
    reads marginals back, here the broadcast product is formed directly: same value and gradient,
    exponential in the number of jointly enumerated guide variables per cost);
  * model enumeration must be no more global than guide enumeration (the reference's
    _check_model_guide_enumeration_constraint, traceenum_elbo.py:50-65).
Sequential enumeration is supported for guide sites (one trace per joint assignment).
"""
from .util import is_validation_enabled
import math
import warnings
from collections import OrderedDict

import torch

from ..distributions.fused import grad_sink as _grad_sink

from .. import kernels, poutine
from ..distributions.util import scale_and_mask
from ..ops.contract import LazyGather, Term, _eliminate, align, contract_tensor_tree, pack
from ..poutine.util import prune_subsample_sites
from ..util import check_traceenum_requirements, torch_item, warn_if_nan
from .elbo import ELBO
from .enum import check_model_guide_match, check_site_shapes, config_enumerate  # noqa: F401
from .trace_elbo import _signed_sum


def _ordinal(site):
    return frozenset(f for f in site["cond_indep_stack"] if f.vectorized)


def _packed(site, lp, first_enum_dim):
    """The site's log-probability tensor as a packed, id-named Term."""
    return pack(lp, site["infer"].get("_dim_to_id", {}), -1 - first_enum_dim, _ordinal(site))


def _check_local_sampling(model_trace, guide_trace):
    """Warnings about ``num_samples`` sites (traceenum_elbo.py:68-108): different draw counts across
    guide sites may bias the estimate; a site multiply sampled in the MODEL is summed like an
    enumerated one, which is not an unbiased gradient."""
    counts = {site["infer"]["num_samples"] for site in guide_trace.nodes.values()
              if site["type"] == "sample" and site["infer"].get("enumerate") == "parallel"
              and site["infer"].get("num_samples") is not None}
    if len(counts) > 1:
        warnings.warn("\n".join([
            "Using different numbers of Monte Carlo samples for different guide sites in "
            "TraceEnum_ELBO.", "This may be biased if the guide is not factorized"]), UserWarning)
    for name, site in model_trace.nodes.items():
        if site["type"] == "sample" and site["infer"].get("enumerate") == "parallel" \
                and site["infer"].get("num_samples") and name not in guide_trace:
            warnings.warn("\n".join([
                "Site {} is multiply sampled in model,".format(site["name"]),
                "expect incorrect gradient estimates from TraceEnum_ELBO.",
                "Consider using exact enumeration or guide sampling if possible."]), RuntimeWarning)


def _enum_log_prob(site):
    """log_prob of an enumerated site at its own enumerated support.  For a Categorical that was
    expanded over plates (``Categorical(doc_topics)`` inside the words plate of examples/lda.py) the
    answer is the log-probability table itself with the support axis moved to the enumeration dim --
    a VIEW of the un-expanded table, [T, 1.., batch] -- instead of a gather out of the expanded
    table, whose result ([T, words, docs]) and whose autograd dual (zeros of that shape, a scatter,
    a sum back over the words) are 200 MB tensors at 1e5 documents."""
    fn, value = site["fn"], site["value"]
    base = getattr(fn, "_base_logits", None)
    from ..distributions import Categorical
    if base is not None and type(fn).log_prob is Categorical.log_prob \
            and site["infer"].get("_enumerate_dim") is not None and value.dim() >= 1 \
            and site["infer"].get("num_samples") is None:
        T, n = base.shape[-1], value.dim()
        batch = base.shape[:-1]
        if value.shape == (T,) + (1,) * (n - 1) and len(batch) <= n - 1 \
                and site["infer"]["_enumerate_dim"] == -n:
            return base.movedim(-1, 0).reshape((T,) + (1,) * (n - 1 - len(batch)) + tuple(batch))
    return fn.log_prob(value)


def _lazy_gather(site, first_enum_dim):
    """Observed Categorical whose logits are [T, 1.., V] expanded over two plates and whose value
    is int64 [W, D]: keep the factor as (table, index) for the fused kernel."""
    fn, value = site["fn"], site["value"]
    if not isinstance(fn, torch.distributions.Categorical) or value.dtype != torch.int64:
        return None
    if value.dim() != 2 or site["mask"] is not None:
        return None
    logits = fn.logits
    if logits.dim() < 4 or logits.shape[-3:-1] != value.shape:
        return None
    lead = logits.shape[:-3]
    nz = [i for i, s in enumerate(lead) if s > 1]
    if len(nz) != 1 or logits.stride(-2) != 0 or logits.stride(-3) != 0:
        return None
    edim = nz[0] - len(lead) - 2          # tensor dim of the enumerated variable in the factor
    if edim > first_enum_dim or edim not in site["infer"].get("_dim_to_id", {}):
        return None
    T, V = lead[nz[0]], logits.shape[-1]
    base = getattr(fn, "_base_logits", None)
    if base is not None and base.numel() == T * V and base.shape[-1] == V:
        table = base.reshape(T, V)        # the table before Categorical.expand (no huge backward)
    else:
        idx = (0,) * nz[0] + (slice(None),) + (0,) * (len(lead) - nz[0] - 1) + (0, 0, slice(None))
        table = logits[idx].reshape(T, V)  # [T, V] view of the un-expanded log-probabilities
    return site["infer"]["_dim_to_id"][edim], LazyGather(table, value)


def _lazy_family(site, first_enum_dim):
    """An observed element-wise site under the data plate (dim -1) -- and at most one outer plate (vectorised
    chains / particles) -- whose parameters were indexed by ONE enumerated value: ``Normal(locs[z], scale)`` with
    float data [N].  Keep the [K, (B,) N] factor as (family, data, parameters per k and b) for the mixture leaf
    kernel (ops/contract.py::_try_fused_mixture).  -> (enum id, LazyFamily) or None."""
    from ..ops import contract
    fn, value = site["fn"], site["value"]
    if not contract.FUSED_MIXTURE or not site["is_observed"] or not (site["mask"] is None or site["mask"] is True):
        return None
    ev = tuple(getattr(fn, "event_shape", ()))
    if not isinstance(value, torch.Tensor) or value.dim() != 1 + len(ev) or value.requires_grad \
            or value.dtype not in (torch.float32, torch.float64) or not kernels.on_device(value):
        return None
    if len(ev) > 1 or (len(ev) == 1 and (ev[0] > kernels.MIXTURE_MAX_D or value.shape[1] != ev[0])):
        return None
    D = ev[0] if ev else None                 # a diagonal family over D features (to_event(1)) or scalar data
    plate_dims = sorted(f.dim for f in site["cond_indep_stack"] if f.vectorized)
    if not plate_dims or plate_dims[-1] != -1 or len(plate_dims) > 2:
        return None
    entry = getattr(fn, "fused_site_entry", None)
    bs = tuple(getattr(fn, "batch_shape", ()))
    if entry is None or len(bs) < 2 or bs[-1] not in (1, value.shape[0]):
        return None
    nb = len(bs)
    batch_dim = plate_dims[0] if len(plate_dims) == 2 else None
    if batch_dim is not None and -batch_dim > nb:
        batch_dim = None                      # the outer plate does not reach this site's parameters
    nz = [i for i, s_ in enumerate(bs[:-1]) if s_ > 1 and i - nb != batch_dim]
    if len(nz) != 1 or (batch_dim is not None and nz[0] - nb > batch_dim):
        return None
    edim = nz[0] - nb
    if edim > first_enum_dim or edim not in site["infer"].get("_dim_to_id", {}):
        return None
    K = bs[nz[0]]
    B = 1 if batch_dim is None else bs[batch_dim]
    if batch_dim is not None and B == 1:
        batch_dim = None
    ent = entry(value, 1.0, None)
    if ent is None or ent[0] not in kernels.MIXTURE_FAMILIES or K > kernels.MIXTURE_MAX_K:
        return None
    if D is not None and ent[0] != kernels._lib.DIST_NORMAL:
        return None                           # (the event-shaped leaf scores the diagonal Normal only)
    params = []
    for p in (ent[2], ent[3]):
        if p is None:
            params.append(None)
            continue
        if not isinstance(p, torch.Tensor) or p.dtype != value.dtype or not kernels.on_device(p):
            return None
        ne = 0 if D is None else 1            # trailing event dims of the parameter
        if p.dim() > nb + ne:
            return None
        shp = (1,) * (nb + ne - p.dim()) + tuple(p.shape)
        Dp = 1
        if ne:
            Dp, shp = shp[-1], shp[:-1]
            if Dp not in (1, D):
                return None
        for i, s_ in enumerate(shp):
            d = i - nb
            ok = s_ == 1 or (d == edim and s_ == K) or (batch_dim is not None and d == batch_dim and s_ == B)
            if not ok:
                return None                   # a parameter that varies along the data plate: the generic path
        Kp = shp[nz[0]]
        Bp = 1 if batch_dim is None else shp[batch_dim]
        # (the enumeration dim lies left of the plate dim, the event dim right of both)
        params.append(p.reshape(Kp, Bp) if D is None else p.reshape(Kp, Bp, Dp))
    if params[0] is None:
        return None

    def packed():
        lp = fn.log_prob(value, *site["args"], **site["kwargs"])
        return _packed(site, lp, first_enum_dim).tensor

    if D is not None and params[1] is None:
        return None
    return site["infer"]["_dim_to_id"][edim], contract.LazyFamily(ent[0], value, K, params[0], params[1], packed,
                                                                  batch_dim, B, D)


class TraceEnum_ELBO(ELBO):
    def _get_traces(self, model, guide, args, kwargs):
        """As ELBO._get_traces; guide sites marked for SEQUENTIAL enumeration multiply the traces:
        one (model, guide) pair per joint assignment of those sites (pyro/infer/enum.py:88-135),
        each weighted through the sites' DiCE factors, the estimate is their sum."""
        self._seq_queue, self._seq_assignment = [], None
        for pair in super()._get_traces(model, guide, args, kwargs):
            yield pair
            queue = self._seq_queue
            while queue:
                assignment = queue.pop(0)
                self._seq_assignment = assignment
                try:
                    if self.vectorize_particles:
                        yield self._get_vectorized_trace(model, guide, args, kwargs)
                    else:
                        yield self._get_trace(model, guide, args, kwargs)
                finally:
                    self._seq_assignment = None

    def _get_trace(self, model, guide, args, kwargs):
        if self.max_plate_nesting == float("inf"):
            self._guess_max_plate_nesting(model, guide, args, kwargs)
        first_enum_dim = -1 - self.max_plate_nesting
        # guide sites enumerate first; the model continues on the dims after them
        # (traceenum_elbo.py:352-360: the two EnumMessengers share the global allocator)
        from ..poutine.handlers import SequentialEnumMessenger
        if getattr(self, "_seq_queue", None) is None:
            self._seq_queue = []
        seq = SequentialEnumMessenger(getattr(self, "_seq_assignment", None) or {}, self._seq_queue)
        guide_enum = poutine.enum(seq(guide), first_available_dim=first_enum_dim)
        from ..ops import lazy
        with lazy.watch_histograms():      # (examples/lda.py's word histogram: see ops/lazy.py)
            guide_trace = poutine.trace(guide_enum).get_trace(*args, **kwargs)
        model_enum = poutine.enum(model)
        model_trace = poutine.trace(poutine.replay(model_enum, trace=guide_trace)).get_trace(
            *args, **kwargs)
        if is_validation_enabled():
            check_model_guide_match(model_trace, guide_trace, self.max_plate_nesting)
        guide_trace = prune_subsample_sites(guide_trace)
        model_trace = prune_subsample_sites(model_trace)
        if is_validation_enabled():
            check_site_shapes(model_trace, guide_trace, self.max_plate_nesting)
            check_traceenum_requirements(model_trace, guide_trace)
            _check_local_sampling(model_trace, guide_trace)
            enumerating = any(site["infer"].get("enumerate") for trace in (guide_trace, model_trace)
                              for site in trace.nodes.values() if site["type"] == "sample")
            if self.strict_enumeration_warning and not enumerating:
                warnings.warn(
                    "TraceEnum_ELBO found no sample sites configured for enumeration. If you want to "
                    "enumerate sites, you need to @config_enumerate or set "
                    'infer={"enumerate": "sequential"} or infer={"enumerate": "parallel"}? If you do '
                    "not want to enumerate, consider using Trace_ELBO instead.")
        model_trace._first_enum_dim = first_enum_dim
        return model_trace, guide_trace

    # ---- reference: _compute_model_factors + contract + sum (all DiCE weights are 1) ----------
    def _dice_elbo(self, model_trace, guide_trace, dice, enum_names, enum_ids):
        """sum over cost terms of cost * exp(sum of the DiCE log-factors of the guide sites whose
        plate context is contained in the cost's), summed over plates and enumerated names --
        value: the ELBO estimate, gradient: pathwise + score-function + exact-expectation terms
        (pyro/infer/util.py:264-326, traceenum_elbo.py:112-214).  All tensors are packed Terms."""
        first_enum_dim = model_trace._first_enum_dim

        marginals = {}
        # a guide site enumerated SEQUENTIALLY splits the run into one trace per value; a cost that is
        # not downstream of it (its plate context does not contain the site's) shows up unchanged in
        # every one of those traces and has to be counted once (infer/util.py:238-262)
        repeats = {}
        for site in guide_trace.nodes.values():
            if site["type"] == "sample" and site["infer"].get("enumerate") == "sequential" \
                    and site["infer"].get("_enum_total") is not None:
                o = _ordinal(site)
                repeats[o] = repeats.get(o, 0.0) + math.log(site["infer"]["_enum_total"])

        def once(cost):
            log_denom = sum(v for o, v in repeats.items() if not o <= cost.ordinal)
            return math.exp(-log_denom) if log_denom else 1.0

        def expectation(cost):
            """cost . P(names the cost depends on): the weights of the upstream guide sites, with every
            name the cost does NOT carry summed out first (variable elimination; never the joint
            table over all enumerated guide sites -- infer/util.py:264-326 asks the marginals of the
            same sum-product)."""
            fs = [f for f in dice if f.ordinal <= cost.ordinal]
            if not fs:
                return cost.tensor.sum() * once(cost)
            key = (cost.ordinal, frozenset(cost.ids))
            prob = marginals.get(key)
            ids = sorted(cost.ids)
            if prob is None:
                other = set().union(*(f.ids for f in fs)) - set(cost.ids)
                left = _eliminate(fs, other) if other else fs
                total = None
                for f in left:
                    extra = set(f.ids) - set(ids)
                    assert not extra, extra
                    x = align(f, ids)
                    total = x if total is None else total + x
                prob = marginals[key] = total.exp()
            c = align(cost, ids)
            # zero-probability branches contribute nothing even where the cost is infinite
            c = torch.where(prob > 0, c, torch.zeros((), dtype=c.dtype, device=c.device))
            return (prob * c).sum() * once(cost)

        costs = []
        factors = OrderedDict()
        scales = []
        for name, site in model_trace.nodes.items():
            if site["type"] != "sample":
                continue
            lp = _enum_log_prob(site) if name in enum_names else \
                site["fn"].log_prob(site["value"], *site["args"], **site["kwargs"])
            if name in enum_names:
                # the enumerated site's own factor is its UNmasked, unscaled log-probability
                # (traceenum_elbo.py:168-174): summed over its support it is exactly 0
                term = _packed(site, lp, first_enum_dim)
                factors.setdefault(term.ordinal, []).append(term)
                scales.append(site["scale"])
                continue
            term = _packed(site, scale_and_mask(lp, mask=site["mask"]), first_enum_dim)
            if term.dims & enum_ids:
                # mask inside, scale outside the log-expectation (traceenum_elbo.py:158-167)
                factors.setdefault(term.ordinal, []).append(term)
                scales.append(site["scale"])
            else:
                term.tensor = scale_and_mask(term.tensor, site["scale"], None)
                costs.append(term)
        if factors:
            min_ordinal = frozenset.intersection(*factors.keys())
            for name, site in guide_trace.nodes.items():
                if site["type"] == "sample" and site["infer"].get("_enumerate_dim") is not None:
                    for f in site["cond_indep_stack"]:
                        if f.vectorized and f not in min_ordinal:
                            raise ValueError(
                                "Expected model enumeration to be no more global than guide "
                                "enumeration, but found model enumeration sites upstream of guide "
                                "site '{}' in plate('{}'). Try converting some model enumeration "
                                "sites to guide enumeration sites.".format(name, f.name))
            scale = scales[0]
            for sc in scales[1:]:
                if sc != scale:
                    raise ValueError("Expected all enumerated sample sites to share a common "
                                     "poutine.scale, but found different scales")
            for ordinal, terms in contract_tensor_tree(factors, enum_ids).items():
                for term in terms:
                    if not isinstance(scale, float) or scale != 1.0:
                        term.tensor = term.tensor * scale
                    costs.append(term)
        for name, site in guide_trace.nodes.items():
            if site["type"] != "sample":
                continue
            lq = site["fn"].log_prob(site["value"], *site["args"], **site["kwargs"])
            costs.append(_packed(site, -scale_and_mask(lq, site["scale"], site["mask"]),
                                 first_enum_dim))
        elbo = 0.0
        for c in costs:
            elbo = elbo + expectation(c)
        return elbo

    def _elbo_tensor(self, model_trace, guide_trace):
        first_enum_dim = model_trace._first_enum_dim
        enum_names = [n for n, s in model_trace.nodes.items()
                      if s["type"] == "sample" and s["infer"].get("_enumerate_dim") is not None
                      and n not in guide_trace.nodes]
        enum_ids = {model_trace.nodes[n]["infer"]["_dim_to_id"][
            model_trace.nodes[n]["infer"]["_enumerate_dim"]] for n in enum_names}
        dice = []
        for name, site in guide_trace.nodes.items():
            if site["type"] != "sample":
                continue
            # enumerated in parallel (own tensor dim) or sequentially (one value per trace): the
            # site's probability itself weights the downstream costs
            enumerated = site["infer"].get("_enumerate_dim") is not None or \
                site["infer"].get("_enum_total") is not None
            if enumerated or not getattr(site["fn"], "has_rsample", False):
                lq = site["fn"].log_prob(site["value"], *site["args"], **site["kwargs"])
                lq = scale_and_mask(lq, 1.0, site["mask"])     # masked, never scaled
                draws = site["infer"].get("num_samples") if site["infer"].get("enumerate") else None
                if draws is not None:
                    # n local draws instead of the support: each weighs 1/n, and a draw that is not
                    # reparameterised carries its score function (infer/util.py:176-186)
                    score = lq * 0.0 if getattr(site["fn"], "has_rsample", False) \
                        else lq - lq.detach()
                    dice.append(_packed(site, score - math.log(draws), first_enum_dim))
                elif enumerated:
                    dice.append(_packed(site, lq, first_enum_dim))
                elif lq.requires_grad:
                    dice.append(_packed(site, lq - lq.detach(), first_enum_dim))
        if dice:
            return self._dice_elbo(model_trace, guide_trace, dice, enum_names, enum_ids)
        plain, signs, const = [], [], 0.0
        factors = OrderedDict()
        scales = []
        for name, site in model_trace.nodes.items():
            if site["type"] != "sample":
                continue
            if name in enum_names:
                # unmasked, unscaled (traceenum_elbo.py:168-174): a masked-out plate slice of an
                # enumerated variable must still sum to probability one
                lp = _enum_log_prob(site)
                term = _packed(site, lp, first_enum_dim)
                factors.setdefault(term.ordinal, []).append(term)
                scales.append(site["scale"])
                continue
            lazy = _lazy_gather(site, first_enum_dim) if enum_ids else None
            if lazy is not None and lazy[0] in enum_ids:
                factors.setdefault(_ordinal(site), []).append(
                    Term(None, (lazy[0],), _ordinal(site), lazy=lazy[1]))
                scales.append(site["scale"])
                continue
            lazy = _lazy_family(site, first_enum_dim) if enum_ids else None
            if lazy is not None and lazy[0] in enum_ids:
                # a plated mixture's likelihood: never materialised when the leaf pattern matches
                factors.setdefault(_ordinal(site), []).append(
                    Term(None, (lazy[0],), _ordinal(site), lazy=lazy[1]))
                scales.append(site["scale"])
                continue
            if not enum_ids or not self._depends_on_enum(site, first_enum_dim):
                fused = model_trace._site_sum(name, site)     # one-kernel plate sum
                if isinstance(fused, torch.Tensor):
                    plain.append(fused)
                    signs.append(1.0)
                else:
                    const += fused
                continue
            lp = site["fn"].log_prob(site["value"], *site["args"], **site["kwargs"])
            term = _packed(site, scale_and_mask(lp, mask=site["mask"]), first_enum_dim)
            if term.dims & enum_ids:          # mask inside, scale outside the log-expectation
                factors.setdefault(term.ordinal, []).append(term)
                scales.append(site["scale"])
            else:
                plain.append(scale_and_mask(lp, site["scale"], site["mask"]).sum())
                signs.append(1.0)
        if factors:
            scale = scales[0]
            for s in scales[1:]:
                if s != scale:
                    raise ValueError("Expected all enumerated sample sites to share a common "
                                     "poutine.scale, but found different scales")
            for ordinal, terms in contract_tensor_tree(factors, enum_ids, reduce_all=True).items():
                for term in terms:
                    t = term.tensor.sum()
                    plain.append(t * scale if not isinstance(scale, float) or scale != 1.0 else t)
                    signs.append(1.0)
        guide_trace.compute_log_prob_sums()
        for site in guide_trace.nodes.values():
            if site["type"] == "sample":
                x = site["log_prob_sum"]
                if isinstance(x, torch.Tensor):
                    plain.append(x)
                    signs.append(-1.0)
                else:
                    const -= x
        if not plain:
            return const
        total = _signed_sum(plain, signs)
        return total + const if const != 0.0 else total

    # ---- posterior of the enumerated model sites (reference: traceenum_elbo.py:224-313, 473-520;
    #      MarginalRing / SampleRing of pyro/ops/rings.py:274-329 through the adjoint of the
    #      sum-product).  Here the adjoint IS autograd: the marginal of an enumerated site is the
    #      gradient of the log-partition function with respect to a probe added to that site's
    #      log-factor (for a chain written with pyro.markov the fused kernel pa_logchain_fwd_bwd
    #      returns exactly these posteriors as its gradient), and sampling conditions on the sites
    #      already drawn by adding their indicator as evidence. -----------------------------------
    def _enum_sites(self, model_trace, guide_trace):
        names = [n for n, s in model_trace.nodes.items()
                 if s["type"] == "sample" and s["infer"].get("_enumerate_dim") is not None
                 and n not in guide_trace.nodes]
        ids = {model_trace.nodes[n]["infer"]["_dim_to_id"][
            model_trace.nodes[n]["infer"]["_enumerate_dim"]] for n in names}
        return names, ids

    def _log_partition(self, model_trace, enum_names, enum_ids, extra):
        """log Z of the enumerated part of the model: every factor that carries an enumerated name,
        summed out over names and plates.  ``extra[name]`` is added to site ``name``'s factor."""
        first_enum_dim = model_trace._first_enum_dim
        factors = OrderedDict()
        for name, site in model_trace.nodes.items():
            if site["type"] != "sample":
                continue
            if name in enum_names:
                lp = _enum_log_prob(site)
                if name in extra:
                    lp = lp + extra[name]
                term = _packed(site, lp, first_enum_dim)
                factors.setdefault(term.ordinal, []).append(term)
                continue
            if not self._depends_on_enum(site, first_enum_dim):
                continue
            lp = site["fn"].log_prob(site["value"], *site["args"], **site["kwargs"])
            term = _packed(site, scale_and_mask(lp, mask=site["mask"]), first_enum_dim)
            if term.dims & enum_ids:
                factors.setdefault(term.ordinal, []).append(term)
        total = 0.0
        for ordinal, terms in contract_tensor_tree(factors, enum_ids, reduce_all=True).items():
            for term in terms:
                total = total + term.tensor.sum()
        return total

    def _posterior_of(self, model_trace, enum_names, enum_ids, name, evidence):
        """Normalised posterior of enumerated site ``name`` given the evidence tensors of the
        sites already fixed: [.., K] with the plate dims of the site in front."""
        site = model_trace.nodes[name]
        with torch.enable_grad():
            lp0 = site["fn"].log_prob(site["value"])
            edim0 = site["infer"]["_enumerate_dim"]
            fed = model_trace._first_enum_dim
            # the probe lives on the site's own enumeration dim and its plate dims only: other
            # variables' enumeration dims in the factor (a Markov parent) broadcast
            shape = [s_ if (i - lp0.dim() > fed or i - lp0.dim() == edim0) else 1
                     for i, s_ in enumerate(lp0.shape)]
            probe = torch.zeros(shape, dtype=lp0.dtype, device=lp0.device, requires_grad=True)
            extra = dict(evidence)
            extra[name] = probe if name not in evidence else evidence[name] + probe
            log_z = self._log_partition(model_trace, enum_names, enum_ids, extra)
            (g,) = torch.autograd.grad(log_z, [probe])
        edim = site["infer"]["_enumerate_dim"]
        g = g.clamp(min=0.0)
        p = g / g.sum(edim, keepdim=True)
        p = p.unsqueeze(-1).transpose(-1, edim - 1)
        while p.dim() > 1 and p.shape[0] == 1:
            p = p.squeeze(0)
        return p

    @staticmethod
    def _make_dist(fn, probs):
        """The site's distribution family with the given probabilities over its support."""
        from .. import distributions as dist
        import torch.distributions as td
        base = fn
        while hasattr(base, "base_dist"):
            base = base.base_dist
        if isinstance(base, td.Bernoulli):
            return dist.Bernoulli(probs=probs[..., 1])
        if isinstance(base, td.OneHotCategorical):
            return dist.OneHotCategorical(probs=probs)
        if isinstance(base, td.Categorical):
            return dist.Categorical(probs=probs)
        raise NotImplementedError("marginals of an enumerated {} site".format(type(base).__name__))

    def _posterior_traces(self, what, model, guide, args, kwargs):
        if self.num_particles != 1:
            raise NotImplementedError("TraceEnum_ELBO.{}() is not compatible with multiple "
                                      "particles.".format(what))
        model_trace, guide_trace = next(iter(self._get_traces(model, guide, args, kwargs)))
        for site in guide_trace.nodes.values():
            if site["type"] == "sample" and ("_enumerate_dim" in site["infer"]
                                             or "_enum_total" in site["infer"]):
                raise NotImplementedError("TraceEnum_ELBO.{}() is not compatible with guide "
                                          "enumeration.".format(what))
        return model_trace, guide_trace

    def compute_marginals(self, model, guide, *args, **kwargs):
        """Marginal distribution of every model-enumerated sample site given all observations:
        an OrderedDict site name -> distribution of the site's family."""
        model_trace, guide_trace = self._posterior_traces("compute_marginals", model, guide, args,
                                                          kwargs)
        enum_names, enum_ids = self._enum_sites(model_trace, guide_trace)
        out = OrderedDict()
        for name in enum_names:
            probs = self._posterior_of(model_trace, enum_names, enum_ids, name, {})
            out[name] = self._make_dist(model_trace.nodes[name]["fn"], probs.detach())
        return out

    def sample_posterior(self, model, guide, *args, **kwargs):
        """One joint draw of all model-enumerated sites from their posterior given the
        observations (forward filtering / backward sampling in the order the model visits the
        sites); returns what the model returns with those sites at their drawn values."""
        import warnings
        from ..poutine.runtime import Messenger
        with poutine.block(), warnings.catch_warnings():
            warnings.filterwarnings("ignore", "Found vars in model but not guide")
            model_trace, guide_trace = self._posterior_traces("sample_posterior", model, guide,
                                                              args, kwargs)
        enum_names, enum_ids = self._enum_sites(model_trace, guide_trace)
        elbo = self

        class _BackwardSample(Messenger):
            def __init__(self):
                super().__init__()
                self.evidence = {}

            def _pyro_sample(self, msg):
                name = msg["name"]
                if name not in enum_names or msg["is_observed"]:
                    return
                probs = elbo._posterior_of(model_trace, enum_names, enum_ids, name, self.evidence)
                msg["fn"] = elbo._make_dist(msg["fn"], probs.detach())
                msg["infer"] = dict(msg["infer"])
                msg["infer"].pop("enumerate", None)        # an ordinary draw from the posterior

            def _pyro_post_sample(self, msg):
                name = msg["name"]
                if name not in enum_names or msg["is_observed"]:
                    return
                site = model_trace.nodes[name]
                support, value = site["value"], msg["value"]
                ev = len(site["fn"].event_shape)
                # indicator of the drawn value on the site's enumerated support, shaped like the
                # site's log-factor: 0 where the support equals the draw, -inf elsewhere
                v = value.to(support.dtype)
                hit = (support == v)
                for _ in range(ev):
                    hit = hit.all(-1)
                neg = torch.full((), -float("inf"), dtype=torch.get_default_dtype(), device=hit.device)
                zero = torch.zeros((), dtype=neg.dtype, device=hit.device)
                lp_dtype = site["fn"].log_prob(support).dtype
                self.evidence[name] = torch.where(hit, zero, neg).to(lp_dtype)

        with _BackwardSample():
            return poutine.replay(model, trace=guide_trace)(*args, **kwargs)

    @staticmethod
    def _depends_on_enum(site, first_enum_dim):
        """A site whose distribution batch shape or value reaches into the enumerated dims."""
        fn = site["fn"]
        bs = getattr(fn, "batch_shape", ())
        n = len(bs)
        if any(bs[i] > 1 for i in range(n) if i - n <= first_enum_dim):
            return True
        v = site["value"]
        ev = len(getattr(fn, "event_shape", ()))
        n = v.dim() - ev
        return any(v.shape[i] > 1 for i in range(n) if i - n <= first_enum_dim)

    def loss(self, model, guide, *args, **kwargs):
        elbo = 0.0
        with torch.no_grad():
            for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
                elbo = elbo + self._elbo_tensor(model_trace, guide_trace) / self.num_particles
        loss = -torch_item(elbo)
        warn_if_nan(loss, "loss")
        return loss

    def differentiable_loss(self, model, guide, *args, **kwargs):
        elbo = 0.0
        for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
            elbo = elbo + self._elbo_tensor(model_trace, guide_trace) / self.num_particles
        loss = -elbo
        warn_if_nan(loss, "loss")
        return loss

    @_grad_sink()
    def loss_and_grads_device(self, model, guide, *args, **kwargs):
        loss = None
        c = -1.0 / self.num_particles
        for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
            e = self._elbo_tensor(model_trace, guide_trace)
            if not isinstance(e, torch.Tensor):
                term = e * c
            else:
                sl = e * c
                term = sl.detach()
                if sl.requires_grad:
                    sl.backward(retain_graph=self.retain_graph)
            loss = term if loss is None else loss + term
        return 0.0 if loss is None else loss

    def loss_and_grads(self, model, guide, *args, **kwargs):
        loss = torch_item(self.loss_and_grads_device(model, guide, *args, **kwargs))
        warn_if_nan(loss, "loss")
        return loss
