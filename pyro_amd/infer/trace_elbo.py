"""Trace_ELBO: the plugin behind SVI for the plated-model hot path
(reference: pyro/infer/trace_elbo.py:20-159).

Same estimator (pathwise term + score-function term with plate-aware Rao-Blackwellisation
through MultiFrameTensor); the difference is how the per-site terms are produced:

  * when every guide site is reparameterised (the common case: AutoNormal & friends) only the
    *sums* of the scaled+masked log-probabilities are needed, so model and guide sites go
    through ``fused_log_prob_sum`` (one HIP kernel per site for forward, one for backward; the
    observed GLM site: one kernel for both) and the loss is assembled on the device with a single
    host synchronisation per step (the reference synchronises once per site,
    trace_elbo.py:90,97);
  * otherwise (non-reparameterised guide sites) the un-reduced path of the reference is taken.
"""
from .util import is_validation_enabled
import torch

from ..distributions.fused import grad_sink as _grad_sink

from ..distributions.util import is_identically_zero
from ..util import check_if_enumerated, torch_item, warn_if_nan
from .elbo import ELBO
from .enum import get_importance_trace
from .util import MultiFrameTensor, get_plate_stacks


def _compute_log_r(model_trace, guide_trace):
    log_r = MultiFrameTensor()
    stacks = get_plate_stacks(model_trace)
    for name, model_site in model_trace.nodes.items():
        if model_site["type"] == "sample":
            log_r_term = model_site["log_prob"]
            if not model_site["is_observed"]:
                log_r_term = log_r_term - guide_trace.nodes[name]["log_prob"]
            log_r.add((stacks[name], log_r_term.detach()))
    return log_r


_SIGN_CACHE = {}
_UNIT_GRADS = {}


def _unit_grad(x):
    """Cached 0-dim one used as the root gradient of backward() (autograd would launch a fill
    kernel for it on every step)."""
    key = (x.dtype, x.device)
    g = _UNIT_GRADS.get(key)
    if g is None:
        g = torch.ones((), dtype=x.dtype, device=x.device)
        _UNIT_GRADS[key] = g
    return g


def _signed_sum(terms, signs):
    """sum_i signs[i] * terms[i] for 0-dim tensors: one stack + one dot (the sign vector is cached
    on the device per pattern)."""
    t0 = terms[0]
    key = (tuple(signs), t0.dtype, t0.device)
    sv = _SIGN_CACHE.get(key)
    if sv is None:
        sv = torch.tensor(signs, dtype=t0.dtype, device=t0.device)
        _SIGN_CACHE[key] = sv
    return torch.dot(torch.stack(terms), sv)


class Trace_ELBO(ELBO):
    def _guide_is_reparameterized(self, guide_trace):
        for site in guide_trace.nodes.values():
            if site["type"] == "sample" and not site["is_observed"] \
                    and not getattr(site["fn"], "has_rsample", False):
                return False
        return True

    def _get_trace(self, model, guide, args, kwargs):
        # nothing is scored yet: the fully-reparameterised case batches all per-site sums into one
        # launch (see _batched_total), the general case takes the un-reduced path of the reference
        model_trace, guide_trace = get_importance_trace("flat", self.max_plate_nesting, model,
                                                        guide, args, kwargs, fused_sums="defer")
        if is_validation_enabled():
            check_if_enumerated(guide_trace)
        if self._guide_is_reparameterized(guide_trace):
            guide_trace._fully_reparam = True
        else:
            model_trace.compute_log_prob()
            guide_trace.compute_score_parts()
            guide_trace._fully_reparam = False
        return model_trace, guide_trace

    @staticmethod
    def _batched_total(model_trace, guide_trace, coef=1.0):
        """coef * (sum of model log_prob_sums - sum of guide log_prob_sums) as a 0-dim device
        tensor (or a float if no site produced a tensor): every small site and the reduced terms of
        the large ones in ONE fused launch, forward and backward (distributions.fused.SiteBatch)."""
        from ..distributions.fused import SiteBatch
        from ..poutine import settings

        batch = SiteBatch()
        left = model_trace.collect_log_prob_sums(batch, 1.0)
        left += guide_trace.collect_log_prob_sums(batch, -1.0)
        total = batch.total(coef)
        for sign, term in left:
            total = total + (coef * sign) * term
        if is_validation_enabled() and isinstance(total, torch.Tensor) \
                and not bool(torch.isfinite(total.detach())):
            # name the offending site(s) the way the reference does (trace_struct.py:279-286)
            with torch.no_grad():
                model_trace.compute_log_prob_sums()
                guide_trace.compute_log_prob_sums()
        return total

    # ---- per-particle (or per vectorised batch of particles) terms ---------------------------
    def _surrogate_and_elbo(self, model_trace, guide_trace):
        """Returns (elbo tensor, surrogate elbo tensor), both 0-dim on the device."""
        if getattr(guide_trace, "_fully_reparam", False):
            # every term enters elbo and surrogate alike (entropy_term == log_prob for
            # reparameterised sites, distribution.py:98-125)
            total = self._batched_total(model_trace, guide_trace)
            if not isinstance(total, torch.Tensor):
                return total, total
            return total.detach(), total
        model_trace.compute_log_prob_sums()
        elbo = 0.0
        surrogate = 0.0
        for site in model_trace.nodes.values():
            if site["type"] == "sample":
                x = site["log_prob_sum"]
                elbo = elbo + (x.detach() if isinstance(x, torch.Tensor) else x)
                surrogate = surrogate + x
        log_r = None
        for name, site in guide_trace.nodes.items():
            if site["type"] != "sample":
                continue
            log_prob, score_function_term, entropy_term = site["score_parts"]
            lps = site["log_prob_sum"]
            elbo = elbo - (lps.detach() if isinstance(lps, torch.Tensor) else lps)
            if not is_identically_zero(entropy_term):
                surrogate = surrogate - entropy_term.sum()
            if not is_identically_zero(score_function_term):
                if log_r is None:
                    log_r = _compute_log_r(model_trace, guide_trace)
                r = log_r.sum_to(site["cond_indep_stack"])
                surrogate = surrogate + (r * score_function_term).sum()
        return elbo, surrogate

    def loss(self, model, guide, *args, **kwargs):
        elbo = 0.0
        with torch.no_grad():
            for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
                e, _ = self._surrogate_and_elbo(model_trace, guide_trace)
                elbo = elbo + e / self.num_particles
        loss = -torch_item(elbo)
        warn_if_nan(loss, "loss")
        return loss

    def differentiable_loss(self, model, guide, *args, **kwargs):
        loss = 0.0
        surrogate_loss = 0.0
        for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
            e, s = self._surrogate_and_elbo(model_trace, guide_trace)
            surrogate_loss = surrogate_loss - s / self.num_particles
            loss = loss - e / self.num_particles
        warn_if_nan(surrogate_loss, "loss")
        if not isinstance(surrogate_loss, torch.Tensor):       # a pair without sample sites
            return loss
        return loss + (surrogate_loss - surrogate_loss.detach())

    def loss_and_grads(self, model, guide, *args, **kwargs):
        """Backward on the surrogate; returns the ELBO estimate as a float (one host sync)."""
        loss = torch_item(self.loss_and_grads_device(model, guide, *args, **kwargs))
        warn_if_nan(loss, "loss")
        return loss

    @_grad_sink()
    def loss_and_grads_device(self, model, guide, *args, **kwargs):
        """Same, but the loss stays a 0-dim device tensor and nothing synchronises with the host
        (what a captured hipGraph step needs, see SVI(hip_graph=True))."""
        loss = None
        c = -1.0 / self.num_particles
        for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
            trainable = any(site["type"] == "param" for trace in (model_trace, guide_trace)
                            for site in trace.nodes.values())
            if getattr(guide_trace, "_fully_reparam", False):
                # surrogate loss == loss value; the -1/num_particles factor rides in the kernel.
                # backward() follows immediately with a unit root gradient, so the assembly may
                # produce its gradients in the forward launch (fused.EAGER_GRAD)
                from ..distributions import fused
                eager = fused.EAGER_GRAD
                eager["on"] = bool(trainable) and torch.is_grad_enabled()
                try:
                    sl = self._batched_total(model_trace, guide_trace, coef=c)
                finally:
                    eager["on"] = False
                if isinstance(sl, torch.Tensor):
                    eager["unit_ptr"] = _unit_grad(sl).data_ptr()
                if not isinstance(sl, torch.Tensor):
                    term, sl = sl, None
                else:
                    term = sl.detach()
            else:
                e, s = self._surrogate_and_elbo(model_trace, guide_trace)
                sl = s * c if isinstance(s, torch.Tensor) else None
                term = e * c
            loss = term if loss is None else loss + term
            if trainable and sl is not None and sl.requires_grad:
                sl.backward(_unit_grad(sl), retain_graph=self.retain_graph)
        return 0.0 if loss is None else loss


class JitTrace_ELBO(Trace_ELBO):
    """Trace_ELBO whose ``differentiable_loss`` is recorded once with ``torch.jit.trace`` and replayed
    (reference: pyro/infer/trace_elbo.py:162-257, on pyro/ops/jit.py).  The recorded graph consists of
    ``pyro_amd::*`` dispatcher ops (ops/torch_library.py: guide draw, ELBO assembly, the fused GLM site,
    ...) and ATen glue; gradients of a replay come from the ops' registered autograd formulas.

    Same limits as the reference: static model structure, tensor inputs as positional arguments,
    everything else as keyword arguments (one trace per distinct set).  ``SVI(hip_graph=True)`` is the
    faster way to run a fixed step on this backend (one hipGraph replay, no operator dispatch at all);
    this class is the drop-in for code written against the reference's JIT estimators."""

    def _traced(self, model, guide):
        import weakref

        from ..ops import jit
        key = (id(model), id(guide))
        cache = self.__dict__.setdefault("_jit_cache", {})
        if key not in cache:
            weakself = weakref.ref(self)

            @jit.trace(ignore_warnings=self.ignore_jit_warnings, jit_options=self.jit_options)
            def differentiable_loss(*args, **kwargs):
                me = weakself()
                return Trace_ELBO.differentiable_loss(me, model, guide, *args, **kwargs)

            cache[key] = (differentiable_loss, model, guide)     # (the ids stay valid while cached)
        return cache[key][0]

    def differentiable_loss(self, model, guide, *args, **kwargs):
        return self._traced(model, guide)(*args, **kwargs)

    def loss(self, model, guide, *args, **kwargs):
        with torch.no_grad():
            return torch_item(self.differentiable_loss(model, guide, *args, **kwargs))

    def loss_and_grads(self, model, guide, *args, **kwargs):
        loss = self.differentiable_loss(model, guide, *args, **kwargs)
        if isinstance(loss, torch.Tensor) and loss.requires_grad:
            loss.backward(retain_graph=self.retain_graph)
        loss = torch_item(loss)
        warn_if_nan(loss, "loss")
        return loss

    loss_and_grads_device = None          # (a traced estimator is not captured into a hipGraph as well)
