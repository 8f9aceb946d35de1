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
import torch

from ..distributions.util import is_identically_zero
from ..util import torch_item, warn_if_nan
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


class Trace_ELBO(ELBO):
    def _guide_is_reparameterized(self, guide_trace):
        for site in guide_trace.nodes.values():
            if site["type"] == "sample" and not site["is_observed"] \
                    and not getattr(site["fn"], "has_rsample", False):
                return False
        return True

    def _get_trace(self, model, guide, args, kwargs):
        # fused sums for the model always; guide handled after inspecting reparameterisation
        model_trace, guide_trace = get_importance_trace("flat", self.max_plate_nesting, model,
                                                        guide, args, kwargs, fused_sums=True)
        if self._guide_is_reparameterized(guide_trace):
            guide_trace.compute_log_prob_sums()
            guide_trace._fully_reparam = True
        else:
            model_trace.compute_log_prob()
            guide_trace.compute_score_parts()
            guide_trace._fully_reparam = False
        return model_trace, guide_trace

    # ---- per-particle (or per vectorised batch of particles) terms ---------------------------
    def _surrogate_and_elbo(self, model_trace, guide_trace):
        """Returns (elbo tensor, surrogate elbo tensor), both 0-dim on the device."""
        elbo = 0.0
        surrogate = 0.0
        for site in model_trace.nodes.values():
            if site["type"] == "sample":
                elbo = elbo + site["log_prob_sum"].detach()
                surrogate = surrogate + site["log_prob_sum"]
        if getattr(guide_trace, "_fully_reparam", False):
            for site in guide_trace.nodes.values():
                if site["type"] == "sample":
                    # entropy_term == log_prob for reparameterised sites (distribution.py:98-125)
                    elbo = elbo - site["log_prob_sum"].detach()
                    surrogate = surrogate - site["log_prob_sum"]
            return elbo, surrogate
        log_r = None
        for name, site in guide_trace.nodes.items():
            if site["type"] != "sample":
                continue
            log_prob, score_function_term, entropy_term = site["score_parts"]
            elbo = elbo - site["log_prob_sum"].detach()
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
        return loss + (surrogate_loss - surrogate_loss.detach())

    def loss_and_grads(self, model, guide, *args, **kwargs):
        """Backward on the surrogate; returns the ELBO estimate as a float (one host sync)."""
        loss = 0.0
        for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
            e, s = self._surrogate_and_elbo(model_trace, guide_trace)
            loss = loss - e / self.num_particles
            trainable = any(site["type"] == "param" for trace in (model_trace, guide_trace)
                            for site in trace.nodes.values())
            if trainable and getattr(s, "requires_grad", False):
                (-s / self.num_particles).backward(retain_graph=self.retain_graph)
        loss = torch_item(loss)
        warn_if_nan(loss, "loss")
        return loss
