"""TraceGraph_ELBO: the score-function ELBO estimator with baselines
(reference: pyro/infer/tracegraph_elbo.py:28-101 baselines, :178-236 _compute_elbo, :290-380).

For every non-reparameterised guide site z the surrogate carries
    log q(z) * stop_gradient(downstream_cost(z) - baseline(z))
and, for trainable baselines, the regression loss (downstream_cost - baseline)^2.  Baselines are
configured per site exactly as in the reference, through
``infer={"baseline": {"use_decaying_avg_baseline": True, "baseline_beta": 0.9, "baseline_value":
tensor, "nn_baseline": module, "nn_baseline_input": tensor}}``.

Dependency structure (tracegraph_elbo.py:178-236).  The downstream cost of z is the sum of the
log p / -log q terms that DEPEND on z -- found by data-flow provenance: values drawn at
non-reparameterised sites are wrapped by ``TrackNonReparam`` in ``ops.provenance.ProvenanceTensor``,
every torch function propagates the set of site names, and a term belongs to z's downstream cost
when z is in the provenance of the term's site (its value, its distribution's parameters) --
reduced to the plates of z (MultiFrameTensor.sum_to).  With a fully reparameterised guide this class
is Trace_ELBO, fused paths included.
"""
import torch

from collections import defaultdict

from ..distributions.util import is_identically_zero
from ..ops.provenance import detach_provenance, site_provenance, track_provenance
from ..params import _PARAM_STORE
from ..poutine.runtime import Messenger
from ..poutine.util import site_is_subsample
from .trace_elbo import Trace_ELBO
from .util import MultiFrameTensor


class TrackNonReparam(Messenger):
    """Tags the value of every non-reparameterised latent sample site with the site's name
    (tracegraph_elbo.py:239-287); whatever is computed from it carries the tag on."""

    def _pyro_post_sample(self, msg):
        if msg["type"] == "sample" and not site_is_subsample(msg) and not msg["is_observed"] \
                and not getattr(msg["fn"], "has_rsample", False):
            msg["value"] = track_provenance(msg["value"], frozenset({msg["name"]}))


def _get_baseline_options(site):
    options = dict(site["infer"].get("baseline", {}))
    out = (options.pop("nn_baseline", None), options.pop("nn_baseline_input", None),
           options.pop("use_decaying_avg_baseline", False), options.pop("baseline_beta", 0.90),
           options.pop("baseline_value", None))
    if options:
        raise ValueError("Unrecognized baseline options: {}".format(options.keys()))
    return out


def _construct_baseline(name, guide_site, downstream_cost):
    """(use_baseline, baseline_loss, baseline) of one site (tracegraph_elbo.py:48-100)."""
    nn_baseline, nn_input, use_avg, beta, value = _get_baseline_options(guide_site)
    baseline, baseline_loss = 0.0, 0.0
    assert not (nn_baseline is not None and value is not None), \
        "cannot use baseline_value and nn_baseline simultaneously"
    if use_avg:
        pname = "__baseline_avg_downstream_cost_" + name
        with torch.no_grad():
            if pname not in _PARAM_STORE:
                _PARAM_STORE.setdefault(pname, torch.zeros_like(downstream_cost))
            old = _PARAM_STORE[pname].detach()
            _PARAM_STORE[pname] = (1 - beta) * downstream_cost + beta * old
        baseline = baseline + old
    if nn_baseline is not None:
        # the baseline's input is detached: only the baseline loss trains the network
        baseline = baseline + nn_baseline(nn_input.detach())
    elif value is not None:
        baseline = baseline + value
    if nn_baseline is not None or value is not None:
        baseline_loss = torch.pow(downstream_cost.detach() - baseline, 2.0).sum()
    use = use_avg or nn_baseline is not None or value is not None
    if use and isinstance(baseline, torch.Tensor) and isinstance(downstream_cost, torch.Tensor) \
            and downstream_cost.shape != baseline.shape:
        raise ValueError("Expected baseline at site {} to be {} instead got {}".format(
            name, tuple(downstream_cost.shape), tuple(baseline.shape)))
    return use, baseline_loss, baseline


class TraceGraph_ELBO(Trace_ELBO):
    def _get_trace(self, model, guide, args, kwargs):
        with TrackNonReparam():
            return super()._get_trace(model, guide, args, kwargs)

    def _surrogate_and_elbo(self, model_trace, guide_trace):
        if getattr(guide_trace, "_fully_reparam", False):
            return super()._surrogate_and_elbo(model_trace, guide_trace)
        model_trace.compute_log_prob()
        elbo, surrogate = 0.0, 0.0
        downstream = defaultdict(MultiFrameTensor)       # non-reparam site -> the costs it influences
        for site in model_trace.nodes.values():
            if site["type"] != "sample":
                continue
            x = detach_provenance(site["log_prob_sum"])
            elbo = elbo + (x.detach() if isinstance(x, torch.Tensor) else x)
            surrogate = surrogate + x
            for key in site_provenance(site):
                downstream[key].add((site["cond_indep_stack"],
                                     detach_provenance(site["log_prob"]).detach()))
        for name, site in guide_trace.nodes.items():
            if site["type"] != "sample":
                continue
            entropy_term = site["score_parts"].entropy_term
            lps = detach_provenance(site["log_prob_sum"])
            elbo = elbo - (lps.detach() if isinstance(lps, torch.Tensor) else lps)
            if not is_identically_zero(entropy_term):
                surrogate = surrogate - detach_provenance(entropy_term).sum()
            for key in site_provenance(site):
                downstream[key].add((site["cond_indep_stack"],
                                     -detach_provenance(site["log_prob"]).detach()))
        for name, cost in downstream.items():
            site = guide_trace.nodes[name]
            downstream_cost = cost.sum_to(site["cond_indep_stack"])
            score_function_term = detach_provenance(site["score_parts"].score_function)
            use, baseline_loss, baseline = _construct_baseline(name, site, downstream_cost)
            if use:
                downstream_cost = downstream_cost - (
                    baseline.detach() if isinstance(baseline, torch.Tensor) else baseline)
            surrogate = surrogate + (score_function_term * downstream_cost).sum()
            # the surrogate is MAXIMISED: the baseline regression loss enters with a minus
            surrogate = surrogate - baseline_loss
        return elbo, surrogate
