"""TraceGraph_ELBO: the score-function ELBO estimator with baselines
(reference: pyro/infer/tracegraph_elbo.py:28-101 baselines, :178-236 _compute_elbo, :290-380).

For every non-reparameterised guide site z the surrogate carries
    log q(z) * stop_gradient(downstream_cost(z) - baseline(z))
and, for trainable baselines, the regression loss (downstream_cost - baseline)^2.  Baselines are
configured per site exactly as in the reference, through
``infer={"baseline": {"use_decaying_avg_baseline": True, "baseline_beta": 0.9, "baseline_value":
tensor, "nn_baseline": module, "nn_baseline_input": tensor}}``.

Dependency structure.  The downstream cost here is Rao-Blackwellised by the PLATE structure -- the
sum of every log p - log q term, reduced to the plates of z (what Trace_ELBO uses,
trace_elbo.py:20-29 + MultiFrameTensor.sum_to) -- whereas the reference additionally drops terms
that data-flow provenance (pyro/ops/provenance.py) shows to be independent of z.  Both are unbiased
estimators of the same gradient; the reference's has the lower variance on models with long chains
of dependent non-reparameterised sites.  With a fully reparameterised guide this class is
Trace_ELBO, fused paths included.
"""
import torch

from ..distributions.util import is_identically_zero
from ..params import _PARAM_STORE
from .trace_elbo import Trace_ELBO, _compute_log_r


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
    def _surrogate_and_elbo(self, model_trace, guide_trace):
        if getattr(guide_trace, "_fully_reparam", False):
            return super()._surrogate_and_elbo(model_trace, guide_trace)
        model_trace.compute_log_prob_sums()
        elbo, surrogate = 0.0, 0.0
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
                downstream_cost = log_r.sum_to(site["cond_indep_stack"])
                use, baseline_loss, baseline = _construct_baseline(name, site, downstream_cost)
                if use:
                    downstream_cost = downstream_cost - (
                        baseline.detach() if isinstance(baseline, torch.Tensor) else baseline)
                surrogate = surrogate + (downstream_cost * score_function_term).sum()
                # the surrogate is MAXIMISED: the baseline regression loss enters with a minus
                surrogate = surrogate - baseline_loss
        return elbo, surrogate
