"""Importance-trace construction shared by the ELBO estimators
(reference: pyro/infer/enum.py:45-85 get_importance_trace, :138-220 config_enumerate)."""
from .util import is_validation_enabled
import numbers

from .. import poutine
from ..poutine.util import prune_subsample_sites
from ..util import check_model_guide_match, check_site_shape  # noqa: F401


def check_site_shapes(model_trace, guide_trace, max_plate_nesting):
    """``check_site_shape`` at every sample site of both traces (enum.py:76-83).  Needs no un-reduced
    log_prob: on the fused path the shape is derived from the distribution and the value."""
    for trace in (model_trace, guide_trace):
        for site in trace.nodes.values():
            if site["type"] == "sample":
                check_site_shape(site, max_plate_nesting)


def get_importance_trace(graph_type, max_plate_nesting, model, guide, args, kwargs, detach=False,
                         fused_sums=False):
    """Guide trace, then the model replayed against it.  With ``fused_sums`` the per-site
    quantities are produced by the fused one-kernel reductions (log_prob_sum only)."""
    guide_trace = poutine.trace(guide, graph_type=graph_type).get_trace(*args, **kwargs)
    if detach:
        guide_trace.detach_()
    model_trace = poutine.trace(poutine.replay(model, trace=guide_trace),
                                graph_type=graph_type).get_trace(*args, **kwargs)
    if is_validation_enabled():
        check_model_guide_match(model_trace, guide_trace, max_plate_nesting)
    guide_trace = prune_subsample_sites(guide_trace)
    model_trace = prune_subsample_sites(model_trace)
    if fused_sums == "defer":
        pass        # the caller batches the per-site sums (Trace.collect_log_prob_sums)
    elif fused_sums:
        model_trace.compute_log_prob_sums()
    else:
        model_trace.compute_log_prob()
        guide_trace.compute_score_parts()
    if is_validation_enabled():
        check_site_shapes(model_trace, guide_trace, max_plate_nesting)
    return model_trace, guide_trace


def _config_fn(default, expand, num_samples, tmc):
    def fn(site):
        if site["type"] != "sample" or site["is_observed"]:
            return {}
        if not getattr(site["fn"], "has_enumerate_support", False):
            return {}
        if site["infer"].get("enumerate") is not None:
            return {}
        return {"enumerate": default, "expand": expand}

    return fn


def config_enumerate(guide=None, default="parallel", expand=False, num_samples=None, tmc="diagonal"):
    """Mark every enumerable site of ``guide`` (or model) for enumeration."""
    if default not in ("sequential", "parallel", None):
        raise ValueError("Invalid default value. Expected 'sequential', 'parallel', or None")
    if num_samples is not None:
        # (reference: enum.py:138-220 + enumerate_site's Monte-Carlo branch) -- not built; summing
        # the support exactly instead would silently change shapes and variance
        raise NotImplementedError("pyro_amd: config_enumerate(num_samples=...) (Monte-Carlo "
                                  "enumeration) is not implemented; enumerate exactly")
    if guide is None:
        return lambda g: config_enumerate(g, default=default, expand=expand,
                                          num_samples=num_samples, tmc=tmc)

    cfg = _config_fn(default, expand, num_samples, tmc)

    class _Infer(poutine.Messenger):
        def _pyro_sample(self, msg):
            msg["infer"].update(cfg(msg))

    return _Infer()(guide)
