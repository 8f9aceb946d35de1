"""Importance-trace construction shared by the ELBO estimators
(reference: pyro/infer/enum.py:45-85 get_importance_trace, :138-220 config_enumerate)."""
import numbers

from .. import poutine
from ..poutine.util import prune_subsample_sites


def get_importance_trace(graph_type, max_plate_nesting, model, guide, args, kwargs, detach=False,
                         fused_sums=False):
    """Guide trace, then the model replayed against it.  With ``fused_sums`` the per-site
    quantities are produced by the fused one-kernel reductions (log_prob_sum only)."""
    guide_trace = poutine.trace(guide, graph_type=graph_type).get_trace(*args, **kwargs)
    if detach:
        guide_trace.detach_()
    model_trace = poutine.trace(poutine.replay(model, trace=guide_trace),
                                graph_type=graph_type).get_trace(*args, **kwargs)
    if poutine.settings.validation_enabled():
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
    return model_trace, guide_trace


def check_model_guide_match(model_trace, guide_trace, max_plate_nesting=float("inf")):
    """Site-level sanity checks (reference: pyro/util.py:314-398, condensed)."""
    import warnings

    guide_vars = {name for name, site in guide_trace.nodes.items()
                  if site["type"] == "sample" and not poutine.util.site_is_subsample(site)}
    aux_vars = {name for name, site in guide_trace.nodes.items()
                if site["type"] == "sample" and site["infer"].get("is_auxiliary")}
    model_vars = {name for name, site in model_trace.nodes.items()
                  if site["type"] == "sample" and not site["is_observed"]
                  and not poutine.util.site_is_subsample(site)}
    enum_vars = {name for name, site in model_trace.nodes.items()
                 if site["type"] == "sample" and not site["is_observed"]
                 and site["infer"].get("_enumerate_dim") is not None
                 and name not in guide_vars}
    if aux_vars & model_vars:
        warnings.warn("Found auxiliary vars in the model: {}".format(aux_vars & model_vars))
    if not (guide_vars <= model_vars | aux_vars):
        warnings.warn("Found non-auxiliary vars in guide but not model, consider marking these "
                      "infer={{'is_auxiliary': True}}:\n{}".format(guide_vars - aux_vars - model_vars))
    if not (model_vars <= guide_vars | enum_vars):
        warnings.warn("Found vars in model but not guide: {}".format(
            model_vars - guide_vars - enum_vars))
    for name in model_vars & guide_vars:
        m, g = model_trace.nodes[name], guide_trace.nodes[name]
        if hasattr(m["fn"], "shape") and hasattr(g["fn"], "shape"):
            ms, gs = m["fn"].shape(*m["args"], **m["kwargs"]) if False else m["fn"].shape(), \
                g["fn"].shape()
            if ms != gs and not (m["infer"].get("_enumerate_dim") is not None):
                # allow broadcastable differences only
                for a, b in zip(reversed(ms), reversed(gs)):
                    if a != b and a != 1 and b != 1:
                        raise ValueError("Model and guide shapes disagree at site '{}': {} vs {}"
                                         .format(name, tuple(ms), tuple(gs)))


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
