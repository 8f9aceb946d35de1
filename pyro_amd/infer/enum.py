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


def iter_discrete_escape(trace, msg):
    """A latent site marked for sequential enumeration that ``trace`` does not hold yet."""
    return (msg["type"] == "sample" and not msg["is_observed"]
            and msg["infer"].get("enumerate") == "sequential" and msg["name"] not in trace)


def iter_discrete_extend(trace, site, **ignored):
    """One extension of ``trace`` per value in the support of ``site`` (each carrying the size of the
    support as ``infer["_enum_total"]``)."""
    support = site["fn"].enumerate_support(expand=bool(site["infer"].get("expand", False)))
    for value in support:
        chosen = site.copy()
        chosen["infer"] = dict(site["infer"], _enum_total=support.shape[0])
        chosen["value"] = value
        longer = trace.copy()
        longer.add_node(site["name"], **chosen)
        yield longer


def iter_discrete_traces(graph_type, fn, *args, **kwargs):
    """All traces of ``fn`` over the joint support of its sequentially enumerated sites, depth first
    (reference: pyro/infer/enum.py:88-111, on poutine.queue)."""
    from queue import LifoQueue
    pending = LifoQueue()
    pending.put(poutine.Trace())
    traced = poutine.trace(poutine.queue(fn, pending, escape_fn=iter_discrete_escape,
                                         extend_fn=iter_discrete_extend), graph_type=graph_type)
    while not pending.empty():
        yield traced.get_trace(*args, **kwargs)


def _config_fn(default, expand, num_samples, tmc):
    def fn(site):
        if site["type"] != "sample" or site["is_observed"]:
            return {}
        if type(site["fn"]).__name__ == "_Subsample":
            return {}
        infer = site["infer"]
        if num_samples is not None:         # local sampling applies to every latent site
            return {"enumerate": infer.get("enumerate", default),
                    "num_samples": infer.get("num_samples", num_samples),
                    "expand": infer.get("expand", expand), "tmc": infer.get("tmc", tmc)}
        if getattr(site["fn"], "has_enumerate_support", False):
            return {"enumerate": infer.get("enumerate", default),
                    "expand": infer.get("expand", expand)}
        return {}

    return fn


def config_enumerate(guide=None, default="parallel", expand=False, num_samples=None, tmc="diagonal"):
    """Mark the sites of ``guide`` (or of a model) for enumeration: every site that can enumerate its
    support -- or, with ``num_samples=n``, every latent site for n local Monte-Carlo draws on an
    enumeration dim.  A site's own ``infer`` entries win.  Usable as a decorator
    (reference: pyro/infer/enum.py:138-220)."""
    if default not in ("sequential", "parallel", "flat", None):
        raise ValueError("Invalid default value. Expected 'sequential', 'parallel', or None, but got "
                         "{}".format(repr(default)))
    if expand not in (True, False):
        raise ValueError("Invalid expand value. Expected True or False, but got {}".format(repr(expand)))
    if num_samples is not None:
        if not (isinstance(num_samples, numbers.Number) and num_samples > 0):
            raise ValueError("Invalid num_samples, expected None or positive integer, but got "
                             "{}".format(repr(num_samples)))
        if default == "sequential":
            raise ValueError('Local sampling does not support "sequential" sampling; use "parallel" '
                             "sampling instead.")
    if tmc == "full" and num_samples is not None and num_samples > 1:
        expand = True
    if guide is None:
        return lambda g: config_enumerate(g, default=default, expand=expand,
                                          num_samples=num_samples, tmc=tmc)
    return poutine.infer_config(guide, config_fn=_config_fn(default, expand, num_samples, tmc))
