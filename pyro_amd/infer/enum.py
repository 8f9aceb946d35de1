"""Importance-trace construction shared by the ELBO estimators
(reference: pyro/infer/enum.py:45-85 get_importance_trace, :138-220 config_enumerate)."""
import functools
import numbers

from .util import is_validation_enabled

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
    from ..ops import lazy
    with lazy.watch_histograms():      # (examples/lda.py's word histogram: see ops/lazy.py)
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


def iter_discrete_traces(graph_type, fn, *args, **kwargs):
    """Every trace of ``fn`` over the joint support of its sites marked
    ``infer={"enumerate": "sequential"}``, depth first.  Each such site carries the size of its support in
    ``infer["_enum_total"]``.  (The reference builds this on ``poutine.queue`` with escape / extend callbacks,
    pyro/infer/enum.py:17-111; here one run = one assignment of support indices handed to
    ``SequentialEnumMessenger``, which also reports the alternatives of every newly met site.)"""
    from ..poutine.handlers import SequentialEnumMessenger
    pending = [{}]                              # assignments (site name -> support index) still to run
    while pending:
        assignment = pending.pop()
        found = []
        branch = SequentialEnumMessenger(assignment, found)
        yield poutine.trace(branch(fn), graph_type=graph_type).get_trace(*args, **kwargs)
        pending.extend(reversed(found))         # the run just made took index 0 of each new site


class _EnumerationConfig:
    """The ``infer`` entries ``config_enumerate`` adds to a site; entries the site already has win."""

    def __init__(self, default, expand, num_samples, tmc):
        self.default, self.expand, self.num_samples, self.tmc = default, expand, num_samples, tmc

    def __call__(self, site):
        if site["type"] != "sample" or site["is_observed"] or type(site["fn"]).__name__ == "_Subsample":
            return {}
        local = self.num_samples is not None              # local sampling applies to every latent site
        if not local and not getattr(site["fn"], "has_enumerate_support", False):
            return {}
        offered = {"enumerate": self.default, "expand": self.expand}
        if local:
            offered.update(num_samples=self.num_samples, tmc=self.tmc)
        return {key: site["infer"].get(key, value) for key, value in offered.items()}


def config_enumerate(guide=None, default="parallel", expand=False, num_samples=None, tmc="diagonal"):
    """Mark the sites of ``guide`` (or of a model) for enumeration: every site that can enumerate its
    support -- or, with ``num_samples=n``, every latent site for n local Monte-Carlo draws on an
    enumeration dim.  Usable as a decorator, with or without arguments (pyro/infer/enum.py:138-220)."""
    if default not in ("sequential", "parallel", "flat", None):
        raise ValueError("Invalid default value. Expected 'sequential', 'parallel', or None, but got "
                         "{}".format(repr(default)))
    if expand is not True and expand is not False:
        raise ValueError("Invalid expand value. Expected True or False, but got {}".format(repr(expand)))
    if num_samples is not None:
        if not isinstance(num_samples, numbers.Number) or num_samples <= 0:
            raise ValueError("Invalid num_samples, expected None or positive integer, but got "
                             "{}".format(repr(num_samples)))
        if default == "sequential":
            raise ValueError('Local sampling does not support "sequential" sampling; use "parallel" '
                             "sampling instead.")
        if tmc == "full" and num_samples > 1:
            expand = True
    configure = functools.partial(poutine.infer_config,
                                  config_fn=_EnumerationConfig(default, expand, num_samples, tmc))
    return configure if guide is None else configure(guide)
