"""Trace utilities (reference: pyro/poutine/util.py:40-48)."""
from .handlers import _Subsample


def site_is_subsample(site):
    return site["type"] == "sample" and isinstance(site["fn"], _Subsample)


def prune_subsample_sites(trace):
    trace = trace.copy()
    for name, site in list(trace.nodes.items()):
        if site_is_subsample(site):
            trace.remove_node(name)
    return trace
