"""Trace utilities (reference: pyro/poutine/util.py)."""
from . import settings
from .handlers import _Subsample


def enable_validation(is_validate):
    settings.enable_validation(is_validate)


def is_validation_enabled():
    return settings.validation_enabled()


def site_is_subsample(site):
    return site["type"] == "sample" and isinstance(site["fn"], _Subsample)


def site_is_factor(site):
    return site["type"] == "sample" and type(site["fn"]).__name__ == "Unit"


def prune_subsample_sites(trace):
    trace = trace.copy()
    for name, site in list(trace.nodes.items()):
        if site_is_subsample(site):
            trace.remove_node(name)
    return trace


# ---- sequential search over traces (poutine.queue; util.py:52-146) ----------------------------------
def _extended(trace, msg, value):
    site = msg.copy()
    site["value"] = value
    longer = trace.copy()
    longer.add_node(msg["name"], **site)
    return longer


def enum_extend(trace, msg, num_samples=None):
    """One copy of ``trace`` per value in the support of the site ``msg``, each with the site added
    (at most ``num_samples + 1`` of them when that is given and non-negative)."""
    limit = -1 if num_samples is None else num_samples
    out = []
    for i, value in enumerate(msg["fn"].enumerate_support(*msg["args"], **msg["kwargs"])):
        if 0 <= limit < i:
            break
        out.append(_extended(trace, msg, value))
    return out


def mc_extend(trace, msg, num_samples=None):
    """``num_samples`` (default 1) copies of ``trace``, each with a fresh draw of the site added."""
    return [_extended(trace, msg, msg["fn"](*msg["args"], **msg["kwargs"]))
            for _ in range(1 if num_samples is None else num_samples)]


def all_escape(trace, msg):
    """A latent sample site that ``trace`` does not hold yet."""
    return (msg["type"] == "sample" and not msg["is_observed"] and msg["name"] is not None
            and msg["name"] not in trace)


def discrete_escape(trace, msg):
    """As :func:`all_escape`, for sites whose distribution can enumerate its support."""
    return all_escape(trace, msg) and bool(getattr(msg["fn"], "has_enumerate_support", False))
