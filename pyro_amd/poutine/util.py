"""Helpers around traces and sites (the names of pyro.poutine.util)."""
from . import settings
from .handlers import _Subsample


# ---- validation switch of the handler layer ---------------------------------------------------------------
def enable_validation(is_validate):
    settings.enable_validation(is_validate)


def is_validation_enabled():
    return settings.validation_enabled()


# ---- kinds of sample sites ------------------------------------------------------------------------------------
def _is_sample(site):
    return site["type"] == "sample"


def site_is_subsample(site):
    """The bookkeeping site a subsampling ``pyro.plate`` records."""
    return _is_sample(site) and isinstance(site["fn"], _Subsample)


def site_is_factor(site):
    """A ``pyro.factor`` statement."""
    return _is_sample(site) and type(site["fn"]).__name__ == "Unit"


def prune_subsample_sites(trace):
    """A copy of ``trace`` without the plates' bookkeeping sites."""
    pruned = trace.copy()
    for name in [n for n, site in pruned.nodes.items() if site_is_subsample(site)]:
        pruned.remove_node(name)
    return pruned


# ---- building blocks of poutine.queue: when to leave a run, and how to continue it -----------------------------
def all_escape(trace, msg):
    """Leave at a latent sample site that ``trace`` does not hold yet."""
    return _is_sample(msg) and not msg["is_observed"] and msg["name"] is not None and msg["name"] not in trace


def discrete_escape(trace, msg):
    """As :func:`all_escape`, for sites whose distribution can enumerate its support."""
    return all_escape(trace, msg) and bool(getattr(msg["fn"], "has_enumerate_support", False))


def _continuations(trace, msg, values):
    """One copy of ``trace`` per value, each with the site ``msg`` added at that value."""
    out = []
    for value in values:
        site = dict(msg, value=value)
        longer = trace.copy()
        longer.add_node(msg["name"], **site)
        out.append(longer)
    return out


def enum_extend(trace, msg, num_samples=None):
    """Continue with every value in the site's support (the first ``num_samples + 1`` of them when
    ``num_samples`` is given and non-negative)."""
    support = msg["fn"].enumerate_support(*msg["args"], **msg["kwargs"])
    if num_samples is not None and num_samples >= 0:
        support = support[:num_samples + 1]
    return _continuations(trace, msg, support)


def mc_extend(trace, msg, num_samples=None):
    """Continue with ``num_samples`` (default 1) fresh draws of the site."""
    count = 1 if num_samples is None else num_samples
    return _continuations(trace, msg, (msg["fn"](*msg["args"], **msg["kwargs"]) for _ in range(count)))
