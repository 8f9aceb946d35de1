"""Where autoguides and HMC/NUTS start: one starting value per latent site (the strategies of
pyro.infer.autoguide.initialization, same names and arguments).

Every strategy obeys one calling protocol, implemented once by ``@_strategy``: called with a site it returns
that site's starting value; called without one -- ``init_to_median(num_samples=50)`` -- it returns itself with
the options bound.  Values are produced under ``torch.no_grad()``.
"""
import functools

import torch
from torch.distributions import biject_to

from ...poutine.runtime import Messenger


def _strategy(fn):
    @functools.wraps(fn)
    def entry(site=None, *args, **options):
        if site is None:
            return functools.partial(entry, **options) if options or args else entry
        with torch.no_grad():
            return fn(site, *args, **options)

    return entry


def _in_unconstrained_space(site, make):
    """``make(u)`` with ``u`` shaped like the site's value in unconstrained coordinates, mapped back."""
    to_support = biject_to(site["fn"].support)
    template = to_support.inv(site["fn"].sample())
    return to_support(make(template))


def _finite(value):
    return value is not None and bool(torch.isfinite(value).all())


# ---- the strategies ---------------------------------------------------------------------------------------
@_strategy
def init_to_feasible(site):
    """The image of 0 in unconstrained space: feasible whatever the distribution's parameters are."""
    return _in_unconstrained_space(site, torch.zeros_like)


@_strategy
def init_to_sample(site):
    """A draw from the prior."""
    return site["fn"].sample()


@_strategy
def init_to_median(site, num_samples=15, *, fallback=init_to_feasible):
    """The element-wise median of ``num_samples`` prior draws; ``fallback`` where that is not finite."""
    value = None
    try:
        value = site["fn"].sample(sample_shape=(num_samples,)).median(dim=0).values
    except (RuntimeError, NotImplementedError):
        pass
    return value if _finite(value) else fallback(site)


@_strategy
def init_to_mean(site, *, fallback=init_to_median):
    """The prior mean; ``fallback`` for families without one (or with an infinite one)."""
    value = None
    try:
        value = site["fn"].mean
    except (NotImplementedError, AttributeError):
        pass
    return value.detach().clone() if _finite(value) else fallback(site)


@_strategy
def init_to_uniform(site, radius=2.0):
    """Uniform in (-radius, radius) in unconstrained space (Stan's default)."""
    return _in_unconstrained_space(site, lambda u: torch.empty_like(u).uniform_(-radius, radius))


@_strategy
def init_to_value(site, values=None, *, fallback=init_to_uniform):
    """``values[name]`` for the sites named there, ``fallback`` (None: ValueError) for the rest."""
    name = site["name"]
    if values is not None and name in values:
        return values[name]
    if fallback is None:
        raise ValueError("No init strategy specified for site {!r}".format(name))
    return fallback(site)


class _PerExecution:
    """A strategy that is re-made once per execution of the model: meeting a site name for the second time
    is how a new execution is recognised."""

    def __init__(self, generate):
        self.generate, self._current, self._met = generate, None, set()

    def __call__(self, site):
        name = site["name"]
        if self._current is None or name in self._met:
            self._current, self._met = self.generate(), set()
        self._met.add(name)
        return self._current(site)


def init_to_generated(site=None, generate=lambda: init_to_uniform):
    """Start from the strategy ``generate()`` returns, asked for afresh for every execution of the model --
    e.g. an ``init_to_value`` over newly drawn values."""
    strategy = _PerExecution(generate)
    return strategy if site is None else strategy(site)


# ---- applying a strategy --------------------------------------------------------------------------------------
class InitMessenger(Messenger):
    """Give every continuous latent site the value its strategy says instead of a draw."""

    def __init__(self, init_fn):
        super().__init__()
        self.init_fn = init_fn

    def _pyro_sample(self, msg):
        fn = msg["fn"]
        if msg["done"] or msg["is_observed"] or type(fn).__name__ == "_Subsample":
            return
        if getattr(fn, "has_enumerate_support", False):
            return          # discrete: there is no unconstrained space to start in; drawn or enumerated
        with torch.no_grad():
            value = self.init_fn(msg)
        if value is None:
            return
        wanted = fn.shape() if hasattr(fn, "shape") else value.shape          # plates may have widened it
        if value.shape != wanted:
            try:
                value = value.expand(wanted)
            except RuntimeError:
                pass
        msg["value"], msg["done"] = value, True
