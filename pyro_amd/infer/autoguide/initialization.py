"""Site initialisation strategies for autoguides and MCMC
(reference: pyro/infer/autoguide/initialization.py:35-256)."""
import functools

import torch
from torch.distributions import biject_to

from ...poutine.runtime import Messenger


def init_to_feasible(site=None):
    """Initialise to an arbitrary feasible point (0 in unconstrained space), ignoring the
    distribution's parameters."""
    if site is None:
        return init_to_feasible
    with torch.no_grad():
        value = site["fn"].sample()
        t = biject_to(site["fn"].support)
        return t(torch.zeros_like(t.inv(value)))


def init_to_sample(site=None):
    if site is None:
        return init_to_sample
    with torch.no_grad():
        return site["fn"].sample()


def init_to_median(site=None, num_samples=15, *, fallback=init_to_feasible):
    if site is None:
        return functools.partial(init_to_median, num_samples=num_samples, fallback=fallback)
    try:
        with torch.no_grad():
            samples = site["fn"].sample(sample_shape=(num_samples,))
            value = samples.median(dim=0)[0]
            if torch.isfinite(value).all():
                return value
    except (RuntimeError, NotImplementedError):
        pass
    return fallback(site)


def init_to_mean(site=None, *, fallback=init_to_median):
    if site is None:
        return functools.partial(init_to_mean, fallback=fallback)
    try:
        with torch.no_grad():
            value = site["fn"].mean
            if torch.isfinite(value).all():
                return value.detach().clone()
    except (NotImplementedError, AttributeError):
        pass
    return fallback(site)


def init_to_uniform(site=None, radius=2.0):
    """Uniform in (-radius, radius) in unconstrained space."""
    if site is None:
        return functools.partial(init_to_uniform, radius=radius)
    with torch.no_grad():
        value = site["fn"].sample()
        t = biject_to(site["fn"].support)
        u = t.inv(value)
        return t(torch.empty_like(u).uniform_(-radius, radius))


def init_to_value(site=None, values=None, *, fallback=init_to_uniform):
    if site is None:
        return functools.partial(init_to_value, values=values or {}, fallback=fallback)
    if values and site["name"] in values:
        return values[site["name"]]
    if fallback is None:
        raise ValueError("No init strategy specified for site {}".format(site["name"]))
    return fallback(site)


class _InitToGenerated:
    def __init__(self, generate):
        self.generate = generate
        self._init = None
        self._seen = set()

    def __call__(self, site):
        # a site name coming round again means a new execution of the model: new strategy
        if self._init is None or site["name"] in self._seen:
            self._init = self.generate()
            self._seen = set()
        self._seen.add(site["name"])
        return self._init(site)


def init_to_generated(site=None, generate=lambda: init_to_uniform):
    """Initialise with the strategy ``generate()`` returns, asked for once per execution of the model
    -- e.g. an ``init_to_value`` over freshly drawn values (initialization.py:183-217)."""
    init = _InitToGenerated(generate)
    return init if site is None else init(site)


class InitMessenger(Messenger):
    """Set the value of each latent site with an init strategy instead of sampling."""

    def __init__(self, init_fn):
        super().__init__()
        self.init_fn = init_fn

    def _pyro_sample(self, msg):
        if msg["done"] or msg["is_observed"] or type(msg["fn"]).__name__ == "_Subsample":
            return
        if getattr(msg["fn"], "has_enumerate_support", False):
            return      # discrete: no unconstrained space to initialise in; drawn (or enumerated)
        with torch.no_grad():
            value = self.init_fn(msg)
        if value is not None:
            # expand to the (plate-broadcast) shape of the site
            shape = msg["fn"].shape() if hasattr(msg["fn"], "shape") else value.shape
            if value.shape != shape:
                try:
                    value = value.expand(shape)
                except RuntimeError:
                    pass
            msg["value"] = value
            msg["done"] = True
