"""Sampling the base of a TransformedDistribution (reference: pyro/infer/reparam/transform.py)."""
import torch

from ... import distributions as dist
from ...primitives import sample
from .reparam import Reparam


class TransformReparam(Reparam):
    """A latent ``TransformedDistribution`` site: sample ``<name>_base`` from the base distribution and push
    it through the transforms."""

    def apply(self, msg):
        name, fn, value, is_observed = msg["name"], msg["fn"], msg["value"], msg["is_observed"]
        fn, event_dim = self._unwrap(fn)
        assert isinstance(fn, torch.distributions.TransformedDistribution)
        value_base = value
        if value is not None:
            for t in reversed(fn.transforms):
                value_base = t.inv(value_base)
        base_event_dim = event_dim
        for t in reversed(fn.transforms):
            base_event_dim += t.domain.event_dim - t.codomain.event_dim
        value_base = sample("{}_base".format(name), self._wrap(fn.base_dist, base_event_dim),
                            obs=value_base, infer={"is_observed": is_observed})
        if value is None:
            value = value_base
            for t in fn.transforms:
                value = t(value)
        return {"fn": dist.Delta(value, event_dim=event_dim).mask(False), "value": value,
                "is_observed": True}
