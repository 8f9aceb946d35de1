"""Non-centring of location-scale families (reference: pyro/infer/reparam/loc_scale.py)."""
import torch
from torch.distributions import constraints

from ... import distributions as dist
from ...distributions.util import is_identically_one
from ...primitives import param, sample
from ..util import is_validation_enabled
from .reparam import Reparam


class LocScaleReparam(Reparam):
    """Partial non-centring of a location-scale site: with ``c = centered`` in [0, 1] the auxiliary site
    ``<name>_decentered ~ Family(c loc, scale^c)`` is sampled and
    ``value = loc + scale^(1-c) (decentered - c loc)``.  ``centered=None`` learns c per element
    (``<name>_centered``); ``shape_params`` names the other constructor arguments to carry over
    (default: all of ``arg_constraints`` except loc and scale)."""

    def __init__(self, centered=None, shape_params=None):
        assert centered is None or isinstance(centered, (float, torch.Tensor))
        if shape_params is not None:
            assert isinstance(shape_params, (tuple, list)) and all(isinstance(n, str) for n in shape_params)
        if is_validation_enabled() and centered is not None:
            c = torch.as_tensor(centered)
            assert bool((0 <= c).all()) and bool((c <= 1).all())
        self.centered, self.shape_params = centered, shape_params

    def apply(self, msg):
        name, fn, value, is_observed = msg["name"], msg["fn"], msg["value"], msg["is_observed"]
        centered = self.centered
        if is_identically_one(centered):
            return msg
        event_shape = fn.event_shape
        fn, event_dim = self._unwrap(fn)
        if self.shape_params is None:
            self.shape_params = tuple(k for k in fn.arg_constraints if k not in ("loc", "scale"))
        params = {key: getattr(fn, key) for key in self.shape_params}
        if centered is None:
            centered = param("{}_centered".format(name), lambda: fn.loc.new_full(event_shape, 0.5),
                             constraint=constraints.unit_interval)
        params["loc"] = fn.loc * centered
        params["scale"] = fn.scale ** centered
        decentered_fn = type(fn)(**params)
        decentered_value = None
        if value is not None:
            decentered_value = (value - fn.loc) * fn.scale.pow(centered - 1) + centered * fn.loc
        decentered_value = sample("{}_decentered".format(name), self._wrap(decentered_fn, event_dim),
                                  obs=decentered_value, infer={"is_observed": is_observed})
        if value is None:
            value = fn.loc + fn.scale.pow(1 - centered) * (decentered_value - centered * fn.loc)
        return {"fn": dist.Delta(value, event_dim=event_dim).mask(False), "value": value,
                "is_observed": True}
