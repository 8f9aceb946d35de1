"""Non-centring of location-scale families (role of pyro/infer/reparam/loc_scale.py; Gorinova, Moore,
Hoffman 2019): the funnel geometry of ``x ~ F(loc, scale)`` with random loc / scale is removed by sampling
in coordinates where the site's law does not depend on them."""
import torch
from torch.distributions import constraints

from ...distributions.util import is_identically_one
from ...primitives import param
from ..util import is_validation_enabled
from .reparam import Reparam


def _in_unit_interval(c):
    c = torch.as_tensor(c)
    return bool(((c >= 0) & (c <= 1)).all())


class LocScaleReparam(Reparam):
    """With centring ``c`` in [0, 1] (per site or per element) the auxiliary site is
    ``<name>_decentered ~ F(c * loc, scale ** c)`` and ``x = loc + scale ** (1 - c) * (aux - c * loc)``:
    c = 1 changes nothing, c = 0 samples the standardised variable.  ``centered=None`` makes c a learnable
    parameter ``<name>_centered`` (initially 0.5).  ``shape_params``: names of the family's other
    constructor arguments to carry over (default: everything in ``arg_constraints`` but loc and scale)."""

    def __init__(self, centered=None, shape_params=None):
        if centered is not None:
            assert isinstance(centered, (float, torch.Tensor)), centered
            if is_validation_enabled():
                assert _in_unit_interval(centered), "centered must lie in [0, 1]"
        if shape_params is not None:
            assert isinstance(shape_params, (tuple, list))
            assert all(isinstance(name, str) for name in shape_params)
        self.centered = centered
        self.shape_params = shape_params

    def _centering(self, name, family, event_shape):
        if self.centered is not None:
            return self.centered
        return param(name + "_centered", lambda: family.loc.new_full(event_shape, 0.5),
                     constraint=constraints.unit_interval)

    def apply(self, msg):
        if is_identically_one(self.centered):
            return msg
        family, event_dim = self._unwrap(msg["fn"])
        loc, scale = family.loc, family.scale
        c = self._centering(msg["name"], family, msg["fn"].event_shape)
        if self.shape_params is None:
            self.shape_params = tuple(k for k in family.arg_constraints if k != "loc" and k != "scale")
        kept = {k: getattr(family, k) for k in self.shape_params}
        aux_fn = type(family)(loc=c * loc, scale=scale ** c, **kept)
        spread = scale ** (1 - c)                          # what is left of the scale outside the aux site

        def to_aux(x):
            return (x - loc) / spread + c * loc

        def from_aux(a):
            return loc + spread * (a - c * loc)

        return self._through_auxiliary(msg, msg["name"] + "_decentered", self._wrap(aux_fn, event_dim),
                                       event_dim, to_aux, from_aux)
