"""Reparameterisers (reference: pyro/infer/reparam/{reparam,loc_scale,transform}.py) for
``poutine.reparam``: a sample site is replaced by an auxiliary site in better-conditioned coordinates
followed by a deterministic map back -- the usual cure for funnel-shaped posteriors under HMC / NUTS and
mean-field guides.  Only the two strategies the hot paths are used with are provided: non-centring of
location-scale families and sampling the base of a TransformedDistribution."""
from .loc_scale import LocScaleReparam  # noqa: F401
from .reparam import Reparam  # noqa: F401
from .transform import TransformReparam  # noqa: F401
