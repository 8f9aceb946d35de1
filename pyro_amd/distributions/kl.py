"""KL pairs registered on top of torch's, the ones the reference adds (pyro/distributions/kl.py,
pyro/distributions/torch_distribution.py): ``TraceMeanField_ELBO`` asks ``kl_divergence(guide_fn, model_fn)``
for whatever the two traces hold, so this table decides which sites take the analytic route.

* ``Delta || anything`` = ``-q.log_prob(point)``: autoguides emit Delta sites; the Delta's own ``log_density``
  does not enter, as in the reference;
* ``Independent || Independent`` with different numbers of reinterpreted dims (torch only knows equal ones):
  the common dims are summed out of the KL of what remains wrapped;
* ``Independent(Delta or Normal, 1) || MultivariateNormal`` in closed form;
* ``Masked || Masked``: the KL of the bases where both masks hold.
"""
import math

import torch.distributions as td
from torch.distributions import kl_divergence, register_kl

from .base import Delta, MaskedDistribution
from .util import scale_and_mask, sum_rightmost

_HALF_LOG_2PI_E = 0.5 * (1.0 + math.log(2.0 * math.pi))       # entropy of a unit normal coordinate


def _peel(d, keep):
    """An Independent with only ``keep`` of its reinterpreted dims left (its base when none are left)."""
    return type(d)(d.base_dist, keep) if keep else d.base_dist


@register_kl(Delta, td.Distribution)
def _point_mass_against(p, q):
    return -q.log_prob(p.v)


@register_kl(td.Independent, td.Independent)
def _independent_pair(p, q):
    common = min(p.reinterpreted_batch_ndims, q.reinterpreted_batch_ndims)
    inner = kl_divergence(_peel(p, p.reinterpreted_batch_ndims - common),
                          _peel(q, q.reinterpreted_batch_ndims - common))
    return sum_rightmost(inner, common) if common else inner


@register_kl(td.Independent, td.MultivariateNormal)
def _diagonal_against_mvn(p, q):
    base = p.base_dist
    if p.reinterpreted_batch_ndims != 1 or not isinstance(base, (Delta, td.Normal)):
        raise NotImplementedError
    if isinstance(base, Delta):
        return -q.log_prob(base.v)
    # E_p[-log q] - H[p] for p = N(loc, diag(scale^2)): only the diagonal of q's precision meets p's covariance
    n = q.event_shape[0]
    trace_term = 0.5 * (base.scale.pow(2) * q.precision_matrix.diagonal(dim1=-2, dim2=-1)).sum(-1)
    entropy = base.scale.log().sum(-1) + n * _HALF_LOG_2PI_E
    return trace_term - q.log_prob(base.loc) - entropy


def _both(m1, m2):
    # conjunction of two masks, each a bool or a bool tensor
    if m1 is False or m2 is False:
        return False
    if m1 is True:
        return m2
    if m2 is True or m1 is m2:
        return m1
    return m1 & m2


@register_kl(MaskedDistribution, MaskedDistribution)
def _masked_pair(p, q):
    mask = _both(p._mask, q._mask)
    if mask is False:
        return 0.0                    # a float: there is no tensor to take a device from
    kl = kl_divergence(p.base_dist, q.base_dist)
    return kl if mask is True else scale_and_mask(kl, mask=mask)
