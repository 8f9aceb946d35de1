"""KL divergences the reference registers on top of torch's (pyro/distributions/kl.py:19-56,
pyro/distributions/torch_distribution.py:529-547) -- ``TraceMeanField_ELBO`` calls
``kl_divergence(guide_fn, model_fn)`` on whatever the two traces hold, so the set of registered
pairs decides which sites take the analytic route:

* Delta || anything: ``-q.log_prob(p.v)`` (autoguides emit Delta sites; note that the Delta's own
  ``log_density`` does not enter, exactly as in the reference);
* Independent || Independent with DIFFERENT numbers of reinterpreted dims (torch only knows equal
  ones): the shared dims are summed, the rest stays wrapped;
* Independent(Delta | Normal, 1) || MultivariateNormal in closed form;
* MaskedDistribution || MaskedDistribution: the KL of the bases under the conjunction of the masks.
"""
import math

from torch.distributions import Independent as _TorchIndependent
from torch.distributions import MultivariateNormal, Normal, kl_divergence, register_kl
from torch.distributions.distribution import Distribution as _TorchDistribution

from .base import Delta, MaskedDistribution
from .util import scale_and_mask, sum_rightmost


@register_kl(Delta, _TorchDistribution)
def _kl_delta(p, q):
    return -q.log_prob(p.v)


@register_kl(_TorchIndependent, _TorchIndependent)
def _kl_independent_independent(p, q):
    shared = min(p.reinterpreted_batch_ndims, q.reinterpreted_batch_ndims)
    p_rest = p.reinterpreted_batch_ndims - shared
    q_rest = q.reinterpreted_batch_ndims - shared
    p = type(p)(p.base_dist, p_rest) if p_rest else p.base_dist
    q = type(q)(q.base_dist, q_rest) if q_rest else q.base_dist
    kl = kl_divergence(p, q)
    return sum_rightmost(kl, shared) if shared else kl


@register_kl(_TorchIndependent, MultivariateNormal)
def _kl_independent_mvn(p, q):
    if isinstance(p.base_dist, Delta) and p.reinterpreted_batch_ndims == 1:
        return -q.log_prob(p.base_dist.v)
    if isinstance(p.base_dist, Normal) and p.reinterpreted_batch_ndims == 1:
        dim = q.event_shape[0]
        p_cov = p.base_dist.scale ** 2
        q_precision = q.precision_matrix.diagonal(dim1=-2, dim2=-1)
        return (0.5 * (p_cov * q_precision).sum(-1) - 0.5 * dim * (1 + math.log(2 * math.pi))
                - q.log_prob(p.base_dist.loc) - p.base_dist.scale.log().sum(-1))
    raise NotImplementedError


@register_kl(MaskedDistribution, MaskedDistribution)
def _kl_masked_masked(p, q):
    if p._mask is False or q._mask is False:
        mask = False
    elif p._mask is True:
        mask = q._mask
    elif q._mask is True:
        mask = p._mask
    elif p._mask is q._mask:
        mask = p._mask
    else:
        mask = p._mask & q._mask
    if mask is False:
        return 0.0      # a float: the device cannot be known
    kl = kl_divergence(p.base_dist, q.base_dist)
    return kl if mask is True else scale_and_mask(kl, mask=mask)
