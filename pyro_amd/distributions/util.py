"""Shape / scaling helpers of the distribution layer (reference: pyro/distributions/util.py)."""
import numbers

import torch

_VALIDATION_ENABLED = __debug__


def enable_validation(is_validate=True):
    global _VALIDATION_ENABLED
    _VALIDATION_ENABLED = bool(is_validate)
    torch.distributions.Distribution.set_default_validate_args(bool(is_validate))


def is_validation_enabled():
    return _VALIDATION_ENABLED


def _is_the_number(x, k):
    return isinstance(x, numbers.Number) and x == k


def is_identically_zero(x):
    return _is_the_number(x, 0)


def is_identically_one(x):
    return _is_the_number(x, 1)


def _sum_block(value, count, rightmost):
    """Sum ``count`` dims at one end of ``value`` (count < 0: all but ``-count`` dims at the OTHER end);
    plain numbers pass through (a site log-density that is a Python constant)."""
    if isinstance(value, numbers.Number):
        return value
    nd = value.dim()
    count = count + nd if count < 0 else count
    if count <= 0:
        return value
    if count >= nd:
        return value.sum()
    dims = tuple(range(nd - count, nd)) if rightmost else tuple(range(count))
    return value.sum(dims)


def sum_rightmost(value, dim):
    """Sum out ``dim`` rightmost dims; ``dim < 0`` keeps ``-dim`` leftmost dims (the helper of the same
    name the reference's distributions use, pyro/distributions/util.py:253-276)."""
    return _sum_block(value, dim, True)


def sum_leftmost(value, dim):
    """Sum out ``dim`` leftmost dims; ``dim < 0`` keeps ``-dim`` rightmost dims."""
    return _sum_block(value, dim, False)


def scale_and_mask(tensor, scale=1.0, mask=None):
    """The UN-fused form of SURVEY 8a row a5 (pyro/distributions/util.py:311-328: a site's log-density
    times its plate scale, zero where the mask is off) for log-densities that come from torch's own
    classes; the fused families take scale and mask as KERNEL ARGUMENTS (fused_log_prob_sum) and never
    get here."""
    if mask is False:
        return tensor if is_identically_zero(tensor) else torch.zeros_like(tensor)
    unscaled = is_identically_one(scale)
    if is_identically_zero(tensor) or (unscaled and (mask is None or mask is True)):
        return tensor
    scaled = tensor if unscaled else tensor * scale
    if mask is None or mask is True:
        return scaled
    return torch.where(mask, scaled, scaled.new_zeros(()))


def broadcast_shape(*shapes, strict=False):
    """numpy-style broadcast of shape tuples; ``strict``: sizes must agree exactly wherever two shapes
    both have the dim (no 1 -> n)."""
    out = [None] * max((len(s) for s in shapes), default=0)     # None: no shape has reached the dim yet
    for shape in shapes:
        for k in range(1, len(shape) + 1):
            have, size = out[-k], shape[-k]
            if have is None or (have == 1 and not strict):
                out[-k] = size
            elif have != size and (strict or size != 1):
                raise ValueError("shape mismatch: objects cannot be broadcast to a single shape: "
                                 + " vs ".join(str(tuple(s)) if not isinstance(s, torch.Size) else str(s)
                                               for s in shapes))
    return tuple(out)
