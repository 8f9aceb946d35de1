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


def is_identically_zero(x):
    if isinstance(x, numbers.Number):
        return x == 0
    return False


def is_identically_one(x):
    if isinstance(x, numbers.Number):
        return x == 1
    return False


def sum_rightmost(value, dim):
    """Sum out ``dim`` rightmost dims (negative dim: keep ``-dim`` leftmost dims)
    (reference: pyro/distributions/util.py:253-276)."""
    if isinstance(value, numbers.Number):
        return value
    if dim < 0:
        dim += value.dim()
    if dim == 0:
        return value
    if dim >= value.dim():
        return value.sum()
    return value.reshape(value.shape[:-dim] + (-1,)).sum(-1)


def sum_leftmost(value, dim):
    if isinstance(value, numbers.Number):
        return value
    if dim < 0:
        dim += value.dim()
    if dim == 0:
        return value
    if dim >= value.dim():
        return value.sum()
    return value.reshape(-1, *value.shape[dim:]).sum(0)


def scale_and_mask(tensor, scale=1.0, mask=None):
    """tensor*scale where mask else 0 (reference: pyro/distributions/util.py:311-328)."""
    if is_identically_zero(tensor) or (mask is None and is_identically_one(scale)):
        return tensor
    if mask is None or mask is True:
        return tensor * scale
    if mask is False:
        return torch.zeros_like(tensor)
    return torch.where(mask, tensor * scale, tensor.new_zeros(()))


def broadcast_shape(*shapes, strict=False):
    reversed_shape = []
    for shape in shapes:
        for i, size in enumerate(reversed(shape)):
            if i >= len(reversed_shape):
                reversed_shape.append(size)
            elif reversed_shape[i] == 1 and not strict:
                reversed_shape[i] = size
            elif reversed_shape[i] != size and (size != 1 or strict):
                raise ValueError("shape mismatch: objects cannot be broadcast to a single shape: "
                                 + " vs ".join(map(str, shapes)))
    return tuple(reversed(reversed_shape))
