"""Small shared utilities (reference: pyro/util.py:107-146 warn_if_nan/inf, pyro/infer/util.py
torch_item / zero_grads)."""
import math
import numbers
import warnings

import torch


def torch_item(x):
    return x if isinstance(x, numbers.Number) else x.item()


def torch_isnan(x):
    if isinstance(x, numbers.Number):
        return x != x
    return torch.isnan(x)


def warn_if_nan(value, msg="", *, filename=None, lineno=None):
    if torch.is_tensor(value):
        if value.requires_grad:
            value = value.detach()
        isnan = bool(torch.isnan(value).any())
    else:
        isnan = value != value
    if isnan:
        warnings.warn("Encountered NaN{}".format(": " + msg if msg else "."), stacklevel=2)
    return value


def warn_if_inf(value, msg="", allow_posinf=False, allow_neginf=False, *, filename=None,
                lineno=None):
    if torch.is_tensor(value):
        v = value.detach()
        if not allow_posinf and bool((v == math.inf).any()):
            warnings.warn("Encountered +inf{}".format(": " + msg if msg else "."), stacklevel=2)
        if not allow_neginf and bool((v == -math.inf).any()):
            warnings.warn("Encountered -inf{}".format(": " + msg if msg else "."), stacklevel=2)
    else:
        if not allow_posinf and value == math.inf:
            warnings.warn("Encountered +inf{}".format(": " + msg if msg else "."), stacklevel=2)
        if not allow_neginf and value == -math.inf:
            warnings.warn("Encountered -inf{}".format(": " + msg if msg else "."), stacklevel=2)
    return value


def zero_grads(tensors):
    """Zero the .grad of each tensor in place (the reference re-allocates zeros_like every step,
    pyro/infer/util.py:85-91; in-place keeps the allocator out of the step)."""
    with torch.no_grad():
        for p in tensors:
            if p.grad is not None:
                if p.grad.grad_fn is not None:
                    p.grad = p.grad.detach()
                p.grad.zero_()     # works for gradients that are views of a flat buffer too


def scalar_like(prototype, fill_value):
    return torch.tensor(fill_value, dtype=prototype.dtype, device=prototype.device)
