"""Batch-aligned advanced indexing for enumerated index tensors
(reference: pyro/ops/indexing.py:9-180 vindex / Vindex, the helper examples/hmm.py uses to select
``probs_y[x, y]`` when ``x`` and ``y`` carry enumeration dims to the left of the plates).

``Vindex(t)[..., i, :, j]`` treats the dims covered by the ellipsis as batch dims of ``t`` that are
right-aligned with the batch dims of the integer tensors ``i`` and ``j`` (so a size-1 or missing
dim broadcasts) and keeps every full slice as a trailing dim of the result:

    Vindex(t)[..., i, :, j][b..., e] == t[b..., i[b...], e, j[b...]]

Implemented as ONE gather through torch's advanced indexing: every full slice (and every ellipsis
dim) becomes an ``arange`` placed on its own output dim, counted from the right.
"""
import torch


def vindex(tensor, args):
    if not isinstance(args, tuple):
        return tensor[args]
    if not args:
        return tensor
    if args[0] is Ellipsis:                      # leading ellipsis only: the batch dims of tensor
        args = tuple(args[1:])
        if not args:
            return tensor
        n_batch = tensor.dim() - len(args)
        args = (slice(None),) * n_batch + args
    else:                                        # un-batched tensor: missing dims are full slices
        n_batch = 0
        args = tuple(args) + (slice(None),) * (tensor.dim() - len(args))
    if any(a is Ellipsis for a in args):
        raise NotImplementedError("Vindex supports an ellipsis only in first position")
    if not any(isinstance(a, torch.Tensor) and a.dim() for a in args):
        return tensor[args]                      # nothing to align: plain indexing
    for a in args:
        if isinstance(a, slice) and a != slice(None):
            raise NotImplementedError("Vindex supports only full slices, got {}".format(a))
    # Output layout: broadcast(batch dims of tensor, index batch dims) + one trailing dim per full
    # slice of the event part.  Every slice becomes an arange on its own output dim counted from
    # the right (event slices first, then the tensor's batch dims); index tensors get one trailing
    # unit dim per event slice so that their batch dims line up with the tensor's batch dims.
    event_slices = [p for p in range(n_batch, len(args)) if isinstance(args[p], slice)]
    k = len(event_slices)

    def arange(pos, right):
        return torch.arange(tensor.size(pos), device=tensor.device).reshape((-1,) + (1,) * right)

    out = list(args)
    for pos, a in enumerate(args):
        if isinstance(a, torch.Tensor) and a.dim():
            out[pos] = a.reshape(tuple(a.shape) + (1,) * k)
    for r, pos in enumerate(reversed(event_slices)):
        out[pos] = arange(pos, r)
    for r, pos in enumerate(range(n_batch - 1, -1, -1)):
        out[pos] = arange(pos, k + r)
    return tensor[tuple(out)]


class Vindex:
    """``Vindex(x)[..., i, j, :]`` == ``vindex(x, (Ellipsis, i, j, slice(None)))``."""

    def __init__(self, tensor):
        self._tensor = tensor

    def __getitem__(self, args):
        return vindex(self._tensor, args)


def _flatten_index(args, out):
    # nested index tuples are spliced in; two Ellipsis in a row count as one
    for arg in args:
        if isinstance(arg, tuple):
            _flatten_index(arg, out)
        elif arg is Ellipsis and out and out[-1] is Ellipsis:
            continue
        else:
            out.append(arg)


def index(tensor, args):
    """``tensor[args]`` where ``args`` may hold nested tuples, e.g. ``(Ellipsis, t)`` with
    ``t = (Ellipsis, None)`` meaning ``tensor.unsqueeze(-1)`` (reference: pyro/ops/indexing.py:25-59)."""
    if not isinstance(args, tuple):
        return tensor[args]
    if not args:
        return tensor
    flat = []
    _flatten_index(args, flat)
    return tensor[tuple(flat)]


class Index:
    """``Index(x)[..., i, j, :]`` = ``index(x, (Ellipsis, i, j, slice(None)))``."""

    def __init__(self, tensor):
        self._tensor = tensor

    def __getitem__(self, args):
        return index(self._tensor, args)
