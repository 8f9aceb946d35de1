"""Constant tensors of a captured SVI step.

A model such as SURVEY 8(d)'s ``dist.Normal(X.new_zeros(D), 1.)`` asks for fresh constant tensors
on every execution.  Inside a captured hipGraph each of them is a fill kernel -- a graph node that
costs ~5 us of dispatch to write the same zeros again on every replay (two of the eight nodes of the
round-2 step).  ``ConstantRecorder`` watches one eager step and notes the factory calls with a
constant fill; ``ConstantReplayer`` hands pre-filled tensors (allocated and filled OUTSIDE the
capture) to the same calls while the step is being captured, so the graph holds no fill nodes for
them.  A constant handed out this way must never be written: any in-place operator that targets one
aborts the capture with ``HoistedConstantWritten`` and SVI captures again without hoisting.
"""
import torch
from torch.utils._python_dispatch import TorchDispatchMode

_MAX_ELEMS = 1 << 28        # up to 1 GiB of f32 per constant (config 4: a 410 MB zeros histogram that is never read)


class HoistedConstantWritten(RuntimeError):
    pass


def _factories():
    a = torch.ops.aten
    out = {}
    for name, fill in (("zeros", 0.0), ("ones", 1.0), ("new_zeros", 0.0), ("new_ones", 1.0),
                       ("zeros_like", 0.0), ("ones_like", 1.0), ("full", None), ("new_full", None),
                       ("full_like", None)):
        pkt = getattr(a, name, None)
        if pkt is None:
            continue
        ov = getattr(pkt, "default", None)
        if ov is not None:
            out[ov] = (name, fill)
    return out


_FACTORIES = None


def _key_of(func, args, kwargs):
    """Hashable description of a factory call (shapes, fill value, dtype, device of the result are
    checked on the tensor itself), or None when an argument cannot be described."""
    def norm(x):
        if isinstance(x, torch.Tensor):
            return ("T", tuple(x.shape), x.dtype, x.device)
        if isinstance(x, (list, tuple)):
            return tuple(norm(v) for v in x)
        if isinstance(x, (int, float, bool, str, type(None), torch.dtype, torch.device, torch.layout,
                          torch.memory_format, torch.Size)):
            return x
        raise TypeError
    try:
        return (func, norm(args), tuple(sorted((k, norm(v)) for k, v in (kwargs or {}).items())))
    except TypeError:
        return None


def _fill_of(name, fill, args, kwargs):
    """The fill value of a factory call (None: not a number we can re-create)."""
    if fill is not None:
        return fill
    v = kwargs.get("fill_value") if kwargs else None
    if v is None:
        pos = 2 if name == "new_full" else 1
        v = args[pos] if len(args) > pos else None
    return v if isinstance(v, (int, float, bool)) else None


class ConstantRecorder(TorchDispatchMode):
    """Notes every constant-filled device tensor created while active: ``calls`` = list of
    (key, (shape, strides, dtype, device, fill value)) -- descriptions only: holding the tensors of the
    eager step themselves would keep up to 1 GiB each alive per argument signature."""

    def __init__(self):
        super().__init__()
        global _FACTORIES
        if _FACTORIES is None:
            _FACTORIES = _factories()
        self.calls = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if func in _FACTORIES and isinstance(out, torch.Tensor) and out.is_cuda \
                and out.numel() <= _MAX_ELEMS and not out.requires_grad:
            key = _key_of(func, args, kwargs)
            name, fill = _FACTORIES[func]
            value = _fill_of(name, fill, args, kwargs)
            if key is not None and value is not None:
                self.calls.append((key, (tuple(out.shape), tuple(out.stride()), out.dtype, out.device,
                                         value)))
        return out


class ConstantReplayer(TorchDispatchMode):
    """Answers the recorded factory calls with pre-filled tensors (made by ``prepare`` before the
    capture) and refuses writes into them."""

    def __init__(self, recorder):
        super().__init__()
        self._queues = {}
        self._storages = set()
        self.tensors = []         # keep-alive: a captured graph reads them on every replay
        self.served = 0
        with torch.no_grad():
            for key, (shape, stride, dtype, device, value) in recorder.calls:
                # re-created from the recorded fill value (not cloned from the eager step's tensor: a
                # write the dispatcher cannot see -- a raw-pointer kernel, a .data alias -- would
                # otherwise be frozen into the "constant")
                c = torch.empty_strided(shape, stride, dtype=dtype, device=device).fill_(value)
                self._queues.setdefault(key, []).append(c)
                self._storages.add(c.untyped_storage().data_ptr())
                self.tensors.append(c)

    def _written(self, func, args, kwargs):
        schema = func._schema
        if not schema.is_mutable:
            return
        kwargs = kwargs or {}
        for i, arg in enumerate(schema.arguments):
            info = arg.alias_info
            if info is None or not info.is_write:
                continue
            v = args[i] if i < len(args) else kwargs.get(arg.name)
            for t in (v if isinstance(v, (list, tuple)) else (v,)):
                if isinstance(t, torch.Tensor) and t.is_cuda \
                        and t.untyped_storage().data_ptr() in self._storages:
                    raise HoistedConstantWritten(
                        "{} writes into a tensor created by a constant factory call".format(func))

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if func in _FACTORIES:
            key = _key_of(func, args, kwargs)
            q = self._queues.get(key) if key is not None else None
            if q:
                self.served += 1
                return q.pop(0)
        else:
            self._written(func, args, kwargs)
        return func(*args, **(kwargs or {}))
