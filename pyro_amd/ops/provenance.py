"""Data-flow provenance of non-reparameterised sample sites (the role of pyro/ops/provenance.py in
TraceGraph_ELBO, tracegraph_elbo.py:178-236: "which cost terms depend on this site?").

A value drawn at a non-reparameterised site is wrapped in :class:`ProvenanceTensor`, a
``torch.Tensor`` subclass that carries a frozenset of site names; every torch function applied to
such tensors returns results that carry the union of its inputs' sets.  The fused distribution
classes evaluate their densities in HIP kernels (no torch function in between), so the provenance
of a SITE is not read off its log-probability tensor: :func:`site_provenance` takes the union over
the site's value and every tensor reachable from its distribution object (its parameters, also
through ``Independent`` / ``MaskedDistribution`` / lazy operands), which is the same set the
reference obtains by pushing the parameters through ``log_prob``.
"""
import torch

_EMPTY = frozenset()


class ProvenanceTensor(torch.Tensor):
    """``ProvenanceTensor(data, provenance)``: ``data`` (same storage) carrying the frozenset."""
    _pa_provenance = _EMPTY

    def __new__(cls, data, provenance=_EMPTY, **kwargs):
        if not provenance:
            return data
        out = data.as_subclass(cls)
        out._pa_provenance = frozenset(provenance) | get_provenance(data)
        return out

    def __init__(self, data, provenance=_EMPTY):
        pass

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        found = set()
        _collect(args, found)
        _collect(kwargs, found)
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return _attach(out, frozenset(found)) if found else out


def _collect(x, found, depth=0):
    if isinstance(x, torch.Tensor):
        found.update(getattr(x, "_pa_provenance", _EMPTY))
    elif isinstance(x, (list, tuple, set, frozenset)):
        for v in x:
            _collect(v, found, depth)
    elif isinstance(x, dict):
        for v in x.values():
            _collect(v, found, depth)


def _attach(out, provenance):
    if isinstance(out, torch.Tensor):
        if type(out) is not ProvenanceTensor:
            out = out.as_subclass(ProvenanceTensor)
        out._pa_provenance = provenance | out.__dict__.get("_pa_provenance", _EMPTY)
        return out
    if isinstance(out, tuple):
        items = [_attach(v, provenance) for v in out]
        return type(out)(*items) if hasattr(out, "_fields") else tuple(items)    # namedtuple results
    if isinstance(out, list):
        return [_attach(v, provenance) for v in out]
    return out


def track_provenance(x, provenance):
    """An alias of ``x`` (same storage, same autograd history) carrying ``provenance`` in addition to
    what ``x`` carries; ``x`` itself is left as it is."""
    provenance = frozenset(provenance)
    if not provenance:
        return x
    if isinstance(x, torch.Tensor):
        tagged = x.as_subclass(ProvenanceTensor)     # a new Python object even for a tagged x
        tagged._pa_provenance = provenance | get_provenance(x)
        return tagged
    # containers: every tensor inside, the container type kept
    if isinstance(x, (list, set, frozenset)):
        return type(x)(track_provenance(v, provenance) for v in x)
    if isinstance(x, tuple):
        items = [track_provenance(v, provenance) for v in x]
        return type(x)(*items) if hasattr(x, "_fields") else tuple(items)
    if isinstance(x, dict):
        return type(x)((k, track_provenance(v, provenance)) for k, v in x.items())
    return x


def get_provenance(x):
    found = set()
    _collect(x, found)
    return frozenset(found)


def detach_provenance(x):
    if isinstance(x, ProvenanceTensor):
        return x.as_subclass(torch.Tensor)
    return x


def _walk_object(obj, found, seen, depth):
    if depth > 4 or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        found.update(getattr(obj, "_pa_provenance", _EMPTY))
        return
    if isinstance(obj, (list, tuple)):
        for v in obj:
            _walk_object(v, found, seen, depth + 1)
        return
    if isinstance(obj, dict):
        for v in obj.values():
            _walk_object(v, found, seen, depth + 1)
        return
    d = getattr(obj, "__dict__", None)
    if d is not None and not isinstance(obj, (type, torch.nn.Module)):
        for v in list(d.values()):
            if isinstance(v, (torch.Tensor, list, tuple, dict)) or hasattr(v, "__dict__"):
                _walk_object(v, found, seen, depth + 1)


def site_provenance(site):
    """The non-reparameterised sites the log-probability of ``site`` depends on: through its
    value, its distribution's parameters, its mask and its scale."""
    found, seen = set(), set()
    for part in (site.get("value"), site.get("fn"), site.get("mask"), site.get("scale"),
                 site.get("args"), site.get("kwargs")):
        if part is not None:
            _walk_object(part, found, seen, 0)
    return frozenset(found)
