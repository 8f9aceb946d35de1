"""TEST INFRASTRUCTURE ONLY -- minimal stand-in for the third-party `opt_einsum`
package (pinned by the reference only as ``opt_einsum>=2.3.2``, setup.py:107; absent
from this image, no network).

It exists so that the *unmodified* reference Pyro under /root/reference can be
imported in the build container to generate golden vectors (tests/golden/make_golden.py).
Nothing in the product package `pyro_amd` imports it.

Surface provided (the call sites are listed in SURVEY.md section 8c):
  get_symbol, shared_intermediates, sharing.count_cached_ops, contract,
  contract_expression, contract_path.

``contract`` performs a left-to-right *pairwise* contraction and dispatches every
pairwise step to the backend module named by ``backend=`` (its ``einsum``), which is
what the reference's log-space / adjoint backends rely on.

IMPORTANT: import torch *before* putting this directory on sys.path, otherwise torch
picks the stand-in up for torch.einsum path optimisation.
"""
import contextlib
import importlib

from . import sharing  # noqa: F401
from .sharing import shared_intermediates  # noqa: F401

_BASE = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def get_symbol(i):
    if i < 52:
        return _BASE[i]
    return chr(i + 140)


def _parse(equation, n):
    equation = equation.replace(" ", "")
    if "->" in equation:
        lhs, out = equation.split("->")
    else:
        lhs = equation
        counts = {}
        for c in lhs.replace(",", ""):
            counts[c] = counts.get(c, 0) + 1
        out = "".join(sorted(c for c, k in counts.items() if k == 1))
    ins = lhs.split(",")
    assert len(ins) == n, (equation, n)
    return ins, out


def _backend(name):
    if name in ("auto", "torch"):
        import torch

        class _T:
            einsum = staticmethod(torch.einsum)

        return _T
    return importlib.import_module(name)


def _check_sizes(ins, operands):
    """opt_einsum validates label sizes before contracting (its error text is what the reference's
    tests match on: tests/ops/test_contract.py:727-733)."""
    sizes = {}
    for k, (labels, op) in enumerate(zip(ins, operands)):
        shape = getattr(op, "shape", None)
        if shape is None or len(shape) != len(labels):
            continue
        for c, n in zip(labels, shape):
            n = int(n)
            if c in sizes and sizes[c] != n and 1 not in (sizes[c], n):
                raise ValueError("Size of label '{}' for operand {} ({}) does not match previous "
                                 "terms ({}).".format(c, k, n, sizes[c]))
            sizes[c] = max(sizes.get(c, 1), n)


def contract(equation, *operands, backend="auto", **kwargs):
    ins, out = _parse(equation, len(operands))
    _check_sizes(ins, operands)
    be = _backend(backend)
    cache = sharing.current_cache()
    ins = list(ins)
    ops = list(operands)
    if len(ops) == 1:
        return _cached(be, cache, backend, ins[0] + "->" + out, ops)
    while len(ops) > 1:
        a, b = ops.pop(0), ops.pop(0)
        ia, ib = ins.pop(0), ins.pop(0)
        rest = set("".join(ins)) | set(out)
        keep = [c for c in dict.fromkeys(ia + ib) if c in rest]
        if not ops:
            io = out
        else:
            io = "".join(keep)
        res = _cached(be, cache, backend, ia + "," + ib + "->" + io, [a, b])
        ops.insert(0, res)
        ins.insert(0, io)
    return ops[0]


def _cached(be, cache, backend, eq, ops):
    if cache is None:
        return be.einsum(eq, *ops)
    key = ("einsum", backend, eq) + tuple(id(o) for o in ops)
    if key not in cache:
        # keep operands alive so ids stay unique while the cache lives
        cache[key] = (be.einsum(eq, *ops), ops)
    return cache[key][0]


class _Expr:
    def __init__(self, equation, shapes, kwargs):
        self.equation = equation
        self.kwargs = kwargs

    def __call__(self, *operands, backend="auto", **kw):
        return contract(self.equation, *operands, backend=backend)


def contract_expression(equation, *shapes, **kwargs):
    return _Expr(equation, shapes, kwargs)


def contract_path(equation, *operands, **kwargs):
    n = len(operands)
    return [(0, 1)] * (n - 1) if n > 1 else [(0,)], None
