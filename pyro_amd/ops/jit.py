"""``trace``: torch.jit.trace for functions that read ``pyro.param`` (reference: pyro/ops/jit.py:48-163
``CompiledFunction`` / ``trace``; used by JitTrace_ELBO, pyro/infer/trace_elbo.py:162-257).

What makes this work here is ops/torch_library.py: every kernel-backed operation of this package is a
dispatcher op (``pyro_amd::*``), so the recorded graph holds them as nodes; and the Philox stream: the
draws of a traced function read their block offset relative to a device-resident base word
(rng.GraphCapture's mechanism), which the wrapper sets before every call -- a replay of the graph draws
fresh numbers, the ones the eager function would have drawn at that point of the stream.

As in the reference, the function is traced once per (number of positional arguments, keyword
arguments); positional arguments must be tensors, everything else goes through keyword arguments and
is baked into the graph; all parameters must exist after the first (un-traced) call.
"""
import warnings
import weakref

import torch

from .. import poutine, rng
from ..primitives import param_unconstrained, validation_enabled


def _freeze(value):
    try:
        hash(value)
        return value
    except TypeError:
        if isinstance(value, (list, tuple)):
            return tuple(_freeze(v) for v in value)
        if isinstance(value, dict):
            return tuple(sorted((k, _freeze(v)) for k, v in value.items()))
        if isinstance(value, set):
            return frozenset(_freeze(v) for v in value)
        return ("id", id(value))


class _ReplaySafeDraws:
    """The Philox side of one traced function: relative block offsets against a device base word."""

    def __init__(self, device):
        self.cap = rng.GraphCapture(device) if device is not None and device.type == "cuda" else None

    def run(self, fn, first):
        cap = self.cap
        if cap is None:                       # host tensors (tests): the stream advances by itself
            return fn()
        rng._follow_torch()                   # (torch.manual_seed since the last draw restarts the stream)
        state = rng._STATE
        start = state["offset"]
        cap.base[:1].fill_(start)             # what *base + relative offset resolves against
        if first:
            assert rng._CAPTURE["active"] is None, "tracing inside a graph capture is not supported"
            cap.start = start
            rng._CAPTURE["active"] = cap
            try:
                out = fn()                    # the trace EXECUTES the function: the draws are consumed
            finally:
                rng._CAPTURE["active"] = None
            cap.used = state["offset"] - start
            return out
        out = fn()
        state["offset"] = start + cap.used
        return out


class CompiledFunction:
    """What ``trace`` returns; ``compiled`` maps call signatures to the torch.jit artefacts."""

    def __init__(self, fn, ignore_warnings=False, jit_options=None):
        self.fn = fn
        self.compiled = {}
        self.ignore_warnings = ignore_warnings
        self.jit_options = dict(jit_options or {})
        # (the reference's own option, pyro/ops/jit.py:71-75: timing of the compilation, not torch.jit.trace's)
        self._time_compilation = bool(self.jit_options.pop("time_compilation", False))
        self.compile_time = None
        self.jit_options.setdefault("check_trace", False)
        self._param_names = None
        self._draws = {}
        self._offsets = {}

    def _leaves(self):
        # through the "param" primitive, OUTSIDE the block below: an enclosing handler (SVI's parameter
        # capture) sees every parameter on every call, replayed or not (pyro/ops/jit.py:119-121)
        return [param_unconstrained(name) for name in self._param_names]

    def __call__(self, *args, **kwargs):
        key = (len(args), _freeze(kwargs))
        first = key not in self.compiled
        if first:
            with poutine.block():
                with poutine.trace(param_only=True) as capture:
                    self.fn(*args, **kwargs)          # creates the parameters (an ordinary eager call)
            self._param_names = sorted(capture.trace.nodes.keys())
            weakself = weakref.ref(self)
            n_params = len(self._param_names)

            def of_leaves_and_args(*leaves_and_args):
                # the leaves arrive as the graph's inputs; they ARE the param store's tensors, so
                # every pyro.param / autoguide read inside is a function of the inputs
                me = weakself()
                return me.fn(*leaves_and_args[n_params:], **kwargs)

            device = next((a.device for a in args if isinstance(a, torch.Tensor)), None)
            draws = _ReplaySafeDraws(device)
            example = tuple(self._leaves()) + tuple(args)
            with validation_enabled(False), warnings.catch_warnings():
                if self.ignore_warnings:
                    warnings.filterwarnings("ignore", category=torch.jit.TracerWarning)
                traced = draws.run(lambda: torch.jit.trace(of_leaves_and_args, example, **self.jit_options),
                                   first=True)
            self.compiled[key] = traced
            self._draws[key] = draws
            self._offsets[key] = [t.storage_offset() for t in example]
        inputs = tuple(self._leaves()) + tuple(args)
        # strided views taken inside the function (torch.as_strided in the site kernels' operand
        # frames) are recorded with the ABSOLUTE storage offset their base had under the tracer.  An
        # input that has moved inside its storage since -- the flat optimizer turns every parameter
        # into a view of one buffer at its first step -- is handed over as a (differentiable) copy
        inputs = tuple(t if t.storage_offset() == off else t.clone()
                       for t, off in zip(inputs, self._offsets[key]))
        with poutine.block(hide=self._param_names):
            with poutine.trace(param_only=True) as capture:
                ret = self._draws[key].run(lambda: self.compiled[key](*inputs), first=False)
        for name in capture.trace.nodes.keys():
            if name not in self._param_names:
                raise NotImplementedError("pyro_amd.ops.jit.trace assumes all params are created on "
                                          "first invocation, but found new param: {}".format(name))
        return ret


def trace(fn=None, ignore_warnings=False, jit_options=None):
    """Lazy ``torch.jit.trace`` for functions that call ``pyro.param`` (decorator or call form)."""
    if fn is None:
        return lambda f: trace(f, ignore_warnings=ignore_warnings, jit_options=jit_options)
    return CompiledFunction(fn, ignore_warnings=ignore_warnings, jit_options=jit_options)
