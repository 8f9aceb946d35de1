"""Lazy recognition of the plated GLM in UNMODIFIED model text.

The reference model of BASELINE config 2 writes

    logits = w @ X.t()                      # w: a sampled latent, [D] or [P, 1, D]
    logits = logits.squeeze(-2) ...
    pyro.sample("obs", Bernoulli(logits=logits + b), obs=y)

and the reference then materialises [P, N] logits, log-probs and their autograd duals
(pyro/poutine/trace_struct.py:264-278).  To reach the fused one-pass kernel
(pa_glm_bernoulli*_fwd_bwd) without an API the reference does not have, the values ``pyro.sample``
hands to the model are a transparent ``torch.Tensor`` subclass whose ONLY special behaviour is:
a matmul of such a latent with a constant matrix that is the transpose view of a contiguous
[N, D] float32 device matrix returns a :class:`DeferredMatmul` instead of running.  A
DeferredMatmul supports exactly what such model text does next (``squeeze``, ``+ bias``, shape
queries); ``Bernoulli(logits=...)`` turns it into the fused site; ANY other use (a torch function,
an attribute, arithmetic beyond the bias) materialises the product -- then the model runs exactly
as written, on the unfused route.  Every other operation on a latent returns plain tensors.
"""
import torch

_MATMUL_NAMES = ("matmul", "__matmul__")
MIN_ROWS = 256              # below this the unfused route is as fast and nothing is deferred
ENABLED = {"on": True}


def _plain(t):
    """The plain tensor behind a latent: the very object the trace holds (so that downstream code
    that keys on tensor identity -- e.g. the gradient chaining of the ELBO assembly -- sees the
    site's value, not a view of it)."""
    if type(t) is torch.Tensor or not isinstance(t, torch.Tensor):
        return t
    origin = getattr(t, "_pa_origin", None)
    return origin if origin is not None else t.as_subclass(torch.Tensor)


def _is_design_transpose(m):
    """m == X.t() for a contiguous, constant [N, D] float32 device matrix X?"""
    return (type(m) is torch.Tensor and m.dim() == 2 and not m.requires_grad and m.is_cuda
            and m.dtype == torch.float32 and m.shape[1] >= MIN_ROWS and m.shape[0] <= 128
            and m.stride(0) == 1 and m.stride(1) == m.shape[0])


def _is_design(m):
    return (type(m) is torch.Tensor and m.dim() == 2 and not m.requires_grad and m.is_cuda
            and m.dtype == torch.float32 and m.shape[0] >= MIN_ROWS and m.shape[1] <= 128
            and m.is_contiguous())


class LatentTensor(torch.Tensor):
    """What pyro_amd.sample returns for a float device value (see module docstring)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if any(t is not LatentTensor and t is not torch.Tensor and t.__name__ == "ProvenanceTensor"
               for t in types):
            return NotImplemented          # ops.provenance handles the call (and keeps its tags)
        if ENABLED["on"] and getattr(func, "__name__", "") in _MATMUL_NAMES and len(args) == 2 \
                and not kwargs:
            a, b = args
            if isinstance(a, LatentTensor) and _is_design_transpose(b) and a.dtype == b.dtype \
                    and a.shape[-1] == b.shape[0] and (a.dim() == 1 or a.shape[-2] == 1):
                return DeferredMatmul(b.t(), _plain(a))                     # w @ X.t()
            if isinstance(b, LatentTensor) and _is_design(a) and b.dim() == 1 \
                    and a.shape[1] == b.shape[0] and a.dtype == b.dtype:
                return DeferredMatmul(a, _plain(b))                         # X @ w
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)       # plain tensors come back: laziness ends here


def as_latent(value):
    if (ENABLED["on"] and type(value) is torch.Tensor and value.is_cuda
            and value.dtype == torch.float32 and 1 <= value.dim() and value.shape[-1] <= 128):
        lat = value.as_subclass(LatentTensor)
        lat._pa_origin = value
        return lat
    return value


class DeferredMatmul:
    """``w @ X.t()`` (or ``X @ w``) not yet evaluated.  ``w``: [D] or [..., 1, D]."""

    def __init__(self, X, w, squeezed=False, bias=None):
        self.X, self.w, self.squeezed, self.bias = X, w, squeezed, bias
        N = X.shape[0]
        if w.dim() == 1:
            shape = (N,)
        elif squeezed:
            shape = tuple(w.shape[:-2]) + (N,)
        else:
            shape = tuple(w.shape[:-1]) + (N,)
        if bias is not None and isinstance(bias, torch.Tensor):
            shape = tuple(torch.broadcast_shapes(shape, bias.shape))
        self.shape = torch.Size(shape)
        self.dtype, self.device = X.dtype, X.device

    # ---- what GLM model text does with its logits -----------------------------------------------
    def dim(self):
        return len(self.shape)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    @property
    def ndim(self):
        return len(self.shape)

    def squeeze(self, d=None):
        if self.bias is None and not self.squeezed and self.w.dim() > 1 and d is not None \
                and d % len(self.shape) == len(self.shape) - 2:
            return DeferredMatmul(self.X, self.w, True, None)
        return self.materialize().squeeze() if d is None else self.materialize().squeeze(d)

    def _with_bias(self, other):
        other = _plain(other) if isinstance(other, torch.Tensor) else other
        flat = self.w.dim() == 1 or self.squeezed
        if self.bias is None and flat:
            if isinstance(other, (int, float)):
                return DeferredMatmul(self.X, self.w, self.squeezed,
                                      torch.full((), float(other), dtype=self.dtype, device=self.device))
            if isinstance(other, torch.Tensor) and other.dtype == self.dtype and other.device == self.device \
                    and (other.dim() == 0 or other.shape[-1] == 1) \
                    and other.dim() <= max(len(self.shape), 1):
                return DeferredMatmul(self.X, self.w, self.squeezed, other)
        return self.materialize() + other

    __add__ = _with_bias
    __radd__ = _with_bias

    def as_linear_logits(self):
        """The fused site's lazy operand, or None when this shape is not the plated GLM."""
        from ..distributions.families import LinearLogits
        if self.w.dim() > 1 and not self.squeezed:
            return None
        try:
            return LinearLogits(self.X, self.w, self.bias)
        except ValueError:
            return None

    # ---- everything else: evaluate, then behave as the tensor ------------------------------------
    def materialize(self):
        out = self.w @ self.X.t()
        if self.squeezed:
            out = out.squeeze(-2)
        if self.bias is not None:
            out = out + self.bias
        return out

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def ev(x):
            if isinstance(x, DeferredMatmul):
                return x.materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(ev(v) for v in x)
            return x
        return func(*ev(args), **{k: ev(v) for k, v in (kwargs or {}).items()})

    def __getattr__(self, name):        # only reached for attributes not defined above
        return getattr(self.materialize(), name)

    def _binary(name):
        def op(self, other):
            return getattr(self.materialize(), name)(other)
        op.__name__ = name
        return op

    for _n in ("__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__",
               "__neg__", "__getitem__", "__matmul__", "__rmatmul__", "__pow__", "__lt__", "__gt__",
               "__le__", "__ge__"):
        locals()[_n] = _binary(_n) if _n != "__neg__" else (lambda self: -self.materialize())
    del _n, _binary
