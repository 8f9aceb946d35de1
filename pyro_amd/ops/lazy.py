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


from .torch_library import dispatcher_op as _dispatcher_op
_MATMUL_NAMES = ("matmul", "__matmul__")
MIN_ROWS = 256              # below this the unfused route is as fast and nothing is deferred
ENABLED = {"on": True}


def _on_device(t):
    """Recognition is for device tensors (the kernels have no CPU form); the host tests of the
    recognition LOGIC replace this predicate."""
    return t.is_cuda


def _plain(t):
    """The plain tensor behind a latent: the very object the trace holds (so that downstream code
    that keys on tensor identity -- e.g. the gradient chaining of the ELBO assembly -- sees the
    site's value, not a view of it)."""
    if type(t) is torch.Tensor or not isinstance(t, torch.Tensor):
        return t
    origin = getattr(t, "_pa_origin", None)
    return origin if origin is not None else t.as_subclass(torch.Tensor)


_DTYPES = (torch.float32,)    # what the fused kernels compute in (host tests of the logic widen it)


def _is_design_transpose(m):
    """m == X.t() for a contiguous, constant [N, D] float32 device matrix X?"""
    return (type(m) is torch.Tensor and m.dim() == 2 and not m.requires_grad and _on_device(m)
            and m.dtype in _DTYPES and m.shape[1] >= MIN_ROWS and m.shape[0] <= 128
            and m.stride(0) == 1 and m.stride(1) == m.shape[0])


def _is_design(m):
    return (type(m) is torch.Tensor and m.dim() == 2 and not m.requires_grad and _on_device(m)
            and m.dtype in _DTYPES and m.shape[0] >= MIN_ROWS and m.shape[1] <= 128
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
        if ENABLED["on"] and len(args) == 2 and not kwargs and isinstance(args[0], LatentTensor) \
                and getattr(func, "__name__", "") == "__getitem__":
            ids = _group_gather_index(args[0], args[1])
            if ids is not None:
                return DeferredGroupDot(_plain(args[0]), ids)               # w[..., g, :]
        for a in args:
            if isinstance(a, _DEFERRED):       # the deferred operand's own handler decides
                return type(a).__torch_function__(func, types, args, kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)       # plain tensors come back: laziness ends here


def _group_gather_index(w, index):
    """``index`` selects rows of the second-to-last dim of ``w`` by ONE 1-D int64 device tensor and
    keeps every other dim whole (``w[..., g, :]``, ``w[:, g]``, ``w[g]`` for a 2-D w): the id tensor,
    else None."""
    if w.dim() < 2 or w.shape[-1] > 128:
        return None
    if type(index) is torch.Tensor:
        index = (index,)
    if not isinstance(index, tuple):
        return None
    ids, pos, n_before, seen_ellipsis = None, None, 0, False
    for k, it in enumerate(index):
        if it is Ellipsis:
            if seen_ellipsis:
                return None
            seen_ellipsis = True
        elif isinstance(it, slice):
            if it != slice(None):
                return None
        elif type(it) is torch.Tensor:
            if ids is not None:
                return None
            ids, pos = it, k
        else:
            return None
    if ids is None or ids.dtype != torch.int64 or ids.dim() != 1 or not _on_device(ids) \
            or ids.requires_grad or ids.shape[0] < MIN_ROWS or not ids.is_contiguous():
        return None
    # the dim the tensor index lands on
    if seen_ellipsis:
        e = [k for k, it in enumerate(index) if it is Ellipsis][0]
        dim = pos if pos < e else w.dim() - (len(index) - pos)
    else:
        dim = pos
    return ids if dim == w.dim() - 2 else None


def as_latent(value):
    if (ENABLED["on"] and type(value) is torch.Tensor and _on_device(value)
            and value.dtype in _DTYPES and 1 <= value.dim() and value.shape[-1] <= 128):
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
        if len(args) == 2 and not kwargs and getattr(func, "__name__", "") in ("add", "__add__", "__radd__"):
            d, o = (args[0], args[1]) if isinstance(args[0], DeferredMatmul) else (args[1], args[0])
            if not isinstance(o, DeferredMatmul):
                return d._with_bias(o)                                      # b + logits

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


# ---- the hierarchical GLM in the reference's formulation ---------------------
#     logits = (w[..., g, :] * X).sum(-1) + b            # g: int64 [N], UNSORTED group ids
#     pyro.sample("obs", Bernoulli(logits=logits), obs=y)
# The reference materialises the gathered weights [P, N, D] (82 GB at N=1e7, P=64, D=32), the product,
# the logits, the log-probs and their autograd duals.  Here the advanced index on a latent returns a
# DeferredGroupDot in stage "gather"; multiplying by the constant [N, D] design matrix moves it to
# stage "product", ``.sum(-1)`` to stage "logits", ``+ b`` attaches the bias; ``Bernoulli(logits=...)``
# turns stage "logits" into the grouped plane-image site (kernels.grouped_rows_of: the sort permutation
# and segment table, built once per id tensor).  ANY other use evaluates exactly what was written.
class DeferredGroupDot:
    def __init__(self, w, ids, stage="gather", X=None, bias=None):
        self.w, self.ids, self.stage, self.X, self.bias = w, ids, stage, X, bias
        N = ids.shape[0]
        lead = tuple(w.shape[:-2])
        if stage == "logits":
            shape = lead + (N,)
            if isinstance(bias, torch.Tensor):
                shape = tuple(torch.broadcast_shapes(shape, bias.shape))
        else:
            shape = lead + (N, w.shape[-1])
        self.shape = torch.Size(shape)
        self.dtype, self.device = w.dtype, w.device

    def dim(self):
        return len(self.shape)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    @property
    def ndim(self):
        return len(self.shape)

    def _times(self, other):
        if self.stage == "gather" and _is_design(other) and other.shape[0] == self.ids.shape[0] \
                and other.shape[1] == self.w.shape[-1] and other.dtype == self.dtype \
                and other.device == self.device:
            return DeferredGroupDot(self.w, self.ids, "product", other)
        return self.materialize() * other

    __mul__ = _times
    __rmul__ = _times

    def sum(self, *args, **kwargs):
        dim = args[0] if args else kwargs.get("dim")
        if self.stage == "product" and isinstance(dim, int) and not isinstance(dim, bool) \
                and dim % len(self.shape) == len(self.shape) - 1 and len(args) <= 1 \
                and not kwargs.get("keepdim", False) and kwargs.get("dtype") is None \
                and set(kwargs) <= {"dim", "keepdim", "dtype"}:
            return DeferredGroupDot(self.w, self.ids, "logits", self.X)
        return self.materialize().sum(*args, **kwargs)

    def _with_bias(self, other):
        other = _plain(other) if isinstance(other, torch.Tensor) else other
        if self.stage == "logits" and self.bias is None:
            if isinstance(other, (int, float)):
                return DeferredGroupDot(self.w, self.ids, "logits", self.X,
                                        torch.full((), float(other), dtype=self.dtype, device=self.device))
            if isinstance(other, torch.Tensor) and other.dtype == self.dtype and other.device == self.device \
                    and (other.dim() == 0 or other.shape[-1] == 1) and other.dim() <= len(self.shape):
                return DeferredGroupDot(self.w, self.ids, "logits", self.X, other)
        return self.materialize() + other

    __add__ = _with_bias
    __radd__ = _with_bias

    def as_grouped_linear_logits(self):
        """The fused site's lazy operand, or None when this is not (yet) the hierarchical GLM or the
        grouped kernel cannot serve it."""
        if self.stage != "logits":
            return None
        from .. import kernels
        from ..distributions.families import GroupedLinearLogits
        try:
            segs = kernels.grouped_rows_of(self.ids, self.w.shape[-2])
            if segs is None:
                return None
            return GroupedLinearLogits(self.X, self.w, self.bias, segs)
        except (ValueError, kernels.Unsupported):
            return None

    def materialize(self):
        out = self.w[..., self.ids, :]
        if self.stage != "gather":
            out = out * self.X
        if self.stage == "logits":
            out = out.sum(-1)
            if self.bias is not None:
                out = out + self.bias
        return out

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if len(args) == 2 and not kwargs:
            a, b = args
            if name in ("mul", "__mul__", "__rmul__", "multiply"):
                d, o = (a, b) if isinstance(a, DeferredGroupDot) else (b, a)
                if not isinstance(o, DeferredGroupDot):
                    return d._times(o)
            if name in ("add", "__add__", "__radd__"):
                d, o = (a, b) if isinstance(a, DeferredGroupDot) else (b, a)
                if not isinstance(o, DeferredGroupDot):
                    return d._with_bias(o)
        if name == "sum" and args and isinstance(args[0], DeferredGroupDot):
            return args[0].sum(*args[1:], **kwargs)

        def ev(x):
            if isinstance(x, DeferredGroupDot):
                return x.materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(ev(v) for v in x)
            return x
        return func(*ev(args), **{k: ev(v) for k, v in kwargs.items()})

    def __getattr__(self, name):        # only reached for attributes not defined above
        return getattr(self.materialize(), name)

    def _binary(name):
        def op(self, other):
            return getattr(self.materialize(), name)(other)
        op.__name__ = name
        return op

    for _n in ("__sub__", "__rsub__", "__truediv__", "__rtruediv__", "__neg__", "__getitem__",
               "__matmul__", "__rmatmul__", "__pow__", "__lt__", "__gt__", "__le__", "__ge__"):
        locals()[_n] = _binary(_n) if _n != "__neg__" else (lambda self: -self.materialize())
    del _n, _binary


_DEFERRED = (DeferredMatmul, DeferredGroupDot)


# ---- the word histogram of an amortised guide (examples/lda.py:113-121) --------------------------
# The reference guide writes
#     counts = torch.zeros(V, B).scatter_add(0, data, torch.ones(data.shape))
#     doc_topics = predictor(counts.transpose(0, 1))          # nn.Sequential(nn.Linear(V, H), ...)
# on every step: a dense [V, B] matrix that is a pure function of the (constant) corpus, then four
# f32 GEMMs over it.  While a guide / model runs under ``watch_histograms()`` (the ELBO estimators
# switch it on around their traces), a scatter_add of FRESH ones into FRESH zeros along dim 0 with a
# 2-D int64 index returns a DeferredCounts instead of running; transposing keeps it deferred;
# ``torch.nn.functional.linear`` on the [B, V] orientation takes the bag-of-words kernels
# (pa_bow_linear_fwd / _bwd over the corpus's cached histogram images); ANY other use materialises
# exactly what the user wrote.
class DeferredCounts:
    def __init__(self, words, V, dtype, transposed=False):
        self.words, self.V, self.dtype, self.transposed = words, int(V), dtype, transposed
        self.device = words.device
        vb = (self.V, words.shape[1])
        self.shape = torch.Size(vb[::-1] if transposed else vb)

    def dim(self):
        return 2

    ndim = 2

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def transpose(self, d0, d1):
        if {d0 % 2, d1 % 2} == {0, 1}:
            return DeferredCounts(self.words, self.V, self.dtype, not self.transposed)
        return self

    def t(self):
        return DeferredCounts(self.words, self.V, self.dtype, not self.transposed)

    T = property(t)
    mT = property(t)

    def materialize(self):
        w = self.words
        with torch._C.DisableTorchFunction():        # (the watcher must not defer this one again)
            out = torch.zeros((self.V, w.shape[1]), dtype=self.dtype, device=w.device).scatter_add(
                0, w, torch.ones(w.shape, dtype=self.dtype, device=w.device))
        return out.transpose(0, 1) if self.transposed else out

    def _linear(self, weight, bias):
        """F.linear(self, weight, bias) on the kernels, or None when they do not cover the call."""
        from .. import kernels
        if not (ENABLED["on"] and self.transposed and self.dtype == torch.float32
                and isinstance(weight, torch.Tensor) and weight.dim() == 2 and _on_device(weight)
                and weight.dtype == torch.float32 and weight.shape[1] == self.V
                and weight.shape[0] <= 128 and self.V % 128 == 0
                and (bias is None or (isinstance(bias, torch.Tensor) and bias.dtype == torch.float32))):
            return None
        images = kernels.bow_images_of(self.words, self.V)
        if images is None:
            return None
        B = self.words.shape[1]
        if B >= TALL_MIN_ROWS and FUSE_ACTIVATION["on"]:
            return DeferredLinear("bow", (weight, bias, images[0], images[1], B), (B, weight.shape[0]), weight)
        out = _BowLinear.invoke(weight, bias, images[0], images[1], B)
        return out.as_subclass(TallActivation) if out.shape[0] >= TALL_MIN_ROWS else out

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.nn.functional.linear and args and isinstance(args[0], DeferredCounts):
            weight = args[1] if len(args) > 1 else kwargs.get("weight")
            bias = args[2] if len(args) > 2 else kwargs.get("bias")
            out = args[0]._linear(weight, bias)
            if out is not None:
                return out

        def ev(x):
            if isinstance(x, DeferredCounts):
                return x.materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(ev(v) for v in x)
            return x
        return func(*ev(args), **{k: ev(v) for k, v in kwargs.items()})

    def __getattr__(self, name):        # only reached for attributes not defined above
        return getattr(self.materialize(), name)


# ---- the layers after it: activations of a large batch ---------------------------------------------
# What the bag-of-words layer returns is a plain tensor in every respect but one: it remembers that it
# is the activation of a LARGE batch, and so do the results of element-wise functions of it.  A
# following ``F.linear`` then takes its weight gradient dW = d_out^T input through pa_tsgemm_tn (the
# long dimension split over the chip) instead of torch's mm backward (one rocBLAS product over 1e5
# rows: 342 us for a 100 x 100 weight at B = 1e5).  Forward and input gradient stay torch's.
TALL_MIN_ROWS = 4096
_ELEMENTWISE = {"sigmoid", "tanh", "relu", "softmax", "log_softmax", "softplus", "gelu", "elu",
                "leaky_relu", "dropout", "silu", "exp", "log"}


class TallActivation(torch.Tensor):
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if func is torch.nn.functional.linear and args and isinstance(args[0], TallActivation) \
                and ENABLED["on"]:
            x = args[0]
            weight = args[1] if len(args) > 1 else kwargs.get("weight")
            bias = args[2] if len(args) > 2 else kwargs.get("bias")
            if (x.dim() == 2 and x.shape[0] >= TALL_MIN_ROWS and x.shape[1] <= 128 and _on_device(x)
                    and x.dtype == torch.float32 and type(weight) in (torch.Tensor, torch.nn.Parameter)
                    and weight.dim() == 2 and weight.shape[0] <= 128 and weight.dtype == torch.float32):
                if FUSE_ACTIVATION["on"]:
                    return DeferredLinear("tall", (x.as_subclass(torch.Tensor), weight, bias),
                                          (x.shape[0], weight.shape[0]), weight)
                out = _TallLinear.invoke(x.as_subclass(torch.Tensor), weight, bias)
                return out.as_subclass(TallActivation)
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if name in _ELEMENTWISE and isinstance(out, torch.Tensor) and args \
                and isinstance(args[0], TallActivation) and out.shape[:1] == args[0].shape[:1] \
                and out.dim() == 2:
            return out.as_subclass(TallActivation)
        return out


FUSE_ACTIVATION = {"on": True}


class DeferredLinear:
    """A Linear layer over a tall batch that has not been launched yet.  nn.Sequential(Linear, Sigmoid, ...)
    (examples/lda.py:84-87) hands it to ``torch.sigmoid`` next: the layer then runs with the sigmoid in its
    kernel's epilogue, and its backward takes the gradient THROUGH the sigmoid in the operand loads of dx /
    dW / db (pa_tall_linear_act, pa_tall_wgrad_act, pa_bow_linear_*_act) -- torch's two extra passes over
    [B, features] in each direction are gone.  ANY other use launches the plain layer and goes on with its
    result (a TallActivation), exactly as before."""

    def __init__(self, kind, args, shape, like):
        self.kind, self.args = kind, args
        self.shape = torch.Size(shape)
        self.dtype, self.device = like.dtype, like.device
        self._plain = None
        self._sigmoid = None

    ndim = 2

    def dim(self):
        return 2

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def materialize(self, sigmoid=False):
        if sigmoid:
            if self._sigmoid is None:          # (a second torch.sigmoid of the same object: the same launch's result)
                fn = _BowLinearSigmoid if self.kind == "bow" else _TallLinearSigmoid
                self._sigmoid = fn.invoke(*self.args).as_subclass(TallActivation)
            return self._sigmoid
        if self._plain is None:
            fn = _BowLinear if self.kind == "bow" else _TallLinear
            self._plain = fn.invoke(*self.args).as_subclass(TallActivation)
        return self._plain

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.sigmoid, torch.nn.functional.sigmoid, torch.Tensor.sigmoid) and len(args) == 1 \
                and not kwargs and isinstance(args[0], DeferredLinear) and ENABLED["on"]:
            return args[0].materialize(sigmoid=True)

        def ev(x):
            if isinstance(x, DeferredLinear):
                return x.materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(ev(v) for v in x)
            return x
        return func(*ev(args), **{k: ev(v) for k, v in kwargs.items()})

    def sigmoid(self):
        return self.materialize(sigmoid=True) if ENABLED["on"] else self.materialize().sigmoid()

    def __getattr__(self, name):        # only reached for attributes not defined above
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def _binary(name):
        def op(self, *other):
            return getattr(self.materialize(), name)(*other)
        op.__name__ = name
        return op

    for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__",
               "__rtruediv__", "__floordiv__", "__rfloordiv__", "__mod__", "__rmod__", "__neg__", "__abs__",
               "__getitem__", "__matmul__", "__rmatmul__", "__pow__", "__rpow__", "__lt__", "__gt__", "__le__",
               "__ge__", "__eq__", "__ne__", "__len__", "__iter__", "__bool__", "__float__", "__int__",
               # (the operators a tensor has and the proxy lacked; in-place forms act on the
               #  materialised layer output, which the proxy keeps handing out afterwards)
               "__invert__", "__and__", "__rand__", "__or__", "__ror__", "__xor__", "__rxor__", "__setitem__",
               "__contains__", "__iadd__", "__isub__", "__imul__", "__itruediv__", "__pos__", "__lshift__",
               "__rshift__", "__index__", "__array__", "__repr__", "__format__"):
        locals()[_n] = _binary(_n)
    del _n, _binary
    __hash__ = object.__hash__          # (``==`` is the tensor's: identity keeps the object usable as a key)


@_dispatcher_op("tall_linear_sigmoid")
class _TallLinearSigmoid(torch.autograd.Function):
    """sigmoid(F.linear(x, weight, bias)) over a tall batch: the sigmoid in pa_tall_linear_act's epilogue; the
    backward never forms g * (1 - y) * y -- dx, dW and db read y next to g."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .. import kernels
        x = x.contiguous()
        w = weight.detach().contiguous()
        n_out, n_in = w.shape
        y = kernels.tall_linear(x, w, 1, n_in, n_out, None if bias is None else bias.detach().contiguous(),
                                sigmoid=True)
        ctx.save_for_backward(x, w, y)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import kernels
        x, w, y = ctx.saved_tensors
        g = g.contiguous()
        n_out, n_in = w.shape
        dx = kernels.tall_linear(g, w, n_in, 1, n_in, y_mul=y) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = kernels.tall_wgrad(g, x, want_bias=ctx.has_bias and ctx.needs_input_grad[2], y_mul=y)
        return dx, dW, db


@_dispatcher_op("bow_linear_sigmoid")
class _BowLinearSigmoid(torch.autograd.Function):
    """sigmoid(bias + counts @ W.T): see _BowLinear and _TallLinearSigmoid; the bias gradient comes from the
    partial sums the operand-split pass of the backward leaves behind."""

    @staticmethod
    def forward(ctx, weight, bias, image_a, image_b, B):
        from .. import kernels
        w = weight.detach().contiguous()
        y = kernels.bow_linear_fwd(image_a, w, None if bias is None else bias.detach().contiguous(), B,
                                   sigmoid=True)
        ctx.save_for_backward(y)
        ctx.image_b, ctx.V, ctx.H, ctx.has_bias = image_b, w.shape[1], w.shape[0], bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import kernels
        y, = ctx.saved_tensors
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        if not ctx.needs_input_grad[0] and not want_b:
            return None, None, None, None, None
        out = kernels.bow_linear_bwd(ctx.image_b, g, ctx.V, y_mul=y, want_bias=want_b)
        if not want_b:
            return out, None, None, None, None
        dW, part = out
        # [4, nkt, 32] -> [4, 32] in two coalesced stages (a thread per (chunk, unit) walks its chunk with its
        # neighbours on consecutive units; one long strided walk per unit is a cache line per element)
        nkt = part.shape[1]
        d = max((q for q in range(1, min(nkt, 256) + 1) if nkt % q == 0), default=1)
        db = part.view(4, nkt // d, d, 32).sum(2).sum(1)
        return dW, db.reshape(-1)[:ctx.H], None, None, None


@_dispatcher_op("tall_linear")
class _TallLinear(torch.autograd.Function):
    """F.linear over a tall batch on the kernels of csrc/tall.hip: forward and dx are pa_tall_linear
    with the weight addressed through strides (no transposed copy), dW and db come from ONE pass
    over the batch (pa_tall_wgrad)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .. import kernels
        x = x.contiguous()
        w = weight.detach().contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        n_out, n_in = w.shape
        # Wm = weight^T: Wm[r][c] = weight[c][r]
        return kernels.tall_linear(x, w, 1, n_in, n_out, None if bias is None else bias.detach().contiguous())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import kernels
        x, w = ctx.saved_tensors
        g = g.contiguous()
        n_out, n_in = w.shape
        dx = kernels.tall_linear(g, w, n_in, 1, n_in) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = kernels.tall_wgrad(g, x, want_bias=ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dW, db


@_dispatcher_op("bow_linear")
class _BowLinear(torch.autograd.Function):
    """h = bias + counts @ W.T over the corpus's histogram images; d W = d_h.T @ counts, d bias =
    sum d_h; the histogram itself carries no gradient."""

    @staticmethod
    def forward(ctx, weight, bias, image_a, image_b, B):
        from .. import kernels
        w = weight.detach().contiguous()
        out = kernels.bow_linear_fwd(image_a, w, None if bias is None else bias.detach().contiguous(), B)
        ctx.image_b, ctx.V, ctx.has_bias = image_b, w.shape[1], bias is not None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import kernels
        dW = kernels.bow_linear_bwd(ctx.image_b, g, ctx.V) if ctx.needs_input_grad[0] else None
        db = g.sum(0) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        return dW, db, None, None, None


class _HistogramWatcher(torch.overrides.TorchFunctionMode):
    def __init__(self):
        super().__init__()
        self._fresh = {}                   # id(tensor) -> (weak reference, fill value)

    def _note(self, t, value):
        import weakref
        if type(t) is torch.Tensor and _on_device(t):
            self._fresh[id(t)] = (weakref.ref(t), value)

    def _is_fresh(self, t, value):
        ent = self._fresh.get(id(t))
        return ent is not None and ent[0]() is t and ent[1] == value and t._version == 0

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.zeros or func is torch.ones:
            out = func(*args, **kwargs)
            self._note(out, 0.0 if func is torch.zeros else 1.0)
            return out
        if func is torch.Tensor.scatter_add or func is torch.scatter_add:
            full = list(args) + [kwargs[k] for k in ("dim", "index", "src") if k in kwargs]
            if len(full) == 4:
                base, dim, index, src = full
                if (ENABLED["on"] and isinstance(dim, int) and dim == 0 and type(index) is torch.Tensor
                        and index.dtype == torch.int64 and index.dim() == 2 and _on_device(index)
                        and type(base) is torch.Tensor and base.dim() == 2
                        and base.shape[1] == index.shape[1] and base.is_floating_point()
                        and type(src) is torch.Tensor and src.shape == index.shape
                        and src.dtype == base.dtype and not base.requires_grad and not src.requires_grad
                        and self._is_fresh(base, 0.0) and self._is_fresh(src, 1.0)):
                    return DeferredCounts(index, base.shape[0], base.dtype)
        return func(*args, **kwargs)


def watch_histograms():
    """Context manager: recognise the histogram construction above in code run inside (a no-op
    context when the recognition is switched off)."""
    import contextlib
    return _HistogramWatcher() if ENABLED["on"] else contextlib.nullcontext()

