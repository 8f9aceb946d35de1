"""HIP-fused distribution families (the hot exponential-family sites of SURVEY 8a).

Each class IS the corresponding torch.distributions class (same constructor, attributes,
support, ``expand``) with ``log_prob`` routed to the fused element-wise kernel, ``rsample``
drawing from the Philox stream, and ``fused_log_prob_sum`` providing the one-kernel
log_prob -> scale_and_mask -> plate-sum reduction.  They require HIP tensors: CPU tensors raise.
"""
import torch

from ..ops import lazy as _lazy
from torch.distributions import constraints

from .. import _lib, rng
from . import fused
from .base import TorchDistribution, TorchDistributionMixin


def _maskable(mask):
    return mask is None or isinstance(mask, torch.Tensor)


_CONSTANTS = {}


def device_constant(value, dtype, device):
    """A cached read-only 0-dim tensor holding ``value`` on ``device``: created once by a fill
    kernel, then reused by every distribution object that is built from a Python number (one fill
    launch per constant per step otherwise; and a host-to-device copy -- a synchronisation that is
    not permitted while a hipGraph is being captured -- if built with torch.tensor)."""
    key = (float(value), dtype, device)
    t = _CONSTANTS.get(key)
    if t is None:
        t = torch.full((), float(value), dtype=dtype, device=device)
        _CONSTANTS[key] = t
    return t


def _on_device(*values):
    """Python-number parameters -> cached 0-dim tensors ON the device of the tensor parameters."""
    proto = next((v for v in values if isinstance(v, torch.Tensor)), None)
    if proto is None:
        return values
    dtype = proto.dtype if proto.is_floating_point() else torch.get_default_dtype()
    return tuple(v if isinstance(v, torch.Tensor) or v is None
                 else device_constant(v, dtype, proto.device)
                 for v in values)


class _FusedElementwise:
    """Overrides that must precede the torch class in the MRO (torch defines log_prob too)."""

    _dist_id = None

    def _params(self):
        raise NotImplementedError

    def log_prob(self, value):
        if self._validate_args:
            self._validate_sample(value)
        p0, p1 = self._params()
        base = getattr(self, "_base_params", None)
        if base is not None and base[0] is not None and isinstance(value, torch.Tensor):
            # the parameters as given BEFORE .expand() wherever the value carries the batch shape anyway: the
            # kernels broadcast by stride, and a parameter derived lazily from an expanded one (logits from
            # expanded probs) would be derived once per ELEMENT instead of once per table entry
            full = torch.broadcast_shapes(value.shape, p0.shape, p1.shape if p1 is not None else ())
            try:
                small = torch.broadcast_shapes(value.shape, base[0].shape,
                                               base[1].shape if base[1] is not None else ())
            except RuntimeError:
                small = None
            if small == full:
                p0, p1 = base
        return fused.log_prob(self._dist_id, value, p0, p1)

    def fused_log_prob_sum(self, value, scale=1.0, mask=None):
        if not _maskable(mask) or isinstance(scale, torch.Tensor):
            return None
        if self._validate_args:
            self._validate_sample(value)
        # un-expanded parameters (see fused_site_entry): the kernels broadcast by stride and
        # reduce the gradient to the operand's own shape themselves
        p0, p1 = getattr(self, "_base_params", None) or self._params()
        return fused.log_prob_sum(self._dist_id, value, p0, p1, mask, scale)

    def fused_site_entry(self, value, scale=1.0, mask=None):
        """(dist_id, value, p0, p1, mask, scale) for fused.SiteBatch, which sums many small sites
        in one launch; None if the site cannot be described that way."""
        if not _maskable(mask) or isinstance(scale, torch.Tensor):
            return None
        if self._validate_args:
            self._validate_sample(value)
        # the parameters as given BEFORE any .expand(): stride-0 expanded views would make the
        # gradient land on the expanded shape and be reduced by a separate autograd node
        p0, p1 = getattr(self, "_base_params", None) or self._params()
        return self._dist_id, value, p0, p1, mask, scale

    def expand(self, batch_shape, _instance=None):
        new = super().expand(batch_shape, _instance)
        new._base_params = getattr(self, "_base_params", None) or self._params()
        return new


class Normal(_FusedElementwise, torch.distributions.Normal, TorchDistributionMixin):
    _dist_id = _lib.DIST_NORMAL
    # a draw made ahead of time for this very distribution object (fused.meanfield_sample draws all
    # sites of a mean-field guide in one launch); handed out by rsample() when the shape matches
    _presampled = None
    # ... and what it was drawn from: (z [P, n] as the kernel wrote it, loc_out [n], scale [n], P).  A site
    # whose value IS that draw is scored in closed form (fused_linear_term below).
    _drawn = None

    def __init__(self, loc, scale, validate_args=None):
        loc, scale = _on_device(loc, scale)
        super().__init__(loc, scale, validate_args=validate_args)

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        new = type(self)(self.loc.expand(batch_shape), self.scale.expand(batch_shape),
                         validate_args=False)
        new._validate_args = self._validate_args
        new._base_params = getattr(self, "_base_params", None) or (self.loc, self.scale)
        new._presampled = self._presampled
        new._drawn = self._drawn
        if "has_rsample" in self.__dict__:      # has_rsample_() set on this instance
            new.has_rsample = self.__dict__["has_rsample"]
        return new

    def _params(self):
        return self.loc, self.scale

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        pre = self._presampled
        if pre is not None:
            if pre.shape != shape:
                raise ValueError("pre-drawn value of shape {} does not fit the requested {}".format(
                    tuple(pre.shape), tuple(shape)))
            return pre
        return fused.normal_rsample(self.loc, self.scale, shape)

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)

    def fused_score_term(self, value, scale=1.0, mask=None):
        """A guide site scored at its own reparameterised draw z = loc + scale * eps: a small vector of
        partial sums of scale * sum log q(z) from ONE pass over z, whose backward is one multiply.  With
        eps fixed the derivative through z cancels the eps terms of the direct ones (d/d loc = 0, d/d scale
        = -1/scale), which is what autograd arrives at through Normal.log_prob's three paths (reference:
        trace_elbo.py:142-160) up to rounding.  Offered after the many-small-sites launch has declined
        the site (too large); None when the value is not this object's own draw."""
        drawn = self._drawn
        if drawn is None or value is not self._presampled or mask is not None \
                or isinstance(scale, torch.Tensor):
            return None
        from .. import kernels
        z, loc, sc, P = drawn
        if not kernels.on_device(z):
            return None
        return fused.drawn_score(z, loc, sc, P, float(scale))


class LogNormal(_FusedElementwise, torch.distributions.LogNormal, TorchDistributionMixin):
    _dist_id = _lib.DIST_LOG_NORMAL

    def __init__(self, loc, scale, validate_args=None):
        loc, scale = _on_device(loc, scale)
        super().__init__(loc, scale, validate_args=validate_args)

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        new = type(self)(self.loc.expand(batch_shape), self.scale.expand(batch_shape),
                         validate_args=False)
        new._validate_args = self._validate_args
        new._base_params = getattr(self, "_base_params", None) or (self.loc, self.scale)
        return new

    def _params(self):
        return self.loc, self.scale

    def rsample(self, sample_shape=torch.Size()):
        return fused.normal_rsample(self.loc, self.scale, self._extended_shape(sample_shape)).exp()

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)


def _zero_loc_base(base_cls, scale):
    """base_cls(0, scale) with the zero location as a cached device constant: torch builds it with
    torch.tensor(0, device=...), a host-to-device copy (a synchronisation, and not permitted while
    a hipGraph is being captured)."""
    if isinstance(scale, torch.Tensor):
        zero = device_constant(0.0, scale.dtype if scale.is_floating_point()
                               else torch.get_default_dtype(), scale.device)
        return base_cls(zero, scale, validate_args=False)
    return base_cls(0, scale, validate_args=False)


class HalfCauchy(_FusedElementwise, torch.distributions.HalfCauchy, TorchDistributionMixin):
    _dist_id = _lib.DIST_HALF_CAUCHY

    def __init__(self, scale, validate_args=None):
        base = _zero_loc_base(torch.distributions.Cauchy, scale)
        torch.distributions.TransformedDistribution.__init__(
            self, base, torch.distributions.transforms.AbsTransform(), validate_args=validate_args)

    def expand(self, batch_shape, _instance=None):
        new = type(self)(self.scale.expand(torch.Size(batch_shape)), validate_args=False)
        new._validate_args = self._validate_args
        new._base_params = getattr(self, "_base_params", None) or self._params()
        return new

    def _params(self):
        return self.scale, None

    def log_prob(self, value):
        if self._validate_args:
            self._validate_sample(value)
        return fused.log_prob(self._dist_id, value, self.scale, None)


class HalfNormal(_FusedElementwise, torch.distributions.HalfNormal, TorchDistributionMixin):
    _dist_id = _lib.DIST_HALF_NORMAL

    def __init__(self, scale, validate_args=None):
        base = _zero_loc_base(torch.distributions.Normal, scale)
        torch.distributions.TransformedDistribution.__init__(
            self, base, torch.distributions.transforms.AbsTransform(), validate_args=validate_args)

    def expand(self, batch_shape, _instance=None):
        new = type(self)(self.scale.expand(torch.Size(batch_shape)), validate_args=False)
        new._validate_args = self._validate_args
        new._base_params = getattr(self, "_base_params", None) or self._params()
        return new

    def _params(self):
        return self.scale, None

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        eps = rng.normal(shape, self.scale.dtype, self.scale.device)
        return (eps * self.scale).abs()


class Exponential(_FusedElementwise, torch.distributions.Exponential, TorchDistributionMixin):
    _dist_id = _lib.DIST_EXPONENTIAL

    def _params(self):
        return self.rate, None


class _GammaFunctionFamily(_FusedElementwise):
    """Families whose normaliser needs lgamma / digamma (dist_fam.h t_lgamma, t_digamma): same
    fused log_prob / log_prob_sum / site-entry routes as the others; ``rsample`` stays torch's
    (``_standard_gamma`` rejection sampler with its implicit reparameterisation gradient)."""

    def _value(self, value):
        return value

    def log_prob(self, value):
        if self._validate_args:
            self._validate_sample(value)
        p0, p1 = self._params()
        return fused.log_prob(self._dist_id, self._value(value), p0, p1)

    def fused_log_prob_sum(self, value, scale=1.0, mask=None):
        return super().fused_log_prob_sum(self._value(value), scale, mask)

    def fused_site_entry(self, value, scale=1.0, mask=None):
        return super().fused_site_entry(self._value(value), scale, mask)


def _native_draws(t):
    """Draw through the HIP Philox sampler?  (device tensors of a supported dtype; when a test has
    replaced the normal source, torch's samplers keep their own generator semantics)"""
    from .. import kernels, rng
    return (isinstance(t, torch.Tensor) and kernels.on_device(t) and rng.normal is rng._default_normal
            and t.dtype in (torch.float32, torch.float64) and NATIVE_GAMMA["on"])


NATIVE_GAMMA = {"on": True}


class Gamma(_GammaFunctionFamily, torch.distributions.Gamma, TorchDistributionMixin):
    _dist_id = _lib.DIST_GAMMA

    def __init__(self, concentration, rate, validate_args=None):
        concentration, rate = _on_device(concentration, rate)
        super().__init__(concentration, rate, validate_args=validate_args)

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        new = type(self)(self.concentration.expand(batch_shape), self.rate.expand(batch_shape),
                         validate_args=False)
        new._validate_args = self._validate_args
        new._base_params = getattr(self, "_base_params", None) or self._params()
        return new

    def _params(self):
        return self.concentration, self.rate

    def rsample(self, sample_shape=torch.Size()):
        # torch: _standard_gamma(concentration.expand(shape)) / rate.expand(shape), clamped away
        # from zero (gamma.py:80-88); the draw and its implicit gradient come from one HIP launch
        if not _native_draws(self.concentration):
            return super().rsample(sample_shape)
        shape = self._extended_shape(sample_shape)
        value = fused.standard_gamma(self.concentration, shape) / self.rate.expand(shape)
        value.detach().clamp_(min=torch.finfo(value.dtype).tiny)
        return value


class Beta(_GammaFunctionFamily, torch.distributions.Beta, TorchDistributionMixin):
    _dist_id = _lib.DIST_BETA

    def __init__(self, concentration1, concentration0, validate_args=None):
        concentration1, concentration0 = _on_device(concentration1, concentration0)
        if not isinstance(concentration1, torch.Tensor):       # Beta(1.0, 1.0): all Python numbers
            concentration1, concentration0 = torch.distributions.utils.broadcast_all(
                concentration1, concentration0)
        super().__init__(concentration1, concentration0, validate_args=validate_args)
        # the operands as given: torch keeps only their stack (the Dirichlet it samples from), and
        # reading them back out of it would put a select + stack-backward pair on every gradient
        self._given = (concentration1, concentration0)

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        new = type(self)(self._given[0].expand(batch_shape), self._given[1].expand(batch_shape),
                         validate_args=False)
        new._validate_args = self._validate_args
        new._base_params = getattr(self, "_base_params", None) or self._given
        return new

    def _params(self):
        # stride-0 views of the given operands on the batch shape (log_prob has the batch shape)
        return tuple(p.expand(self.batch_shape) for p in self._given)

    def rsample(self, sample_shape=torch.Size()):
        # X = Ga / (Ga + Gb) with independent Gamma(concentration1), Gamma(concentration0) draws:
        # the pathwise gradient flows through the two draws' implicit gradients (an unbiased
        # reparameterisation gradient; torch's Beta draws through Dirichlet._dirichlet_grad, a
        # different -- equally unbiased -- estimator of the same derivative)
        c1, c0 = self._given
        if not (isinstance(c1, torch.Tensor) and _native_draws(c1)):
            return super().rsample(sample_shape)
        shape = self._extended_shape(sample_shape)
        ga = fused.standard_gamma(c1, shape)
        gb = fused.standard_gamma(c0 if isinstance(c0, torch.Tensor) else c1.new_tensor(float(c0)), shape)
        value = ga / (ga + gb)
        eps = torch.finfo(value.dtype).eps
        return value.clamp(min=eps, max=1 - eps)


class Dirichlet(torch.distributions.Dirichlet, TorchDistributionMixin):
    """torch's Dirichlet (constructor, ``rsample`` through ``_Dirichlet`` with its implicit
    reparameterisation gradient, ``expand``) with ``log_prob`` in one HIP launch each way
    (csrc/dirichlet.hip) on device tensors."""

    def log_prob(self, value):
        if self._validate_args:
            self._validate_sample(value)
        conc = self.concentration
        if value.is_cuda and conc.is_cuda and value.dtype == conc.dtype \
                and value.dtype in (torch.float32, torch.float64) and value.shape[-1] == conc.shape[-1]:
            base = getattr(self, "_base_concentration", None)
            return fused.dirichlet_log_prob(value, conc if base is None else base)
        return super().log_prob(value)

    def rsample(self, sample_shape=torch.Size()):
        # x = g / sum(g), g_k ~ Gamma(concentration_k): pathwise gradient through the draws' implicit
        # gradients (see Beta.rsample)
        if not _native_draws(self.concentration):
            return super().rsample(sample_shape)
        shape = self._extended_shape(sample_shape)
        g = fused.standard_gamma(self.concentration, shape)
        value = g / g.sum(-1, keepdim=True)
        tiny = torch.finfo(value.dtype).tiny
        return value.clamp(min=tiny)

    def expand(self, batch_shape, _instance=None):
        new = super().expand(batch_shape, _instance)
        # the concentration as given: a shared vector is read with row stride 0 instead of through
        # the expanded view (whose gradient would be reduced by a separate node)
        new._base_concentration = getattr(self, "_base_concentration", None)
        if new._base_concentration is None and self.concentration.dim() == 1:
            new._base_concentration = self.concentration
        return new


class _CountFamily(_GammaFunctionFamily):
    def _value(self, value):
        p0 = self._params()[0]
        return value if value.dtype == p0.dtype else value.to(p0.dtype)


class Poisson(_CountFamily, torch.distributions.Poisson, TorchDistributionMixin):
    _dist_id = _lib.DIST_POISSON

    def _params(self):
        return self.rate, None


class Binomial(_CountFamily, torch.distributions.Binomial, TorchDistributionMixin):
    """log_prob restates the reference's override (pyro/distributions/torch.py:83-101) with
    ``approx_log_prob_tol = 0``, its default."""
    _dist_id = _lib.DIST_BINOMIAL_LOGITS

    def _params(self):
        logits, n = self.logits, self.total_count
        return logits, (n if n.dtype == logits.dtype else n.to(logits.dtype))


class LinearLogits:
    """Lazy ``logits = (w @ X^T).squeeze(-2) + b`` of a plated GLM (never materialised).

    X: [N, D] data.  w: [D] or [..., 1, D] (the singleton is the data-plate dim of a
    vectorised-particle latent, pyro/infer/elbo.py:186-203).  b: scalar / [..., 1] / None.
    ``Bernoulli(logits=LinearLogits(X, w, b))`` evaluates log-likelihood AND gradient with the
    fused one-pass kernel instead of materialising the [P, N] logits.
    """

    def __init__(self, X, w, b=None):
        # the site values themselves, not the latent-tensor aliases model code holds (ops/lazy.py):
        # the ELBO assembly chains gradients by tensor identity
        w = _lazy._plain(w)
        b = _lazy._plain(b) if isinstance(b, torch.Tensor) else b
        if X.dim() != 2:
            raise ValueError("LinearLogits: X must be [N, D], got {}".format(tuple(X.shape)))
        if w.shape[-1] != X.shape[-1]:
            raise ValueError("LinearLogits: w[..., D] does not match X[N, D]: {} vs {}".format(
                tuple(w.shape), tuple(X.shape)))
        if w.dim() > 1 and w.shape[-2] != 1:
            raise ValueError("LinearLogits: expected w of shape [..., 1, D] (data-plate dim must "
                             "be a singleton), got {}".format(tuple(w.shape)))
        self.X, self.w, self.b = X, w, b
        lead = w.shape[:-2] if w.dim() > 1 else torch.Size()
        if b is not None and b.dim() > 0:
            if b.shape[-1] != 1:
                raise ValueError("LinearLogits: expected b of shape [..., 1], got {}".format(
                    tuple(b.shape)))
            lead = torch.broadcast_shapes(lead, b.shape[:-1])
        self.shape = torch.Size(lead) + (X.shape[0],)
        self.dtype, self.device = X.dtype, X.device

    def dim(self):
        return len(self.shape)

    def materialize(self):
        w = self.w
        out = (w @ self.X.t())
        if w.dim() > 1:
            out = out.squeeze(-2)
        if self.b is not None:
            out = out + self.b
        return out

    def flat_params(self):
        """(w2 [P, D], b1 [P] or None) with the particle dims flattened."""
        lead = self.shape[:-1]
        P = 1
        for s in lead:
            P *= s
        D = self.X.shape[1]
        w2 = self.w.expand(lead + (1, D)).reshape(P, D) if self.w.dim() > 1 else \
            self.w.expand(lead + (D,)).reshape(P, D)
        b1 = None
        if self.b is not None:
            b = self.b if self.b.dim() > 0 else self.b.reshape(1)
            b1 = b.expand(lead + (1,)).reshape(P)
        return w2.contiguous(), (b1.contiguous() if b1 is not None else None)


def linear_logits(X, w, b=None):
    return LinearLogits(X, w, b)


class GroupedLinearLogits:
    """Lazy ``logits[..., n] = w[..., g(n), :] . X[n] + b`` of a hierarchical GLM whose rows are
    SORTED BY GROUP (BASELINE config 5).  X: [N, D]; w: [..., G, D] (the plate dim of the
    ``groups`` plate is the G axis, particle dims lead); b: scalar / [..., 1] / None;
    ``segments``: kernels.GroupSegments built once from the group offsets."""

    def __init__(self, X, w, b, segments):
        w = _lazy._plain(w)
        b = _lazy._plain(b) if isinstance(b, torch.Tensor) else b
        if X.dim() != 2 or w.dim() < 2 or w.shape[-1] != X.shape[-1] or w.shape[-2] != segments.G:
            raise ValueError("GroupedLinearLogits: expected X [N, D], w [..., G={}, D], got {} and {}"
                             .format(segments.G, tuple(X.shape), tuple(w.shape)))
        if segments.N != X.shape[0]:
            raise ValueError("GroupedLinearLogits: group offsets cover {} rows, X has {}".format(
                segments.N, X.shape[0]))
        self.X, self.w, self.b, self.segments = X, w, b, segments
        lead = w.shape[:-2]
        if b is not None and b.dim() > 0:
            if b.shape[-1] != 1:
                raise ValueError("GroupedLinearLogits: expected b of shape [..., 1]")
            lead = torch.broadcast_shapes(lead, b.shape[:-1])
        self.shape = torch.Size(lead) + (X.shape[0],)
        self.dtype, self.device = X.dtype, X.device

    def dim(self):
        return len(self.shape)

    def group_of_row(self):
        import numpy as np
        if self.segments.ids is not None:            # rows in their original order
            return self.segments.ids
        off = self.segments.group_offsets
        return torch.as_tensor(np.repeat(np.arange(len(off) - 1), np.diff(off)), device=self.X.device)

    def materialize(self):
        wg = self.w[..., self.group_of_row(), :]                   # [..., N, D]
        out = (wg * self.X).sum(-1)
        if self.b is not None:
            out = out + self.b
        return out

    def flat_params(self):
        lead = self.shape[:-1]
        P = 1
        for s in lead:
            P *= s
        G, D = self.w.shape[-2:]
        w3 = self.w.expand(lead + (G, D)).reshape(P, G, D).contiguous()
        b1 = None
        if self.b is not None:
            b = self.b if self.b.dim() > 0 else self.b.reshape(1)
            b1 = b.expand(lead + (1,)).reshape(P).contiguous()
        return w3, b1


def grouped_linear_logits(X, w, b, segments):
    return GroupedLinearLogits(X, w, b, segments)


class _BernoulliLinear(TorchDistribution):
    """Bernoulli whose logits are a lazy LinearLogits: the plated GLM observed site."""

    arg_constraints = {}
    support = constraints.boolean
    has_enumerate_support = False
    _allow_f64 = False  # the HIP GLM kernel is float32; other dtypes take the unfused route

    def __init__(self, lazy, validate_args=None):
        self.lazy = lazy
        super().__init__(lazy.shape, torch.Size(), validate_args=False)

    @property
    def logits(self):
        return self.lazy.materialize()

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        if batch_shape == self.batch_shape:
            return self
        # a wider batch than the lazy form can express: materialise (unfused path)
        return Bernoulli(logits=self.lazy.materialize()).expand(batch_shape)

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return Bernoulli(logits=self.lazy.materialize()).sample(sample_shape)

    def log_prob(self, value):
        return fused.log_prob(_lib.DIST_BERNOULLI_LOGITS, value, self.lazy.materialize(), None)

    def fused_log_prob_sum(self, value, scale=1.0, mask=None):
        ll = self.fused_log_prob_batch(value, scale, mask)
        return None if ll is None else ll.sum()

    def fused_linear_term(self, value, scale=1.0, mask=None):
        """(ll[lead shape], [(w, d ll.sum()/d w), (b, d ll.sum()/d b)]) from ONE pass over X, with
        no autograd node: fused.SiteBatch carries the two gradients in its own backward launch."""
        lz = self.lazy
        from ..ops import torch_library
        if torch_library._routing_now():
            return None      # a tracer is recording: the site must be a graph node (fused_log_prob_batch)
        args = self._glm_args(value, scale, mask)
        if args is None or not isinstance(lz.w, torch.Tensor):
            return None
        # the batch carries gradients of small tensors only; decide BEFORE running the kernel
        if lz.w.requires_grad and lz.w.numel() > _lib.MULTI_MAX_ELEMS:
            return None
        w2, b1, mask = args
        with torch.no_grad():
            if isinstance(lz, GroupedLinearLogits):
                from .. import kernels
                ll, gw, gb = kernels.glm_bernoulli_grouped_fwd_bwd(lz.X, value.contiguous(), w2, b1,
                                                                   mask, float(scale), lz.segments)
            else:
                from .. import kernels
                ll, gw, gb = kernels.glm_bernoulli_fwd_bwd(lz.X, value.contiguous(), w2, b1, mask,
                                                           float(scale))
        targets = []
        if lz.w.numel() == gw.numel():
            targets.append((lz.w, gw.reshape(lz.w.shape)))
        elif lz.w.requires_grad:
            return None            # w was broadcast over particles: take the autograd route
        if lz.b is not None and isinstance(lz.b, torch.Tensor):
            if lz.b.numel() == gb.numel():
                targets.append((lz.b, gb.reshape(lz.b.shape)))
            elif lz.b.requires_grad:
                return None
        return ll.reshape(lz.shape[:-1]), targets

    def _glm_args(self, value, scale, mask):
        lz = self.lazy
        N = lz.X.shape[0]
        if isinstance(scale, torch.Tensor) or not _maskable(mask):
            return None
        if value.dim() != 1 or value.shape[0] != N:
            return None
        if lz.X.dtype != torch.float32 and not self._allow_f64:
            return None
        if mask is not None:
            if mask.dim() != 1 or mask.shape[0] != N:
                return None
            mask = mask.contiguous()
        if lz.X.shape[1] > 128 or not lz.X.is_contiguous():
            return None
        if isinstance(lz, GroupedLinearLogits):
            from .. import kernels
            if not kernels.glm_grouped_rows_servable(lz.X, value, mask, lz.segments):
                return None
        w2, b1 = lz.flat_params()
        return w2, b1, mask

    def fused_log_prob_batch(self, value, scale=1.0, mask=None):
        """Per-particle / per-chain sums over the data plate: a tensor of the lazy logits'
        leading shape (one fused GLM pass for all of them)."""
        lz = self.lazy
        N = lz.X.shape[0]
        if isinstance(scale, torch.Tensor) or not _maskable(mask):
            return None
        if value.dim() != 1 or value.shape[0] != N:
            return None
        if lz.X.dtype != torch.float32 and not self._allow_f64:
            return None
        if mask is not None:
            if mask.dim() != 1 or mask.shape[0] != N:
                return None
            mask = mask.contiguous()
        if lz.X.shape[1] > 128 or not lz.X.is_contiguous():
            return None
        if isinstance(lz, GroupedLinearLogits):
            from .. import kernels
            if not kernels.glm_grouped_rows_servable(lz.X, value, mask, lz.segments):
                return None
        w2, b1 = lz.flat_params()
        if isinstance(lz, GroupedLinearLogits):
            ll = fused.glm_bernoulli_grouped_ll(lz.X, value.contiguous(), w2, b1, mask, scale,
                                                lz.segments)
        else:
            ll = fused.glm_bernoulli_ll(lz.X, value.contiguous(), w2, b1, mask, scale)
        return ll.reshape(lz.shape[:-1])


class Bernoulli(_FusedElementwise, torch.distributions.Bernoulli, TorchDistributionMixin):
    """Bernoulli; with ``logits=`` given, log_prob runs the fused -BCE-with-logits kernel
    (torch: torch/distributions/bernoulli.py:121-125)."""

    _dist_id = _lib.DIST_BERNOULLI_LOGITS

    def __new__(cls, probs=None, logits=None, validate_args=None):
        if isinstance(logits, _lazy.DeferredMatmul):
            # unmodified model text (w @ X.t() ... + b): the plated GLM if its shape says so;
            # otherwise torch's constructor evaluates the product (broadcast_all is a torch function)
            logits = logits.as_linear_logits() or logits
        elif isinstance(logits, _lazy.DeferredGroupDot):
            # the hierarchical GLM as the reference writes it: (w[..., g, :] * X).sum(-1) + b
            logits = logits.as_grouped_linear_logits() or logits.materialize()
        if isinstance(logits, (LinearLogits, GroupedLinearLogits)):
            return _BernoulliLinear(logits, validate_args)
        return super().__new__(cls)

    def _params(self):
        return self.logits, None

    def enumerate_support(self, expand=True):
        return super().enumerate_support(expand)

    @property
    def has_enumerate_support(self):
        return True
