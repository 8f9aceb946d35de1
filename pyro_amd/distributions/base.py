"""Mixin that gives torch.distributions classes the interface Pyro's handlers call
(reference seam 1: pyro/distributions/torch_distribution.py:31-299, distribution.py:98-125,
score_parts.py:21-38), plus Delta / Unit / MaskedDistribution.

Fused protocol (new in this backend): a distribution may define

    fused_log_prob_sum(value, scale, mask) -> 0-dim tensor

returning ``scale_and_mask(log_prob(value), scale, mask).sum()`` from ONE HIP kernel;
``Trace.compute_log_prob`` uses it when nobody needs the un-reduced log_prob tensor.
"""
from collections import namedtuple

import torch
from torch.distributions import constraints

from .util import broadcast_shape, is_identically_zero, scale_and_mask, sum_rightmost


class ScoreParts(namedtuple("ScoreParts", ["log_prob", "score_function", "entropy_term"])):
    """(log_prob, score_function, entropy_term) used by the ELBO surrogate."""

    def scale_and_mask(self, scale=1.0, mask=None):
        # score_function is a multiplier, it is masked but never scaled
        log_prob = scale_and_mask(self.log_prob, scale, mask)
        score_function = self.score_function
        if mask is not None and not is_identically_zero(score_function):
            score_function = scale_and_mask(score_function, 1.0, mask)
        entropy_term = scale_and_mask(self.entropy_term, scale, mask)
        return ScoreParts(log_prob, score_function, entropy_term)


class TorchDistributionMixin:
    has_rsample = False
    has_enumerate_support = False

    def __call__(self, sample_shape=torch.Size()):
        # rsample when reparameterised, else sample (torch_distribution.py:31-52)
        return self.rsample(sample_shape) if self.has_rsample else self.sample(sample_shape)

    @property
    def event_dim(self):
        return len(self.event_shape)

    def shape(self, sample_shape=torch.Size()):
        return torch.Size(sample_shape) + self.batch_shape + self.event_shape

    def score_parts(self, x, *args, **kwargs):
        log_prob = self.log_prob(x, *args, **kwargs)
        if self.has_rsample:
            return ScoreParts(log_prob=log_prob, score_function=0, entropy_term=log_prob)
        return ScoreParts(log_prob=log_prob, score_function=log_prob, entropy_term=0)

    def expand_by(self, sample_shape):
        return self.expand(torch.Size(sample_shape) + self.batch_shape)

    def has_rsample_(self, value):
        """Force reparameterised or detached sampling on this instance (distribution.py:180-199)."""
        if not (value is True or value is False):
            raise ValueError("Expected value in [False,True], actual {}".format(value))
        self.has_rsample = value
        return self

    def reshape(self, sample_shape=None, extra_event_dims=None):
        raise Exception(".reshape(sample_shape=s, extra_event_dims=n) was renamed: "
                        "use .expand_by(s).to_event(n)")

    def to_event(self, reinterpreted_batch_ndims=None):
        if reinterpreted_batch_ndims is None:
            reinterpreted_batch_ndims = len(self.batch_shape)
        if reinterpreted_batch_ndims == 0:
            return self
        from . import Independent
        base, n = self, reinterpreted_batch_ndims
        while isinstance(base, torch.distributions.Independent):
            n += base.reinterpreted_batch_ndims
            base = base.base_dist
        if n < 0:
            raise ValueError("cannot remove event dims that were never added")
        return Independent(base, n) if n else base

    def independent(self, reinterpreted_batch_ndims=None):
        return self.to_event(reinterpreted_batch_ndims)

    def mask(self, mask):
        return MaskedDistribution(self, mask)

    # ---- fused protocol default: peel wrappers, delegate to the base family -----------------
    def fused_log_prob_sum(self, value, scale=1.0, mask=None):
        return None


class TorchDistribution(torch.distributions.Distribution, TorchDistributionMixin):
    """Base class for distributions implemented directly in this package -- and for the user's own:
    a subclass that does not write ``expand`` still works inside plates, through the generic wrapper
    below (reference: torch_distribution.py:279-299)."""

    def expand(self, batch_shape, _instance=None):
        return ExpandedDistribution(self, batch_shape)


class ExpandedDistribution(TorchDistribution):
    """``base_dist`` seen with a larger batch shape: independent draws along every dim that was added or
    stretched from size 1, the base distribution's own log_prob broadcast over them
    (reference: torch_distribution.py:399-526)."""

    arg_constraints = {}

    def __init__(self, base_dist, batch_shape=torch.Size()):
        self.base_dist = base_dist
        super().__init__(base_dist.batch_shape, base_dist.event_shape, validate_args=False)
        self.expand(batch_shape)

    @staticmethod
    def _grown(old, new):
        """``old`` broadcast up to ``new`` (never down): the resulting shape, or ValueError."""
        old, new = tuple(old), tuple(new)
        if len(new) < len(old):
            raise ValueError("Cannot broadcast distribution of shape {} to shape {}".format(old, new))
        out = list(new[:len(new) - len(old)])
        for have, want in zip(old, new[len(new) - len(old):]):
            if have != want and have != 1:
                raise ValueError("Cannot broadcast distribution of shape {} to shape {}".format(old, new))
            out.append(want if have == 1 else have)
        return torch.Size(out)

    def expand(self, batch_shape, _instance=None):
        # in place, as the reference does: an expanded distribution only ever grows
        grown = self._grown(self.batch_shape, batch_shape)
        self._batch_shape = self._grown(self.base_dist.batch_shape, grown)
        return self

    has_rsample = property(lambda self: self.base_dist.has_rsample)
    has_enumerate_support = property(lambda self: self.base_dist.has_enumerate_support)

    @constraints.dependent_property
    def support(self):
        return self.base_dist.support

    def _draw(self, draw, sample_shape):
        sample_shape = torch.Size(sample_shape)
        base, full = tuple(self.base_dist.batch_shape), tuple(self.batch_shape)
        lead = full[:len(full) - len(base)]
        stretched = [(j, want) for j, (have, want) in enumerate(zip(base, full[len(lead):]))
                     if have == 1 and want != 1]
        x = draw(sample_shape + torch.Size(lead) + torch.Size([size for _, size in stretched]))
        # x: sample_shape + lead + stretched sizes + base batch + event; each stretched axis swaps
        # places with the size-1 axis it fills
        first = len(sample_shape) + len(lead)
        for i, (j, _) in enumerate(stretched):
            x = x.transpose(first + i, first + len(stretched) + j)
        return x.reshape(sample_shape + self.batch_shape + self.event_shape)

    def sample(self, sample_shape=torch.Size()):
        return self._draw(self.base_dist.sample, sample_shape)

    def rsample(self, sample_shape=torch.Size()):
        return self._draw(self.base_dist.rsample, sample_shape)

    def _value_batch(self, value):
        return broadcast_shape(self.batch_shape, value.shape[:value.dim() - self.event_dim])

    def log_prob(self, value):
        return self.base_dist.log_prob(value).expand(self._value_batch(value))

    def score_parts(self, value):
        shape = self._value_batch(value)
        parts = self.base_dist.score_parts(value)
        if self.batch_shape == self.base_dist.batch_shape:
            return parts
        return ScoreParts(*(p.expand(shape) if isinstance(p, torch.Tensor) else p for p in parts))

    def enumerate_support(self, expand=True):
        values = self.base_dist.enumerate_support(expand=False)
        values = values.reshape(values.shape[:1] + (1,) * len(self.batch_shape))
        return values.expand(values.shape[:1] + self.batch_shape) if expand else values

    @property
    def mean(self):
        return self.base_dist.mean.expand(self.batch_shape + self.event_shape)

    @property
    def variance(self):
        return self.base_dist.variance.expand(self.batch_shape + self.event_shape)

    def conjugate_update(self, other):
        updated, log_normalizer = self.base_dist.conjugate_update(other)
        return updated.expand(self.batch_shape), log_normalizer.expand(self.batch_shape)



def _of_base(name):
    return property(lambda self: getattr(self.base_dist, name))


def _density_dtype(value):
    """A log-density is floating point whatever the value's dtype (an integer-valued site switched off by
    mask=False would otherwise hand an int64 zero to logsumexp / einsum)."""
    return value.dtype if value.is_floating_point() else torch.get_default_dtype()


class MaskedDistribution(TorchDistribution):
    """``base_dist.mask(m)``: the density counts only where ``m`` holds (what the reference's class of
    the same name provides, pyro/distributions/torch_distribution.py:302-396).  Here the mask is not an
    extra pass over the log-density: it is the u8 ``mask`` ARGUMENT every site kernel already takes
    (include/pyro_amd.h pa_dist_log_prob_sum / pa_site_entry), so the class reduces to one rule --
    ``_gate`` -- that says what reaches the kernel: None (everything counts), False (nothing does) or
    a bool tensor, combined with the mask a ``poutine.mask`` handler brings."""

    arg_constraints = {}
    # everything that does not involve the density is the base distribution's
    has_rsample, has_enumerate_support, support = _of_base("has_rsample"), _of_base("has_enumerate_support"), _of_base("support")
    mean, variance = _of_base("mean"), _of_base("variance")

    def __init__(self, base_dist, mask):
        if not isinstance(mask, bool):
            shape = broadcast_shape(mask.shape, base_dist.batch_shape)
            mask = mask.bool().expand(shape)
            if base_dist.batch_shape != shape:
                base_dist = base_dist.expand(shape)
        self.base_dist, self._mask = base_dist, mask
        super().__init__(base_dist.batch_shape, base_dist.event_shape, validate_args=False)

    def _gate(self, other=None):
        mine = None if self._mask is True else self._mask
        if mine is None or other is None:
            return mine if other is None else other
        if mine is False or other is False:
            return False
        return mine & other

    def _nothing(self, value):
        """Zeros of the log-density's shape without evaluating the base distribution (a False mask is
        how models switch a site off on data that may lie outside its support)."""
        lead = value.shape[:value.dim() - self.event_dim]
        return value.new_zeros((), dtype=_density_dtype(value)).expand(
            broadcast_shape(self.base_dist.batch_shape, lead))

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        m = self._mask
        return MaskedDistribution(self.base_dist.expand(batch_shape),
                                  m if isinstance(m, bool) else m.expand(batch_shape))

    def sample(self, sample_shape=torch.Size()):
        return self.base_dist.sample(sample_shape)

    def rsample(self, sample_shape=torch.Size()):
        return self.base_dist.rsample(sample_shape)

    def enumerate_support(self, expand=True):
        return self.base_dist.enumerate_support(expand=expand)

    def log_prob(self, value):
        gate = self._gate()
        if gate is False:
            return self._nothing(value)
        lp = self.base_dist.log_prob(value)
        return lp if gate is None else scale_and_mask(lp, mask=gate)

    def score_parts(self, value):
        gate = self._gate()
        if gate is None or gate is False:
            return super().score_parts(value)          # built from self.log_prob: masked already
        return self.base_dist.score_parts(value).scale_and_mask(mask=gate)

    def conjugate_update(self, other):
        updated, log_normalizer = self.base_dist.conjugate_update(other)
        return updated.mask(self._mask), scale_and_mask(log_normalizer, mask=self._mask)

    # ---- the fused protocol: the gate goes to the kernel with the handler's mask ---------------------
    def _fused(self, method, value, scale, mask):
        f = getattr(self.base_dist, method, None)
        gate = self._gate(mask)
        if f is None or (gate is not None and gate is not False and self.event_dim != 0):
            return None        # a tensor mask over event dims is not a per-element kernel mask
        return f, gate

    def fused_log_prob_sum(self, value, scale=1.0, mask=None):
        if self._mask is False:
            return value.new_zeros((), dtype=_density_dtype(value))
        hit = self._fused("fused_log_prob_sum", value, scale, mask)
        return None if hit is None else hit[0](value, scale, hit[1])

    def fused_site_entry(self, value, scale=1.0, mask=None):
        if self._mask is False:
            return None
        hit = self._fused("fused_site_entry", value, scale, mask)
        return None if hit is None else hit[0](value, scale, hit[1])


class Delta(TorchDistribution):
    """Point mass at ``v`` with log-density ``log_density`` (reference: delta.py:73-77)."""

    has_rsample = True
    arg_constraints = {"v": constraints.dependent, "log_density": constraints.real}

    def __init__(self, v, log_density=0.0, event_dim=0, validate_args=None):
        if event_dim > v.dim():
            raise ValueError("Expected event_dim <= v.dim(), actual {} vs {}".format(
                event_dim, v.dim()))
        batch_dim = v.dim() - event_dim
        batch_shape, event_shape = v.shape[:batch_dim], v.shape[batch_dim:]
        # a python-number log-density of 0 (identity transforms of the autoguides) is remembered so
        # that scoring the site costs no kernel at all
        self._zero_density = isinstance(log_density, (int, float)) and log_density == 0
        if isinstance(log_density, (int, float)):
            from .families import device_constant   # cached 0-dim constant, expanded: no kernel
            log_density = device_constant(log_density, v.dtype, v.device).expand(batch_shape)
        elif log_density.shape != batch_shape:
            raise ValueError("Expected log_density.shape = {}, actual {}".format(
                log_density.shape, batch_shape))
        self.v, self.log_density = v, log_density
        super().__init__(batch_shape, event_shape, validate_args=validate_args)

    @constraints.dependent_property
    def support(self):
        return constraints.independent(constraints.real, len(self.event_shape))

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        v = self.v.expand(batch_shape + self.event_shape)
        new = Delta(v, self.log_density.expand(batch_shape), len(self.event_shape),
                    validate_args=False)
        new._zero_density = self._zero_density
        return new

    def rsample(self, sample_shape=torch.Size()):
        shape = torch.Size(sample_shape) + self.v.shape
        return self.v if shape == self.v.shape else self.v.expand(shape)

    def log_prob(self, x):
        if x is self.v:
            # the site's own draw: (x == v).log() is 0 (it is -inf only where v is NaN, and then the
            # model site scoring the same NaN value already makes the estimate NaN)
            return self.log_density
        v = self.v.expand(self.batch_shape + self.event_shape)
        log_prob = (x == v).type(x.dtype).log()
        log_prob = sum_rightmost(log_prob, len(self.event_shape))
        return log_prob + self.log_density

    def fused_log_prob_sum(self, value, scale=1.0, mask=None):
        if value is self.v and self._zero_density:
            return 0.0      # exactly zero whatever the scale / mask: no kernel, no tensor
        return None

    def fused_score_term(self, value, scale=1.0, mask=None):
        """The site's own draw, unscaled and unmasked: its score IS ``log_density``, handed over
        un-summed so that the ELBO's one batched reduction adds it up (no reduction of its own)."""
        if value is self.v and not self._zero_density and mask is None \
                and not isinstance(scale, torch.Tensor) and scale == 1.0 and isinstance(self.log_density, torch.Tensor) and self.log_density.numel() > 0:
            return self.log_density
        return None

    @property
    def mean(self):
        return self.v

    @property
    def variance(self):
        return torch.zeros_like(self.v)


class Unit(TorchDistribution):
    """Trivial distribution over the empty event, carrying a log_factor (pyro.factor)."""

    arg_constraints = {"log_factor": constraints.real}
    support = constraints.real

    def __init__(self, log_factor, *, has_rsample=None, validate_args=None):
        log_factor = torch.as_tensor(log_factor)
        self.log_factor = log_factor
        if has_rsample is not None:
            # an instance attribute only when the caller said so: a guide-side pyro.factor has to
            # (pyro/util.py:447-462 looks into __dict__)
            self.has_rsample = has_rsample
        super().__init__(log_factor.shape, torch.Size((0,)), validate_args=validate_args)

    def expand(self, batch_shape, _instance=None):
        return Unit(self.log_factor.expand(torch.Size(batch_shape)),
                    has_rsample=self.__dict__.get("has_rsample"))

    def sample(self, sample_shape=torch.Size()):
        return self.log_factor.new_empty(torch.Size(sample_shape) + self.shape())

    rsample = sample

    def log_prob(self, value):
        shape = broadcast_shape(self.batch_shape, value.shape[:-1])
        return self.log_factor.expand(shape)

