"""Distribution layer: torch.distributions classes with the Pyro mixin interface
(reference: pyro/distributions/torch.py:395-419 wraps every torch.distributions class the
same way), plus HIP-fused overrides for the hot exponential-family sites.
"""
import torch
from torch.distributions import constraints, transforms  # noqa: F401
from torch.distributions import biject_to, transform_to, kl_divergence  # noqa: F401

from .base import (Delta, ExpandedDistribution, MaskedDistribution, ScoreParts, TorchDistribution,  # noqa: F401
                   TorchDistributionMixin, Unit)
from .families import (Bernoulli, Beta, Binomial, Dirichlet, Exponential, Gamma,  # noqa: F401
                       GroupedLinearLogits, HalfCauchy, HalfNormal, LinearLogits, LogNormal, Normal,
                       Poisson, grouped_linear_logits, linear_logits)
from .util import enable_validation, is_validation_enabled  # noqa: F401
from . import kl as _kl  # noqa: F401,E402  (registers the reference's extra kl_divergence pairs)

# ---- everything else: torch.distributions + mixin, arithmetic by ATen on the GPU ----------------
_FUSED = {"Normal", "Bernoulli", "HalfCauchy", "HalfNormal", "LogNormal", "Exponential",
          "Gamma", "Beta", "Poisson", "Binomial", "Dirichlet"}
__all__ = ["Delta", "Unit", "MaskedDistribution", "TorchDistribution", "ScoreParts",
           "LinearLogits", "linear_logits", "GroupedLinearLogits", "grouped_linear_logits"] + sorted(_FUSED)


def _wrap_all():
    import torch.distributions as td
    g = globals()
    for name in td.__all__:
        cls = getattr(td, name, None)
        if not isinstance(cls, type) or not issubclass(cls, td.Distribution):
            continue
        if cls is td.Distribution or name in g:
            continue
        g[name] = type(name, (cls, TorchDistributionMixin), {
            "__doc__": "torch.distributions.%s with the Pyro mixin interface." % name,
            "__module__": __name__})
        __all__.append(name)


_wrap_all()


# pyro-style overrides that matter for the enumeration path (reference: torch.py:124-149)
_SUPPORTS = {}


class Categorical(torch.distributions.Categorical, TorchDistributionMixin):
    def expand(self, batch_shape, _instance=None):
        """Expansion keeps the log-probability table a stride-0 VIEW of the un-expanded one.
        (torch expands ``probs`` first and materialises ``logits`` [batch.., V] from the expanded
        tensor on first use: for examples/lda.py that is T x words x docs x V elements.)"""
        new = self._get_checked_instance(Categorical, _instance)
        batch_shape = torch.Size(batch_shape)
        param_shape = batch_shape + torch.Size((self._num_events,))
        new.logits = self.logits.expand(param_shape)       # computed once on the small table
        # the un-expanded table: consumers that only need the table itself (the fused enumeration
        # kernels) must not index the expanded view -- the autograd dual of select/expand
        # materialises a zero tensor of the EXPANDED shape (T x words x docs x V elements)
        new._base_logits = getattr(self, "_base_logits", self.logits)
        if "probs" in self.__dict__:
            new.probs = self.probs.expand(param_shape)
        new._param = new.logits
        new._num_events = self._num_events
        super(torch.distributions.Categorical, new).__init__(batch_shape, validate_args=False)
        new._validate_args = self._validate_args
        return new

    def enumerate_support(self, expand=True):
        # (torch/distributions/categorical.py:149-156: arange(num_events) on a fresh leftmost dim; the arange
        # itself is kept per (size, device) -- a step of a pyro.markov loop would launch one per time step)
        n, dev = self._num_events, self._param.device
        values = _SUPPORTS.get((n, dev))
        if values is None:
            values = torch.arange(n, dtype=torch.long, device=dev)
            if not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
                _SUPPORTS[(n, dev)] = values
        result = values.view((-1,) + (1,) * len(self._batch_shape))
        if expand:
            result = result.expand((-1,) + self._batch_shape)
        else:
            result._pyro_categorical_support = id(self)
        return result

    def log_prob(self, value):
        if getattr(value, "_pyro_categorical_support", None) == id(self):
            # value is the un-expanded support arange(T) on a fresh leftmost dim: reshape, no gather
            if not torch._C._get_tracing_state():
                if self._validate_args:
                    self._validate_sample(value)
                assert value.size(0) == self.logits.size(-1)
            logits = self.logits
            if logits.dim() <= value.dim():
                logits = logits.reshape((1,) * (1 + value.dim() - logits.dim()) + logits.shape)
            if not torch._C._get_tracing_state():
                assert logits.size(-1 - value.dim()) == 1
            return logits.transpose(-1 - value.dim(), -1).squeeze(-1)
        return super().log_prob(value)


class Independent(torch.distributions.Independent, TorchDistributionMixin):
    @property
    def _validate_args(self):
        return self.base_dist._validate_args

    @_validate_args.setter
    def _validate_args(self, value):
        self.base_dist._validate_args = value

    @property
    def has_enumerate_support(self):
        return False

    def fused_log_prob_sum(self, value, scale=1.0, mask=None):
        # the plate/event sum of an Independent is the plain sum of its base log_prob
        if isinstance(mask, torch.Tensor):
            mask = mask.reshape(mask.shape + (1,) * self.reinterpreted_batch_ndims)
        f = getattr(self.base_dist, "fused_log_prob_sum", None)
        return None if f is None else f(value, scale, mask)

    def fused_site_entry(self, value, scale=1.0, mask=None):
        if isinstance(mask, torch.Tensor):
            mask = mask.reshape(mask.shape + (1,) * self.reinterpreted_batch_ndims)
        f = getattr(self.base_dist, "fused_site_entry", None)
        return None if f is None else f(value, scale, mask)

    def fused_score_term(self, value, scale=1.0, mask=None):
        f = getattr(self.base_dist, "fused_score_term", None) if mask is None else None
        return None if f is None else f(value, scale, mask)
from .hmm import DiscreteHMM  # noqa: E402,F401
