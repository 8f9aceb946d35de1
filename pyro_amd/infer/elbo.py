"""The estimator base class SVI consumes (constructor arguments and entry points of pyro.infer.ELBO, so that
``Trace_ELBO(num_particles=64, vectorize_particles=True)`` means here what it means there).

What the base class owns is the PARTICLE dimension: with ``vectorize_particles`` model and guide run once
inside an outermost ``plate("num_particles_vectorized", P)`` placed just left of the model's own plates, which
is what turns the particle batch into the leading tensor dim the kernels stream over; otherwise the pair is
traced ``num_particles`` times.  Subclasses provide ``_get_trace`` (one traced pair) and the three entry
points ``loss`` / ``loss_and_grads`` / ``differentiable_loss``.
"""
import abc
import math
import warnings

import torch

from .. import poutine
from ..poutine.util import prune_subsample_sites
from ..primitives import plate

_PARTICLE_PLATE = "num_particles_vectorized"


class ELBOModule(torch.nn.Module):
    """``elbo(model, guide)``: a module whose forward is the differentiable loss (for torch optimisers)."""

    def __init__(self, model, guide, elbo):
        super().__init__()
        self.model, self.guide, self.elbo = model, guide, elbo

    def forward(self, *args, **kwargs):
        return self.elbo.differentiable_loss(self.model, self.guide, *args, **kwargs)


def deepest_plate(model, guide, args, kwargs, validate):
    """How many vectorised plate dims the pair uses, found by one un-enumerated run of guide and model.
    With ``validate`` the log_prob shapes of that run are checked against the plates now -- later, with a
    finite ``max_plate_nesting``, whatever lies left of it may broadcast freely."""
    from ..util import check_site_shape
    with poutine.block():
        from ..ops import lazy
        with lazy.watch_histograms():
            guide_trace = poutine.trace(guide).get_trace(*args, **kwargs)
        model_trace = poutine.trace(poutine.replay(model, trace=guide_trace)).get_trace(*args, **kwargs)
    depth = 0
    for trace in (prune_subsample_sites(model_trace), prune_subsample_sites(guide_trace)):
        for site in trace.nodes.values():
            if site["type"] != "sample":
                continue
            if validate:
                check_site_shape(site, max_plate_nesting=math.inf)
            depth = max([depth] + [-f.dim for f in site["cond_indep_stack"] if f.vectorized])
    return depth


class ELBO(abc.ABC):
    def __init__(self, num_particles=1, max_plate_nesting=math.inf, max_iarange_nesting=None,
                 vectorize_particles=False, strict_enumeration_warning=True, ignore_jit_warnings=False,
                 jit_options=None, retain_graph=None, tail_adaptive_beta=-1.0):
        if max_iarange_nesting is not None:
            warnings.warn("max_iarange_nesting is deprecated; use max_plate_nesting instead",
                          DeprecationWarning)
            max_plate_nesting = max_iarange_nesting
        self.num_particles, self.vectorize_particles = num_particles, vectorize_particles
        self.max_plate_nesting = max_plate_nesting + self._particle_dims
        self.strict_enumeration_warning = strict_enumeration_warning
        self.retain_graph = retain_graph
        # accepted for signature compatibility; there is no tracing compiler to configure
        self.ignore_jit_warnings, self.jit_options = ignore_jit_warnings, jit_options
        self.tail_adaptive_beta = tail_adaptive_beta

    def __call__(self, model, guide):
        return ELBOModule(model, guide, self)

    # ---- the particle dimension ------------------------------------------------------------------------------
    @property
    def _particle_dims(self):
        return int(bool(self.vectorize_particles) and self.num_particles > 1)

    def _guess_max_plate_nesting(self, model, guide, args, kwargs):
        from .util import is_validation_enabled
        self.max_plate_nesting = deepest_plate(model, guide, args, kwargs, is_validation_enabled()) \
            + self._particle_dims

    def _vectorized_num_particles(self, fn):
        """``fn`` run inside the particle plate (as it is for a single particle)."""
        if self.num_particles == 1:
            return fn

        def in_particle_plate(*args, **kwargs):
            with plate(_PARTICLE_PLATE, self.num_particles, dim=-self.max_plate_nesting):
                return fn(*args, **kwargs)

        return in_particle_plate

    def _get_vectorized_trace(self, model, guide, args, kwargs):
        wrap = self._vectorized_num_particles
        return self._get_trace(wrap(model), wrap(guide), args, kwargs)

    def _get_traces(self, model, guide, args, kwargs):
        """The traced (model, guide) pairs one evaluation averages over."""
        if not self.vectorize_particles:
            for _ in range(self.num_particles):
                yield self._get_trace(model, guide, args, kwargs)
            return
        if self.max_plate_nesting == math.inf:
            self._guess_max_plate_nesting(model, guide, args, kwargs)
        yield self._get_vectorized_trace(model, guide, args, kwargs)

    @abc.abstractmethod
    def _get_trace(self, model, guide, args, kwargs):
        """One (model_trace, guide_trace) pair."""
