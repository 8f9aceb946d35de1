"""ELBO base class: the plugin seam SVI consumes (reference: pyro/infer/elbo.py:28-237).

Same constructor arguments and the same three entry points (loss, loss_and_grads,
differentiable_loss); ``vectorize_particles`` wraps model and guide in an outermost
plate("num_particles_vectorized", P, dim=-max_plate_nesting) exactly like the reference
(elbo.py:186-216) so the particle batch becomes the leading tensor dim the kernels stream over.
"""
import warnings
from abc import ABCMeta, abstractmethod

import torch

from .. import poutine
from ..primitives import plate


class ELBOModule(torch.nn.Module):
    def __init__(self, model, guide, elbo):
        super().__init__()
        self.model, self.guide, self.elbo = model, guide, elbo

    def forward(self, *args, **kwargs):
        return self.elbo.differentiable_loss(self.model, self.guide, *args, **kwargs)


class ELBO(metaclass=ABCMeta):
    def __init__(self, num_particles=1, max_plate_nesting=float("inf"), max_iarange_nesting=None,
                 vectorize_particles=False, strict_enumeration_warning=True,
                 ignore_jit_warnings=False, jit_options=None, retain_graph=None,
                 tail_adaptive_beta=-1.0):
        if max_iarange_nesting is not None:
            warnings.warn("max_iarange_nesting is deprecated; use max_plate_nesting",
                          DeprecationWarning)
            max_plate_nesting = max_iarange_nesting
        self.max_plate_nesting = max_plate_nesting
        self.num_particles = num_particles
        self.vectorize_particles = vectorize_particles
        self.retain_graph = retain_graph
        if self.vectorize_particles and self.num_particles > 1:
            self.max_plate_nesting += 1
        self.strict_enumeration_warning = strict_enumeration_warning
        self.ignore_jit_warnings = ignore_jit_warnings
        self.jit_options = jit_options
        self.tail_adaptive_beta = tail_adaptive_beta

    def __call__(self, model, guide):
        return ELBOModule(model, guide, self)

    def _guess_max_plate_nesting(self, model, guide, args, kwargs):
        """Run model and guide once to find the deepest vectorised plate."""
        with poutine.block():
            guide_trace = poutine.trace(guide).get_trace(*args, **kwargs)
            model_trace = poutine.trace(poutine.replay(model, trace=guide_trace)).get_trace(
                *args, **kwargs)
        from ..poutine.util import prune_subsample_sites
        from ..util import check_site_shape
        from .util import is_validation_enabled
        guide_trace = prune_subsample_sites(guide_trace)
        model_trace = prune_subsample_sites(model_trace)
        sites = [site for trace in (model_trace, guide_trace) for site in trace.nodes.values()
                 if site["type"] == "sample"]
        # shapes are checked now, against the un-enumerated run: once max_plate_nesting is finite,
        # whatever sits left of it is allowed to broadcast (elbo.py:160-168)
        if is_validation_enabled():
            for site in sites:
                check_site_shape(site, max_plate_nesting=float("inf"))
        dims = [frame.dim for site in sites for frame in site["cond_indep_stack"] if frame.vectorized]
        self.max_plate_nesting = -min(dims) if dims else 0
        if self.vectorize_particles and self.num_particles > 1:
            self.max_plate_nesting += 1

    def _vectorized_num_particles(self, fn):
        def wrapped_fn(*args, **kwargs):
            if self.num_particles == 1:
                return fn(*args, **kwargs)
            with plate("num_particles_vectorized", self.num_particles,
                       dim=-self.max_plate_nesting):
                return fn(*args, **kwargs)

        return wrapped_fn

    def _get_vectorized_trace(self, model, guide, args, kwargs):
        return self._get_trace(self._vectorized_num_particles(model),
                               self._vectorized_num_particles(guide), args, kwargs)

    @abstractmethod
    def _get_trace(self, model, guide, args, kwargs):
        raise NotImplementedError

    def _get_traces(self, model, guide, args, kwargs):
        if self.vectorize_particles:
            if self.max_plate_nesting == float("inf"):
                self._guess_max_plate_nesting(model, guide, args, kwargs)
            yield self._get_vectorized_trace(model, guide, args, kwargs)
        else:
            for _ in range(self.num_particles):
                yield self._get_trace(model, guide, args, kwargs)
