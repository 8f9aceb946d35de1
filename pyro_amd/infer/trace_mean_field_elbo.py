"""TraceMeanField_ELBO: analytic KL divergences where torch knows them
(reference: pyro/infer/trace_mean_field_elbo.py:21-156).

Same traces as Trace_ELBO; for every latent site that appears in both traces the sampled pair
``log p(z) - log q(z)`` is replaced by ``-KL(q || p)`` when ``torch.distributions.kl_divergence`` is
registered for the pair (e.g. Normal/Normal, Independent(Normal)/Independent(Normal)), and falls back
on the sampled terms otherwise (:131-137).  Observed sites, fall-back sites and auxiliary guide sites
go through the batched one-launch reduction of Trace_ELBO (distributions.fused.SiteBatch); the KL
terms (small tensors, torch arithmetic) ride in the same launch as already-computed terms.
"""
from .util import is_validation_enabled
import warnings

import torch

from ..distributions.fused import grad_sink as _grad_sink
from torch.distributions import kl_divergence

from .. import poutine
from ..distributions.util import scale_and_mask
from ..util import torch_item, warn_if_nan
from .trace_elbo import Trace_ELBO, _unit_grad


def _check_mean_field_requirement(model_trace, guide_trace):
    """Sufficient (not necessary) check: shared sample sites occur in the same order
    (trace_mean_field_elbo.py:21-47)."""
    model_sites = [name for name, site in model_trace.nodes.items()
                   if site["type"] == "sample" and name in guide_trace.nodes]
    guide_sites = [name for name, site in guide_trace.nodes.items()
                   if site["type"] == "sample" and name in model_trace.nodes]
    assert set(model_sites) == set(guide_sites)
    if model_sites != guide_sites:
        warnings.warn("Failed to verify mean field restriction on the guide. To eliminate this "
                      "warning, ensure model and guide sites occur in the same order.\n"
                      "Model sites:\n  " + "\n  ".join(model_sites) +
                      "Guide sites:\n  " + "\n  ".join(guide_sites))


def _normal_operands(fn):
    """(loc, scale, (event dims, family), shape) of a Normal / LogNormal, possibly under ``to_event`` -- the parameters as
    given before any expand (the site kernels broadcast by stride) -- or None."""
    from ..distributions import families
    event = 0
    while isinstance(fn, torch.distributions.Independent):
        event += fn.reinterpreted_batch_ndims
        fn = fn.base_dist
    if type(fn) not in (families.Normal, families.LogNormal):
        return None
    loc, scale = getattr(fn, "_base_params", None) or (fn.loc, fn.scale)
    # (a LogNormal pair has the KL of its base Normals: torch kl.py _kl_transformed_transformed)
    return loc, scale, (event, type(fn)), fn.batch_shape


def _add_normal_kl(batch, gsite, msite):
    """KL(q || p) of a Normal / Normal pair (same event dims) as two entries of the batch's launch
    instead of ~8 element-wise torch kernels and their autograd duals; False = not applicable."""
    q, p = _normal_operands(gsite["fn"]), _normal_operands(msite["fn"])
    if q is None or p is None or q[2] != p[2]:
        return False
    shape = q[3]
    try:
        if torch.broadcast_shapes(shape, p[3]) != shape:
            return False
    except RuntimeError:
        return False
    mask = gsite["mask"]
    if mask is not None:
        if not isinstance(mask, torch.Tensor):
            return False
        if q[2][0]:                       # the mask spans batch dims only
            mask = mask.reshape(mask.shape + (1,) * q[2][0])
    return batch.add_kl_normal(q[0], q[1], p[0], p[1], shape, mask, gsite["scale"], -1.0)


class TraceMeanField_ELBO(Trace_ELBO):
    def _get_trace(self, model, guide, args, kwargs):
        model_trace, guide_trace = super()._get_trace(model, guide, args, kwargs)
        if not getattr(guide_trace, "_fully_reparam", False):
            raise NotImplementedError("TraceMeanField_ELBO requires every guide site to be "
                                      "reparameterised (check_fully_reparametrized in the reference)")
        if is_validation_enabled():
            _check_mean_field_requirement(model_trace, guide_trace)
        return model_trace, guide_trace

    @staticmethod
    def _batched_total(model_trace, guide_trace, coef=1.0):
        """coef * ELBO particle of trace_mean_field_elbo.py:104-150 as a 0-dim tensor."""
        from ..distributions.fused import SiteBatch

        from ..distributions.base import Delta
        batch = SiteBatch()
        analytic, delta_sites, left = set(), set(), []
        for name, msite in model_trace.nodes.items():
            if msite["type"] != "sample" or msite["is_observed"] or name not in guide_trace.nodes:
                continue
            gsite = guide_trace.nodes[name]
            if _add_normal_kl(batch, gsite, msite):
                analytic.add(name)
                continue
            if type(gsite["fn"]) is Delta and not isinstance(gsite["scale"], torch.Tensor) \
                    and gsite["scale"] == msite["scale"] and gsite["mask"] is msite["mask"]:
                # kl_divergence(Delta(v), p) = -p.log_prob(v) (distributions/kl.py, the Delta's own
                # log_density does not enter): -KL is the model site's log-probability at the
                # replayed value -- it rides in the batch as an ordinary model entry, and the
                # guide's Delta contributes nothing
                delta_sites.add(name)
                continue
            try:
                kl_qp = kl_divergence(gsite["fn"], msite["fn"])
            except NotImplementedError:
                continue        # fall back on the sampled terms for this site
            kl_qp = scale_and_mask(kl_qp, scale=gsite["scale"], mask=gsite["mask"])
            if torch.is_tensor(kl_qp):
                assert kl_qp.shape == gsite["fn"].batch_shape
                kl_sum = kl_qp.sum() if kl_qp.dim() else kl_qp
            else:
                kl_sum = kl_qp * torch.Size(gsite["fn"].batch_shape).numel()
            analytic.add(name)
            if not batch.add_term(kl_sum, -1.0):
                left.append((-1.0, kl_sum))
        left += model_trace.collect_log_prob_sums(
            batch, 1.0, lambda name, site: name not in analytic)
        left += guide_trace.collect_log_prob_sums(
            batch, -1.0, lambda name, site: name not in analytic and name not in delta_sites)
        total = batch.total(coef)
        for sign, term in left:
            total = total + (coef * sign) * term
        return total

    def loss(self, model, guide, *args, **kwargs):
        loss = 0.0
        with torch.no_grad():
            for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
                loss = loss + self._batched_total(model_trace, guide_trace,
                                                  coef=-1.0 / self.num_particles)
        loss = torch_item(loss)
        warn_if_nan(loss, "loss")
        return loss

    def differentiable_loss(self, model, guide, *args, **kwargs):
        loss = 0.0
        for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
            loss = loss + self._batched_total(model_trace, guide_trace,
                                              coef=-1.0 / self.num_particles)
        warn_if_nan(loss, "loss")
        return loss

    @_grad_sink()
    def loss_and_grads_device(self, model, guide, *args, **kwargs):
        loss = None
        for model_trace, guide_trace in self._get_traces(model, guide, args, kwargs):
            sl = self._batched_total(model_trace, guide_trace, coef=-1.0 / self.num_particles)
            term = sl.detach() if isinstance(sl, torch.Tensor) else sl
            loss = term if loss is None else loss + term
            if isinstance(sl, torch.Tensor) and sl.requires_grad:
                sl.backward(_unit_grad(sl), retain_graph=self.retain_graph)
        return 0.0 if loss is None else loss
