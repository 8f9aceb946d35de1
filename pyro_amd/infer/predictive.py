"""Predictive: posterior-predictive sampling from SVI guides or MCMC samples
(reference: pyro/infer/predictive.py:79-160 _predictive, :162-325 Predictive).

Vectorised over the posterior draws through a leading plate (``parallel=True``), which is how the
draws stay one batch of device work instead of a Python loop per sample; ``parallel=False`` runs the
model once per draw as the reference does.
"""
import torch

from .. import poutine
from ..primitives import plate


def _guess_max_plate_nesting(model, args, kwargs):
    with poutine.block():
        trace = poutine.trace(model).get_trace(*args, **kwargs)
    dims = [frame.dim for site in trace.nodes.values() if site["type"] == "sample"
            for frame in site["cond_indep_stack"] if frame.vectorized]
    return -min(dims) if dims else 0


def _predictive_sequential(model, posterior_samples, model_args, model_kwargs, num_samples,
                           return_site_shapes):
    collected = {name: [] for name in return_site_shapes}
    traces = []
    for i in range(num_samples):
        draw = {k: v[i] for k, v in posterior_samples.items()}
        trace = poutine.trace(poutine.condition(model, draw)).get_trace(*model_args, **model_kwargs)
        traces.append(trace)
        for name in return_site_shapes:
            collected[name].append(trace.nodes[name]["value"])
    samples = {}
    for name, vals in collected.items():
        shape = return_site_shapes[name]
        samples[name] = vals if shape is None else torch.stack(vals).reshape(shape)
    return samples, traces


def _predictive(model, posterior_samples, num_samples, return_sites=(), parallel=False,
                model_args=(), model_kwargs=None, mask=True):
    """predictive.py:79-160: shapes are [num_samples] + [1] * (max_plate_nesting - batch rank) +
    site shape; observed sites are sampled afresh (the model runs under mask(False) so that scoring
    them costs nothing -- and so that code guarded by ``poutine.get_mask() is not False`` is
    skipped).  Returns (samples, trace): the vectorised trace, or the list of per-draw traces."""
    model_kwargs = model_kwargs or {}
    base = poutine.mask(model, mask=False) if mask else model
    max_plate_nesting = _guess_max_plate_nesting(base, model_args, model_kwargs)
    with poutine.block():
        model_trace = poutine.util.prune_subsample_sites(
            poutine.trace(base).get_trace(*model_args, **model_kwargs))
    reshaped = {}
    for name, sample in posterior_samples.items():
        sample_shape = tuple(sample.shape[1:])
        reshaped[name] = sample[:num_samples].reshape(
            (num_samples,) + (1,) * (max_plate_nesting - len(sample_shape)) + sample_shape)
    return_site_shapes = {}
    for name, site in model_trace.nodes.items():
        if site["type"] != "sample":
            continue
        append_ndim = max_plate_nesting - len(site["fn"].batch_shape)
        site_shape = (num_samples,) + (1,) * append_ndim + tuple(site["value"].shape)
        if return_sites:
            if name in return_sites:
                return_site_shapes[name] = site_shape
        elif return_sites is None:
            return_site_shapes[name] = site_shape
        elif name not in posterior_samples:
            return_site_shapes[name] = site_shape
    if return_sites is not None and "_RETURN" in return_sites:      # the model's return value
        value = model_trace.nodes["_RETURN"]["value"]
        return_site_shapes["_RETURN"] = (num_samples,) + tuple(value.shape) \
            if torch.is_tensor(value) else None
    if not parallel:
        return _predictive_sequential(base, posterior_samples, model_args, model_kwargs,
                                      num_samples, return_site_shapes)

    def vectorized(*args, **kwargs):
        with plate("_num_predictive_samples", num_samples, dim=-max_plate_nesting - 1):
            return base(*args, **kwargs)

    trace = poutine.trace(poutine.condition(vectorized, reshaped)).get_trace(*model_args,
                                                                            **model_kwargs)
    out = {}
    for name, shape in return_site_shapes.items():
        value = trace.nodes[name]["value"]
        if name == "_RETURN" and shape is None:
            out[name] = value
            continue
        out[name] = value.expand(shape) if value.numel() < torch.Size(shape).numel() \
            else value.reshape(shape)
    return out, trace


class Predictive(torch.nn.Module):
    """``Predictive(model, posterior_samples=...)`` or ``Predictive(model, guide=guide,
    num_samples=S)``: draws of the model's sites given posterior draws of the latents
    (predictive.py:162-325).  By default the sites NOT in ``posterior_samples`` are returned; with a
    guide, all of the model's sites (the guide's draws included)."""

    def __init__(self, model, posterior_samples=None, guide=None, num_samples=None,
                 return_sites=(), parallel=False):
        super().__init__()
        if posterior_samples is None:
            if num_samples is None:
                raise ValueError("Either posterior_samples or num_samples must be specified.")
            posterior_samples = {}
        for name, sample in posterior_samples.items():
            batch_size = sample.shape[0]
            if num_samples is None:
                num_samples = batch_size
            elif num_samples != batch_size:
                import warnings
                warnings.warn("Sample's leading dimension size {} is different from the provided {} "
                              "num_samples argument. Defaulting to {}.".format(
                                  batch_size, num_samples, batch_size), UserWarning)
                num_samples = batch_size
        if num_samples is None:
            raise ValueError("No sample sites in posterior samples to infer num_samples.")
        if guide is not None and posterior_samples:
            raise ValueError("`posterior_samples` cannot be provided with the `guide` argument.")
        if return_sites is not None:
            assert isinstance(return_sites, (list, tuple, set))
        self.model, self.posterior_samples, self.guide = model, posterior_samples, guide
        self.num_samples, self.return_sites, self.parallel = num_samples, return_sites, parallel

    def call(self, *args, **kwargs):
        """forward() with the values as a tuple ordered by site name (predictive.py:262-275)."""
        result = self.forward(*args, **kwargs)
        return tuple(v for _, v in sorted(result.items()))

    def _run(self, args, kwargs, return_sites, parallel):
        posterior_samples = self.posterior_samples
        if self.guide is not None:
            # draws of every guide site, auxiliary ones included: conditioning the model on names
            # it does not have is a no-op
            posterior_samples, _ = _predictive(self.guide, posterior_samples, self.num_samples,
                                               return_sites=None, parallel=self.parallel,
                                               model_args=args, model_kwargs=kwargs)
        return _predictive(self.model, posterior_samples, self.num_samples,
                           return_sites=return_sites, parallel=parallel, model_args=args,
                           model_kwargs=kwargs)

    @torch.no_grad()
    def forward(self, *args, **kwargs):
        return_sites = self.return_sites
        if self.guide is not None:
            return_sites = None if not return_sites else return_sites   # all sites with a guide
        return self._run(args, kwargs, return_sites, self.parallel)[0]

    def get_samples(self, *args, **kwargs):
        import warnings
        warnings.warn("The method `.get_samples` has been deprecated in favor of `.forward`.",
                      DeprecationWarning)
        return self.forward(*args, **kwargs)

    @torch.no_grad()
    def get_vectorized_trace(self, *args, **kwargs):
        """One vectorised trace of the predictive distribution (every batch dim of the model must be
        declared through ``plate``; predictive.py:311-333)."""
        return self._run(args, kwargs, self.return_sites, True)[1]
