"""Automatic guides (reference: pyro/infer/autoguide/guides.py: AutoGuide :60-200,
AutoDelta :330-412, AutoNormal :415-603, AutoDiagonalNormal :968-1029).

AutoNormal generates exactly the guide sites the fused kernels see: one
``Normal(loc, scale).to_event(k)`` auxiliary site per latent plus a ``Delta`` carrying the
transform's log-abs-det-Jacobian.
"""
import contextlib
from contextlib import ExitStack

import torch
from torch.distributions import biject_to, constraints

from ... import distributions as dist
from ... import poutine
from ...distributions.util import sum_rightmost
from ...primitives import param, plate, sample
from ...ops.torch_library import dispatcher_op as _dispatcher_op
from .initialization import InitMessenger, init_to_feasible, init_to_median


def _softplus_inv(y):
    return y + torch.log(-torch.expm1(-y))


class _SoftplusPositive(constraints.Constraint):
    """scale constraint of AutoNormal: positive, parameterised through softplus."""

    is_discrete = False
    event_dim = 0

    def check(self, value):
        return value > 0


softplus_positive = _SoftplusPositive()


@dist.transform_to.register(_SoftplusPositive)
def _transform_to_softplus_positive(constraint):
    return torch.distributions.transforms.SoftplusTransform()


_IDENTITY = biject_to(constraints.real)


def _exp_lower(transform):
    from ...distributions import fused
    return fused.exp_lower_bound_of(transform)


def _checked_init_scale(value):
    """The guides' ``init_scale`` argument: a positive python float (same message as the reference's
    constructors, which its tests match on)."""
    if isinstance(value, float) and value > 0:
        return value
    raise ValueError("Expected init_scale > 0. but got {}".format(value))


def _is_identity(transform):
    while isinstance(transform, torch.distributions.transforms.IndependentTransform):
        transform = transform.base_transform
    return transform is _IDENTITY


@contextlib.contextmanager
def helpful_support_errors(site):
    """A latent site without an unconstrained coordinate system (discrete, on a sphere) fails deep
    inside ``biject_to``; say what to do instead (reference: autoguide/utils.py:62-86)."""
    try:
        yield
    except NotImplementedError as e:
        support = site["fn"].support
        name = site["name"]
        if getattr(support, "is_discrete", False):
            raise ValueError(
                "Continuous inference cannot handle discrete sample site '{0}'. Consider enumerating "
                "that variable as documented in https://pyro.ai/examples/enumeration.html . If you are "
                "already enumerating, take care to hide this site when constructing an autoguide, e.g. "
                "guide = AutoNormal(poutine.block(model, hide=['{0}'])).".format(name)) from None
        if "sphere" in repr(support).lower():
            raise ValueError(
                "Continuous inference cannot handle spherical sample site '{0}'. Consider using "
                "ProjectedNormal distribution together with a reparameterizer, e.g. "
                "poutine.reparam(config={{'{0}': ProjectedNormalReparam()}}).".format(name)) from None
        raise e from None


def periodic_repeat(tensor, size, dim):
    """``tensor`` repeated along ``dim`` (period = its size there) and cut to ``size``: how the value
    drawn for one subsample initialises the parameter of the FULL plate
    (reference: pyro/ops/tensor_utils.py periodic_repeat)."""
    if dim >= 0:
        dim -= tensor.dim()
    period = tensor.size(dim)
    repeats = [1] * tensor.dim()
    repeats[dim] = -(-size // period)
    return tensor.repeat(*repeats).narrow(dim, 0, size)


def _full_plate_value(value, site, event_dim):
    """A value observed under subsampled plates, laid out for the full plates."""
    for frame in site["cond_indep_stack"]:
        full_size = getattr(frame, "full_size", None) or frame.size
        if full_size != frame.size:
            value = periodic_repeat(value, full_size, frame.dim - event_dim).contiguous()
    return value


class AutoGuide:
    def __init__(self, model, *, create_plates=None):
        self.model = model
        self.create_plates = create_plates
        self.prototype_trace = None
        self._prototype_frames = {}
        self.prefix = type(self).__name__
        self.master = None                  # weakref to the AutoGuideList this guide is a part of

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def call(self, *args, **kwargs):
        """The draws as a tuple ordered by site name (reference: guides.py:108-115)."""
        result = self(*args, **kwargs)
        return tuple(v for _, v in sorted(result.items()))

    def _create_plates(self, *args, **kwargs):
        master = getattr(self, "master", None)
        if master is not None:
            # a part of an AutoGuideList: the list made the plates of this call (one subsample for all)
            assert self.create_plates is None, "Cannot pass create_plates() to non-master guide"
            return master().plates
        if self.create_plates is None:
            plates = {}
        else:
            plates = self.create_plates(*args, **kwargs)
            if isinstance(plates, plate):
                plates = [plates]
            plates = {p.name: p for p in plates}
        for name, frame in sorted(self._prototype_frames.items()):
            if name not in plates:
                full_size = getattr(frame, "full_size", frame.size)
                plates[name] = plate(name, full_size, dim=frame.dim,
                                     subsample_size=frame.size if frame.size != full_size else None)
        self.plates = plates
        return plates

    def _setup_prototype(self, *args, **kwargs):
        model = poutine.block(InitMessenger(self.init_loc_fn)(self.model))
        # prototype: run the (blocked) model once, unwrapped from any enumeration
        with poutine.block():
            self.prototype_trace = poutine.trace(
                InitMessenger(self.init_loc_fn)(self.model)).get_trace(*args, **kwargs)
        del model
        self.prototype_trace = poutine.util.prune_subsample_sites(self.prototype_trace)
        self._prototype_frames = {}
        for name, site in self.prototype_trace.iter_stochastic_nodes():
            for frame in site["cond_indep_stack"]:
                if frame.vectorized:
                    self._prototype_frames[frame.name] = frame
                else:
                    raise NotImplementedError("AutoGuide does not support sequential pyro.plate")

    def median(self, *args, **kwargs):
        raise NotImplementedError


class AutoNormal(AutoGuide):
    scale_constraint = softplus_positive

    def __init__(self, model, *, init_loc_fn=init_to_feasible, init_scale=0.1, create_plates=None):
        self.init_loc_fn = init_loc_fn
        self._init_scale = _checked_init_scale(init_scale)
        super().__init__(model, create_plates=create_plates)
        self._event_dims = {}
        self._inits = {}

    def _setup_prototype(self, *args, **kwargs):
        super()._setup_prototype(*args, **kwargs)
        for name, site in self.prototype_trace.iter_stochastic_nodes():
            if site["infer"].get("enumerate") == "parallel":
                continue
            with torch.no_grad(), helpful_support_errors(site):
                init_loc = biject_to(site["fn"].support).inv(site["value"].detach()).detach()
            event_dim = site["fn"].event_dim + init_loc.dim() - site["value"].dim()
            self._event_dims[name] = event_dim
            init_loc = _full_plate_value(init_loc, site, event_dim)
            self._inits[name] = (init_loc.contiguous().clone(),
                                 torch.full_like(init_loc, self._init_scale))

    def _latent_sites(self):
        for name, site in self.prototype_trace.iter_stochastic_nodes():
            if name in self._inits:
                yield name, site

    def _get_loc_and_scale(self, name):
        init_loc, init_scale = self._inits[name]
        event_dim = self._event_dims[name]
        loc = param("{}.locs.{}".format(self.prefix, name), init_loc, constraints.real,
                    event_dim=event_dim)
        scale = param("{}.scales.{}".format(self.prefix, name), init_scale, self.scale_constraint,
                      event_dim=event_dim)
        return loc, scale

    def _fused_draw(self):
        """All latent sites drawn by ONE launch (distributions.fused.meanfield_sample) instead of a
        softplus + rsample chain per site; returns {name: Normal with the draw attached} or None
        when the fused form does not apply (parameters not on the device, a replaced eps source,
        subsampled plates, outer plates interleaved with a site's own dims, a guide prefix whose
        scale constraint is not the softplus one)."""
        from ... import kernels, rng
        from ...distributions import fused
        from ...poutine.runtime import _PYRO_STACK
        from ...primitives import param_unconstrained

        if self.scale_constraint is not softplus_positive or rng.normal is not rng._default_normal:
            return None
        names = [name for name, _ in self._latent_sites()]
        if not names:
            return None
        own = set()
        for name, site in self._latent_sites():
            for frame in site["cond_indep_stack"]:
                if getattr(frame, "full_size", frame.size) != frame.size:
                    return None                       # subsampled plate: per-site path
                own.add(frame.name)
        outer = [(m.dim, m.size) for m in _PYRO_STACK
                 if isinstance(m, plate) and m._vectorized is True and m.name not in own]
        if any(m.size != m.subsample_size for m in _PYRO_STACK
               if isinstance(m, plate) and m._vectorized is True and m.name not in own):
            return None
        locs, rhos = [], []
        for name in names:
            init_loc, init_scale = self._inits[name]
            ed = self._event_dims[name]
            loc = param_unconstrained("{}.locs.{}".format(self.prefix, name), init_loc,
                                      constraints.real, event_dim=ed)
            rho = param_unconstrained("{}.scales.{}".format(self.prefix, name), init_scale,
                                      self.scale_constraint, event_dim=ed)
            if not kernels.on_device(loc) or loc.dtype != rho.dtype \
                    or loc.dtype not in (torch.float32, torch.float64) \
                    or not loc.is_contiguous() or not rho.is_contiguous():
                return None
            # every outer (particle) plate must sit to the left of the site's own batch dims
            batch_rank = loc.dim() - ed
            if any(-d <= batch_rank for d, _ in outer):
                return None
            locs.append(loc)
            rhos.append(rho)
        if any(l.dtype != locs[0].dtype or l.device != locs[0].device for l in locs):
            return None
        P = 1
        for _, size in outer:
            P *= size
        drawn = fused.meanfield_sample(locs, rhos, P)
        out = {}
        for name, loc, (z, scale, loc_out) in zip(names, locs, drawn):
            ed = self._event_dims[name]
            batch_rank = loc.dim() - ed
            lead = [1] * (max([-d for d, _ in outer], default=batch_rank) - batch_rank)
            for d, size in outer:
                lead[len(lead) + batch_rank + d] = size
            fn = dist.Normal(loc_out.reshape(loc.shape), scale.reshape(loc.shape))
            fn._presampled = z.reshape(tuple(lead) + tuple(loc.shape))
            fn._drawn = (z, loc_out, scale, P)
            out[name] = fn
        return out

    def forward(self, *args, **kwargs):
        if self.prototype_trace is None:
            self._setup_prototype(*args, **kwargs)
        from ... import kernels
        from ...distributions import fused
        plates = self._create_plates(*args, **kwargs)
        fused_fns = self._fused_draw()
        result = {}
        for name, site in self._latent_sites():
            transform = biject_to(site["fn"].support)
            with ExitStack() as stack:
                for frame in site["cond_indep_stack"]:
                    if frame.vectorized:
                        stack.enter_context(plates[frame.name])
                if fused_fns is not None:
                    base_fn = fused_fns[name]
                else:
                    site_loc, site_scale = self._get_loc_and_scale(name)
                    base_fn = dist.Normal(site_loc, site_scale)
                unconstrained_latent = sample(
                    name + "_unconstrained", base_fn.to_event(self._event_dims[name]),
                    infer={"is_auxiliary": True})
                lower = _exp_lower(transform)
                if (lower is not None and poutine.get_mask() is not False
                        and type(unconstrained_latent) is torch.Tensor
                        and kernels.on_device(unconstrained_latent)
                        and unconstrained_latent.dtype in (torch.float32, torch.float64)):
                    # support (lower, inf): value and the Jacobian term from one kernel
                    value, log_density = fused.exp_site(unconstrained_latent, site["fn"].event_dim, lower)
                else:
                    value = transform(unconstrained_latent)
                    if poutine.get_mask() is False or _is_identity(transform):
                        log_density = 0.0      # real support: the Jacobian term is identically zero
                    else:
                        log_density = transform.inv.log_abs_det_jacobian(value, unconstrained_latent)
                        log_density = sum_rightmost(
                            log_density, log_density.dim() - value.dim() + site["fn"].event_dim)
                delta_dist = dist.Delta(value, log_density=log_density,
                                        event_dim=site["fn"].event_dim)
                result[name] = sample(name, delta_dist)
        return result

    @torch.no_grad()
    def median(self, *args, **kwargs):
        out = {}
        for name, site in self._latent_sites():
            loc, _ = self._get_loc_and_scale(name)
            out[name] = biject_to(site["fn"].support)(loc).clone()
        return out

    @torch.no_grad()
    def quantiles(self, quantiles, *args, **kwargs):
        out = {}
        for name, site in self._latent_sites():
            loc, scale = self._get_loc_and_scale(name)
            q = torch.tensor(quantiles, dtype=loc.dtype, device=loc.device)
            q = q.reshape((-1,) + (1,) * loc.dim())
            vals = torch.distributions.Normal(loc, scale).icdf(q)
            out[name] = biject_to(site["fn"].support)(vals)
        return out


# ---- AutoContinuous family (reference: guides.py:605-1029) -----------------------------------------
class _UnitLowerCholesky(constraints.Constraint):
    """Lower-triangular square matrices with a unit diagonal (pyro constraints.unit_lower_cholesky)."""

    event_dim = 2
    is_discrete = False

    def check(self, value):
        tril = value.tril()
        lower = (tril == value).reshape(value.shape[:-2] + (-1,)).min(-1)[0]
        ones = (value.diagonal(dim1=-2, dim2=-1) == 1).min(-1)[0]
        return lower & ones


unit_lower_cholesky = _UnitLowerCholesky()


class _UnitLowerCholeskyTransform(torch.distributions.transforms.Transform):
    """x -> tril(x, -1) + I (pyro/distributions/transforms/cholesky.py UnitLowerCholeskyTransform)."""

    domain = constraints.independent(constraints.real, 2)
    codomain = unit_lower_cholesky
    bijective = True

    def __eq__(self, other):
        return isinstance(other, _UnitLowerCholeskyTransform)

    def _call(self, x):
        return x.tril(-1) + torch.eye(x.size(-1), device=x.device, dtype=x.dtype)

    def _inverse(self, y):
        return y.tril(-1)

    def log_abs_det_jacobian(self, x, y):
        return x.new_zeros(x.shape[:-2])


@dist.transform_to.register(_UnitLowerCholesky)
def _transform_to_unit_lower_cholesky(constraint):
    return _UnitLowerCholeskyTransform()


class _GuideMVN(torch.distributions.MultivariateNormal, dist.TorchDistributionMixin):
    """MultivariateNormal whose reparameterised draw takes its standard normals from the backend's
    Philox stream (pyro_amd.rng.normal) and applies the affine map as ONE dense product over all
    particles: z = loc + eps @ scale_tril^T -- the genuine dense matvec of this guide family
    (rocBLAS / MFMA for large latent spaces)."""

    def rsample(self, sample_shape=torch.Size()):
        from ... import rng
        shape = self._extended_shape(sample_shape)
        eps = rng.normal(shape, self.loc.dtype, self.loc.device)
        return self.loc + torch.matmul(eps, self._unbroadcasted_scale_tril.transpose(-1, -2))

    def expand(self, batch_shape, _instance=None):
        new = torch.distributions.MultivariateNormal.expand(
            self, batch_shape, _instance=self._get_checked_instance(_GuideMVN, _instance))
        return new


class _FusedGuideMVN(dist.TorchDistribution):
    """The posterior of AutoMultivariateNormal as ONE fused draw: rsample() produces the value
    and its log-density together (distributions.fused.mvn_tril_sample: the draw knows its own
    standard normals, so log q needs no triangular solve), log_prob(that value) hands the density
    back.  Any other use (log_prob of a foreign value, mean, ...) goes through the ordinary
    MultivariateNormal built on demand."""

    arg_constraints = {}
    support = constraints.real_vector
    has_rsample = True

    def __init__(self, loc, rho, A, build, batch_shape=torch.Size()):
        self._leaves, self._build = (loc, rho, A), build
        self._value = self._logq = None
        super().__init__(torch.Size(batch_shape), torch.Size((loc.numel(),)), validate_args=False)

    def expand(self, batch_shape, _instance=None):
        return _FusedGuideMVN(*self._leaves, self._build, batch_shape=batch_shape)

    def rsample(self, sample_shape=torch.Size()):
        from ...distributions import fused
        shape = self._extended_shape(sample_shape)
        z, logq = fused.mvn_tril_sample(*self._leaves, shape)
        self._value, self._logq = z.reshape(shape), logq.reshape(shape[:-1])
        return self._value

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)

    def log_prob(self, value):
        if value is self._value:
            return self._logq
        return self._build().expand(self.batch_shape).log_prob(value)

    @property
    def mean(self):
        return self._build().mean

    @property
    def variance(self):
        return self._build().variance


@_dispatcher_op("split_latent")
class _SplitLatent(torch.autograd.Function):
    """latent[..., sum(sizes)] -> one CONTIGUOUS tensor per site; the backward is one concatenation
    (slicing views would cost a zero-fill + strided copy per site and an add per extra site in
    autograd, and a .contiguous() copy wherever a kernel consumes the slice)."""

    @staticmethod
    def forward(ctx, latent, sizes):
        ctx.sizes, ctx.meta = sizes, (latent.shape, latent.dtype, latent.device)
        out, pos = [], 0
        for size in sizes:
            out.append(latent[..., pos:pos + size].contiguous())
            pos += size
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        shape, dtype, device = ctx.meta
        parts = [g if g is not None else torch.zeros(shape[:-1] + (size,), dtype=dtype, device=device)
                 for g, size in zip(grads, ctx.sizes)]
        return torch.cat(parts, dim=-1), None


def _product(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


class AutoContinuous(AutoGuide):
    """All latent sites transformed to unconstrained space and concatenated into ONE latent vector;
    subclasses provide the distribution over it (guides.py:605-828)."""

    def __init__(self, model, init_loc_fn=init_to_median):
        self.init_loc_fn = init_loc_fn
        super().__init__(model)

    def _setup_prototype(self, *args, **kwargs):
        super()._setup_prototype(*args, **kwargs)
        self._unconstrained_shapes = {}
        self._cond_indep_stacks = {}
        for name, site in self.prototype_trace.iter_stochastic_nodes():
            with torch.no_grad(), helpful_support_errors(site):
                self._unconstrained_shapes[name] = \
                    biject_to(site["fn"].support).inv(site["value"]).shape
            self._cond_indep_stacks[name] = site["cond_indep_stack"]
        self.latent_dim = sum(_product(shape) for shape in self._unconstrained_shapes.values())
        if self.latent_dim == 0:
            raise RuntimeError("{} found no latent variables; Use an empty guide instead".format(
                type(self).__name__))

    def _init_loc(self):
        parts = []
        for name, site in self.prototype_trace.iter_stochastic_nodes():
            with torch.no_grad():
                parts.append(biject_to(site["fn"].support).inv(site["value"].detach()).reshape(-1))
        latent = torch.cat(parts)
        assert latent.size() == (self.latent_dim,)
        return latent

    def get_posterior(self, *args, **kwargs):
        raise NotImplementedError

    def sample_latent(self, *args, **kwargs):
        pos_dist = self.get_posterior(*args, **kwargs)
        return sample("_{}_latent".format(self.prefix), pos_dist, infer={"is_auxiliary": True})

    def _unpack_latent(self, latent):
        batch_shape = latent.shape[:-1]   # plates outside of _setup_prototype, e.g. parallel particles
        sites = list(self.prototype_trace.iter_stochastic_nodes())
        sizes = tuple(_product(self._unconstrained_shapes[name]) for name, _ in sites)
        assert sum(sizes) == latent.size(-1)
        parts = _SplitLatent.invoke(latent, sizes) if latent.requires_grad else \
            torch.split(latent, sizes, dim=-1)
        for (name, site), part in zip(sites, parts):
            constrained_shape = site["value"].shape
            unconstrained_shape = self._unconstrained_shapes[name]
            event_dim = site["fn"].event_dim + len(unconstrained_shape) - len(constrained_shape)
            unconstrained_shape = torch.broadcast_shapes(unconstrained_shape,
                                                         batch_shape + (1,) * event_dim)
            yield site, part.reshape(unconstrained_shape)

    def forward(self, *args, **kwargs):
        if self.prototype_trace is None:
            self._setup_prototype(*args, **kwargs)
        latent = self.sample_latent(*args, **kwargs)
        plates = self._create_plates(*args, **kwargs)
        result = {}
        for site, unconstrained_value in self._unpack_latent(latent):
            name = site["name"]
            transform = biject_to(site["fn"].support)
            value = transform(unconstrained_value)
            if poutine.get_mask() is False or _is_identity(transform):
                log_density = 0.0
            else:
                log_density = transform.inv.log_abs_det_jacobian(value, unconstrained_value)
                log_density = sum_rightmost(
                    log_density, log_density.dim() - value.dim() + site["fn"].event_dim)
            delta_dist = dist.Delta(value, log_density=log_density, event_dim=site["fn"].event_dim)
            with ExitStack() as stack:
                for frame in self._cond_indep_stacks[name]:
                    if frame.vectorized:
                        stack.enter_context(plates[frame.name])
                result[name] = sample(name, delta_dist)
        return result

    def _loc_scale(self, *args, **kwargs):
        raise NotImplementedError

    @torch.no_grad()
    def median(self, *args, **kwargs):
        loc, _ = self._loc_scale(*args, **kwargs)
        return {site["name"]: biject_to(site["fn"].support)(unconstrained_value).clone()
                for site, unconstrained_value in self._unpack_latent(loc.detach())}

    @torch.no_grad()
    def quantiles(self, quantiles, *args, **kwargs):
        loc, scale = self._loc_scale(*args, **kwargs)
        q = torch.tensor(quantiles, dtype=loc.dtype, device=loc.device).unsqueeze(-1)
        latents = torch.distributions.Normal(loc, scale).icdf(q)
        result = {}
        for latent in latents:
            for site, unconstrained_value in self._unpack_latent(latent):
                result.setdefault(site["name"], []).append(
                    biject_to(site["fn"].support)(unconstrained_value))
        return {k: torch.stack(v) for k, v in result.items()}


class AutoDiagonalNormal(AutoContinuous):
    """Diagonal Normal over the concatenated unconstrained latent vector (guides.py:968-1029):
    parameters ``<prefix>.loc`` and ``<prefix>.scale`` (softplus-positive)."""

    scale_constraint = softplus_positive

    def __init__(self, model, init_loc_fn=init_to_median, init_scale=0.1):
        self._init_scale = _checked_init_scale(init_scale)
        super().__init__(model, init_loc_fn=init_loc_fn)

    def _setup_prototype(self, *args, **kwargs):
        super()._setup_prototype(*args, **kwargs)
        self._loc0 = self._init_loc()

    def _params(self):
        loc = param("{}.loc".format(self.prefix), lambda: self._loc0.clone(), constraints.real)
        scale = param("{}.scale".format(self.prefix),
                      lambda: torch.full_like(self._loc0, self._init_scale), self.scale_constraint)
        return loc, scale

    def get_posterior(self, *args, **kwargs):
        loc, scale = self._params()
        return dist.Normal(loc, scale).to_event(1)

    def _loc_scale(self, *args, **kwargs):
        return self._params()


class AutoMultivariateNormal(AutoContinuous):
    """Full-covariance Normal through its Cholesky factor (guides.py:855-965): parameters
    ``<prefix>.loc``, ``<prefix>.scale`` (softplus-positive) and ``<prefix>.scale_tril`` (unit lower
    Cholesky); scale_tril of the posterior = scale[..., None] * scale_tril."""

    scale_constraint = softplus_positive
    scale_tril_constraint = unit_lower_cholesky

    def __init__(self, model, init_loc_fn=init_to_median, init_scale=0.1):
        self._init_scale = _checked_init_scale(init_scale)
        super().__init__(model, init_loc_fn=init_loc_fn)

    def _setup_prototype(self, *args, **kwargs):
        super()._setup_prototype(*args, **kwargs)
        self._loc0 = self._init_loc()

    def _params(self):
        loc = param("{}.loc".format(self.prefix), lambda: self._loc0.clone(), constraints.real)
        scale = param("{}.scale".format(self.prefix),
                      lambda: torch.full_like(self._loc0, self._init_scale), self.scale_constraint)
        scale_tril = param("{}.scale_tril".format(self.prefix),
                           lambda: torch.eye(self.latent_dim, dtype=self._loc0.dtype,
                                             device=self._loc0.device), self.scale_tril_constraint)
        return loc, scale, scale_tril

    def _plain_posterior(self):
        loc, scale, scale_tril = self._params()
        return _GuideMVN(loc, scale_tril=scale[..., None] * scale_tril)

    def get_posterior(self, *args, **kwargs):
        from ... import kernels
        from ...primitives import param_unconstrained
        if self.scale_constraint is softplus_positive and \
                self.scale_tril_constraint is unit_lower_cholesky and self.latent_dim <= 4096:
            d = self.latent_dim
            loc = param_unconstrained("{}.loc".format(self.prefix),
                                      lambda: self._loc0.clone(), constraints.real)
            rho = param_unconstrained("{}.scale".format(self.prefix),
                                      lambda: torch.full_like(self._loc0, self._init_scale),
                                      self.scale_constraint)
            A = param_unconstrained("{}.scale_tril".format(self.prefix),
                                    lambda: torch.eye(d, dtype=self._loc0.dtype,
                                                      device=self._loc0.device),
                                    self.scale_tril_constraint)
            if kernels.on_device(loc) and A.is_contiguous() \
                    and loc.dtype in (torch.float32, torch.float64) \
                    and loc.dtype == rho.dtype == A.dtype:
                return _FusedGuideMVN(loc, rho, A, self._plain_posterior)
        return self._plain_posterior()

    def _loc_scale(self, *args, **kwargs):
        loc, scale, scale_tril = self._params()
        st = scale[..., None] * scale_tril
        return loc, st.pow(2).sum(-1).sqrt()


class AutoLowRankMultivariateNormal(AutoContinuous):
    """Low-rank plus diagonal Normal over the concatenated unconstrained latent vector (reference:
    guides.py:968-1029): parameters ``<prefix>.loc``, ``<prefix>.scale`` (softplus-positive) and
    ``<prefix>.cov_factor`` [latent_dim, rank]; covariance = scale (W W^T + I) scale.  ``rank`` defaults to
    round(sqrt(latent_dim))."""

    scale_constraint = softplus_positive

    def __init__(self, model, init_loc_fn=init_to_median, init_scale=0.1, rank=None):
        self._init_scale = _checked_init_scale(init_scale)
        if rank is not None and not (isinstance(rank, int) and rank > 0):
            raise ValueError("Expected rank > 0 but got {}".format(rank))
        self.rank = rank
        super().__init__(model, init_loc_fn=init_loc_fn)

    def _setup_prototype(self, *args, **kwargs):
        super()._setup_prototype(*args, **kwargs)
        self._loc0 = self._init_loc()
        if self.rank is None:
            self.rank = int(round(self.latent_dim ** 0.5))

    def _params(self):
        loc0 = self._loc0
        loc = param("{}.loc".format(self.prefix), lambda: loc0.clone(), constraints.real)
        scale = param("{}.scale".format(self.prefix),
                      lambda: torch.full_like(loc0, 0.5 ** 0.5 * self._init_scale), self.scale_constraint)
        cov_factor = param("{}.cov_factor".format(self.prefix),
                           lambda: loc0.new_empty(self.latent_dim, self.rank).normal_(
                               0, 1 / self.rank ** 0.5), constraints.real)
        return loc, scale, cov_factor

    def get_posterior(self, *args, **kwargs):
        loc, scale, cov_factor = self._params()
        return dist.LowRankMultivariateNormal(loc, cov_factor * scale.unsqueeze(-1), scale * scale)

    def _loc_scale(self, *args, **kwargs):
        loc, scale, cov_factor = self._params()
        return loc, scale * (cov_factor.pow(2).sum(-1) + 1).sqrt()


class AutoCallable(AutoGuide):
    """A hand-written guide function as a part of an :class:`AutoGuideList` (guides.py:279-316)."""

    def __init__(self, model, guide, median=lambda *args, **kwargs: {}):
        super().__init__(model)
        self._guide = guide
        self.median = median

    def forward(self, *args, **kwargs):
        result = self._guide(*args, **kwargs)
        return {} if result is None else result


class AutoGuideList(AutoGuide):
    """Several guides, each for a part of the model (made by ``poutine.block``-ing the model down to some
    of its sites), run one after the other; the plates are created once per call and shared
    (reference: guides.py:184-276)::

        guide = AutoGuideList(model)
        guide.append(AutoDelta(poutine.block(model, expose=["drift"])))
        guide.append(AutoNormal(poutine.block(model, hide=["drift"])))
    """

    init_loc_fn = staticmethod(init_to_feasible)

    def __init__(self, model, *, create_plates=None):
        super().__init__(model, create_plates=create_plates)
        self._parts = []

    def __len__(self):
        return len(self._parts)

    def __iter__(self):
        return iter(self._parts)

    def __getitem__(self, index):
        return self._parts[index]

    def append(self, part):
        if not isinstance(part, AutoGuide):
            part = AutoCallable(self.model, part)
        if getattr(part, "master", None) is not None:
            raise RuntimeError("The module `{}` is already added.".format(part.prefix))
        import weakref
        part.master = weakref.ref(self)
        part.prefix = "{}.{}".format(self.prefix, len(self._parts))     # parameter names: <list>.<i>...
        self._parts.append(part)

    def add(self, part):
        """Old spelling of ``append`` (kept, with the reference's deprecation notice)."""
        import warnings
        warnings.warn("The method `.add` has been deprecated in favor of `.append`.", DeprecationWarning,
                      stacklevel=2)
        self.append(part)

    def _collect(self, ask):
        """One dict out of every part's answer, in the order the parts were appended."""
        merged = {}
        for part in self._parts:
            merged.update(ask(part))
        return merged

    def forward(self, *args, **kwargs):
        if self.prototype_trace is None:
            self._setup_prototype(*args, **kwargs)
        self._create_plates(*args, **kwargs)               # once per call: the parts share them
        return self._collect(lambda part: part(*args, **kwargs))

    def median(self, *args, **kwargs):
        return self._collect(lambda part: part.median(*args, **kwargs))

    def quantiles(self, quantiles, *args, **kwargs):
        return self._collect(lambda part: part.quantiles(quantiles, *args, **kwargs))


class AutoDelta(AutoGuide):
    """MAP guide: a learnable point estimate per latent site."""

    def __init__(self, model, init_loc_fn=init_to_median, *, create_plates=None):
        self.init_loc_fn = init_loc_fn
        super().__init__(model, create_plates=create_plates)

    def forward(self, *args, **kwargs):
        if self.prototype_trace is None:
            self._setup_prototype(*args, **kwargs)
        plates = self._create_plates(*args, **kwargs)
        result = {}
        for name, site in self.prototype_trace.iter_stochastic_nodes():
            with ExitStack() as stack:
                for frame in site["cond_indep_stack"]:
                    if frame.vectorized:
                        stack.enter_context(plates[frame.name])
                event_dim = site["fn"].event_dim
                # the point estimate covers the FULL plates; inside the (subsampling) plates entered
                # above, pyro.param hands back the rows of the current subsample
                init = _full_plate_value(site["value"].detach(), site, event_dim).clone()
                with helpful_support_errors(site):
                    value = param("{}.{}".format(self.prefix, name), init, site["fn"].support,
                                  event_dim=event_dim)
                result[name] = sample(name, dist.Delta(value, event_dim=site["fn"].event_dim))
        return result

    @torch.no_grad()
    def median(self, *args, **kwargs):
        return {k: v.clone() for k, v in self(*args, **kwargs).items()}
