"""User-facing primitives: sample / param / plate / factor / deterministic
(reference: pyro/primitives.py:57-91,125-192,283-389)."""
import warnings

import torch

from . import distributions as dist
from . import rng
from .ops import lazy as _lazy
from .params import _PARAM_STORE
from .poutine import settings as _poutine_settings
from .poutine.handlers import PlateMessenger
from .poutine.runtime import am_i_wrapped, apply_stack, new_message


def get_param_store():
    return _PARAM_STORE


def clear_param_store():
    _PARAM_STORE.clear()


def set_rng_seed(seed):
    rng.set_rng_seed(seed)


def enable_validation(is_validate=True):
    from .infer import util as _infer_util
    dist.enable_validation(is_validate)
    _infer_util.enable_validation(is_validate)
    _poutine_settings.enable_validation(is_validate)


class validation_enabled:
    def __init__(self, is_validate=True):
        self.is_validate = is_validate

    def __enter__(self):
        from .infer import util as _infer_util
        self.prev = (dist.is_validation_enabled(), _poutine_settings.validation_enabled(),
                     _infer_util.is_validation_enabled())
        enable_validation(self.is_validate)

    def __exit__(self, *a):
        from .infer import util as _infer_util
        dist.enable_validation(self.prev[0])
        _poutine_settings.enable_validation(self.prev[1])
        _infer_util.enable_validation(self.prev[2])


def _partially_observed(name, fn, obs, obs_mask, *args, **kwargs):
    """``obs_mask`` (reference: pyro/primitives.py:94-122): the site splits into
    ``<name>_observed`` (scored where the mask holds) and ``<name>_unobserved`` (a latent, scored
    where it does not); the model continues with their interleaving, recorded as the deterministic
    site ``<name>``."""
    from .poutine import mask as _mask
    with _mask(mask=obs_mask):
        observed = sample(name + "_observed", fn, *args, obs=obs, **kwargs)
    with _mask(mask=~obs_mask):
        unobserved = sample(name + "_unobserved", fn, *args, **kwargs)
    batch_mask = obs_mask.reshape(tuple(obs_mask.shape) + (1,) * fn.event_dim)
    try:
        value = torch.where(batch_mask, observed, unobserved)
    except RuntimeError as e:
        if "must match the size of tensor" in str(e):
            shape = torch.broadcast_shapes(observed.shape, unobserved.shape)
            raise ValueError("Invalid obs_mask shape {}; should be broadcastable to batch_shape = {}"
                             .format(tuple(obs_mask.shape),
                                     tuple(shape[:len(shape) - fn.event_dim]))) from e
        raise
    return deterministic(name, value, event_dim=fn.event_dim)


def sample(name, fn, *args, obs=None, obs_mask=None, infer=None, **kwargs):
    """Sample (or observe) a value at a named site."""
    infer = {} if infer is None else infer.copy()
    is_observed = infer.pop("is_observed", obs is not None)
    if obs_mask is not None:
        return _partially_observed(name, fn, obs, obs_mask, *args, **kwargs)
    if not am_i_wrapped():
        if obs is not None and not infer.get("_deterministic"):
            warnings.warn("trying to observe a value outside of inference at " + name,
                          RuntimeWarning)
            return obs
        return fn(*args, **kwargs)
    msg = new_message("sample", name, fn, args, kwargs, obs, is_observed, infer)
    apply_stack(msg)
    value = msg["value"]
    if not is_observed and isinstance(value, torch.Tensor) and \
            (msg.get("_replayed") or msg["is_observed"]):
        # a latent whose value was fixed from outside (replayed from a guide trace, or conditioned
        # as the HMC/NUTS potential does) reaches MODEL code as a transparent tensor subclass that
        # lets unmodified GLM model text (w @ X.t()) reach the fused kernel (ops/lazy.py); the
        # trace keeps the plain tensor, and guide code keeps getting plain tensors (an alias there
        # would split the latent's gradient over two autograd edges)
        value = _lazy.as_latent(value)
    return value


def factor(name, log_factor, *, has_rsample=None):
    unit = dist.Unit(log_factor, has_rsample=has_rsample)
    unit_value = unit.sample()
    sample(name, unit, obs=unit_value, infer={"is_auxiliary": True})


def deterministic(name, value, event_dim=None):
    event_dim = value.dim() if event_dim is None else event_dim
    return sample(name, dist.Delta(value, event_dim=event_dim).mask(False), obs=value,
                  infer={"_deterministic": True})


def param(name, init_tensor=None, constraint=dist.constraints.real, event_dim=None):
    """Fetch (creating on first use) a named learnable parameter."""
    if not am_i_wrapped():
        return _PARAM_STORE.get_param(name, init_tensor, constraint, event_dim)
    # the message reads like the reference's (primitives.py:88-89): fn = the store's get_param,
    # args = (name[, init]), kwargs = constraint / event_dim -- poutine.lift rewrites these
    args = (name,) if init_tensor is None else (name, init_tensor)
    msg = new_message("param", name, _PARAM_STORE.get_param, args,
                      {"constraint": constraint, "event_dim": event_dim})
    apply_stack(msg)
    return msg["value"]


def param_unconstrained(name, init_tensor=None, constraint=dist.constraints.real, event_dim=None):
    """Like ``param`` (same site name, same "param" message, created on first use) but the site's
    value is the UNCONSTRAINED leaf tensor: for callers that apply the constraint's transform inside
    a fused kernel (AutoNormal: scale = softplus(rho) in pa_meanfield_normal_sample) and would
    otherwise pay a transform launch per parameter per step.  ``value.unconstrained()`` returns the
    leaf itself, which is what SVI collects."""
    import weakref

    def fn(*a, **kw):
        if name not in _PARAM_STORE:
            _PARAM_STORE.get_param(name, init_tensor, constraint, event_dim)
        leaf = _PARAM_STORE._params[name]
        try:
            leaf.unconstrained = weakref.ref(leaf)
        except AttributeError:
            pass
        return leaf

    if not am_i_wrapped():
        return fn()
    msg = new_message("param", name, fn, (), {"event_dim": event_dim},
                      infer={"_unconstrained_value": True})
    apply_stack(msg)
    return msg["value"]


class plate(PlateMessenger):
    """Conditional-independence context, vectorised (``with``) or sequential (``for``)."""


def plate_stack(prefix, sizes, rightmost_dim=-1):
    from contextlib import ExitStack

    class _Stack(ExitStack):
        def __enter__(self_):
            super().__enter__()
            for i, size in enumerate(reversed(sizes)):
                self_.enter_context(plate("{}_{}".format(prefix, i), size, dim=rightmost_dim - i))
            return self_

    return _Stack()


def module(name, nn_module, update_module_params=False):
    """Register every trainable parameter of a torch.nn.Module in the param store under
    ``<name>$$$<parameter name>`` (reference: pyro/primitives.py:403-503).  A parameter the store
    already holds under that name (a loaded checkpoint, an earlier module object) wins: with
    ``update_module_params=True`` the stored tensors are put INTO the module in place of its own."""
    import inspect
    import warnings as _warnings
    from operator import attrgetter
    if "$$$" in name:
        raise AssertionError("improper module name, since contains $$$")
    if inspect.isclass(nn_module):
        raise NotImplementedError("pyro.module does not support class constructors for the "
                                  "argument nn_module")
    stored = {}
    for pname, p in nn_module.named_parameters():
        if p.requires_grad:
            returned = param("{}$$${}".format(name, pname), p)
            if returned.data_ptr() != p.data_ptr() or returned.shape != p.shape:
                stored[pname] = returned
        elif nn_module.training:
            _warnings.warn("{} was not registered in the param store because requires_grad=False. "
                           "You can silence this warning by calling my_module.train(False)"
                           .format(pname))
    if stored and update_module_params:
        for pname in [n for n, _ in nn_module.named_parameters()]:
            if pname not in stored:
                continue
            head, _, leaf = pname.rpartition(".")
            owner = attrgetter(head)(nn_module) if head else nn_module
            owner._parameters[leaf] = stored[pname]
    return nn_module


def subsample(data, event_dim):
    """``data`` cut down to the subsamples of the enclosing plates: each plate selects along its own dim,
    counted from the right of ``data``'s batch part, i.e. left of the ``event_dim`` rightmost dims
    (reference: pyro/primitives.py:237-280).  Outside of plates: ``data`` itself."""
    assert isinstance(event_dim, int) and event_dim >= 0
    if not am_i_wrapped():
        return data
    msg = new_message("subsample", None, None, (data, event_dim), {"event_dim": event_dim},
                      value=data)
    msg["done"] = True
    apply_stack(msg)
    return msg["value"]


def random_module(name, nn_module, prior, *args, **kwargs):
    """DEPRECATED in the reference too (primitives.py:506-543): a callable that returns a copy of
    ``nn_module`` whose parameters are drawn from ``prior`` (``poutine.lift`` over ``pyro.module``)."""
    import copy
    import warnings as _warnings
    from . import poutine as _poutine
    _warnings.warn("The `random_module` primitive is deprecated, and will be removed in a future "
                   "release. Use `pyro.nn.Module` to create Bayesian modules from `torch.nn.Module` "
                   "instances.", FutureWarning)
    assert hasattr(nn_module, "parameters"), "Module is not a NN module."
    lifted = _poutine.lift(module, prior=prior)

    def _fn():
        # update_module_params=True: the draws have to end up in the returned copy
        return lifted(name, copy.deepcopy(nn_module), *args, update_module_params=True, **kwargs)

    return _fn


class iarange(plate):
    """Deprecated spelling of :class:`plate` as a context manager (pyro/primitives.py:392-397)."""

    def __init__(self, *args, **kwargs):
        import warnings as _warnings
        _warnings.warn("pyro.iarange is deprecated; use pyro.plate instead", DeprecationWarning)
        super().__init__(*args, **kwargs)


class irange(plate):
    """Deprecated spelling of :class:`plate` as a loop (pyro/primitives.py:400-405)."""

    def __init__(self, *args, **kwargs):
        import warnings as _warnings
        _warnings.warn("pyro.irange is deprecated; use pyro.plate instead", DeprecationWarning)
        super().__init__(*args, **kwargs)


def barrier(data):
    """EXPERIMENTAL in the reference (a synchronisation point for lazily evaluated funsor values under
    ``poutine.collapse``): values are always ground here, so this is the identity."""
    return data
