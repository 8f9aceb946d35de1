"""``PyroModule``: torch.nn.Modules whose attributes can be Pyro parameters and random variables (the role of
pyro.nn.module; the subset models for the SVI / MCMC paths are written with).

* ``PyroModule[nn.Linear](3, 1)`` -- the torch class mixed with ``PyroModule``; its ``nn.Parameter`` s are
  registered with ``pyro.param`` under dotted names (``"linear.weight"``) when they are read;
* ``self.scale = PyroParam(torch.ones(3), constraint=constraints.positive)`` -- a constrained learnable
  attribute: the module holds the unconstrained ``nn.Parameter`` (``scale_unconstrained``), reading
  ``self.scale`` goes through ``pyro.param`` and returns the constrained value;
* ``self.weight = PyroSample(dist.Normal(0., 1.).expand([1, 3]).to_event(2))`` -- reading ``self.weight``
  is a ``pyro.sample`` statement named after the attribute; the prior may be a function of the module.

Inside one ``module(...)`` call every attribute is evaluated once (a second read returns the same draw) and is
a named Pyro statement.  Outside of a call the module does not know its place in a module tree yet (names are
fixed when it is attached to its parent), so reads are plain values: the constrained parameter, a draw of the
prior, the ``nn.Parameter`` itself.
"""
import collections
import functools
import weakref

import torch
from torch.distributions import constraints, transform_to

from .. import primitives
from ..params import _PARAM_STORE


_MODULE_LOCAL_PARAMS = False      # pyro.settings "module_local_params": parameters stay inside the module


def _local_params():
    return _MODULE_LOCAL_PARAMS


class PyroParam(collections.namedtuple("PyroParam", ["init_value", "constraint", "event_dim"])):
    """``module.attr = PyroParam(init, constraint, event_dim)``; or, in a class body, a decorator on a method
    that computes the initial value lazily: ``@PyroParam`` / ``@PyroParam(constraint=...)``."""

    def __new__(cls, init_value=None, constraint=constraints.real, event_dim=None):
        return super().__new__(cls, init_value, constraint, event_dim)

    def __call__(self, initialiser):                      # @PyroParam(constraint=...) def attr(self): ...
        assert self.init_value is None
        return PyroParam(initialiser, self.constraint, self.event_dim)

    def __get__(self, obj, owner=None):                   # the decorated method read as an attribute
        if obj is None:
            return self
        name = self.init_value.__name__
        if name not in obj.__dict__["_pyro_params"]:
            setattr(obj, name, PyroParam(functools.partial(self.init_value, obj), self.constraint,
                                         self.event_dim))
        return obj.__getattr__(name)


class PyroSample:
    """``module.attr = PyroSample(prior)`` with ``prior`` a distribution or a function of the module; or a
    decorator on a method returning the prior."""

    def __init__(self, prior):
        self.prior = prior
        if not isinstance(prior, torch.distributions.Distribution):
            self.__name__ = getattr(prior, "__name__", type(prior).__name__)

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        name = self.prior.__name__
        obj.__dict__["_pyro_samples"].setdefault(name, self.prior)
        return obj.__getattr__(name)


def _dotted(prefix, name):
    return "{}.{}".format(prefix, name) if prefix else name


class _CallScope:
    """Shared by a module tree: how deep we are inside ``__call__`` s, and what was read so far."""

    def __init__(self):
        self.depth, self.memo = 0, {}

    def __enter__(self):
        self.depth += 1
        return self

    def __exit__(self, *exc):
        self.depth -= 1
        if self.depth == 0:
            self.memo.clear()

    @property
    def active(self):
        return self.depth > 0

    @property
    def cache(self):
        return self.memo

    def recall(self, name, make):
        if name not in self.memo:
            self.memo[name] = make()
        return self.memo[name]


class _PyroModuleMeta(type):
    _mixed = {}

    def __getitem__(cls, Module):
        assert isinstance(Module, type) and issubclass(Module, torch.nn.Module), Module
        if Module is torch.nn.Module:
            return PyroModule
        if issubclass(Module, PyroModule):
            return Module
        if Module not in cls._mixed:
            cls._mixed[Module] = type("Pyro" + Module.__name__, (Module, PyroModule),
                                      {"__module__": Module.__module__, "_pyro_mixed_from": Module})
        return cls._mixed[Module]


class PyroModule(torch.nn.Module, metaclass=_PyroModuleMeta):
    def __init__(self, *args, name="", **kwargs):
        self.__dict__.update(_pyro_name=name, _pyro_scope=_CallScope(), _pyro_params={}, _pyro_samples={})
        super().__init__(*args, **kwargs)

    # ---- naming: children learn their dotted name (and share the scope) when they are attached -----------
    def _pyro_adopt(self, name, scope):
        self.__dict__.update(_pyro_name=name, _pyro_scope=scope)
        for key, child in self._modules.items():
            if isinstance(child, PyroModule):
                child._pyro_adopt(_dotted(name, key), scope)

    def add_module(self, name, module):
        if isinstance(module, PyroModule):
            module._pyro_adopt(_dotted(self._pyro_name, name), self._pyro_scope)
        super().add_module(name, module)

    def __call__(self, *args, **kwargs):
        with self._pyro_scope:
            return super().__call__(*args, **kwargs)

    # ---- attribute protocol ---------------------------------------------------------------------------------
    @property
    def _pyro_context(self):
        return self._pyro_scope

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor) and not isinstance(value, torch.nn.Parameter) \
                and name in self.__dict__.get("_pyro_params", ()):
            # a new constrained value for a declared parameter: written through the constraint
            constraint, _ = self._pyro_params[name]
            leaf = torch.nn.Module.__getattr__(self, name + "_unconstrained")
            with torch.no_grad():
                leaf.data = transform_to(constraint).inv(value.detach()).clone()
            return
        if isinstance(value, PyroModule):
            value._pyro_adopt(_dotted(self._pyro_name, name), self._pyro_scope)
        elif isinstance(value, (PyroParam, PyroSample)) or (
                isinstance(value, torch.nn.Parameter) and name in self._pyro_params):
            self._pyro_forget(name)
            if isinstance(value, PyroSample):
                self._pyro_samples[name] = value.prior
            else:
                self._pyro_declare_param(name, value)
            return
        super().__setattr__(name, value)

    def _pyro_declare_param(self, name, value):
        if isinstance(value, torch.nn.Parameter):
            value = PyroParam(value.data, *self._pyro_params.get(name, (constraints.real, None)))
        init, constraint, event_dim = value
        if callable(init) and not isinstance(init, torch.Tensor):     # lazy initialiser
            init = init(self) if _wants_self(init) else init()
        with torch.no_grad():
            unconstrained = transform_to(constraint).inv(init.detach()).clone().contiguous()
        self._pyro_params[name] = (constraint, event_dim)
        super().__setattr__(name + "_unconstrained", torch.nn.Parameter(unconstrained))

    def _pyro_forget(self, name):
        self._pyro_samples.pop(name, None)
        if self._pyro_params.pop(name, None) is not None:
            super().__delattr__(name + "_unconstrained")
        elif name in self._parameters or name in self._modules or name in self._buffers \
                or name in self.__dict__:
            super().__delattr__(name)

    def __delattr__(self, name):
        if name in self._pyro_params or name in self._pyro_samples:
            self._pyro_forget(name)
        else:
            super().__delattr__(name)

    def __getattr__(self, name):
        state = self.__dict__
        scope = state.get("_pyro_scope")
        in_call = scope is not None and scope.active
        full = _dotted(state.get("_pyro_name", ""), name)
        if "_pyro_params" in state and name in state["_pyro_params"]:
            if in_call and not _local_params():
                return scope.recall(full, functools.partial(self._pyro_read_param, name))
            constraint, _ = state["_pyro_params"][name]
            leaf = torch.nn.Module.__getattr__(self, name + "_unconstrained")
            value = transform_to(constraint)(leaf)
            value.unconstrained = weakref.ref(leaf)
            return value
        if "_pyro_samples" in state and name in state["_pyro_samples"]:
            if in_call:
                return scope.recall(full, functools.partial(self._pyro_read_sample, name))
            prior = state["_pyro_samples"][name]
            if not isinstance(prior, torch.distributions.Distribution) and callable(prior):
                prior = prior(self)
            return prior if isinstance(prior, torch.Tensor) else prior()
        value = super().__getattr__(name)
        if isinstance(value, PyroModule) and scope is not None and \
                (value._pyro_scope is not scope or (not value._pyro_name and full)):
            value._pyro_adopt(full, scope)          # converted or attached behind our back: name it now
        if in_call and not _local_params():
            if isinstance(value, torch.nn.Parameter) and not name.endswith("_unconstrained"):
                return scope.recall(full, lambda: primitives.param(full, value))
            if isinstance(value, torch.nn.Module) and not isinstance(value, PyroModule):
                # a plain torch module inside a PyroModule: its parameters join as <name>$$$<param>
                return scope.recall(full, lambda: primitives.module(full, value))
        return value

    def _pyro_read_param(self, name):
        constraint, event_dim = self._pyro_params[name]
        full = _dotted(self._pyro_name, name)
        leaf = torch.nn.Module.__getattr__(self, name + "_unconstrained")
        if _PARAM_STORE._params.get(full) is not leaf:
            # the store's unconstrained leaf IS the module's parameter: optimisers see one tensor
            if full in _PARAM_STORE._params:
                _PARAM_STORE._param_to_name.pop(_PARAM_STORE._params[full], None)
            _PARAM_STORE._params[full] = leaf
            _PARAM_STORE._param_to_name[leaf] = full
            _PARAM_STORE._constraints[full] = constraint
            _PARAM_STORE.generation += 1
        return primitives.param(full, event_dim=event_dim)

    def _pyro_read_sample(self, name):
        prior = self._pyro_samples[name]
        if not isinstance(prior, torch.distributions.Distribution) and callable(prior):
            prior = prior(self)
        full = _dotted(self._pyro_name, name)
        if isinstance(prior, torch.Tensor):            # a function of other attributes: recorded, not scored
            return primitives.deterministic(full, prior, event_dim=0)
        return primitives.sample(full, prior)

    def named_pyro_params(self, prefix="", recurse=True):
        """(dotted name, constrained value) of every parameter, PyroParam or plain."""
        owners = self.named_modules(prefix=prefix) if recurse else [(prefix, self)]
        for module_prefix, module in owners:
            declared = getattr(module, "_pyro_params", {})
            for key in list(module._parameters):
                attr = key[:-len("_unconstrained")] if key.endswith("_unconstrained") \
                    and key[:-len("_unconstrained")] in declared else key
                yield _dotted(module_prefix, attr), getattr(module, attr)


def _wants_self(fn):
    import inspect
    try:
        return len(inspect.signature(fn).parameters) == 1
    except (TypeError, ValueError):
        return False


def pyro_method(fn):
    """Decorator for methods other than ``forward`` that read Pyro attributes: one evaluation per call."""
    @functools.wraps(fn)
    def scoped(self, *args, **kwargs):
        with self._pyro_scope:
            return fn(self, *args, **kwargs)

    return scoped


def clear(mod):
    """Remove the module tree's parameters from the global parameter store."""
    assert isinstance(mod, PyroModule)
    for name in [n for n, _ in mod.named_pyro_params()]:
        if name in _PARAM_STORE:
            del _PARAM_STORE[name]


def to_pyro_module_(m, recurse=True):
    """Turn an existing ``nn.Module`` (and, by default, its children) into PyroModules, in place."""
    if not isinstance(m, torch.nn.Module):
        raise TypeError("Expected an nn.Module instance but got a {}".format(type(m)))
    if not isinstance(m, PyroModule):
        m.__class__ = PyroModule[type(m)]
        m.__dict__.update(_pyro_name="", _pyro_scope=_CallScope(), _pyro_params={}, _pyro_samples={})
    if recurse:
        for name, child in list(m._modules.items()):
            if child is not None:
                to_pyro_module_(child)
                child._pyro_adopt(_dotted(m._pyro_name, name), m._pyro_scope)
