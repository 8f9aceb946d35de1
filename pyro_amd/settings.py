"""Global settings registry (the interface of pyro/settings.py: get / set / context / register).

A setting is an alias for a module-level constant or class attribute that lives where it is used;
the registry only knows where to find it.  ``with settings.context(alias=value): ...`` (also a
decorator) overrides temporarily.
"""
import functools
from contextlib import contextmanager
from importlib import import_module

_REGISTRY = {}      # alias -> (module name, dotted attribute path, validator or None)


def _owner_and_attr(alias):
    modulename, deepname, _ = _REGISTRY[alias]
    owner = import_module(modulename)
    *path, attr = deepname.split(".")
    for name in path:
        owner = getattr(owner, name)
    return owner, attr


def get(alias=None):
    """One setting, or all of them as a dict when ``alias`` is omitted."""
    if alias is None:
        return {a: get(a) for a in sorted(_REGISTRY)}
    owner, attr = _owner_and_attr(alias)
    return getattr(owner, attr)


def set(**kwargs):
    """``settings.set(alias=value, ...)``; each value goes through the setting's validator first."""
    for alias, value in kwargs.items():
        validator = _REGISTRY[alias][2]
        if validator is not None:
            validator(value)
        owner, attr = _owner_and_attr(alias)
        setattr(owner, attr, value)


@contextmanager
def context(**kwargs):
    saved = {alias: get(alias) for alias in kwargs}
    try:
        set(**kwargs)
        yield
    finally:
        set(**saved)


def register(alias, modulename, deepname, validator=None):
    """Declare a setting: ``register("my_setting", __name__, "MY_CONSTANT")``, or as a decorator on
    the function that validates new values."""
    _REGISTRY[alias] = (modulename, deepname, validator)
    if validator is not None:
        return validator
    return functools.partial(register, alias, modulename, deepname)


def _is_bool(value):
    assert isinstance(value, bool)


register("validate_distributions_pyro", "pyro_amd.distributions.util", "_VALIDATION_ENABLED", _is_bool)
register("validate_poutine", "pyro_amd.poutine.settings", "_VALIDATE", _is_bool)
register("validate_infer", "pyro_amd.infer.util", "_VALIDATION_ENABLED", _is_bool)


@register("validate_distributions_torch", "torch.distributions.distribution",
          "Distribution._validate_args")
def _validate_torch_flag(value):
    assert isinstance(value, bool)
