"""Global settings (the interface of pyro.settings: ``get`` / ``set`` / ``context`` / ``register``).

A setting is an ALIAS for a module-level constant or a class attribute that lives where it is used; this module
only keeps the address book.  ``with settings.context(alias=value): ...`` -- also usable as a decorator --
overrides temporarily; ``register`` doubles as a decorator for the function that validates new values.
"""
import contextlib
import importlib
import operator


class _Address:
    """Where a setting lives: ``module`` + dotted attribute path, and who checks new values."""

    __slots__ = ("module", "path", "validator")

    def __init__(self, module, path, validator):
        self.module, self.path, self.validator = module, path.split("."), validator

    def _owner(self):
        holder = importlib.import_module(self.module)
        return operator.attrgetter(".".join(self.path[:-1]))(holder) if len(self.path) > 1 else holder

    def read(self):
        return getattr(self._owner(), self.path[-1])

    def write(self, value):
        if self.validator is not None:
            self.validator(value)
        setattr(self._owner(), self.path[-1], value)


_BOOK = {}          # alias -> _Address


def get(alias=None):
    """One setting, or a dict of all of them (sorted by alias) when called without one."""
    if alias is not None:
        return _BOOK[alias].read()
    return {name: _BOOK[name].read() for name in sorted(_BOOK)}


def set(**values):
    """``settings.set(alias=value, ...)``; unknown aliases are a KeyError, rejected values whatever the
    setting's validator raises."""
    for alias, value in values.items():
        _BOOK[alias].write(value)


@contextlib.contextmanager
def context(**values):
    before = {alias: get(alias) for alias in values}
    set(**values)
    try:
        yield
    finally:
        set(**before)


def register(alias, modulename, deepname, validator=None):
    """``register("my_setting", __name__, "MY_CONSTANT")`` declares a setting; the returned callable accepts
    the validator, so that the same line works as a decorator::

        @register("my_setting", __name__, "MY_CONSTANT")
        def _check(value):
            assert value > 0
    """
    entry = _BOOK[alias] = _Address(modulename, deepname, validator)

    def with_validator(fn):
        entry.validator = fn
        return fn

    return validator if validator is not None else with_validator


def _must_be_bool(value):
    assert isinstance(value, bool), value


for _alias, _module, _name in (
        ("validate_distributions_pyro", "pyro_amd.distributions.util", "_VALIDATION_ENABLED"),
        ("validate_distributions_torch", "torch.distributions.distribution", "Distribution._validate_args"),
        ("validate_poutine", "pyro_amd.poutine.settings", "_VALIDATE"),
        ("validate_infer", "pyro_amd.infer.util", "_VALIDATION_ENABLED"),
        ("module_local_params", "pyro_amd.nn.module", "_MODULE_LOCAL_PARAMS")):
    register(_alias, _module, _name, _must_be_bool)
del _alias, _module, _name


def _auto_or_bool(value):
    assert value in ("auto", True, False), value


# what runs when nobody says: SVI(model, guide, optim, loss) captures its step / NUTS replays its rounds
# from a hipGraph when the arguments live on the GPU ("auto"), always (True), or only when asked (False)
register("svi_capture_steps", "pyro_amd.infer.svi", "CAPTURE_STEPS", _auto_or_bool)
register("mcmc_capture_rounds", "pyro_amd.infer.mcmc.nuts", "CAPTURE_ROUNDS", _auto_or_bool)
