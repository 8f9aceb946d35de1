from .module import PyroModule, PyroParam, PyroSample, clear, pyro_method, to_pyro_module_  # noqa: F401
