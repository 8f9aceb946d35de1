"""pyro.poutine.reparam_messenger: the reference's module path."""
from .handlers import ReparamMessenger  # noqa: F401
