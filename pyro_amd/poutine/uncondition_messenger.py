"""pyro.poutine.uncondition_messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .handlers import UnconditionMessenger  # noqa: F401
