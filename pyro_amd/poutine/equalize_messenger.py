"""pyro.poutine.equalize_messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .handlers import EqualizeMessenger  # noqa: F401
