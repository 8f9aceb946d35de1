"""pyro.poutine.escape_messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .handlers import EscapeMessenger  # noqa: F401
