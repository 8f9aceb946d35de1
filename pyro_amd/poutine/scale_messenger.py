"""pyro.poutine.scale_messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .handlers import ScaleMessenger  # noqa: F401
