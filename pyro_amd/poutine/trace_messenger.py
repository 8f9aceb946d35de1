"""pyro.poutine.trace_messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .handlers import TraceMessenger, _TraceHandler as TraceHandler  # noqa: F401
