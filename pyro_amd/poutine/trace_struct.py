"""pyro.poutine.trace_struct: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .trace import Trace  # noqa: F401
