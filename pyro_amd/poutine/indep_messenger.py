"""pyro.poutine.indep_messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .handlers import CondIndepStackFrame, PlateMessenger as IndepMessenger  # noqa: F401
