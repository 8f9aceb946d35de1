"""pyro.poutine.infer_config_messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .handlers import InferConfigMessenger  # noqa: F401
