"""pyro.poutine.messenger: the reference's module path for these names (they live in handlers.py /
runtime.py / trace.py here)."""
from .runtime import Messenger, block_messengers  # noqa: F401


def unwrap(fn):
    """The callable underneath any number of handler wrappers."""
    from .runtime import _BoundHandler
    while isinstance(fn, _BoundHandler):
        fn = fn.fn
    return fn
